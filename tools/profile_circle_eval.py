"""GPU probe: host profile of one objective + gradient evaluation of the circle workflow (resampled traces)."""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
from seismic import demo_model, AcquisitionGeometry
from fwi import fm_multi, fwi_loss
from misfit import least_square

shape, spacing, nbl, so = (201, 201), (10., 10.), 40, 6
kw = dict(origin=(0, 0), shape=shape, spacing=spacing, space_order=so, nbl=nbl, dt=1.)
true_model = demo_model('circle-isotropic', vp_circle=3.6, vp_background=3, r=60, **kw)
init_model = demo_model('circle-isotropic', vp_circle=3, vp_background=3, r=60, **kw)
nsrc = 11
src = np.empty((nsrc, 2)); src[:, 1] = np.linspace(0, true_model.domain_size[0], nsrc); src[:, 0] = 20.
rec = np.empty((shape[0], 2)); rec[:, 1] = np.linspace(10., true_model.domain_size[0] - 10., shape[0]); rec[:, 0] = 1980.
g1 = AcquisitionGeometry(true_model, rec, src, 0., 1000., f0=0.010, src_type='Ricker')
g0 = AcquisitionGeometry(init_model, rec, src, 0., 1000., f0=0.010, src_type='Ricker')
g1.resample(1.); g0.resample(1.)
obs = fm_multi(g1, save=False)
v0 = init_model.vp.data[nbl:-nbl, nbl:-nbl]
m0 = 1.0 / (v0.reshape(-1).astype(np.float64)) ** 2
mask = np.ones(shape, np.float32)
fwi_loss(m0, g0, obs, least_square, None, mask, True)
t = time.perf_counter(); fwi_loss(m0, g0, obs, least_square, None, mask, True); print("gradient evaluation %.1f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter(); fwi_loss(m0, g0, obs, least_square, None, mask, True, calc_grad=False); print("forward-only evaluation %.1f ms" % ((time.perf_counter() - t) * 1e3))
pr = cProfile.Profile(); pr.enable()
fwi_loss(m0, g0, obs, least_square, None, mask, True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
