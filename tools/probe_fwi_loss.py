"""GPU probe: wall time of fwi_loss (gradient evaluation and forward-only line-search trial) through the
drop-in Python interface, Marmousi-II, 8 shots, vs the GPU time inside libmifwi."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
from seismic import Model, AcquisitionGeometry
from fwi import fm_multi, fwi_loss
from misfit import least_square, qWasserstein

so = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nshots = 8
d = os.path.join(ROOT, "tests", "golden", "model_data", "SMARM2")
shape = (340, 140)
true_vp = np.fromfile(os.path.join(d, "vp.true"), dtype=np.float32).reshape(shape) / 1000
smooth_vp = np.fromfile(os.path.join(d, "vp.smooth_20"), dtype=np.float32).reshape(shape) / 1000
kw = dict(origin=(0, 0), spacing=(30., 30.), shape=shape, space_order=so, nbl=40, fs=False, dt=3.0)
true_model, init_model = Model(vp=true_vp, **kw), Model(vp=smooth_vp, **kw)
const_model = Model(vp=np.ones(shape) * 1.5, **kw)
src = np.empty((nshots, 2)); src[:, 0] = np.linspace(0, true_model.domain_size[0], nshots); src[:, 1] = 60.
rec = np.empty((shape[0], 2)); rec[:, 0] = np.linspace(30., true_model.domain_size[0] - 30., shape[0]); rec[:, 1] = 60.
g1 = AcquisitionGeometry(true_model, rec, src, 0., 4500., f0=0.007, src_type='Ricker')
g0 = AcquisitionGeometry(init_model, rec, src, 0., 4500., f0=0.007, src_type='Ricker')
g2 = AcquisitionGeometry(const_model, rec, src, 0., 4500., f0=0.007, src_type='Ricker')
mask = np.ones(shape, dtype=np.float32); mask[:, :15] = 0
t = time.perf_counter(); obs = fm_multi(g1); t_obs = time.perf_counter() - t
direct = fm_multi(g2)
m0 = 1. / (smooth_vp.reshape(-1).astype(np.float64)) ** 2
for dw, label in ((None, "no direct wave"), (direct, "with direct wave")):
    fwi_loss(m0, g0, obs, least_square, dw, mask, True)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); f, g, r = fwi_loss(m0, g0, obs, least_square, dw, mask, True); ts.append(time.perf_counter() - t)
    tf = []
    for _ in range(3):
        t = time.perf_counter(); f2, _, _ = fwi_loss(m0, g0, obs, least_square, dw, mask, True, calc_grad=False); tf.append(time.perf_counter() - t)
    print("%-18s fwi_loss grad %.1f ms, forward-only %.1f ms (fm_multi first call %.0f ms)  fval %.6e |g| %.4e" %
          (label, min(ts) * 1e3, min(tf) * 1e3, t_obs * 1e3, f, np.linalg.norm(g)), flush=True)

w2 = qWasserstein(gamma=1.01, method='1d')
fwi_loss(m0, g0, obs, w2, None, mask, True)
ts = []
for _ in range(2):
    t = time.perf_counter(); f, g, r = fwi_loss(m0, g0, obs, w2, None, mask, True); ts.append(time.perf_counter() - t)
t = time.perf_counter()
for k in range(nshots):
    w2(np.asarray(r[k]), obs[k].data)
tm = time.perf_counter() - t
print("trace-wise W2 misfit: fwi_loss grad %.1f ms, of which the misfit calls themselves ~%.1f ms   fval %.6e" % (min(ts) * 1e3, tm * 1e3, f))
