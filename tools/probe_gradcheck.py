"""Directional-derivative check of fwi.fwi_loss on the Marmousi setup (GPU)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'devito-fwi_amd'))
from seismic import Model, AcquisitionGeometry
from fwi import fm_multi, fwi_loss
from misfit import least_square
from scipy.ndimage import gaussian_filter

shape, spacing, origin = (300, 106), (30., 30.), (0, 0)
data = os.path.join(ROOT, 'tests', 'golden', 'model_data', 'SMARMN')
true_vp = np.fromfile(os.path.join(data, 'vp.true'), dtype=np.float32).reshape(shape) / 1000
smooth_vp = np.fromfile(os.path.join(data, 'vp.smooth_20'), dtype=np.float32).reshape(shape) / 1000
mk = lambda vp: Model(origin=origin, spacing=spacing, shape=shape, space_order=8, vp=vp, nbl=40, fs=False, dt=2.95)
nsrc = 5
src = np.empty((nsrc, 2)); src[:, 0] = np.linspace(0, 8970, nsrc); src[:, 1] = 60
rec = np.empty((300, 2)); rec[:, 0] = np.linspace(30, 8940, 300); rec[:, 1] = 60
g1 = AcquisitionGeometry(mk(true_vp), rec, src, 0., 3000., f0=0.007, src_type='Ricker')
g0 = AcquisitionGeometry(mk(smooth_vp), rec, src, 0., 3000., f0=0.007, src_type='Ricker')
g2 = AcquisitionGeometry(mk(np.ones(shape) * 1.5), rec, src, 0., 3000., f0=0.007, src_type='Ricker')
obs = fm_multi(g1); dw = fm_multi(g2)
m0 = 1. / smooth_vp.reshape(-1).astype(np.float64) ** 2
rng = np.random.default_rng(0)
for use_dw in (None, dw):
    f0, g, _ = fwi_loss(m0, g0, obs, least_square, use_dw, None, False)
    for trial in range(2):
        p = gaussian_filter(rng.standard_normal(shape), 4).reshape(-1)
        p[np.arange(p.size) % 106 < 10] = 0      # keep away from sources/receivers
        p *= 0.01 * np.abs(m0).max() / np.abs(p).max()
        for eps in (1.0, 0.1):
            fp = fwi_loss(m0 + eps * p, g0, obs, least_square, use_dw, None, False, calc_grad=False)[0]
            fm = fwi_loss(m0 - eps * p, g0, obs, least_square, use_dw, None, False, calc_grad=False)[0]
            print('dw' if use_dw else 'no-dw', trial, eps, 'fd', (fp - fm) / (2 * eps), 'g.p', g @ p, 'f0', f0)
