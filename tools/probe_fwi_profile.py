"""GPU probe: cProfile of fwi.fwi_loss on the bench's Marmousi-II survey, forward-only and gradient calls apart
(host functions by own time; the library call itself shows as the ctypes entry)."""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
import numpy as np, torch, bench
from seismic import AcquisitionGeometry
from fwi import fm_multi, fwi_loss
from misfit import least_square
from copy import deepcopy

true_model, init_model = bench.marmousi2(16)
src_all, rec = bench.acquisition(true_model, 8)
init_model = deepcopy(init_model)
g_true = AcquisitionGeometry(true_model, rec, src_all, 0., 4500., f0=0.007, src_type='Ricker')
g_init = AcquisitionGeometry(init_model, rec, src_all, 0., 4500., f0=0.007, src_type='Ricker')
obs = fm_multi(g_true)
mask = np.ones(init_model.shape, dtype=np.float32)
mask[:, :15] = 0
nbl = init_model.nbl
m0 = 1.0 / np.asarray(init_model.vp.data[nbl:-nbl, nbl:-nbl], dtype=np.float64).reshape(-1) ** 2
for grad in (True, False):
    for _ in range(3):
        fwi_loss(m0, g_init, obs, least_square, None, mask, True, calc_grad=grad)
    pr = cProfile.Profile()
    n = 40
    pr.enable()
    for j in range(n):
        fwi_loss(m0 * (1 + 1e-4 * j), g_init, obs, least_square, None, mask, True, calc_grad=grad)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print("==== calc_grad=%s, %d calls (divide by %d) ====" % (grad, n, n))
    print(s.getvalue())
