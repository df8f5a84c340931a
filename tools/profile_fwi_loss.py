"""cProfile of fwi.fwi_loss on the benchmark survey (Marmousi-II, 8 shots): where the host time of one
objective evaluation goes beyond the GPU time loops."""
import cProfile, pstats, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
import numpy as np
import torch
import bench
from copy import deepcopy
from seismic import AcquisitionGeometry
from fwi import fm_multi, fwi_loss
from misfit import least_square
true_model, init_model = bench.marmousi2(16)
src_all, rec = bench.acquisition(true_model, 8)
init_model = deepcopy(init_model)
g_true = AcquisitionGeometry(true_model, rec, src_all, 0., 4500., f0=0.007, src_type='Ricker')
g_init = AcquisitionGeometry(init_model, rec, src_all, 0., 4500., f0=0.007, src_type='Ricker')
obs = fm_multi(g_true)
mask = np.ones(init_model.shape, dtype=np.float32); mask[:, :15] = 0
m0 = 1.0 / np.asarray(init_model.vp.data[40:-40, 40:-40], dtype=np.float64).reshape(-1) ** 2
for _ in range(2):
    fwi_loss(m0, g_init, obs, least_square, None, mask, True)
    fwi_loss(m0, g_init, obs, least_square, None, mask, True, calc_grad=False)
for grad in (True, False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        fwi_loss(m0, g_init, obs, least_square, None, mask, True, calc_grad=grad)
    torch.cuda.synchronize()
    print("calc_grad=%s: %.2f ms per call" % (grad, (time.perf_counter() - t0) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    fwi_loss(m0, g_init, obs, least_square, None, mask, True)
    fwi_loss(m0, g_init, obs, least_square, None, mask, True, calc_grad=False)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue())
