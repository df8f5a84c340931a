"""cProfile of forward-only fwi.fwi_loss evaluations on the bench workload (host-side overheads)."""
import cProfile, pstats, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
import bench
from seismic import AcquisitionGeometry
from fwi import fm_multi, fwi_loss
from misfit import least_square
true_model, init_model = bench.marmousi2(16)
src_all, rec = bench.acquisition(true_model, 8)
g_true = AcquisitionGeometry(true_model, rec, src_all, 0., 4500., f0=0.007, src_type='Ricker')
g_init = AcquisitionGeometry(init_model, rec, src_all, 0., 4500., f0=0.007, src_type='Ricker')
obs = fm_multi(g_true)
nbl = init_model.nbl
m0 = 1.0 / np.asarray(init_model.vp.data[nbl:-nbl, nbl:-nbl], dtype=np.float64).reshape(-1) ** 2
for _ in range(2):
    fwi_loss(m0, g_init, obs, least_square, None, None, True, calc_grad=False)
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    fwi_loss(m0, g_init, obs, least_square, None, None, True, calc_grad=False)
pr.disable()
print("per evaluation: %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue())
