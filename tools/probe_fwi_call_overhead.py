"""GPU probe: where the time of fwi.fwi_loss goes beyond the two time loops -- wall time of each library call
against the loop times the library reports for it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
import numpy as np, torch, bench
import mifwi
from mifwi import _lib
rows = []
orig = _lib.Plan.l2_misfit_gradient
def timed(self, *a, **kw):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = orig(self, *a, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    st = self.stats()
    rows.append((bool(kw.get("calc_grad", True)), dt, st["fwd_ms"], st["bwd_ms"] if kw.get("calc_grad", True) else 0.0))
    return out
_lib.Plan.l2_misfit_gradient = timed
orig_set = _lib.Plan.set_model
set_ms = []
def timed_set(self, *a, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig_set(self, *a, **kw)
    torch.cuda.synchronize(); set_ms.append((time.perf_counter() - t0) * 1e3)
    return r
_lib.Plan.set_model = timed_set
true_model, init_model = bench.marmousi2(16)
src_all, rec = bench.acquisition(true_model, 8)
ms = bench.fwi_iteration(true_model, init_model, src_all, rec, reps=3)[0]
rows = rows[-12:]
for grad in (True, False):
    sel = [r for r in rows if r[0] == grad]
    print("calc_grad=%s: call %.2f ms  forward loop %.2f  backward loop %.2f  -> %.2f ms outside the loops (n=%d)" %
          (grad, np.mean([r[1] for r in sel]), np.mean([r[2] for r in sel]), np.mean([r[3] for r in sel]),
           np.mean([r[1] - r[2] - r[3] for r in sel]), len(sel)))
print("set_model: %.3f ms per call" % np.mean(set_ms[-12:]), " fwi_iteration (with the extra synchronisations): %.2f ms" % ms)
