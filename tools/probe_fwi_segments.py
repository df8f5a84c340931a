"""GPU probe: wall time of the pieces of fwi.fwi_loss (device synchronised around each), per gradient call and per
forward-only call of bench.fwi_iteration."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
import numpy as np, torch, bench
import fwi
from mifwi import _lib

acc = collections.defaultdict(list)
state = {"grad": True}


def wrap(obj, name, label):
    orig = getattr(obj, name)

    def timed(*a, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = orig(*a, **kw)
        torch.cuda.synchronize()
        acc[(label, state["grad"])].append((time.perf_counter() - t0) * 1e3)
        return r
    setattr(obj, name, timed)


orig_loss = fwi.fwi_loss


def loss(*a, **kw):
    state["grad"] = kw.get("calc_grad", True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = orig_loss(*a, **kw)
    torch.cuda.synchronize()
    acc[("fwi_loss", state["grad"])].append((time.perf_counter() - t0) * 1e3)
    return r


wrap(fwi, "_batch_entry", "_batch_entry (set_model inside)")
wrap(fwi, "_resident_batch", "_resident_batch")
wrap(fwi, "_wavelet", "_wavelet")
wrap(fwi, "_compute_batch", "_compute_batch (all of the above + library call)")
wrap(fwi, "fwi_obj_multi", "fwi_obj_multi")
wrap(_lib.Plan, "l2_misfit_gradient", "library call")
wrap(_lib.Plan, "set_model", "set_model")
import seismic.model as _sm
wrap(_sm.SeismicModel, "update", "model.update")
fwi.fwi_loss = loss
true_model, init_model = bench.marmousi2(16)
src_all, rec = bench.acquisition(true_model, 8)
ms = bench.fwi_iteration(true_model, init_model, src_all, rec, reps=5)[0]
print("fwi_iteration with the probes' synchronisations: %.2f ms" % ms)
for grad in (True, False):
    print("calc_grad = %s" % grad)
    for (label, g), v in sorted(acc.items()):
        if g == grad:
            print("   %-55s %.3f ms (median of the last %d)" % (label, float(np.median(v[-4:])), len(v[-4:])))
