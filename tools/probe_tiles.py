import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import mifwi
from helpers import Case
for shape in ((340, 140), (336, 140), (304, 140), (340, 108)):
    case = Case(shape, (25., 25.), nbl=40, space_order=16, nt=600, nshots=8, nrec=340, seed=1)
    plan = case.plan(mifwi)
    nsteps = case.nt - 2
    t = []
    for _ in range(5):
        plan.forward(case.wavelet, save=False, want_rec=False)
        t.append(plan.stats()["fwd_ms"] * 1e3 / nsteps)
    sp = case.prob.shape_pad
    print(shape, "padded", sp, "tiles", -(-sp[0] // 32) * -(-sp[1] // 32), "fwd %.2f us/step" % min(t), flush=True)
    plan.close()
