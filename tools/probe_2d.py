"""GPU probe: per-step time of the 2-D forward loop, saved vs rotating buffers, by order and batch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import mifwi
from helpers import Case

flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for so in (8, 16):
    for ns in (8, 32):
        case = Case((340, 140), (30., 30.), nbl=40, space_order=so, nt=600, nshots=ns, nrec=340, seed=1)
        plan = case.plan(mifwi, flags=flags)
        out = []
        for save in (False, True):
            plan.forward(case.wavelet, save=save, want_rec=False)
            t = []
            for _ in range(3):
                plan.forward(case.wavelet, save=save, want_rec=False)
                st = plan.stats()
                t.append(st["fwd_ms"] * 1e3 / st["fwd_launches"])
            out.append(min(t))
        rec = plan.forward(case.wavelet, save=True)
        g = []
        for _ in range(3):
            plan.gradient(rec, want_illum=True)
            st = plan.stats()
            g.append(st["bwd_ms"] * 1e3 / st["bwd_launches"])
        pts = 420 * 220 * ns
        print("so=%2d shots=%2d  fwd nosave %.2f us  save %.2f us  grad %.2f us   (fwd %.0f Gpts/s, %.0f GB/s alg)" %
              (so, ns, out[0], out[1], min(g), pts / out[1] / 1e3, 20 * pts / out[1] / 1e3), flush=True)
        plan.close()
