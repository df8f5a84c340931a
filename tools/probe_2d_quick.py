"""GPU probe: per-step time of the 2-D forward (no save / save) and gradient loops at the bench size."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import mifwi
from helpers import Case
so = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nxp, nzp = [int(v) for v in os.environ.get("PROBE_SHAPE", "340,140").split(",")]   # physical grid; + 2 * 40 of sponge
case = Case((nxp, nzp), (25., 25.), nbl=40, space_order=so, nt=int(os.environ.get("PROBE_NT", "600")), nshots=ns, nrec=340, seed=1)
plan = case.plan(mifwi)
nsteps = case.nt - 2   # per time step, whatever the number of launches
out = []
for save in (False, True):
    plan.forward(case.wavelet, save=save, want_rec=False)
    t = []
    for _ in range(4):
        plan.forward(case.wavelet, save=save, want_rec=False)
        st = plan.stats()
        t.append(st["fwd_ms"] * 1e3 / nsteps)
    out.append(min(t))
rec = plan.forward(case.wavelet, save=True)
g, g2, ad = [], [], []
for _ in range(4):
    plan.gradient(rec, want_illum=True)
    st = plan.stats()
    g.append(st["bwd_ms"] * 1e3 / nsteps)
    plan.gradient(rec, want_illum=False)
    st = plan.stats()
    g2.append(st["bwd_ms"] * 1e3 / nsteps)
    plan.adjoint(rec)
    st = plan.stats()
    ad.append(st["bwd_ms"] * 1e3 / nsteps)
print("%s grid %dx%d so=%2d shots=%2d  fwd nosave %.2f us  save %.2f us  grad %.2f us  grad-noillum %.2f us  adjoint %.2f us" %
      (os.path.basename(os.environ.get("MIFWI_LIB", "libmifwi.so")), nxp + 80, nzp + 80, so, ns, out[0], out[1], min(g), min(g2), min(ad)),
      flush=True)
