"""GPU probe: the gradient loop that updates its accumulators every second step (MIFWI_PAIR_GRAD=1, the
default of the tiled kernels) against the one-update-per-step loop, with the full save and with
checkpointing.  Expected: full and checkpointed runs agree bit for bit within each schedule (asserted);
the two schedules differ in the gradient at rounding level only (the plain and the gradient variant of
the step kernel round the update differently), and not at all in the illumination."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import mifwi
from helpers import Case

def diff(a, b):
    n = int((a != b).sum())
    return "%d differ, max |d| %.3e of %.3e" % (n, float(np.abs(a - b).max()), float(np.abs(a).max()))

for shape, so, nt in (((50, 44), 8, 131), ((100, 70), 16, 150), ((100, 70), 16, 151), ((37, 90), 12, 64)):
    case = Case(shape, (10.,) * 2, 8, so, nt=nt, nshots=3, nrec=11, seed=5, f0=0.03)
    res = np.random.default_rng(0).standard_normal((3, nt, 11)).astype(np.float32)
    out = {}
    for pair in (0, 1):
        for flags in (0, mifwi.FLAG_CHECKPOINT):
            os.environ["MIFWI_PAIR_GRAD"] = str(pair)
            plan = case.plan(mifwi, flags=flags)
            plan.forward(case.wavelet, save=True)
            g, il = plan.gradient(res, want_illum=True)
            v = plan.read_wavefield(which=2, shot=1)
            out[(pair, flags)] = (g, il, v)
            plan.close()
    for pair in (0, 1):
        for i in range(3):
            assert np.array_equal(out[(pair, 0)][i], out[(pair, mifwi.FLAG_CHECKPOINT)][i]), (shape, pair, i)
    assert np.abs(out[(1, 0)][0] - out[(0, 0)][0]).max() <= 1e-5 * np.abs(out[(0, 0)][0]).max()
    assert np.array_equal(out[(1, 0)][1], out[(0, 0)][1])
    ref = out[(0, 0)]
    for key, val in out.items():
        print(shape, so, nt, "pair=%d flags=%d" % key, "| grad:", diff(ref[0], val[0]), "| illum:", diff(ref[1], val[1]),
              "| v:", diff(ref[2], val[2]))
