"""GPU probe: calls that use fewer shots than the plan holds (nshots < max_shots) give the same per-shot
results as a plan sized exactly, for the acoustic and the TTI entry points."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import mifwi
from helpers import Case, rel_l2, smooth_random_vp
from oracle import oracle as orc

for shape, so in (((70, 52), 8), ((24, 26, 30), 4)):
    nd = len(shape)
    case = Case(shape, (10.,) * nd, 8, so, nt=70, nshots=4, nrec=9, seed=3, f0=0.03)
    res = np.random.default_rng(0).standard_normal((4, 70, 9)).astype(np.float32)
    for flags in (0, mifwi.FLAG_SAVE_F16, mifwi.FLAG_CHECKPOINT):
        big = case.plan(mifwi, flags=flags)                       # holds 4 shots
        rec4 = big.forward(case.wavelet, save=True); g4 = big.gradient(res)
        big.set_geometry(case.src[:2], case.rec)
        rec2 = big.forward(case.wavelet, nshots=2, save=True); g2 = big.gradient(res[:2], nshots=2)
        a2 = big.adjoint(res[:2], nshots=2)
        out = big.l2_misfit_gradient(case.wavelet, rec4[:2] * 0.9, nshots=2)
        assert np.array_equal(rec2, rec4[:2]) and np.array_equal(g2, g4[:2]), (shape, flags)
        assert out["grad"].shape[0] == 2 and np.isfinite(out["grad"]).all() and a2.shape[0] == 2
        big.close()
    if so % 4 == 0:
        vp = smooth_random_vp(shape, seed=3)
        prob = orc.TTIProblem(vp, (10.,) * nd, 8, so, .3 * (vp - 1.5), .2 * (vp - 1.5), .5 * (vp - 1.5),
                              .25 * (vp - 1.5) if nd == 3 else None, dtype=np.float32)
        dt = float(prob.critical_dt)
        w = orc.ricker(0.03, dt * np.arange(70)).reshape(70, 1).astype(np.float32)
        plan = mifwi.Plan(prob.shape_pad, prob.spacing, prob.origin_pad, so, 70, dt, 1, 9, max_shots=4)
        plan.set_model(prob.vp, prob.damp); plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
        plan.set_geometry(case.src, case.rec)
        d4 = plan.tti_forward(w, save=True); g4 = plan.tti_gradient(res)
        dm = np.zeros(prob.shape_pad, np.float32); dm[tuple(slice(10, -10) for _ in shape)] = 0.01
        b4 = plan.tti_born(w, dm)
        d2 = plan.tti_forward(w, nshots=2, save=True); g2 = plan.tti_gradient(res[:2], nshots=2)
        b2 = plan.tti_born(w, dm, nshots=2)
        a2 = plan.tti_adjoint(res[:2], nshots=2)
        assert np.array_equal(d2, d4[:2]) and np.array_equal(g2, g4[:2]) and np.array_equal(b2, b4[:2]), shape
        assert np.isfinite(a2).all()
        plan.close()
    print("ok", shape, so)
