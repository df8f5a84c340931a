"""GPU probe: per-step time of the 2-D acoustic and TTI kernels for small shot batches (Marmousi-II-sized grid),
to compare tile shapes (MIFWI_2D_TX environment hook)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import mifwi
from helpers import smooth_random_vp
from oracle import oracle as orc
nt = 600
shape = (340, 140)
vp = smooth_random_vp(shape, seed=2, passes=2).astype(np.float32)
rng = np.random.default_rng(3)
dom = np.array([(n - 1) * 10. for n in shape])
for nshots in (1, 2, 4, 8):
    src = ((rng.random((nshots, 1, 2)) * 0.5 + 0.25) * dom).astype(np.float32)
    rec = ((rng.random((300, 2)) * 0.8 + 0.1) * dom).astype(np.float32)
    line = "TX=%s shots=%d:" % (os.environ.get("MIFWI_2D_TX", "auto"), nshots)
    for so in (8, 16):
        prob = orc.Problem(vp, (10., 10.), 40, so)
        dt = float(prob.critical_dt)
        w = orc.ricker(0.01, dt * np.arange(nt)).astype(np.float32).reshape(nt, 1)
        plan = mifwi.Plan(prob.shape_pad, (10., 10.), prob.origin_pad, so, nt, dt, 1, 300, max_shots=nshots)
        plan.set_model(prob.vp, prob.damp); plan.set_geometry(src, rec)
        bf = bg = 1e9
        for _ in range(3):
            d = plan.forward(w, save=True); bf = min(bf, plan.stats()["fwd_ms"] / (nt - 2))
            plan.gradient(d, want_illum=True); bg = min(bg, plan.stats()["bwd_ms"] / (nt - 2))
        plan.close()
        line += "  so=%d acoustic fwd(save) %.2f grad %.2f" % (so, bf * 1e3, bg * 1e3)
    so = 8
    eps, delta, theta = .3 * (vp - 1.5), .2 * (vp - 1.5), .5 * (vp - 1.5)
    prob = orc.TTIProblem(vp, (10., 10.), 40, so, eps, delta, theta, None)
    dt = float(prob.critical_dt)
    w = orc.ricker(0.01, dt * np.arange(nt)).astype(np.float32).reshape(nt, 1)
    plan = mifwi.Plan(prob.shape_pad, (10., 10.), prob.origin_pad, so, nt, dt, 1, 300, max_shots=nshots)
    plan.set_model(prob.vp, prob.damp); plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
    plan.set_geometry(src, rec)
    bf = bg = 1e9
    for _ in range(3):
        d = plan.tti_forward(w, save=True); bf = min(bf, plan.stats()["fwd_ms"] / (nt - 2))
        plan.tti_gradient(d); bg = min(bg, plan.stats()["bwd_ms"] / (nt - 2))
    plan.close()
    print(line + "  | TTI so=8 fwd(save) %.2f grad %.2f us/step" % (bf * 1e3, bg * 1e3), flush=True)
