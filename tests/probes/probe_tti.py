"""GPU probe: per-step time of the TTI propagators (forward and adjoint) on a Marmousi-II-sized 2-D grid."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import mifwi
from helpers import smooth_random_vp
from oracle import oracle as orc

nshots, nt = int(os.environ.get("NSHOTS", 8)), 600
shape = (340, 140)
vp = smooth_random_vp(shape, seed=2, passes=2).astype(np.float32)
rng = np.random.default_rng(3)
dom = np.array([(n - 1) * 10. for n in shape])
src = ((rng.random((nshots, 1, 2)) * 0.5 + 0.25) * dom).astype(np.float32)
rec = ((rng.random((300, 2)) * 0.8 + 0.1) * dom).astype(np.float32)
for so in (8, 16):
    eps, delta, theta = .3 * (vp - 1.5), .2 * (vp - 1.5), .5 * (vp - 1.5)
    prob = orc.TTIProblem(vp, (10., 10.), 40, so, eps, delta, theta, None)
    dt = float(prob.critical_dt)
    w = orc.ricker(0.01, dt * np.arange(nt)).astype(np.float32).reshape(nt, 1)
    for flags, name in ((0, "tiled"), (mifwi.FLAG_GENERIC, "untiled")):
        plan = mifwi.Plan(prob.shape_pad, (10., 10.), prob.origin_pad, so, nt, dt, 1, 300, max_shots=nshots, flags=flags)
        plan.set_model(prob.vp, prob.damp)
        plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
        plan.set_geometry(src, rec)
        ng = float(np.prod(prob.shape_pad)) * nshots
        bf = ba = bs = bb = bg = 1e9
        dm = np.zeros(prob.shape_pad, np.float32)
        dm[60:-60, 60:-60] = 0.01
        for _ in range(3):
            r = plan.tti_forward(w)
            bf = min(bf, plan.stats()["fwd_ms"] / (nt - 2))
            plan.tti_adjoint(r)
            ba = min(ba, plan.stats()["bwd_ms"] / (nt - 2))
            plan.tti_forward(w, save=True)
            bs = min(bs, plan.stats()["fwd_ms"] / (nt - 2))
            plan.tti_gradient(r)
            bg = min(bg, plan.stats()["bwd_ms"] / (nt - 2))
            plan.tti_born(w, dm)
            bb = min(bb, plan.stats()["fwd_ms"] / (nt - 2))
        print("TTI 2-D %dx%d x %d shots so=%2d %-8s forward %.1f us/step (%.1f Gpts/s)  adjoint %.1f  forward(save) %.1f  "
              "gradient %.1f  born %.1f us/step" %
              (prob.shape_pad + (nshots, so, name, bf * 1e3, ng / bf / 1e6, ba * 1e3, bs * 1e3, bg * 1e3, bb * 1e3)),
              flush=True)
        plan.close()
