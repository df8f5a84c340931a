"""GPU probe: tiled TTI forward step time only (2-D, 8 shots), for kernel experiments."""
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
    plan = mifwi.Plan(prob.shape_pad, (10., 10.), prob.origin_pad, so, nt, dt, 1, 300, max_shots=nshots)
    plan.set_model(prob.vp, prob.damp)
    plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
    plan.set_geometry(src, rec)
    bf = ba = 1e9
    for _ in range(4):
        r = plan.tti_forward(w)
        bf = min(bf, plan.stats()["fwd_ms"] / (nt - 2))
        plan.tti_adjoint(r)
        ba = min(ba, plan.stats()["bwd_ms"] / (nt - 2))
    print("%s so=%2d forward %.2f adjoint %.2f us/step  |rec| %.6e" % (os.environ.get("MIFWI_LIB", "default")[-12:], so, bf * 1e3, ba * 1e3, float(np.linalg.norm(r))), flush=True)
    plan.close()
