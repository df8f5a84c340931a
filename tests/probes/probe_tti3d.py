"""GPU probe: per-step time of the 3-D float32 TTI forward loop (256^3 + 2 x 40 of sponge by default, one shot)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import mifwi
from helpers import smooth_random_vp
from oracle import oracle as orc
so = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(os.environ.get("PROBE_N", "256"))
nt = 24
shape = (n, n, n)
vp = smooth_random_vp(shape, seed=2, passes=2).astype(np.float32)
spacing = (10.,) * 3
rng = np.random.default_rng(3)
dom = np.array([(v - 1) * 10. for v in shape])
src = ((rng.random((1, 1, 3)) * 0.5 + 0.25) * dom).astype(np.float32)
rec = ((rng.random((300, 3)) * 0.8 + 0.1) * dom).astype(np.float32)
eps, delta, theta, phi = .3 * (vp - 1.5), .2 * (vp - 1.5), .5 * (vp - 1.5), .25 * (vp - 1.5)
prob = orc.TTIProblem(vp, spacing, 40, so, eps, delta, theta, phi)
dt = float(prob.critical_dt)
w = orc.ricker(0.01, dt * np.arange(nt)).astype(np.float32).reshape(nt, 1)
plan = mifwi.Plan(prob.shape_pad, spacing, prob.origin_pad, so, nt, dt, 1, 300, max_shots=1)
plan.set_model(prob.vp, prob.damp)
plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
plan.set_geometry(src, rec)
best = 1e9
for _ in range(4):
    plan.tti_forward(w)
    best = min(best, plan.stats()["fwd_ms"] / (nt - 2))
ng = float(np.prod(prob.shape_pad))
if os.environ.get("PROBE_GRAD"):   # adjoint, saving forward, gradient and Born loops as well
    d = plan.tti_forward(w)
    t = {"adjoint": 1e9, "forward(save)": 1e9, "gradient": 1e9, "born": 1e9}
    dm = np.zeros(prob.shape_pad, np.float32)
    dm[60:90, 60:90, 60:90] = 0.01
    for _ in range(3):
        plan.tti_adjoint(d)
        t["adjoint"] = min(t["adjoint"], plan.stats()["bwd_ms"] / (nt - 2))
        plan.tti_forward(w, save=True)
        t["forward(save)"] = min(t["forward(save)"], plan.stats()["fwd_ms"] / (nt - 2))
        plan.tti_gradient(d)
        t["gradient"] = min(t["gradient"], plan.stats()["bwd_ms"] / (nt - 2))
        plan.tti_born(w, dm)
        t["born"] = min(t["born"], plan.stats()["fwd_ms"] / (nt - 2))
    print("  " + "  ".join("%s %.1f us/step" % (k, v * 1e3) for k, v in t.items()), flush=True)
print("%s TTI 3-D %d^3 so=%d: %.1f us/step = %.1f Gpts/s" % (os.path.basename(os.environ.get("MIFWI_LIB", "libmifwi.so")), n + 80, so, best * 1e3, ng / best / 1e6), flush=True)
