"""GPU probe: throughput of the solver variants beyond the plain OT2 propagators (free surface, Born,
OT4, TTI) on a Marmousi-II-sized 2-D grid (8 shots) and a 256^3 + sponge 3-D grid (1 shot)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import mifwi
from helpers import smooth_random_vp
from oracle import oracle as orc


def run(shape, so, nshots, nt, label):
    nd = len(shape)
    vp = smooth_random_vp(shape, seed=2, passes=2).astype(np.float32)
    spacing = (10.,) * nd
    rng = np.random.default_rng(3)
    dom = np.array([(n - 1) * 10. for n in shape])
    src = ((rng.random((nshots, 1, nd)) * 0.5 + 0.25) * dom).astype(np.float32)
    src[..., -1] = 30.
    rec = ((rng.random((300, nd)) * 0.8 + 0.1) * dom).astype(np.float32)
    rec[:, -1] = 30.
    for name, fs, flags in (("OT2", False, 0), ("OT2 free surface", True, 0), ("OT4", False, mifwi.FLAG_OT4)):
        prob = orc.Problem(vp, spacing, 40, so, fs=fs)
        dt = float(prob.critical_dt) * (1.73 if flags else 1.0)
        w = orc.ricker(0.01, dt * np.arange(nt)).astype(np.float32).reshape(nt, 1)
        plan = mifwi.Plan(prob.shape_pad, spacing, prob.origin_pad, so, nt, dt, 1, 300, max_shots=nshots,
                          flags=flags | (mifwi.FLAG_FREE_SURFACE if fs else 0))
        plan.set_model(prob.vp, prob.damp)
        plan.set_geometry(src, rec)
        ng = float(np.prod(prob.shape_pad)) * nshots
        best = {}
        for _ in range(3):
            r = plan.forward(w, save=False)
            st = plan.stats()
            best["forward"] = min(best.get("forward", 1e9), st["fwd_ms"] / (nt - 2))
            if name != "OT4" or nd == 2:
                dm = np.zeros(prob.shape_pad, np.float32)
                dm[tuple(slice(45, 60) for _ in range(nd))] = 0.01
                plan.born(w, dm)
                st = plan.stats()
                best["born"] = min(best.get("born", 1e9), st["fwd_ms"] / (nt - 2))
        print("%s %-18s so=%2d  " % (label, name, so) +
              "  ".join("%s %.1f us/step = %.1f Gpts/s" % (k, v * 1e3, ng / v / 1e6) for k, v in best.items()), flush=True)
        plan.close()
    # TTI
    if so % 4 == 0:
        eps, delta, theta = .3 * (vp - 1.5), .2 * (vp - 1.5), .5 * (vp - 1.5)
        phi = .25 * (vp - 1.5) if nd == 3 else None
        prob = orc.TTIProblem(vp, spacing, 40, so, eps, delta, theta, phi)
        dt = float(prob.critical_dt)
        w = orc.ricker(0.01, dt * np.arange(nt)).astype(np.float32).reshape(nt, 1)
        plan = mifwi.Plan(prob.shape_pad, spacing, prob.origin_pad, so, nt, dt, 1, 300, max_shots=nshots)
        plan.set_model(prob.vp, prob.damp)
        plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
        plan.set_geometry(src, rec)
        ng = float(np.prod(prob.shape_pad)) * nshots
        best = 1e9
        for _ in range(3):
            plan.tti_forward(w)
            best = min(best, plan.stats()["fwd_ms"] / (nt - 2))
        print("%s %-18s so=%2d  forward %.1f us/step = %.1f Gpts/s (two coupled fields)" % (label, "TTI centered", so, best * 1e3, ng / best / 1e6), flush=True)
        plan.close()


run((340, 140), 16, 8, 300, "2-D 420x220 x 8 shots")
run((340, 140), 8, 8, 300, "2-D 420x220 x 8 shots")
run((256, 256, 256), 8, 1, 24, "3-D 336^3 x 1 shot  ")
