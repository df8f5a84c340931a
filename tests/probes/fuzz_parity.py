"""GPU probe: randomised parity sweep of the C-ABI entry points against the CPU oracle (shapes, orders, shot
counts, receiver layouts, free surface, half-precision saving, TTI).  Prints the worst relative errors."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import mifwi
from helpers import Case, rel_l2, smooth_random_vp
from oracle import oracle as orc

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
worst = {}
def note(k, v, info):
    if v > worst.get(k, (0, None))[0]:
        worst[k] = (v, info)

for c in range(ncases):
    rng = np.random.default_rng(seed0 * 1000 + c)
    nd = 2 if rng.random() < 0.7 else 3
    so = int(rng.choice([2, 4, 6, 8, 10, 12, 14, 16]))
    shape = tuple(int(x) for x in (rng.integers(20, 110, nd) if nd == 2 else rng.integers(14, 44, nd)))
    nshots = int(rng.integers(1, 5))
    nrec = int(rng.integers(1, 14))
    nsrc = int(rng.integers(1, 3))
    fs = bool(rng.random() < 0.25)
    nbl = int(rng.integers(4, 12))
    nt = int(rng.integers(30, 90))
    flags = mifwi.FLAG_SAVE_F16 if rng.random() < 0.3 else 0
    if rng.random() < 0.15:
        flags |= mifwi.FLAG_CHECKPOINT
    if rng.random() < 0.15:
        flags |= mifwi.FLAG_NO_GRAPH
    info = dict(shape=shape, so=so, nshots=nshots, nrec=nrec, nsrc=nsrc, fs=fs, nbl=nbl, nt=nt, flags=flags)
    case = Case(shape, (10.,) * nd, nbl, so, nt=nt, nshots=nshots, nrec=nrec, nsrc=nsrc, seed=c, f0=0.03, fs=fs)
    plan = case.plan(mifwi, flags=flags)
    f16 = bool(flags & mifwi.FLAG_SAVE_F16) and not (flags & mifwi.FLAG_CHECKPOINT)
    rec = plan.forward(case.wavelet, save=True)
    res = np.random.default_rng(c).standard_normal(rec.shape).astype(np.float32)
    grad, il = plan.gradient(res, want_illum=True)
    srca = plan.adjoint(res)
    for s in range(nshots):
        r_ref, u_ref = case.oracle_forward(s, save=True)
        note("forward", rel_l2(rec[s], r_ref), info)
        g_ref = case.prob.gradient(case.rec, res[s], u_ref, dt=case.dt)
        note("gradient_f16" if f16 else "gradient", rel_l2(grad[s], g_ref), info)
        il_ref = (u_ref.astype(np.float64) ** 2).sum(0)
        note("illum_f16" if f16 else "illum", rel_l2(il[s], il_ref), info)
        a_ref = case.prob.adjoint(case.src[s], case.rec, res[s], dt=case.dt)
        note("adjoint", rel_l2(srca[s], a_ref), info)
    plan.close()
    # TTI on the same grid when the order allows it (no free surface)
    if so % 4 == 0 and not fs:
        vp = smooth_random_vp(shape, seed=c)
        eps, delta, theta = .3 * (vp - 1.5), .2 * (vp - 1.5), .5 * (vp - 1.5)
        phi = .25 * (vp - 1.5) if nd == 3 else None
        prob = orc.TTIProblem(vp, (10.,) * nd, nbl, so, eps, delta, theta, phi, dtype=np.float32)
        dt = float(prob.critical_dt)
        w = orc.ricker(0.03, dt * np.arange(nt)).reshape(nt, 1).astype(np.float32)
        src, recp = case.src[:, :1], case.rec
        plan = mifwi.Plan(prob.shape_pad, prob.spacing, prob.origin_pad, so, nt, dt, 1, nrec, max_shots=nshots, dtype=np.float32)
        plan.set_model(prob.vp, prob.damp)
        if c % 2:
            plan.set_geometry(src, recp); plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
        else:
            plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi); plan.set_geometry(src, recp)
        d = plan.tti_forward(w, save=True)
        g = plan.tti_gradient(res[:, :, :nrec] if res.shape[2] >= nrec else res)
        for s in range(nshots):
            ref = prob.forward(src[s], w, recp, dt=dt)
            note("tti_forward", rel_l2(d[s], ref), info)
            note("tti_gradient", rel_l2(g[s], prob.gradient(src[s], w, recp, res[s], dt=dt)), info)
        plan.close()
    print("case %d ok %s" % (c, info), flush=True)
print("---- worst relative errors ----")
for k, (v, info) in sorted(worst.items()):
    print("%-14s %.3e  at %s" % (k, v, info))
