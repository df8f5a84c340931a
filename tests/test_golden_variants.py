"""Golden vectors of the solver variants (tests/golden/variants_small.npz, written by
tests/golden/make_variants_golden.py from the float64 oracle): the oracle still reproduces them (CPU), and the
GPU library matches them through the C ABI in float64 (1e-6, the fixture is stored as float32) and float32
(5e-5 records, 2e-4 gradients)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from helpers import rel_l2, smooth_random_vp
from oracle import oracle as orc


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(os.path.join(HERE, "golden", "variants_small.npz")))


def test_oracle_reproduces_variant_fixtures(golden):
    import make_variants_golden as mk
    now = mk.cases()
    assert set(now) == set(golden)
    for k, v in now.items():
        assert rel_l2(np.asarray(v, np.float32), golden[k]) <= 1e-6, k


def _setup(nd):
    shape, so = ((44, 38), 8) if nd == 2 else ((20, 18, 22), 4)
    vp = smooth_random_vp(shape, seed=11)
    dom = np.array([(n - 1) * 10. for n in shape])
    src = (0.45 * dom + 2.7)[None, :]
    rec = np.stack([np.linspace(0.2, 0.8, 6) * dom[d] + 1.3 * d for d in range(nd)], axis=1)
    nt = 80 if nd == 2 else 40
    return shape, so, vp, src, rec, nt


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 5e-5)])
@pytest.mark.parametrize("nd", [2, 3])
def test_gpu_matches_variant_fixtures(golden, nd, dtype, tol):
    import mifwi
    shape, so, vp, src, rec, nt = _setup(nd)
    tag = "%dd" % nd
    src, rec = src.astype(dtype), rec.astype(dtype)
    eps, delta, theta = .3 * (vp - 1.5), .2 * (vp - 1.5), .5 * (vp - 1.5)
    phi = .25 * (vp - 1.5) if nd == 3 else None
    tti = orc.TTIProblem(vp, (10.,) * nd, 6, so, eps, delta, theta, phi, dtype=np.float64)    # set-up only
    dt = float(tti.critical_dt)
    w = orc.ricker(0.03, dt * np.arange(nt)).reshape(nt, 1).astype(dtype)
    dm = np.zeros(tti.shape_pad)
    inner = [s - 18 for s in tti.shape_pad]
    dm[tuple(slice(9, -9) for _ in shape)] = 0.02 * np.sin(np.arange(np.prod(inner))).reshape(inner)
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dtype)

    def plan_for(prob, step, flags=0):
        p = mifwi.Plan(prob.shape_pad, prob.spacing, prob.origin_pad, so, nt, step, 1, 6, max_shots=1, dtype=dtype,
                       flags=flags)
        p.set_model(c(prob.vp), c(prob.damp))
        p.set_geometry(src.reshape(1, 1, -1), rec)
        return p

    plan = plan_for(tti, dt)
    plan.tti_set_model(c(tti.epsilon), c(tti.delta), c(tti.theta), c(tti.phi))
    d = plan.tti_forward(w, save=True)
    assert rel_l2(d[0], golden["tti_forward_" + tag]) <= tol
    assert rel_l2(plan.tti_gradient(c(golden["tti_forward_" + tag])[None])[0], golden["tti_gradient_" + tag]) <= 4 * tol
    assert rel_l2(plan.tti_adjoint(c(golden["tti_forward_" + tag])[None])[0], golden["tti_adjoint_" + tag]) <= 4 * tol
    assert rel_l2(plan.tti_born(w, c(dm))[0], golden["tti_born_" + tag]) <= 4 * tol
    plan.close()

    iso = orc.Problem(vp, (10.,) * nd, 6, so, dtype=np.float64)
    dti = float(iso.critical_dt)
    wi = orc.ricker(0.03, dti * np.arange(nt)).reshape(nt, 1).astype(dtype)
    plan = plan_for(iso, dti)
    assert rel_l2(plan.born(wi, c(dm))[0], golden["born_" + tag]) <= 4 * tol
    plan.close()
    fsp = orc.Problem(vp, (10.,) * nd, 6, so, dtype=np.float64, fs=True)
    plan = plan_for(fsp, dti, mifwi.FLAG_FREE_SURFACE)
    assert rel_l2(plan.forward(wi)[0], golden["free_surface_forward_" + tag]) <= tol
    plan.close()
    ot4 = orc.Problem(vp, (10.,) * nd, 6, so, dtype=np.float64, kernel='OT4')
    d4 = float(ot4.solver_dt)
    w4 = orc.ricker(0.03, d4 * np.arange(nt)).reshape(nt, 1).astype(dtype)
    plan = plan_for(ot4, d4, mifwi.FLAG_OT4)
    r4 = plan.forward(w4, save=True)
    assert rel_l2(r4[0], golden["ot4_forward_" + tag]) <= tol
    assert rel_l2(plan.gradient(c(golden["ot4_forward_" + tag])[None])[0], golden["ot4_gradient_" + tag]) <= 4 * tol
    plan.close()
