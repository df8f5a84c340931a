"""GPU parity of the TTI propagators (mifwi_tti_set_model / mifwi_tti_forward / mifwi_tti_adjoint;
reference seismic/tti/operators.py:198-313,430-519) against the CPU oracle, the adjoint-pair
identity on the GPU, and the reference's stability test through the drop-in solver.
Tolerances: rel-L2 <= 2e-5 on records (float32; two coupled fields and nested first derivatives
accumulate a little more rounding than the acoustic kernels), 1e-11 in float64.  The reference pins
no TTI numbers, so parity here is against the restatement only."""
import numpy as np
import pytest

from helpers import rel_l2
from test_oracle_tti import tti_case
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mifwi():
    import mifwi as m
    return m


def _plan(mifwi, prob, nt, dt, nsrc, nrec, src, rec):
    plan = mifwi.Plan(prob.shape_pad, prob.spacing, prob.origin_pad, prob.space_order, nt, dt, nsrc, nrec,
                      max_shots=1, dtype=prob.dtype)
    plan.set_model(prob.vp, prob.damp)
    plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
    plan.set_geometry(src.reshape(1, nsrc, -1), rec)
    return plan


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-5), (np.float64, 1e-11)])
@pytest.mark.parametrize("shape,so", [((41, 37), 8), ((45, 33), 4), ((19, 17, 21), 8), ((21, 18, 70), 4)])
def test_tti_forward_adjoint_parity(mifwi, shape, so, dtype, tol):
    prob, vp, src, rec, rng = tti_case(shape, so=so, dtype=dtype)
    dt = float(prob.critical_dt)
    nt = 100 if len(shape) == 2 else 45
    w = orc.ricker(0.03, dt * np.arange(nt)).reshape(nt, 1).astype(dtype)
    src, rec = src.astype(dtype), rec.astype(dtype)
    plan = _plan(mifwi, prob, nt, dt, 1, rec.shape[0], src, rec)
    d = plan.tti_forward(w)[0]
    ref = prob.forward(src, w, rec, dt=dt)
    assert np.abs(ref).max() > 0
    assert rel_l2(d, ref) <= tol
    a = plan.tti_adjoint(ref[None])[0]
    assert rel_l2(a, prob.adjoint(src, rec, ref, dt=dt)) <= 5 * tol
    u, v = plan.read_wavefield(3), plan.read_wavefield(4)
    assert u.shape == (3,) + prob.shape_pad and np.abs(u).max() > 0 and np.abs(v).max() > 0
    plan.close()


@pytest.mark.parametrize("shape,so", [((70, 48), 16), ((53, 80), 8), ((90, 30), 12), ((33, 45), 4)])
def test_tti_tiled_kernel_matches_untiled_and_oracle(mifwi, shape, so):
    """float32 2-D plans run the tiled kernel (both tile shapes occur over these grids); FLAG_GENERIC
    keeps the untiled gz + step + sparse launches.  Three shots with their own sources and receivers,
    checked shot by shot against the oracle as well."""
    prob, vp, src0, rec0, rng = tti_case(shape, so=so, dtype=np.float32)
    dt, nt, nshots = float(prob.critical_dt), 90, 3
    dom = np.array([(n - 1) * 10. for n in shape])
    src = ((rng.random((nshots, 2, 2)) * 0.5 + 0.25) * dom).astype(np.float32)
    rec = ((rng.random((nshots, 9, 2)) * 0.9 + 0.05) * dom).astype(np.float32)
    rec[0, 0] = 0.0          # a receiver on the corner of the physical domain
    rec[1, 1] = dom          # and on the opposite one
    w = (orc.ricker(0.03, dt * np.arange(nt))[None, :, None] * np.array([1., -.5])[None, None, :] *
         (1 + np.arange(nshots))[:, None, None]).astype(np.float32)
    out = {}
    for name, flags in (("tiled", 0), ("untiled", mifwi.FLAG_GENERIC)):
        plan = mifwi.Plan(prob.shape_pad, prob.spacing, prob.origin_pad, so, nt, dt, 2, 9, max_shots=nshots,
                          dtype=np.float32, flags=flags)
        plan.set_model(prob.vp, prob.damp)
        if so == 8:      # geometry first: the first tti_set_model re-tiles the plan and rebuilds the sparse tables
            plan.set_geometry(src, rec, rec_per_shot=True)
            plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
        else:
            plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
            plan.set_geometry(src, rec, rec_per_shot=True)
        d = plan.tti_forward(w, wavelet_per_shot=True)
        a = plan.tti_adjoint(d)
        out[name] = (d, a, plan.read_wavefield(3, shot=2), plan.read_wavefield(6, shot=1))
        plan.close()
    for x, y in zip(out["tiled"], out["untiled"]):
        assert np.abs(y).max() > 0 and rel_l2(x, y) <= 5e-5   # float32 rounding (different contraction)
    for s in range(nshots):
        ref = prob.forward(src[s], w[s], rec[s], dt=dt)
        assert rel_l2(out["tiled"][0][s], ref) <= 2e-5
        assert rel_l2(out["tiled"][1][s], prob.adjoint(src[s], rec[s], ref, dt=dt)) <= 1e-4


@pytest.mark.parametrize("tx", ["32", "16"])
def test_tti_tiled_kernel_on_the_large_tiles(mifwi, monkeypatch, tx):
    """Large batches at high space orders keep the 32-row (or 16 x 64) tiles of the acoustic kernels for the TTI
    kernel; small test grids never reach that rule, so the tile height is pinned here (MIFWI_2D_TX also keeps
    tti_set_model from re-tiling the plan)."""
    monkeypatch.setenv("MIFWI_2D_TX", tx)
    shape, so = (70, 48), 16
    prob, vp, src0, rec0, rng = tti_case(shape, so=so, dtype=np.float32)
    dt, nt, nshots = float(prob.critical_dt), 60, 2
    dom = np.array([(n - 1) * 10. for n in shape])
    src = ((rng.random((nshots, 1, 2)) * 0.5 + 0.25) * dom).astype(np.float32)
    rec = ((rng.random((nshots, 9, 2)) * 0.9 + 0.05) * dom).astype(np.float32)
    w = (orc.ricker(0.03, dt * np.arange(nt))[None, :, None] * (1 + np.arange(nshots))[:, None, None]).astype(np.float32)
    out = {}
    for name, flags in (("tiled", 0), ("untiled", mifwi.FLAG_GENERIC)):
        plan = mifwi.Plan(prob.shape_pad, prob.spacing, prob.origin_pad, so, nt, dt, 1, 9, max_shots=nshots,
                          dtype=np.float32, flags=flags)
        plan.set_model(prob.vp, prob.damp)
        plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
        plan.set_geometry(src, rec, rec_per_shot=True)
        d = plan.tti_forward(w, wavelet_per_shot=True)
        out[name] = (d, plan.tti_adjoint(d), plan.read_wavefield(3, shot=1))
        plan.close()
    for x, y in zip(out["tiled"], out["untiled"]):
        assert np.abs(y).max() > 0 and rel_l2(x, y) <= 5e-5
    assert rel_l2(out["tiled"][0][0], prob.forward(src[0], w[0], rec[0], dt=dt)) <= 2e-5


@pytest.mark.parametrize("shape,so", [((40, 21, 70), 16), ((30, 19, 23), 12), ((25, 37, 131), 8), ((20, 18, 30), 4)])
def test_tti_3d_marching_kernels_match_untiled_and_oracle(mifwi, monkeypatch, shape, so):
    """float32 3-D plans step with the marching kernels (every radius they are built for, ragged tiles in y and z,
    several chunks of planes, two shots); FLAG_GENERIC keeps the kernels with one thread per point."""
    prob, vp, src0, rec0, rng = tti_case(shape, so=so, dtype=np.float32)
    dt, nt, nshots = float(prob.critical_dt), 40, 2
    dom = np.array([(n - 1) * 10. for n in shape])
    src = ((rng.random((nshots, 1, 3)) * 0.5 + 0.25) * dom).astype(np.float32)
    rec = ((rng.random((nshots, 7, 3)) * 0.9 + 0.05) * dom).astype(np.float32)
    w = (orc.ricker(0.03, dt * np.arange(nt))[None, :, None] * (1 + np.arange(nshots))[:, None, None]).astype(np.float32)
    out = {}
    variants = [("march", 0, None), ("untiled", mifwi.FLAG_GENERIC, None)]
    if so <= 8:   # these orders take one march by default; the two-march form of the same orders is kept as well
        variants.append(("two marches", 0, "1"))
    for name, flags, two_pass in variants:
        plan = mifwi.Plan(prob.shape_pad, prob.spacing, prob.origin_pad, so, nt, dt, 1, 7, max_shots=nshots,
                          dtype=np.float32, flags=flags)
        plan.set_model(prob.vp, prob.damp)
        if two_pass:
            monkeypatch.setenv("MIFWI_TTI3D_TWOPASS", two_pass)   # read by tti_set_model
        plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
        monkeypatch.delenv("MIFWI_TTI3D_TWOPASS", raising=False)
        plan.set_geometry(src, rec, rec_per_shot=True)
        d = plan.tti_forward(w, wavelet_per_shot=True)
        a = plan.tti_adjoint(d)
        out[name] = (d, a, plan.read_wavefield(3, shot=1), plan.read_wavefield(6, shot=0))
        plan.close()
    for name in out:
        if name != "untiled":
            for x, y in zip(out[name], out["untiled"]):
                assert np.abs(y).max() > 0 and rel_l2(x, y) <= 5e-5   # float32 rounding (different contraction)
    if "two marches" in out:   # and the two forms do differ in rounding, i.e. both really ran
        assert any(not np.array_equal(x, y) for x, y in zip(out["march"], out["two marches"]))
    ref = prob.forward(src[1], w[1], rec[1], dt=dt)
    assert rel_l2(out["march"][0][1], ref) <= 2e-5
    assert rel_l2(out["march"][1][1], prob.adjoint(src[1], rec[1], ref, dt=dt)) <= 1e-4


@pytest.mark.parametrize("dtype,tol", [(np.float32, 5e-5), (np.float64, 1e-10)])
@pytest.mark.parametrize("shape,so", [((41, 37), 8), ((60, 75), 16), ((17, 15, 19), 4)])
def test_tti_jacobian_pair_parity(mifwi, shape, so, dtype, tol):
    """mifwi_tti_born / mifwi_tti_forward(save) + mifwi_tti_gradient against the oracle's restatement
    of seismic/tti/operators.py:529-639 (rel-L2; float32 gradients to 2e-4 like the acoustic ones)."""
    prob, vp, src, rec, rng = tti_case(shape, so=so, nbl=10, dtype=dtype)
    dt = float(prob.critical_dt)
    nt = 90 if len(shape) == 2 else 40
    w = orc.ricker(0.03, dt * np.arange(nt)).reshape(nt, 1).astype(dtype)
    src, rec = src.astype(dtype), rec.astype(dtype)
    dm = np.zeros(prob.shape_pad, dtype)
    inner = tuple(slice(12, -12) for _ in shape)
    dm[inner] = 0.02 * rng.standard_normal(dm[inner].shape)
    plan = _plan(mifwi, prob, nt, dt, 1, rec.shape[0], src, rec)
    jd = plan.tti_born(w, dm)[0]
    ref = prob.born(src, w, rec, dm, dt=dt)
    assert np.abs(ref).max() > 0 and rel_l2(jd, ref) <= tol
    r = rng.standard_normal(ref.shape).astype(dtype)
    d = plan.tti_forward(w, save=True)[0]
    assert rel_l2(d, prob.forward(src, w, rec, dt=dt)) <= tol
    u0 = plan.read_wavefield(7)
    assert u0.shape == (nt,) + prob.shape_pad and np.abs(u0[nt - 1]).max() > 0 and not u0[:2].any()
    g = plan.tti_gradient(r[None])[0]
    gref = prob.gradient(src, w, rec, r, dt=dt)
    assert np.abs(gref).max() > 0 and rel_l2(g, gref) <= 4 * tol
    g2 = plan.tti_gradient(r[None], grad=g[None].copy(), accumulate=True)[0]
    assert rel_l2(g2, 2 * gref) <= 4 * tol
    if dtype == np.float64:      # <J dm, r> = <dm, J^T r> on the device
        lhs, rhs = float(np.sum(jd * r)), float(np.sum(dm * g))
        assert abs(lhs - rhs) <= 1e-10 * max(abs(lhs), abs(rhs))
    plan.close()


def test_tti_gradient_needs_saved_fields(mifwi):
    prob, vp, src, rec, rng = tti_case((33, 35), so=8, dtype=np.float32)
    nt, dt = 30, float(prob.critical_dt)
    plan = _plan(mifwi, prob, nt, dt, 1, rec.shape[0], src.astype(np.float32), rec.astype(np.float32))
    with pytest.raises(RuntimeError, match="save"):
        plan.tti_gradient(np.zeros((1, nt, rec.shape[0]), np.float32))
    plan.close()


def test_tti_adjoint_pair_on_gpu(mifwi):
    prob, vp, src, rec, rng = tti_case((37, 41), so=8, dtype=np.float64)
    dt, nt = float(prob.critical_dt), 120
    plan = _plan(mifwi, prob, nt, dt, 1, rec.shape[0], src, rec)
    s = rng.standard_normal((nt, 1))
    r = rng.standard_normal((1, nt, rec.shape[0]))
    fs = plan.tti_forward(s)
    ftr = plan.tti_adjoint(r)
    lhs, rhs = float(np.sum(fs * r)), float(np.sum(s * ftr[0]))
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))
    plan.close()


def test_tti_requires_model_and_order(mifwi):
    plan = mifwi.Plan((30, 30), (10., 10.), (0., 0.), 4, 10, 1.0, 1, 1)
    plan.set_model(np.full((30, 30), 2.0, np.float32), np.zeros((30, 30), np.float32))
    plan.set_geometry(np.array([[[100., 100.]]], np.float32), np.array([[150., 150.]], np.float32))
    with pytest.raises(mifwi.MifwiError):
        plan.tti_forward(np.zeros((10, 1), np.float32))
    with pytest.raises(mifwi.MifwiError):
        plan.tti_born(np.zeros((10, 1), np.float32), np.zeros((30, 30), np.float32))
    with pytest.raises(mifwi.MifwiError):
        plan.tti_gradient(np.zeros((1, 10, 1), np.float32))
    z = np.zeros((30, 30), np.float32)
    plan.tti_set_model(z, z, z)
    with pytest.raises(mifwi.MifwiError):                  # more shots than the plan holds
        plan.tti_forward(np.zeros((10, 1), np.float32), nshots=2)
    with pytest.raises(mifwi.MifwiError) as e:             # no saved fields yet
        plan.tti_gradient(np.zeros((1, 10, 1), np.float32))
    assert e.value.code == -4
    with pytest.raises(mifwi.MifwiError):                  # saved TTI fields are not readable before a saving run
        plan.read_wavefield(7)
    plan.tti_forward(np.zeros((10, 1), np.float32), save=True)
    assert not plan.read_wavefield(8).any()
    assert not plan.tti_gradient(np.zeros((1, 10, 1), np.float32)).any()
    plan.close()
    plan = mifwi.Plan((30, 30), (10., 10.), (0., 0.), 4, 10, 1.0, 1, 1, flags=mifwi.FLAG_FREE_SURFACE)
    with pytest.raises(mifwi.MifwiError):                  # no TTI with a free surface
        plan.tti_set_model(z, z, z)
    plan.close()
    plan = mifwi.Plan((30, 30), (10., 10.), (0., 0.), 6, 10, 1.0, 1, 1)
    z = np.zeros((30, 30), np.float32)
    with pytest.raises(mifwi.MifwiError):
        plan.tti_set_model(z, z, z)
    plan.close()


@pytest.mark.parametrize("ndim", [2, 3])
def test_tti_stability_through_solver(ndim):
    """tti_example.py:42-49 through AnisotropicWaveSolver: finite records after tn = 16000."""
    from seismic import demo_model, setup_geometry
    from seismic.tti import AnisotropicWaveSolver
    model = demo_model('layers-tti', shape=(11,) * ndim, spacing=(20.,) * ndim, space_order=4, nbl=0)
    geometry = setup_geometry(model, 16000.0)
    solver = AnisotropicWaveSolver(model, geometry, space_order=4)
    rec, u, v, summary = solver.forward()
    assert np.isfinite(np.linalg.norm(rec.data.astype(np.float64))) and np.abs(rec.data).max() > 0
    srca, p, r, _ = solver.adjoint(rec)
    assert np.isfinite(np.linalg.norm(srca.data.astype(np.float64)))
    assert u.data.shape == (3,) + model.grid.shape and np.abs(v.data).max() > 0


def test_tti_solver_jacobian_pair():
    """AnisotropicWaveSolver.jacobian / jacobian_adjoint (tti/wavesolver.py:219-357): the pair satisfies
    <J dm, r> = <dm, J^T r> to float32 accuracy, device-resident and host-copied u0/v0 give the same
    gradient, and the gradient accumulates into a supplied dm."""
    from seismic import demo_model, setup_geometry, Function, Receiver
    from seismic.tti import AnisotropicWaveSolver
    model = demo_model('layers-tti', shape=(51, 47), spacing=(10., 10.), space_order=8, nbl=12)
    geometry = setup_geometry(model, 300.0)
    solver = AnisotropicWaveSolver(model, geometry, space_order=8)
    rng = np.random.default_rng(5)
    dm = np.zeros(model.grid.shape, np.float32)
    dm[20:-20, 20:-20] = 0.02 * rng.standard_normal(dm[20:-20, 20:-20].shape)
    lin = Receiver(name='lin', grid=model.grid, time_range=geometry.time_axis, coordinates=geometry.rec_positions)
    lin, u0, v0, du, dv, _ = solver.born(dm, rec=lin)
    jd = lin.data.astype(np.float64)
    assert np.abs(jd).max() > 0 and du.data.shape == (3,) + model.grid.shape
    res = Receiver(name='res', grid=model.grid, time_range=geometry.time_axis, coordinates=geometry.rec_positions)
    res.data[:] = rng.standard_normal(res.data.shape)
    rec, u, v, _ = solver.forward(save=True)
    g, _ = solver.jacobian_adjoint(res, u, v)
    lhs, rhs = float(np.sum(jd * res.data)), float(np.sum(dm.astype(np.float64) * g.data))
    assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs))
    assert u.data.shape == (geometry.nt,) + model.grid.shape and not u.on_device()
    acc = Function(name='acc', grid=model.grid)
    acc.data[...] = g.data
    g2, _ = solver.jacobian_adjoint(res, u, v, dm=acc)          # host copies: fields are modelled again
    assert g2 is acc and rel_l2(acc.data, 2 * g.data) <= 1e-6


@pytest.mark.parametrize("shape,so,flags", [((70, 48), 8, 0), ((53, 40), 8, 4), ((19, 17, 21), 4, 0)])
def test_tti_checkpointed_gradient_is_bit_identical(mifwi, shape, so, flags, monkeypatch):
    """mifwi_tti_forward(save=1) keeps two levels of u and v per segment when every level does not fit (forced here
    with MIFWI_FLAG_CHECKPOINT) and mifwi_tti_gradient propagates each segment again before crossing it: the
    same kernels in the same order, hence the same gradient bits as with every level stored
    (seismic/tti/wavesolver.py:282-357 is the reference's checkpointed counterpart).  Tiled 2-D, untiled 2-D, 3-D.
    Checkpointed loops run a kernel per step; "the same bits" holds between plans on that schedule, so the tiled 2-D
    case switches the whole-loop launches off (they are compared, to rounding, at the end)."""
    if len(shape) == 2 and flags == 0:
        monkeypatch.setenv("MIFWI_TTI_PERSIST", "0")
    prob, vp, src, rec, rng = tti_case(shape, so=so, dtype=np.float32)
    dt = float(prob.critical_dt)
    nt = 83 if len(shape) == 2 else 41
    w = orc.ricker(0.03, dt * np.arange(nt)).reshape(nt, 1).astype(np.float32)
    src, rec = src.astype(np.float32), rec.astype(np.float32)
    res = rng.standard_normal((1, nt, rec.shape[0])).astype(np.float32)
    out = {}
    for name, fl in (("full", flags), ("ckpt", flags | mifwi.FLAG_CHECKPOINT)):
        plan = mifwi.Plan(prob.shape_pad, prob.spacing, prob.origin_pad, prob.space_order, nt, dt, 1, rec.shape[0],
                          max_shots=1, dtype=np.float32, flags=fl)
        plan.set_model(prob.vp, prob.damp)
        plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
        plan.set_geometry(src.reshape(1, 1, -1), rec)
        d = plan.tti_forward(w, save=True)
        if name == "ckpt":
            plan.tti_forward(2.0 * w[::-1].copy())        # other modelling in between must not disturb the checkpoints
            with pytest.raises(mifwi.MifwiError):
                plan.read_wavefield(7)                    # levels are not stored one by one
        out[name] = (d, plan.tti_gradient(res))
        plan.close()
    assert np.array_equal(out["full"][0], out["ckpt"][0])
    assert np.abs(out["full"][1]).max() > 0
    assert np.array_equal(out["full"][1], out["ckpt"][1])
    if len(shape) == 2 and flags == 0:
        # the checkpointed plan of a build that runs whole loops (its forward loop without saving does, its
        # checkpointed loops do not) agrees with both to rounding
        monkeypatch.delenv("MIFWI_TTI_PERSIST")
        plan = mifwi.Plan(prob.shape_pad, prob.spacing, prob.origin_pad, prob.space_order, nt, dt, 1, rec.shape[0],
                          max_shots=1, dtype=np.float32, flags=mifwi.FLAG_CHECKPOINT)
        plan.set_model(prob.vp, prob.damp)
        plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
        plan.set_geometry(src.reshape(1, 1, -1), rec)
        d = plan.tti_forward(w, save=True)
        g = plan.tti_gradient(res)
        assert plan.stats()["ckpt_steps"] >= 0 and rel_l2(d, out["full"][0]) < 2e-5 and rel_l2(g, out["full"][1]) < 1e-4
        plan.close()
