"""
GPU parity of the C-ABI entry points against the CPU oracle on small seeded problems.

Tolerances (float32; the reference states none for its generated code, whose summation order
is Devito-internal): rel-L2 <= 1e-5 on shot records and wavefields, <= 1e-4 on gradients and
illumination; float64: 1e-11 / 1e-10.
"""
import numpy as np
import pytest

from helpers import Case, rel_l2

pytestmark = pytest.mark.gpu

TOL = {np.dtype(np.float32): (1e-5, 1e-4), np.dtype(np.float64): (1e-11, 1e-10)}


@pytest.fixture(scope="module")
def mifwi():
    import mifwi as m
    return m


def _check_case(mifwi, case, flags=0):
    tol_d, tol_g = TOL[case.dtype]
    plan = case.plan(mifwi, flags=flags)
    # forward without saving
    rec = plan.forward(case.wavelet)
    refs = [case.oracle_forward(s, save=True) for s in range(case.nshots)]
    for s in range(case.nshots):
        assert np.abs(refs[s][0]).max() > 0
        assert rel_l2(rec[s], refs[s][0]) <= tol_d, ("rec", s)
    # forward with saving: same records, wavefield matches
    rec2 = plan.forward(case.wavelet, save=True)
    assert np.array_equal(rec, rec2)
    for s in range(case.nshots):
        u = plan.read_wavefield(0, shot=s)
        assert u.shape == refs[s][1].shape
        assert rel_l2(u, refs[s][1]) <= tol_d, ("u", s)
        assert np.all(u[:2] == 0)
    # adjoint of the records
    srca = plan.adjoint(rec)
    for s in range(case.nshots):
        ref = case.oracle_adjoint(s, refs[s][0])
        assert rel_l2(srca[s], ref) <= tol_g, ("srca", s)
    # gradient of a residual (use the record itself as residual), with illumination
    grad, illum = plan.gradient(rec, want_illum=True)
    for s in range(case.nshots):
        g_ref = case.oracle_gradient(refs[s][0], refs[s][1])
        assert rel_l2(grad[s], g_ref) <= tol_g, ("grad", s)
        il_ref = (refs[s][1] * refs[s][1]).sum(axis=0)
        assert rel_l2(illum[s], il_ref) <= tol_g, ("illum", s)
    # accumulate semantics (Devito Inc): adding twice doubles
    g2 = np.array(grad)
    plan.gradient(rec, grad=g2, accumulate=True)
    assert rel_l2(g2, 2 * grad) <= 1e-6
    st = plan.stats()
    assert st["points"] == (case.nt - 2) * plan.ngrid * case.nshots
    plan.close()


# flags 0: tiled fast path where available; 4 (MIFWI_FLAG_GENERIC): untiled reference kernels
@pytest.mark.parametrize("flags", [0, 4])
@pytest.mark.parametrize("so", [2, 4, 8, 16])
def test_2d_orders(mifwi, so, flags):
    _check_case(mifwi, Case((45, 38), (10.0, 12.5), nbl=9, space_order=so, nt=140, nshots=2, nrec=13), flags)


@pytest.mark.parametrize("flags", [0, 4])
@pytest.mark.parametrize("so", [6, 10, 12, 14])
def test_2d_other_orders(mifwi, so, flags):
    _check_case(mifwi, Case((31, 29), (10.0, 10.0), nbl=8, space_order=so, nt=90, nshots=1, nrec=7, seed=3), flags)


@pytest.mark.parametrize("shape", [(75, 83), (130, 41), (20, 200)])
def test_2d_many_tiles_ragged(mifwi, shape):
    # several tiles per dimension, widths that are not multiples of 4 / of the tile, 3 shots, dense receivers
    _check_case(mifwi, Case(shape, (10.0, 10.0), nbl=9, space_order=8, nt=120, nshots=3, nrec=57, seed=13))


@pytest.mark.parametrize("flags", [0, 4])
@pytest.mark.parametrize("so", [4, 8, 16])
def test_3d_orders(mifwi, so, flags):
    _check_case(mifwi, Case((14, 12, 13), (10.0, 10.0, 10.0), nbl=6, space_order=so, nt=70, nshots=2, nrec=9, seed=5),
                flags)


@pytest.mark.parametrize("so", [2, 6, 10, 12, 14])
def test_3d_other_orders(mifwi, so):
    _check_case(mifwi, Case((11, 10, 12), (10.0, 12.0, 9.0), nbl=5, space_order=so, nt=50, nshots=1, nrec=6, seed=6))


@pytest.mark.parametrize("shape", [(40, 20, 70), (21, 37, 13)])
def test_3d_many_tiles_chunks_ragged(mifwi, shape):
    # several (y,z) tiles, several x chunks, z widths that are not multiples of 4, dense receivers
    _check_case(mifwi, Case(shape, (10.0, 10.0, 10.0), nbl=6, space_order=8, nt=60, nshots=2, nrec=80, seed=17))


def test_1d(mifwi):
    _check_case(mifwi, Case((101,), (10.0,), nbl=20, space_order=4, nt=200, nshots=2, nrec=5, seed=7))


def test_2d_f64(mifwi):
    _check_case(mifwi, Case((33, 30), (10.0, 10.0), nbl=8, space_order=8, nt=100, nshots=2, nrec=9,
                            dtype=np.float64, seed=9))


def test_multi_source_per_shot(mifwi):
    _check_case(mifwi, Case((40, 36), (10.0, 10.0), nbl=8, space_order=4, nt=110, nshots=2, nrec=9, nsrc=3, seed=11))


@pytest.mark.parametrize("shape,so", [((44, 40), 18), ((44, 40), 20), ((26, 24, 28), 18), ((26, 24, 28), 20)])
def test_space_orders_above_the_tiled_kernels(mifwi, shape, so):
    """Space orders 18 and 20 (the reference's 07_DRP_schemes.ipynb runs at 20) take the untiled kernels."""
    _check_case(mifwi, Case(shape, (10.0,) * len(shape), nbl=6, space_order=so, nt=70, nshots=2, nrec=7, seed=so))


def test_error_paths(mifwi):
    case = Case((20, 20), (10.0, 10.0), nbl=4, space_order=4, nt=20, nshots=1, nrec=3)
    with pytest.raises(mifwi.MifwiError):
        mifwi.Plan((20, 20), (10., 10.), (0., 0.), 5, 20, 1.0, 1, 3)       # odd order
    with pytest.raises(mifwi.MifwiError):
        mifwi.Plan((20, 20), (10., 10.), (0., 0.), 22, 20, 1.0, 1, 3)      # order too high
    plan = mifwi.Plan(case.prob.shape_pad, case.prob.spacing, case.prob.origin_pad, 4, case.nt, case.dt, 1, 3)
    with pytest.raises(mifwi.MifwiError):
        plan.forward(case.wavelet, nshots=1)                               # no model yet
    plan.set_model(case.prob.vp, case.prob.damp)
    plan.set_geometry(case.src, case.rec)
    with pytest.raises(mifwi.MifwiError):
        plan.gradient(np.zeros((1, case.nt, 3), np.float32))               # no saved wavefield
    plan.close()


@pytest.mark.parametrize("kind", ["2d", "3d", "2d-generic"])
def test_checkpointed_gradient_is_bit_identical(mifwi, kind, monkeypatch):
    """Two-level checkpointing (segments recomputed before the backward pass crosses them) must give
    exactly the gradient of the fully saved wavefield -- the 'blocked == unblocked' equivalence of
    the reference's tutorials/12_time_blocking.ipynb, here bit for bit (both with a kernel per step: a
    checkpointed loop is never run as one launch)."""
    monkeypatch.setenv("MIFWI_PERSIST", "0")
    if kind == "3d":
        case = Case((20, 18, 22), (10.0, 10.0, 10.0), nbl=6, space_order=8, nt=97, nshots=2, nrec=12, seed=21)
        flags = 0
    else:
        case = Case((50, 44), (10.0, 10.0), nbl=8, space_order=8, nt=131, nshots=2, nrec=15, seed=22)
        flags = 4 if kind == "2d-generic" else 0
    full = case.plan(mifwi, flags=flags)
    rec = full.forward(case.wavelet, save=True)
    g_full, il_full = full.gradient(rec, want_illum=True)
    full.close()
    ck = case.plan(mifwi, flags=flags | mifwi.FLAG_CHECKPOINT)
    rec2 = ck.forward(case.wavelet, save=True)
    assert np.array_equal(rec, rec2)
    g_ck, il_ck = ck.gradient(rec2, want_illum=True)
    assert np.array_equal(g_full, g_ck)
    assert np.array_equal(il_full, il_ck)
    out = ck.l2_misfit_gradient(case.wavelet, 0.5 * rec, want_illum=True)
    assert np.isfinite(out["fval"]).all() and np.abs(out["grad"]).max() > 0
    with pytest.raises(mifwi.MifwiError):
        ck.read_wavefield(0, shot=0)          # levels are not stored one by one
    ck.close()


@pytest.mark.parametrize("shape,so,flags", [((50, 41), 8, 0), ((50, 41), 8, 4), ((20, 23, 66), 8, 0)])
def test_per_shot_receivers_and_wavelets(mifwi, shape, so, flags):
    """Receiver tables and wavelets that differ from shot to shot (rec_per_shot / wavelet_per_shot of the
    ABI): every shot against its own oracle run, forward, adjoint and gradient."""
    case = Case(shape, (10.0,) * len(shape), nbl=6, space_order=so, nt=70, nshots=3, nrec=9, seed=21)
    rng = np.random.default_rng(5)
    dom = np.array([(n - 1) * 10.0 for n in shape])
    recs = ((rng.random((3, 9, len(shape))) * 0.7 + 0.15) * dom).astype(np.float32)
    wav = np.stack([case.wavelet * (1.0 + 0.5 * s) for s in range(3)]).astype(np.float32)
    p = case.prob
    plan = mifwi.Plan(p.shape_pad, p.spacing, p.origin_pad, so, case.nt, case.dt, 1, 9, max_shots=3, flags=flags)
    plan.set_model(p.vp, p.damp)
    plan.set_geometry(case.src, recs, rec_per_shot=True)
    rec = plan.forward(wav, wavelet_per_shot=True, save=True)
    refs = [p.forward(case.src[s], wav[s], recs[s], dt=case.dt, save=True) for s in range(3)]
    for s in range(3):
        assert rel_l2(rec[s], refs[s][0]) <= 1e-5, s
    srca = plan.adjoint(rec)
    grad = plan.gradient(rec)
    for s in range(3):
        assert rel_l2(srca[s], p.adjoint(case.src[s], recs[s], refs[s][0], dt=case.dt)) <= 1e-4, s
        assert rel_l2(grad[s], p.gradient(recs[s], refs[s][0], refs[s][1], dt=case.dt)) <= 1e-4, s
    plan.close()


@pytest.mark.parametrize("flags", [0, 4])
def test_points_on_and_beyond_the_grid_edge(mifwi, flags):
    """Sources / receivers on the last cell of the padded grid, exactly on a grid line, and one cell outside
    (the reference's generated code accepts indices -1..n, sa_01_iso_implementation1.ipynb:1255-1316)."""
    case = Case((40, 36), (10.0, 10.0), nbl=5, space_order=4, nt=90, nshots=1, nrec=6, seed=3)
    p = case.prob
    lo = np.array(p.origin_pad)
    hi = lo + (np.array(p.shape_pad) - 1) * 10.0
    case.src[0, 0] = [200.0, 150.0]                              # exactly on grid lines
    case.rec[:] = [[lo[0], 100.0], [hi[0], 120.0], [150.0, hi[1]], [lo[0] - 4.0, 90.0], [160.0, hi[1] + 6.0],
                   [200.0, 150.0]]
    plan = case.plan(mifwi, flags=flags)
    rec = plan.forward(case.wavelet)
    ref, _ = case.oracle_forward(0)
    assert rel_l2(rec[0], ref) <= 1e-5
    srca = plan.adjoint(rec)
    assert rel_l2(srca[0], case.oracle_adjoint(0, ref)) <= 1e-4
    plan.close()


def test_no_receivers_and_no_sources(mifwi):
    """nrec = 0: forward still propagates (wavefield equals the oracle's); nsrc = 0: adjoint runs."""
    case = Case((38, 35), (10.0, 10.0), nbl=6, space_order=8, nt=60, nshots=1, nrec=5, seed=8)
    p = case.prob
    plan = mifwi.Plan(p.shape_pad, p.spacing, p.origin_pad, 8, case.nt, case.dt, 1, 0, max_shots=1)
    plan.set_model(p.vp, p.damp)
    plan.set_geometry(case.src, np.zeros((0, 2), np.float32))
    rec = plan.forward(case.wavelet, save=True)
    assert rec.shape == (1, case.nt, 0)
    _, u = case.oracle_forward(0, save=True)
    assert rel_l2(plan.read_wavefield(0), u) <= 1e-5
    plan.close()
    plan = mifwi.Plan(p.shape_pad, p.spacing, p.origin_pad, 8, case.nt, case.dt, 0, 5, max_shots=1)
    plan.set_model(p.vp, p.damp)
    plan.set_geometry(np.zeros((1, 0, 2), np.float32), case.rec)
    data = np.random.default_rng(0).standard_normal((1, case.nt, 5)).astype(np.float32)
    srca = plan.adjoint(data)
    assert srca.shape == (1, case.nt, 0)
    v = plan.read_wavefield(2)
    assert np.isfinite(v).all() and np.abs(v).max() > 0
    plan.close()


def test_user_stream_async_and_device_buffers(mifwi, monkeypatch):
    """mifwi_set_stream + MIFWI_FLAG_ASYNC with inputs and outputs that live on the GPU (torch tensors):
    same numbers as the synchronous host-buffer calls (run with a kernel per step, as asynchronous plans do)."""
    import torch
    monkeypatch.setenv("MIFWI_PERSIST", "0")
    case = Case((45, 38), (10.0, 12.5), nbl=9, space_order=8, nt=120, nshots=2, nrec=13, seed=4)
    ref_plan = case.plan(mifwi)
    rec_ref = ref_plan.forward(case.wavelet, save=True)
    grad_ref, illum_ref = ref_plan.gradient(rec_ref, want_illum=True)
    ref_plan.close()
    dev = torch.device("cuda")
    stream = torch.cuda.Stream()
    plan = case.plan(mifwi, flags=mifwi.FLAG_ASYNC)
    plan.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        w = torch.from_numpy(case.wavelet).to(dev)
        rec = torch.empty((2, case.nt, case.nrec), dtype=torch.float32, device=dev)
        grad = torch.zeros((2,) + case.prob.shape_pad, dtype=torch.float32, device=dev)
        illum = torch.zeros_like(grad)
    stream.synchronize()
    plan.forward(w, rec_out=rec, save=True)
    plan.gradient(rec, grad=grad, illum=illum)
    plan.synchronize()
    assert np.array_equal(rec.cpu().numpy(), rec_ref)
    assert np.array_equal(grad.cpu().numpy(), grad_ref)
    assert np.array_equal(illum.cpu().numpy(), illum_ref)
    plan.close()


@pytest.mark.parametrize("flags", [0, 8, 64])
def test_short_time_axes_of_either_parity(mifwi, flags):
    """The tiled kernels pair time steps (two saved levels per second forward step, two steps' gradient terms
    per accumulator update); which steps pair depends on the parity of the time index and of nt.  Every
    nt from 3 (one step) to 10, with the full save, checkpointing (8) and half-precision levels (64),
    against the oracle: saved levels, gradient, illumination."""
    for shape in ((40, 37), (14, 13, 20)):
        for nt in range(3, 11):
            case = Case(shape, (10.0,) * len(shape), nbl=5, space_order=4, nt=nt, nshots=2, nrec=7, seed=nt, f0=0.05)
            res = np.random.default_rng(nt).standard_normal((2, nt, 7)).astype(np.float32)
            plan = case.plan(mifwi, flags=flags)
            plan.forward(case.wavelet, save=True)
            grad, illum = plan.gradient(res, want_illum=True)
            tol = 2e-3 if flags == 64 else 1e-4
            for s in range(2):
                _, u = case.oracle_forward(s, save=True)
                if flags != 8:
                    assert rel_l2(plan.read_wavefield(0, shot=s), u) <= (1e-3 if flags == 64 else 1e-5), (shape, nt, s)
                g_ref = case.oracle_gradient(res[s], u)
                if np.abs(g_ref).max() == 0:      # the first steps: the two fields do not overlap yet
                    assert nt <= 5 and not np.any(grad[s]), (shape, nt, s)
                else:
                    assert rel_l2(grad[s], g_ref) <= tol, (shape, nt, s, "grad")
                assert rel_l2(illum[s], (u * u).sum(axis=0)) <= tol, (shape, nt, s, "illum")
            plan.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_l2_misfit_of_given_gathers(mifwi, dtype):
    """mifwi_l2_misfit against misfit/misfit.py:5-9 restated in numpy: residual bit for bit, objective to
    float64 rounding; host arrays, device arrays, and the same bits on every call."""
    import torch
    case = Case((30, 28), (10.0, 10.0), nbl=5, space_order=4, nt=333, nshots=3, nrec=17, dtype=dtype, seed=2)
    plan = case.plan(mifwi)
    rng = np.random.default_rng(4)
    syn = rng.standard_normal((3, case.nt, 17)).astype(dtype)
    obs = rng.standard_normal((3, case.nt, 17)).astype(dtype)
    f, r = plan.l2_misfit(syn, obs)
    assert np.array_equal(r, syn - obs)
    ref = 0.5 * ((syn - obs).astype(np.float64) ** 2).reshape(3, -1).sum(axis=1)
    assert np.allclose(f, ref, rtol=1e-13, atol=0)
    d_syn, d_obs = torch.from_numpy(syn).cuda(), torch.from_numpy(obs).cuda()
    d_res = torch.empty_like(d_syn)
    d_f = torch.zeros(3, dtype=torch.float64, device="cuda")
    plan.l2_misfit(d_syn, d_obs, residual=d_res, fval=d_f)
    assert np.array_equal(d_res.cpu().numpy(), r) and np.array_equal(d_f.cpu().numpy(), f)
    f2, _ = plan.l2_misfit(syn[:2], obs[:2], nshots=2, want_residual=False)
    assert np.array_equal(f2, f[:2])
    with pytest.raises(mifwi.MifwiError):
        plan.l2_misfit(syn, obs, nshots=4)
    plan.close()


@pytest.mark.parametrize("kind", ["3d", "2d-generic"])
def test_partly_checkpointed_gradient_is_bit_identical(mifwi, kind, monkeypatch):
    """When only part of the saved wavefield fits, the first steps are checkpointed and the rest stored level
    by level (MIFWI_SAVE_BUDGET_MB stands in for a full GPU): same gradient bits as the full save, for a
    range of budgets, hence of split points and segment lengths."""
    if kind == "3d":
        case = Case((20, 18, 22), (10.0, 10.0, 10.0), nbl=6, space_order=8, nt=97, nshots=2, nrec=12, seed=21)
        flags = 0
    else:
        case = Case((50, 44), (10.0, 10.0), nbl=8, space_order=8, nt=131, nshots=2, nrec=15, seed=22)
        flags = 4
    full = case.plan(mifwi, flags=flags)
    rec = full.forward(case.wavelet, save=True)
    g_full, il_full = full.gradient(rec, want_illum=True)
    st = full.stats()
    assert st["ckpt_steps"] == 0
    full.close()
    # bytes of one time level of the batch as the library lays it out: halo of R in every dimension, rows padded
    # to 32 elements with a left z halo rounded up to 4
    n = (1,) * (3 - case.ndim) + tuple(case.prob.shape_pad)
    R = case.prob.space_order // 2
    zl = (R + 3) // 4 * 4
    pitch = (zl + (n[2] + 3) // 4 * 4 + zl + 31) // 32 * 32
    rows = n[1] + (2 * R if case.ndim >= 2 else 0)
    planes = n[0] + (2 * R if case.ndim >= 3 else 0)
    level_mb = case.nshots * (pitch * rows * planes + 288) * 4 / 2 ** 20
    seen = set()
    for frac in (0.85, 0.7, 0.55):
        monkeypatch.setenv("MIFWI_SAVE_BUDGET_MB", str(max(1, int(np.ceil(level_mb * case.nt * frac)))))
        part = case.plan(mifwi, flags=flags)
        rec2 = part.forward(case.wavelet, save=True)
        g, il = part.gradient(rec2, want_illum=True)
        st = part.stats()
        assert 0 < st["ckpt_steps"] < case.nt - 2, (frac, st["ckpt_steps"], st["ckpt_segment"])
        seen.add(st["ckpt_steps"])
        assert np.array_equal(rec, rec2)
        assert np.array_equal(g_full, g) and np.array_equal(il_full, il), (frac, st["ckpt_steps"], st["ckpt_segment"])
        out = part.l2_misfit_gradient(case.wavelet, 0.5 * rec)
        assert np.isfinite(out["fval"]).all() and np.abs(out["grad"]).max() > 0
        part.close()
    assert len(seen) >= 2


@pytest.mark.parametrize("shape,so,nt", [((40, 20, 70), 8, 61), ((70, 21, 37), 16, 40), ((33, 18, 20), 4, 30)])
def test_3d_two_steps_per_sweep_is_bit_identical(mifwi, shape, so, nt, monkeypatch):
    """MIFWI_FLAG_TIME_BLOCK: the 3-D forward loop advances two time steps per sweep over x (slabs, the second
    step one slab behind the first).  Same kernels on the same inputs: shot records and final wavefield equal the
    plain loop's bit for bit, for even and odd numbers of steps and several slab widths."""
    case = Case(shape, (10.0, 10.0, 10.0), nbl=6, space_order=so, nt=nt, nshots=2, nrec=40, seed=31)
    plain = case.plan(mifwi)
    rec = plain.forward(case.wavelet)
    u = [plain.read_wavefield(1, shot=s) for s in range(2)]
    plain.close()
    assert rel_l2(rec[0], case.oracle_forward(0)[0]) <= 1e-5
    for slab in (None, "16", "24"):
        if slab:
            monkeypatch.setenv("MIFWI_3D_SLAB", slab)
            monkeypatch.setenv("MIFWI_3D_SLAB_CHUNKS", "1")
        tb = case.plan(mifwi, flags=mifwi.FLAG_TIME_BLOCK)
        rec2 = tb.forward(case.wavelet)
        # several slab launches per step once the grid is longer than a slab (the default slab is 64 planes)
        assert tb.stats()["fwd_launches"] > nt - 2 or (slab is None and shape[0] + 12 <= 64)
        assert np.array_equal(rec, rec2), slab
        for s in range(2):
            assert np.array_equal(u[s], tb.read_wavefield(1, shot=s)), (slab, s)
        tb.close()
