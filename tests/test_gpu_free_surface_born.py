"""
GPU parity of the free-surface variant (MIFWI_FLAG_FREE_SURFACE; reference
seismic/acoustic/operators.py:8-35) and of linearised Born modelling (mifwi_born; reference
operators.py:228-273, wavesolver.py:207-242) against the CPU oracle, plus the reference's
free-surface known-answer test (seismic/acoustic/acoustic_example.py:75-79: norm 369.955,
rtol 1e-3, float32) through the drop-in solver API.

Tolerances as in test_gpu_parity_abi.py: rel-L2 <= 1e-5 on records/wavefields, <= 1e-4 on
gradients (float32); 1e-11 / 1e-10 in float64.
"""
import numpy as np
import pytest

from helpers import Case, rel_l2, smooth_random_vp
from test_gpu_parity_abi import _check_case, TOL
from kat_setups import layers3d_kat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mifwi():
    import mifwi as m
    return m


def _shallow(case):
    """Move sources and receivers into the first cells below the surface, where the mirror acts."""
    h = case.prob.spacing[-1]
    case.src[..., -1] = (1.0 + np.arange(case.src.shape[0])[:, None] * 0.37) * h
    case.rec[:, -1] = 2.3 * h
    return case


@pytest.mark.parametrize("flags", [0, 4])
@pytest.mark.parametrize("so", [4, 8, 16])
def test_2d_free_surface(mifwi, so, flags):
    case = _shallow(Case((47, 40), (10.0, 10.0), nbl=9, space_order=so, nt=150, nshots=2, nrec=13, fs=True))
    assert case.prob.shape_pad == (65, 49)
    _check_case(mifwi, case, flags)


def test_2d_free_surface_differs_from_absorbing_top(mifwi):
    fs = _shallow(Case((47, 40), (10.0, 10.0), nbl=9, space_order=8, nt=150, nshots=1, nrec=13, fs=True))
    ab = Case((47, 40), (10.0, 10.0), nbl=9, space_order=8, nt=150, nshots=1, nrec=13, fs=False)
    ab.src[:], ab.rec[:] = fs.src, fs.rec
    r_fs = fs.plan(mifwi).forward(fs.wavelet)
    r_ab = ab.plan(mifwi).forward(ab.wavelet)
    assert rel_l2(r_fs, r_ab) > 0.3          # the surface reflection is a first-order effect


@pytest.mark.parametrize("flags", [0, 4])
def test_3d_free_surface(mifwi, flags):
    case = _shallow(Case((22, 27, 70), (10.0, 10.0, 10.0), nbl=6, space_order=8, nt=60, nshots=2, nrec=9,
                         fs=True, seed=4))
    _check_case(mifwi, case, flags)


def test_3d_free_surface_so16(mifwi):
    case = _shallow(Case((20, 21, 66), (10.0, 10.0, 10.0), nbl=5, space_order=16, nt=40, nshots=1, nrec=9,
                         fs=True, seed=5))
    _check_case(mifwi, case, 0)


def test_f64_free_surface(mifwi):
    case = _shallow(Case((30, 33), (10.0, 10.0), nbl=7, space_order=6, nt=100, nshots=1, nrec=9, fs=True,
                         dtype=np.float64))
    _check_case(mifwi, case, 0)


def _dm(case, seed=9):
    rng = np.random.default_rng(seed)
    dm = smooth_random_vp(case.prob.shape_pad, vmin=-0.02, vmax=0.02, seed=seed, passes=2)
    return np.ascontiguousarray(dm, dtype=case.dtype)


@pytest.mark.parametrize("flags", [0, 4])
@pytest.mark.parametrize("fs", [False, True])
def test_2d_born(mifwi, fs, flags):
    case = Case((45, 38), (10.0, 12.5), nbl=9, space_order=8, nt=160, nshots=2, nrec=13, fs=fs)
    if fs:
        _shallow(case)
    dm = _dm(case)
    plan = case.plan(mifwi, flags=flags)
    lin = plan.born(case.wavelet, dm)
    for s in range(case.nshots):
        ref = case.oracle_born(s, dm)
        assert np.abs(ref).max() > 0
        assert rel_l2(lin[s], ref) <= 1e-4, s
    # the background field left behind equals that of a plain forward run
    u_born = plan.read_wavefield(1, shot=1)
    plan.forward(case.wavelet)
    assert rel_l2(u_born, plan.read_wavefield(1, shot=1)) <= 1e-5   # two differently scheduled loops
    plan.close()


@pytest.mark.parametrize("flags", [0, 4])
def test_3d_born(mifwi, flags):
    case = Case((20, 23, 66), (10.0, 10.0, 10.0), nbl=5, space_order=8, nt=50, nshots=1, nrec=9, seed=6)
    dm = _dm(case)
    plan = case.plan(mifwi, flags=flags)
    lin = plan.born(case.wavelet, dm)
    assert rel_l2(lin[0], case.oracle_born(0, dm)) <= 1e-4
    plan.close()


def test_born_f64_and_adjoint_pair(mifwi):
    """<J dm, r> = <dm, J^T r> on the GPU (float64, generic kernels): the pairing the reference's
    solver exposes as jacobian / jacobian_adjoint."""
    case = Case((33, 31), (10.0, 10.0), nbl=8, space_order=4, nt=120, nshots=1, nrec=11, dtype=np.float64)
    dm = _dm(case)
    dm[:case.prob.nbl] = dm[-case.prob.nbl:] = 0
    dm[:, :case.prob.nbl] = dm[:, -case.prob.nbl:] = 0
    plan = case.plan(mifwi)
    lin = plan.born(case.wavelet, dm)
    assert rel_l2(lin[0], case.oracle_born(0, dm)) <= 1e-11
    rng = np.random.default_rng(2)
    r = rng.standard_normal(lin.shape) * (np.arange(case.nt)[None, :, None] > 2)
    plan.forward(case.wavelet, save=True)
    g = plan.gradient(r)
    lhs, rhs = float(np.sum(lin * r)), float(np.sum(g[0] * dm))
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), abs(rhs))
    plan.close()


def test_kat_free_surface_through_solver_api():
    """acoustic_example.py:75-79 with fs=True: ||rec|| = 369.955 (rtol 1e-3), float32."""
    from seismic import demo_model, setup_geometry
    from seismic.acoustic import AcousticWaveSolver
    c = layers3d_kat()
    model = demo_model('layers-isotropic', space_order=4, shape=c["shape"], nbl=c["nbl"], dtype=np.float32,
                       spacing=c["spacing"], fs=True)
    assert model.grid.shape == (130, 130, 90)
    geometry = setup_geometry(model, c["tn"])
    solver = AcousticWaveSolver(model, geometry, kernel='OT2', space_order=4)
    rec, u, summary = solver.forward(save=False)
    nrm = np.linalg.norm(rec.data.ravel().astype(np.float64))
    print("free-surface KAT on GPU:", nrm)
    assert np.isclose(nrm, 369.955, rtol=1e-3, atol=0)


def test_solver_jacobian_matches_oracle():
    from seismic import demo_model, setup_geometry
    from seismic.acoustic import AcousticWaveSolver
    from oracle import oracle as orc
    model = demo_model('layers-isotropic', space_order=8, shape=(60, 50), nbl=12, dtype=np.float32,
                       spacing=(10.0, 10.0), nlayers=3)
    geometry = setup_geometry(model, 400.0)
    solver = AcousticWaveSolver(model, geometry, space_order=8)
    dm_phys = smooth_random_vp(model.shape, vmin=-0.03, vmax=0.03, seed=21, passes=2).astype(np.float32)
    rec, u, U, summary = solver.born(dm_phys)
    nbl = model.nbl
    prob = orc.Problem(np.asarray(model.vp.data)[nbl:-nbl, nbl:-nbl], model.spacing, nbl, 8,
                       origin=model.origin, dt=float(model.critical_dt))
    ref = prob.born(geometry.src_positions, geometry.src.data, geometry.rec_positions,
                    np.pad(dm_phys, nbl), dt=float(model.critical_dt))
    assert rel_l2(rec.data, ref) <= 1e-4
    assert U.data.shape == (3,) + model.grid.shape and np.abs(U.data).max() > 0
    assert summary.gpointss > 0


# ---- OT4 kernel (operators.py:38-56,216-219): no number of the reference pins it (its own test,
# acoustic_example.py:66-72, checks finiteness only), so parity is against the restatement alone.
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("so", [4, 8])
def test_2d_ot4(mifwi, so, dtype):
    _check_case(mifwi, Case((45, 38), (10.0, 12.5), nbl=9, space_order=so, nt=110, nshots=2, nrec=13,
                            kernel='OT4', dtype=dtype))


def test_3d_ot4(mifwi):
    _check_case(mifwi, Case((20, 23, 30), (10.0, 10.0, 10.0), nbl=5, space_order=4, nt=40, nshots=1, nrec=9,
                            kernel='OT4', seed=8))


# The float32 OT4 plans run on the tiled kernels: a launch writes z = u + dt^2/12 laplace(u)/m and the step
# stencils z (flags 0); flags 4 selects the untiled kernels.
@pytest.mark.parametrize("flags", [0, 4])
def test_2d_ot4_so16_batch(mifwi, flags):
    _check_case(mifwi, Case((70, 90), (10.0, 10.0), nbl=12, space_order=16, nt=120, nshots=3, nrec=17,
                            kernel='OT4', seed=11), flags)


def test_3d_ot4_tiles(mifwi):
    # several tiles in y and z, two shots, order 8
    _check_case(mifwi, Case((24, 37, 75), (10.0, 10.0, 10.0), nbl=6, space_order=8, nt=40, nshots=2, nrec=9,
                            kernel='OT4', seed=12), 0)


@pytest.mark.parametrize("shape", [(70, 90), (24, 37, 75)])
def test_ot4_checkpointed_gradient_is_bit_identical(mifwi, shape):
    """MIFWI_FLAG_CHECKPOINT (8): the recomputed segments run the same launches as the forward pass."""
    nd = len(shape)
    case = Case(shape, (10.0,) * nd, nbl=6, space_order=8, nt=61, nshots=2, nrec=9, kernel='OT4', seed=13)
    out = []
    for flags in (0, 8):
        plan = case.plan(mifwi, flags=flags)
        rec = plan.forward(case.wavelet, save=True)
        g, il = plan.gradient(rec, want_illum=True)
        out.append((np.array(rec), np.array(g), np.array(il), plan.stats()))
        plan.close()
    assert out[1][3]["ckpt_steps"] > 0 and out[0][3]["ckpt_steps"] == 0
    assert np.abs(out[0][1]).max() > 0
    for a, b in zip(out[0][:3], out[1][:3]):
        assert np.array_equal(a, b)


def test_ot4_tiled_born_matches_untiled(mifwi):
    case = Case((45, 38), (10.0, 12.5), nbl=9, space_order=8, nt=110, nshots=2, nrec=13, kernel='OT4')
    dm = _dm(case)
    out = []
    for flags in (0, 4):
        plan = case.plan(mifwi, flags=flags)
        out.append(np.array(plan.born(case.wavelet, dm)))
        plan.close()
    assert np.abs(out[1]).max() > 0
    # the Born tolerance of this file (second time differences of float32 fields)
    # (the CPU restatement of Born modelling has no OT4 form: the two GPU paths check each other)
    assert rel_l2(out[0], out[1]) <= 1e-4


def test_ot4_stability_and_solver_dt():
    """acoustic_example.py:66-72: tiny grid, no absorbing layer, tn = 20000, OT4: finite records;
    the solver steps with 1.73 x the CFL time step (wavesolver.py:41-46)."""
    from seismic import demo_model, setup_geometry
    from seismic.acoustic import AcousticWaveSolver
    for ndim in (1, 2, 3):
        model = demo_model('layers-isotropic', space_order=4, shape=(11,) * ndim, nbl=0, dtype=np.float32,
                           spacing=(20.0,) * ndim)
        geometry = setup_geometry(model, 20000.0)
        solver = AcousticWaveSolver(model, geometry, kernel='OT4', space_order=4)
        assert np.isclose(float(solver.dt), 1.73 * float(model.critical_dt), rtol=1e-6)
        rec, _, _ = solver.forward(save=False)
        assert np.isfinite(np.linalg.norm(rec.data))


# OT4 with a free surface (operators.py:8-35 applied to the OT4 update of operators.py:52-56): the reference's
# freesurface() mirrors every function read through a negative z offset, the factors of the biharmonic included.
# No number of the reference pins it (parity unpinned); the library runs it on the untiled kernels in both
# precisions and is checked against the restatement, sources and receivers in the first cells below the surface.
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("so", [4, 8])
def test_2d_ot4_free_surface(mifwi, so, dtype):
    case = _shallow(Case((47, 40), (10.0, 10.0), nbl=9, space_order=so, nt=130, nshots=2, nrec=13, fs=True,
                         kernel='OT4', dtype=dtype, seed=21))
    _check_case(mifwi, case)
    plan = case.plan(mifwi)
    plan.forward(case.wavelet)
    assert plan.stats()["fwd_kernel"] == "step_generic_kernel"
    plan.close()


def test_3d_ot4_free_surface(mifwi):
    case = _shallow(Case((20, 23, 30), (10.0, 10.0, 10.0), nbl=5, space_order=4, nt=40, nshots=1, nrec=9,
                         fs=True, kernel='OT4', seed=22))
    _check_case(mifwi, case)


def test_ot4_free_surface_differs_from_absorbing_top(mifwi):
    kw = dict(nbl=9, space_order=8, nshots=1, nrec=13, seed=23)
    both = _shallow(Case((47, 40), (10.0, 10.0), nt=130, fs=True, kernel='OT4', **kw))
    ab = Case((47, 40), (10.0, 10.0), nt=130, fs=False, kernel='OT4', **kw)
    ab.src[:], ab.rec[:] = both.src, both.rec
    r_both = both.plan(mifwi).forward(both.wavelet)
    r_ab = ab.plan(mifwi).forward(ab.wavelet)
    assert rel_l2(r_both, r_ab) > 0.3


def test_ot4_free_surface_through_the_solver():
    """acoustic_example.py:26-63 run(fs=True, kernel='OT4'): the reference builds this operator; here it must run."""
    from seismic import demo_model, setup_geometry
    from seismic.acoustic import AcousticWaveSolver
    model = demo_model('layers-isotropic', space_order=4, shape=(31, 33), nbl=8, dtype=np.float32, spacing=(15.0, 15.0),
                       fs=True)
    geometry = setup_geometry(model, 300.0)
    solver = AcousticWaveSolver(model, geometry, kernel='OT4', space_order=4)
    rec, u, _ = solver.forward(save=True)
    assert np.isfinite(np.linalg.norm(rec.data)) and np.linalg.norm(rec.data) > 0
    grad, _ = solver.jacobian_adjoint(rec, u)
    assert np.isfinite(grad.data).all() and np.abs(grad.data).max() > 0
