"""
Time loops of 2-D float32 plans run as ONE launch (every tile keeps a block resident and synchronises with its
neighbours) instead of one kernel per step.  The arithmetic is the reference's (seismic/acoustic/operators.py:
59-95 forward, 143-180 adjoint, 183-225 gradient) either way, so the two schedules must agree to rounding and
both must agree with the CPU oracle; MIFWI_FLAG_NO_GRAPH selects the kernel per step.

Tolerances between the schedules: shot records, wavefields, adjoint traces and illumination 1e-5 relative (same
sums; the compiler contracts the multiply-adds of the two kernels differently, measured 2e-6 after 150 steps),
gradient 5e-5 (the kernel-per-step loop adds two steps' terms per update from a differently rounded field
update); against the oracle the usual 1e-5 (records) / 1e-4 (gradient, illumination).
"""
import os

import numpy as np
import pytest

from helpers import Case, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mifwi():
    import mifwi as m
    return m


def _run(plan, case, res):
    rec = plan.forward(case.wavelet, save=True)
    st_f = plan.stats()
    u = plan.read_wavefield(0, shot=case.nshots - 1, first=case.nt - 3, count=3)
    g, il = plan.gradient(res, want_illum=True)
    st_b = plan.stats()
    srca = plan.adjoint(res)
    rec2 = plan.forward(case.wavelet, save=False)
    st_e = plan.stats()
    return dict(rec=rec, u=u, g=g, il=il, srca=srca, rec2=rec2, fwd_kernel=st_f["fwd_kernel"], bwd_kernel=st_b["bwd_kernel"],
                launches=(st_f["fwd_launches"], st_b["bwd_launches"]), fallbacks=st_e["persist_fallbacks"])


# shape, space order, shots, sources per shot: 32 x 32 and 16 x 64 tiles, ragged edges in both directions, rows whose
# length is not a multiple of 4, radius 2 / 4 / 8, more shots than XCDs
CASES = [((70, 50), 16, 3, 1), ((61, 83), 8, 2, 2), ((45, 110), 4, 2, 1), ((30, 150), 16, 1, 1), ((50, 41), 12, 11, 1)]


@pytest.mark.parametrize("fs", [False, True])
@pytest.mark.parametrize("shape,so,nshots,nsrc", CASES)
def test_whole_loop_launch_equals_kernel_per_step_and_oracle(mifwi, shape, so, nshots, nsrc, fs):
    if fs and nshots > 3:
        pytest.skip("the free surface is covered by the smaller batches")
    case = Case(shape, (10., 10.), 10, so, nt=150, nshots=nshots, nrec=13, nsrc=nsrc, seed=3, f0=0.03, fs=fs)
    rng = np.random.default_rng(5)
    res = rng.standard_normal((nshots, case.nt, case.nrec)).astype(np.float32)
    out = {}
    for name, flags in (("loop", 0), ("steps", mifwi.FLAG_NO_GRAPH)):
        plan = case.plan(mifwi, flags=flags)
        out[name] = _run(plan, case, res)
        plan.close()
    a, b = out["loop"], out["steps"]
    assert a["fwd_kernel"] == "persist2d_kernel" and a["bwd_kernel"] == "persist2d_kernel<GRAD>", a
    assert a["launches"] == ((nshots + 7) // 8, (nshots + 7) // 8)
    assert a["fallbacks"] == 0 and b["fallbacks"] == 0
    assert b["fwd_kernel"] == "step2d_kernel" and b["launches"][0] == case.nt - 2
    assert np.array_equal(a["rec"], a["rec2"])          # saving changes nothing
    for key, tol in (("rec", 1e-5), ("u", 1e-5), ("srca", 1e-5), ("g", 5e-5), ("il", 1e-5)):
        assert rel_l2(a[key], b[key]) < tol, (key, rel_l2(a[key], b[key]))
        assert np.abs(b[key]).max() > 0
    # against the oracle, first and last shot
    for s in {0, nshots - 1}:
        syn, u = case.oracle_forward(s, save=True)
        assert rel_l2(a["rec"][s], syn) < 1e-5
        g = case.oracle_gradient(res[s], u)
        assert rel_l2(a["g"][s], g) < 1e-4 and rel_l2(a["il"][s], (u * u).sum(axis=0)) < 1e-4
        assert rel_l2(a["srca"][s], case.oracle_adjoint(s, res[s])) < 1e-5


def test_points_on_tile_corners_and_edges(mifwi):
    """Sources and receivers placed on grid nodes at the corners where four tiles meet, on the tile edges and on the
    last row and column of the grid: each is injected and sampled once, by the tile that owns it."""
    case = Case((70, 70), (10., 10.), 10, 8, nt=120, nshots=2, nrec=9, seed=1, f0=0.03)
    h, nbl = 10.0, 10
    node = lambda ix, iz: ((ix - nbl) * h, (iz - nbl) * h)   # padded index -> physical coordinate
    case.src[0, 0] = node(32, 32)
    case.src[1, 0] = node(31, 64)
    for i, (ix, iz) in enumerate([(32, 32), (31, 31), (63, 32), (64, 63), (32, 64), (10, 10), (79, 79), (31.5, 31.5), (63.999, 32.0)]):
        case.rec[i] = node(ix, iz)
    rng = np.random.default_rng(2)
    res = rng.standard_normal((2, case.nt, case.nrec)).astype(np.float32)
    out = {}
    for name, flags in (("loop", 0), ("steps", mifwi.FLAG_NO_GRAPH)):
        plan = case.plan(mifwi, flags=flags)
        out[name] = _run(plan, case, res)
        plan.close()
    assert out["loop"]["fwd_kernel"] == "persist2d_kernel"
    for key, tol in (("rec", 1e-5), ("srca", 1e-5), ("g", 5e-5)):
        assert rel_l2(out["loop"][key], out["steps"][key]) < tol, key
    syn, _ = case.oracle_forward(0)
    assert rel_l2(out["loop"]["rec"][0], syn) < 1e-5


def test_abandoned_whole_loop_is_repeated_with_a_kernel_per_step(mifwi, monkeypatch):
    """A block that waits too long for a neighbour raises a flag, every block leaves, and the library runs the loop
    again the other way: with a spin limit of zero the first unsatisfied poll does that."""
    case = Case((70, 50), (10., 10.), 10, 16, nt=100, nshots=3, nrec=7, seed=4, f0=0.03)
    rng = np.random.default_rng(9)
    res = rng.standard_normal((3, case.nt, case.nrec)).astype(np.float32)
    ref = case.plan(mifwi, flags=mifwi.FLAG_NO_GRAPH)
    want = _run(ref, case, res)
    ref.close()
    monkeypatch.setenv("MIFWI_PERSIST_SPIN", "0")
    plan = case.plan(mifwi)
    got = _run(plan, case, res)
    plan.close()
    assert got["fwd_kernel"] == "step2d_kernel" and got["bwd_kernel"] == "step2d_kernel<GRAD>"
    assert got["fallbacks"] >= 1 and want["fallbacks"] == 0
    for key in ("rec", "u", "g", "il", "srca", "rec2"):
        assert np.array_equal(got[key], want[key]), key


def test_a_shot_whose_xcd_takes_no_part_is_noticed(mifwi, monkeypatch):
    """On a device that deals no block to one of the XCD ids (a partition of the GPU, a part with fewer dies) the
    shot dealt to that id would stay untouched while nobody waits for it.  The step counts of every tile are
    checked behind the launch: the loop is then run again with a kernel per step, and counted."""
    case = Case((70, 50), (10., 10.), 10, 8, nt=80, nshots=3, nrec=7, seed=4, f0=0.03)
    rng = np.random.default_rng(9)
    res = rng.standard_normal((3, case.nt, case.nrec)).astype(np.float32)
    ref = case.plan(mifwi, flags=mifwi.FLAG_NO_GRAPH)
    want = _run(ref, case, res)
    ref.close()
    monkeypatch.setenv("MIFWI_PERSIST_DROP_SLOT", "1")
    plan = case.plan(mifwi)
    got = _run(plan, case, res)
    plan.close()
    assert got["fwd_kernel"] == "step2d_kernel" and got["fallbacks"] >= 1
    for key in ("rec", "u", "g", "il", "srca", "rec2"):
        assert np.array_equal(got[key], want[key]), key
    assert np.abs(got["rec"][1]).max() > 0 and np.abs(got["g"][1]).max() > 0   # the dropped shot was modelled after all


def test_devices_without_eight_xcds_keep_the_kernel_per_step(mifwi, monkeypatch):
    case = Case((40, 40), (10., 10.), 10, 8, nt=40, nshots=2, nrec=5, seed=4, f0=0.03)
    for fake in ("4", "1", "0"):
        monkeypatch.setenv("MIFWI_FAKE_XCDS", fake)
        plan = case.plan(mifwi)
        plan.forward(case.wavelet)
        st = plan.stats()
        assert st["fwd_kernel"] == "step2d_kernel" and st["persist_fallbacks"] == 0, fake
        plan.close()


def test_switch_and_flags_that_keep_the_kernel_per_step(mifwi, monkeypatch):
    case = Case((40, 40), (10., 10.), 10, 8, nt=40, nshots=1, nrec=5, seed=4, f0=0.03)
    for flags in (mifwi.FLAG_ASYNC, mifwi.FLAG_SAVE_F16, mifwi.FLAG_CHECKPOINT):
        plan = case.plan(mifwi, flags=flags)
        plan.forward(case.wavelet, save=bool(flags & (mifwi.FLAG_SAVE_F16 | mifwi.FLAG_CHECKPOINT)))
        assert plan.stats()["fwd_kernel"] == "step2d_kernel", flags
        plan.close()
    monkeypatch.setenv("MIFWI_PERSIST", "0")
    plan = case.plan(mifwi)
    plan.forward(case.wavelet)
    assert plan.stats()["fwd_kernel"] == "step2d_kernel"
    plan.close()
