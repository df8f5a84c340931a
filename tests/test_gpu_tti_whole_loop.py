"""
2-D TTI time loops of float32 plans at space orders 4 and 8 run as ONE launch (csrc/kernels_persist_tti2d.hip: every
32 x 32 tile keeps a block resident and synchronises with its neighbours) instead of one kernel per step.  The
arithmetic is the reference's 'centered' kernel (seismic/tti/operators.py:180-243, 430-481 forward, 484-527 adjoint,
590-639 Jacobian adjoint) either way, so the two schedules must agree to rounding and both with the CPU restatement;
MIFWI_FLAG_NO_GRAPH selects the kernel per step.  The reference holds no TTI numbers: parity unpinned.

Tolerances between the schedules: records, fields and adjoint traces 2e-5 relative (same sums; the two kernels
contract their multiply-adds differently and one divides through a refined reciprocal), gradient 1e-4; against the
restatement 2e-5 (records) / 1e-4 (adjoint traces, gradient), as in test_gpu_tti.py.
"""
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


def _setup(shape, so, nshots, nsrc=1, nrec=9, nt=120, seed=0):
    prob, vp, src0, rec0, rng = tti_case(shape, so=so, dtype=np.float32, seed=seed)
    dt = float(prob.critical_dt)
    dom = np.array([(n - 1) * 10. for n in shape])
    src = ((rng.random((nshots, nsrc, 2)) * 0.5 + 0.25) * dom).astype(np.float32)
    rec = ((rng.random((nrec, 2)) * 0.9 + 0.05) * dom).astype(np.float32)
    rec[0] = 0.0           # a receiver on the corner of the physical domain
    rec[1] = dom           # and on the opposite one
    w = (orc.ricker(0.03, dt * np.arange(nt))[None, :, None] * (1 + 0.3 * np.arange(nsrc))[None, None, :] *
         (1 + np.arange(nshots))[:, None, None]).astype(np.float32)
    return prob, dt, nt, src, rec, w, rng


def _run(mifwi, prob, dt, nt, src, rec, w, res, flags):
    nshots, nsrc = src.shape[0], src.shape[1]
    plan = mifwi.Plan(prob.shape_pad, prob.spacing, prob.origin_pad, prob.space_order, nt, dt, nsrc, rec.shape[0],
                      max_shots=nshots, dtype=np.float32, flags=flags)
    plan.set_model(prob.vp, prob.damp)
    plan.set_geometry(src, rec)
    plan.tti_set_model(prob.epsilon, prob.delta, prob.theta, prob.phi)
    d0 = np.array(plan.tti_forward(w, wavelet_per_shot=True))
    st0 = plan.stats()
    u3 = np.array(plan.read_wavefield(3, shot=nshots - 1))
    a = np.array(plan.tti_adjoint(res))
    st_a = plan.stats()
    r6 = np.array(plan.read_wavefield(6, shot=0))
    d1 = np.array(plan.tti_forward(w, wavelet_per_shot=True, save=True))
    u_saved = np.array(plan.read_wavefield(7, shot=0, first=nt - 2, count=2))
    g = np.array(plan.tti_gradient(res))
    st_g = plan.stats()
    plan.close()
    return dict(d0=d0, d1=d1, a=a, g=g, u3=u3, r6=r6, u_saved=u_saved, fwd_kernel=st0["fwd_kernel"],
                adj_kernel=st_a["bwd_kernel"], grad_kernel=st_g["bwd_kernel"], launches=(st0["fwd_launches"], st_g["bwd_launches"]),
                fallbacks=st_g["persist_fallbacks"])


# shape, space order, shots, sources per shot: one and several tiles, ragged edges in both directions, rows whose length
# is not a multiple of 4, both radii, more shots than XCDs
CASES = [((70, 50), 8, 3, 1), ((61, 83), 4, 2, 2), ((45, 110), 8, 2, 1), ((30, 150), 4, 1, 1), ((50, 41), 8, 11, 1)]


@pytest.mark.parametrize("shape,so,nshots,nsrc", CASES)
def test_whole_loop_tti_equals_kernel_per_step_and_oracle(mifwi, shape, so, nshots, nsrc):
    prob, dt, nt, src, rec, w, rng = _setup(shape, so, nshots, nsrc)
    res = rng.standard_normal((nshots, nt, rec.shape[0])).astype(np.float32)
    a = _run(mifwi, prob, dt, nt, src, rec, w, res, 0)
    b = _run(mifwi, prob, dt, nt, src, rec, w, res, mifwi.FLAG_NO_GRAPH)
    assert a["fwd_kernel"] == "persist_tti2d_kernel" and a["adj_kernel"] == "persist_tti2d_kernel<BWD>" and \
        a["grad_kernel"] == "persist_tti2d_kernel<BWD>", a
    assert a["launches"] == ((nshots + 7) // 8, (nshots + 7) // 8) and a["fallbacks"] == 0
    assert b["fwd_kernel"] == "step2d_kernel" and b["launches"][0] == nt - 2
    assert np.array_equal(a["d0"], a["d1"])            # saving changes nothing
    for key, tol in (("d0", 2e-5), ("u3", 2e-5), ("r6", 2e-5), ("u_saved", 2e-5), ("a", 2e-5), ("g", 1e-4)):
        assert np.abs(b[key]).max() > 0, key
        assert rel_l2(a[key], b[key]) < tol, (key, rel_l2(a[key], b[key]))
    for s in {0, nshots - 1}:
        ref = prob.forward(src[s], w[s], rec, dt=dt)
        assert rel_l2(a["d0"][s], ref) <= 2e-5
        assert rel_l2(a["a"][s], prob.adjoint(src[s], rec, res[s], dt=dt)) <= 1e-4
        assert rel_l2(a["g"][s], prob.gradient(src[s], w[s], rec, res[s], dt=dt)) <= 1e-4


def test_abandoned_whole_loop_tti_is_repeated_with_a_kernel_per_step(mifwi, monkeypatch):
    prob, dt, nt, src, rec, w, rng = _setup((70, 50), 8, 3, nt=80)
    res = rng.standard_normal((3, nt, rec.shape[0])).astype(np.float32)
    # the same bits need the same tile shape: a plan that never runs whole loops re-tiles itself to 8 x 32 at the first
    # tti_set_model, and the compiler contracts the multiply-adds of each instance of the kernel in its own way
    monkeypatch.setenv("MIFWI_2D_TX", "32")
    want = _run(mifwi, prob, dt, nt, src, rec, w, res, mifwi.FLAG_NO_GRAPH)
    monkeypatch.delenv("MIFWI_2D_TX")
    monkeypatch.setenv("MIFWI_PERSIST_SPIN", "0")
    got = _run(mifwi, prob, dt, nt, src, rec, w, res, 0)
    assert got["fwd_kernel"] == "step2d_kernel" and got["fallbacks"] >= 1
    for key in ("d0", "d1", "a", "g", "u3", "r6"):
        assert np.array_equal(got[key], want[key]), key


def test_a_tti_shot_whose_xcd_takes_no_part_is_noticed(mifwi, monkeypatch):
    prob, dt, nt, src, rec, w, rng = _setup((70, 50), 8, 3, nt=60)
    res = rng.standard_normal((3, nt, rec.shape[0])).astype(np.float32)
    monkeypatch.setenv("MIFWI_2D_TX", "32")
    want = _run(mifwi, prob, dt, nt, src, rec, w, res, mifwi.FLAG_NO_GRAPH)
    monkeypatch.delenv("MIFWI_2D_TX")
    monkeypatch.setenv("MIFWI_PERSIST_DROP_SLOT", "1")
    got = _run(mifwi, prob, dt, nt, src, rec, w, res, 0)
    assert got["fwd_kernel"] == "step2d_kernel" and got["fallbacks"] >= 1
    for key in ("d0", "a", "g"):
        assert np.array_equal(got[key], want[key]), key
    assert np.abs(got["d0"][1]).max() > 0


def test_space_orders_without_a_whole_loop_kernel_keep_the_kernel_per_step(mifwi):
    prob, dt, nt, src, rec, w, rng = _setup((60, 44), 16, 2, nt=40)
    res = rng.standard_normal((2, nt, rec.shape[0])).astype(np.float32)
    got = _run(mifwi, prob, dt, nt, src, rec, w, res, 0)
    assert got["fwd_kernel"] == "step2d_kernel" and got["fallbacks"] == 0


def test_whole_loop_tti_repeats_bit_for_bit_when_tiles_share_a_cu(mifwi, monkeypatch):
    """Run-to-run identity with several resident tiles per CU.  The first version of the kernel overwrote the data
    registers of its 16-byte stores in the next issue slot; on gfx950 that reaches the store when another wave keeps
    the SIMD busy, so results differed from run to run as soon as two tiles shared a CU (any grid of more than 16
    tiles, i.e. the Marmousi-II benchmark), while every small test grid passed.  Surplus blocks (they leave at once)
    make the 16 tiles of this grid share CUs; without them every tile has a CU of its own."""
    prob, dt, nt, src, rec, w, rng = _setup((120, 120), 8, 1, 1, nt=200)
    res = rng.standard_normal((1, nt, rec.shape[0])).astype(np.float32)
    monkeypatch.setenv("MIFWI_TTI_SURPLUS_BLOCKS", "0")
    alone = _run(mifwi, prob, dt, nt, src, rec, w, res, 0)
    monkeypatch.setenv("MIFWI_TTI_SURPLUS_BLOCKS", "100")
    for _ in range(4):
        shared = _run(mifwi, prob, dt, nt, src, rec, w, res, 0)
        assert shared["fwd_kernel"] == "persist_tti2d_kernel" and shared["fallbacks"] == 0
        for key in ("d0", "d1", "u3", "a", "r6", "u_saved", "g"):
            assert np.array_equal(alone[key], shared[key]), key
