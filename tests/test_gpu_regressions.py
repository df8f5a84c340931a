"""
Regression tests of the C ABI's state handling on the GPU: asynchronous plans, ordering against
torch's streams, lazily downloaded wavefields, half-precision scales, checkpoint recomputation
inputs and the legacy default stream.  Checker: the library's own synchronous results (bit for
bit) or the CPU oracle at the tolerances of test_gpu_parity_abi.py.
"""
import time

import numpy as np
import pytest

from helpers import Case, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mifwi():
    import mifwi as m
    return m


def test_async_calls_return_before_the_work_is_done(mifwi, monkeypatch):
    """MIFWI_FLAG_ASYNC: a propagation call only enqueues its time loop.  The host returns long before the
    GPU has finished, two plans can be kept busy from one thread, and times and results are there after
    mifwi_synchronize / mifwi_get_stats.  (The synchronous reference runs a kernel per step too.)"""
    monkeypatch.setenv("MIFWI_PERSIST", "0")
    case = Case((300, 260), (10.0, 10.0), nbl=20, space_order=8, nt=4000, nshots=4, nrec=50, seed=2)
    ref = case.plan(mifwi)
    ref.forward(case.wavelet)            # first run: graph capture, cold clocks
    rec_ref = ref.forward(case.wavelet)
    loop_ms = ref.stats()["fwd_ms"]
    ref.close()
    assert loop_ms > 4.0, loop_ms      # long enough for the comparison below to mean something
    import torch
    dev = torch.device("cuda")
    w = torch.from_numpy(case.wavelet).to(dev)
    plans = [case.plan(mifwi, flags=mifwi.FLAG_ASYNC) for _ in range(2)]
    recs = [torch.empty((4, case.nt, case.nrec), dtype=torch.float32, device=dev) for _ in plans]
    for p, r in zip(plans, recs):       # first call captures the graph: not timed
        p.forward(w, rec_out=r)
        p.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p, r in zip(plans, recs):
        p.forward(w, rec_out=r)
    host_ms = (time.perf_counter() - t0) * 1e3
    for p in plans:
        p.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    print("loop %.2f ms on the GPU; both asynchronous calls returned after %.2f ms, done after %.2f ms"
          % (loop_ms, host_ms, wall_ms))
    assert host_ms < 0.4 * loop_ms, (host_ms, loop_ms)
    assert wall_ms > 0.9 * loop_ms      # the work did take its time: the calls had returned before it was done
    for p, r in zip(plans, recs):
        st = p.stats()
        assert 0.5 * loop_ms < st["fwd_ms"] < 4.0 * loop_ms and st["kernel_ms"] == st["fwd_ms"]
        assert np.array_equal(r.cpu().numpy(), rec_ref)
        p.close()


def test_async_fused_call_reports_the_sum_of_both_loops(mifwi):
    case = Case((60, 50), (10.0, 10.0), nbl=10, space_order=4, nt=200, nshots=2, nrec=9, seed=5)
    sync = case.plan(mifwi, flags=mifwi.FLAG_NO_GRAPH)   # a kernel per step, as asynchronous plans run
    obs = 0.5 * sync.forward(case.wavelet)
    out_ref = sync.l2_misfit_gradient(case.wavelet, obs)
    sync.close()
    import torch
    dev = torch.device("cuda")
    plan = case.plan(mifwi, flags=mifwi.FLAG_ASYNC)
    g = torch.zeros((2,) + case.prob.shape_pad, dtype=torch.float32, device=dev)
    il = torch.zeros_like(g)
    f = torch.zeros(2, dtype=torch.float64, device=dev)
    plan.l2_misfit_gradient(torch.from_numpy(case.wavelet).to(dev), torch.from_numpy(obs).to(dev), grad=g, illum=il,
                            fval=f)
    st = plan.stats()     # blocks until both loops have run
    assert st["fwd_ms"] > 0 and st["bwd_ms"] > 0
    assert abs(st["kernel_ms"] - (st["fwd_ms"] + st["bwd_ms"])) < 1e-9
    plan.synchronize()
    assert np.array_equal(g.cpu().numpy(), out_ref["grad"])
    assert np.array_equal(f.cpu().numpy(), out_ref["fval"])
    plan.close()


def test_inputs_produced_by_queued_torch_work_are_waited_for(mifwi):
    """A residual that torch is still computing when plan.gradient is called (a long queue of kernels on
    torch's stream ends in the tensor handed over) must be complete when the plan's own stream reads it."""
    import torch
    case = Case((120, 100), (10.0, 10.0), nbl=10, space_order=8, nt=300, nshots=2, nrec=64, seed=3)
    plan = case.plan(mifwi)
    rec = plan.forward(case.wavelet, save=True)
    g_ref = plan.gradient(rec)
    dev = torch.device("cuda")
    base = torch.from_numpy(rec).to(dev)
    big = torch.randn(4096, 4096, device=dev)
    torch.cuda.synchronize()
    for trial in range(3):
        acc = big
        for _ in range(20):                  # tens of milliseconds of queued work
            acc = acc @ big * 1e-2
        # depends on the queue above: zero only once it has run (0 * finite), and then added to the data
        res = base + 0.0 * torch.nan_to_num(acc[:1, :1]).sum()
        g = plan.gradient(res)
        assert np.array_equal(g, g_ref), trial
    plan.close()


def test_lazy_wavefield_of_a_non_saving_run_survives_a_saving_run(mifwi):
    """forward(save=False) hands back a wavefield that stays on the GPU until read.  A later saving run on the
    same cached plan steps on the same rotating levels: the first wavefield must have been fetched by then."""
    from seismic import demo_model, AcquisitionGeometry
    from seismic.acoustic import AcousticWaveSolver
    model = demo_model('layers-isotropic', shape=(61, 55), spacing=(10., 10.), nbl=10, space_order=4, nlayers=3)
    rec_pos = np.stack([np.linspace(0, 600, 21), np.full(21, 20.0)], 1)
    geom = AcquisitionGeometry(model, rec_pos, np.array([[300.0, 20.0]]), 0.0, 300.0, f0=0.015, src_type='Ricker')
    solver = AcousticWaveSolver(model, geom, space_order=4)
    vp_a = np.array(model.vp.data)
    # reference copy of what u1 must hold: the same run again, read at once
    _, u1b, _ = solver.forward(vp=vp_a, save=False)
    expect = np.array(u1b.data)
    _, u1, _ = solver.forward(vp=vp_a, save=False)
    vp_b = vp_a * 0.9
    _, u2, _ = solver.forward(vp=vp_b, save=True)
    assert np.array_equal(np.array(u1.data), expect)
    assert np.array(u2.data).shape[0] == geom.nt


def test_half_precision_scale_ignores_unused_injection_slots(mifwi):
    """Two sources of a shot in one cell share its corners, so the injection table has fewer cells than
    slots; the scale of the half-precision saved levels must come from the used slots only."""
    case = Case((70, 60), (10.0, 10.0), nbl=10, space_order=8, nt=220, nshots=2, nrec=12, nsrc=2, seed=9)
    case.src[:, 1] = case.src[:, 0] + np.float32(1.5)       # same cell
    res = np.random.default_rng(1).standard_normal((2, case.nt, 12)).astype(np.float32)
    full = case.plan(mifwi)
    full.forward(case.wavelet, save=True)
    g_ref = full.gradient(res)
    full.close()
    for trial in range(2):       # the second plan reuses freed (dirty) device memory
        half = case.plan(mifwi, flags=mifwi.FLAG_SAVE_F16)
        half.forward(case.wavelet, save=True)
        g = half.gradient(res)
        for s in range(2):
            assert rel_l2(g[s], g_ref[s]) < 5e-3, (trial, s, rel_l2(g[s], g_ref[s]))   # white-noise residual: worst case for the rounding
        half.close()


@pytest.mark.parametrize("flags", [0, 4])
def test_checkpoints_are_recomputed_with_the_wavelet_that_took_them(mifwi, flags):
    """Other modelling between forward(save=1) and the gradient must not leak into the recomputed segments
    (a kept copy of the wavelet); a new model invalidates the checkpoints with a clear error."""
    case = Case((50, 44), (10.0, 10.0), nbl=8, space_order=8, nt=131, nshots=2, nrec=15, seed=22)
    plan = case.plan(mifwi, flags=flags | mifwi.FLAG_CHECKPOINT)
    rec = plan.forward(case.wavelet, save=True)
    g_ref = plan.gradient(rec)
    rec2 = plan.forward(case.wavelet, save=True)
    assert np.array_equal(rec, rec2)
    plan.forward(3.0 * case.wavelet[::-1].copy(), save=False)      # another wavelet, nothing saved
    g = plan.gradient(rec)
    assert np.array_equal(g, g_ref)
    plan.set_model(case.prob.vp * 1.05, case.prob.damp)
    with pytest.raises(mifwi.MifwiError, match="set_model"):
        plan.gradient(rec)
    plan.forward(case.wavelet, save=True)
    assert np.abs(plan.gradient(rec)).max() > 0
    plan.close()


def test_legacy_default_stream(mifwi):
    """mifwi_set_stream(NULL) selects the legacy default stream (torch's current stream unless told
    otherwise); it cannot be captured into a graph, so the loops are launched kernel by kernel."""
    case = Case((45, 38), (10.0, 12.5), nbl=9, space_order=8, nt=120, nshots=2, nrec=13, seed=4)
    ref = case.plan(mifwi)
    rec_ref = ref.forward(case.wavelet, save=True)
    g_ref = ref.gradient(rec_ref)
    ref.close()
    plan = case.plan(mifwi)
    plan.set_stream(0)
    rec = plan.forward(case.wavelet, save=True)
    g = plan.gradient(rec)
    assert np.array_equal(rec, rec_ref) and np.array_equal(g, g_ref)
    plan.close()
