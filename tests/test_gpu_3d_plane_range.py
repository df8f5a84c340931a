"""Time loops of the 2.5-D kernel compute nothing where the wave has not arrived yet: the library keeps, per shot, the
box of points that hold a value other than zero, widens it by the stencil radius per step and leaves out the x planes
and the (y, z) tiles outside it (forward loops, recomputed segments, adjoint loops, gradient loops without illumination
sums).  The result must be the one of the full sweep, value for value: `MIFWI_3D_BOX=0` switches the skipping off.
The reference has no counterpart (Devito sweeps the whole grid every step); parity of the 3-D kernels with the CPU
restatement is tested in test_gpu_parity_abi.py and test_gpu_golden.py with the skipping on."""
import numpy as np
import pytest

from helpers import Case, smooth_random_vp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mifwi():
    import mifwi as m
    return m


def _long_case(scale=1.0, nt=44, seed=5):
    """A grid long in x with the sources at 20 % and 70 % of its length: 44 steps do not take the wave to the ends."""
    shape = (150, 22, 34)
    vp = (smooth_random_vp(shape, seed=seed) * scale).astype(np.float32)
    c = Case(shape, (10.0, 10.0, 10.0), nbl=5, space_order=8, nt=nt, nshots=2, nrec=13, seed=seed, vp=vp)
    dom = np.array([(n - 1) * 10.0 for n in shape], dtype=np.float32)
    c.src[0, 0, :] = dom * np.array([0.2, 0.5, 0.2], np.float32) + 3.3
    c.src[1, 0, :] = dom * np.array([0.7, 0.4, 0.25], np.float32) + 1.7
    return c


def _wide_case(nt=22, seed=9):
    """A grid of several tiles in y (16 rows each) and z (64 points each) with both sources in one corner region: most
    tiles stay out of the wave's reach for these steps."""
    shape = (40, 150, 246)
    c = Case(shape, (10.0, 10.0, 10.0), nbl=5, space_order=4, nt=nt, nshots=2, nrec=13, seed=seed)
    dom = np.array([(n - 1) * 10.0 for n in shape], dtype=np.float32)
    c.src[0, 0, :] = dom * np.array([0.5, 0.08, 0.05], np.float32) + 2.1
    c.src[1, 0, :] = dom * np.array([0.4, 0.12, 0.08], np.float32) + 4.4
    return c


def _run(mifwi, case, monkeypatch, box, budget_mb=None, dt=None, model_first=None, flags=0):
    monkeypatch.setenv("MIFWI_3D_BOX", "1" if box else "0")
    if budget_mb is None:
        monkeypatch.delenv("MIFWI_SAVE_BUDGET_MB", raising=False)
    else:
        monkeypatch.setenv("MIFWI_SAVE_BUDGET_MB", str(budget_mb))
    p = case.prob
    plan = mifwi.Plan(p.shape_pad, p.spacing, p.origin_pad, p.space_order, case.nt, dt or case.dt, case.nsrc, case.nrec,
                      max_shots=case.nshots, flags=flags)
    plan.set_geometry(case.src, case.rec)
    if model_first is not None:   # an evaluation with another model first: whatever it leaves behind must not show
        plan.set_model(model_first.vp, model_first.damp)
        r0 = plan.forward(case.wavelet, save=True)
        plan.gradient(r0)
    plan.set_model(p.vp, p.damp)
    rec_plain = np.array(plan.forward(case.wavelet))
    last = np.array(plan.read_wavefield(1, shot=1))
    rec = np.array(plan.forward(case.wavelet, save=True))
    st = plan.stats()
    saved = None if (st["ckpt_steps"] or flags & mifwi.FLAG_SAVE_F16) else np.array(plan.read_wavefield(0, shot=0))
    grad, illum = plan.gradient(rec, want_illum=True)
    grad, illum = np.array(grad), np.array(illum)
    # without the illumination sums the backward loop skips what its field has not reached as well
    grad_only = np.array(plan.gradient(rec))
    srca = np.array(plan.adjoint(rec))
    out = dict(rec_plain=rec_plain, last=last, rec=rec, saved=saved, grad=grad, illum=illum, grad_only=grad_only, srca=srca,
               ckpt_steps=st["ckpt_steps"])
    plan.close()
    return out


def _mb_of_levels(case, n):
    """Device memory of n saved levels of both shots, in MiB (halo of the space order included)."""
    field = np.prod(np.array(case.prob.shape_pad) + case.prob.space_order)
    return int(n * case.nshots * field * 4 / 2 ** 20) + 1


def _same(a, b, what):
    for k in ("rec_plain", "last", "rec", "saved", "grad", "illum", "grad_only", "srca"):
        if a[k] is None or b[k] is None:
            assert a[k] is None and b[k] is None, (what, k)
            continue
        assert np.array_equal(a[k], b[k]), (what, k, float(np.abs(a[k] - b[k]).max()))


def test_skipped_planes_equal_the_full_sweep(mifwi, monkeypatch):
    case = _long_case()
    full = _run(mifwi, case, monkeypatch, box=False)
    skip = _run(mifwi, case, monkeypatch, box=True)
    assert np.abs(full["rec"]).max() > 0 and np.abs(full["grad"]).max() > 0
    # the premise: after these steps the ends of the grid are still exactly zero (shot 1 sits at 70 % of the length)
    level = full["last"][(case.nt - 1) % 3]
    assert not level[:20].any() and level[90:130].any()
    _same(full, skip, "stored levels")
    assert np.array_equal(skip["grad"], skip["grad_only"])


def test_skipped_tiles_equal_the_full_sweep(mifwi, monkeypatch):
    case = _wide_case()
    full = _run(mifwi, case, monkeypatch, box=False)
    skip = _run(mifwi, case, monkeypatch, box=True)
    level = full["last"][(case.nt - 1) % 3]
    assert not level[:, 128:, :].any() and not level[:, :, 192:].any() and level[:, :48, :64].any()
    _same(full, skip, "tiles out of reach")
    ck = _run(mifwi, case, monkeypatch, box=True, budget_mb=_mb_of_levels(case, 20))
    assert ck["ckpt_steps"] > 0
    for k in ("rec_plain", "rec", "grad", "illum", "grad_only", "srca"):
        assert np.array_equal(full[k], ck[k]), k


def test_skipped_planes_with_checkpointed_levels(mifwi, monkeypatch):
    case = _long_case()
    full = _run(mifwi, case, monkeypatch, box=False)
    skip = _run(mifwi, case, monkeypatch, box=True, budget_mb=_mb_of_levels(case, 32))
    assert skip["ckpt_steps"] > 0
    for k in ("rec_plain", "rec", "grad", "illum", "grad_only", "srca"):
        assert np.array_equal(full[k], skip[k]), k


def test_a_wider_wavefield_of_an_earlier_evaluation_leaves_nothing_behind(mifwi, monkeypatch):
    case = _long_case()
    fast = _long_case(scale=1.6)
    dt = min(case.dt, fast.dt)
    full = _run(mifwi, case, monkeypatch, box=False, dt=dt)
    skip = _run(mifwi, case, monkeypatch, box=True, dt=dt, model_first=fast.prob)
    _same(full, skip, "after a faster model")
    ck = _run(mifwi, case, monkeypatch, box=True, dt=dt, model_first=fast.prob,
              budget_mb=_mb_of_levels(case, 32))
    assert ck["ckpt_steps"] > 0
    for k in ("rec_plain", "rec", "grad", "illum", "grad_only", "srca"):
        assert np.array_equal(full[k], ck[k]), k


def test_half_precision_levels_and_several_sources_per_shot(mifwi, monkeypatch):
    """The half-precision copy of a level gets its zeros too (an evaluation with a faster model first), and the box
    of the injected cells spans all sources of a shot."""
    case = _long_case()
    fast = _long_case(scale=1.6)
    dt = min(case.dt, fast.dt)
    full = _run(mifwi, case, monkeypatch, box=False, dt=dt, flags=mifwi.FLAG_SAVE_F16)
    skip = _run(mifwi, case, monkeypatch, box=True, dt=dt, model_first=fast.prob, flags=mifwi.FLAG_SAVE_F16)
    _same(full, skip, "half-precision levels")
    shape = (150, 22, 34)
    multi = Case(shape, (10.0, 10.0, 10.0), nbl=5, space_order=8, nt=40, nshots=2, nrec=9, nsrc=3, seed=21)
    dom = np.array([(n - 1) * 10.0 for n in shape], dtype=np.float32)
    multi.src[..., 0] = (0.15 + 0.1 * np.arange(3, dtype=np.float32))[None, :] * dom[0] + np.array([[0.0], [600.0]], np.float32)
    _same(_run(mifwi, multi, monkeypatch, box=False), _run(mifwi, multi, monkeypatch, box=True), "three sources per shot")
