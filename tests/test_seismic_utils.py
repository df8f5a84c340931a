"""
Host-side data model tests, modelled on the reference's seismic/test_seismic_utils.py:17-97
(damp centre value and re-initialisation, user dt, default geometry, resampling, naming) plus
checks of the numpy mirror against the oracle's restatement of the same formulas.
"""
import numpy as np
import pytest

from seismic import Model, setup_geometry, AcquisitionGeometry, demo_model, TimeAxis, RickerSource
from oracle import oracle as orc


def not_bcs(bc):
    return ("mask", 1) if bc == "damp" else ("damp", 0)


@pytest.mark.parametrize('nbl, bcs', [(20, ("mask", 1)), (0, ("mask", 1)), (20, ("damp", 0)), (0, ("damp", 0))])
def test_damp(nbl, bcs):
    shape = (21, 21)
    vp = np.ones(shape)
    model = Model((0, 0), (10, 10), shape, 4, vp, nbl=nbl, bcs=bcs[0])
    try:
        center = model.damp.data[tuple(s // 2 for s in model.damp.shape)]
    except AttributeError:
        center = model.damp
    assert all([s == s0 + 2 * nbl for s, s0 in zip(model.vp.shape, shape)])
    assert center == bcs[1]
    switch_bcs = not_bcs(bcs[0])
    model._initialize_bcs(bcs=switch_bcs[0])
    try:
        center = model.damp.data[tuple(s // 2 for s in model.damp.shape)]
    except AttributeError:
        center = model.damp
    assert center == switch_bcs[1]


@pytest.mark.parametrize('shape', [(41,), (21, 21), (11, 11, 11)])
def test_default_geom(shape):
    vp = np.ones(shape)
    o = tuple([0] * len(shape))
    d = tuple([10] * len(shape))
    model = Model(o, d, shape, 4, vp, nbl=20, dt=1)
    assert model.critical_dt == 1
    geometry = setup_geometry(model, 250)
    nrec = shape[0] * (shape[1] if len(shape) > 2 else 1)
    assert geometry.grid == model.grid
    assert geometry.nrec == nrec
    assert geometry.nsrc == 1
    assert geometry.src_type == "Ricker"
    assert geometry.rec.shape == (251, nrec)
    assert np.linalg.norm(geometry.rec.data) == 0
    assert geometry.src.shape == (251, 1)
    assert np.linalg.norm(geometry.new_src(src_type=None).data) == 0
    rec2 = geometry.rec.resample(num=501)
    assert rec2.shape == (501, nrec)
    assert rec2.grid == model.grid
    assert geometry.new_rec(name="bonjour").name == "bonjour"
    assert geometry.new_src(name="bonjour").name == "bonjour"


@pytest.mark.parametrize('shape', [(41,), (21, 21), (11, 11, 11)])
def test_geom(shape):
    vp = np.ones(shape)
    o = tuple([0] * len(shape))
    d = tuple([10] * len(shape))
    model = Model(o, d, shape, 4, vp, nbl=20, dt=1)
    nrec, nsrc = 31, 4
    geometry = AcquisitionGeometry(model, np.ones((nrec, len(shape))), np.ones((nsrc, len(shape))), t0=0.0, tn=250)
    assert geometry.nrec == nrec and geometry.nsrc == nsrc and geometry.src_type is None
    assert geometry.rec.shape == (251, nrec)
    assert geometry.src.shape == (251, nsrc)
    assert np.linalg.norm(geometry.src.data) == 0
    assert geometry.rec.resample(num=501).shape == (501, nrec)


def test_user_dt_above_cfl_raises():
    model = Model((0, 0), (10, 10), (21, 21), 4, np.full((21, 21), 3.0), nbl=5, dt=50.0)
    with pytest.raises(ValueError):
        model.critical_dt


@pytest.mark.parametrize('so, ndim', [(2, 2), (4, 2), (8, 2), (16, 2), (4, 3), (8, 3)])
def test_model_matches_oracle_formulas(so, ndim):
    shape = (23, 19, 17)[:ndim]
    spacing = (10.0, 12.5, 7.5)[:ndim]
    rng = np.random.default_rng(so)
    vp = (1.5 + 2 * rng.random(shape)).astype(np.float32)
    model = Model((0.,) * ndim, spacing, shape, so, vp, nbl=7, bcs="damp")
    prob = orc.Problem(vp, spacing, 7, so)
    assert model.critical_dt == prob.critical_dt
    np.testing.assert_array_equal(model.vp.data, prob.vp)
    np.testing.assert_allclose(model.damp.data, prob.damp, rtol=2e-6, atol=0)
    assert tuple(float(o) for o in model.grid.origin) == prob.origin_pad


def test_time_axis_and_ricker_match_oracle():
    ta = TimeAxis(start=0.0, stop=1000.0, step=2.041)
    nt, tv = orc.time_axis(0.0, 1000.0, 2.041)
    assert ta.num == nt == 491
    np.testing.assert_array_equal(ta.time_values, tv)
    model = demo_model('constant-isotropic', space_order=4)
    assert model.critical_dt == np.float32(3.698)
    geom = setup_geometry(model, 1000.)
    src = geom.src
    assert isinstance(src, RickerSource)
    ref = orc.ricker(0.010, src.time_values).astype(np.float32)
    np.testing.assert_array_equal(src.data[:, 0], ref)
    with pytest.raises(ValueError):
        TimeAxis(start=0, stop=1, step=0.1, num=11)


def test_update_repads_and_m():
    model = demo_model('circle-isotropic', nbl=6, shape=(31, 31))
    new = np.full((31, 31), 2.0, dtype=np.float64)
    new[0, :] = 3.0
    model.update('vp', new)
    assert model.vp.data.dtype == np.float32
    assert np.all(model.vp.data[:7, :] == 3.0) and np.all(model.vp.data[7:, :] == 2.0)
    np.testing.assert_allclose(model.m, 1 / model.vp.data ** 2)
    with pytest.raises(ValueError):
        model.update('vp', np.ones((5, 5)))


def test_qwasserstein_1d_closed_forms():
    """Shifted positive bumps: W2^2/2 = shift^2/2 in normalised time; identical signals: zero
    loss and zero adjoint source; the adjoint source correlates with the finite-difference
    derivative (the reference's expression, misfit/misfit.py:64-68, is a scaled approximation)."""
    from misfit import qWasserstein, least_square, Misfit
    n = 801
    t = np.linspace(0, 1, n)

    def bump(c):
        return np.exp(-0.5 * ((t - c) / 0.03) ** 2)
    w = qWasserstein(trans_type='none', method='1d')
    loss, adj = w(bump(0.4), bump(0.55))
    assert loss == pytest.approx(0.5 * 0.15 ** 2, rel=2e-3)
    loss0, adj0 = w(bump(0.4) + 0.05, bump(0.4) + 0.05)     # strictly increasing CDFs
    assert loss0 == pytest.approx(0, abs=1e-12) and np.abs(adj0).max() < 1e-9
    # gathers: column-wise sum, shape preserved, 'linear' makes signed data positive
    wl = qWasserstein(trans_type='linear', gamma=1.01, method='1d')
    f = np.stack([bump(0.3) - 0.2 * bump(0.5), bump(0.6) - 0.1 * bump(0.2)], axis=1)
    g = np.stack([bump(0.35) - 0.2 * bump(0.55), bump(0.5) - 0.1 * bump(0.1)], axis=1)
    loss2, adj2 = wl(f, g)
    assert adj2.shape == f.shape and loss2 > 0
    eps, fd = 1e-6, np.zeros(40)
    idx = np.linspace(100, 700, 40).astype(int)
    for k, i in enumerate(idx):
        fp, fm = f.copy(), f.copy()
        fp[i, 0] += eps
        fm[i, 0] -= eps
        fd[k] = (wl(fp, g)[0] - wl(fm, g)[0]) / (2 * eps)
    # the shift constant depends on min(f, g) only, so the perturbation leaves it unchanged
    corr = np.corrcoef(fd, adj2[idx, 0])[0, 1]
    assert corr > 0.9
    w2d = qWasserstein(gamma=1.01, method='2d', num_steps=15, step_scale=1.)     # as the drivers build it
    loss2d, adj2d = w2d(f, g)                                                     # details: tests/test_w2_2d.py
    assert loss2d > 0 and adj2d.shape == f.shape and np.isfinite(adj2d).all()
    assert Misfit(least_square)(np.ones(3), np.zeros(3))[0] == pytest.approx(1.5)


def test_plotting_helpers_render(tmp_path):
    """The drivers' figures (seismic/plotting.py of the reference) can be drawn headless."""
    matplotlib = pytest.importorskip("matplotlib")
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from seismic import demo_model, plot_velocity, plot_shotrecord, plot_image, plot_perturbation
    model = demo_model('layers-isotropic', shape=(40, 30), spacing=(10., 10.), nbl=5, space_order=4)
    other = demo_model('constant-isotropic', shape=(40, 30), spacing=(10., 10.), nbl=5, space_order=4, vp=2.0)
    src = np.array([[200., 20.]])
    rec = np.stack([np.linspace(0, 390, 12), np.full(12, 20.)], axis=1)
    plot_velocity(model, source=src, receiver=rec, show=False)
    plt.savefig(tmp_path / "v.png"); plt.clf()
    plot_perturbation(model, model, show=False); plt.clf()
    plot_shotrecord(np.random.default_rng(0).standard_normal((50, 12)), model, 0., 500., show=False); plt.clf()
    plot_image(np.asarray(model.vp.data)[model.physical_slices], cmap="jet", show=False)
    plt.savefig(tmp_path / "i.png"); plt.clf()
    assert (tmp_path / "v.png").stat().st_size > 1000 and (tmp_path / "i.png").stat().st_size > 1000
    del other


@pytest.mark.parametrize("trans", ["linear", "square", "exp", "softplus"])
def test_qwasserstein_batched_form_matches_trace_loop(trans):
    """qWasserstein.batch_device (all traces of all shots in tensor operations) against the trace-by-trace
    host loop of misfit/misfit.py:49-104, on CPU tensors: losses to 1e-9, adjoint sources to 1e-6 (where a
    density is ~1e-80 the cumulative sums are flat and the optimal map is ill-conditioned, so the two
    summation orders differ there by more than rounding)."""
    import torch
    from misfit import qWasserstein
    rng = np.random.default_rng(4)
    ns, nt, ntr = 3, 200, 7
    t = np.linspace(0, 1, nt)[None, :, None]
    c = 0.3 + 0.4 * rng.random((ns, 1, ntr))
    f = np.exp(-((t - c) / 0.05) ** 2) * np.cos(40 * (t - c)) * (1 + rng.random((ns, 1, ntr)))
    g = np.exp(-((t - c - 0.08) / 0.06) ** 2) * np.cos(35 * (t - c - 0.08))
    f[1, :, 2] = g[1, :, 2]                                   # one identical pair: zero loss
    w2 = qWasserstein(trans_type=trans, gamma=1.01, method='1d')
    loss_b, grad_b = w2.batch_device(torch.from_numpy(f), torch.from_numpy(g))
    for k in range(ns):
        loss, grad = w2(f[k], g[k])
        assert abs(float(loss_b[k]) - loss) <= 1e-9 * max(1e-3, abs(loss))
        assert np.linalg.norm(grad_b[k].numpy() - grad) <= 1e-6 * np.linalg.norm(grad)


@pytest.mark.parametrize("order", [1, 3, 5])
def test_spline_resample_equals_per_trace_fit(order):
    """seismic.source.spline_resample (one cached sparse operator per pair of time axes) against the
    per-trace splrep/splev of the reference (source.py:160-166, fwi.py:47-57): 1e-12 relative."""
    from scipy import interpolate
    from seismic.source import spline_resample
    rng = np.random.default_rng(6)
    t0 = np.arange(700) * 0.6897
    t = np.arange(0, 482) * 1.0
    x = np.cumsum(rng.standard_normal((700, 9)), axis=0).astype(np.float32)
    ref = np.stack([interpolate.splev(t, interpolate.splrep(t0, x[:, i], k=order)) for i in range(9)], axis=1)
    for _ in range(2):                                      # second call takes the cached operator
        out = spline_resample(x, t0, t, order)
        assert out.shape == ref.shape and np.abs(out - ref).max() <= 1e-12 * np.abs(ref).max()
    # and through the Receiver object
    from seismic import Receiver, TimeAxis, demo_model
    model = demo_model('constant-isotropic', shape=(21, 21), spacing=(10., 10.), nbl=2)
    rec = Receiver(name='r', grid=model.grid, time_range=TimeAxis(start=0., step=0.6897, num=700),
                   coordinates=np.zeros((9, 2)))
    rec.data[:] = x
    rs = rec.resample(dt=1.0)
    tt = rs.time_values
    ref2 = np.stack([interpolate.splev(tt, interpolate.splrep(t0, x[:, i], k=3)) for i in range(9)], axis=1)
    assert np.abs(rs.data - ref2).max() <= 1e-5 * np.abs(ref2).max()
