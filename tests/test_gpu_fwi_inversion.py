"""End-to-end inversion on the GPU: the Marmousi workflow of the reference's marmousi_fwi.py
(observed + direct-wave data, bathymetry mask, illumination preconditioner, bounds, L-BFGS with
the bracketing line search) for a few iterations."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _example(name="marmousi_fwi"):
    spec = importlib.util.spec_from_file_location(name + "_example", os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_marmousi_lbfgs_reduces_misfit(tmp_path, capsys):
    ex = _example()
    vp, err0, err1 = ex.main(["--maxiter", "4", "--nsrc", "6", "--tn", "2500", "--odir", str(tmp_path)])
    log = tmp_path / "log0"
    misfit = np.loadtxt(log / "misfit")
    assert misfit.shape == (4, 2)
    assert np.all(np.diff(misfit[:, 0]) < 0), misfit
    assert misfit[-1, 0] < 0.8 * misfit[0, 0]
    assert vp.shape == (300, 106) and vp.min() >= 1.5 - 1e-6 and vp.max() <= 5.2 + 1e-6
    assert np.all(vp[:, :7] == pytest.approx(1.5, abs=1e-3))        # water column is masked out
    for name in ("optim_info", "fval", "step_length", "sim_count", "theta"):
        assert (log / name).exists()
    assert (log / "model_est" / "v_0").stat().st_size == 300 * 106 * 4


def test_marmousi_w2_and_nlcg_run(tmp_path, capsys):
    ex = _example()
    vp, err0, err1 = ex.main(["--maxiter", "2", "--nsrc", "3", "--tn", "1500", "--misfit", "1",
                              "--method", "NLCG", "--odir", str(tmp_path)])
    misfit = np.loadtxt(tmp_path / "log1" / "misfit")
    assert misfit.shape == (2, 2) and misfit[1, 0] < misfit[0, 0]
    assert np.isfinite(vp).all()


def test_marmousi_2d_wasserstein_runs(tmp_path, capsys):
    """--misfit 2 of the reference's drivers (marmousi_fwi.py:130-140): the whole gather as one density,
    back-and-forth optimal transport for the adjoint source."""
    ex = _example()
    vp, err0, err1 = ex.main(["--maxiter", "2", "--nsrc", "3", "--tn", "1500", "--misfit", "2",
                              "--odir", str(tmp_path)])
    misfit = np.loadtxt(tmp_path / "log2" / "misfit")
    assert misfit.shape == (2, 2) and misfit[1, 0] < misfit[0, 0]
    assert np.isfinite(vp).all()


def test_marmousi2_driver_and_first_gradient(tmp_path, capsys):
    """marmousi2_fwi.py:61-179 through the same example (--model SMARM2), with the first-gradient dump."""
    ex = _example()
    vp, err0, err1 = ex.main(["--model", "SMARM2", "--maxiter", "2", "--nsrc", "4", "--tn", "2000",
                              "--check-gradient", "1", "--odir", str(tmp_path)])
    g = np.fromfile(tmp_path / "1st_grad_0", dtype=np.float64)
    assert g.size == 340 * 140 and np.isfinite(g).all() and np.abs(g).max() > 0
    assert not g.reshape(340, 140)[:, :15].any()                     # bathymetry mask
    misfit = np.loadtxt(tmp_path / "log0" / "misfit")
    assert misfit.shape == (2, 2) and misfit[1, 0] < misfit[0, 0]
    assert vp.shape == (340, 140) and vp.max() <= 5.0 + 1e-6


def test_forward_modelling_driver(tmp_path, capsys):
    """marmousi_fm.py:30-113: observed, synthetic and direct-wave gathers written as raw float32."""
    ex = _example("marmousi_fm")
    gathers = ex.main(["--nsrc", "3", "--tn", "1200", "--odir", str(tmp_path)])
    nt, nrec = gathers["obs"][0].data.shape
    assert nrec == 300
    for name in ("obs", "syn", "direct"):
        for i in range(3):
            d = np.fromfile(tmp_path / "data" / (name + str(i)), dtype=np.float32).reshape(nt, nrec)
            assert np.array_equal(d, gathers[name][i].data) and np.abs(d).max() > 0
    assert not np.array_equal(gathers["obs"][1].data, gathers["syn"][1].data)


def test_circle_driver(tmp_path, capsys):
    ex = _example("circle_fwi")
    vp, err0, err1 = ex.main(["--maxiter", "2", "--nsrc", "3", "--odir", str(tmp_path)])
    misfit = np.loadtxt(tmp_path / "log0" / "misfit")
    assert misfit.shape == (2, 2) and misfit[1, 0] < misfit[0, 0]
    assert vp.shape == (201, 201) and vp.min() >= 2.5 - 1e-6 and vp.max() <= 4.0 + 1e-6


def test_circle_workflow_with_resampled_traces(tmp_path, capsys):
    """The set-up of the reference's circle_fwi.py:60-168 with fewer shots: demo_model ignores `dt`, so the
    solver steps with the CFL time step (different for the true and the initial model) while the misfit
    is taken on traces resampled to 1 ms (geometry.resample), which sends every shot through the
    resampling branch of fwi_obj_single (fwi.py:140-160).  With these sizes both resampled axes have
    1003 samples, as in the reference's own run."""
    from seismic import demo_model, AcquisitionGeometry
    from fwi import fm_multi, fwi_loss
    from misfit import least_square
    from optimize import LBFGS
    from minimize import minimize
    shape, spacing, nbl, so = (201, 201), (10., 10.), 40, 6
    kw = dict(origin=(0, 0), shape=shape, spacing=spacing, space_order=so, nbl=nbl, dt=1.)
    true_model = demo_model('circle-isotropic', vp_circle=3.6, vp_background=3, r=60, **kw)
    init_model = demo_model('circle-isotropic', vp_circle=3, vp_background=3, r=60, **kw)
    assert float(true_model.critical_dt) > 1.0                      # dt=1 was not taken over
    nsrc = 3
    src = np.empty((nsrc, 2))
    src[:, 1] = np.linspace(0, true_model.domain_size[0], num=nsrc)
    src[:, 0] = 20.
    rec = np.empty((shape[0], 2))
    rec[:, 1] = np.linspace(spacing[0], true_model.domain_size[0] - spacing[0], num=shape[0])
    rec[:, 0] = 1980.
    g1 = AcquisitionGeometry(true_model, rec, src, 0., 1000., f0=0.010, src_type='Ricker')
    g0 = AcquisitionGeometry(init_model, rec, src, 0., 1000., f0=0.010, src_type='Ricker')
    g1.resample(1.)
    g0.resample(1.)
    obs = fm_multi(g1, save=False)
    # observed data live on the modelling axis (CFL step), not on the resampled one
    assert obs[0].data.shape[0] != g1.nt and np.isclose(obs[0].time_range.step, float(true_model.critical_dt))
    m0 = 1. / (np.full(shape, 3.0).reshape(-1).astype(np.float64)) ** 2
    f0_, grad, res = fwi_loss(m0, g0, obs, least_square, None, None, True)
    assert f0_ > 0 and np.isfinite(grad).all() and np.abs(grad).max() > 0
    log = tmp_path / "log"
    opt = LBFGS(memory=5, ls_method='Bracket', step_len_init=0.05, max_ls=5, log_path=str(log))
    m = minimize(opt, maxIter=2, ftol=1e-5, gtol=1e-10, log_path=str(log)).run(
        m0, g0, obs, least_square, None, None, True, [1.0 / 4.0 ** 2, 1.0 / 2.0 ** 2])
    misfit = np.loadtxt(log / "misfit")
    assert misfit[1, 0] < misfit[0, 0]
    assert m.shape == m0.shape
