"""End-to-end inversion on the GPU: the Marmousi workflow of the reference's marmousi_fwi.py
(observed + direct-wave data, bathymetry mask, illumination preconditioner, bounds, L-BFGS with
the bracketing line search) for a few iterations."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _example():
    spec = importlib.util.spec_from_file_location("marmousi_fwi_example",
                                                  os.path.join(ROOT, "examples", "marmousi_fwi.py"))
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
