"""The optimisers and the iteration loop against trajectories of the reference's own optimize
package (tests/golden/optimize_traj.npz, made by tests/golden/make_optimize_golden.py)."""
import os

import numpy as np
import pytest

import optimize
from minimize import minimize, divides
from optimize.math import backtrack2, polyfit2, angle
from optimize_driver import CASES, PROBLEMS, drive


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "optimize_traj.npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_trajectory_matches_reference(name, golden, tmp_path, capsys):
    problem, cls, kwargs, iters, bounds = CASES[name]
    fg, m0 = PROBLEMS[problem]()
    opt = getattr(optimize, cls)(log_path=str(tmp_path), **kwargs)
    models, trials, fvals = drive(opt, fg, m0.copy(), iters, bounds)
    ref_trials = golden[name + "/trials"]
    assert trials.shape == ref_trials.shape
    np.testing.assert_array_equal(trials[:, 0], ref_trials[:, 0])
    np.testing.assert_allclose(trials[:, 1:], ref_trials[:, 1:], rtol=1e-10)
    np.testing.assert_allclose(models, golden[name + "/models"], rtol=1e-10)
    np.testing.assert_allclose(fvals, golden[name + "/fvals"], rtol=1e-10)
    # the text logs
    assert (tmp_path / "optim_info").read_text() == str(golden[name + "/optim_info"])
    for stat in ("fval", "step_length", "step_count", "theta", "restarted", "slope"):
        got = np.loadtxt(tmp_path / stat, ndmin=1)
        np.testing.assert_allclose(got, golden[name + "/stat_" + stat], rtol=1e-5, atol=1e-12,
                                   err_msg=stat)


def test_lbfgs_secant_pairs_converge_faster(tmp_path):
    fg, m0 = PROBLEMS["quadratic"]()
    end = {}
    for pairs in ("reference", "secant"):
        opt = optimize.LBFGS(memory=5, max_ls=8, log_path=str(tmp_path / pairs), pairs=pairs)
        _, _, fvals = drive(opt, fg, m0.copy(), 12)
        end[pairs] = fvals[-1]
    assert end["secant"] < 0.1 * end["reference"]


def test_helpers():
    assert backtrack2(1.0, -1.0, 1.0, 2.0) == pytest.approx(0.25)
    assert backtrack2(1.0, -1.0, 1.0, 0.5) == 0.5          # upper safeguard b2*x1
    assert backtrack2(1.0, -0.01, 1.0, 50.0) == 0.1        # lower safeguard b1*x1
    x = np.array([0.0, 1.0, 2.0])
    assert polyfit2(x, (x - 1.2) ** 2) == pytest.approx(1.2)
    assert angle(np.array([1.0, 0.0]), np.array([0.0, 2.0])) == pytest.approx(np.pi / 2)
    assert divides(10, 5) and not divides(3, 2) and not divides(4, 0)


def test_line_search_failure_paths(tmp_path):
    ls = optimize.line_search.Bracket(step_count_max=2, path=str(tmp_path))
    alpha, status = ls.initialize(0.0, 1.0, 4.0, -4.0)
    assert (alpha, status) == (0.25, 0)
    # objective keeps rising: backtracks until the trial budget is spent, then fails
    for _ in range(3):
        alpha, status = ls.update(alpha, 2.0)
    assert status == -1 and alpha == 0
    # the maximum step length ends a growing search with success
    ls = optimize.line_search.Bracket(step_count_max=5, step_len_max=0.3, path=str(tmp_path))
    alpha, status = ls.initialize(0.0, 1.0, 4.0, -4.0)
    alpha, status = ls.update(alpha, 0.5)
    assert (alpha, status) == (0.3, 1)


def _loss_from(fg):
    calls = {"grad": 0, "nograd": 0}

    def loss(m, geometry, obs, misfit, direct_wave=None, mask=None, precond=True, calc_grad=True):
        f, g = fg(m)
        calls["grad" if calc_grad else "nograd"] += 1
        return f, (g if calc_grad else None), [np.arange(6.0).reshape(3, 2) * f]
    return loss, calls


def test_minimize_loop_files_and_iterates(golden, tmp_path, capsys):
    name = "lbfgs_bounds_quadratic"
    problem, cls, kwargs, iters, bounds = CASES[name]
    fg, m0 = PROBLEMS[problem]()
    loss, calls = _loss_from(fg)
    log = tmp_path / "log"
    opt = getattr(optimize, cls)(log_path=str(log), **kwargs)
    runner = minimize(opt, maxIter=iters, ftol=1e-9, log_path=str(log), save_model_freq=3,
                      save_grad_freq=2, save_res_freq=4, loss_func=loss)
    m = runner.run(m0.copy(), None, None, None, bounds=list(bounds))
    np.testing.assert_allclose(m, golden[name + "/models"][-1], rtol=1e-10)
    assert calls["grad"] == iters and calls["nograd"] == len(golden[name + "/trials"])
    misfit = np.loadtxt(log / "misfit")
    np.testing.assert_allclose(misfit[:, 0], golden[name + "/fvals"], rtol=2e-3)
    assert sorted(os.listdir(log / "model_est")) == ["v_0", "v_3", "v_6"]
    assert sorted(os.listdir(log / "gradient")) == ["g_0", "g_2", "g_4", "g_6"]
    assert sorted(os.listdir(log / "residual")) == ["0", "4"]
    v = np.fromfile(log / "model_est" / "v_6", dtype=np.float32)
    np.testing.assert_allclose(v, 1 / np.sqrt(golden[name + "/models"][7]), rtol=1e-6)
    sims = np.loadtxt(log / "sim_count")
    steps = np.loadtxt(log / "step_count")
    # 3 simulations + trials for a first (or restarted) direction, 2 + trials otherwise
    extra = sims - steps
    assert extra[0] == 3 and extra[1] == 2 and set(extra) <= {2.0, 3.0}


def test_minimize_stops_on_ftol(tmp_path, capsys):
    fg, m0 = PROBLEMS["quadratic"]()
    loss, calls = _loss_from(fg)
    opt = optimize.LBFGS(memory=5, max_ls=8, log_path=str(tmp_path), pairs="secant")
    runner = minimize(opt, maxIter=50, ftol=1e-2, log_path=str(tmp_path), loss_func=loss)
    m = runner.run(m0.copy(), None, None, None)
    assert calls["grad"] < 50
    assert fg(m)[0] < 1e-2 * fg(m0)[0]
    with pytest.raises(ValueError):
        runner.apply_bounds(m, [1.0])
