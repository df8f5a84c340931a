"""Analytic problems and a protocol loop for checking the optimisers.

`drive` exercises an optimiser object through the same call sequence as the reference's
minimize.py:37-100 and records every trial, so the same function can be run on the reference's
classes (when generating the golden trajectories) and on this package's classes (in the tests).
"""
import numpy as np


def quadratic(n=40, seed=3):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    a = q @ np.diag(np.logspace(0, 2, n)) @ q.T
    x_star = 1.0 + rng.random(n)

    def fg(x):
        d = x - x_star
        return 0.5 * d @ a @ d, a @ d
    return fg, np.full(n, 1.5)


def rosenbrock(n=12):
    def fg(x):
        f = np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
        g = np.zeros_like(x)
        g[:-1] = -400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200.0 * (x[1:] - x[:-1] ** 2)
        return f, g
    return fg, np.full(n, 0.7) + 0.05 * np.arange(n)


PROBLEMS = {'quadratic': quadratic, 'rosenbrock': rosenbrock}

# name -> (problem, optimiser class name, constructor kwargs, iterations, bounds)
CASES = {
    'lbfgs_bracket_quadratic': ('quadratic', 'LBFGS', dict(memory=5, ls_method='Bracket', max_ls=8), 12, None),
    'lbfgs_backtrack_rosenbrock': ('rosenbrock', 'LBFGS', dict(memory=4, ls_method='Backtrack', max_ls=10,
                                                                step_len_max=None), 15, None),
    'lbfgs_bounds_quadratic': ('quadratic', 'LBFGS', dict(memory=3, max_ls=6, step_len_init=0.02), 8, (1.1, 1.9)),
    'nlcg_quadratic': ('quadratic', 'NLCG', dict(thresh=1.0, max_ls=8), 10, None),
    'nlcg_default_rosenbrock': ('rosenbrock', 'NLCG', dict(max_ls=8), 8, None),
    'sd_rosenbrock': ('rosenbrock', 'SteepestDescent', dict(max_ls=8, ls_method='Backtrack'), 8, None),
}


def clip(x, bounds):
    if bounds is not None:
        x[x < bounds[0]] = bounds[0]
        x[x > bounds[1]] = bounds[1]
    return x


def drive(opt, fg, m, iters, bounds=None):
    """Returns (models [iters+1, n], trials: flat list of (iteration, alpha, f_try),
    per-iteration objective)."""
    opt.setup()
    models, trials, fvals = [m.copy()], [], []
    for it in range(iters):
        fval, g = fg(m)
        fvals.append(fval)
        p = opt.compute_direction(m, g)
        searching = True
        while searching:
            alpha = opt.initialize_search(m, g, p, fval)
            while True:
                f_try, _ = fg(clip(m + alpha * p, bounds))
                trials.append((it, float(alpha), float(f_try)))
                alpha, status = opt.update_search(alpha, f_try)
                if status > 0:
                    opt.finalize_search(g, p)
                    searching = False
                    break
                if status < 0:
                    if not opt.retry_status(g, p):
                        return np.array(models), np.array(trials), np.array(fvals)
                    opt.restart()
                    break
        m = clip(m + alpha * p, bounds)
        models.append(m.copy())
    return np.array(models), np.array(trials), np.array(fvals)
