"""Generates tests/golden/optimize_traj.npz: iterates of the reference's own optimisers
(/root/reference/optimize, plain numpy) on analytic problems, driven by tests/optimize_driver.py.

Run from the repository root in the build container:  python tests/golden/make_optimize_golden.py
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, '/root/reference')

import optimize as ref_optimize          # noqa: E402  (the reference's package)
from optimize_driver import CASES, PROBLEMS, drive   # noqa: E402

assert ref_optimize.__file__.startswith('/root/reference')

# optimize/optimizer/LBFGS.py:82 tests `S==[]` on the memory-mapped history. Older numpy answered
# False to that comparison; numpy >= 2 raises. Give the arrays handed to `apply` the old answer
# and leave the algorithm itself untouched.
from optimize.optimizer.LBFGS import lbfgs as _ref_lbfgs   # noqa: E402


class _OldEq(np.memmap):
    def __eq__(self, other):
        if isinstance(other, list):
            return False
        return np.ndarray.__eq__(self, other)


_ref_update = _ref_lbfgs.update


def _update(self, m, g):
    S, Y = _ref_update(self, m, g)
    return S.view(_OldEq), Y.view(_OldEq)


_ref_lbfgs.update = _update
out = {}
for name, (problem, cls, kwargs, iters, bounds) in CASES.items():
    fg, m0 = PROBLEMS[problem]()
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        opt = getattr(ref_optimize, cls)(log_path=tmp, **kwargs)
        models, trials, fvals = drive(opt, fg, m0.copy(), iters, bounds)
        with open(os.path.join(tmp, 'optim_info')) as fh:
            info = fh.read()
        stats = {s: np.loadtxt(os.path.join(tmp, s), ndmin=1)
                 for s in ('fval', 'step_length', 'step_count', 'theta', 'restarted', 'slope')}
    out[name + '/models'] = models
    out[name + '/trials'] = trials
    out[name + '/fvals'] = fvals
    out[name + '/optim_info'] = np.array(info)
    for s, v in stats.items():
        out[name + '/stat_' + s] = v
    print(name, 'iterations', len(models) - 1, 'trials', len(trials), 'f', fvals[0], '->', fvals[-1])
np.savez_compressed(os.path.join(HERE, 'optimize_traj.npz'), **out)
