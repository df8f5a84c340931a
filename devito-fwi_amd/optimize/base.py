"""Optimiser front ends: a search-direction engine plus a line search plus the text logs.

Class names, constructor arguments, methods and log files are those of the reference's
optimize/base.py:6-193, LBFGS.py:5-45, NLCG.py:5-42 and steepest_descent.py:4-34, so that
minimize.minimize and marmousi_fwi.py-style drivers work unchanged.
"""
import os
import numpy as np

from . import line_search as _line_search
from . import optimizer as _engines
from .math import angle

_STAT_FILES = ('factor', 'gradient_norm_L1', 'gradient_norm_L2', 'fval', 'restarted', 'slope',
               'step_count', 'sim_count', 'step_length', 'theta')


def dot(x, y):
    return np.dot(np.squeeze(x.flatten()), np.squeeze(y.flatten()))


class Writer(object):
    """Appends one '%e' value per call to the file of the given name (base.py:180-193)."""

    def __init__(self, path='.'):
        self.path = path
        if not os.path.exists(path):
            os.makedirs(path)
        self('step_count', 0)

    def __call__(self, filename, val):
        with open(os.path.join(self.path, filename), 'a') as fh:
            fh.write('%e\n' % val)


class base(object):
    """Common machinery (base.py:6-172).

    m is the current model, g its gradient, p the search direction; initialize_search /
    update_search / finalize_search wrap the line search; retry_status / restart implement the
    recovery after a failed search.
    """

    def __init__(self, line_search_method='Bracket', max_ls=10, step_len_init=None,
                 step_len_max=None, log_path='.', verbose=1):
        assert line_search_method in ['Backtrack', 'Bracket']
        self.line_search_method = line_search_method
        self.max_ls = max_ls
        self.log_path = log_path
        self.step_len_init = step_len_init
        self.step_len_max = step_len_max
        self.verbose = verbose
        self.restarted = 0

    @property
    def name(self):
        raise NotImplementedError("")

    @property
    def call_count(self):
        raise NotImplementedError("")

    def setup(self):
        self.writer = Writer(self.log_path)
        self.line_search = getattr(_line_search, self.line_search_method)(
            step_count_max=self.max_ls, path=self.log_path)
        self.check_path()

    def compute_direction(self, m, g):
        return -g

    def initialize_search(self, m, g, p, fval):
        """First trial step length for direction p (base.py:60-81)."""
        norm_m = np.abs(m).max()
        norm_p = np.abs(p).max()
        if self.restarted:
            self.line_search.clear_history()
        if self.step_len_max:
            self.line_search.step_len_max = self.step_len_max * norm_m / norm_p
        alpha, _ = self.line_search.initialize(0., fval, dot(g, g), dot(g, p))
        # relative first step, only while the search has no history to scale from
        if self.step_len_init and len(self.line_search.step_lens) <= 1:
            alpha = self.step_len_init * norm_m / norm_p
        return alpha

    def update_search(self, alpha, fval):
        return self.line_search.update(alpha, fval)

    def finalize_search(self, g, p):
        """Append this iteration's statistics to the log files (base.py:96-113)."""
        x, f = self.line_search.search_history()[:2]
        slope = (f[1] - f[0]) / (x[1] - x[0])
        self.writer('factor', -dot(g, g) ** -0.5 * slope)
        self.writer('gradient_norm_L1', np.linalg.norm(g, 1))
        self.writer('gradient_norm_L2', np.linalg.norm(g, 2))
        self.writer('fval', f[0])
        self.writer('restarted', self.restarted)
        self.writer('slope', slope)
        self.writer('step_count', self.line_search.step_count)
        self.writer('step_length', x[f.argmin()])
        self.writer('theta', 180. / np.pi * angle(p, -g))
        self.line_search.writer.newline()

    def check_path(self):
        """Start every statistics file afresh."""
        for name in _STAT_FILES:
            path = os.path.join(self.log_path, name)
            if os.path.exists(path):
                os.remove(path)

    def retry_status(self, g, p):
        """1 when the failed direction differed from -g, so that a restart can help."""
        theta = angle(p, -g)
        if self.verbose >= 2:
            print('\t theta: %.3f' % theta)
        return 0 if abs(theta) < 1e-3 else 1

    def restart(self):
        self.line_search.clear_history()
        self.restarted = 1
        self.line_search.writer.iter -= 1
        self.line_search.writer.newline()


class _EngineFrontEnd(base):
    """A front end whose direction comes from one engine object."""
    _defaults = dict(ls_method='Bracket', max_ls=5, step_len_init=0.05, step_len_max=0.5,
                     log_path='.', verbose=1)

    def _init_base(self, ls_method, max_ls, step_len_init, step_len_max, log_path, verbose):
        base.__init__(self, line_search_method=ls_method, max_ls=max_ls,
                      step_len_init=step_len_init, step_len_max=step_len_max,
                      log_path=log_path, verbose=verbose)
        self.engine = None

    @property
    def call_count(self):
        return self.engine.call_count

    def compute_direction(self, m, g):
        p, self.restarted = self.engine.compute_direction(m, g)
        return p

    def restart(self):
        base.restart(self)
        self.engine.restart()


class LBFGS(_EngineFrontEnd):
    """Limited-memory BFGS (LBFGS.py:5-45). `pairs` is passed to optimizer.lbfgs."""

    def __init__(self, memory=5, max_call=np.inf, thresh=0, ls_method='Bracket', max_ls=5,
                 step_len_init=0.05, step_len_max=0.5, log_path='.', verbose=1,
                 pairs='reference'):
        self._init_base(ls_method, max_ls, step_len_init, step_len_max, log_path, verbose)
        self.memory = memory
        self.max_call = max_call
        self.thresh = thresh
        self.pairs = pairs

    @property
    def name(self):
        return 'LBFGS'

    def setup(self):
        base.setup(self)
        self.engine = self.lbfgs = _engines.lbfgs(memory=self.memory, max_call=self.max_call,
                                                  thresh=self.thresh, path=self.log_path,
                                                  pairs=self.pairs)


class NLCG(_EngineFrontEnd):
    """Nonlinear conjugate gradients (NLCG.py:5-42)."""

    def __init__(self, max_call=np.inf, thresh=0, ls_method='Bracket', max_ls=5,
                 step_len_init=0.05, step_len_max=0.5, log_path='.', verbose=1):
        self._init_base(ls_method, max_ls, step_len_init, step_len_max, log_path, verbose)
        self.max_call = max_call
        self.thresh = thresh

    @property
    def name(self):
        return 'NLCG'

    def setup(self):
        base.setup(self)
        self.engine = self.nlcg = _engines.nlcg(max_call=self.max_call, thresh=self.thresh)


class SteepestDescent(_EngineFrontEnd):
    """Steepest descent (steepest_descent.py:4-34); never needs a restart."""

    def __init__(self, ls_method='Bracket', max_ls=5, step_len_init=0.05, step_len_max=0.5,
                 log_path='.', verbose=1):
        self._init_base(ls_method, max_ls, step_len_init, step_len_max, log_path, verbose)

    @property
    def name(self):
        return 'steepest descent'

    def setup(self):
        base.setup(self)
        self.engine = self.sd = _engines.steepest_descent()

    def restart(self):
        pass
