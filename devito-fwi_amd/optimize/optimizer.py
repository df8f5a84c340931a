"""Search-direction engines: steepest descent, nonlinear conjugate gradients, L-BFGS.

Each engine's compute_direction(m, g) returns (direction, restarted). Behaviour follows the
reference's optimize/optimizer/base.py:3-12, NLCG.py:7-98 and LBFGS.py:6-134.
"""
import numpy as np

from .math import angle, dot


class steepest_descent(object):
    def __init__(self):
        self.call_count = 0

    def compute_direction(self, m, g):
        self.call_count += 1
        return -g, 0


Base = steepest_descent


def _ratio(num, den):
    return num / den if den != 0 else 0


_BETA = {
    'FR': lambda gn, go, po: _ratio(dot(gn, gn), dot(go, go)),
    'PR': lambda gn, go, po: max(_ratio(dot(gn, gn - go), dot(go, go)), 0),
    'HS': lambda gn, go, po: _ratio(-dot(gn, gn - go), dot(po, gn - go)),
    'DY': lambda gn, go, po: _ratio(-dot(gn, gn), dot(po, gn - go)),
}


class nlcg(steepest_descent):
    """Nonlinear conjugate gradients (optimizer/NLCG.py:7-62). Falls back to -g (and reports a
    restart) after max_call directions, when |g_new.g_old|/|g_new|^2 exceeds thresh, or when the
    new direction is not a descent direction. With the default thresh=0 the conjugacy test
    restarts on practically every call, as in the reference."""

    def __init__(self, beta_type='FR', max_call=np.inf, thresh=0.):
        assert beta_type in _BETA
        self.beta_type = beta_type
        self.g_old = self.g_new = self.p_old = self.p_new = None
        self.thresh = thresh
        self.max_call = max_call
        self.call_count = 0

    def compute_direction(self, m, g):
        self.g_old, self.p_old = self.g_new, self.p_new
        self.g_new = g
        self.call_count += 1
        if self.call_count == 1:
            self.p_new = -g
            return -g, 0
        if self.call_count > self.max_call:
            print('restarting NLCG... [periodic restart]')
            self.restart()
            return -g, 1

        beta = _BETA[self.beta_type](self.g_new, self.g_old, self.p_old)
        self.p_new = -self.g_new + beta * self.p_old

        gg = dot(self.g_new, self.g_new)
        if abs(dot(self.g_new, self.g_old) / gg) > self.thresh:
            print('restarting NLCG... [loss of conjugacy]')
            self.restart()
            return -g, 1
        if dot(self.p_new, self.g_new) / gg > 0.:
            print('restarting NLCG... [not a descent direction]')
            self.restart()
            return -g, 1
        return self.p_new, 0

    def restart(self):
        self.call_count = 0


class lbfgs(steepest_descent):
    """Limited-memory BFGS with the two-loop recursion and the Liu-Nocedal scaling
    (optimizer/LBFGS.py:6-134).

    The pairs are kept in float32 like the reference's memory-mapped S and Y files, newest in
    column 0, but in memory rather than on disk. `pairs` selects what goes into Y:
      'reference' - the previous gradient itself, which is what the reference stores
                    (optimizer/LBFGS.py:56, `y = g = self.g`); default, so that iterates match it;
      'secant'    - the textbook difference g - g_previous.
    """

    def __init__(self, memory=10, thresh=0., max_call=np.inf, path='.', pairs='reference'):
        assert pairs in ('reference', 'secant')
        self.memory = memory
        self.max_call = max_call
        self.thresh = thresh
        self.pairs = pairs
        self.call_count = 0
        self.memory_used = 0
        self.g = None
        self.m = None
        self.S = None
        self.Y = None

    def compute_direction(self, m, g):
        self.call_count += 1
        if self.call_count == 1:
            self.g, self.m = g, m
            return -g, 0
        if self.call_count > self.max_call:
            print('restarting LBFGS... [periodic restart]')
            self.restart()
            return -g, 1

        S, Y = self.update(m, g)
        q = self.apply(g, S, Y)
        self.g, self.m = g, m
        if self.check_status(g, q) != 0:
            self.restart()
            return -g, 1
        return -q, 0

    def update(self, m, g):
        """Push the newest (s, y) pair into column 0."""
        s = m - self.m
        y = self.g if self.pairs == 'reference' else g - self.g
        if self.S is None or self.S.shape != (len(s), self.memory):
            self.S = np.zeros((len(s), self.memory), dtype=np.float32)
            self.Y = np.zeros((len(s), self.memory), dtype=np.float32)
        if self.memory_used > 0:
            self.S[:, 1:] = self.S[:, :-1].copy()
            self.Y[:, 1:] = self.Y[:, :-1].copy()
        self.S[:, 0] = s
        self.Y[:, 0] = y
        self.memory_used = min(self.memory_used + 1, self.memory)
        return self.S, self.Y

    def apply(self, q, S=None, Y=None):
        """Inverse-Hessian estimate times q."""
        S = self.S if S is None else S
        Y = self.Y if Y is None else Y
        k = self.memory_used
        rho = np.zeros(k)
        alpha = np.zeros(k)
        for i in range(k):
            rho[i] = 1 / np.dot(Y[:, i], S[:, i])
            alpha[i] = rho[i] * np.dot(S[:, i], q)
            q = q - alpha[i] * Y[:, i]
        r = q * (np.dot(Y[:, 0], S[:, 0]) / np.dot(Y[:, 0], Y[:, 0]))
        for i in range(k - 1, -1, -1):
            beta = rho[i] * np.dot(Y[:, i], r)
            r = r + S[:, i] * (alpha[i] - beta)
        return r

    def restart(self):
        self.call_count = 0
        self.memory_used = 0
        if self.S is not None:
            self.S[:] = 0.
            self.Y[:] = 0.

    def check_status(self, g, r):
        theta = 180. / np.pi * angle(g, r)
        if not 0. < theta < 90.:
            print('restarting LBFGS... [not a descent direction]')
            return 1
        if theta > 90. - self.thresh:
            print('restarting LBFGS... [practical safeguard]')
            return 1
        return 0
