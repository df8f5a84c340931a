"""Line searches driven one trial at a time: the caller evaluates the objective, the search
answers with the next step length and a status (>0 done, 0 continue, <0 failed).

Behaviour follows the reference's optimize/line_search/base.py:9-147, bracket.py:8-107 and
backtrack.py:7-68, including the 'optim_info' text log.
"""
import os
import numpy as np

from .math import backtrack2, polyfit2

GOLDEN = 1.618034


class InfoLog(object):
    """The three-column ITER / STEPLEN / MISFIT table 'optim_info' (line_search/base.py:102-147).
    A row with step length 0 opens a new iteration; trial steps are indented under it."""

    def __init__(self, path='.'):
        self.iter = 0
        self.filename = os.path.join(path, 'optim_info')
        if os.path.exists(self.filename):
            os.remove(self.filename)
        self.write_header()

    def _append(self, text):
        with open(self.filename, 'a') as fh:
            fh.write(text)

    def write_header(self):
        cols = ('ITER', 'STEPLEN', 'MISFIT')
        self._append(''.join('%10s  ' % c for c in cols) + '\n' +
                     ''.join('%10s  ' % (10 * '=') for _ in cols) + '\n')

    def __call__(self, steplen=None, funcval=None):
        if self.iter == 0 or steplen == 0.:
            self.iter += 1
            self._append('%10d  %10.3e  %10.3e\n' % (self.iter, steplen, funcval))
        else:
            self._append(12 * ' ' + '%10.3e  %10.3e\n' % (steplen, funcval))

    def newline(self):
        self._append('\n')


class Base(object):
    """History keeping shared by the searches (line_search/base.py:9-99). step_lens / func_vals
    run over ALL trials since the last clear_history(); a step length of 0 marks the start of a
    model update, so the number of zeros minus one is the number of completed updates."""

    def __init__(self, step_count_max=10, step_len_max=np.inf, path='.'):
        self.step_count_max = step_count_max
        self.step_len_max = step_len_max
        self.writer = InfoLog(path)
        self.step_count = 0
        self.clear_history()

    @property
    def name(self):
        return 'constant'

    def clear_history(self):
        self.func_vals = []
        self.step_lens = []
        self.gtg = []
        self.gtp = []

    def search_history(self, sort=True):
        """(x, f, gtg, gtp, step_count, update_count) of the search in progress; x and f hold
        the starting point and the trials so far, ordered by |x| when sort is set."""
        n = self.step_count
        updates = int(np.sum(np.array(self.step_lens) == 0)) - 1
        x = np.array(self.step_lens[len(self.step_lens) - n - 1:])
        f = np.array(self.func_vals[len(self.func_vals) - n - 1:])
        if sort:
            order = np.abs(x).argsort()
            x, f = x[order], f[order]
        return x, f, self.gtg, self.gtp, n, updates

    def initialize(self, step_len, func_val, gtg, gtp):
        self.step_count = 0
        self.step_lens.append(step_len)
        self.func_vals.append(func_val)
        self.gtg.append(gtg)
        self.gtp.append(gtp)
        self.writer(step_len, func_val)
        return self.calculate_step()

    def update(self, step_len, func_val):
        self.step_count += 1
        self.step_lens.append(step_len)
        self.func_vals.append(func_val)
        self.writer(step_len, func_val)
        return self.calculate_step()

    def calculate_step(self):
        return self.step_lens[0]


def _bracketed(x, f):
    """A sample below f[0] with a higher sample beyond it (bracket.py:81-89)."""
    lo = int(f.argmin())
    return bool(f[lo] < f[0] and np.any(f[lo:] > f[lo]))


def _near_parabola_vertex(x, f, thresh=np.log10(1.2)):
    """Some trial lies within a factor 1.2 of the parabolic estimate (bracket.py:92-103)."""
    if not _bracketed(x, f):
        return False
    return bool(np.any(np.abs(np.log10(x[1:] / polyfit2(x, f))) < thresh))


class Bracket(Base):
    """Bracketing search with parabolic refinement (bracket.py:8-78)."""

    @property
    def name(self):
        return 'bracket'

    def calculate_step(self):
        x, f, gtg, gtp, steps, updates = self.search_history()
        status = 0
        if steps == 0 and updates == 0:
            alpha = gtg[-1] ** -1                      # first step of all: 1/|g|^2
        elif steps == 0:
            # scale the best step of the previous search by the ratio of directional derivatives
            best = int(np.argmin(self.func_vals[:-1]))
            alpha = self.step_lens[best] * gtp[-2] / gtp[-1]
        elif _bracketed(x, f) and _near_parabola_vertex(x, f):
            alpha, status = x[f.argmin()], 1
        elif _bracketed(x, f):
            alpha = polyfit2(x, f)
        elif steps <= self.step_count_max and np.all(f <= f[0]):
            alpha = GOLDEN * x[-1]                     # still going down: reach further
        elif steps <= self.step_count_max:
            alpha = backtrack2(f[0], gtp[-1] / gtg[-1], x[1], f[1], b1=0.1, b2=0.5)
        else:
            alpha, status = 0, -1

        # step length safeguard
        if alpha > self.step_len_max and steps == 0:
            alpha, status = (GOLDEN - 1.) * self.step_len_max, 0
        elif alpha > self.step_len_max:
            alpha, status = self.step_len_max, 1
        return alpha, status


class Backtrack(Bracket):
    """Backtracking search for well-scaled (quasi-Newton) directions; before the first model
    update it behaves like Bracket (backtrack.py:7-58)."""

    @property
    def name(self):
        return 'backtrack'

    def calculate_step(self):
        x, f, gtg, gtp, steps, updates = self.search_history()
        if updates == 0:
            return super(Backtrack, self).calculate_step()
        if steps == 0:
            return min(1., self.step_len_max), 0
        if f.min() < f[0]:
            return x[f.argmin()], 1
        if steps <= self.step_count_max:
            return backtrack2(f[0], gtp[-1] / gtg[-1], x[1], f[1], b1=0.1, b2=0.5), 0
        return None, -1
