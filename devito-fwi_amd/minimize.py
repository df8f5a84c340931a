"""The FWI iteration loop: gradient, search direction, line search, model update, logs.

Interface and files written follow the reference's minimize.py:14-180: `minimize(optimizer,
maxIter, ftol, gtol, log_path, save_model_freq, save_grad_freq, save_res_freq).run(m, geometry,
obs_data, misfit_func, direct_wave, mask, precond, bounds)` returns the final squared slowness.
The objective is fwi.fwi_loss unless `loss_func` (same signature) is given, which lets the loop be
exercised on an analytic function.
"""
import os
import numpy as np


def divides(i, j):
    """True if j divides i; j == 0 divides nothing (minimize.py:5-12)."""
    return j != 0 and i % j == 0


class minimize(object):
    def __init__(self, optimizer, maxIter=10, ftol=1e-2, gtol=1e-3, log_path='./log',
                 save_model_freq=5, save_grad_freq=5, save_res_freq=10, loss_func=None):
        assert optimizer.name in ['LBFGS', 'NLCG', 'SteepestDescent', 'steepest descent']
        self.optimizer = optimizer
        self.ftol = ftol
        self.gtol = gtol
        self.maxIter = maxIter
        self.log_path = log_path
        self.save_model_freq = save_model_freq
        self.save_grad_freq = save_grad_freq
        self.save_res_freq = save_res_freq
        if loss_func is None:
            from fwi import fwi_loss as loss_func
        self.loss_func = loss_func
        self.optimizer.setup()
        self.check_path()

    def run(self, m, geometry, obs_data, misfit_func, direct_wave=None, mask=None, precond=True,
            bounds=None):
        """Up to maxIter model updates (minimize.py:37-100). Stops early when the accepted trial
        objective falls below ftol times the first objective, or when a line search fails along
        the steepest-descent direction."""
        opt = self.optimizer
        for iter_count in range(self.maxIter):
            print('Starting iteration', iter_count + 1)
            print('\t Computing gradient')
            fval, g, res = self.loss_func(m, geometry, obs_data, misfit_func, direct_wave, mask,
                                          precond)
            if iter_count == 0:
                self.f0 = fval
            self.save_misfit(fval, g)
            if divides(iter_count, self.save_grad_freq):
                self.save_gradient(g, iter_count)
            if divides(iter_count, self.save_res_freq):
                self.save_residual(res, iter_count)
            print('\t Computing search direction')
            p = opt.compute_direction(m, g)
            print('\t Computing step length')

            searching = True
            while searching:
                alpha = opt.initialize_search(m, g, p, fval)
                while True:
                    print(" trial step", opt.line_search.step_count + 1)
                    m_try = self.apply_bounds(m + alpha * p, bounds)
                    fval_try, _, _ = self.loss_func(m_try, geometry, obs_data, misfit_func,
                                                    direct_wave, mask, precond, calc_grad=False)
                    print('\t fval_try: %10.3e' % fval_try)
                    alpha, status = opt.update_search(alpha, fval_try)
                    if status > 0:
                        opt.finalize_search(g, p)
                        searching = False
                        break
                    if status < 0:
                        if not opt.retry_status(g, p):
                            print(' Line search failed\n\n Aborting...')
                            return m
                        print(' Line search failed\n\n Retrying...')
                        opt.restart()
                        break       # search again along the same p with a cleared history

            m = self.apply_bounds(m + alpha * p, bounds)
            stop = self.finalize(m, g, fval, fval_try, iter_count)
            print('')
            if stop:
                return m
        return m

    def apply_bounds(self, x, bounds):
        """Clip x in place to [bounds[0], bounds[1]] (minimize.py:102-109)."""
        if bounds is not None:
            if len(bounds) != 2:
                raise ValueError('The bounds should only have two values')
            x[x < bounds[0]] = bounds[0]
            x[x > bounds[1]] = bounds[1]
        return x

    def finalize(self, m, g, fk, fkp1, iter_count):
        self.write_count()
        if divides(iter_count, self.save_model_freq):
            self.save_model(m, iter_count)
        return self.check_stopping_criteria(fk, fkp1, g)

    def check_stopping_criteria(self, fk, fkp1, g):
        """1 when the new objective has dropped below ftol * first objective
        (minimize.py:117-133; gtol is accepted but not tested there either)."""
        return 1 if fkp1 / self.f0 < self.ftol else 0

    def _dir(self, *parts):
        path = os.path.join(self.log_path, *parts)
        if not os.path.exists(path):
            os.makedirs(path)
        return path

    def save_model(self, m, k):
        """Velocity 1/sqrt(m) as raw float32 in model_est/v_<k>."""
        (1. / np.sqrt(m)).astype(np.float32).tofile(os.path.join(self._dir('model_est'),
                                                                 'v_' + str(k)))

    def save_gradient(self, g, k):
        g.astype(np.float32).tofile(os.path.join(self._dir('gradient'), 'g_' + str(k)))

    def save_misfit(self, fval, g):
        norm_g = np.max(np.abs(g))
        with open(os.path.join(self.log_path, 'misfit'), 'a') as fh:
            fh.write('%10.3e  %10.3e\n' % (fval, norm_g))
        print('\t\t f: %10.3e \t |g|: %10.3e' % (fval, norm_g))

    def save_residual(self, res, k):
        path = self._dir('residual', str(k))
        for i, r in enumerate(res):
            r.astype(np.float32).tofile(os.path.join(path, 'res' + str(i)))

    def check_path(self):
        self._dir()
        path = os.path.join(self.log_path, 'misfit')
        if os.path.exists(path):
            os.remove(path)

    def write_count(self):
        """Number of wave simulations per shot this iteration, appended to 'sim_count'
        (minimize.py:165-180): forward + adjoint + gradient correlation count 3, each trial 1;
        L-BFGS after its first direction is charged 2."""
        opt = self.optimizer
        count = 3 + opt.line_search.step_count
        if opt.name == 'LBFGS' and opt.call_count != 1:
            count = 2 + opt.line_search.step_count
        opt.writer('sim_count', count)
