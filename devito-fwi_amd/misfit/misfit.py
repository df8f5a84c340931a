import numpy as np


def least_square(x, y):
    """0.5 * ||x - y||^2 and the residual x - y (reference misfit/misfit.py:5-9)."""
    residual = x - y
    fval = .5 * np.linalg.norm(residual.flatten()) ** 2
    return fval, residual


class qWasserstein(object):
    """Quadratic Wasserstein misfit between synthetic f and observed g, called as
    `loss, adjoint_source = qWasserstein(...)(f, g)` (reference misfit/misfit.py:11-104).

    method='1d' compares trace by trace (each column of an (nt, ntr) gather) through the
    closed-form 1-D optimal map; the signals are first made positive by `trans_type`
    ('linear' shifts by gamma * |min|, 'square', 'exp', 'softplus'). method='2d' treats the whole gather
    as one density on the unit square and solves the transport problem with the back-and-forth method
    (misfit/bfm.py here; misfit/misfit.py:71-81 and misfit/bfm.py of the reference).

    No output of the reference exists for these misfits in its repository, so the numbers are
    checked against finite differences and closed forms only (parity unpinned).
    """

    def __init__(self, trans_type='linear', gamma=1.0, method='1d', num_steps=10, step_scale=1.):
        assert method in ['1d', '2d']
        self.gamma = gamma
        self.method = method
        self.trans_type = trans_type
        self.num_steps, self.step_scale = num_steps, step_scale
        self.bfm = None            # built on first use: its helper library is only needed for method='2d'

    def _transform(self, f, g):
        """Positive densities mu(f), nu(g) and d mu / d f (misfit/misfit.py:21-47)."""
        shift = 0
        if self.trans_type == 'square':
            mu, nu, d = f * f, g * g, 2 * f
        elif self.trans_type == 'exp':
            mu, nu = np.exp(self.gamma * f), np.exp(self.gamma * g)
            d = self.gamma * mu
        elif self.trans_type == 'softplus':
            mu, nu = np.log(np.exp(self.gamma * f) + 1), np.log(np.exp(self.gamma * g) + 1)
            d = self.gamma / (1 + np.exp(-self.gamma * f))
        else:
            mu, nu, d = f, g, np.ones(f.shape)
            if self.trans_type == 'linear':
                lowest = min(f.min(), g.min())
                shift = -lowest * self.gamma if lowest < 0 else 0
        return mu + shift, nu + shift, d

    @staticmethod
    def _trace(f, g):
        """W2^2/2 between two positive 1-D signals on [0, 1] and its derivative with respect to f
        (misfit/misfit.py:49-69)."""
        mass = f.sum()
        mu, nu = f / mass, g / g.sum()
        t = np.linspace(0, 1, mu.size)
        shift = t - np.interp(np.cumsum(mu), np.cumsum(nu), t)     # t - T(t), T the optimal map
        loss = .5 * (shift ** 2 * mu).sum()
        grad = np.cumsum(shift) - shift.sum()
        return loss, (grad - (grad * mu).sum()) / mass

    def batch_device(self, f, g):
        """The same misfit for a whole batch of gathers held as torch tensors (ns, nt, ntr) on one
        device: returns (losses (ns,) float64, adjoint sources like f).  Every trace of every shot is
        handled at once in float64 tensor operations, with np.interp's rule for the optimal map
        (bracketing by binary search, end values outside the range), so the numbers agree with the
        trace-by-trace host loop to rounding while the gathers never leave the device."""
        import torch
        if self.method != '1d':
            raise NotImplementedError("only the trace-wise misfit has a batched form")
        ns, nt, ntr = f.shape
        f64, g64 = f.to(torch.float64), g.to(torch.float64)
        gamma = self.gamma
        if self.trans_type == 'square':
            mu, nu, d = f64 * f64, g64 * g64, 2 * f64
        elif self.trans_type == 'exp':
            mu, nu = torch.exp(gamma * f64), torch.exp(gamma * g64)
            d = gamma * mu
        elif self.trans_type == 'softplus':
            mu, nu = torch.log(torch.exp(gamma * f64) + 1), torch.log(torch.exp(gamma * g64) + 1)
            d = gamma / (1 + torch.exp(-gamma * f64))
        else:
            mu, nu, d = f64, g64, None
            if self.trans_type == 'linear':      # one shift per gather, from the minimum over both signals
                lowest = torch.minimum(f64.amin(dim=(1, 2)), g64.amin(dim=(1, 2)))
                shift = torch.where(lowest < 0, -lowest * gamma, torch.zeros_like(lowest)).reshape(ns, 1, 1)
                mu, nu = mu + shift, nu + shift
        # traces as rows: (ns * ntr, nt)
        mu = mu.permute(0, 2, 1).reshape(ns * ntr, nt)
        nu = nu.permute(0, 2, 1).reshape(ns * ntr, nt)
        mass = mu.sum(dim=1, keepdim=True)
        mu = mu / mass
        nu = nu / nu.sum(dim=1, keepdim=True)
        t = torch.linspace(0, 1, nt, dtype=torch.float64, device=f.device)
        x, xp = torch.cumsum(mu, dim=1).contiguous(), torch.cumsum(nu, dim=1).contiguous()
        j = (torch.searchsorted(xp, x, right=True) - 1).clamp_(0, nt - 2)
        x0, x1 = torch.gather(xp, 1, j), torch.gather(xp, 1, j + 1)
        t0, t1 = t[j], t[j + 1]
        dx = x1 - x0
        tmap = torch.where(dx > 0, (t1 - t0) / torch.where(dx > 0, dx, torch.ones_like(dx)) * (x - x0) + t0, t0)
        tmap = torch.where(x < xp[:, :1], t[0].expand_as(tmap), tmap)
        tmap = torch.where(x >= xp[:, -1:], t[-1].expand_as(tmap), tmap)
        shift = t.unsqueeze(0) - tmap
        loss = .5 * (shift * shift * mu).sum(dim=1)
        grad = torch.cumsum(shift, dim=1) - shift.sum(dim=1, keepdim=True)
        grad = (grad - (grad * mu).sum(dim=1, keepdim=True)) / mass
        grad = grad.reshape(ns, ntr, nt).permute(0, 2, 1)
        if d is not None:
            grad = grad * d
        return loss.reshape(ns, ntr).sum(dim=1), grad.to(f.dtype).contiguous()

    def __call__(self, f, g):
        shape = f.shape
        if self.method == '2d' and (f.ndim == 1 or shape[1] <= 1):
            raise ValueError("Can not use 2d method for 1D input.")
        mu, nu, d = self._transform(f, g)
        if self.method == '2d':
            # misfit/misfit.py:71-81: the solver normalises both densities itself; its derivative is divided
            # by the mean of mu once more, as the reference does
            if self.bfm is None:
                from .bfm import bfm
                self.bfm = bfm(num_steps=self.num_steps, step_scale=self.step_scale)
            loss, grad = self.bfm.gradient(mu, nu)
            return loss, grad / (mu.sum() / mu.size) * d
        if f.ndim == 1 or shape[1] <= 1:
            loss, grad = self._trace(np.squeeze(mu), np.squeeze(nu))
            return loss, grad.reshape(shape) * d
        loss, grad = 0, np.zeros(shape)
        for i in range(shape[1]):
            value, grad[:, i] = self._trace(mu[:, i], nu[:, i])
            loss += value
        return loss, grad * d


class Misfit(object):
    def __init__(self, operator):
        self.operator = operator

    def __call__(self, x, y):
        return self.operator(x, y)
