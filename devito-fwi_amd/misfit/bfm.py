"""Quadratic optimal transport between two positive 2-D densities by the back-and-forth method, as the
reference's 2-D Wasserstein misfit uses it (misfit/bfm.py:16-131 for the iteration, misfit/QW2D/src/fot2d.c
for its grid kernels, which the reference runs as the separate executable QW2D/bin/bfm2d).

The iteration is host Python: Poisson solves by DCT (scipy.fft), the step-size rule and the dual objective;
the two grid kernels (Legendre transform, push-forward of a density) come from libmifwi_ot.so
(include/mifwi_ot.h).  Everything is float64.

The reference's solver cannot be built here (it needs FFTW) and its repository holds no output of it, so
this restatement is checked against facts of optimal transport only: the transform is an involution on
convex functions, the distance between translates of one density is half the squared shift, the potential
of a translation is linear (parity unpinned).
"""
import ctypes
import os

import numpy as np
from scipy.fft import dctn, idctn

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    """libmifwi_ot.so, built in-tree by `make -C devito-fwi_amd/csrc`.  Fails loudly if absent."""
    global _LIB
    if _LIB is None:
        path = os.environ.get("MIFWI_OT_LIB") or os.path.join(os.path.dirname(_HERE), "libmifwi_ot.so")
        if not os.path.exists(path):
            raise ImportError("libmifwi_ot.so not found at %s: build it with `make -C devito-fwi_amd/csrc`" % path)
        lib = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        lib.mifwi_ot_legendre2d.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp]
        lib.mifwi_ot_pushforward2d.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, dp, ctypes.c_double]
        _LIB = lib
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def legendre2d(u):
    """Discrete Legendre transform of u (n2, n1) on the unit square, at the cell centres."""
    u = np.ascontiguousarray(u, dtype=np.float64)
    out = np.empty_like(u)
    rc = _lib().mifwi_ot_legendre2d(u.shape[1], u.shape[0], _ptr(u), _ptr(out))
    assert rc == 0, rc
    return out


def pushforward2d(potential, mu, total_mass=1.0):
    """(grad potential)_# mu on the cell grid, rescaled to mean `total_mass`."""
    potential = np.ascontiguousarray(potential, dtype=np.float64)
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    out = np.empty_like(mu)
    rc = _lib().mifwi_ot_pushforward2d(mu.shape[1], mu.shape[0], _ptr(potential), _ptr(mu), _ptr(out), float(total_mass))
    assert rc == 0, rc
    return out


class bfm(object):
    """`value, grad = bfm(num_steps, step_scale).gradient(f, g)`: half the squared 2-Wasserstein distance
    between f and g (both scaled to unit mean on the unit square) and its derivative with respect to f,
    with the normalisation of the reference's solver (fot2d.c:606-656)."""

    def __init__(self, num_steps=10, step_scale=8., verbose=False):
        self.num_steps = int(num_steps)
        self.step_scale = float(step_scale)
        self.upper, self.lower, self.scale_down = .75, .25, .8
        self.verbose = verbose
        self.values, self.residuals = [], []

    def _setup(self, shape):
        n2, n1 = shape
        jj, ii = np.arange(n1), np.arange(n2)
        # eigenvalues of the negative Neumann Laplacian under the DCT (fot2d.c:4-17)
        self._kernel = (2 * n1 * n1 * (1 - np.cos(np.pi * jj / n1)))[None, :] + \
                       (2 * n2 * n2 * (1 - np.cos(np.pi * ii / n2)))[:, None]
        self._kernel[0, 0] = 1.0
        x, y = (jj + .5) / n1, (ii + .5) / n2
        self._half_sq = .5 * (x[None, :] ** 2 + y[:, None] ** 2)

    def _ascend(self, potential, rho, target, sigma):
        """potential += sigma * (-Laplace)^-1 (rho - target); returns the H^-1 norm squared of the residual
        (fot2d.c:459-482)."""
        r = rho - target
        w = dctn(r, norm='ortho') / self._kernel
        w[0, 0] = 0.0
        w = idctn(w, norm='ortho')
        potential += sigma * w
        return float(np.sum(w * r) / r.size)

    def _objective(self, phi, psi, mu, nu):
        return float(np.sum(self._half_sq * (mu + nu) - nu * phi - mu * psi) / phi.size)

    def _step(self, sigma, value, old, grad_sq):
        gain = value - old
        if gain > grad_sq * sigma * self.upper:
            return sigma / self.scale_down
        if gain < grad_sq * sigma * self.lower:
            return sigma * self.scale_down
        return sigma

    def solve(self, f, g):
        """Returns (value, phi, psi, mean of f); phi and psi are the Kantorovich potentials
        (half |x|^2 minus the convex potentials of the iteration, fot2d.c:591-598)."""
        f = np.ascontiguousarray(f, dtype=np.float64)
        g = np.ascontiguousarray(g, dtype=np.float64)
        if f.ndim != 2 or f.shape != g.shape:
            raise ValueError("two 2-D arrays of one shape are needed")
        mean_f, mean_g = f.sum() / f.size, g.sum() / g.size
        if mean_f <= 0 or mean_g <= 0:
            raise ValueError("Non-positive density function unsupported!")
        self._setup(f.shape)
        mu, nu = f / mean_f, g / mean_g
        sigma = self.step_scale / max(mu.max(), nu.max())
        phi, psi = self._half_sq.copy(), self._half_sq.copy()
        rho = mu.copy()
        old = self._objective(phi, psi, mu, nu)
        self.values, self.residuals = [], []
        value = old
        for k in range(self.num_steps):
            grad_sq = self._ascend(phi, rho, nu, sigma)
            psi = legendre2d(phi)                       # convexify: psi = phi*, phi = psi*
            phi = legendre2d(psi)
            value = self._objective(phi, psi, mu, nu)
            sigma = self._step(sigma, value, old, grad_sq)
            old = value
            rho = pushforward2d(phi, nu)
            grad_sq = self._ascend(psi, rho, mu, sigma)
            phi = legendre2d(psi)
            psi = legendre2d(phi)
            rho = pushforward2d(psi, mu)
            value = self._objective(phi, psi, mu, nu)
            sigma = self._step(sigma, value, old, grad_sq)
            old = value
            self.values.append(value)
            self.residuals.append(grad_sq)
            if self.verbose:
                print("iter %d, step %e, W2 value %e, H^-1 residual %e" % (k, sigma, value, grad_sq))
        return value, self._half_sq - phi, self._half_sq - psi, mean_f

    def gradient(self, f, g):
        value, _, psi, mean_f = self.solve(f, g)
        mu = np.asarray(f, dtype=np.float64) / mean_f
        return value, (psi - np.sum(mu * psi) / psi.size) / mean_f


# the reference's drivers import the solver under this name (misfit/misfit.py:1)
bfmx = bfm
