"""The 2-D quadratic Wasserstein misfit (reference misfit/misfit.py:71-104, misfit/bfm.py, misfit/QW2D/src/fot2d.c).
The reference's solver needs FFTW and cannot be built here, and its repository holds none of its outputs, so
the restatement is checked against facts of optimal transport (parity unpinned): the discrete Legendre
transform of a quadratic, W2^2 / 2 = |shift|^2 / 2 between translates, a linear Kantorovich potential for a
translation, and the derivative against finite differences of the solver's own value."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _grid(n2, n1):
    x, y = (np.arange(n1) + .5) / n1, (np.arange(n2) + .5) / n2
    return np.meshgrid(x, y)


def _blob(X, Y, cx, cy, s=0.07):
    return np.exp(-((X - cx) ** 2 + (Y - cy) ** 2) / (2 * s * s))


def test_ot_library_exports_header_symbols():
    from misfit import bfm
    lib = bfm._lib()
    header = open(os.path.join(ROOT, "include", "mifwi_ot.h")).read()
    declared = set(re.findall(r"\b(mifwi_ot_[a-z0-9_]+)\s*\(", header))
    assert declared == {"mifwi_ot_legendre2d", "mifwi_ot_pushforward2d"}
    for name in declared:
        assert hasattr(lib, name), name
    null = ctypes.POINTER(ctypes.c_double)()
    assert lib.mifwi_ot_legendre2d(0, 4, null, null) == -1
    assert lib.mifwi_ot_pushforward2d(4, 4, null, null, null, 1.0) == -1


def test_legendre_transform_of_a_quadratic():
    from misfit.bfm import legendre2d
    X, Y = _grid(120, 96)
    a, c = 2.0, (0.4, 0.55)
    u = a / 2 * ((X - c[0]) ** 2 + (Y - c[1]) ** 2)
    d = legendre2d(u)
    exact = X * c[0] + Y * c[1] + (X ** 2 + Y ** 2) / (2 * a)      # where the maximiser c + s / a is inside
    inside = (c[0] + X / a < 1) & (c[1] + Y / a < 1)
    assert np.abs(d - exact)[inside].max() <= 5e-5                 # O(h^2) of the 1/96 grid
    # involution wherever the slope of u lies in the slope grid (0, 1)
    back = legendre2d(d)
    sl = (a * (X - c[0]) > 0.02) & (a * (X - c[0]) < 0.98) & (a * (Y - c[1]) > 0.02) & (a * (Y - c[1]) < 0.98)
    assert np.abs(back - u)[sl].max() <= 1e-4
    assert np.all(back <= u + 1e-12)                               # the double transform is the convex envelope


def test_pushforward_by_the_identity_keeps_an_interior_density():
    from misfit.bfm import pushforward2d
    X, Y = _grid(96, 128)
    mu = _blob(X, Y, 0.5, 0.45, 0.1)
    rho = pushforward2d(.5 * (X ** 2 + Y ** 2), mu, mu.mean())
    assert abs(rho.mean() - mu.mean()) <= 1e-12 * mu.mean()
    assert np.abs(rho - mu).max() <= 5e-3 * mu.max()               # the sampling smooths by [1/8, 3/4, 1/8]


def test_distance_and_potential_of_a_translation():
    from misfit.bfm import bfm
    X, Y = _grid(128, 128)
    shift = (0.1, -0.06)
    f, g = _blob(X, Y, 0.45, 0.5), _blob(X, Y, 0.45 + shift[0], 0.5 + shift[1])
    solver = bfm(num_steps=40, step_scale=8.)
    value, grad = solver.gradient(f, g)
    exact = .5 * (shift[0] ** 2 + shift[1] ** 2)
    assert abs(value - exact) <= 5e-3 * exact
    assert solver.residuals[-1] < 1e-2 * solver.residuals[0]
    w = (f / f.sum()).ravel()
    A = np.stack([X.ravel(), Y.ravel(), np.ones(X.size)], 1) * np.sqrt(w)[:, None]
    slope = np.linalg.lstsq(A, grad.ravel() * np.sqrt(w), rcond=None)[0][:2] * f.mean()
    assert np.allclose(slope, [-shift[0], -shift[1]], rtol=2e-2)
    # identical densities: zero distance
    assert abs(bfm(num_steps=5).gradient(f, f)[0]) <= 1e-6


def test_derivative_against_finite_differences():
    from misfit.bfm import bfm
    X, Y = _grid(64, 64)
    f = _blob(X, Y, 0.42, 0.5, 0.09) + 0.05
    g = _blob(X, Y, 0.55, 0.46, 0.08) + 0.05
    h = _blob(X, Y, 0.38, 0.55, 0.06) * np.cos(9 * X)               # a smooth perturbation of f
    solver = bfm(num_steps=60, step_scale=8.)
    value, grad = solver.gradient(f, g)
    eps = 1e-3
    fd = (solver.gradient(f + eps * h, g)[0] - solver.gradient(f - eps * h, g)[0]) / (2 * eps)
    # d value / d f_ij = grad_ij / (number of cells): the solver works with means, not sums
    an = float(np.sum(grad * h)) / f.size
    assert abs(fd - an) <= 0.05 * abs(fd), (fd, an)


def test_qwasserstein_2d_wrapper():
    from misfit import qWasserstein
    from misfit.bfm import bfm
    X, Y = _grid(60, 48)
    f = _blob(X, Y, 0.4, 0.5, 0.1) * np.cos(14 * Y)                  # signed "gathers"
    g = _blob(X, Y, 0.5, 0.45, 0.1) * np.cos(14 * Y)
    w2 = qWasserstein(gamma=1.01, method='2d', num_steps=10, step_scale=1.)
    loss, adj = w2(f, g)
    assert loss > 0 and adj.shape == f.shape and np.isfinite(adj).all()
    shift = 1.01 * -min(f.min(), g.min())
    ref_loss, ref_grad = bfm(num_steps=10, step_scale=1.).gradient(f + shift, g + shift)
    assert loss == ref_loss
    np.testing.assert_allclose(adj, ref_grad / ((f + shift).sum() / f.size), rtol=1e-12)
    with pytest.raises(ValueError):
        w2(f[:, 0], g[:, 0])
