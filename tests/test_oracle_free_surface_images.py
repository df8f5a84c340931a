"""
The free-surface restatement against the method of images, on the CPU.

The reference's freesurface() (seismic/acoustic/operators.py:8-35) replaces every function read through a negative
z offset by sign(z') * f[|z'|]: the update then equals that of the field's antisymmetric extension about z = 0.
So a free-surface run must give, in float64 to rounding, the lower half of a run WITHOUT a free surface on the
model mirrored about z = 0 (absorbing layers at both ends) with an image source of opposite sign above the surface,
as long as nothing is injected into row z = 0 itself (where the literal form's sign(0) = 0 differs from the
extension).  This checks the OT2 form, pinned besides by the reference's known-answer test 369.955, and the OT4
form, where the rewrite also reaches into the factors of the biharmonic term and no number of the reference exists.
"""
import numpy as np
import pytest

from helpers import rel_l2, smooth_random_vp
from oracle import oracle as orc


def _pair(kernel, so, shape=(37, 29), nbl=8, h=10.0, nt=140, seed=3):
    nx, nz = shape
    vp = smooth_random_vp(shape, seed=seed).astype(np.float64)
    fs = orc.Problem(vp, (h, h), nbl, so, dtype=np.float64, fs=True, kernel=kernel)
    vp_m = np.concatenate([vp[:, :0:-1], vp], axis=1)                  # rows -(nz-1) .. nz-1
    full = orc.Problem(vp_m, (h, h), nbl, so, origin=(0.0, -(nz - 1) * h), dtype=np.float64, fs=False, kernel=kernel)
    dt = float(fs.solver_dt)
    assert np.isclose(dt, float(full.solver_dt))
    tv = dt * np.arange(nt)
    w = orc.ricker(0.02, tv).reshape(nt, 1)
    src = np.array([[0.43 * (nx - 1) * h, 1.6 * h]])                   # between rows 1 and 2: row 0 is not injected
    rec = np.stack([np.linspace(0.1, 0.9, 9) * (nx - 1) * h, np.linspace(1.2, 4.7, 9) * h], axis=1)
    return fs, full, dt, w, src, rec, nz, nbl


@pytest.mark.parametrize("kernel,so", [("OT2", 4), ("OT2", 8), ("OT4", 4), ("OT4", 8)])
def test_free_surface_equals_antisymmetric_images(kernel, so):
    fs, full, dt, w, src, rec, nz, nbl = _pair(kernel, so)
    r_fs, u_fs = fs.forward(src, w, rec, dt=dt, save=True)
    src2 = np.concatenate([src, src * np.array([[1.0, -1.0]])])        # image above the surface
    w2 = np.concatenate([w, -w], axis=1)
    r_full, u_full = full.forward(src2, w2, rec, dt=dt, save=True)
    assert np.abs(r_fs).max() > 0
    assert rel_l2(r_fs, r_full) < 1e-10
    z0 = nbl + nz - 1                                                  # row of the surface in the mirrored grid
    assert rel_l2(u_fs, u_full[:, :, z0:]) < 1e-10
    assert np.abs(u_full[:, :, z0]).max() < 1e-12 * np.abs(u_full).max()   # the extension vanishes on the surface
    # backward propagation (adjoint) through the same update: receivers and their images as sources
    rec2 = np.concatenate([rec, rec * np.array([[1.0, -1.0]])])
    a_fs = fs.adjoint(src, rec, r_fs, dt=dt)
    a_full = full.adjoint(src, rec2, np.concatenate([r_fs, -r_fs], axis=1), dt=dt)
    assert rel_l2(a_fs, a_full) < 1e-10


def test_ot4_free_surface_is_not_ot2():
    fs4, _, dt, w, src, rec, _, _ = _pair("OT4", 8)
    fs2, _, _, _, _, _, _, _ = _pair("OT2", 8)
    dt2 = float(fs2.solver_dt)      # stable for both kernels
    r4 = fs4.forward(src, w, rec, dt=dt2)[0]
    r2 = fs2.forward(src, w, rec, dt=dt2)[0]
    assert rel_l2(r4, r2) > 1e-4
