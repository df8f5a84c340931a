"""
TEST INFRASTRUCTURE ONLY.  CPU oracle for the acoustic FWI hot path of LongyanU/devito-fwi.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker / reported CPU baseline.  The shipped package never does.

Two layers:
  * ctypes bindings to oracle/liboracle.so (acoustic_oracle.c): forward / adjoint / gradient
    time loops, sponge profile, FD weights, CFL coefficient.
  * numpy restatements of the host-side set-up and FWI glue the reference runs around the
    propagators, each citing the reference file:line it follows.

Parity status: PINNED (tests/test_oracle_kat.py reproduces seismic/inversion/fwi.py:95-97,121
and seismic/acoustic/acoustic_example.py:75-79).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/liboracle.so with the committed Makefile (gcc + OpenMP)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("acoustic_oracle.c", "acoustic_oracle_impl.h", "Makefile")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.oracle_cfl_coeff.restype = ctypes.c_double
        _LIB.oracle_cfl_coeff.argtypes = [ctypes.c_int, ctypes.c_int]
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _ia(v):
    return (ctypes.c_int * len(v))(*[int(x) for x in v])


def _da(v):
    return (ctypes.c_double * len(v))(*[float(x) for x in v])


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(dtype)


# ----------------------------------------------------------------------------------------------
# numerical constants
# ----------------------------------------------------------------------------------------------
def fd2_weights(radius):
    """w[0..R] of the centred 2nd-derivative stencil (sympy fd_w(2, range(-R,R+1), 0)[-1][-1])."""
    w = np.zeros(radius + 1)
    lib().oracle_fd2_weights(ctypes.c_int(radius), _p(w))
    return w


def cfl_coeff(space_order, ndim):
    """seismic/model.py:338-353."""
    return lib().oracle_cfl_coeff(space_order, ndim)


def critical_dt(vp_max, spacing, space_order, dtype=np.float32, dt=None):
    """seismic/model.py:355-370, including the local edit honouring a user dt."""
    ndim = len(spacing)
    cdt = cfl_coeff(space_order, ndim) * np.min(spacing) / vp_max
    cdt = np.dtype(dtype).type("%.3e" % cdt)
    if dt:
        if dt > cdt:
            raise ValueError("Critical dt: %f, set dt: %f" % (cdt, dt))
        return dt
    return cdt


def sponge(shape_pad, nbl, spacing, dtype=np.float32, fs=False):
    """seismic/model.py:13-51 (abc_type='damp'); with fs no layer at the low end of the last dim."""
    dtype = np.dtype(dtype)
    out = np.zeros(tuple(shape_pad), dtype=dtype)
    fn = getattr(lib(), "oracle_sponge_" + _sfx(dtype))
    rc = fn(len(shape_pad), _ia(shape_pad), int(nbl), _da(spacing), _p(out), int(bool(fs)))
    assert rc == 0
    return out


def pad_vp(vp, nbl, fs=False):
    """seismic/model.py:157,174,387: initialize_function replicates edge values into the layer
    (padsizes (nbl, nbl) per dimension, (0, nbl) for the last one under a free surface)."""
    if nbl <= 0:
        return np.array(vp)
    widths = [(nbl, nbl)] * vp.ndim
    if fs:
        widths[-1] = (0, nbl)
    return np.pad(vp, widths, mode="edge")


def time_axis(t0, tn, dt):
    """seismic/source.py:42-75: num = ceil((stop-start+step)/step); values = linspace."""
    num = int(np.ceil((tn - t0 + dt) / dt))
    stop = dt * (num - 1) + t0
    return num, np.linspace(t0, stop, num)


def ricker(f0, time_values, t0=None, a=None):
    """seismic/source.py:272-277."""
    t0 = t0 or 1.0 / f0
    a = a or 1
    r = np.pi * f0 * (time_values - t0)
    return a * (1 - 2.0 * r ** 2) * np.exp(-r ** 2)


# ----------------------------------------------------------------------------------------------
# propagators
# ----------------------------------------------------------------------------------------------
class Problem(object):
    """
    One padded acoustic model + discretisation, as `SeismicModel` + `AcousticWaveSolver`
    define it (seismic/model.py:91-124,227-283; seismic/acoustic/wavesolver.py:28-46).

    vp      physical-domain velocity (km/s), any of 1-3 dims
    """

    def __init__(self, vp, spacing, nbl, space_order, origin=None, dtype=np.float32, dt=None,
                 model_space_order=None, fs=False, kernel='OT2'):
        self.dtype = np.dtype(dtype)
        vp = np.asarray(vp, dtype=self.dtype)
        self.ndim = vp.ndim
        self.shape = vp.shape
        self.nbl = int(nbl)
        self.spacing = tuple(float(s) for s in spacing)
        self.origin = tuple(float(o) for o in (origin or (0.0,) * self.ndim))
        self.space_order = int(space_order)
        self.model_space_order = int(model_space_order or space_order)
        self.user_dt = dt
        # model.py:100: origin_pml = dtype(o - s*nbl)
        self.fs = bool(fs)
        assert kernel in ('OT2', 'OT4')
        self.ot4 = int(kernel == 'OT4')
        origin_pad = [float(self.dtype.type(o - s * self.nbl)) for o, s in zip(self.origin, self.spacing)]
        shape_pad = [s + 2 * self.nbl for s in self.shape]
        if self.fs:                                   # model.py:106-110
            origin_pad[-1] = float(self.dtype.type(self.origin[-1]))
            shape_pad[-1] -= self.nbl
        self.origin_pad, self.shape_pad = tuple(origin_pad), tuple(shape_pad)
        self.damp = sponge(self.shape_pad, self.nbl, self.spacing, self.dtype, self.fs)
        self.set_vp(vp)

    def set_vp(self, vp):
        vp = np.asarray(vp, dtype=self.dtype)
        if vp.shape == self.shape:
            self.vp = np.ascontiguousarray(pad_vp(vp, self.nbl, self.fs))
        elif vp.shape == self.shape_pad:
            self.vp = np.ascontiguousarray(vp)
        else:
            raise ValueError("bad vp shape %s" % (vp.shape,))

    @property
    def solver_dt(self):
        """Time step the solver uses: 1.73 x critical_dt for OT4 (wavesolver.py:41-46)."""
        return self.dtype.type(1.73 * self.critical_dt) if self.ot4 else self.critical_dt

    @property
    def critical_dt(self):
        return critical_dt(float(self.vp.max()), self.spacing, self.model_space_order, self.dtype, self.user_dt)

    def _coords(self, c):
        return np.ascontiguousarray(np.asarray(c, dtype=self.dtype).reshape(-1, self.ndim))

    def forward(self, src_coords, wavelet, rec_coords, dt=None, save=False, vp=None, return_last=False):
        """operators.py:98-140.  wavelet (nt, nsrc).  Returns rec (nt,nrec) [, u (nt,*shape_pad)]."""
        dt = float(self.critical_dt if dt is None else dt)
        vp = self.vp if vp is None else np.ascontiguousarray(vp, dtype=self.dtype)
        sc, rc_ = self._coords(src_coords), self._coords(rec_coords)
        w = np.ascontiguousarray(np.asarray(wavelet, dtype=self.dtype).reshape(-1, sc.shape[0]))
        nt = w.shape[0]
        rec = np.zeros((nt, rc_.shape[0]), dtype=self.dtype)
        u = np.zeros((nt,) + self.shape_pad, dtype=self.dtype) if save else None
        ul = np.zeros((3,) + self.shape_pad, dtype=self.dtype) if return_last else None
        fn = getattr(lib(), "oracle_forward_" + _sfx(self.dtype))
        rc = fn(self.ndim, _ia(self.shape_pad), _da(self.spacing), _da(self.origin_pad), self.space_order,
                nt, ctypes.c_double(dt), _p(vp), _p(self.damp), sc.shape[0], _p(sc), _p(w),
                rc_.shape[0], _p(rc_), _p(rec), _p(u), _p(ul), int(self.fs), self.ot4)
        assert rc == 0, rc
        out = (rec, u)
        return out + (ul,) if return_last else out

    def adjoint(self, src_coords, rec_coords, rec, dt=None, vp=None):
        """operators.py:143-180.  rec (nt,nrec) -> srca (nt,nsrc)."""
        dt = float(self.critical_dt if dt is None else dt)
        vp = self.vp if vp is None else np.ascontiguousarray(vp, dtype=self.dtype)
        sc, rc_ = self._coords(src_coords), self._coords(rec_coords)
        rec = np.ascontiguousarray(rec, dtype=self.dtype)
        nt = rec.shape[0]
        srca = np.zeros((nt, sc.shape[0]), dtype=self.dtype)
        fn = getattr(lib(), "oracle_adjoint_" + _sfx(self.dtype))
        rc = fn(self.ndim, _ia(self.shape_pad), _da(self.spacing), _da(self.origin_pad), self.space_order,
                nt, ctypes.c_double(dt), _p(vp), _p(self.damp), sc.shape[0], _p(sc), _p(srca),
                rc_.shape[0], _p(rc_), _p(rec), None, int(self.fs), self.ot4)
        assert rc == 0, rc
        return srca

    def gradient(self, rec_coords, residual, u, dt=None, vp=None, grad=None):
        """operators.py:183-225.  Accumulates into grad (padded grid) and returns it."""
        dt = float(self.critical_dt if dt is None else dt)
        vp = self.vp if vp is None else np.ascontiguousarray(vp, dtype=self.dtype)
        rc_ = self._coords(rec_coords)
        residual = np.ascontiguousarray(residual, dtype=self.dtype)
        u = np.ascontiguousarray(u, dtype=self.dtype)
        nt = residual.shape[0]
        assert u.shape == (nt,) + self.shape_pad
        if grad is None:
            grad = np.zeros(self.shape_pad, dtype=self.dtype)
        assert grad.flags.c_contiguous and grad.dtype == self.dtype
        fn = getattr(lib(), "oracle_gradient_" + _sfx(self.dtype))
        rc = fn(self.ndim, _ia(self.shape_pad), _da(self.spacing), _da(self.origin_pad), self.space_order,
                nt, ctypes.c_double(dt), _p(vp), _p(self.damp), rc_.shape[0], _p(rc_), _p(residual),
                _p(u), _p(grad), int(self.fs), self.ot4)
        assert rc == 0, rc
        return grad

    def born(self, src_coords, wavelet, rec_coords, dm, dt=None, vp=None):
        """operators.py:228-273.  dm over the padded grid.  Returns the linearised data (nt,nrec)."""
        dt = float(self.critical_dt if dt is None else dt)
        vp = self.vp if vp is None else np.ascontiguousarray(vp, dtype=self.dtype)
        sc, rc_ = self._coords(src_coords), self._coords(rec_coords)
        w = np.ascontiguousarray(np.asarray(wavelet, dtype=self.dtype).reshape(-1, sc.shape[0]))
        dm = np.ascontiguousarray(dm, dtype=self.dtype)
        assert dm.shape == self.shape_pad
        nt = w.shape[0]
        rec = np.zeros((nt, rc_.shape[0]), dtype=self.dtype)
        fn = getattr(lib(), "oracle_born_" + _sfx(self.dtype))
        rc = fn(self.ndim, _ia(self.shape_pad), _da(self.spacing), _da(self.origin_pad), self.space_order,
                nt, ctypes.c_double(dt), _p(vp), _p(self.damp), _p(dm), sc.shape[0], _p(sc), _p(w),
                rc_.shape[0], _p(rc_), _p(rec), int(self.fs))
        assert rc == 0, rc
        return rec


class TTIProblem(Problem):
    """Tilted transversely isotropic medium, 'centered' kernel (seismic/tti/operators.py,
    seismic/tti/wavesolver.py:80-215; parameters as seismic/model.py:285-315).  epsilon, delta, theta
    (and phi in 3-D) over the physical grid; they are edge-padded into the absorbing layer like vp."""

    def __init__(self, vp, spacing, nbl, space_order, epsilon, delta, theta, phi=None, **kw):
        super(TTIProblem, self).__init__(vp, spacing, nbl, space_order, **kw)
        assert not self.fs and not self.ot4 and space_order % 4 == 0

        def padded(a):
            return None if a is None else np.ascontiguousarray(pad_vp(np.asarray(a, dtype=self.dtype), self.nbl))
        self.epsilon, self.delta, self.theta, self.phi = padded(epsilon), padded(delta), padded(theta), padded(phi)

    @property
    def critical_dt(self):
        """model.py:324-329,355-370: the CFL velocity is scaled by sqrt(1 + 2 max(epsilon))."""
        scale = np.sqrt(1 + 2 * float(self.epsilon.max()))
        return critical_dt(scale * float(self.vp.max()), self.spacing, self.model_space_order, self.dtype,
                           self.user_dt)

    def _run(self, name, src_coords, src, rec_coords, rec, dt):
        fn = getattr(lib(), name + _sfx(self.dtype))
        sc, rc_ = self._coords(src_coords), self._coords(rec_coords)
        nt = src.shape[0]
        r = fn(self.ndim, _ia(self.shape_pad), _da(self.spacing), _da(self.origin_pad), self.space_order, nt,
               ctypes.c_double(dt), _p(self.vp), _p(self.damp), _p(self.epsilon), _p(self.delta), _p(self.theta),
               _p(self.phi), sc.shape[0], _p(sc), _p(src), rc_.shape[0], _p(rc_), _p(rec))
        assert r == 0, r

    def forward(self, src_coords, wavelet, rec_coords, dt=None):
        dt = float(self.critical_dt if dt is None else dt)
        nsrc = self._coords(src_coords).shape[0]
        w = np.ascontiguousarray(np.asarray(wavelet, dtype=self.dtype).reshape(-1, nsrc))
        rec = np.zeros((w.shape[0], self._coords(rec_coords).shape[0]), dtype=self.dtype)
        self._run("oracle_tti_forward_", src_coords, w, rec_coords, rec, dt)
        return rec

    def adjoint(self, src_coords, rec_coords, rec, dt=None):
        dt = float(self.critical_dt if dt is None else dt)
        rec = np.ascontiguousarray(rec, dtype=self.dtype)
        srca = np.zeros((rec.shape[0], self._coords(src_coords).shape[0]), dtype=self.dtype)
        self._run("oracle_tti_adjoint_", src_coords, srca, rec_coords, rec, dt)
        return srca


    def born(self, src_coords, wavelet, rec_coords, dm, dt=None):
        """seismic/tti/operators.py:529-587.  dm over the padded grid; returns the linearised data."""
        dt = float(self.critical_dt if dt is None else dt)
        sc, rc_ = self._coords(src_coords), self._coords(rec_coords)
        w = np.ascontiguousarray(np.asarray(wavelet, dtype=self.dtype).reshape(-1, sc.shape[0]))
        dm = np.ascontiguousarray(dm, dtype=self.dtype)
        assert dm.shape == self.shape_pad
        rec = np.zeros((w.shape[0], rc_.shape[0]), dtype=self.dtype)
        fn = getattr(lib(), "oracle_tti_born_" + _sfx(self.dtype))
        r = fn(self.ndim, _ia(self.shape_pad), _da(self.spacing), _da(self.origin_pad), self.space_order,
               w.shape[0], ctypes.c_double(dt), _p(self.vp), _p(self.damp), _p(self.epsilon), _p(self.delta),
               _p(self.theta), _p(self.phi), _p(dm), sc.shape[0], _p(sc), _p(w), rc_.shape[0], _p(rc_), _p(rec))
        assert r == 0, r
        return rec

    def gradient(self, src_coords, wavelet, rec_coords, residual, dt=None, grad=None):
        """seismic/tti/operators.py:590-639: forward modelling with both fields saved, then the adjoint
        fields driven by `residual` (nt, nrec).  Accumulates into grad (padded grid) and returns it."""
        dt = float(self.critical_dt if dt is None else dt)
        sc, rc_ = self._coords(src_coords), self._coords(rec_coords)
        w = np.ascontiguousarray(np.asarray(wavelet, dtype=self.dtype).reshape(-1, sc.shape[0]))
        residual = np.ascontiguousarray(residual, dtype=self.dtype)
        assert residual.shape == (w.shape[0], rc_.shape[0])
        if grad is None:
            grad = np.zeros(self.shape_pad, dtype=self.dtype)
        assert grad.flags.c_contiguous and grad.dtype == self.dtype
        fn = getattr(lib(), "oracle_tti_gradient_" + _sfx(self.dtype))
        r = fn(self.ndim, _ia(self.shape_pad), _da(self.spacing), _da(self.origin_pad), self.space_order,
               w.shape[0], ctypes.c_double(dt), _p(self.vp), _p(self.damp), _p(self.epsilon), _p(self.delta),
               _p(self.theta), _p(self.phi), sc.shape[0], _p(sc), _p(w), rc_.shape[0], _p(rc_), _p(residual),
               _p(grad))
        assert r == 0, r
        return grad


# ----------------------------------------------------------------------------------------------
# FWI glue (fwi.py of the reference)
# ----------------------------------------------------------------------------------------------
def least_square(syn, obs):
    """misfit/misfit.py:5-9."""
    res = syn - obs
    return 0.5 * np.linalg.norm(res.flatten()) ** 2, res


def fix_source_illumination(shape, spacing, src_pos, rec_pos, g):
    """fwi.py:104-129, including its meshgrid(z, x) transposition quirk (fwi.py:115)."""
    dx, dz = spacing
    sx, sz = src_pos[0], src_pos[1]
    nx, nz = shape
    x = np.arange(0, nx) * dx
    z = np.arange(0, nz) * dz
    xx, zz = np.meshgrid(z, x)
    sigma = dx + dz
    mask = np.exp(-.5 * ((xx - sx) ** 2 + (zz - sz) ** 2) / (sigma ** 2))
    g = g * (1. - mask)
    for i in range(rec_pos.shape[0]):
        rx, rz = rec_pos[i, 0], rec_pos[i, 1]
        mask = np.exp(-.5 * ((xx - rx) ** 2 + (zz - rz) ** 2) / (sigma ** 2))
        g = g * (1. - mask)
    return g


def fwi_obj_single(prob, src_pos, rec_pos, wavelet, obs, dt=None, calc_grad=True):
    """fwi.py:131-173 with the L2 misfit and no resampling (dt of data == dt of solver)."""
    nbl = prob.nbl
    pred, u = prob.forward(src_pos, wavelet, rec_pos, dt=dt, save=calc_grad)
    fval, res = least_square(pred, obs)
    if not calc_grad:
        return fval, None, res, None
    grad = prob.gradient(rec_pos, res, u, dt=dt)
    crop = tuple(slice(nbl, -nbl) for _ in range(prob.ndim))
    g = fix_source_illumination(prob.shape, prob.spacing, np.ravel(src_pos), np.asarray(rec_pos), np.array(grad)[crop])
    illum = (u * u).sum(axis=0)[crop]
    illum = fix_source_illumination(prob.shape, prob.spacing, np.ravel(src_pos), np.asarray(rec_pos), illum)
    return fval, g, res, illum


def fwi_obj_multi(prob, src_positions, rec_pos, wavelet, obs_list, dt=None, mask=None, precond=True,
                  calc_grad=True):
    """fwi.py:175-205."""
    fval = 0.0
    grad = np.zeros(prob.shape)
    illum = np.zeros(prob.shape)
    residuals = []
    for i in range(len(src_positions)):
        f_, g_, r_, il_ = fwi_obj_single(prob, src_positions[i], rec_pos, wavelet, obs_list[i], dt, calc_grad)
        fval += f_
        residuals.append(r_)
        if calc_grad:
            grad += g_
            illum += il_
    if calc_grad:
        if precond:
            grad /= np.sqrt(illum + 1e-30)
        if mask is not None:
            grad *= mask
    return fval, grad.reshape(-1).astype(np.float64), residuals


# ----------------------------------------------------------------------------------------------
# Perfectly matched layer (2-D): numpy restatement of seismic/abc_methods/03_pml.ipynb
# (PDE cell 2, damping function cell 60, discrete equations cells 75-76, operator order cell 85),
# generalised to a layer on all four sides and to the Laplacian of any even space order.
# The reference holds no number of the acoustic hot path with this boundary (the notebook's own last
# cell only bounds ||rec|| to within a factor of two): parity of the GPU path is against this
# restatement, which is itself constrained by properties (tests/test_pml.py).
# ----------------------------------------------------------------------------------------------
def pml_profiles(n, npml, quibar=0.05, low=True, high=True):
    """zeta at the nodes and at the staggered points of a dimension of n nodes (fdamp, cell 60); `low` / `high`:
    whether the dimension has a layer at its first / last nodes (the notebook has none at the top, cell 58)."""
    def zeta(pos):
        a = np.zeros_like(pos)
        if low:
            a = np.where(pos <= npml, (npml - pos) / npml, a)
        if high:
            a = np.where(pos >= n - 1 - npml, (pos - (n - 1 - npml)) / npml, a)
        return quibar * (a - np.sin(2 * np.pi * a) / (2 * np.pi))
    i = np.arange(n, dtype=np.float64)
    return zeta(i), zeta(i + 0.5)


def _support_2d(coords, origin, spacing, shape, dtype):
    """(index pairs, weights) of the bilinear support of each point, in the grid dtype as the generated code
    computes it (sa_01_iso_implementation1.ipynb:1255-1316); corners outside the grid are dropped."""
    out = []
    t = np.dtype(dtype).type
    for c in np.asarray(coords, dtype=dtype).reshape(-1, 2):
        pts = []
        pos = [t(c[d]) - t(origin[d]) for d in range(2)]
        idx = [int(np.floor(pos[d] / t(spacing[d]))) for d in range(2)]
        frac = [(pos[d] - t(spacing[d]) * t(idx[d])) / t(spacing[d]) for d in range(2)]
        for bx in (0, 1):
            for bz in (0, 1):
                i, j = idx[0] + bx, idx[1] + bz
                w = (frac[0] if bx else t(1) - frac[0]) * (frac[1] if bz else t(1) - frac[1])
                if 0 <= i < shape[0] and 0 <= j < shape[1]:
                    pts.append((i, j, t(w)))
        out.append(pts)
    return out


def pml_forward(vp_pad, spacing, origin_pad, space_order, dt, src_coords, wavelet, rec_coords, npml, quibar=0.05,
                dtype=np.float32, return_last=False, sides=(True, True, True, True), edges="halo", vp_staggered=None):
    """Shot record (nt, nrec) of the Grote-Sim scheme on the padded grid vp_pad (km/s).

    Default: a layer on all four sides, fields zero outside the grid.  `sides` = (low x, high x, low z, high z)
    selects the sides that have a layer.  edges="notebook" restates 03_pml.ipynb literally: the damped update and
    the auxiliary fields only at the nodes of the layer regions (its subdomains d1-d3, cells 20-22, 76-81), the plain
    wave equation elsewhere, and after the source term the boundary rows of cell 83 (u = 0 on the first and last x
    node and on the last z node, u[x, 0] = u[x, 1]); `vp_staggered`: the velocity at the staggered points as the
    caller's own array (the notebook's vel1, cells 26, 51-53) instead of the mean of c^2 over a cell's corners."""
    t = np.dtype(dtype).type
    vp = np.asarray(vp_pad, dtype=dtype)
    n1, n2 = vp.shape
    R = space_order // 2
    w = fd2_weights(R)
    hx, hz = float(spacing[0]), float(spacing[1])
    c0 = t(w[0] / hx ** 2 + w[0] / hz ** 2)
    cx = [t(w[k] / hx ** 2) for k in range(R + 1)]
    cz = [t(w[k] / hz ** 2) for k in range(R + 1)]
    m = (t(1) / (vp * vp)).astype(dtype)
    zx0, zx1 = [a.astype(dtype) for a in pml_profiles(n1, npml, quibar, sides[0], sides[1])]
    zz0, zz1 = [a.astype(dtype) for a in pml_profiles(n2, npml, quibar, sides[2], sides[3])]
    notebook = edges == "notebook"
    if notebook:   # staggered profiles: the last entry repeats the one before it (cells 72-74)
        zx1[-1], zz1[-1] = zx1[-2], zz1[-2]
    ZX0, ZZ0 = zx0[:, None], zz0[None, :]
    ZX1, ZZ1 = zx1[:, None], zz1[None, :]
    # c^2 at the staggered points: mean of 1/m over the corners of the cell that lie inside the grid
    inv = (t(1) / m).astype(dtype)
    acc = inv.copy()
    cnt = np.ones((n1, n2), dtype=dtype)
    acc[:-1, :] += inv[1:, :]; cnt[:-1, :] += 1
    acc[:, :-1] += inv[:, 1:]; cnt[:, :-1] += 1
    acc[:-1, :-1] += inv[1:, 1:]; cnt[:-1, :-1] += 1
    c2s = (acc / cnt).astype(dtype)
    if vp_staggered is not None:
        vs = np.asarray(vp_staggered, dtype=dtype)
        c2s = (vs * vs).astype(dtype)
    H = R + 1
    def padz(a):
        return np.pad(a, H)
    sl = (slice(H, H + n1), slice(H, H + n2))
    def sh(ap, dx, dz):
        return ap[H + dx:H + dx + n1, H + dz:H + dz + n2]
    nt = wavelet.shape[0]
    wav = np.asarray(wavelet, dtype=dtype).reshape(nt, -1)
    s_sup = _support_2d(src_coords, origin_pad, spacing, (n1, n2), dtype)
    r_sup = _support_2d(rec_coords, origin_pad, spacing, (n1, n2), dtype)
    rec = np.zeros((nt, len(r_sup)), dtype=dtype)
    u = [np.zeros((n1, n2), dtype=dtype) for _ in range(3)]
    p1 = [np.zeros((n1, n2), dtype=dtype) for _ in range(2)]
    p2 = [np.zeros((n1, n2), dtype=dtype) for _ in range(2)]
    dt_, dt2 = t(dt), t(dt) * t(dt)
    half_s = (t(0.5) * (ZX0 + ZZ0) * dt_).astype(dtype)
    layer = (ZX1 != 0) | (ZZ1 != 0)
    if notebook:   # the layer regions by node index (subdomains d1, d2, d3 and their mirror image at the top)
        i1, i2 = np.arange(n1)[:, None], np.arange(n2)[None, :]
        layer = np.zeros((n1, n2), dtype=bool)
        if sides[0]: layer |= (i1 < npml) & (i2 >= 0)
        if sides[1]: layer |= (i1 >= n1 - npml) & (i2 >= 0)
        if sides[2]: layer |= (i2 < npml) & (i1 >= 0)
        if sides[3]: layer |= (i2 >= n2 - npml) & (i1 >= 0)
    for time in range(1, nt - 1):
        cur, old, out = u[time % 3], u[(time + 2) % 3], u[(time + 1) % 3]
        f1, f2 = padz(p1[time & 1]), padz(p2[time & 1])
        cp = padz(cur)
        lap = c0 * cur
        for k in range(1, R + 1):
            lap = lap + cz[k] * (sh(cp, 0, -k) + sh(cp, 0, k))
        for k in range(1, R + 1):
            lap = lap + cx[k] * (sh(cp, -k, 0) + sh(cp, k, 0))
        P1 = -t(0.5 / hx) * (sh(f1, 0, -1) + sh(f1, 0, 0) - sh(f1, -1, -1) - sh(f1, -1, 0))
        P2 = -t(0.5 / hz) * (sh(f2, -1, 0) + sh(f2, 0, 0) - sh(f2, -1, -1) - sh(f2, 0, -1))
        rhs = t(2) * cur - old * (t(1) - half_s) + dt2 * (lap / m - ZX0 * ZZ0 * cur - P1 - P2)
        out[...] = (rhs / (t(1) + half_s)).astype(dtype)
        if notebook:
            out[...] = np.where(layer, out, (t(2) * cur - old + dt2 * (lap / m)).astype(dtype))
        for p_, pts in enumerate(s_sup):                      # source into u[t+1] (cell 55)
            for i, j, wgt in pts:
                out[i, j] += wgt * wav[time, p_] * dt2 / m[i, j]
        if notebook:                                          # boundary rows, in the order of cell 83
            out[0, :] = 0
            out[n1 - 1, :] = 0
            out[:, n2 - 1] = 0
            out[:, 0] = out[:, 1]
        for p_, pts in enumerate(r_sup):                      # receivers sample u[t] (cell 56)
            rec[time, p_] = sum((wgt * cur[i, j] for i, j, wgt in pts), t(0))
        op, cq = padz(out), padz(cur)
        a = (sh(op, 1, 0) + sh(op, 1, 1) - sh(op, 0, 0) - sh(op, 0, 1)) + (sh(cq, 1, 0) + sh(cq, 1, 1) - sh(cq, 0, 0) - sh(cq, 0, 1))
        b = (sh(op, 0, 1) + sh(op, 1, 1) - sh(op, 0, 0) - sh(op, 1, 0)) + (sh(cq, 0, 1) + sh(cq, 1, 1) - sh(cq, 0, 0) - sh(cq, 1, 0))
        idt = t(1) / dt_
        n1_ = (p1[time & 1] * (idt - t(0.5) * ZX1) + (ZZ1 - ZX1) * c2s * t(0.25 / hx) * a) / (idt + t(0.5) * ZX1)
        n2_ = (p2[time & 1] * (idt - t(0.5) * ZZ1) + (ZX1 - ZZ1) * c2s * t(0.25 / hz) * b) / (idt + t(0.5) * ZZ1)
        p1[(time + 1) & 1][...] = np.where(layer, n1_, 0).astype(dtype)
        p2[(time + 1) & 1][...] = np.where(layer, n2_, 0).astype(dtype)
    if return_last:
        return rec, np.stack([u[k] for k in range(3)])
    return rec
