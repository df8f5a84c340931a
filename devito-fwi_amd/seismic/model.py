"""
Host-side physical model: padded grid, edge-padded parameters, sponge profile, CFL time step.

Mirrors the interface of the reference's seismic/model.py (`SeismicModel`, alias `Model`) for the
acoustic case (isotropic or TTI) with numpy arrays in place of Devito functions; every attribute the FWI
drivers and `AcousticWaveSolver` touch is present with the same meaning (reference file:line
given per member).  Elastic / visco parameters are outside this package's scope and raise.
"""
import functools

import numpy as np

from .fields import Grid, Function, Constant

__all__ = ['SeismicModel', 'Model', 'GenericModel', 'initialize_damp']


@functools.lru_cache(maxsize=None)
def _second_derivative_weights(radius):
    """Centred 2nd-derivative FD weights on -radius..radius, index 0 = centre (closed form).
    The array is shared between callers: read only."""
    w = np.zeros(radius + 1)
    for k in range(1, radius + 1):
        q = 1.0
        for j in range(1, k + 1):
            q *= (radius - k + j) / float(radius + j)
        w[k] = (2.0 if k % 2 else -2.0) * q / (k * k)
    w[0] = -2.0 * w[1:].sum()
    w.setflags(write=False)
    return w


def initialize_damp(damp, padsizes, spacing, abc_type="damp", fs=False):
    """
    Fill `damp.data` with the absorbing profile of reference seismic/model.py:13-51: for each
    dimension and side, cell i of the `nbl` outermost cells (i = 0 at the outer edge) gets
    c * (pos - sin(2 pi pos)/(2 pi)) / h added, pos = (nbl - i + 1)/nbl, c = 1.5 ln(1000)/nbl.
    "mask" starts from 1 and subtracts instead.
    """
    sign = -1.0 if abc_type == "mask" else 1.0
    d = damp.data
    d[...] = 1.0 if abc_type == "mask" else 0.0
    for axis, ((nbl, nbr), h) in enumerate(zip(padsizes, spacing)):
        n = d.shape[axis]
        prof = np.zeros(n, dtype=np.float64)
        for side, width in ((0, nbl), (1, nbr)):
            if width <= 0 or (fs and axis == d.ndim - 1 and side == 0):
                continue
            i = np.arange(width)
            pos = np.abs((width - i + 1) / float(width))
            val = (1.5 * np.log(1.0 / 0.001) / width) * (pos - np.sin(2 * np.pi * pos) / (2 * np.pi))
            if side == 0:
                prof[:width] += val
            else:
                prof[n - 1 - i] += val
        shape = [1] * d.ndim
        shape[axis] = n
        d += (sign * prof / float(h)).reshape(shape).astype(d.dtype)


class GenericModel(object):
    """Grid + boundary layer bookkeeping (reference seismic/model.py:87-224)."""

    def __init__(self, origin, spacing, shape, space_order, nbl=20, dtype=np.float32, subdomains=(),
                 bcs="damp", grid=None, fs=False):
        self.shape = tuple(int(s) for s in shape)
        self.space_order = space_order
        self.nbl = int(nbl)
        dtype = np.dtype(dtype).type
        self.origin = tuple(dtype(o) for o in origin)
        self.fs = fs
        origin_pml = [dtype(o - s * self.nbl) for o, s in zip(origin, spacing)]   # model.py:100
        shape_pml = np.array(self.shape) + 2 * self.nbl
        if fs:
            # free surface at the low end of the last dimension: no layer there (model.py:106-110)
            origin_pml[-1] = dtype(origin[-1])
            shape_pml[-1] -= self.nbl
        if grid is None:
            extent = tuple(np.array(spacing, dtype=np.float64) * (shape_pml - 1))
            self.grid = Grid(extent=extent, shape=tuple(shape_pml), origin=origin_pml, dtype=dtype)
        else:
            self.grid = grid
        self._spacing = tuple(dtype(s) for s in spacing)
        self._physical_parameters = set()
        self.damp = None
        self._bcs = None
        self._initialize_bcs(bcs=bcs)

    def _initialize_bcs(self, bcs="damp"):
        """(Re)build the absorbing profile; the wave solver forces bcs="damp" (wavesolver.py:30)."""
        if self.nbl == 0:
            self.damp = 1 if bcs == "mask" else 0
            return
        if bcs == "pml":
            # extension: the absorbing layer is a perfectly matched layer of nbl cells (2-D forward modelling;
            # seismic/abc_methods/03_pml.ipynb of the reference); the sponge field stays zero
            self.damp = Function(name="damp", grid=self.grid)
        elif callable(bcs):
            self.damp = self.damp if isinstance(self.damp, Function) else Function(name="damp", grid=self.grid)
            bcs(self.damp, self.nbl)
        elif self._bcs != bcs or not isinstance(self.damp, Function):
            if not isinstance(self.damp, Function):
                self.damp = Function(name="damp", grid=self.grid)
            initialize_damp(self.damp, self.padsizes, self.spacing, abc_type=bcs, fs=self.fs)
        self._bcs = bcs
        self._physical_parameters.update(['damp'])

    @property
    def padsizes(self):
        """(low, high) layer widths per dimension (model.py:151-158)."""
        pads = [(self.nbl, self.nbl) for _ in range(self.dim)]
        if self.fs:
            pads[-1] = (0, self.nbl)
        return pads

    @property
    def physical_slices(self):
        """Index of the physical domain inside a padded-grid array."""
        return tuple(slice(lo, n - hi) for (lo, hi), n in zip(self.padsizes, self.grid.shape))

    @property
    def physical_parameters(self):
        return tuple(self._physical_parameters)

    def physical_params(self, **kwargs):
        known = [getattr(self, i) for i in self.physical_parameters]
        return {i.name: kwargs.get(i.name, i) or i for i in known if hasattr(i, "name")}

    @property
    def dim(self):
        return self.grid.dim

    @property
    def spacing(self):
        return self._spacing

    @property
    def space_dimensions(self):
        return self.grid.dimensions

    @property
    def spacing_map(self):
        return self.grid.spacing_map

    @property
    def dtype(self):
        return self.grid.dtype

    @property
    def domain_size(self):
        """Physical extent (shape-1)*spacing, model.py:219-224."""
        return tuple((d - 1) * s for d, s in zip(self.shape, self.spacing))


class SeismicModel(GenericModel):
    """
    Isotropic acoustic model.  `vp` in km/s: array over the physical shape (replicated into the
    absorbing layer, model.py:174,387), array over the padded shape, or a scalar.
    `dt=` (keyword) is the local extension of the reference: a user time step that must not exceed
    the CFL value (model.py:277-278,366-369).
    """
    _known_parameters = ['vp', 'damp', 'epsilon', 'delta', 'theta', 'phi']
    _unsupported = ('vs', 'b', 'qp', 'qs', 'lam', 'mu')

    def __init__(self, origin, spacing, shape, space_order, vp, nbl=20, fs=False, dtype=np.float32,
                 subdomains=(), bcs="mask", grid=None, **kwargs):
        for k in self._unsupported:
            if kwargs.get(k) is not None and not (k == 'b' and np.all(np.asarray(kwargs[k]) == 1)):
                raise NotImplementedError("parameter %r: only acoustic (isotropic or TTI) models are supported" % k)
        super(SeismicModel, self).__init__(origin, spacing, shape, space_order, nbl, dtype, subdomains,
                                           grid=grid, bcs=bcs, fs=fs)
        self.vp = self._gen_phys_param(vp, 'vp', space_order)
        # Thomsen parameters and tilt / azimuth angles of a TTI medium (model.py:285-315)
        for name in ('epsilon', 'delta', 'theta', 'phi'):
            if kwargs.get(name) is not None:
                setattr(self, name, self._gen_phys_param(kwargs.get(name), name, space_order))
        self._dt = kwargs.get('dt')
        self._dt_scale = 1

    def _pad(self, field):
        field = np.asarray(field)
        if field.shape == self.grid.shape:
            return field
        if field.shape != self.shape:
            raise ValueError("Incorrect input size %s for model %s without or %s with padding"
                             % (field.shape, self.shape, self.grid.shape))
        if not self.nbl:
            return field
        return self._pad_into(np.empty(self.grid.shape, dtype=field.dtype), field)

    def _pad_into(self, out, field):
        """Write `field` (physical shape) into `out` (padded shape) with edge padding, one dimension after the other:
        what np.pad(mode='edge') returns, without its generic path and without a temporary."""
        out[self.physical_slices] = field
        for d, (lo, hi) in enumerate(self.padsizes):
            n = self.grid.shape[d]
            idx = [slice(None)] * out.ndim
            src = list(idx)
            if lo:
                idx[d], src[d] = slice(0, lo), slice(lo, lo + 1)
                out[tuple(idx)] = out[tuple(src)]
            if hi:
                idx[d], src[d] = slice(n - hi, n), slice(n - hi - 1, n - hi)
                out[tuple(idx)] = out[tuple(src)]
        return out

    def _gen_phys_param(self, field, name, space_order, is_param=True, default_value=0):
        if field is None:
            return default_value
        if isinstance(field, np.ndarray):
            function = Function(name=name, grid=self.grid, space_order=space_order)
            function.data[...] = self._pad(field)
        else:
            function = Constant(name=name, value=field, dtype=self.grid.dtype)
        self._physical_parameters.update([name])
        return function

    @property
    def _max_vp(self):
        return np.max(self.vp.data)

    @property
    def _thomsen_scale(self):
        """sqrt(1 + 2 max(epsilon)) for anisotropic media (model.py:324-329)."""
        if 'epsilon' in self._physical_parameters:
            return np.sqrt(1 + 2 * np.max(self.epsilon.data))
        return 1

    @property
    def dt_scale(self):
        return self._dt_scale

    @dt_scale.setter
    def dt_scale(self, val):
        self._dt_scale = val

    @property
    def _cfl_coeff(self):
        """sqrt(4 / (dim * sum|c|)), c = 2nd-derivative weights on -so..so (model.py:351-353)."""
        w = _second_derivative_weights(self.space_order)
        return np.sqrt(4.0 / float(self.grid.dim * (abs(w[0]) + 2.0 * np.abs(w[1:]).sum())))

    @property
    def critical_dt(self):
        """CFL time step rounded through '%.3e' in the grid dtype (model.py:355-370)."""
        dt = self._cfl_coeff * np.min(self.spacing) / (self._thomsen_scale * self._max_vp)
        dt = self.dtype("%.3e" % (self.dt_scale * dt))
        if self._dt:
            if self._dt > dt:
                raise ValueError("Critical dt: %f, set dt: %f" % (dt, self._dt))
            return self._dt
        return dt

    def update(self, name, value):
        """Replace a physical parameter; physical-shape arrays are re-padded (model.py:372-393)."""
        try:
            param = getattr(self, name)
        except AttributeError:
            setattr(self, name, self._gen_phys_param(value, name, self.space_order))
            return
        if isinstance(value, np.ndarray):
            if isinstance(param, Constant):
                setattr(self, name, self._gen_phys_param(value, name, self.space_order))
            elif self.nbl and value.shape == self.shape and value.shape != self.grid.shape \
                    and getattr(param.data, 'shape', None) == self.grid.shape:
                self._pad_into(param.data, value)
            else:
                param.data[...] = self._pad(value)
        else:
            param.data = value

    @property
    def m(self):
        """Squared slowness 1/vp^2 (model.py:395-400), as an array/scalar."""
        v = self.vp.data
        return 1 / (v * v)

    def smooth(self, physical_parameters, sigma=5.0):
        from scipy.ndimage import gaussian_filter
        for name in (physical_parameters if not isinstance(physical_parameters, str) else (physical_parameters,)):
            f = getattr(self, name)
            f.data[...] = gaussian_filter(f.data, sigma=sigma, mode='nearest')


Model = SeismicModel
