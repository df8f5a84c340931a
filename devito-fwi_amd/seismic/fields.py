"""
numpy-backed stand-ins for the Devito data types the reference's host code touches
(`Grid`, `Function`, `TimeFunction`, `Constant`): just enough surface for `model.vp.data`,
`grad.data`, `u.data`, `.shape`, `.grid`, `.name` as used by fwi.py:153-171 and
seismic/acoustic/wavesolver.py:104-106,142-143,179-183 of the reference.
"""
import numpy as np


class Grid(object):
    """Cartesian grid: shape, extent, origin, dtype (devito.Grid as used by seismic/model.py:114-120)."""

    def __init__(self, shape, extent=None, origin=None, dtype=np.float32, subdomains=()):
        self.shape = tuple(int(s) for s in shape)
        self.dim = len(self.shape)
        self.dtype = np.dtype(dtype).type
        self.extent = tuple(float(e) for e in (extent if extent is not None else [s - 1 for s in self.shape]))
        self.origin = tuple(self.dtype(o) for o in (origin if origin is not None else (0,) * self.dim))
        self.subdomains = tuple(subdomains)

    @property
    def spacing(self):
        return tuple(self.dtype(e / (s - 1)) if s > 1 else self.dtype(e) for e, s in zip(self.extent, self.shape))

    @property
    def spacing_map(self):
        return {"h_%s" % n: s for n, s in zip("xyz"[:self.dim] if self.dim < 3 else "xyz", self.spacing)}

    @property
    def dimensions(self):
        return tuple("xyz"[:self.dim]) if self.dim == 3 else tuple({1: "x", 2: "xz"}[self.dim])

    def _key(self):
        return (self.shape, tuple(float(e) for e in self.extent), tuple(float(o) for o in self.origin),
                np.dtype(self.dtype).str)

    def __eq__(self, other):
        return isinstance(other, Grid) and self._key() == other._key()

    def __ne__(self, other):
        return not self == other

    def __hash__(self):
        return hash(self._key())

    def __repr__(self):
        return "Grid(shape=%s, extent=%s, origin=%s)" % (self.shape, self.extent, self.origin)


class Function(object):
    """Time-invariant grid field: `.data` is a C-ordered numpy array over the grid."""

    is_Constant = False

    def __init__(self, name, grid, space_order=0, dtype=None, data=None, **kwargs):
        self.name = name
        self.grid = grid
        self.space_order = space_order
        self.dtype = np.dtype(dtype or grid.dtype).type
        self._data = np.zeros(grid.shape, dtype=self.dtype)
        if data is not None:
            self._data[...] = data

    @property
    def data(self):
        return self._data

    @data.setter
    def data(self, value):
        self._data[...] = value

    @property
    def shape(self):
        return self._data.shape

    @property
    def dimensions(self):
        return self.grid.dimensions

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, self.name)


class Constant(object):
    """Scalar parameter (devito.Constant): `.data` is the value."""

    is_Constant = True

    def __init__(self, name, value=0.0, dtype=np.float32):
        self.name = name
        self.dtype = np.dtype(dtype).type
        self._value = self.dtype(value)

    @property
    def data(self):
        return self._value

    @data.setter
    def data(self, value):
        self._value = self.dtype(value)


class TimeFunction(object):
    """
    Time-varying grid field.  `save=None` keeps 3 rotating levels (time order 2, level t at index
    t % 3, as in the code Devito generates); `save=nt` keeps every level.
    `.data` has shape (levels, *grid.shape).
    """

    def __init__(self, name, grid, save=None, time_order=2, space_order=0, dtype=None, **kwargs):
        self.name = name
        self.grid = grid
        self.save = save
        self.time_order = time_order
        self.space_order = space_order
        self.dtype = np.dtype(dtype or grid.dtype).type
        self._levels = int(save) if save else time_order + 1
        self._data = None

    @property
    def shape(self):
        return (self._levels,) + self.grid.shape

    @property
    def data(self):
        if self._data is None:
            self._data = np.zeros(self.shape, dtype=self.dtype)
        return self._data

    @data.setter
    def data(self, value):
        self.data[...] = value


class DeviceTimeFunction(TimeFunction):
    """
    A wavefield that still lives in the GPU plan that produced it.  `.data` downloads it on first
    use; the solver's gradient takes the device copy directly while `token` still matches the
    plan's current save generation.
    """

    def __init__(self, name, grid, save, plan_entry, token, which, shot=0, **kwargs):
        super(DeviceTimeFunction, self).__init__(name, grid, save=save, **kwargs)
        self._entry = plan_entry
        self._token = token
        self._which = which
        self._shot = shot

    def on_device(self):
        return self._data is None and self._entry is not None and self._entry.token(self._which) == self._token

    def materialize(self):
        if self._data is None:
            if self._entry is None or self._entry.token(self._which) != self._token:
                raise RuntimeError("wavefield %s was overwritten on the device before it was read" % self.name)
            self._data = self._entry.plan.read_wavefield(self._which, shot=self._shot)
        self._entry = None
        return self._data

    @property
    def data(self):
        return self.materialize()
