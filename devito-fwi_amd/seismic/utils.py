"""
Acquisition geometry: where the sources and receivers are, the time axis and the wavelet.
Interface of the reference's seismic/utils.py (AcquisitionGeometry :50-189 with its local
`resample` :89-91 and source `filter` :181-184 additions; setup_geometry :12-47).
"""
import warnings

import numpy as np

from .source import TimeAxis, PointSource, Receiver, WaveletSource, RickerSource, GaborSource

__all__ = ['AcquisitionGeometry', 'setup_geometry', 'setup_rec_coords']

sources = {'Wavelet': WaveletSource, 'Ricker': RickerSource, 'Gabor': GaborSource}


def setup_rec_coords(model):
    """One receiver per x (and y) grid column, two cells below the top (utils.py:27-47)."""
    axes = [np.linspace(model.origin[d], model.domain_size[d], model.shape[d]) for d in range(model.dim - 1)]
    if model.dim == 1:
        return np.linspace(model.origin[0], model.domain_size[0], model.shape[0]).reshape(-1, 1)
    mesh = np.meshgrid(*axes, indexing='ij')
    rec = np.empty((mesh[0].size, model.dim))
    for d in range(model.dim - 1):
        rec[:, d] = mesh[d].ravel()
    rec[:, -1] = model.origin[-1] + 2 * model.spacing[-1]
    return rec


def setup_geometry(model, tn, f0=0.010):
    """Single Ricker source in the middle, one cell deep, plus setup_rec_coords (utils.py:12-24)."""
    src = np.empty((1, model.dim))
    src[0, :] = np.array(model.domain_size) * .5
    if model.dim > 1:
        src[0, -1] = model.origin[-1] + model.spacing[-1]
    return AcquisitionGeometry(model, setup_rec_coords(model), src, t0=0.0, tn=tn, src_type='Ricker', f0=f0)


class AcquisitionGeometry(object):
    """Positions, time range and wavelet of one acquisition (utils.py:60-186)."""

    def __init__(self, model, rec_positions, src_positions, t0, tn, **kwargs):
        self.src_positions = np.reshape(src_positions, (-1, model.dim))
        self.rec_positions = np.reshape(rec_positions, (-1, model.dim))
        self._nrec = self.rec_positions.shape[0]
        self._nsrc = self.src_positions.shape[0]
        self._src_type = kwargs.get('src_type')
        assert self._src_type in sources or self._src_type is None
        self._f0 = kwargs.get('f0')
        self._a = kwargs.get('a', None)
        self._t0w = kwargs.get('t0w', None)
        if self._src_type is not None and self._f0 is None:
            raise ValueError("Peak frequency must be provided in KH for source of type %s" % self._src_type)
        self._grid = model.grid
        self._model = model
        self._dt = model.critical_dt
        self._t0 = t0
        self._tn = tn
        self._src_data = kwargs.get('src_data', None)
        self._filter = kwargs.get('filter', None)

    def resample(self, dt):
        """Change the sampling of the traces only; the solver keeps stepping with model.critical_dt."""
        self._dt = dt
        return self

    time_axis = property(lambda self: TimeAxis(start=self.t0, stop=self.tn, step=self.dt))
    src_type = property(lambda self: self._src_type)
    grid = property(lambda self: self._grid)
    model = property(lambda self: self._model)
    f0 = property(lambda self: self._f0)
    tn = property(lambda self: self._tn)
    t0 = property(lambda self: self._t0)
    dt = property(lambda self: self._dt)
    nt = property(lambda self: self.time_axis.num)
    nrec = property(lambda self: self._nrec)
    nsrc = property(lambda self: self._nsrc)
    dtype = property(lambda self: self.grid.dtype)
    rec = property(lambda self: self.new_rec())
    src = property(lambda self: self.new_src())

    def new_rec(self, name='rec'):
        return Receiver(name=name, grid=self.grid, time_range=self.time_axis, npoint=self.nrec,
                        coordinates=self.rec_positions)

    @property
    def adj_src(self):
        if self.src_type is None:
            warnings.warn("No source type defined, returning uninitiallized (zero) shot record")
            return self.new_rec()
        adj = sources[self.src_type](name='rec', grid=self.grid, f0=self.f0, time_range=self.time_axis,
                                     npoint=self.nrec, coordinates=self.rec_positions, t0=self._t0w, a=self._a)
        rev = adj.wavelet[::-1]
        for i in range(self.nrec):
            adj.data[:, i] = rev
        return adj

    def new_src(self, name='src', src_type='self'):
        if self.src_type is None or src_type is None:
            return PointSource(name=name, grid=self.grid, time_range=self.time_axis, npoint=self.nsrc,
                               coordinates=self.src_positions)
        source = sources[self.src_type](name=name, grid=self.grid, f0=self.f0, time_range=self.time_axis,
                                        npoint=self.nsrc, coordinates=self.src_positions, t0=self._t0w, a=self._a)
        if self._filter is not None:
            self._filter.df = 1000 / self._dt
            for i in range(self.nsrc):
                source.data[:, i] = self._filter(source.data[:, i])
        return source
