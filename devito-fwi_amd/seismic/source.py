"""
Time axis, sparse point sources / receivers and analytic wavelets, numpy-backed.

Interface of the reference's seismic/source.py (TimeAxis :15-75, PointSource :78-174,
Receiver = Shot = PointSource :177-178, RickerSource :248-277, GaborSource :280-310,
DGaussSource :313-351); data live in plain arrays instead of Devito SparseTimeFunctions.
"""
import numpy as np
from scipy import interpolate

__all__ = ['PointSource', 'Receiver', 'Shot', 'WaveletSource',
           'RickerSource', 'GaborSource', 'DGaussSource', 'TimeAxis']


class TimeAxis(object):
    """
    start/stop/step/num with exactly three given (source.py:42-64).  When `num` is derived it is
    ceil((stop - start + step)/step) and `stop` is moved to start + step*(num-1).
    """

    def __init__(self, start=None, step=None, num=None, stop=None):
        given = [v is not None for v in (start, step, num, stop)]
        if sum(given) != 3:
            raise ValueError("Three of args start, step, num and stop may be set")
        try:
            if start is None:
                start = stop - step * (num - 1)
            elif step is None:
                step = (stop - start) / (num - 1)
            elif num is None:
                num = int(np.ceil((stop - start + step) / step))
                stop = start + step * (num - 1)
            else:
                stop = start + step * (num - 1)
        except Exception:
            raise ValueError("Three of args start, step, num and stop may be set")
        if not isinstance(num, int):
            raise TypeError("input argument must be of type int")
        self.start, self.stop, self.step, self.num = start, stop, step, num
        self._values = None

    def __str__(self):
        return "TimeAxis: start=%g, stop=%g, step=%g, num=%g" % (self.start, self.stop, self.step, self.num)

    def _rebuild(self):
        return TimeAxis(start=self.start, stop=self.stop, num=self.num)

    @property
    def time_values(self):
        if self._values is None:
            self._values = np.linspace(self.start, self.stop, self.num)
        return self._values


class _Coordinates(object):
    """`.data` holder so that `rec.coordinates.data` reads as in the reference."""

    def __init__(self, array):
        self.data = array

    @property
    def shape(self):
        return self.data.shape


class PointSource(object):
    """
    A set of `npoint` sparse points with one time series each: `.data` is (nt, npoint) in the grid
    dtype, `.coordinates.data` is (npoint, ndim) physical positions in the grid dtype.
    """

    def __init__(self, name=None, grid=None, time_range=None, npoint=None, data=None, coordinates=None,
                 space_order=0, time_order=2, dtype=None, **kwargs):
        if time_range is None or grid is None:
            raise TypeError("Need `grid` and `time_range`")
        if coordinates is None:
            coordinates = kwargs.get('coordinates_data')
        if npoint is None:
            if coordinates is None:
                raise TypeError("Need either `npoint` or `coordinates`")
            npoint = np.asarray(coordinates).reshape(-1, grid.dim).shape[0]
        self.name = name
        self.grid = grid
        self.npoint = int(npoint)
        self.dtype = np.dtype(dtype or grid.dtype).type
        self._time_range = time_range._rebuild()
        self.nt = self._time_range.num
        self.data = np.zeros((self.nt, self.npoint), dtype=self.dtype)
        cdata = np.zeros((self.npoint, grid.dim), dtype=self.dtype)
        if coordinates is not None:
            cdata[...] = np.asarray(coordinates).reshape(self.npoint, grid.dim)
        self.coordinates = _Coordinates(cdata)
        if data is not None:
            self.data[:] = data

    @property
    def shape(self):
        return self.data.shape

    @property
    def time_values(self):
        return self._time_range.time_values

    @property
    def time_range(self):
        return self._time_range

    def resample(self, dt=None, num=None, rtol=1e-5, order=3):
        """Cubic-spline resampling to a new step or sample count; returns self when the step is
        unchanged (source.py:140-170)."""
        if (dt is None) == (num is None):
            raise AssertionError("exactly one of dt and num must be given")
        start, stop = self._time_range.start, self._time_range.stop
        dt0 = self._time_range.step
        if dt is None:
            new_range = TimeAxis(start=start, stop=stop, num=num)
            dt = new_range.step
        else:
            new_range = TimeAxis(start=start, stop=stop, step=dt)
        if np.isclose(dt, dt0):
            return self
        traces = spline_resample(self.data, self._time_range.time_values, new_range.time_values, order)
        return PointSource(name=self.name, grid=self.grid, data=traces, time_range=new_range,
                           coordinates=self.coordinates.data)


Receiver = PointSource
Shot = PointSource


_SPLINE_OPS = {}
_SPLINE_OPS_MAX = 8


def spline_resample(x, t0, t, order=3):
    """Interpolating spline of degree `order` through the columns of x(t0), evaluated at t: what
    `splev(t, splrep(t0, x[:, i], k=order))` gives for every column i (source.py:160-166 of the reference),
    to ~1e-15.  Interpolation is linear in the data, so the map t0 -> t is formed once per pair of time axes
    as a sparse matrix (a sample's influence decays by 0.27 per knot for cubics; entries below 1e-17 of the
    largest are dropped) and every later call is one sparse product instead of a spline fit per trace."""
    from scipy import sparse
    t0 = np.ascontiguousarray(t0, dtype=np.float64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    key = (t0.size, t.size, int(order), t0.tobytes()[:64], t0[-1], t.tobytes()[:64], t[-1])
    op = _SPLINE_OPS.get(key)
    if op is None:
        dense = interpolate.make_interp_spline(t0, np.eye(t0.size), k=order, axis=0)(t)
        dense[np.abs(dense) < 1e-17 * np.abs(dense).max()] = 0.0
        op = sparse.csr_matrix(dense)
        while len(_SPLINE_OPS) >= _SPLINE_OPS_MAX:
            _SPLINE_OPS.pop(next(iter(_SPLINE_OPS)))
        _SPLINE_OPS[key] = op
    return op @ np.asarray(x, dtype=np.float64)


class WaveletSource(PointSource):
    """Point sources carrying an analytic wavelet of peak frequency f0 (kHz), source.py:181-245."""

    def __init__(self, *args, **kwargs):
        kwargs.setdefault('npoint', 1 if kwargs.get('coordinates') is None else None)
        self.f0 = kwargs.get('f0')
        self.a = kwargs.get('a')
        self.t0 = kwargs.get('t0')
        for k in ('f0', 'a', 't0'):
            kwargs.pop(k, None)
        super(WaveletSource, self).__init__(*args, **kwargs)
        w = self.wavelet
        for p in range(self.npoint):
            self.data[:, p] = w

    @property
    def wavelet(self):
        raise NotImplementedError('Wavelet not defined')


class RickerSource(WaveletSource):
    """(1 - 2 r^2) exp(-r^2), r = pi f0 (t - t0), t0 defaulting to 1/f0 (source.py:272-277)."""

    @property
    def wavelet(self):
        t0 = self.t0 or 1 / self.f0
        a = self.a or 1
        r = np.pi * self.f0 * (self.time_values - t0)
        return a * (1 - 2. * r ** 2) * np.exp(-r ** 2)


class GaborSource(WaveletSource):
    """exp(-2 s^2) cos(2 pi s), s = (t - tcut) f0/2 (source.py:304-310)."""

    @property
    def wavelet(self):
        agauss = 0.5 * self.f0
        tcut = self.t0 or 1.5 / agauss
        s = (self.time_values - tcut) * agauss
        a = self.a or 1
        return a * np.exp(-2 * s ** 2) * np.cos(2 * np.pi * s)


class DGaussSource(WaveletSource):
    """First derivative of a Gaussian (source.py:346-351)."""

    @property
    def wavelet(self):
        t0 = self.t0 or 1 / self.f0
        a = self.a or 1
        time = self.time_values - t0
        return -2 * a * time * np.exp(-a * time ** 2)
