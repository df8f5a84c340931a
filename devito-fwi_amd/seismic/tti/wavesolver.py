"""
`AnisotropicWaveSolver` with the call signatures of the reference's
seismic/tti/wavesolver.py:11-357 for the 'centered' kernel, executing on an MI355X through
libmifwi.  forward and adjoint are implemented; the staggered kernel, jacobian and
jacobian_adjoint are not and raise.
"""
import numpy as np

from ..fields import DeviceTimeFunction
from ..acoustic.wavesolver import PerformanceSummary, _as_vp_array

__all__ = ['AnisotropicWaveSolver']


class AnisotropicWaveSolver(object):
    """
    Parameters as in the reference (tti/wavesolver.py:17-30): model (with epsilon, delta, theta and,
    in 3-D, phi), geometry, space_order (even; must be a multiple of 4 here, the reference
    recommends it).
    """

    def __init__(self, model, geometry, space_order=4, **kwargs):
        self.model = model
        self.model._initialize_bcs(bcs="damp")
        self.geometry = geometry
        if space_order % 2 != 0:
            raise ValueError("space_order must be even but got %s" % space_order)
        if space_order % 4 != 0:
            raise NotImplementedError("the TTI kernels need space_order to be a multiple of 4, got %s" % space_order)
        self.space_order = space_order
        self._device = kwargs.pop('device', -1)
        self._kwargs = kwargs

    @property
    def dt(self):
        return self.model.critical_dt

    def _param(self, value, name):
        """Operator argument or model attribute -> padded-grid array (None when the model has none)."""
        if value is None:
            value = getattr(self.model, name, None)
        if value is None:
            return None
        data = value.data if hasattr(value, 'data') else value
        if np.ndim(data) == 0:
            return np.full(self.model.grid.shape, data, dtype=self.model.dtype)
        data = np.asarray(data, dtype=self.model.dtype)
        if data.shape == self.model.shape and data.shape != self.model.grid.shape:
            data = np.pad(data, self.model.padsizes, mode='edge')
        return np.ascontiguousarray(data)

    def _entry(self, dt, nsrc, nrec, nt, vp, epsilon, delta, theta, phi, src_coords, rec_coords):
        from mifwi import engine
        m = self.model
        entry = engine.get_plan(m.grid.shape, m.spacing, m.grid.origin, self.space_order, nt, dt, nsrc, nrec,
                                max_shots=1, dtype=m.dtype, device=self._device)
        damp = m.damp.data if hasattr(m.damp, 'data') else np.full(m.grid.shape, m.damp, dtype=m.dtype)
        entry.plan.set_model(np.ascontiguousarray(_as_vp_array(vp, m), dtype=m.dtype),
                             np.ascontiguousarray(damp, dtype=m.dtype))
        zeros = np.zeros(m.grid.shape, dtype=m.dtype)
        eps, dl, th = self._param(epsilon, 'epsilon'), self._param(delta, 'delta'), self._param(theta, 'theta')
        entry.plan.tti_set_model(zeros if eps is None else eps, zeros if dl is None else dl,
                                 zeros if th is None else th, self._param(phi, 'phi') if m.dim == 3 else None)
        entry.plan.set_geometry(src_coords, rec_coords)
        return entry

    def forward(self, src=None, rec=None, u=None, v=None, vp=None, epsilon=None, delta=None, theta=None, phi=None,
                save=False, kernel='centered', **kwargs):
        """Forward modelling; returns (rec, u, v, summary) (tti/wavesolver.py:80-167)."""
        if kernel != 'centered':
            raise NotImplementedError("only the 'centered' TTI kernel is implemented")
        if save:
            raise NotImplementedError("saving the TTI wavefields is not implemented")
        src = src or self.geometry.src
        rec = rec or self.geometry.rec
        dt = kwargs.pop('dt', self.dt)
        nt = src.data.shape[0]
        entry = self._entry(dt, src.data.shape[1], rec.data.shape[1], nt, vp, epsilon, delta, theta, phi,
                            src.coordinates.data, rec.coordinates.data)
        rec.data[:] = entry.plan.tti_forward(src.data, nshots=1)[0]
        fields = []
        for given, which, name in ((u, 3, 'u'), (v, 4, 'v')):
            data = entry.plan.read_wavefield(which, shot=0)
            if given is None:
                from ..fields import TimeFunction
                given = TimeFunction(name=name, grid=self.model.grid, time_order=2, space_order=self.space_order)
            given.data[...] = data
            fields.append(given)
        return rec, fields[0], fields[1], PerformanceSummary(entry.plan.stats())

    def adjoint(self, rec, srca=None, p=None, r=None, vp=None, epsilon=None, delta=None, theta=None, phi=None,
                **kwargs):
        """Adjoint modelling; returns (srca, p, r, summary) (tti/wavesolver.py:169-215)."""
        srca = srca or self.geometry.new_src(name='srca', src_type=None)
        dt = kwargs.pop('dt', self.dt)
        nt = rec.data.shape[0]
        entry = self._entry(dt, srca.data.shape[1], rec.data.shape[1], nt, vp, epsilon, delta, theta, phi,
                            srca.coordinates.data, rec.coordinates.data)
        srca.data[:] = entry.plan.tti_adjoint(rec.data[None], nshots=1)[0]
        fields = []
        for given, which, name in ((p, 5, 'p'), (r, 6, 'r')):
            data = entry.plan.read_wavefield(which, shot=0)
            if given is None:
                from ..fields import TimeFunction
                given = TimeFunction(name=name, grid=self.model.grid, time_order=2, space_order=self.space_order)
            given.data[...] = data
            fields.append(given)
        return srca, fields[0], fields[1], PerformanceSummary(entry.plan.stats())

    def jacobian(self, *args, **kwargs):
        raise NotImplementedError("TTI Born modelling (tti/wavesolver.py:217-281 of the reference) is not implemented")

    def jacobian_adjoint(self, *args, **kwargs):
        raise NotImplementedError("the TTI gradient (tti/wavesolver.py:283-353 of the reference) is not implemented")

    born = jacobian
    gradient = jacobian_adjoint
