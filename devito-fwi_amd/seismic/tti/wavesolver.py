"""
`AnisotropicWaveSolver` with the call signatures of the reference's
seismic/tti/wavesolver.py:11-357 for the 'centered' kernel, executing on an MI355X through
libmifwi: forward, adjoint, jacobian (Born) and jacobian_adjoint (gradient).  The staggered
kernel is not implemented and raises.
"""
import numpy as np

from ..fields import DeviceTimeFunction, Function
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

    def _fields(self, entry, given, whichs, names, save=None):
        """Lazy device-resident TimeFunctions for the storage ids `whichs` (caller-supplied ones are
        filled after the run instead).  Returns (outputs, finish) where finish() is called after the run."""
        outs, tokens = [], []
        for g, which, name in zip(given, whichs, names):
            out = g
            if g is None:
                out = DeviceTimeFunction(name, self.model.grid, save, entry, None, which, time_order=2,
                                         space_order=self.space_order)
            tokens.append(entry.claim(which, out if g is None else None))
            outs.append(out)

        def finish():
            for g, out, which, token in zip(given, outs, whichs, tokens):
                if g is None:
                    out._token = token
                else:
                    g.data[...] = entry.plan.read_wavefield(which, shot=0)
        return outs, finish

    def forward(self, src=None, rec=None, u=None, v=None, vp=None, epsilon=None, delta=None, theta=None, phi=None,
                save=False, kernel='centered', **kwargs):
        """Forward modelling; returns (rec, u, v, summary) (tti/wavesolver.py:80-167).  With save the
        fields keep every time level; they stay on the device until `.data` is read."""
        if kernel != 'centered':
            raise NotImplementedError("only the 'centered' TTI kernel is implemented")
        src = src or self.geometry.src
        rec = rec or self.geometry.rec
        dt = kwargs.pop('dt', self.dt)
        nt = src.data.shape[0]
        entry = self._entry(dt, src.data.shape[1], rec.data.shape[1], nt, vp, epsilon, delta, theta, phi,
                            src.coordinates.data, rec.coordinates.data)
        (out_u, out_v), finish = self._fields(entry, (u, v), (7, 8) if save else (3, 4), ('u', 'v'),
                                              save=nt if save else None)
        rec.data[:] = entry.plan.tti_forward(src.data, nshots=1, save=bool(save))[0]
        finish()
        return rec, out_u, out_v, PerformanceSummary(entry.plan.stats())

    def adjoint(self, rec, srca=None, p=None, r=None, vp=None, epsilon=None, delta=None, theta=None, phi=None,
                **kwargs):
        """Adjoint modelling; returns (srca, p, r, summary) (tti/wavesolver.py:169-215)."""
        srca = srca or self.geometry.new_src(name='srca', src_type=None)
        dt = kwargs.pop('dt', self.dt)
        nt = rec.data.shape[0]
        entry = self._entry(dt, srca.data.shape[1], rec.data.shape[1], nt, vp, epsilon, delta, theta, phi,
                            srca.coordinates.data, rec.coordinates.data)
        (out_p, out_r), finish = self._fields(entry, (p, r), (5, 6), ('p', 'r'))
        srca.data[:] = entry.plan.tti_adjoint(rec.data[None], nshots=1)[0]
        finish()
        return srca, out_p, out_r, PerformanceSummary(entry.plan.stats())

    def jacobian(self, dm, src=None, rec=None, u0=None, v0=None, du=None, dv=None, vp=None, epsilon=None,
                 delta=None, theta=None, phi=None, save=None, kernel='centered', **kwargs):
        """Linearised Born modelling for the perturbation `dm` of 1/vp^2; returns
        (rec, u0, v0, du, dv, summary) (tti/wavesolver.py:219-280)."""
        if kernel != 'centered':
            raise ValueError('Only centered kernel is supported for the jacobian')
        dt = kwargs.pop('dt', self.dt)
        src = src or self.geometry.src
        rec = rec or self.geometry.rec
        nt = src.data.shape[0]
        dmv = np.asarray(dm.data if hasattr(dm, 'data') else dm, dtype=self.model.dtype)
        if dmv.shape == self.model.shape and dmv.shape != self.model.grid.shape:
            dmv = np.pad(dmv, self.model.padsizes, mode='constant')
        if dmv.shape != self.model.grid.shape:
            raise ValueError("dm of shape %s does not match the model" % (dmv.shape,))
        entry = self._entry(dt, src.data.shape[1], rec.data.shape[1], nt, vp, epsilon, delta, theta, phi,
                            src.coordinates.data, rec.coordinates.data)
        outs, finish = self._fields(entry, (u0, v0, du, dv), (3, 4, 5, 6), ('u0', 'v0', 'du', 'dv'))
        rec.data[:] = entry.plan.tti_born(src.data, np.ascontiguousarray(dmv), nshots=1)[0]
        finish()
        return (rec,) + tuple(outs) + (PerformanceSummary(entry.plan.stats()),)

    def jacobian_adjoint(self, rec, u0, v0, du=None, dv=None, dm=None, vp=None, epsilon=None, delta=None,
                         theta=None, phi=None, checkpointing=False, kernel='centered', **kwargs):
        """Adjoint of the Jacobian (FWI gradient); accumulates into `dm` and returns (dm, summary)
        (tti/wavesolver.py:282-357).  u0 and v0 are used in place while they are the still-resident
        result of `forward(save=True)` on the same model and geometry; otherwise (host copies, or
        `checkpointing=True`, which the reference also answers by re-running the forward operator from
        `geometry.src`) the background fields are modelled again on the device."""
        if kernel != 'centered':
            raise ValueError('Only centered kernel is supported for the jacobian_adj')
        dt = kwargs.pop('dt', self.dt)
        dm = dm or Function(name='dm', grid=self.model.grid)
        nt = rec.data.shape[0]
        src = self.geometry.src
        entry = self._entry(dt, src.data.shape[1], rec.data.shape[1], nt, vp, epsilon, delta, theta, phi,
                            src.coordinates.data, rec.coordinates.data)
        resident = (not checkpointing and
                    all(isinstance(f, DeviceTimeFunction) and f._entry is entry and f.on_device() for f in (u0, v0)))
        if not resident:
            entry.claim(7), entry.claim(8)
            entry.plan.tti_forward(src.data, nshots=1, save=True)
        (out_du, out_dv), finish = self._fields(entry, (du, dv), (5, 6), ('du', 'dv'))
        entry.plan.tti_gradient(rec.data[None], nshots=1, grad=dm.data.reshape((1,) + dm.data.shape), accumulate=True)
        finish()
        return dm, PerformanceSummary(entry.plan.stats())

    born = jacobian
    gradient = jacobian_adjoint
