"""
`AcousticWaveSolver` with the call signatures of the reference's
seismic/acoustic/wavesolver.py:10-246, executing on an MI355X through libmifwi instead of
through Devito-generated C.  forward / adjoint / jacobian (alias born) / jacobian_adjoint (alias
gradient) are implemented for the OT2 and OT4 kernels, both also with a free surface.
`jacobian_adjoint(checkpointing=True)` models the forward wavefield again from `geometry.src`, as the
reference's pyrevolve branch does, with the library's own device checkpointing.
"""
import numpy as np

from ..fields import Function, TimeFunction, DeviceTimeFunction, Constant
from ..source import PointSource

__all__ = ['AcousticWaveSolver', 'PerformanceSummary']


class PerformanceSummary(dict):
    """Timing record of one operator run (stand-in for Devito's PerformanceSummary)."""

    def __init__(self, stats):
        super(PerformanceSummary, self).__init__(stats)
        self.timings = {'section0': stats.get('kernel_ms', 0.0) * 1e-3}
        self.gpointss = stats.get('gpointss', 0.0)
        self.gflopss = None
        self.oi = None


def _as_vp_array(vp, model):
    """Velocity argument -> padded-grid array: Function, Constant, scalar or ndarray."""
    if vp is None:
        vp = model.vp
    data = vp.data if hasattr(vp, 'data') else vp
    if np.ndim(data) == 0:
        return np.full(model.grid.shape, data, dtype=model.dtype)
    data = np.asarray(data)
    if data.shape == model.grid.shape:
        return data
    if data.shape == model.shape:
        return np.pad(data, model.padsizes, mode='edge') if model.nbl else data
    raise ValueError("vp of shape %s does not match the model" % (data.shape,))


class AcousticWaveSolver(object):
    """
    Parameters as in the reference (wavesolver.py:16-27): model, geometry, kernel ('OT2'),
    space_order (default 4).  Extra keyword `device` selects the GPU ordinal.
    """

    def __init__(self, model, geometry, kernel='OT2', space_order=4, **kwargs):
        self.model = model
        # the reference forces the sponge here (wavesolver.py:30); a model built with bcs="pml" keeps its choice
        self._pml = getattr(model, '_bcs', None) == "pml"
        if not self._pml:
            self.model._initialize_bcs(bcs="damp")
        elif model.dim != 2 or kernel != 'OT2' or getattr(model, 'fs', False) or model.nbl < 1:
            raise NotImplementedError("bcs='pml' is available for 2-D OT2 modelling without a free surface")
        self.geometry = geometry
        assert self.model.grid == geometry.grid
        if kernel not in ('OT2', 'OT4'):
            raise ValueError("Unrecognized kernel")
        self.space_order = space_order
        self.kernel = kernel
        self._device = kwargs.pop('device', -1)
        self._kwargs = kwargs

    @property
    def dt(self):
        # wavesolver.py:41-46: the 4th-order kernel steps sqrt(3) ~ 1.73 times further
        if self.kernel == 'OT4':
            return self.model.dtype(1.73 * self.model.critical_dt)
        return self.model.critical_dt

    # -- plumbing -----------------------------------------------------------------------------
    def _entry(self, dt, nsrc, nrec, nt, extra_flags=0):
        from mifwi import engine
        m = self.model
        from mifwi import FLAG_FREE_SURFACE, FLAG_OT4, FLAG_PML
        flags = (FLAG_FREE_SURFACE if getattr(m, 'fs', False) else 0) | (FLAG_OT4 if self.kernel == 'OT4' else 0)
        flags |= (FLAG_PML if self._pml else 0) | extra_flags
        return engine.get_plan(m.grid.shape, m.spacing, m.grid.origin, self.space_order, nt, dt, nsrc, nrec,
                               max_shots=1, dtype=m.dtype, device=self._device, flags=flags)

    def _damp_array(self):
        d = self.model.damp
        return d.data if hasattr(d, 'data') else np.full(self.model.grid.shape, d, dtype=self.model.dtype)

    def _load(self, entry, vp, src_coords, rec_coords):
        vp_arr = np.ascontiguousarray(_as_vp_array(vp, self.model), dtype=self.model.dtype)
        entry.plan.set_model(vp_arr, np.ascontiguousarray(self._damp_array(), dtype=self.model.dtype))
        if self._pml:
            entry.plan.set_pml(self.model.nbl)
        entry.plan.set_geometry(src_coords, rec_coords)

    def _no_pml(self):
        if self._pml:
            raise NotImplementedError("only forward modelling is available with bcs='pml'")

    # -- operators ------------------------------------------------------------------------------
    def forward(self, src=None, rec=None, u=None, vp=None, save=None, **kwargs):
        """Forward modelling; returns (rec, u, summary) (wavesolver.py:76-114)."""
        src = src or self.geometry.src
        rec = rec or self.geometry.rec
        dt = kwargs.pop('dt', self.dt)
        nt = src.data.shape[0]
        if self._pml and save:
            raise NotImplementedError("bcs='pml' models without saving the wavefield")
        entry = self._entry(dt, src.data.shape[1], rec.data.shape[1], nt)
        self._load(entry, vp, src.coordinates.data, rec.coordinates.data)
        which = 0 if save else 1
        out_u = u
        if u is None:
            out_u = DeviceTimeFunction('u', self.model.grid, nt if save else None, entry, None, which,
                                       time_order=2, space_order=self.space_order)
        token = entry.claim(which, out_u if u is None else None)
        if save:
            # saving runs may step on the rotating levels (storage 1) and stream copies into storage 0:
            # a lazy wavefield of an earlier non-saving run must be fetched before that happens
            entry.claim(1)
        data = entry.plan.forward(src.data, nshots=1, save=bool(save))
        rec.data[:] = data[0]
        if u is None:
            out_u._token = token
        else:
            u.data[...] = entry.plan.read_wavefield(which, shot=0)
        return rec, out_u, PerformanceSummary(entry.plan.stats())

    def adjoint(self, rec, srca=None, v=None, vp=None, **kwargs):
        """Adjoint modelling; returns (srca, v, summary) (wavesolver.py:116-151)."""
        self._no_pml()
        srca = srca or self.geometry.new_src(name='srca', src_type=None)
        dt = kwargs.pop('dt', self.dt)
        nt = rec.data.shape[0]
        entry = self._entry(dt, srca.data.shape[1], rec.data.shape[1], nt)
        self._load(entry, vp, srca.coordinates.data, rec.coordinates.data)
        out_v = v
        if v is None:
            out_v = DeviceTimeFunction('v', self.model.grid, None, entry, None, 2, time_order=2,
                                       space_order=self.space_order)
        token = entry.claim(2, out_v if v is None else None)
        srca.data[:] = entry.plan.adjoint(rec.data[None], nshots=1)[0]
        if v is None:
            out_v._token = token
        else:
            v.data[...] = entry.plan.read_wavefield(2, shot=0)
        return srca, out_v, PerformanceSummary(entry.plan.stats())

    def jacobian_adjoint(self, rec, u, v=None, grad=None, vp=None, checkpointing=False, illum=None, **kwargs):
        """Adjoint Jacobian (FWI gradient); accumulates into `grad` and returns (grad, summary)
        (wavesolver.py:153-205).  Extension: a Function passed as `illum` also receives (is incremented
        by) the sum over time of u^2, which the gradient kernel forms from the saved wavefield anyway."""
        self._no_pml()
        dt = kwargs.pop('dt', self.dt)
        grad = grad or Function(name='grad', grid=self.model.grid)
        nt = rec.data.shape[0]
        if checkpointing:
            # wavesolver.py:188-201: `u` is not used; the forward wavefield is modelled again from geometry.src and
            # kept as checkpoints from which the backward pass propagates one segment at a time again (here the
            # library's device checkpointing, MIFWI_FLAG_CHECKPOINT, in place of pyrevolve)
            from mifwi import FLAG_CHECKPOINT
            src = self.geometry.src
            entry = self._entry(dt, src.data.shape[1], rec.data.shape[1], nt, extra_flags=FLAG_CHECKPOINT)
            self._load(entry, vp, src.coordinates.data, rec.coordinates.data)
            entry.claim(0), entry.claim(1)
            entry.plan.forward(src.data, nshots=1, save=True, want_rec=False)
        else:
            entry = self._entry(dt, self.geometry.nsrc, rec.data.shape[1], nt)
            self._load(entry, vp, self.geometry.src_positions, rec.coordinates.data)
        if checkpointing:
            pass
        elif not (isinstance(u, DeviceTimeFunction) and u._entry is entry and u.on_device()):
            ud = np.ascontiguousarray(u.data, dtype=self.model.dtype)
            assert ud.shape[0] == nt, "u must hold every time level (created with save=True)"
            entry.claim(0)
            entry.plan.write_wavefield(ud, shot=0)
        entry.claim(2)
        out = entry.plan.gradient(rec.data[None], nshots=1, grad=grad.data.reshape((1,) + grad.data.shape),
                                  accumulate=True, want_illum=illum is not None)
        if illum is not None:
            illum.data[...] += out[1][0]
        if v is not None:
            v.data[...] = entry.plan.read_wavefield(2, shot=0)
        return grad, PerformanceSummary(entry.plan.stats())

    def jacobian(self, dmin, src=None, rec=None, u=None, U=None, vp=None, **kwargs):
        """Linearised Born modelling for the perturbation `dmin` of 1/vp^2 (array over the padded or
        the physical grid, or a Function); returns (rec, u, U, summary) (wavesolver.py:207-242)."""
        self._no_pml()
        src = src or self.geometry.src
        rec = rec or self.geometry.rec
        dt = kwargs.pop('dt', self.dt)
        nt = src.data.shape[0]
        dm = np.asarray(dmin.data if hasattr(dmin, 'data') else dmin, dtype=self.model.dtype)
        if dm.shape == self.model.shape and dm.shape != self.model.grid.shape:
            dm = np.pad(dm, self.model.padsizes, mode='constant')
        if dm.shape != self.model.grid.shape:
            raise ValueError("dmin of shape %s does not match the model" % (dm.shape,))
        entry = self._entry(dt, src.data.shape[1], rec.data.shape[1], nt)
        self._load(entry, vp, src.coordinates.data, rec.coordinates.data)
        out_u, out_U = u, U
        if u is None:
            out_u = DeviceTimeFunction('u', self.model.grid, None, entry, None, 1, time_order=2,
                                       space_order=self.space_order)
        if U is None:
            out_U = DeviceTimeFunction('U', self.model.grid, None, entry, None, 2, time_order=2,
                                       space_order=self.space_order)
        tok_u = entry.claim(1, out_u if u is None else None)
        tok_U = entry.claim(2, out_U if U is None else None)
        rec.data[:] = entry.plan.born(src.data, np.ascontiguousarray(dm), nshots=1)[0]
        if u is None:
            out_u._token = tok_u
        else:
            u.data[...] = entry.plan.read_wavefield(1, shot=0)
        if U is None:
            out_U._token = tok_U
        else:
            U.data[...] = entry.plan.read_wavefield(2, shot=0)
        return rec, out_u, out_U, PerformanceSummary(entry.plan.stats())

    # Backward compatibility
    born = jacobian
    gradient = jacobian_adjoint
