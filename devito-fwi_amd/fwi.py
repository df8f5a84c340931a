"""
FWI objective glue: forward modelling of all shots, per-shot objective and gradient, the
illumination preconditioner and `fwi_loss(x, ...)` for the optimisers -- the functions of the
reference's fwi.py (fm_single :59-65, fm_multi :67-81, fix_source_illumination :104-129,
fwi_obj_single :131-173, fwi_obj_multi :175-205, fwi_loss :236-246, resample :47-57) with the
same signatures and return values.

Differences in HOW, not in WHAT: all shots of a call are propagated side by side in one GPU plan
(and, under torch.distributed, sharded over the ranks with one all-reduce of gradient,
illumination and objective); for the least-squares misfit the residual never leaves the device.
"""
from copy import deepcopy

import numpy as np
from scipy import interpolate

from seismic import Receiver, AcquisitionGeometry, Function
from seismic.acoustic import AcousticWaveSolver
from misfit import least_square

__all__ = ['fm_single', 'fm_multi', 'fwi_obj_single', 'fwi_obj_multi', 'fwi_loss',
           'fix_source_illumination', 'resample', 'Filter', 'seismic_filter', 'shot_partition']


# ---------------------------------------------------------------------------------------------
# small host utilities
# ---------------------------------------------------------------------------------------------
def seismic_filter(data, filter_type, freqmin=None, freqmax=None, df=None, corners=16, zerophase=False, axis=-1):
    """Butterworth band/low/high-pass of a trace (reference fwi.py:10-29 over seismic/filter)."""
    from scipy.signal import iirfilter, zpk2sos, sosfilt
    kind = filter_type.lower()
    assert kind in ('bandpass', 'lowpass', 'highpass')
    fe = 0.5 * df
    if kind == 'bandpass':
        if not (freqmin and freqmax and df):
            raise ValueError
        wn, btype = [freqmin / fe, freqmax / fe], 'band'
    elif kind == 'lowpass':
        if not (freqmax and df):
            raise ValueError
        wn, btype = freqmax / fe, 'lowpass'
    else:
        if not (freqmin and df):
            raise ValueError
        wn, btype = freqmin / fe, 'highpass'
    z, p, k = iirfilter(corners, wn, btype=btype, ftype='butter', output='zpk')
    sos = zpk2sos(z, p, k)
    if zerophase:
        first = sosfilt(sos, data, axis=axis)
        return np.flip(sosfilt(sos, np.flip(first, axis=axis), axis=axis), axis=axis)
    return sosfilt(sos, data, axis=axis)


class Filter(object):
    """Callable wavelet filter handed to AcquisitionGeometry(filter=...) (reference fwi.py:31-44)."""

    def __init__(self, filter_type, freqmin=None, freqmax=None, df=None, corners=10, zerophase=False, axis=-1):
        self.filter_type, self.freqmin, self.freqmax = filter_type, freqmin, freqmax
        self.df, self.corners, self.zerophase, self.axis = df, corners, zerophase, axis

    def __call__(self, data):
        return seismic_filter(data, self.filter_type, self.freqmin, self.freqmax, self.df, self.corners,
                              self.zerophase, self.axis)


def resample(x, t, t0, order=3):
    """Spline-resample traces x(t0) onto t; identity when the steps agree (reference fwi.py:47-57)."""
    dt = t[1] - t[0]
    dt0 = t0[1] - t0[0]
    if np.isclose(dt, dt0):
        return x
    new_x = np.zeros((t.size, x.shape[1]), dtype=np.float32)
    for i in range(x.shape[1]):
        tck = interpolate.splrep(t0, x[:, i], k=order)
        new_x[:, i] = interpolate.splev(t, tck)
    return new_x


_MUTE_CACHE = {}


def _mute_factor(shape, spacing, src_pos, rec_positions):
    """prod over source and receivers of (1 - exp(-r^2 / (2 sigma^2))), sigma = dx + dz, built in
    the order and with the coordinate grids of reference fwi.py:113-127 (note its
    meshgrid(z, x): `xx` carries z values and `zz` x values, which is kept)."""
    key = (tuple(shape), tuple(float(s) for s in spacing), np.asarray(src_pos, np.float64).tobytes(),
           np.asarray(rec_positions, np.float64).tobytes())
    m = _MUTE_CACHE.get(key)
    if m is None:
        dx, dz = spacing
        nx, nz = shape
        x = np.arange(0, nx) * dx
        z = np.arange(0, nz) * dz
        xx, zz = np.meshgrid(z, x)
        sigma = dx + dz
        sx, sz = src_pos[0], src_pos[1]
        m = 1. - np.exp(-.5 * ((xx - sx) ** 2 + (zz - sz) ** 2) / (sigma ** 2))
        for i in range(rec_positions.shape[0]):
            rx, rz = rec_positions[i, 0], rec_positions[i, 1]
            m = m * (1. - np.exp(-.5 * ((xx - rx) ** 2 + (zz - rz) ** 2) / (sigma ** 2)))
        if len(_MUTE_CACHE) > 256:
            _MUTE_CACHE.clear()
        _MUTE_CACHE[key] = m
    return m


def fix_source_illumination(geometry, g):
    """Mute `g` around the source and every receiver (reference fwi.py:104-129)."""
    if geometry.src_positions.shape[0] > 1:
        raise ValueError("Only single source valid.")
    nx, nz = geometry.model.shape
    if g.shape != (nx, nz):
        raise ValueError("Shape does not match!")
    return g * _mute_factor((nx, nz), geometry.model.spacing, geometry.src_positions[0], geometry.rec_positions)


def _shot_geometry(geometry, i):
    return AcquisitionGeometry(geometry.model, geometry.rec_positions, geometry.src_positions[i, :],
                               geometry.t0, geometry.tn, f0=geometry.f0, src_type=geometry.src_type,
                               filter=geometry._filter)


# ---------------------------------------------------------------------------------------------
# distributed helpers: shots are independent, so ranks own disjoint subsets
# ---------------------------------------------------------------------------------------------
def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except ImportError:
        pass
    return None


def shot_partition(nshots, rank=None, world=None):
    """Shot indices owned by `rank`: contiguous blocks, sizes differing by at most one."""
    if rank is None or world is None:
        dist = _dist()
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
    base, extra = divmod(nshots, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def _all_reduce_sum(fval, arrays):
    """Sum a scalar and a list of float64 numpy arrays over the ranks (RCCL when the arrays'
    home is a GPU process group, gloo on CPU)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return fval, arrays
    import torch
    backend = dist.get_backend()
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    flat = np.concatenate([np.array([fval], dtype=np.float64)] + [a.ravel() for a in arrays])
    t = torch.from_numpy(flat).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    flat = t.cpu().numpy()
    out, pos = [], 1
    for a in arrays:
        out.append(flat[pos:pos + a.size].reshape(a.shape))
        pos += a.size
    return float(flat[0]), out


# ---------------------------------------------------------------------------------------------
# batched engine access
# ---------------------------------------------------------------------------------------------
def _batch_entry(geometry, shots):
    """GPU plan holding `len(shots)` shots of this acquisition, loaded with the current model."""
    from mifwi import engine
    model = geometry.model
    model._initialize_bcs(bcs="damp")
    nt = geometry.nt
    entry = engine.get_plan(model.grid.shape, model.spacing, model.grid.origin, model.space_order, nt,
                            model.critical_dt, 1, geometry.nrec, max_shots=len(shots), dtype=model.dtype)
    vp = model.vp.data
    if np.ndim(vp) == 0:
        vp = np.full(model.grid.shape, vp, dtype=model.dtype)
    damp = model.damp.data if hasattr(model.damp, 'data') else np.full(model.grid.shape, model.damp, model.dtype)
    entry.plan.set_model(np.ascontiguousarray(vp, dtype=model.dtype), np.ascontiguousarray(damp, dtype=model.dtype))
    gkey = (geometry.src_positions[shots].tobytes(), geometry.rec_positions.tobytes())
    if entry.geom_key != gkey:
        entry.plan.set_geometry(geometry.src_positions[shots].reshape(len(shots), 1, model.dim), geometry.rec_positions)
        entry.geom_key = gkey
    for which in (0, 1, 2):
        entry.claim(which)
    return entry


def _wavelet(geometry):
    """(nt, 1) source time function of one shot (all shots share type, f0 and filter)."""
    return _shot_geometry(geometry, 0).src.data


def _compute_batch(geometry, shots, obs, direct_wave, misfit_func, calc_grad):
    """Objective pieces of the shots `shots` (indices into the survey) as one GPU batch:
    (fvals per shot, residuals (ns, nt, nrec), padded-grid gradients (ns, *grid) or None,
    illumination likewise).  With the least-squares misfit and no direct wave the residual is
    formed on the device; otherwise the misfit runs on the host between the two passes."""
    model = geometry.model
    entry = _batch_entry(geometry, shots)
    plan = entry.plan
    ns = len(shots)
    wavelet = _wavelet(geometry)
    obs_b = np.stack([obs[i].data for i in shots]).astype(model.dtype, copy=False)
    dw_b = None
    if direct_wave is not None:
        dw_b = np.stack([direct_wave[i].data for i in shots]).astype(model.dtype, copy=False)
    if misfit_func is least_square and dw_b is None:
        out = plan.l2_misfit_gradient(wavelet, obs_b, nshots=ns, calc_grad=calc_grad, want_illum=True,
                                      want_residual=True)
        return out['fval'], out['residual'], out['grad'], out['illum']
    syn_b = plan.forward(wavelet, nshots=ns, save=calc_grad)
    res_b = np.empty_like(syn_b)
    fvals = []
    for k in range(ns):
        s_, o_ = syn_b[k], obs_b[k]
        if dw_b is not None:
            s_, o_ = s_ - dw_b[k], o_ - dw_b[k]
        f_, r_ = misfit_func(s_, o_)
        fvals.append(f_)
        res_b[k] = r_
    g_b = il_b = None
    if calc_grad:
        g_b, il_b = plan.gradient(res_b, nshots=ns, want_illum=True)
    return fvals, res_b, g_b, il_b


# ---------------------------------------------------------------------------------------------
# modelling
# ---------------------------------------------------------------------------------------------
def fm_single(geometry, save=False):
    """Forward modelling of one shot: (Receiver, wavefield) (reference fwi.py:59-65)."""
    solver = AcousticWaveSolver(geometry.model, geometry, space_order=geometry.model.space_order)
    data, u = solver.forward(vp=geometry.model.vp, save=save)[0:2]
    return data, u


def fm_multi(geometry, save=False):
    """Forward modelling of every shot of `geometry`: list of Receiver objects, one per shot
    (reference fwi.py:67-81).  Under torch.distributed only this rank's shots are modelled; the
    other entries are None."""
    nsrc = geometry.nsrc
    mine = shot_partition(nsrc)
    shots = [None] * nsrc
    if not mine:
        return shots
    entry = _batch_entry(geometry, mine)
    data = entry.plan.forward(_wavelet(geometry), nshots=len(mine), save=False)
    for k, i in enumerate(mine):
        rec = _shot_geometry(geometry, i).rec
        rec.data[:] = data[k]
        shots[i] = rec
    return shots


def fwi_obj_single(geometry, obs, misfit_func, direct_wave=None, resample_dt=None, calc_grad=False):
    """Objective, muted gradient, residual and muted illumination of one shot
    (reference fwi.py:131-173)."""
    solver = AcousticWaveSolver(geometry.model, geometry, space_order=geometry.model.space_order)
    pred, wfd = solver.forward(vp=geometry.model.vp, save=calc_grad)[0:2]
    dw = direct_wave
    if resample_dt is None:
        resample_dt = geometry.dt
    else:
        obs = deepcopy(obs).resample(resample_dt)
        pred = pred.resample(resample_dt)
        if direct_wave is not None:
            dw = deepcopy(direct_wave).resample(resample_dt)
    syn_data = pred.data
    obs_data = obs.data
    if direct_wave is not None:
        syn_data = syn_data - dw.data
        obs_data = obs_data - dw.data
    fval, residual_data = misfit_func(syn_data, obs_data)
    residual = Receiver(name="rec", grid=geometry.model.grid, time_range=geometry.time_axis,
                        coordinates=geometry.rec_positions)
    residual.data[:] = resample(residual_data, geometry.time_axis.time_values, pred.time_values)[:]
    illum, crop_grad = None, None
    if calc_grad:
        grad = Function(name="grad", grid=geometry.model.grid)
        solver.gradient(rec=residual, u=wfd, vp=geometry.model.vp, grad=grad)
        nbl = geometry.model.nbl
        crop_grad = np.array(grad.data[:])[nbl:-nbl, nbl:-nbl]
        crop_grad = fix_source_illumination(geometry, crop_grad)
        illum = (wfd.data * wfd.data).sum(axis=0)[nbl:-nbl, nbl:-nbl]
        illum = fix_source_illumination(geometry, illum)
    return fval, crop_grad, residual.data, illum


def _same_sampling(geometry, objs):
    dt = geometry.model.critical_dt
    if not np.isclose(geometry.dt, dt):
        return False
    return all(o is None or np.isclose(o.time_range.step, dt) for o in objs)


def fwi_obj_multi(geometry, obs, misfit_func, direct_wave=None, mask=None, precond=True, calc_grad=False):
    """Sum over shots of objective / gradient / illumination, then preconditioning and masking:
    returns (fval, grad.ravel() as float64, residuals) (reference fwi.py:175-205).

    All shots owned by this rank run as one GPU batch.  With the least-squares misfit and data on
    the solver's time grid the residual is formed on the device; with a direct wave to subtract
    (fwi.py:148-150) or any other misfit function, the misfit is evaluated on the host between the
    batched forward and backward passes, in the reference's own float32 arithmetic."""
    model = geometry.model
    nbl = model.nbl
    nsrc = geometry.nsrc
    mine = shot_partition(nsrc)
    fval = .0
    grad = np.zeros(model.shape)
    illum = np.zeros(model.shape)
    residuals = [None] * nsrc
    if mine and not _same_sampling(geometry, [obs[i] for i in mine] +
                                   ([direct_wave[i] for i in mine] if direct_wave is not None else [])):
        # resampled data: shot-by-shot, as the reference does it
        for i in mine:
            f_, g_, r_, il_ = fwi_obj_single(_shot_geometry(geometry, i), obs[i], misfit_func,
                                             direct_wave[i] if direct_wave is not None else None, geometry.dt,
                                             calc_grad)
            fval += f_
            residuals[i] = r_
            if calc_grad:
                grad += g_
                illum += il_
    elif mine:
        fvals, res_b, g_b, il_b = _compute_batch(geometry, mine, obs, direct_wave, misfit_func, calc_grad)
        crop = tuple(slice(nbl, -nbl) if nbl else slice(None) for _ in range(model.dim))
        for k, i in enumerate(mine):
            fval += float(fvals[k])
            residuals[i] = res_b[k]
            if calc_grad:
                mute = _mute_factor(model.shape, model.spacing, geometry.src_positions[i], geometry.rec_positions)
                grad += g_b[k][crop] * mute
                illum += il_b[k][crop] * mute
    if calc_grad:
        fval, (grad, illum) = _all_reduce_sum(fval, [grad, illum])
        if precond:
            grad /= np.sqrt(illum + 1e-30)
        if mask is not None:
            grad *= mask
    else:
        fval, _ = _all_reduce_sum(fval, [])
    return fval, grad.reshape(-1).astype(np.float64), residuals


def fwi_loss(x, geometry, obs, misfit_func, direct_wave=None, mask=None, precond=True, calc_grad=True):
    """Objective and gradient with respect to squared slowness x = 1/vp^2 (reference fwi.py:236-246)."""
    v = 1. / np.sqrt(x.reshape(geometry.model.shape))
    geometry.model.update('vp', v.reshape(geometry.model.shape))
    return fwi_obj_multi(geometry, obs, misfit_func, direct_wave, mask, precond, calc_grad)
