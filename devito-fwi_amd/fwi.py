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

from seismic import Receiver, AcquisitionGeometry, Function
from seismic.source import spline_resample
from seismic.acoustic import AcousticWaveSolver
from misfit import least_square

__all__ = ['fm_single', 'fm_multi', 'fwi_obj_single', 'fwi_obj_multi', 'fwi_loss',
           'fix_source_illumination', 'resample', 'Filter', 'seismic_filter', 'shot_partition',
           'clear_caches']


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
    return spline_resample(x, t0, t, order).astype(np.float32)


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


_MUTE_DEV = {}


def _mute_factors_device(model, geometry, shots, device):
    """(ns, nx, nz) float64 tensor of the shots' mute factors on `device`, kept for the survey."""
    import torch
    key = (tuple(model.shape), tuple(float(s) for s in model.spacing), geometry.src_positions[shots].tobytes(),
           geometry.rec_positions.tobytes(), str(device))
    t = _MUTE_DEV.get(key)
    if t is None:
        m = np.stack([_mute_factor(model.shape, model.spacing, geometry.src_positions[i], geometry.rec_positions)
                      for i in shots])
        t = torch.from_numpy(np.ascontiguousarray(m, dtype=np.float64)).to(device)
        if len(_MUTE_DEV) > 16:
            _MUTE_DEV.clear()
        _MUTE_DEV[key] = t
    return t


_MASK_DEV = {}


def _mask_device(mask, device):
    """float64 tensor of the gradient mask on `device`, kept while the same array is passed with the same content."""
    import torch
    a = np.ascontiguousarray(mask, dtype=np.float64)
    key = (id(mask), a.shape, str(device))
    finger = float(a.reshape(-1)[::97].sum())
    hit = _MASK_DEV.get(key)
    if hit is None or hit[0] != finger:
        if len(_MASK_DEV) > 8:
            _MASK_DEV.clear()
        hit = (finger, torch.as_tensor(a).to(device), mask)
        _MASK_DEV[key] = hit
    return hit[1]


def _all_reduce_sum_device(packed):
    """Sum a float64 tensor over the ranks where it lives: RCCL for a GPU tensor under the nccl backend;
    under gloo (CPU rehearsals) it travels through the host."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return packed
    if dist.get_backend() == 'nccl' or not packed.is_cuda:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        return packed
    host = packed.cpu()
    dist.all_reduce(host, op=dist.ReduceOp.SUM)
    return host.to(packed.device)


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
def _batch_entry(geometry, shots, dt=None):
    """GPU plan holding `len(shots)` shots of this acquisition, loaded with the current model.  `dt`: the model's
    critical time step when the caller has it at hand (forming it takes a pass over the velocity model)."""
    from mifwi import engine
    from seismic.source import TimeAxis
    model = geometry.model
    model._initialize_bcs(bcs="damp")
    dt = model.critical_dt if dt is None else dt
    # modelling always runs on the model's own time axis: the per-shot geometries the reference builds
    # (fwi.py:73-76,185-188) do not inherit geometry.resample(), which only affects the misfit sampling
    nt = TimeAxis(start=geometry.t0, stop=geometry.tn, step=dt).num
    from mifwi import FLAG_FREE_SURFACE, FLAG_SAVE_F16
    flags = (FLAG_FREE_SURFACE if getattr(model, 'fs', False) else 0) | (FLAG_SAVE_F16 if SAVE_HALF else 0)
    entry = engine.get_plan(model.grid.shape, model.spacing, model.grid.origin, model.space_order, nt,
                            dt, 1, geometry.nrec, max_shots=len(shots), dtype=model.dtype,
                            flags=flags)
    vp = model.vp.data
    if np.ndim(vp) == 0:
        vp = np.full(model.grid.shape, vp, dtype=model.dtype)
    damp = model.damp.data if hasattr(model.damp, 'data') else np.full(model.grid.shape, model.damp, model.dtype)
    # the damping profile changes with the model's boundary set-up only: it stays on the device between evaluations
    damp = np.ascontiguousarray(damp, dtype=model.dtype)
    dkey = (id(model.damp), damp.shape, float(damp.reshape(-1)[::97].sum(dtype=np.float64)))
    if getattr(entry, 'damp_key', None) != dkey:
        import torch
        entry.damp_dev = torch.as_tensor(damp).to('cuda')
        entry.damp_key = dkey
    entry.plan.set_model(np.ascontiguousarray(vp, dtype=model.dtype), entry.damp_dev)
    gkey = (geometry.src_positions[shots].tobytes(), geometry.rec_positions.tobytes())
    if entry.geom_key != gkey:
        entry.plan.set_geometry(geometry.src_positions[shots].reshape(len(shots), 1, model.dim), geometry.rec_positions)
        entry.geom_key = gkey
    for which in (0, 1, 2):
        entry.claim(which)
    return entry


_WAVELETS = {}


def _wavelet(geometry, dt=None):
    """(nt, 1) source time function of one shot (all shots share type, f0 and filter).  It depends on the time
    axis and the source parameters only, so it is formed once per such set (through the same classes as ever)."""
    dt = geometry.model.critical_dt if dt is None else dt
    key = (geometry.src_type, geometry.f0, geometry._t0w, geometry._a, float(geometry.t0), float(geometry.tn), float(dt),
           id(geometry._filter), geometry.model.dim, str(geometry.model.dtype))
    hit = _WAVELETS.get(key)
    if hit is None or hit[1] is not geometry._filter:
        if len(_WAVELETS) > 64:
            _WAVELETS.clear()
        hit = (np.ascontiguousarray(_shot_geometry(geometry, 0).src.data), geometry._filter)
        _WAVELETS[key] = hit
    return hit[0]


# Subtracting the direct wave from both synthetic and observed data (reference fwi.py:148-150)
# leaves the least-squares residual and objective unchanged, so that case also takes the route
# that forms the residual on the device.  Set to False to evaluate (syn - dw) - (obs - dw) on the
# host in float32 exactly as the reference does.
FUSE_DIRECT_WAVE = True

# Misfit objects with a `batch_device` method (the trace-wise Wasserstein misfit) are evaluated on the
# device for all shots at once.  Set to False to call them shot by shot on the host as the reference does.
DEVICE_MISFIT = True

# Keep the saved forward wavefield in half precision on the device (half the memory; the gradient
# changes by ~5e-5 relative on Marmousi-II).  Off by default: the default path is float32 throughout.
SAVE_HALF = False

_device_data = {}       # (plan id, ids of the shot arrays) -> (fingerprint, device tensor, arrays)
_DEVICE_DATA_MAX = 4


def clear_caches():
    """Forget the device-resident copies of observed data."""
    _device_data.clear()
    _resampled_data.clear()
    _MASK_DEV.clear()
    _WAVELETS.clear()
    _GRID_OUTPUTS.clear()


def _resident_batch(plan, objs, shots, dtype):
    """(ns, nt, nrec) device tensor of objs[i].data for i in shots, uploaded once and reused for as
    long as the same arrays are passed with unchanged content.  Content is watched through a
    strided checksum, which notices refilled arrays but not single-sample edits: call
    clear_caches() after editing observed data in place."""
    import torch
    arrays = [objs[i].data for i in shots]
    key = (id(plan),) + tuple(id(a) for a in arrays)
    finger = tuple(float(np.asarray(a).reshape(-1)[::4099].sum(dtype=np.float64)) for a in arrays)
    hit = _device_data.get(key)
    if hit is not None and hit[0] == finger:
        return hit[1]
    dev = torch.as_tensor(np.stack(arrays).astype(dtype, copy=False)).to('cuda')
    while len(_device_data) >= _DEVICE_DATA_MAX:
        _device_data.pop(next(iter(_device_data)))
    _device_data[key] = (finger, dev, arrays)
    return dev


class _DeviceResiduals(object):
    """Residuals of one batch kept on the GPU and copied to the host on first access; behaves
    like the (ns, nt, nrec) array it stands for."""

    def __init__(self, tensor):
        self._tensor = tensor
        self._host = None

    def _fetch(self):
        if self._host is None:
            self._host = self._tensor.cpu().numpy()
            self._tensor = None
        return self._host

    def __getitem__(self, k):
        return self._fetch()[k]

    def __len__(self):
        return len(self._fetch())

    def __array__(self, dtype=None, copy=None):
        a = self._fetch()
        return a if dtype is None else a.astype(dtype)


class _ResidualList(object):
    """What fwi_obj_multi returns as `residuals`: indexable per shot like the reference's list of
    (nt, nrec) arrays (fwi.py:184-189), fetching from the device only when an entry is read."""

    def __init__(self, nsrc):
        self._rows = [None] * nsrc

    def _set(self, i, batch, k):
        self._rows[i] = (batch, k)

    def _set_array(self, i, arr):
        self._rows[i] = arr

    def __len__(self):
        return len(self._rows)

    def __getitem__(self, i):
        row = self._rows[i]
        if isinstance(row, tuple):
            return row[0][row[1]]
        return row

    def __iter__(self):
        return (self[i] for i in range(len(self)))


_GRID_OUTPUTS = {}


def _grid_outputs(plan, ns, dtype):
    """Device tensors for the per-shot padded-grid gradients and illuminations of a batch: they are cropped,
    muted and summed where they are, and only the summed physical-grid arrays ever cross to the host.  One pair per
    plan and batch size, overwritten by every evaluation (nothing of it is handed to the caller)."""
    import torch
    key = (id(plan), ns, str(dtype))
    hit = _GRID_OUTPUTS.get(key)
    if hit is None or hit[0] is not plan:
        if len(_GRID_OUTPUTS) > 8:
            _GRID_OUTPUTS.clear()
        shape = (ns,) + tuple(plan.shape)
        hit = (plan, torch.empty(shape, dtype=dtype, device='cuda'), torch.empty(shape, dtype=dtype, device='cuda'))
        _GRID_OUTPUTS[key] = hit
    hit[1]._mifwi_plan = plan
    return hit[1], hit[2]


def _compute_batch(geometry, shots, obs, direct_wave, misfit_func, calc_grad, dt=None):
    """Objective pieces of the shots `shots` (indices into the survey) as one GPU batch:
    (fvals per shot, residuals (ns, nt, nrec), padded-grid gradients (ns, *grid) or None,
    illumination likewise).  With the least-squares misfit the residual is formed on the device
    from device-resident observed data and stays there until somebody reads it; with any other
    misfit function the misfit runs on the host between the two passes."""
    model = geometry.model
    entry = _batch_entry(geometry, shots, dt)
    plan = entry.plan
    ns = len(shots)
    wavelet = _wavelet(geometry, dt)
    if misfit_func is least_square and (direct_wave is None or FUSE_DIRECT_WAVE):
        import torch
        obs_dev = _resident_batch(plan, obs, shots, model.dtype)
        res_dev = torch.empty_like(obs_dev)
        g_dev, il_dev = _grid_outputs(plan, ns, obs_dev.dtype) if calc_grad else (None, None)
        out = plan.l2_misfit_gradient(wavelet, obs_dev, nshots=ns, calc_grad=calc_grad, want_illum=True,
                                      residual=res_dev, grad=g_dev, illum=il_dev)
        return out['fval'], _DeviceResiduals(res_dev), out['grad'], out['illum']
    if DEVICE_MISFIT and getattr(misfit_func, 'method', None) == '1d' and hasattr(misfit_func, 'batch_device'):
        # trace-wise misfits with a batched tensor form: the gathers stay on the device between the passes
        import torch
        obs_dev = _resident_batch(plan, obs, shots, model.dtype)
        syn_dev = torch.empty_like(obs_dev)
        plan.forward(wavelet, nshots=ns, save=calc_grad, rec_out=syn_dev)
        f_dev, o_dev = syn_dev, obs_dev
        if direct_wave is not None:
            dw_dev = _resident_batch(plan, direct_wave, shots, model.dtype)
            f_dev, o_dev = syn_dev - dw_dev, obs_dev - dw_dev
        fvals, res_dev = misfit_func.batch_device(f_dev, o_dev)
        g_b = il_b = None
        if calc_grad:
            g_b, il_b = _grid_outputs(plan, ns, obs_dev.dtype)
            plan.gradient(res_dev, nshots=ns, grad=g_b, illum=il_b)
        return fvals.cpu().numpy(), _DeviceResiduals(res_dev), g_b, il_b
    obs_b = np.stack([obs[i].data for i in shots]).astype(model.dtype, copy=False)
    dw_b = None
    if direct_wave is not None:
        dw_b = np.stack([direct_wave[i].data for i in shots]).astype(model.dtype, copy=False)
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
        import torch
        g_b, il_b = _grid_outputs(plan, ns, torch.float32 if model.dtype == np.float32 else torch.float64)
        plan.gradient(res_b, nshots=ns, grad=g_b, illum=il_b)
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


_resampled_data = {}     # (id of the data array, dt) -> (fingerprint, resampled object, array)


def _resampled(obj, dt):
    """obj.resample(dt), remembered for as long as the same data array is passed with unchanged content
    (observed and direct-wave gathers are resampled identically at every objective evaluation)."""
    arr = obj.data
    key = (id(arr), float(dt))
    finger = float(np.asarray(arr).reshape(-1)[::4099].sum(dtype=np.float64))
    hit = _resampled_data.get(key)
    if hit is not None and hit[0] == finger:
        return hit[1]
    out = obj.resample(dt)
    while len(_resampled_data) >= 256:
        _resampled_data.pop(next(iter(_resampled_data)))
    _resampled_data[key] = (finger, out, arr)
    return out


def fwi_obj_single(geometry, obs, misfit_func, direct_wave=None, resample_dt=None, calc_grad=False):
    """Objective, muted gradient, residual and muted illumination of one shot
    (reference fwi.py:131-173)."""
    solver = AcousticWaveSolver(geometry.model, geometry, space_order=geometry.model.space_order)
    pred, wfd = solver.forward(vp=geometry.model.vp, save=calc_grad)[0:2]
    dw = direct_wave
    if resample_dt is None:
        resample_dt = geometry.dt
    else:
        obs = _resampled(obs, resample_dt)
        pred = pred.resample(resample_dt)
        if direct_wave is not None:
            dw = _resampled(direct_wave, resample_dt)
    syn_data = pred.data
    obs_data = obs.data
    pred_times = pred.time_values
    # The modelling time step follows max(vp) of each model, so the resampled axes of observed and
    # predicted data can differ by a sample once the model has been updated (the reference stops with
    # a shape error there); compare them on the samples they share.
    n = min(syn_data.shape[0], obs_data.shape[0], dw.data.shape[0] if direct_wave is not None else 1 << 30)
    syn_data, obs_data, pred_times = syn_data[:n], obs_data[:n], pred_times[:n]
    if direct_wave is not None:
        syn_data = syn_data - dw.data[:n]
        obs_data = obs_data - dw.data[:n]
    fval, residual_data = misfit_func(syn_data, obs_data)
    residual = Receiver(name="rec", grid=geometry.model.grid, time_range=geometry.time_axis,
                        coordinates=geometry.rec_positions)
    residual.data[:] = resample(residual_data, geometry.time_axis.time_values, pred_times)[:]
    illum, crop_grad = None, None
    if calc_grad:
        grad = Function(name="grad", grid=geometry.model.grid)
        # sum over time of u^2 (fwi.py:171 of the reference reduces the saved wavefield on the host):
        # accumulated by the gradient kernel while the wavefield is still on the device
        illum_f = Function(name="illum", grid=geometry.model.grid)
        solver.gradient(rec=residual, u=wfd, vp=geometry.model.vp, grad=grad, illum=illum_f)
        crop = geometry.model.physical_slices
        crop_grad = np.array(grad.data[:])[crop]
        crop_grad = fix_source_illumination(geometry, crop_grad)
        illum = np.array(illum_f.data[:])[crop]
        illum = fix_source_illumination(geometry, illum)
    return fval, crop_grad, residual.data, illum


def _close(a, b):
    """np.isclose(a, b) for two scalars (its defaults), without its array machinery."""
    a, b = float(a), float(b)
    return abs(a - b) <= 1e-8 + 1e-5 * abs(b)


def _same_sampling(geometry, objs, dt=None):
    dt = geometry.model.critical_dt if dt is None else dt
    if not _close(geometry.dt, dt):
        return False
    return all(o is None or _close(o.time_range.step, dt) for o in objs)


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
    grad = illum = None
    if calc_grad:
        grad = np.zeros(model.shape)
        illum = np.zeros(model.shape)
    residuals = _ResidualList(nsrc)
    dt_model = model.critical_dt      # once per evaluation: a pass over the velocity model
    if mine and not _same_sampling(geometry, [obs[i] for i in mine] +
                                   ([direct_wave[i] for i in mine] if direct_wave is not None else []), dt_model):
        # resampled data: shot-by-shot, as the reference does it
        for i in mine:
            f_, g_, r_, il_ = fwi_obj_single(_shot_geometry(geometry, i), obs[i], misfit_func,
                                             direct_wave[i] if direct_wave is not None else None, geometry.dt,
                                             calc_grad)
            fval += f_
            residuals._set_array(i, r_)
            if calc_grad:
                grad += g_
                illum += il_
    elif mine:
        fvals, res_b, g_b, il_b = _compute_batch(geometry, mine, obs, direct_wave, misfit_func, calc_grad, dt_model)
        crop = model.physical_slices
        for k, i in enumerate(mine):
            fval += float(fvals[k])
            residuals._set(i, res_b, k)
        if calc_grad and type(g_b).__module__.startswith('torch'):
            # device route: crop, mute (per shot, before the sum, as fwi.py:166-171 of the reference does), shot sum,
            # all-reduce, preconditioner and mask on the GPU, in float64 like the host route; one small copy back
            # (mifwi_sum_shots, mifwi_precondition: the arithmetic of the host route below, bit for bit)
            import torch
            plan = g_b._mifwi_plan
            n = int(np.prod(model.shape))
            packed = torch.empty(1 + 2 * n, dtype=torch.float64, device=g_b.device)
            g_sum, il_sum = packed[1:1 + n], packed[1 + n:]
            mutes = _mute_factors_device(model, geometry, mine, g_b.device)
            dist = _dist()
            if dist is not None and dist.get_world_size() > 1:
                packed[:1].fill_(fval)
            plan.sum_shots(g_b, il_b, mutes, [c.start or 0 for c in crop], model.shape, g_sum=g_sum, il_sum=il_sum)
            if dist is not None and dist.get_world_size() > 1:
                packed = _all_reduce_sum_device(packed)
                g_sum, il_sum = packed[1:1 + n], packed[1 + n:]
                fval = float(packed[0].item())
            mask_dev = _mask_device(mask, g_b.device).reshape(-1) if mask is not None else None
            host = plan.precondition(g_sum, il_sum if precond else None, mask_dev)
            return fval, host, residuals
        if calc_grad:
            for k, i in enumerate(mine):
                mute = _mute_factor(model.shape, model.spacing, geometry.src_positions[i], geometry.rec_positions)
                grad += np.asarray(g_b[k])[crop] * mute
                illum += np.asarray(il_b[k])[crop] * mute
    if calc_grad:
        fval, (grad, illum) = _all_reduce_sum(fval, [grad, illum])
        if precond:
            grad /= np.sqrt(illum + 1e-30)
        if mask is not None:
            grad *= mask
    else:
        fval, _ = _all_reduce_sum(fval, [])
        grad = np.zeros(model.shape)
    return fval, grad.reshape(-1).astype(np.float64), residuals


def fwi_loss(x, geometry, obs, misfit_func, direct_wave=None, mask=None, precond=True, calc_grad=True):
    """Objective and gradient with respect to squared slowness x = 1/vp^2 (reference fwi.py:236-246)."""
    v = 1. / np.sqrt(x.reshape(geometry.model.shape))
    geometry.model.update('vp', v.reshape(geometry.model.shape))
    return fwi_obj_multi(geometry, obs, misfit_func, direct_wave, mask, precond, calc_grad)
