import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

F32, F64 = 0, 1
FLAG_ASYNC, FLAG_NO_GRAPH, FLAG_GENERIC, FLAG_CHECKPOINT, FLAG_FREE_SURFACE, FLAG_OT4 = 1, 2, 4, 8, 16, 32
FLAG_SAVE_F16 = 64
FLAG_TIME_BLOCK = 128
FLAG_PML = 256

# every symbol include/mifwi.h declares
EXPORTED_SYMBOLS = (
    "mifwi_version", "mifwi_last_error", "mifwi_plan_create", "mifwi_plan_destroy", "mifwi_set_stream",
    "mifwi_synchronize", "mifwi_set_model", "mifwi_set_geometry", "mifwi_forward", "mifwi_adjoint",
    "mifwi_gradient", "mifwi_l2_misfit_gradient", "mifwi_read_wavefield", "mifwi_write_wavefield",
    "mifwi_get_stats", "mifwi_born", "mifwi_tti_set_model", "mifwi_tti_forward", "mifwi_tti_adjoint",
    "mifwi_tti_born", "mifwi_tti_gradient", "mifwi_l2_misfit", "mifwi_l2_misfit_noalloc",
    "mifwi_set_pml", "mifwi_set_pml_layout", "mifwi_sum_shots", "mifwi_precondition",
)


class MifwiError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("mifwi error %d: %s" % (code, text))
        self.code = code


class _Config(ctypes.Structure):
    _fields_ = [("ndim", ctypes.c_int32), ("shape", ctypes.c_int32 * 3), ("spacing", ctypes.c_double * 3),
                ("origin", ctypes.c_double * 3), ("space_order", ctypes.c_int32), ("nt", ctypes.c_int32),
                ("dt", ctypes.c_double), ("dtype", ctypes.c_int32), ("nsrc", ctypes.c_int32),
                ("nrec", ctypes.c_int32), ("max_shots", ctypes.c_int32), ("device", ctypes.c_int32),
                ("flags", ctypes.c_int32)]


class _Stats(ctypes.Structure):
    _fields_ = [("kernel_ms", ctypes.c_double), ("points", ctypes.c_double), ("gpointss", ctypes.c_double),
                ("algorithmic_bytes", ctypes.c_double), ("fwd_ms", ctypes.c_double), ("bwd_ms", ctypes.c_double),
                ("launches", ctypes.c_int32), ("fwd_launches", ctypes.c_int32), ("bwd_launches", ctypes.c_int32),
                ("kernel_kind", ctypes.c_int32), ("device_bytes", ctypes.c_int64), ("ckpt_steps", ctypes.c_int32),
                ("ckpt_segment", ctypes.c_int32), ("fwd_kernel_kind", ctypes.c_int32),
                ("bwd_kernel_kind", ctypes.c_int32), ("persist_fallbacks", ctypes.c_int32),
                ("reserved_", ctypes.c_int32)]

KERNEL_NAMES = {0: ("step_generic_kernel", "step_generic_kernel<GRAD>"),
                1: ("step2d_kernel", "step2d_kernel<GRAD>"),
                2: ("step3d_kernel", "step3d_kernel<GRAD>"),
                3: ("persist2d_kernel", "persist2d_kernel<GRAD>"),
                5: ("persist_tti2d_kernel", "persist_tti2d_kernel<BWD>")}


_LIB = None


def library_path():
    # MIFWI_LIB selects another build of the same library (kernel tuning experiments)
    return os.environ.get("MIFWI_LIB") or os.path.join(os.path.dirname(_HERE), "libmifwi.so")


def load_library():
    """Load libmifwi.so (built in-tree by `make -C devito-fwi_amd/csrc`).  Fails loudly if absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError("libmifwi.so not found at %s: build it with `make -C devito-fwi_amd/csrc` "
                          "(there is no CPU fallback)" % path)
    # PyTorch wheels carry their own HIP runtime.  When this library is loaded first, the system
    # runtime it links against is the one the process keeps, and torch then fails to see any GPU;
    # loading torch's runtime first works for both.  torch is optional for the C ABI itself.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    vp = ctypes.c_void_p
    i32 = ctypes.c_int32
    lib.mifwi_version.restype = ctypes.c_int
    lib.mifwi_last_error.restype = ctypes.c_char_p
    lib.mifwi_plan_create.argtypes = [ctypes.POINTER(_Config), ctypes.POINTER(vp)]
    lib.mifwi_plan_destroy.argtypes = [vp]
    lib.mifwi_set_stream.argtypes = [vp, vp]
    lib.mifwi_synchronize.argtypes = [vp]
    lib.mifwi_set_model.argtypes = [vp, vp, vp]
    lib.mifwi_set_geometry.argtypes = [vp, i32, vp, vp, i32]
    lib.mifwi_forward.argtypes = [vp, i32, vp, i32, vp, i32]
    lib.mifwi_adjoint.argtypes = [vp, i32, vp, vp]
    lib.mifwi_born.argtypes = [vp, i32, vp, i32, vp, vp]
    lib.mifwi_tti_set_model.argtypes = [vp, vp, vp, vp, vp]
    lib.mifwi_tti_forward.argtypes = [vp, i32, vp, i32, vp, i32]
    lib.mifwi_tti_born.argtypes = [vp, i32, vp, i32, vp, vp]
    lib.mifwi_tti_gradient.argtypes = [vp, i32, vp, vp, i32]
    lib.mifwi_tti_adjoint.argtypes = [vp, i32, vp, vp]
    lib.mifwi_gradient.argtypes = [vp, i32, vp, vp, i32, vp]
    lib.mifwi_l2_misfit_gradient.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp]
    lib.mifwi_set_pml.argtypes = [vp, i32, ctypes.c_double]
    lib.mifwi_set_pml_layout.argtypes = [vp, i32, ctypes.c_double, i32, i32, vp]
    lib.mifwi_l2_misfit.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.mifwi_l2_misfit_noalloc.argtypes = [i32, i32, ctypes.c_int64, vp, vp, vp, vp, vp]
    lib.mifwi_sum_shots.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.mifwi_precondition.argtypes = [vp, ctypes.c_int64, vp, vp, vp, vp]
    lib.mifwi_read_wavefield.argtypes = [vp, i32, i32, i32, i32, vp]
    lib.mifwi_write_wavefield.argtypes = [vp, i32, i32, i32, vp]
    lib.mifwi_get_stats.argtypes = [vp, ctypes.POINTER(_Stats)]
    for name in EXPORTED_SYMBOLS:
        getattr(lib, name)
        if name not in ("mifwi_last_error",):
            getattr(lib, name).restype = ctypes.c_int
    _LIB = lib
    return lib


def _is_torch(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _order_after_torch(x):
    """A plan works on a stream of its own, which is not ordered with torch's streams.  Before a tensor's
    memory is handed to the library, wait for the work torch has queued on its current stream of that
    device: kernels still producing an input, or still reading an output buffer the call will overwrite.
    (The other direction needs nothing: a call returns after the plan's stream has drained, and with
    MIFWI_FLAG_ASYNC the caller synchronises the plan before touching results.)"""
    if x.is_cuda:
        import torch
        torch.cuda.current_stream(x.device).synchronize()


def l2_misfit_noalloc(syn, obs, residual, fval, stream=0):
    """mifwi_l2_misfit_noalloc on torch tensors of the current device: syn, obs, residual (nshots, ...) float32 or
    float64, fval (nshots,) float64; enqueued on `stream` (a raw hipStream_t, 0 = default stream).  Allocates
    nothing and does not wait."""
    lib = load_library()
    for t in (syn, obs, residual, fval):
        assert _is_torch(t) and t.is_cuda and t.is_contiguous()
    assert syn.shape == obs.shape == residual.shape and syn.dtype == obs.dtype == residual.dtype
    nshots = int(syn.shape[0])
    n = int(syn.numel() // max(1, nshots))
    dt = F64 if syn.element_size() == 8 else F32
    rc = lib.mifwi_l2_misfit_noalloc(dt, nshots, n, ctypes.c_void_p(syn.data_ptr()), ctypes.c_void_p(obs.data_ptr()),
                                     ctypes.c_void_p(residual.data_ptr()), ctypes.c_void_p(fval.data_ptr()),
                                     ctypes.c_void_p(int(stream)))
    if rc != 0:
        raise MifwiError(rc, lib.mifwi_last_error().decode())


class Plan(object):
    """
    One discretised acquisition on one GPU: grid, stencil order, time axis, batch capacity.
    Array arguments are numpy arrays (host) or torch tensors on the plan's device; both are
    passed to the C ABI as plain pointers.
    """

    def __init__(self, shape, spacing, origin, space_order, nt, dt, nsrc, nrec, max_shots=1,
                 dtype=np.float32, device=-1, flags=0):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise TypeError("dtype must be float32 or float64")
        self.shape = tuple(int(s) for s in shape)
        self.ndim = len(self.shape)
        self.nt, self.nsrc, self.nrec, self.max_shots = int(nt), int(nsrc), int(nrec), int(max_shots)
        self.dt = float(dt)
        self.space_order = int(space_order)
        cfg = _Config()
        cfg.ndim = self.ndim
        for d in range(self.ndim):
            cfg.shape[d] = self.shape[d]
            cfg.spacing[d] = float(spacing[d])
            cfg.origin[d] = float(origin[d])
        cfg.space_order = self.space_order
        cfg.nt = self.nt
        cfg.dt = self.dt
        cfg.dtype = F64 if self.dtype == np.float64 else F32
        cfg.nsrc, cfg.nrec, cfg.max_shots = self.nsrc, self.nrec, self.max_shots
        cfg.device = int(device)
        cfg.flags = int(flags)
        self._check(self._lib.mifwi_plan_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        self.ngrid = int(np.prod(self.shape))

    # -- plumbing -------------------------------------------------------------------------
    def _check(self, rc):
        self._ordered = False       # the next call's first tensor argument waits for torch again
        if rc != 0:
            raise MifwiError(rc, self._lib.mifwi_last_error().decode())

    def _order(self, x):
        """_order_after_torch once per library call: the arguments of one call are gathered without any torch
        work in between, so the wait for the first of them covers the rest."""
        if not getattr(self, "_ordered", False):
            _order_after_torch(x)
            self._ordered = x.is_cuda

    def _in(self, x, shape=None):
        """(pointer, keep-alive) of an input array in the plan dtype."""
        if x is None:
            return None, None
        if _is_torch(x):
            assert x.is_contiguous()
            self._order(x)
            return ctypes.c_void_p(x.data_ptr()), x
        a = np.ascontiguousarray(x, dtype=self.dtype)
        if shape is not None:
            a = a.reshape(shape)
        return a.ctypes.data_as(ctypes.c_void_p), a

    def _out(self, x, shape, dtype=None):
        """(pointer, array) of an output: a new numpy array when x is None."""
        dtype = dtype or self.dtype
        if x is None:
            x = np.empty(shape, dtype=dtype)
        if _is_torch(x):
            assert x.is_contiguous()
            self._order(x)
            return ctypes.c_void_p(x.data_ptr()), x
        assert isinstance(x, np.ndarray) and x.flags.c_contiguous and x.dtype == dtype, "bad output array"
        assert x.size == int(np.prod(shape)), "bad output size"
        return x.ctypes.data_as(ctypes.c_void_p), x

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.mifwi_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- API ----------------------------------------------------------------------------------
    def set_stream(self, hip_stream):
        self._check(self._lib.mifwi_set_stream(self._h, ctypes.c_void_p(int(hip_stream))))

    def synchronize(self):
        self._check(self._lib.mifwi_synchronize(self._h))

    def set_model(self, vp, damp=None):
        pv, kv = self._in(vp)
        pd, kd = self._in(damp)
        n = kv.numel() if _is_torch(kv) else kv.size
        assert n == self.ngrid, "vp must cover the padded grid"
        self._check(self._lib.mifwi_set_model(self._h, pv, pd))
        # device tensors are read by a kernel that may still be queued: their memory must not go back to torch's
        # allocator before the plan has waited, so they are held until the next model replaces them
        self._model_keep = (kv, kd)

    def set_pml(self, npml, quibar=0.05, sides=None, edges=False, vp_staggered=None):
        """Layer width (cells) and strength of a FLAG_PML plan.  sides: (low x, high x, low z, high z) booleans, default
        all; edges=True: regions and boundary rows as the reference's notebook has them; vp_staggered: the caller's
        velocity at the staggered points (padded grid)."""
        if sides is None and not edges and vp_staggered is None:
            self._check(self._lib.mifwi_set_pml(self._h, int(npml), float(quibar)))
            return
        mask = 15 if sides is None else sum(1 << k for k, on in enumerate(sides) if on)
        pv, kv = self._in(vp_staggered)
        self._check(self._lib.mifwi_set_pml_layout(self._h, int(npml), float(quibar), mask, int(bool(edges)), pv))

    def set_geometry(self, src_coords, rec_coords, rec_per_shot=False, nshots=None):
        sc = np.ascontiguousarray(src_coords, dtype=self.dtype)
        if self.nsrc == 0:      # no sources: the shot count cannot be read off the coordinates
            nshots = int(nshots or (sc.shape[0] if sc.ndim == 3 else 1))
            sc = np.zeros((nshots, 1, self.ndim), dtype=self.dtype)
        else:
            sc = sc.reshape(-1, self.nsrc, self.ndim)
        nshots = sc.shape[0]
        rc_ = np.ascontiguousarray(rec_coords, dtype=self.dtype)
        rc_ = rc_.reshape((nshots, self.nrec, self.ndim) if rec_per_shot else (self.nrec, self.ndim))
        self._check(self._lib.mifwi_set_geometry(self._h, nshots, sc.ctypes.data_as(ctypes.c_void_p),
                                                 rc_.ctypes.data_as(ctypes.c_void_p), int(bool(rec_per_shot))))
        self.nshots = nshots
        return nshots

    def forward(self, wavelet, nshots=None, save=False, rec_out=None, wavelet_per_shot=False, want_rec=True):
        nshots = nshots or self.nshots
        pw, kw = self._in(wavelet)
        pr, rec = (None, None)
        if want_rec:
            pr, rec = self._out(rec_out, (nshots, self.nt, self.nrec))
        self._check(self._lib.mifwi_forward(self._h, nshots, pw, int(bool(wavelet_per_shot)), pr, int(bool(save))))
        return rec

    def born(self, wavelet, dm, nshots=None, rec_out=None, wavelet_per_shot=False):
        """Linearised data (nshots, nt, nrec) for the model perturbation dm (padded grid)."""
        nshots = nshots or self.nshots
        pw, kw = self._in(wavelet)
        pd, kd = self._in(dm)
        n = kd.numel() if _is_torch(kd) else kd.size
        assert n == self.ngrid, "dm must cover the padded grid"
        pr, rec = self._out(rec_out, (nshots, self.nt, self.nrec))
        self._check(self._lib.mifwi_born(self._h, nshots, pw, int(bool(wavelet_per_shot)), pd, pr))
        return rec

    # -- TTI ('centered' kernel) -----------------------------------------------------------------
    def tti_set_model(self, epsilon, delta, theta, phi=None):
        """Thomsen parameters and angles over the padded grid (phi only in 3-D, may be None)."""
        ptrs, keep = [], []
        for a in (epsilon, delta, theta, phi):
            pa, ka = self._in(a)
            if ka is not None:
                n = ka.numel() if _is_torch(ka) else ka.size
                assert n == self.ngrid, "TTI parameters must cover the padded grid"
            ptrs.append(pa)
            keep.append(ka)
        self._check(self._lib.mifwi_tti_set_model(self._h, *ptrs))

    def tti_forward(self, wavelet, nshots=None, rec_out=None, wavelet_per_shot=False, save=False):
        nshots = nshots or self.nshots
        pw, kw = self._in(wavelet)
        pr, rec = self._out(rec_out, (nshots, self.nt, self.nrec))
        self._check(self._lib.mifwi_tti_forward(self._h, nshots, pw, int(bool(wavelet_per_shot)), pr,
                                                int(bool(save))))
        return rec

    def tti_born(self, wavelet, dm, nshots=None, rec_out=None, wavelet_per_shot=False):
        """Linearised TTI data (nshots, nt, nrec) for the model perturbation dm (padded grid)."""
        nshots = nshots or self.nshots
        pw, kw = self._in(wavelet)
        pd, kd = self._in(dm)
        n = kd.numel() if _is_torch(kd) else kd.size
        assert n == self.ngrid, "dm must cover the padded grid"
        pr, rec = self._out(rec_out, (nshots, self.nt, self.nrec))
        self._check(self._lib.mifwi_tti_born(self._h, nshots, pw, int(bool(wavelet_per_shot)), pd, pr))
        return rec

    def tti_gradient(self, residual, nshots=None, grad=None, accumulate=False):
        """Per-shot TTI gradients (nshots, *shape) from the fields kept by tti_forward(save=True)."""
        nshots = nshots or self.nshots
        pi, ki = self._in(residual)
        pg, grad = self._out(grad, (nshots,) + self.shape)
        self._check(self._lib.mifwi_tti_gradient(self._h, nshots, pi, pg, int(bool(accumulate))))
        return grad

    def tti_adjoint(self, rec_in, nshots=None, srca_out=None):
        nshots = nshots or self.nshots
        pi, ki = self._in(rec_in)
        po, srca = self._out(srca_out, (nshots, self.nt, self.nsrc))
        self._check(self._lib.mifwi_tti_adjoint(self._h, nshots, pi, po))
        return srca

    def adjoint(self, rec_in, nshots=None, srca_out=None):
        nshots = nshots or self.nshots
        pi, ki = self._in(rec_in)
        po, srca = self._out(srca_out, (nshots, self.nt, self.nsrc))
        self._check(self._lib.mifwi_adjoint(self._h, nshots, pi, po))
        return srca

    def gradient(self, residual, nshots=None, grad=None, accumulate=False, illum=None, want_illum=False):
        nshots = nshots or self.nshots
        pi, ki = self._in(residual)
        pg, grad = self._out(grad, (nshots,) + self.shape)
        pl, il = (None, None)
        if want_illum or illum is not None:
            pl, il = self._out(illum, (nshots,) + self.shape)
        self._check(self._lib.mifwi_gradient(self._h, nshots, pi, pg, int(bool(accumulate)), pl))
        return (grad, il) if pl is not None else grad

    def l2_misfit_gradient(self, wavelet, obs, nshots=None, wavelet_per_shot=False, calc_grad=True,
                           grad=None, illum=None, want_illum=True, want_residual=False, want_syn=False, fval=None,
                           residual=None):
        nshots = nshots or self.nshots
        pw, kw = self._in(wavelet)
        po, ko = self._in(obs)
        pf, fval = self._out(fval, (nshots,), np.dtype(np.float64))
        pg = pl = pr = ps = None
        il = res = syn = None
        if calc_grad:
            pg, grad = self._out(grad, (nshots,) + self.shape)
            if want_illum or illum is not None:
                pl, il = self._out(illum, (nshots,) + self.shape)
        else:
            grad = None
        if want_residual or residual is not None:
            pr, res = self._out(residual, (nshots, self.nt, self.nrec))
        if want_syn:
            ps, syn = self._out(None, (nshots, self.nt, self.nrec))
        self._check(self._lib.mifwi_l2_misfit_gradient(self._h, nshots, pw, int(bool(wavelet_per_shot)), po, pf,
                                                       pg, pl, pr, ps))
        return dict(fval=fval, grad=grad, illum=il, residual=res, syn=syn)

    def l2_misfit(self, syn, obs, nshots=None, residual=None, want_residual=True, fval=None):
        """(fval per shot, residual) of the least-squares misfit for gathers of shape (nshots, nt, nrec)."""
        nshots = nshots or self.nshots
        ps, ks = self._in(syn)
        po, ko = self._in(obs)
        pr, res = (None, None)
        if want_residual or residual is not None:
            pr, res = self._out(residual, (nshots, self.nt, self.nrec))
        pf, fval = self._out(fval, (nshots,), np.dtype(np.float64))
        self._check(self._lib.mifwi_l2_misfit(self._h, nshots, ps, po, pr, pf))
        return fval, res

    def sum_shots(self, grad, illum, mute, lo, phys_shape, g_sum=None, il_sum=None):
        """Shot sums of the cropped, muted per-shot gradients and illuminations (device tensors (nshots, *shape));
        mute: device float64 (nshots, *phys_shape) or None.  Returns (g_sum, il_sum) float64 over the physical grid
        (numpy arrays unless tensors are passed in); il_sum is None when illum is."""
        nshots = int(grad.shape[0])
        f64 = np.dtype(np.float64)
        pg, kg = self._in(grad)
        pi, ki = self._in(illum)
        pm = ctypes.c_void_p(mute.data_ptr()) if mute is not None else None
        lo_a = (ctypes.c_int32 * self.ndim)(*[int(v) for v in lo])
        ph_a = (ctypes.c_int32 * self.ndim)(*[int(v) for v in phys_shape])
        shape = tuple(int(v) for v in phys_shape)
        ps, g_sum = self._out(g_sum, shape, f64)
        pl, il_sum = self._out(il_sum, shape, f64) if illum is not None else (None, None)
        self._check(self._lib.mifwi_sum_shots(self._h, nshots, pg, pi, pm, lo_a, ph_a, ps, pl))
        return g_sum, il_sum

    def precondition(self, g_sum, il_sum=None, mask=None, out=None):
        """g_sum / sqrt(il_sum + 1e-30) * mask on device float64 tensors (il_sum, mask optional); `out` a numpy
        array (default: a new one) or a device tensor."""
        n = int(g_sum.numel())
        self._order(g_sum)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        po, out = self._out(out, (n,), np.dtype(np.float64))
        self._check(self._lib.mifwi_precondition(self._h, n, ptr(g_sum), ptr(il_sum), ptr(mask), po))
        return out

    def read_wavefield(self, which=0, shot=0, first=0, count=None, out=None):
        nlev = self.nt if which in (0, 7, 8) else 3
        count = nlev - first if count is None else count
        po, arr = self._out(out, (count,) + self.shape)
        self._check(self._lib.mifwi_read_wavefield(self._h, which, shot, first, count, po))
        return arr

    def write_wavefield(self, u, shot=0, first=0):
        pi, ki = self._in(u)
        n = ki.numel() if _is_torch(ki) else ki.size
        count = n // self.ngrid
        self._check(self._lib.mifwi_write_wavefield(self._h, shot, first, count, pi))

    def stats(self):
        st = _Stats()
        self._check(self._lib.mifwi_get_stats(self._h, ctypes.byref(st)))
        out = {k: getattr(st, k) for k, _ in _Stats._fields_}
        # names of the kernels the last forward and the last backward loop of the plan really ran on
        out["fwd_kernel"] = KERNEL_NAMES.get(st.fwd_kernel_kind, ("?", "?"))[0]
        out["bwd_kernel"] = KERNEL_NAMES.get(st.bwd_kernel_kind, ("?", "?"))[1]
        out.pop("reserved_", None)
        return out
