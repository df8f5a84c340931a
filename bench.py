#!/usr/bin/env python
"""
Benchmark of the acoustic FWI hot path on MI355X.

Metric: Gpoints/s (forward + adjoint/gradient) on Marmousi-II, 8 shots per GPU.
One "step" = one gradient evaluation of the L2 objective for this rank's 8 shots:
forward modelling with the wavefield saved on the device, residual against the observed data,
backward propagation with zero-lag cross-correlation, per-shot source/receiver muting and the
sum over shots; with N > 1 ranks the summed gradient, illumination and objective are
all-reduced over RCCL.  Each rank owns 8 different shots of an 8N-shot survey (weak scaling).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

    ... bench.py --gpus N --shots-total 32     # strong scaling: BASELINE config 3, a fixed 32-shot survey

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (the gradient time step);
`cpu_baseline` is the CPU oracle (a restatement of the Devito CPU path, not Devito itself) timed
on the host cores for a bounded sample of the same workload, in two modes: shots one after the other
with every core on one shot (how the reference runs), and one shot per core side by side.
At N = 1 `shot_batches` times the same gradient evaluation for batches of 4 and 32 shots: their ratio is
what eight GPUs with 4 shots each can gain over one GPU with all 32 (before the all-reduce).  At every N
`strong_scaling` times a fixed 32-shot survey (BASELINE config 3) and a fixed 8-shot survey split evenly over the
ranks, next to the weak-scaling headline (8 shots per GPU).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "devito-fwi_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
SHOTS_PER_GPU = 8


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same
    command (the newest profiles/r*_traffic.json), or None when no such measurement is recorded."""
    return _profile_entry(kernel).get("bytes")


PROFILE_FILES = ("r3_traffic.json", "r2_traffic.json")    # newest first


def _profile_entry(kernel):
    for name in PROFILE_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return dict(json.load(f).get(kernel, {}), source="profiles/" + name)
        except (OSError, ValueError):
            continue
    return {}


def exchange_floor_us():
    """Cost of one step of the whole-loop kernel's neighbour exchange alone (counters, halo ring from the XCD's L2,
    tile store, acknowledgement; no stencil, no saved level), measured by tools/micro/xcdsplit.hip on this part and
    recorded in profiles/: the bound of that kernel."""
    for name in ("r3_exchange_floor.json", "r2_exchange_floor.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
                return float(d["us_per_step"]), "profiles/" + name
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def profile_launch_us(kernel):
    """Average duration of one launch of `kernel` in the committed rocprofv3 kernel trace of this same
    command (microseconds), or None."""
    return _profile_entry(kernel).get("avg_us")


def marmousi2(space_order):
    """Marmousi-II set-up of the reference's marmousi2_fwi.py:61-103 (rasters are the reference's
    model_data/SMARM2 files, kept as fixtures)."""
    from seismic import Model
    d = os.path.join(ROOT, "tests", "golden", "model_data", "SMARM2")
    shape = (340, 140)
    true_vp = np.fromfile(os.path.join(d, "vp.true"), dtype=np.float32).reshape(shape) / 1000
    smooth_vp = np.fromfile(os.path.join(d, "vp.smooth_20"), dtype=np.float32).reshape(shape) / 1000
    kw = dict(origin=(0, 0), spacing=(30., 30.), shape=shape, space_order=space_order, nbl=40, fs=False, dt=3.0,
              bcs="damp")
    return Model(vp=true_vp, **kw), Model(vp=smooth_vp, **kw)


def acquisition(model, nshots_total):
    spacing = model.spacing
    src = np.empty((nshots_total, 2))
    src[:, 0] = np.linspace(0, model.domain_size[0], num=nshots_total)
    src[:, -1] = 2 * spacing[0]
    nrec = model.shape[0]
    rec = np.empty((nrec, 2))
    rec[:, 0] = np.linspace(spacing[0], model.domain_size[0] - spacing[0], num=nrec)
    rec[:, 1] = 2 * spacing[0]
    return src, rec


def _cpu_one_shot(args):
    """One shot (forward with saved wavefield + gradient) on one core; runs in a worker process."""
    os.environ["OMP_NUM_THREADS"] = "1"
    vp_true, vp_init, spacing, nbl, so, dt, src, rec, wavelet = args
    from oracle import oracle as orc
    p_true = orc.Problem(vp_true, spacing, nbl, so, dt=dt)
    p_init = orc.Problem(vp_init, spacing, nbl, so, dt=dt)
    obs = p_true.forward(src, wavelet, rec, dt=dt)[0]
    t0 = time.perf_counter()
    syn, u = p_init.forward(src, wavelet, rec, dt=dt, save=True)
    p_init.gradient(rec, syn - obs, u, dt=dt)
    return time.perf_counter() - t0


def cpu_baseline(true_model, init_model, src, rec, wavelet, dt, space_order, nshots=2):
    """Oracle (CPU restatement of the Devito path) on a bounded sample of the same workload, two ways:
    (i) as the reference runs it, shots one after the other, OpenMP over the grid of one shot (`value`);
    (ii) best case for a CPU node, one shot per core side by side, one thread each (`shots_across_cores`)."""
    from oracle import oracle as orc
    nthreads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    nbl = true_model.nbl
    vt = np.ascontiguousarray(true_model.vp.data[nbl:-nbl, nbl:-nbl])
    vi = np.ascontiguousarray(init_model.vp.data[nbl:-nbl, nbl:-nbl])
    p_true = orc.Problem(vt, true_model.spacing, nbl, space_order, dt=dt)
    p_init = orc.Problem(vi, true_model.spacing, nbl, space_order, dt=dt)
    obs = [p_true.forward(src[i], wavelet, rec, dt=dt)[0] for i in range(nshots)]
    per_shot = 2 * (wavelet.shape[0] - 2) * int(np.prod(p_init.shape_pad))
    t0 = time.perf_counter()
    pts, reps = 0, 0
    while reps < 1 or (time.perf_counter() - t0 < 8.0 and reps < 50):     # about 8 s of CPU work
        for i in range(nshots):
            syn, u = p_init.forward(src[i], wavelet, rec, dt=dt, save=True)
            p_init.gradient(rec, syn - obs[i], u, dt=dt)
            pts += per_shot
        reps += 1
    el = time.perf_counter() - t0
    out = {"value": pts / el / 1e9, "unit": "Gpoints/s", "cores": nthreads, "kind": "port",
           "omp_num_threads": nthreads, "binding": "none (threads placed by the OS scheduler)",
           "mode": "shots serial, every core on one shot (OpenMP over the grid), as the reference's fwi.py:183-199 runs them",
           "sample": "%d shots of the same Marmousi-II workload (forward with saved wavefield + gradient, "
                     "full nt=%d), %d passes; %.1f s" % (nshots, wavelet.shape[0], reps, el)}
    # (ii) one single-threaded shot per core, all at once
    try:
        import multiprocessing as mp
        nproc = max(1, min(nthreads, 16))
        args = [(vt, vi, tuple(true_model.spacing), nbl, space_order, dt, np.asarray(src[i % len(src)]), rec, wavelet)
                for i in range(nproc)]
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(nproc) as pool:
            inner = pool.map(_cpu_one_shot, args)
        wall = time.perf_counter() - t0
        out["shots_across_cores"] = {
            "value": nproc * per_shot / max(inner) / 1e9, "unit": "Gpoints/s", "cores": nproc, "omp_num_threads": 1,
            "binding": "none (one process per shot, placed by the OS scheduler)",
            "sample": "%d shots at once, one process and one thread each, full nt; slowest shot %.1f s "
                      "(%.1f s with process start-up)" % (nproc, max(inner), wall)}
    except Exception as e:      # the baseline is a report, never a reason to lose the bench line
        out["shots_across_cores"] = {"error": repr(e)}
    return out


class _env(object):
    """An environment variable set for the creation of a plan (the library reads its switches there)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = os.environ.get(self.name)
        os.environ[self.name] = self.value

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.old


def roofline_3d(mifwi, device, space_order=16, n=512, nbl=40, nt=14):
    """Forward time step of the 2.5-D kernel on the synthetic 3-D stress grid (512^3 + sponge = 592^3,
    one shot): algorithmic 20 B per point update / kernel duration (HIP events around the time loop
    on the plan's stream; one launch per step, launches last ~1 ms so gaps are negligible)."""
    from seismic import Model
    shape = (n, n, n)
    vp = np.empty(shape, dtype=np.float32)
    vp[:] = np.linspace(1.5, 4.5, n, dtype=np.float32)[None, None, :]
    model = Model(origin=(0., 0., 0.), spacing=(10., 10., 10.), shape=shape, space_order=space_order, vp=vp,
                  nbl=nbl, bcs="damp")
    dt = float(model.critical_dt)
    sp = model.grid.shape
    ngrid = float(np.prod(sp))
    src = np.array([[n * 5.0, n * 5.0, 20.0]], dtype=np.float32)
    g = np.linspace(0, (n - 1) * 10.0, 64, dtype=np.float32)
    rec = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    rec = np.concatenate([rec, np.full((rec.shape[0], 1), 20.0, np.float32)], 1)
    # every point computed: a forward loop otherwise skips what the wave has not reached, which after these few steps
    # is most of the grid
    with _env("MIFWI_3D_BOX", "0"):
        plan = mifwi.Plan(sp, model.spacing, model.grid.origin, space_order, nt, dt, 1, rec.shape[0], max_shots=1,
                          device=device)
    plan.set_model(model.vp.data, model.damp.data)
    plan.set_geometry(src, rec)
    t = np.arange(nt) * dt
    r = np.pi * 0.010 * (t - 100.0)
    wav = ((1 - 2 * r ** 2) * np.exp(-r ** 2)).astype(np.float32).reshape(nt, 1)
    best = None
    for _ in range(3):
        plan.forward(wav, save=False, want_rec=False)
        st = plan.stats()
        ms = st["fwd_ms"] / st["fwd_launches"]
        best = ms if best is None else min(best, ms)
    plan.close()
    ach = 20.0 * ngrid / (best * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": pmc_traffic("step3d_kernel"), "kernel": "step3d_kernel", "avg_launch_us": best * 1e3,
            "profile_avg_launch_us": profile_launch_us("step3d_kernel"),
            "gradient_step_profile": {"kernel": "step3d_kernel<GRAD>", "traffic": pmc_traffic("step3d_kernel<GRAD>"),
                                      "profile_avg_launch_us": profile_launch_us("step3d_kernel<GRAD>"),
                                      "what": "the launch of the space_order=8 gradient loop that advances the adjoint "
                                              "field two steps' worth of gradient terms (config_3d_512)"},
            "gpoints_per_s": ngrid / (best * 1e-3) / 1e9,
            "workload": "synthetic 3-D %d^3 (+nbl %d = %d^3), space_order=%d, 1 shot, forward time step"
                        % (n, nbl, sp[0], space_order)}


def tti_leg(mifwi, device, true_model, src, rec, space_order=8, nshots=2):
    """2-D TTI on the Marmousi-II grid at space order 8, two shots (the per-GPU share of 16 shots on 8 GPUs):
    forward with both fields saved + gradient (Jacobian adjoint), full nt.  The reference ships no anisotropy
    rasters for Marmousi-II, so epsilon, delta and theta follow its `layers-tti` recipe applied to vp."""
    from seismic import Model, AcquisitionGeometry
    vp = np.asarray(true_model.vp.data[true_model.nbl:-true_model.nbl, true_model.nbl:-true_model.nbl], np.float32)
    scale = float(vp.max() - 1.5)
    model = Model(origin=(0, 0), spacing=true_model.spacing, shape=vp.shape, space_order=space_order, vp=vp,
                  nbl=true_model.nbl, bcs="damp", epsilon=0.3 * (vp - 1.5) / scale, delta=0.2 * (vp - 1.5) / scale,
                  theta=0.5 * (vp - 1.5) / scale)
    dt = float(model.critical_dt)
    geom = AcquisitionGeometry(model, rec, src[0], 0., 4500., f0=0.007, src_type='Ricker')
    nt = geom.nt
    wavelet = np.ascontiguousarray(geom.src.data)
    plan = mifwi.Plan(model.grid.shape, model.spacing, model.grid.origin, space_order, nt, dt, 1, rec.shape[0],
                      max_shots=nshots, device=device)
    plan.set_model(model.vp.data, model.damp.data)
    plan.tti_set_model(model.epsilon.data, model.delta.data, model.theta.data)
    plan.set_geometry(np.asarray(src[:nshots], np.float32).reshape(nshots, 1, 2), rec)
    ngrid = float(np.prod(model.grid.shape))
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        d = plan.tti_forward(wavelet, save=True)
        fwd = plan.stats()["fwd_ms"]
        plan.tti_gradient(d)
        bwd = plan.stats()["bwd_ms"]
        wall = (time.perf_counter() - t0) * 1e3
        if best is None or wall < best[0]:
            best = (wall, fwd, bwd)
    plan.close()
    pts = 2.0 * (nt - 2) * ngrid * nshots
    return {"workload": "2-D TTI ('centered' kernel) on the Marmousi-II grid %s, space_order=%d, dt=%.3f ms, nt=%d, "
                        "%d shots: forward with both fields saved + Jacobian-adjoint gradient" %
                        (list(model.grid.shape), space_order, dt, nt, nshots),
            "gpoints_per_s": pts / (best[0] * 1e-3) / 1e9, "wall_ms": best[0], "forward_loop_ms": best[1],
            "gradient_loop_ms": best[2], "forward_us_per_step": best[1] / (nt - 2) * 1e3,
            "gradient_us_per_step": best[2] / (nt - 2) * 1e3,
            "algorithmic_bytes_per_point": {"forward": 64, "gradient": 80}}


def c4_leg(mifwi, device, space_order=8, n=512, nbl=40, nt=500):
    """Synthetic 3-D 512^3 (+ sponge = 592^3), space order 8, one shot, nt = 500: forward + L2 gradient through
    the C ABI, inputs and outputs resident on the device.  500 float32 levels (460 GB with halos) do not fit one
    GPU: the library stores the levels of as many of the last steps as fit and checkpoints + recomputes the first
    ones.  The first evaluation (which allocates and clears that memory) is reported apart from the repeated one."""
    import torch
    from seismic import Model
    shape = (n, n, n)
    vp = np.empty(shape, dtype=np.float32)
    vp[:] = np.linspace(1.5, 4.5, n, dtype=np.float32)[None, None, :]
    kw = dict(origin=(0., 0., 0.), spacing=(10., 10., 10.), shape=shape, space_order=space_order, nbl=nbl, bcs="damp")
    model, model0 = Model(vp=vp, **kw), Model(vp=np.full(shape, 3.0, dtype=np.float32), **kw)
    dt = float(model.critical_dt)
    sp = model.grid.shape
    src = np.array([[n * 5.0, n * 5.0, 20.0]], dtype=np.float32)
    g = np.linspace(0, (n - 1) * 10.0, 64, dtype=np.float32)
    rec = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    rec = np.concatenate([rec, np.full((rec.shape[0], 1), 20.0, np.float32)], 1)
    t = np.arange(nt) * dt
    r = np.pi * 0.010 * (t - 100.0)
    dev = torch.device("cuda", device)
    wav = torch.from_numpy(((1 - 2 * r ** 2) * np.exp(-r ** 2)).astype(np.float32).reshape(nt, 1)).to(dev)
    obs = torch.empty((1, nt, rec.shape[0]), dtype=torch.float32, device=dev)
    fval = torch.zeros(1, dtype=torch.float64, device=dev)
    pts = 2.0 * (nt - 2) * float(np.prod(sp))
    out, grad_f32 = None, None
    # the float32 default first; then the same evaluation with the saved levels in half precision
    # (MIFWI_FLAG_SAVE_F16: every level fits, nothing is propagated twice; the gradient differs in the 5th digit)
    # and in between the float32 evaluation with every point of every step computed (MIFWI_3D_BOX=0): by default the
    # loops compute nothing where the wave has not arrived yet, which is exactly zero
    box_default = os.environ.get("MIFWI_3D_BOX", "1")   # "0" from outside (profiling the full sweep): no second float32 run
    variants = [(0, box_default)] + ([(0, "0")] if box_default != "0" else []) + [(mifwi.FLAG_SAVE_F16, box_default)]
    for flags, box in variants:
        with _env("MIFWI_3D_BOX", box):
            plan = mifwi.Plan(sp, model.spacing, model.grid.origin, space_order, nt, dt, 1, rec.shape[0], max_shots=1,
                              device=device, flags=flags)
        plan.set_geometry(src, rec)
        if flags == 0 and box == box_default:
            plan.set_model(model.vp.data, model.damp.data)
            plan.forward(wav, rec_out=obs)
        plan.set_model(model0.vp.data, model0.damp.data)
        grad = torch.empty((1,) + sp, dtype=torch.float32, device=dev)
        walls = []
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.l2_misfit_gradient(wav, obs, grad=grad, fval=fval, want_illum=False)
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
        st = plan.stats()
        objective = float(fval.item())
        plan.close()
        leg = {"gpoints_per_s": pts / walls[1] / 1e9,
               "gpoints_per_s_counts": "every grid point of every step, computed or known to be zero",
               "wall_ms": walls[1] * 1e3, "first_call_wall_ms": walls[0] * 1e3,
               "forward_loop_ms": st["fwd_ms"], "gradient_loop_ms_incl_recomputation": st["bwd_ms"],
               "checkpointed_steps": st["ckpt_steps"], "device_bytes": st["device_bytes"], "objective": objective}
        if flags == 0 and box == box_default:
            out = dict({"workload": "synthetic 3-D %d^3 (+nbl %d = %d^3), space_order=%d, nt=%d, 1 shot: forward + L2 "
                                    "gradient; the first %d steps checkpointed in segments of %d and propagated twice, "
                                    "the levels of the other %d stored; %s" %
                                    (n, nbl, sp[0], space_order, nt, st["ckpt_steps"], st["ckpt_segment"],
                                     nt - 2 - st["ckpt_steps"],
                                     "every point of every step computed (MIFWI_3D_BOX=0)" if box_default == "0" else
                                     "time loops compute nothing where the wave has not arrived (exact zeros)")}, **leg)
            grad_f32 = grad
        elif flags == 0:
            leg["gradient_equal_to_default"] = bool(torch.equal(grad, grad_f32))
            leg["what"] = "same evaluation with every point of every step computed (MIFWI_3D_BOX=0)"
            out["every_plane_computed"] = leg
        else:
            num = float(torch.linalg.vector_norm((grad - grad_f32).double()))
            den = float(torch.linalg.vector_norm(grad_f32.double()))
            leg["gradient_rel_l2_vs_float32_levels"] = num / den
            leg["what"] = "same evaluation with MIFWI_FLAG_SAVE_F16 (opt-in): saved levels in half precision, all stored"
            out["saved_levels_f16"] = leg
        del grad
        torch.cuda.empty_cache()
    del grad_f32, obs
    torch.cuda.empty_cache()
    return out


def fwi_iteration(true_model, init_model, src_all, rec, k=3, reps=3):
    """Wall time of one FWI iteration through the drop-in Python API (fwi.fwi_loss): one
    objective+gradient evaluation followed by k forward-only line-search evaluations, all shots of
    the survey sharded over the ranks.  Called on every rank; returns the per-iteration time."""
    import torch
    from seismic import AcquisitionGeometry
    from fwi import fm_multi, fwi_loss
    from misfit import least_square
    from copy import deepcopy
    init_model = deepcopy(init_model)            # fwi_loss writes trial velocities into the model
    g_true = AcquisitionGeometry(true_model, rec, src_all, 0., 4500., f0=0.007, src_type='Ricker')
    g_init = AcquisitionGeometry(init_model, rec, src_all, 0., 4500., f0=0.007, src_type='Ricker')
    obs = fm_multi(g_true)
    mask = np.ones(init_model.shape, dtype=np.float32)
    mask[:, :15] = 0
    m0 = 1.0 / np.asarray(init_model.vp.data[init_model.nbl:-init_model.nbl,
                                             init_model.nbl:-init_model.nbl], dtype=np.float64).reshape(-1) ** 2
    fwi_loss(m0, g_init, obs, least_square, None, mask, True)                       # warm-up of both kinds of call
    fwi_loss(m0, g_init, obs, least_square, None, mask, True, calc_grad=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        f, g, _ = fwi_loss(m0, g_init, obs, least_square, None, mask, True)
        p = -g * (0.01 * np.abs(m0).max() / np.abs(g).max())
        for j in range(k):
            fwi_loss(m0 + (j + 1) * p, g_init, obs, least_square, None, mask, True, calc_grad=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, k, float(f)


class Survey(object):
    """Device-resident state of one rank's share of a Marmousi-II survey: plan, observed data, mutes, outputs.
    step() is one gradient evaluation of the L2 objective for these shots (+ all-reduce when distributed)."""

    def __init__(self, mifwi, torch, dev, local_rank, true_model, init_model, src, rec, so, flags, dist=None):
        from seismic import AcquisitionGeometry
        from fwi import _mute_factor
        self.torch, self.dist = torch, dist
        self.ns = ns = src.shape[0]
        geom = AcquisitionGeometry(true_model, rec, src[0], 0., 4500., f0=0.007, src_type='Ricker')
        self.nt, self.dt = geom.nt, float(true_model.critical_dt)
        self.wavelet = np.ascontiguousarray(geom.src.data)                       # (nt, 1) float32
        self.shape_pad = true_model.grid.shape
        self.ngrid = int(np.prod(self.shape_pad))
        self.nbl = true_model.nbl
        self.plan = mifwi.Plan(self.shape_pad, true_model.spacing, true_model.grid.origin, so, self.nt, self.dt, 1,
                               rec.shape[0], max_shots=ns, dtype=np.float32, device=local_rank, flags=flags)
        self.plan.set_geometry(src.reshape(ns, 1, 2), rec)
        # everything the timed region reads is resident in HBM before it starts
        d_vp_true = torch.from_numpy(np.ascontiguousarray(true_model.vp.data)).to(dev)
        d_vp_init = torch.from_numpy(np.ascontiguousarray(init_model.vp.data)).to(dev)
        d_damp = torch.from_numpy(np.ascontiguousarray(true_model.damp.data)).to(dev)
        self.d_wavelet = torch.from_numpy(self.wavelet).to(dev)
        self.d_obs = torch.empty((ns, self.nt, rec.shape[0]), dtype=torch.float32, device=dev)
        self.plan.set_model(d_vp_true, d_damp)
        self.plan.forward(self.d_wavelet, nshots=ns, rec_out=self.d_obs)        # observed data (untimed)
        self.plan.set_model(d_vp_init, d_damp)
        mute = np.stack([_mute_factor(true_model.shape, true_model.spacing, src[k], rec) for k in range(ns)])
        self.d_mute = torch.from_numpy(mute.astype(np.float32)).to(dev)
        self.d_grad = torch.zeros((ns,) + self.shape_pad, dtype=torch.float32, device=dev)
        self.d_illum = torch.zeros_like(self.d_grad)
        self.d_fval = torch.zeros(ns, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()

    def step(self):
        torch, nbl = self.torch, self.nbl
        self.plan.l2_misfit_gradient(self.d_wavelet, self.d_obs, nshots=self.ns, grad=self.d_grad, illum=self.d_illum,
                                     fval=self.d_fval, want_illum=True)
        st = self.plan.stats()
        g = (self.d_grad[:, nbl:-nbl, nbl:-nbl] * self.d_mute).sum(0)
        il = (self.d_illum[:, nbl:-nbl, nbl:-nbl] * self.d_mute).sum(0)
        packed = torch.cat([g.reshape(-1).double(), il.reshape(-1).double(), self.d_fval.sum().reshape(1)])
        if self.dist is not None:
            self.dist.all_reduce(packed)
        return packed, st

    def forward_only(self):
        """Objective without gradient (a line-search trial): forward modelling + residual + norm."""
        self.plan.l2_misfit_gradient(self.d_wavelet, self.d_obs, nshots=self.ns, calc_grad=False, fval=self.d_fval)
        f = self.d_fval.sum().reshape(1)
        if self.dist is not None:
            self.dist.all_reduce(f)
        return f

    def close(self):
        self.plan.close()


def shot_batches(mifwi, torch, dev, local_rank, true_model, init_model, rec, so, flags, sizes=(4, 32), reps=3):
    """The gradient evaluation of Survey.step for other batch sizes on this one GPU (shots spread over the line as in
    a 32-shot survey).  BASELINE config 3 runs 32 shots on 8 GPUs, 4 each: time(32 shots) / time(4 shots) is the
    speed-up that sharding can reach before the all-reduce (one message of 2 x 47 600 doubles + 1) is paid."""
    out = {}
    src32, _ = acquisition(true_model, 32)
    for ns in sizes:
        pick = src32[:: 32 // ns][:ns]
        sv = Survey(mifwi, torch, dev, local_rank, true_model, init_model, pick, rec, so, flags)
        sv.step()
        torch.cuda.synchronize()
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            _, st = sv.step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            if best is None or ms < best[0]:
                best = (ms, st["fwd_ms"], st["bwd_ms"])
        steps = sv.nt - 2
        out["%d_shots" % ns] = {"gradient_eval_ms": best[0], "forward_us_per_step": best[1] / steps * 1e3,
                                "gradient_us_per_step": best[2] / steps * 1e3,
                                "gpoints_per_s": 2.0 * steps * sv.ngrid * ns / best[0] / 1e6}
        sv.close()
    if "4_shots" in out and "32_shots" in out:
        out["speedup_32_shots_on_1_gpu_over_4_shots_per_gpu"] = (out["32_shots"]["gradient_eval_ms"] /
                                                                 out["4_shots"]["gradient_eval_ms"])
    return out


def strong_scaling(mifwi, torch, dist, use_dist, dev, local_rank, rank, world, true_model, init_model, rec, so, flags,
                   totals=(32, 8), reps=3):
    """Fixed surveys sharded over the ranks (strong scaling), timed like the headline: barrier + synchronize on both
    sides, maximum over ranks.  32 shots = BASELINE config 3 (32 / N per GPU), 8 shots = the survey of the BASELINE
    metric's wording (8 / N per GPU).  Run at every N, so that the driver's runs at N = 1, 2, 4, 8 give the curve
    time(N = 1) / time(N): the shot-parallel gradient speed-up."""
    out = {}

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for total in totals:
        key = "%d_shot_survey" % total
        if total % world:
            out[key] = {"skipped": "%d shots do not divide over %d GPUs" % (total, world)}
            continue
        per = total // world
        src_all, _ = acquisition(true_model, total)
        sv = Survey(mifwi, torch, dev, local_rank, true_model, init_model, src_all[rank * per:(rank + 1) * per], rec, so,
                    flags, dist if use_dist else None)
        sv.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            _, st = sv.step()
        barrier()
        el = (time.perf_counter() - t0) / reps
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        steps = sv.nt - 2
        out[key] = {"shots_total": total, "shots_per_gpu": per, "n_gpus": world, "gradient_eval_ms": el * 1e3,
                    "gpoints_per_s": 2.0 * steps * sv.ngrid * total / el / 1e9,
                    "forward_us_per_step": st["fwd_ms"] / steps * 1e3, "gradient_us_per_step": st["bwd_ms"] / steps * 1e3,
                    "fwd_kernel": st["fwd_kernel"], "bwd_kernel": st["bwd_kernel"],
                    "persist_fallbacks": st["persist_fallbacks"]}
        sv.close()
    out["what"] = ("gradient evaluation (forward with saved wavefield + gradient + muting + shot sum + all-reduce) of a "
                   "FIXED survey split evenly over the GPUs; speed-up at N GPUs = gradient_eval_ms at n_gpus 1 / "
                   "gradient_eval_ms at n_gpus N")
    return out


def main():
    # The result line is the only thing this program writes to its standard output: libraries that print banners
    # there (RCCL prints its version when a communicator is created) are pointed at standard error instead.
    result_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--space-order", type=int, default=16)
    ap.add_argument("--shots-total", type=int, default=0,
                    help="strong scaling: a fixed survey of this many shots split evenly over the GPUs (BASELINE config 3: "
                         "32); default 0 = weak scaling, 8 shots per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-3d", action="store_true", help="skip the 3-D measurements (stencil roofline, 512^3 gradient)")
    ap.add_argument("--no-tti", action="store_true", help="skip the 2-D TTI measurement")
    ap.add_argument("--no-batches", action="store_true", help="skip the 4- and 32-shot batch timings")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling legs (32- and 8-shot surveys sharded over the GPUs)")
    ap.add_argument("--flags", type=int, default=0, help="MIFWI_FLAG_* bits for the plan (debugging)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import mifwi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    # BENCH_REHEARSAL=1: every rank uses GPU 0 and the collectives run over gloo, so that the N > 1
    # code path can be exercised on a one-GPU machine (timings of such a run mean nothing)
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # BENCH_FORCE_DIST=1: a one-rank job under torch.distributed.run still goes through RCCL
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    so = args.space_order
    true_model, init_model = marmousi2(so)
    strong = args.shots_total > 0
    if strong and args.shots_total % world:
        raise SystemExit("--shots-total %d is not a multiple of %d GPUs" % (args.shots_total, world))
    per_gpu = args.shots_total // world if strong else SHOTS_PER_GPU
    nshots_total = per_gpu * world
    src_all, rec = acquisition(true_model, nshots_total)
    src = src_all[rank * per_gpu:(rank + 1) * per_gpu]

    coll = dist if use_dist else None
    sv = Survey(mifwi, torch, dev, local_rank, true_model, init_model, src, rec, so, args.flags, coll)
    nt, ngrid, shape_pad = sv.nt, sv.ngrid, sv.shape_pad

    fwd_stats, grad_stats = [], []

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist:
        # communicator set-up is lazy (first collective); keep it out of the timed region even with --warmup 0
        dist.all_reduce(torch.zeros(1, dtype=torch.float64, device=dev))
    for _ in range(args.warmup):
        sv.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        packed, st = sv.step()
        fwd_stats.append((st["fwd_ms"], st["fwd_launches"]))
        grad_stats.append((st["bwd_ms"], st["bwd_launches"]))
        if os.environ.get("BENCH_VERBOSE"):
            print("step: fwd %.3f ms  bwd %.3f ms" % (st["fwd_ms"], st["bwd_ms"]), file=sys.stderr)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # forward-only objective evaluations (line-search trials), same protocol
    sv.forward_only()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sv.forward_only()
    barrier()
    fwd_only_ms = (time.perf_counter() - t0) / args.steps * 1e3

    if st["persist_fallbacks"] != 0:
        # a whole-loop launch that was abandoned and run again a kernel per step would make the line describe a
        # mixture of two schedules
        print("warning: %d whole-loop launches were repeated with a kernel per step" % st["persist_fallbacks"], file=sys.stderr)
    pts_per_step = 2.0 * (nt - 2) * ngrid * per_gpu * world
    value = pts_per_step * args.steps / elapsed / 1e9
    fval = float(packed[-1].item())

    # roofline of the dominant kernel (gradient time step): algorithmic 32 B per point update,
    # points per launch = padded grid x shots in the launch; duration = HIP-event time of the
    # backward time loop on the plan's stream / launches of that loop.
    bwd_ms = float(np.mean([m for m, _ in grad_stats]))
    bwd_launches = grad_stats[-1][1]
    fwd_ms = float(np.mean([m for m, _ in fwd_stats]))
    fwd_launches = fwd_stats[-1][1]
    steps_per_loop = nt - 2
    g_bytes_loop = 32.0 * ngrid * per_gpu * steps_per_loop
    f_bytes_loop = 20.0 * ngrid * per_gpu * steps_per_loop
    g_ach = g_bytes_loop / (bwd_ms * 1e-3) / 1e9
    f_ach = f_bytes_loop / (fwd_ms * 1e-3) / 1e9
    prof_us = profile_launch_us(st["bwd_kernel"])
    roofline = {"bound": "hbm", "achieved": g_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": g_ach / HBM_PEAK_GBS, "traffic": pmc_traffic(st["bwd_kernel"]),
                "kernel": st["bwd_kernel"], "launches_per_loop": bwd_launches,
                "avg_launch_us": bwd_ms * 1e3 / max(1, bwd_launches),
                "bytes_per_launch": g_bytes_loop / max(1, bwd_launches),
                "timing": "HIP events around the backward time loop on the plan's stream / launches "
                          "(includes inter-kernel gaps)",
                "profile_avg_launch_us": prof_us,
                "profile_frac": (g_bytes_loop / max(1, bwd_launches)) / (prof_us * 1e-6) / 1e9 / HBM_PEAK_GBS
                if prof_us else None,
                "profile": "%s: rocprofv3 --kernel-trace of this command with --no-batches, mean over the backward loop's "
                           "launches (plain and two-step gradient launches weighted by count)" %
                           _profile_entry(st["bwd_kernel"]).get("source"),
                "profile_note": "the tracer slows these ~5 us launches: HIP events around the same loop read 8.7 us "
                                "per launch under rocprofv3 and 5.6 us without it (profiles/r2_01_bench_line_under_trace.json "
                                "vs profiles/r2_01_bench_line_untraced.json), so the traced kernel-only mean exceeds the "
                                "untraced launch-to-launch mean; the 0.9 ms 3-D launch agrees within 1 %",
                "schedule": "the loop alternates a plain step (step2d_kernel mode 0) and a step that adds the "
                            "gradient terms of both (mode 4); launch time, bytes and traffic are means over the loop"}
    whole_loop = st["bwd_kernel"].startswith("persist2d")
    if whole_loop:
        # One launch runs the whole backward loop; the model, the two newest levels and the gradient sums of a tile
        # stay in registers / LDS, so the bytes a kernel per step must move are not moved and HBM is not what
        # binds.  What binds is the chain of trips to the XCD's L2 that every step makes to exchange halos with the
        # neighbouring tiles.  The roofline is therefore a latency one: time per step against the measured cost of
        # that exchange alone.
        floor_us, floor_src = exchange_floor_us()
        step_us = bwd_ms * 1e3 / steps_per_loop
        tr = _profile_entry(st["bwd_kernel"])
        roofline = {
            "bound": "xcd_l2_exchange_latency",
            "achieved": 1.0 / step_us, "peak": (1.0 / floor_us) if floor_us else None, "unit": "time steps/us",
            "frac": (floor_us / step_us) if floor_us else None,
            "traffic": tr.get("bytes"),
            "kernel": st["bwd_kernel"], "launches_per_loop": bwd_launches, "steps_per_launch": steps_per_loop,
            "us_per_step": step_us, "exchange_floor_us_per_step": floor_us, "exchange_floor_source": floor_src,
            "avg_launch_us": bwd_ms * 1e3 / max(1, bwd_launches),
            "timing": "HIP events around the backward time loop on the plan's stream (one launch runs the whole loop)",
            "profile_avg_launch_us": prof_us,
            "profile": "%s: rocprofv3 --kernel-trace of this command with --no-batches, mean over the whole-loop "
                       "launches of the gradient variant" % tr.get("source"),
            "schedule": "one launch per loop: every tile keeps a block resident and synchronises with its neighbours "
                        "through per-tile counters in the L2 of the XCD that holds the shot",
            "hbm": {"bytes_counter": tr.get("bytes"),
                    "achieved_GBps": (tr["bytes"] / (bwd_ms * 1e-3) / 1e9) if tr.get("bytes") else None,
                    "frac": (tr["bytes"] / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr.get("bytes") else None,
                    "peak_GBps": HBM_PEAK_GBS,
                    "what": "fabric-side bytes of one launch from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE) over "
                            "the launch time measured here: the saved forward field read once"},
            "algorithmic_equivalent_GBps": g_ach,
            "algorithmic_equivalent_note": "32 B per point update (what a kernel per step has to move) x points x steps "
                                           "/ loop time; not a fraction of anything this kernel is bound by"}
    roofline_fwd = {"bound": "hbm", "achieved": f_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": f_ach / HBM_PEAK_GBS, "traffic": pmc_traffic(st["fwd_kernel"]), "kernel": st["fwd_kernel"],
                    "launches_per_loop": fwd_launches, "avg_launch_us": fwd_ms * 1e3 / max(1, fwd_launches),
                    "profile_avg_launch_us": profile_launch_us(st["fwd_kernel"])}

    if st["fwd_kernel"].startswith("persist2d"):
        roofline_fwd.update({"steps_per_launch": steps_per_loop, "us_per_step": fwd_ms * 1e3 / steps_per_loop})
    out = {
        "metric": "Gpoints/s (fwd+adj), Marmousi-II, %s" % ("%d-shot survey sharded over the GPUs" % nshots_total if strong
                                                           else "8 shots per GPU"),
        "value": value, "unit": "Gpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f32",
        "data": "Marmousi-II vp rasters of the reference (model_data/SMARM2); observed data synthesised by "
                "forward modelling the true model; start model = smooth_20",
        "config": {"workload": "Marmousi-II (SMARM2 340x140, nbl 40 -> 420x220) acoustic, space_order=%d, "
                               "dt=3.0 ms, nt=%d, %d shots per GPU%s, L2 misfit gradient (forward with saved "
                               "wavefield + adjoint/gradient + per-shot muting + shot sum%s)"
                               % (so, nt, per_gpu, " (BASELINE config 3: %d shots in all)" % nshots_total if strong else "",
                                  " + RCCL all-reduce" if world > 1 else ""),
                   "shots_total": nshots_total, "grid_padded": list(shape_pad), "nt": nt, "space_order": so,
                   "parallelism": "shots sharded %d per GPU" % per_gpu},
        "gradient_eval_ms": elapsed / args.steps * 1e3,
        "forward_only_eval_ms": fwd_only_ms,
        "kernels": {"forward_loop": st["fwd_kernel"], "backward_loop": st["bwd_kernel"],
                    "whole_loop_launches_repeated_kernel_per_step": st["persist_fallbacks"]},
        "objective": fval,
        "roofline": roofline, "roofline_forward": roofline_fwd,
    }
    sv.close()
    it_ms, it_k, it_f = fwi_iteration(true_model, init_model, src_all, rec)
    if use_dist:
        t = torch.tensor([it_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        it_ms = float(t.item())
    out["fwi_iteration"] = {"ms": it_ms, "gradient_evaluations": 1, "forward_only_evaluations": it_k,
                            "shots": nshots_total, "objective": it_f,
                            "expected_from_parts_ms": out["gradient_eval_ms"] + it_k * fwd_only_ms,
                            "what": "fwi.fwi_loss through the Python API: model update, batched shots, "
                                    "muting, preconditioner, mask, all-reduce"}
    if not args.no_strong:
        # every rank takes part (the surveys are sharded over the ranks); rank 0 keeps the result
        out["strong_scaling"] = strong_scaling(mifwi, torch, dist, use_dist, dev, local_rank, rank, world, true_model,
                                               init_model, rec, so, args.flags)
    if rank == 0 and world == 1 and not args.no_batches and not strong:
        out["shot_batches"] = shot_batches(mifwi, torch, dev, local_rank, true_model, init_model, rec, so, args.flags)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(true_model, init_model, src_all, rec, sv.wavelet, sv.dt, so)
    if rank == 0 and world == 1 and not args.no_3d:
        # the north-star stencil: 16th-order 3-D forward step against the HBM roofline
        out["roofline_3d"] = roofline_3d(mifwi, local_rank, space_order=16)
        # the same step at the space order of config_3d_512 (the counters were collected at order 16 only)
        o8 = roofline_3d(mifwi, local_rank, space_order=8)
        out["roofline_3d"]["space_order_8"] = {k: o8[k] for k in ("achieved", "frac", "avg_launch_us", "gpoints_per_s", "workload")}
        out["config_3d_512"] = c4_leg(mifwi, local_rank)
    if rank == 0 and world == 1 and not args.no_tti:
        out["config_tti_2d"] = tti_leg(mifwi, local_rank, true_model, src_all, rec)
    if rank == 0:
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
