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

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (the gradient time step);
`cpu_baseline` is the CPU oracle (a restatement of the Devito CPU path, not Devito itself) timed
on the host cores for a bounded sample of the same workload.
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
    command (profiles/r1_traffic.json), or None when no such measurement is recorded."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_traffic.json")) as f:
            return json.load(f).get(kernel, {}).get("bytes")
    except (OSError, ValueError):
        return None


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


def cpu_baseline(true_model, init_model, src, rec, wavelet, dt, space_order, nshots=2):
    """Oracle (CPU restatement of the Devito path) on `nshots` shots of the same workload."""
    from oracle import oracle as orc
    nthreads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    nbl = true_model.nbl
    p_true = orc.Problem(true_model.vp.data[nbl:-nbl, nbl:-nbl], true_model.spacing, nbl, space_order, dt=dt)
    p_init = orc.Problem(init_model.vp.data[nbl:-nbl, nbl:-nbl], true_model.spacing, nbl, space_order, dt=dt)
    obs = [p_true.forward(src[i], wavelet, rec, dt=dt)[0] for i in range(nshots)]
    t0 = time.perf_counter()
    pts, reps = 0, 0
    while reps < 1 or (time.perf_counter() - t0 < 10.0 and reps < 50):     # about 10 s of CPU work
        for i in range(nshots):
            syn, u = p_init.forward(src[i], wavelet, rec, dt=dt, save=True)
            p_init.gradient(rec, syn - obs[i], u, dt=dt)
            pts += 2 * (wavelet.shape[0] - 2) * int(np.prod(p_init.shape_pad))
        reps += 1
    el = time.perf_counter() - t0
    return {"value": pts / el / 1e9, "unit": "Gpoints/s", "cores": nthreads, "kind": "port",
            "sample": "%d shots of the same Marmousi-II workload (forward with saved wavefield + gradient, "
                      "full nt=%d), %d passes, shots serial, OpenMP over the grid; %.1f s"
                      % (nshots, wavelet.shape[0], reps, el)}


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
    the C ABI.  500 float32 levels (415 GB) do not fit one GPU, so the default path checkpoints and recomputes."""
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
    plan = mifwi.Plan(sp, model.spacing, model.grid.origin, space_order, nt, dt, 1, rec.shape[0], max_shots=1,
                      device=device)
    plan.set_geometry(src, rec)
    t = np.arange(nt) * dt
    r = np.pi * 0.010 * (t - 100.0)
    wav = ((1 - 2 * r ** 2) * np.exp(-r ** 2)).astype(np.float32).reshape(nt, 1)
    plan.set_model(model.vp.data, model.damp.data)
    obs = plan.forward(wav)
    plan.set_model(model0.vp.data, model0.damp.data)
    t0 = time.perf_counter()
    out = plan.l2_misfit_gradient(wav, obs, want_illum=False)
    wall = time.perf_counter() - t0
    st = plan.stats()
    plan.close()
    pts = 2.0 * (nt - 2) * float(np.prod(sp))
    return {"workload": "synthetic 3-D %d^3 (+nbl %d = %d^3), space_order=%d, nt=%d, 1 shot: forward + L2 gradient, "
                        "two-level checkpointing (the saved wavefield would be 415 GB)" % (n, nbl, sp[0], space_order, nt),
            "gpoints_per_s": pts / wall / 1e9, "wall_ms": wall * 1e3, "forward_loop_ms": st["fwd_ms"],
            "gradient_loop_ms_incl_recomputation": st["bwd_ms"], "device_bytes": st["device_bytes"],
            "objective": float(out["fval"][0])}


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
    fwi_loss(m0, g_init, obs, least_square, None, mask, True)                       # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        f, g, _ = fwi_loss(m0, g_init, obs, least_square, None, mask, True)
        p = -g * (0.01 * np.abs(m0).max() / np.abs(g).max())
        for j in range(k):
            fwi_loss(m0 + (j + 1) * p, g_init, obs, least_square, None, mask, True, calc_grad=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, k, float(f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--space-order", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-3d", action="store_true", help="skip the 3-D measurements (stencil roofline, 512^3 gradient)")
    ap.add_argument("--no-tti", action="store_true", help="skip the 2-D TTI measurement")
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
    nshots_total = SHOTS_PER_GPU * world
    src_all, rec = acquisition(true_model, nshots_total)
    mine = list(range(rank * SHOTS_PER_GPU, (rank + 1) * SHOTS_PER_GPU))
    src = src_all[mine]

    from seismic import AcquisitionGeometry
    geom = AcquisitionGeometry(true_model, rec, src[0], 0., 4500., f0=0.007, src_type='Ricker')
    nt, dt = geom.nt, float(true_model.critical_dt)
    wavelet = np.ascontiguousarray(geom.src.data)                       # (nt, 1) float32
    shape_pad = true_model.grid.shape
    ngrid = int(np.prod(shape_pad))
    nbl = true_model.nbl

    plan = mifwi.Plan(shape_pad, true_model.spacing, true_model.grid.origin, so, nt, dt, 1, rec.shape[0],
                      max_shots=SHOTS_PER_GPU, dtype=np.float32, device=local_rank, flags=args.flags)
    plan.set_geometry(src.reshape(SHOTS_PER_GPU, 1, 2), rec)

    # everything the timed region reads is resident in HBM before it starts
    d_vp_true = torch.from_numpy(np.ascontiguousarray(true_model.vp.data)).to(dev)
    d_vp_init = torch.from_numpy(np.ascontiguousarray(init_model.vp.data)).to(dev)
    d_damp = torch.from_numpy(np.ascontiguousarray(true_model.damp.data)).to(dev)
    d_wavelet = torch.from_numpy(wavelet).to(dev)
    d_obs = torch.empty((SHOTS_PER_GPU, nt, rec.shape[0]), dtype=torch.float32, device=dev)
    plan.set_model(d_vp_true, d_damp)
    plan.forward(d_wavelet, nshots=SHOTS_PER_GPU, rec_out=d_obs)        # observed data (untimed)
    plan.set_model(d_vp_init, d_damp)

    from fwi import _mute_factor
    mute = np.stack([_mute_factor(true_model.shape, true_model.spacing, src[k], rec) for k in range(SHOTS_PER_GPU)])
    d_mute = torch.from_numpy(mute.astype(np.float32)).to(dev)
    d_grad = torch.zeros((SHOTS_PER_GPU,) + shape_pad, dtype=torch.float32, device=dev)
    d_illum = torch.zeros_like(d_grad)
    d_fval = torch.zeros(SHOTS_PER_GPU, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    fwd_stats, grad_stats = [], []

    def step():
        plan.l2_misfit_gradient(d_wavelet, d_obs, nshots=SHOTS_PER_GPU, grad=d_grad, illum=d_illum, fval=d_fval,
                                want_illum=True)
        st = plan.stats()
        g = (d_grad[:, nbl:-nbl, nbl:-nbl] * d_mute).sum(0)
        il = (d_illum[:, nbl:-nbl, nbl:-nbl] * d_mute).sum(0)
        packed = torch.cat([g.reshape(-1).double(), il.reshape(-1).double(), d_fval.sum().reshape(1)])
        if use_dist:
            dist.all_reduce(packed)
        return packed, st

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist:
        # communicator set-up is lazy (first collective); keep it out of the timed region even with --warmup 0
        dist.all_reduce(torch.zeros(1, dtype=torch.float64, device=dev))
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        packed, st = step()
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

    pts_per_step = 2.0 * (nt - 2) * ngrid * SHOTS_PER_GPU * world
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
    g_bytes_loop = 32.0 * ngrid * SHOTS_PER_GPU * steps_per_loop
    f_bytes_loop = 20.0 * ngrid * SHOTS_PER_GPU * steps_per_loop
    g_ach = g_bytes_loop / (bwd_ms * 1e-3) / 1e9
    f_ach = f_bytes_loop / (fwd_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": g_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": g_ach / HBM_PEAK_GBS, "traffic": pmc_traffic(st["bwd_kernel"]),
                "kernel": st["bwd_kernel"], "launches_per_loop": bwd_launches,
                "avg_launch_us": bwd_ms * 1e3 / max(1, bwd_launches),
                "bytes_per_launch": g_bytes_loop / max(1, bwd_launches),
                "timing": "HIP events around the backward time loop on the plan's stream / launches "
                          "(includes inter-kernel gaps)",
                "schedule": "the loop alternates a plain step (step2d_kernel mode 0) and a step that adds the "
                            "gradient terms of both (mode 4); launch time, bytes and traffic are means over the loop"}
    roofline_fwd = {"bound": "hbm", "achieved": f_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": f_ach / HBM_PEAK_GBS, "traffic": pmc_traffic(st["fwd_kernel"]), "kernel": st["fwd_kernel"],
                    "launches_per_loop": fwd_launches, "avg_launch_us": fwd_ms * 1e3 / max(1, fwd_launches)}

    out = {
        "metric": "Gpoints/s (fwd+adj), Marmousi-II, 8 shots per GPU",
        "value": value, "unit": "Gpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32",
        "data": "Marmousi-II vp rasters of the reference (model_data/SMARM2); observed data synthesised by "
                "forward modelling the true model; start model = smooth_20",
        "config": {"workload": "Marmousi-II (SMARM2 340x140, nbl 40 -> 420x220) acoustic, space_order=%d, "
                               "dt=3.0 ms, nt=%d, %d shots per GPU, L2 misfit gradient (forward with saved "
                               "wavefield + adjoint/gradient + per-shot muting + shot sum%s)"
                               % (so, nt, SHOTS_PER_GPU, " + RCCL all-reduce" if world > 1 else ""),
                   "shots_total": nshots_total, "grid_padded": list(shape_pad), "nt": nt, "space_order": so,
                   "parallelism": "shots sharded %d per GPU" % SHOTS_PER_GPU},
        "gradient_eval_ms": elapsed / args.steps * 1e3,
        "objective": fval,
        "roofline": roofline, "roofline_forward": roofline_fwd,
    }
    plan.close()
    it_ms, it_k, it_f = fwi_iteration(true_model, init_model, src_all, rec)
    if use_dist:
        t = torch.tensor([it_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        it_ms = float(t.item())
    out["fwi_iteration"] = {"ms": it_ms, "gradient_evaluations": 1, "forward_only_evaluations": it_k,
                            "shots": nshots_total, "objective": it_f,
                            "what": "fwi.fwi_loss through the Python API: model update, batched shots, "
                                    "muting, preconditioner, mask, all-reduce"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(true_model, init_model, src_all, rec, wavelet, dt, so)
    if rank == 0 and world == 1 and not args.no_3d:
        # the north-star stencil: 16th-order 3-D forward step against the HBM roofline
        out["roofline_3d"] = roofline_3d(mifwi, local_rank, space_order=16)
        out["config_3d_512"] = c4_leg(mifwi, local_rank)
    if rank == 0 and world == 1 and not args.no_tti:
        out["config_tti_2d"] = tti_leg(mifwi, local_rank, true_model, src_all, rec)
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
