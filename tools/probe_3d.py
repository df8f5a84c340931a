"""GPU probe: per-step time and algorithmic bandwidth of the 3-D forward/adjoint/gradient kernels on a
large synthetic grid (default 512^3 + nbl 40 = 592^3)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
os.environ.setdefault("MIFWI_3D_BOX", "0")   # steady-state step: every plane computed
import mifwi
from seismic import Model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
orders = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 16]
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nt = 14
shape = (n, n, n)
vp = np.empty(shape, dtype=np.float32)
vp[:] = np.linspace(1.5, 4.5, n, dtype=np.float32)[None, None, :]
for so in orders:
    model = Model(origin=(0., 0., 0.), spacing=(10., 10., 10.), shape=shape, space_order=so, vp=vp, nbl=40, bcs="damp")
    dt = float(model.critical_dt)
    sp = model.grid.shape
    ngrid = float(np.prod(sp))
    src = np.array([[n * 5.0, n * 5.0, 20.0]], dtype=np.float32)
    g = np.linspace(0, (n - 1) * 10.0, 64, dtype=np.float32)
    rec = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    rec = np.concatenate([rec, np.full((rec.shape[0], 1), 20.0, np.float32)], 1)
    plan = mifwi.Plan(sp, model.spacing, model.grid.origin, so, nt, dt, 1, rec.shape[0], max_shots=1, flags=flags)
    plan.set_model(model.vp.data, model.damp.data)
    plan.set_geometry(src, rec)
    t = np.arange(nt) * dt
    r = np.pi * 0.010 * (t - 100.0)
    wav = ((1 - 2 * r ** 2) * np.exp(-r ** 2)).astype(np.float32).reshape(nt, 1)
    best = 1e9
    for _ in range(3):
        recd = plan.forward(wav, save=False)
        st = plan.stats()
        best = min(best, st["fwd_ms"] / st["fwd_launches"])
    print("so=%2d grid %s fwd  %.3f ms/step  %.1f Gpts/s  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)  kind=%d" %
          (so, sp, best, ngrid / best / 1e6, 20 * ngrid / best / 1e6, 20 * ngrid / best / 1e6 / 80, st["kernel_kind"]),
          flush=True)
    best = 1e9
    for _ in range(3):
        plan.adjoint(recd)
        st = plan.stats()
        best = min(best, st["bwd_ms"] / st["bwd_launches"])
    print("so=%2d grid %s adj  %.3f ms/step  %.1f Gpts/s  %.0f GB/s algorithmic" %
          (so, sp, best, ngrid / best / 1e6, 20 * ngrid / best / 1e6), flush=True)
    if os.environ.get("PROBE_GRAD"):
        u = np.zeros((nt,) + tuple(sp), dtype=np.float32) if False else None
        plan.forward(wav, save=True, want_rec=False)
        best = 1e9
        for _ in range(3):
            plan.gradient(recd, want_illum=False)
            st = plan.stats()
            best = min(best, st["bwd_ms"] / st["bwd_launches"])
        print("so=%2d grid %s grad %.3f ms/step  %.1f Gpts/s  %.0f GB/s algorithmic (32 B/pt)" %
              (so, sp, best, ngrid / best / 1e6, 32 * ngrid / best / 1e6), flush=True)
    plan.close()
