"""GPU probe: forward loop of the 2.5-D kernel at 592^3 with two steps per sweep (x slabs, second step one slab
behind the first) against the plain one-step-per-launch loop.  Usage: probe_3d_slab.py [space_order=16] [W,chunks ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
os.environ.setdefault("MIFWI_3D_BOX", "0")   # steady-state step: every plane computed
import mifwi
from seismic import Model
n, so, nt = 512, (int(sys.argv[1]) if len(sys.argv) > 1 else 16), 22
combos = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(0, 1), (16, 1), (24, 1), (32, 1), (32, 2), (48, 2),
                                                                       (64, 2), (64, 3), (96, 3), (148, 4)]
vp = np.empty((n, n, n), dtype=np.float32)
vp[:] = np.linspace(1.5, 4.5, n, dtype=np.float32)[None, None, :]
model = Model(origin=(0., 0., 0.), spacing=(10., 10., 10.), shape=vp.shape, space_order=so, vp=vp, nbl=40, bcs="damp")
dt = float(model.critical_dt)
sp = model.grid.shape
ngrid = float(np.prod(sp))
src = np.array([[n * 5.0, n * 5.0, 20.0]], dtype=np.float32)
g = np.linspace(0, (n - 1) * 10.0, 64, dtype=np.float32)
rec = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
rec = np.concatenate([rec, np.full((rec.shape[0], 1), 20.0, np.float32)], 1)
t = np.arange(nt) * dt
r = np.pi * 0.010 * (t - 100.0)
wav = ((1 - 2 * r ** 2) * np.exp(-r ** 2)).astype(np.float32).reshape(nt, 1)
import torch
d_vp = torch.from_numpy(np.ascontiguousarray(model.vp.data)).cuda()
d_damp = torch.from_numpy(np.ascontiguousarray(model.damp.data)).cuda()
ref = None
for W, ch in combos:
    os.environ["MIFWI_3D_SLAB"] = str(W)
    os.environ["MIFWI_3D_SLAB_CHUNKS"] = str(ch)
    plan = mifwi.Plan(sp, model.spacing, model.grid.origin, so, nt, dt, 1, rec.shape[0], max_shots=1)
    plan.set_model(d_vp, d_damp)
    plan.set_geometry(src, rec)
    best = 1e9
    for _ in range(3):
        d = plan.forward(wav, save=False)
        st = plan.stats()
        best = min(best, st["fwd_ms"] / (nt - 2))
    u = plan.read_wavefield(1, shot=0)
    same = ""
    if ref is None:
        ref = (d.copy(), u.copy())
    else:
        same = "  identical" if (np.array_equal(d, ref[0]) and np.array_equal(u, ref[1])) else "  DIFFERENT"
    print("so=%d slab W=%3d chunks=%d launches=%4d: %.4f ms/step  %.1f Gpts/s  %.1f%% of 8 TB/s  |u| %.8e%s" %
          (so, W, ch, st["fwd_launches"], best, ngrid / best / 1e6, 20 * ngrid / best / 1e6 / 80,
           float(np.linalg.norm(u.astype(np.float64))), same), flush=True)
    plan.close()
