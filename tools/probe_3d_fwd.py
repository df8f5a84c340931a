"""GPU probe: forward step of the 3-D kernel at 592^3 (kernel experiments).  Usage: probe_3d_fwd.py [space_order=16]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
os.environ.setdefault("MIFWI_3D_BOX", "0")   # steady-state step: every plane computed
import mifwi
from seismic import Model
n, so, nt = 512, (int(sys.argv[1]) if len(sys.argv) > 1 else 16), 14
nz = int(sys.argv[2]) if len(sys.argv) > 2 else n      # physical z extent (x and y stay at 512)
vp = np.empty((n, n, nz), dtype=np.float32)
vp[:] = np.linspace(1.5, 4.5, nz, dtype=np.float32)[None, None, :]
model = Model(origin=(0., 0., 0.), spacing=(10., 10., 10.), shape=vp.shape, space_order=so, vp=vp, nbl=40, bcs="damp")
dt = float(model.critical_dt)
sp = model.grid.shape
ngrid = float(np.prod(sp))
src = np.array([[n * 5.0, n * 5.0, 20.0]], dtype=np.float32)
g = np.linspace(0, (n - 1) * 10.0, 64, dtype=np.float32)
rec = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
rec = np.concatenate([rec, np.full((rec.shape[0], 1), 20.0, np.float32)], 1)
plan = mifwi.Plan(sp, model.spacing, model.grid.origin, so, nt, dt, 1, rec.shape[0], max_shots=1)
plan.set_model(model.vp.data, model.damp.data)
plan.set_geometry(src, rec)
t = np.arange(nt) * dt
r = np.pi * 0.010 * (t - 100.0)
wav = ((1 - 2 * r ** 2) * np.exp(-r ** 2)).astype(np.float32).reshape(nt, 1)
best = 1e9
for _ in range(4):
    d = plan.forward(wav, save=False)
    st = plan.stats()
    best = min(best, st["fwd_ms"] / st["fwd_launches"])
u = plan.read_wavefield(1, shot=0)
print("nz=%d " % sp[2] + "so=%d tile=%s chunks=%s: %.4f ms/step  %.1f Gpts/s  %.1f%% of 8 TB/s   |rec| %.8e |u| %.8e" %
      (so, os.environ.get("MIFWI_3D_TILE", "16x64"), os.environ.get("MIFWI_3D_NCHUNK", "auto"), best, ngrid / best / 1e6,
       20 * ngrid / best / 1e6 / 80, float(np.linalg.norm(d.astype(np.float64))), float(np.linalg.norm(u.astype(np.float64)))), flush=True)
