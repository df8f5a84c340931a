"""GPU probe: what the bookkeeping of the box of nonzero points costs once the wave fills the grid: forward loop of
nt steps on a small cube (default 176^3 + 2*40), with and without it.  Usage: probe_box_overhead.py [n=176] [nt=900]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
import mifwi
from seismic import Model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 176
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 900
so = 8
vp = np.full((n, n, n), 3.0, dtype=np.float32)
model = Model(origin=(0., 0., 0.), spacing=(10., 10., 10.), shape=vp.shape, space_order=so, vp=vp, nbl=40, bcs="damp")
dt = float(model.critical_dt)
sp = model.grid.shape
src = np.array([[n * 5.0, n * 5.0, 20.0]], dtype=np.float32)
rec = np.array([[100.0, 100.0, 20.0]], dtype=np.float32)
t = np.arange(nt) * dt
r = np.pi * 0.010 * (t - 100.0)
w = ((1 - 2 * r ** 2) * np.exp(-r ** 2)).astype(np.float32).reshape(nt, 1)
for box in ("0", "1"):
    os.environ["MIFWI_3D_BOX"] = box
    plan = mifwi.Plan(sp, model.spacing, model.grid.origin, so, nt, dt, 1, 1, max_shots=1)
    plan.set_model(model.vp.data, model.damp.data)
    plan.set_geometry(src, rec)
    best = 1e9
    for _ in range(3):
        d = plan.forward(w)
        best = min(best, plan.stats()["fwd_ms"])
    u = plan.read_wavefield(1, shot=0)[(nt - 1) % 3]
    print("grid %s nt=%d MIFWI_3D_BOX=%s: forward loop %.2f ms (%.4f ms per step); planes with a nonzero value at the end: %d of %d; |rec| %.8e" %
          (sp, nt, box, best, best / (nt - 2), int((np.abs(u).max(axis=(1, 2)) > 0).sum()), sp[0], float(np.linalg.norm(d.astype(np.float64)))), flush=True)
    plan.close()
