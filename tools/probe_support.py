"""GPU probe: how many x planes of the forward wavefield hold a nonzero value after n steps (BASELINE config 4's
set-up, 592^3, space order 8, constant 3 km/s).  Usage: probe_support.py [steps,...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
import mifwi
from seismic import Model
n, so = 512, 8
steps = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [30, 60, 120, 180, 224, 300]
vp = np.empty((n, n, n), dtype=np.float32)
vp[:] = np.linspace(1.5, 4.5, n, dtype=np.float32)[None, None, :]
model = Model(origin=(0., 0., 0.), spacing=(10., 10., 10.), shape=vp.shape, space_order=so, vp=vp, nbl=40, bcs="damp")
dt = float(model.critical_dt)
model0 = Model(origin=(0., 0., 0.), spacing=(10., 10., 10.), shape=vp.shape, space_order=so, vp=np.full(vp.shape, 3.0, np.float32), nbl=40, bcs="damp", dt=dt)
sp = model.grid.shape
src = np.array([[n * 5.0, n * 5.0, 20.0]], dtype=np.float32)
rec = np.array([[100.0, 100.0, 20.0]], dtype=np.float32)
for nt in steps:
    plan = mifwi.Plan(sp, model.spacing, model.grid.origin, so, nt, dt, 1, 1, max_shots=1)
    plan.set_model(model0.vp.data, model0.damp.data)
    plan.set_geometry(src, rec)
    t = np.arange(nt) * dt
    r = np.pi * 0.010 * (t - 100.0)
    w = ((1 - 2 * r ** 2) * np.exp(-r ** 2)).astype(np.float32).reshape(nt, 1)
    plan.forward(w)
    u = plan.read_wavefield(1, shot=0)[(nt - 1) % 3]
    a = np.abs(u)
    mx = float(a.max())
    out = []
    for thr in (0.0, 1e-30, 1e-12 * mx):
        nzx = int((a.max(axis=(1, 2)) > thr).sum()); nzy = int((a.max(axis=(0, 2)) > thr).sum()); nzz = int((a.max(axis=(0, 1)) > thr).sum())
        out.append("thr %.0e: x %d y %d z %d" % (thr, nzx, nzy, nzz))
    print("after %d steps (max %.2e): " % (nt - 1, mx) + "; ".join(out), flush=True)
    plan.close()
