"""3-D TTI per-step timing on a layered cube (default 256^3 + 40 = 336^3): forward, adjoint, gradient.
Usage: probe_tti3d.py [n=256] [space_order=8]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
import mifwi
from seismic import Model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
so = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nt = 14
vp = np.empty((n, n, n), dtype=np.float32)
vp[:] = np.linspace(1.5, 3.5, n, dtype=np.float32)[None, None, :]
sc = 0.5 * (vp - 1.5)
model = Model(origin=(0., 0., 0.), spacing=(10., 10., 10.), shape=vp.shape, space_order=so, vp=vp, nbl=40, bcs="damp",
              epsilon=0.3 * sc, delta=0.2 * sc, theta=0.5 * sc, phi=0.3 * sc)
dt = float(model.critical_dt)
sp = model.grid.shape
src = np.array([[n * 5.0, n * 5.0, 20.0]], dtype=np.float32)
g = np.linspace(0, (n - 1) * 10.0, 32, dtype=np.float32)
rec = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
rec = np.concatenate([rec, np.full((rec.shape[0], 1), 20.0, np.float32)], 1)
plan = mifwi.Plan(sp, model.spacing, model.grid.origin, so, nt, dt, 1, rec.shape[0], max_shots=1)
plan.set_model(model.vp.data, model.damp.data)
plan.tti_set_model(model.epsilon.data, model.delta.data, model.theta.data, model.phi.data)
plan.set_geometry(src, rec)
t = np.arange(nt) * dt
r = np.pi * 0.010 * (t - 100.0)
w = ((1 - 2 * r ** 2) * np.exp(-r ** 2)).astype(np.float32).reshape(nt, 1)
best = [1e9] * 3
for _ in range(3):
    d = plan.tti_forward(w); f0 = plan.stats()["fwd_ms"]
    d = plan.tti_forward(w, save=True)
    plan.tti_gradient(d); gq = plan.stats()["bwd_ms"]
    plan.tti_adjoint(d); a = plan.stats()["bwd_ms"]
    best = [min(x, y) for x, y in zip(best, (f0, gq, a))]
print("tti3d %s so=%d rounds=%s: forward %.3f  gradient %.3f  adjoint %.3f ms per step  |rec| %.6e" %
      ((sp, so, os.environ.get("MIFWI_TTI3D_ROUNDS", "8")) + tuple(b / (nt - 2) for b in best) + (float(np.linalg.norm(d.astype(np.float64))),)), flush=True)
