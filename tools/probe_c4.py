"""GPU probe: the large 3-D benchmark configuration on one GPU -- synthetic 3-D 512^3 (+nbl 40), 1 shot, nt=500:
forward with (checkpointed) wavefield + L2 gradient through the C ABI."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
import mifwi
from seismic import Model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
so = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 500
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # e.g. 64 = half-precision saved levels
shape = (n, n, n)
rng = np.random.default_rng(1234)
vp = np.empty(shape, dtype=np.float32)
vp[:] = np.linspace(1.5, 4.5, n, dtype=np.float32)[None, None, :]
vp0 = np.full(shape, 3.0, dtype=np.float32)
kw = dict(origin=(0., 0., 0.), spacing=(10., 10., 10.), shape=shape, space_order=so, nbl=40, bcs="damp")
model, model0 = Model(vp=vp, **kw), Model(vp=vp0, **kw)
dt = float(model.critical_dt)
sp = model.grid.shape
ngrid = float(np.prod(sp))
src = np.array([[n * 5.0, n * 5.0, 20.0]], dtype=np.float32)
g = np.linspace(0, (n - 1) * 10.0, 64, dtype=np.float32)
rec = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
rec = np.concatenate([rec, np.full((rec.shape[0], 1), 20.0, np.float32)], 1)
plan = mifwi.Plan(sp, model.spacing, model.grid.origin, so, nt, dt, 1, rec.shape[0], max_shots=1, flags=flags)
plan.set_geometry(src, rec)
t = np.arange(nt) * dt
r = np.pi * 0.010 * (t - 100.0)
wav = ((1 - 2 * r ** 2) * np.exp(-r ** 2)).astype(np.float32).reshape(nt, 1)
plan.set_model(model.vp.data, model.damp.data)
obs = plan.forward(wav)
st = plan.stats()
print("forward: %.1f ms  %.1f Gpts/s" % (st["fwd_ms"], st["gpointss"]), flush=True)
plan.set_model(model0.vp.data, model0.damp.data)
t0 = time.perf_counter()
out = plan.l2_misfit_gradient(wav, obs, want_illum=False)
el = time.perf_counter() - t0
st = plan.stats()
pts = 2 * (nt - 2) * ngrid
print("C4 gradient evaluation (1 shot, %s, so=%d, nt=%d): wall %.2f s; fwd loop %.0f ms, bwd loop (incl. recomputation) %.0f ms"
      % (sp, so, nt, el, st["fwd_ms"], st["bwd_ms"]))
print("  -> %.1f Gpoints/s (fwd+adj points / wall), device memory %.1f GB, fval %.5e, |grad| %.4e, finite %s"
      % (pts / el / 1e9, st["device_bytes"] / 1e9, out["fval"][0], np.linalg.norm(out["grad"].astype(np.float64)),
         bool(np.isfinite(out["grad"]).all())))
