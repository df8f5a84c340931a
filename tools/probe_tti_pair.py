"""GPU probe: forward/adjoint pair of the full-size 2-D TTI case under both schedules, and the schedules against each other."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mifwi
from seismic import Model

shape, nbl, so = (340, 140), 40, int(os.environ.get("PROBE_SO", "8"))
data = os.path.join(ROOT, "tests", "golden", "model_data", "SMARM2")
vp = np.fromfile(os.path.join(data, "vp.smooth_20"), dtype=np.float32).reshape(shape) / 1000
eps, delta, theta = 0.3 * (vp - 1.5) / 3.2, 0.2 * (vp - 1.5) / 3.2, 0.5 * (vp - 1.5) / 3.2
model = Model(origin=(0, 0), spacing=(25., 25.), shape=shape, space_order=so, vp=vp, nbl=nbl, epsilon=eps, delta=delta,
              theta=theta, bcs="damp")
dt, nt, nshots, nrec = float(model.critical_dt), int(os.environ.get("PROBE_NT", "1501")), 2, 340
src = np.empty((nshots, 1, 2), dtype=np.float32)
src[:, 0, 0] = np.array([0.3, 0.7], dtype=np.float32) * model.domain_size[0]
src[:, 0, 1] = 50.0
rec = np.empty((nrec, 2), dtype=np.float32)
rec[:, 0] = np.linspace(25.0, model.domain_size[0] - 25.0, nrec)
rec[:, 1] = 50.0
rng = np.random.default_rng(13)
s = rng.standard_normal((nshots, nt, 1)).astype(np.float32)
r = rng.standard_normal((nshots, nt, nrec)).astype(np.float32)
out = {}
for name, env in (("loop", None), ("loop2", None), ("steps", "0")):
    if env is None:
        os.environ.pop("MIFWI_TTI_PERSIST", None)
    else:
        os.environ["MIFWI_TTI_PERSIST"] = env
    for dtype in (np.float32,):
        plan = mifwi.Plan(model.grid.shape, model.spacing, model.grid.origin, so, nt, dt, 1, nrec, max_shots=nshots)
        plan.set_model(model.vp.data, model.damp.data)
        plan.tti_set_model(model.epsilon.data, model.delta.data, model.theta.data)
        plan.set_geometry(src, rec)
        fs = np.array(plan.tti_forward(s, wavelet_per_shot=True))
        k1 = plan.stats()["fwd_kernel"]
        ftr = np.array(plan.tti_adjoint(r))
        k2 = plan.stats()["bwd_kernel"]
        plan.close()
        out[name] = (fs, ftr)
        for k in range(nshots):
            a = float(np.vdot(fs[k].astype(np.float64), r[k].astype(np.float64)))
            b = float(np.vdot(s[k].astype(np.float64), ftr[k].astype(np.float64)))
            sc = max(np.linalg.norm(fs[k].astype(np.float64)) * np.linalg.norm(r[k].astype(np.float64)),
                     np.linalg.norm(s[k].astype(np.float64)) * np.linalg.norm(ftr[k].astype(np.float64)))
            print("%-5s %s %s shot %d: <Fs,r> = %.9e <s,Ftr> = %.9e diff/scale %.2e" % (name, k1, k2, k, a, b, abs(a - b) / sc), flush=True)
rel = lambda x, y: float(np.linalg.norm(x.astype(np.float64) - y) / np.linalg.norm(y.astype(np.float64)))
print("loop vs loop again: records %.2e adjoint traces %.2e" % (rel(out["loop"][0], out["loop2"][0]), rel(out["loop"][1], out["loop2"][1])))
print("loop vs steps: records %.2e adjoint traces %.2e" % (rel(out["loop"][0], out["steps"][0]), rel(out["loop"][1], out["steps"][1])))
