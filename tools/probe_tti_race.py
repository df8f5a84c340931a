"""GPU probe: two identical whole-loop TTI forward runs compared sample by sample (first differing time, receivers, field)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mifwi
from seismic import Model

shape, nbl, so = (340, 140), 40, 8
data = os.path.join(ROOT, "tests", "golden", "model_data", "SMARM2")
vp = np.fromfile(os.path.join(data, "vp.smooth_20"), dtype=np.float32).reshape(shape) / 1000
eps, delta, theta = 0.3 * (vp - 1.5) / 3.2, 0.2 * (vp - 1.5) / 3.2, 0.5 * (vp - 1.5) / 3.2
model = Model(origin=(0, 0), spacing=(25., 25.), shape=shape, space_order=so, vp=vp, nbl=nbl, epsilon=eps, delta=delta,
              theta=theta, bcs="damp")
dt, nt, nshots, nrec = float(model.critical_dt), int(os.environ.get("PROBE_NT", "400")), int(os.environ.get("PROBE_SHOTS", "2")), 340
src = np.empty((nshots, 1, 2), dtype=np.float32)
src[:, 0, 0] = np.linspace(0.3, 0.7, nshots).astype(np.float32) * model.domain_size[0]
src[:, 0, 1] = 50.0
rec = np.empty((nrec, 2), dtype=np.float32)
rec[:, 0] = np.linspace(25.0, model.domain_size[0] - 25.0, nrec)
rec[:, 1] = 50.0
rng = np.random.default_rng(13)
s = rng.standard_normal((nshots, nt, 1)).astype(np.float32)
plan = mifwi.Plan(model.grid.shape, model.spacing, model.grid.origin, so, nt, dt, 1, nrec, max_shots=nshots)
plan.set_model(model.vp.data, model.damp.data)
plan.tti_set_model(model.epsilon.data, model.delta.data, model.theta.data)
plan.set_geometry(src, rec)
runs = []
for k in range(4):
    if os.environ.get("PROBE_KIND") == "acoustic":
        d = np.array(plan.forward(s, wavelet_per_shot=True))
        u = np.array(plan.read_wavefield(1, shot=0))
    else:
        d = np.array(plan.tti_forward(s, wavelet_per_shot=True))
        u = np.array(plan.read_wavefield(3, shot=0))
    runs.append((d, u))
    print("run", k, plan.stats()["fwd_kernel"], plan.stats()["persist_fallbacks"])
for k in range(1, 4):
    d0, u0 = runs[0]
    d1, u1 = runs[k]
    diff = np.argwhere(d0 != d1)
    print("run 0 vs %d: records differ at %d samples; field levels differ at %d points" % (k, len(diff), int((u0 != u1).sum())))
    if len(diff):
        t0 = diff[:, 1].min()
        first = diff[diff[:, 1] == t0]
        print("   first differing time %d: (shot, receiver) %s" % (t0, [(int(a), int(c)) for a, _, c in first[:12]]))
        for t in range(t0, min(t0 + 3, nt)):
            rr = sorted(set(int(c) for a, tt, c in diff if tt == t and a == first[0][0]))
            print("   time %d shot %d receivers %s" % (t, first[0][0], rr[:40]))
        a, t, c = first[0]
        print("   values", d0[a, t, c], d1[a, t, c], "receiver x =", rec[c, 0], "grid x index", rec[c, 0] / 25.0 + 40)
plan.close()
