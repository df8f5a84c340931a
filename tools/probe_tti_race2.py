"""GPU probe: where in the grid do identical whole-loop TTI forward runs first differ?"""
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
dt, nshots, nrec = float(model.critical_dt), 1, 340
src = np.empty((nshots, 1, 2), dtype=np.float32)
src[:, 0, 0] = 0.3 * model.domain_size[0]
src[:, 0, 1] = 50.0
rec = np.empty((nrec, 2), dtype=np.float32)
rec[:, 0] = np.linspace(25.0, model.domain_size[0] - 25.0, nrec)
rec[:, 1] = 50.0
rng = np.random.default_rng(13)
for nt in [int(v) for v in os.environ.get("PROBE_NTS", "8,16,32,64,128").split(",")]:
    s = rng.standard_normal((nshots, nt, 1)).astype(np.float32)
    plan = mifwi.Plan(model.grid.shape, model.spacing, model.grid.origin, so, nt, dt, 1, nrec, max_shots=nshots)
    plan.set_model(model.vp.data, model.damp.data)
    plan.tti_set_model(model.epsilon.data, model.delta.data, model.theta.data)
    plan.set_geometry(src, rec)
    fields = []
    for k in range(6):
        plan.tti_forward(s, wavelet_per_shot=True)
        fields.append((np.array(plan.read_wavefield(3, shot=0)), np.array(plan.read_wavefield(4, shot=0))))
    plan.close()
    os.environ["MIFWI_TTI_PERSIST"] = "0"
    plan = mifwi.Plan(model.grid.shape, model.spacing, model.grid.origin, so, nt, dt, 1, nrec, max_shots=nshots)
    plan.set_model(model.vp.data, model.damp.data)
    plan.tti_set_model(model.epsilon.data, model.delta.data, model.theta.data)
    plan.set_geometry(src, rec)
    plan.tti_forward(s, wavelet_per_shot=True)
    ref = (np.array(plan.read_wavefield(3, shot=0)), np.array(plan.read_wavefield(4, shot=0)))
    plan.close()
    os.environ.pop("MIFWI_TTI_PERSIST")
    for k in range(1, 6):
        for f, name in ((0, "u"), (1, "v")):
            d = np.argwhere(fields[0][f] != fields[k][f])
            if len(d):
                lv = (nt - 1) % 3
                dd = d[d[:, 0] == lv] if (d[:, 0] == lv).any() else d
                print("nt %d run 0 vs %d field %s: %d points differ; level %d x %d..%d z %d..%d; tiles x %s z %s; max abs diff %.3e (field max %.3e)" %
                      (nt, k, name, len(d), dd[0, 0], dd[:, 1].min(), dd[:, 1].max(), dd[:, 2].min(), dd[:, 2].max(),
                       sorted(set((dd[:, 1] // 32).tolist()))[:8], sorted(set((dd[:, 2] // 32).tolist()))[:8],
                       float(np.abs(fields[0][f] - fields[k][f]).max()), float(np.abs(fields[0][f]).max())), flush=True)
    e = [float(np.abs(fields[k][0].astype(np.float64) - ref[0]).max() / max(1e-300, np.abs(ref[0]).max())) for k in range(6)]
    print("nt %d: max |loop - steps| / max |steps| per run: %s" % (nt, " ".join("%.2e" % v for v in e)), flush=True)
