"""GPU probe: whole-loop TTI forward with blocks sharing CUs against the same launch with one block per CU: where
and when do they first differ?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mifwi
from seismic import Model

so = int(os.environ.get("PROBE_SO", "8"))
shape, nbl = (120, 120), 4
rng = np.random.default_rng(3)
vp = (2.0 + rng.random(shape)).astype(np.float32)
model = Model(origin=(0, 0), spacing=(10., 10.), shape=shape, space_order=so, vp=vp, nbl=nbl, epsilon=0.2 * (vp - 2), delta=0.1 * (vp - 2),
              theta=0.4 * (vp - 2), bcs="damp")
dt, nrec = float(model.critical_dt), 16
dom = np.array(model.domain_size)
src = (np.array([[[0.4, 0.3]]]) * dom).astype(np.float32)
rec = (rng.random((nrec, 2)) * dom).astype(np.float32)
for nt in (4, 6, 10, 20, 40):
    s = rng.standard_normal((1, nt, 1)).astype(np.float32)
    res = {}
    for extra in ("0", "100"):
        os.environ["MIFWI_TTI_SURPLUS_BLOCKS"] = extra
        plan = mifwi.Plan(model.grid.shape, model.spacing, model.grid.origin, so, nt, dt, 1, nrec, max_shots=1)
        plan.set_model(model.vp.data, model.damp.data)
        plan.tti_set_model(model.epsilon.data, model.delta.data, model.theta.data)
        plan.set_geometry(src, rec)
        out = []
        for k in range(6):
            plan.tti_forward(s, wavelet_per_shot=True)
            out.append(np.array(plan.read_wavefield(3, shot=0))[(nt - 1) % 3])
        plan.close()
        res[extra] = out
    ref = res["0"][0]
    assert all((ref == x).all() for x in res["0"])
    for k, x in enumerate(res["100"]):
        d = np.argwhere(ref != x)
        if len(d):
            inx, inz = d[:, 0] % 32, d[:, 1] % 32
            print("nt %d run %d: %d points differ, max abs %.2e (field max %.2e); x %d..%d z %d..%d; position in tile x: %s z: %s" %
                  (nt, k, len(d), float(np.abs(ref - x).max()), float(np.abs(ref).max()), d[:, 0].min(), d[:, 0].max(), d[:, 1].min(),
                   d[:, 1].max(), sorted(set(inx.tolist()))[:40], sorted(set(inz.tolist()))[:40]), flush=True)
        else:
            print("nt %d run %d: equal" % (nt, k), flush=True)
