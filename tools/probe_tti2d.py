"""2-D TTI per-step timing on the Marmousi-II grid: whole loop vs kernel per step, 1..8 shots (tools/probe_tti2d.py)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "devito-fwi_amd")):
    sys.path.insert(0, p)
import mifwi
from bench import marmousi2, acquisition
from seismic import Model, AcquisitionGeometry

so = int(os.environ.get("PROBE_SO", "8"))
true_model, _ = marmousi2(16)
vp = np.asarray(true_model.vp.data[40:-40, 40:-40], np.float32)
scale = float(vp.max() - 1.5)
model = Model(origin=(0, 0), spacing=true_model.spacing, shape=vp.shape, space_order=so, vp=vp, nbl=40, bcs="damp",
              epsilon=0.3 * (vp - 1.5) / scale, delta=0.2 * (vp - 1.5) / scale, theta=0.5 * (vp - 1.5) / scale)
dt = float(model.critical_dt)
src32, rec = acquisition(true_model, 32)
geom = AcquisitionGeometry(model, rec, src32[0], 0., 4500., f0=0.007, src_type='Ricker')
nt = geom.nt
w = np.ascontiguousarray(geom.src.data)
for ns in [int(x) for x in os.environ.get("PROBE_SHOTS", "1,2,4,8").split(",")]:
    for name, flags in (("loop", 0), ("steps", mifwi.FLAG_NO_GRAPH if os.environ.get("PROBE_NOGRAPH") else -1)):
        if flags < 0:
            os.environ["MIFWI_TTI_PERSIST"] = "0"
            flags = 0
        else:
            os.environ.pop("MIFWI_TTI_PERSIST", None)
        plan = mifwi.Plan(model.grid.shape, model.spacing, model.grid.origin, so, nt, dt, 1, rec.shape[0], max_shots=ns)
        plan.set_model(model.vp.data, model.damp.data)
        plan.tti_set_model(model.epsilon.data, model.delta.data, model.theta.data)
        pick = src32[:: 32 // ns][:ns]
        plan.set_geometry(np.asarray(pick, np.float32).reshape(ns, 1, 2), rec)
        best = [1e9] * 4
        for _ in range(3):
            d = plan.tti_forward(w); f0 = plan.stats()["fwd_ms"]
            d = plan.tti_forward(w, save=True); f1 = plan.stats()["fwd_ms"]; k = plan.stats()["fwd_kernel"]
            plan.tti_gradient(d); g = plan.stats()["bwd_ms"]
            plan.tti_adjoint(d); a = plan.stats()["bwd_ms"]
            best = [min(x, y) for x, y in zip(best, (f0, f1, g, a))]
        print("so=%d shots=%d %-5s %-22s per step: forward %.2f  forward+save %.2f  gradient %.2f  adjoint %.2f us  (fallbacks %d)" %
              ((so, ns, name, k) + tuple(b / (nt - 2) * 1e3 for b in best) + (plan.stats()["persist_fallbacks"],)), flush=True)
        plan.close()
