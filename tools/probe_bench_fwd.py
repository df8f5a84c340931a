"""GPU probe: forward (save) and gradient per-step time on the exact bench.py workload, by variant."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
import bench, mifwi
from seismic import AcquisitionGeometry
true_model, init_model = bench.marmousi2(16)
src_all, rec = bench.acquisition(true_model, 8)
geom = AcquisitionGeometry(true_model, rec, src_all[0], 0., 4500., f0=0.007, src_type='Ricker')
nt, dt = geom.nt, float(true_model.critical_dt)
wav = np.ascontiguousarray(geom.src.data)
for name, srcs, recs in (("bench geometry", src_all, rec),
                         ("sources moved to z=1000", src_all + np.array([0., 950.]), rec),
                         ("receivers moved to z=1000", src_all, rec + np.array([0., 950.]))):
    plan = mifwi.Plan(true_model.grid.shape, true_model.spacing, true_model.grid.origin, 16, nt, dt, 1, rec.shape[0],
                      max_shots=8)
    plan.set_model(init_model.vp.data, true_model.damp.data)
    plan.set_geometry(srcs.reshape(8, 1, 2).astype(np.float32), recs.astype(np.float32))
    t = []
    for _ in range(3):
        r = plan.forward(wav, save=True)
        st = plan.stats()
        t.append(st["fwd_ms"] * 1e3 / st["fwd_launches"])
    g = []
    for _ in range(3):
        plan.gradient(r, want_illum=True)
        st = plan.stats()
        g.append(st["bwd_ms"] * 1e3 / st["bwd_launches"])
    print("%-28s fwd save %.2f us  grad %.2f us" % (name, min(t), min(g)), flush=True)
    if os.environ.get("PROBE_SUSTAIN"):
        import time
        for k in range(12):
            t0 = time.perf_counter()
            plan.forward(wav, save=True, want_rec=False)
            f = plan.stats()["fwd_ms"]
            plan.gradient(r, want_illum=True)
            b = plan.stats()["bwd_ms"]
            print("   pass %2d: fwd %.3f ms  bwd %.3f ms  wall %.2f ms" % (k, f, b, (time.perf_counter() - t0) * 1e3), flush=True)
        break
    plan.close()
