"""GPU probe: run-to-run determinism of the whole-loop TTI loops (forward, adjoint) on small grids."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mifwi
from seismic import Model

so = int(os.environ.get("PROBE_SO", "8"))
for shape in [tuple(int(v) for v in t.split("x")) for t in os.environ.get("PROBE_SHAPES", "28x24,24x20,60x52,92x84,248x248,312x312,412x212").split(",")]:
    nbl = 4
    rng = np.random.default_rng(3)
    vp = (2.0 + rng.random(shape)).astype(np.float32)
    model = Model(origin=(0, 0), spacing=(10., 10.), shape=shape, space_order=so, vp=vp, nbl=nbl, epsilon=0.2 * (vp - 2), delta=0.1 * (vp - 2),
                  theta=0.4 * (vp - 2), bcs="damp")
    dt, nt, nrec = float(model.critical_dt), int(os.environ.get("PROBE_NT", "300")), 16
    dom = np.array(model.domain_size)
    src = (np.array([[[0.4, 0.3]]]) * dom).astype(np.float32)
    rec = (rng.random((nrec, 2)) * dom).astype(np.float32)
    s = rng.standard_normal((1, nt, 1)).astype(np.float32)
    r = rng.standard_normal((1, nt, nrec)).astype(np.float32)
    plan = mifwi.Plan(model.grid.shape, model.spacing, model.grid.origin, so, nt, dt, 1, nrec, max_shots=1)
    plan.set_model(model.vp.data, model.damp.data)
    plan.tti_set_model(model.epsilon.data, model.delta.data, model.theta.data)
    plan.set_geometry(src, rec)
    f, a = [], []
    for k in range(5):
        plan.tti_forward(s, wavelet_per_shot=True)
        kern = plan.stats()["fwd_kernel"]
        f.append(np.array(plan.read_wavefield(3, shot=0)))
        plan.tti_adjoint(r)
        a.append(np.array(plan.read_wavefield(5, shot=0)))
    plan.close()
    gs = model.grid.shape
    print("grid %s tiles %d x %d %s: forward fields differ in %s runs, adjoint in %s runs (max abs %.2e / %.2e of %.2e / %.2e)" %
          (gs, (gs[0] + 31) // 32, (gs[1] + 31) // 32, kern, sum(int((f[0] != x).any()) for x in f[1:]),
           sum(int((a[0] != x).any()) for x in a[1:]), max(float(np.abs(f[0] - x).max()) for x in f[1:]),
           max(float(np.abs(a[0] - x).max()) for x in a[1:]), float(np.abs(f[0]).max()), float(np.abs(a[0]).max())), flush=True)
