"""GPU probe: margins of the golden-vector tests (worst relative error against the oracle per case)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import test_gpu_golden as t
from helpers import rel_l2
import mifwi
from oracle import oracle as orc
from seismic import Model
for name in sorted(t.CASES):
    c, g = t.load(name)
    ndim = c["true_vp"].ndim
    dt, nt = float(g["dt"]), int(g["nt"])
    kw = dict(origin=(0.,) * ndim, spacing=c["spacing"], shape=c["true_vp"].shape, space_order=c["so"], nbl=c["nbl"], bcs="damp", dt=c["dt"])
    m_true = Model(vp=c["true_vp"], **kw)
    tv = np.linspace(0.0, dt * (nt - 1), nt)
    w = orc.ricker(float(g["f0"]), tv).astype(np.float32).reshape(nt, 1)
    for env in ("1", "0"):
        os.environ["MIFWI_PERSIST"] = env
        plan = mifwi.Plan(m_true.grid.shape, c["spacing"], m_true.grid.origin, c["so"], nt, dt, 1, g["rec"].shape[0])
        plan.set_geometry(g["src"].reshape(1, 1, ndim), g["rec"])
        plan.set_model(m_true.vp.data, m_true.damp.data)
        obs = plan.forward(w)
        k = int(g["stride"])
        print(name, "persist" if env == "1" else "per-step", "obs rel_l2 %.3e" % rel_l2(obs[0][:, ::k], g["obs"]), flush=True)
        plan.close()
