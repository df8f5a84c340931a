"""Print the GPU-vs-golden relative errors for every golden case (tolerance calibration)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd"), os.path.join(ROOT, "tests")]
import mifwi
from golden_cases import CASES, load
from helpers import rel_l2
from oracle import oracle as orc
from seismic import Model
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for name in sorted(CASES):
    c, g = load(name)
    ndim = c["true_vp"].ndim
    nbl, so = c["nbl"], c["so"]
    dt, nt = float(g["dt"]), int(g["nt"])
    kw = dict(origin=(0.,) * ndim, spacing=c["spacing"], shape=c["true_vp"].shape, space_order=so, nbl=nbl, bcs="damp", dt=c["dt"])
    m_true, m_init = Model(vp=c["true_vp"], **kw), Model(vp=c["init_vp"], **kw)
    tv = np.linspace(0.0, dt * (nt - 1), nt)
    w = orc.ricker(float(g["f0"]), tv).astype(np.float32).reshape(nt, 1)
    plan = mifwi.Plan(m_true.grid.shape, c["spacing"], m_true.grid.origin, so, nt, dt, 1, g["rec"].shape[0], flags=flags)
    plan.set_geometry(g["src"].reshape(1, 1, ndim), g["rec"])
    plan.set_model(m_true.vp.data, m_true.damp.data)
    obs = plan.forward(w)
    k = int(g["stride"])
    plan.set_model(m_init.vp.data, m_init.damp.data)
    out = plan.l2_misfit_gradient(w, obs, want_illum=True, want_syn=True)
    crop = tuple(slice(nbl, -nbl) for _ in range(ndim))
    print("%-14s obs %.2e syn %.2e fval %.2e grad %.2e illum %.2e" % (
        name, rel_l2(obs[0][:, ::k], g["obs"]), rel_l2(out["syn"][0][:, ::k], g["syn"]),
        abs(out["fval"][0] - float(g["fval"])) / float(g["fval"]), rel_l2(out["grad"][0][crop], g["grad"]),
        rel_l2(out["illum"][0][crop], g["illum"])), flush=True)
    plan.close()
