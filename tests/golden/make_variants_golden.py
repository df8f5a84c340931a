"""Golden vectors of the solver variants beyond the plain OT2 propagators -- TTI forward / adjoint / Born /
gradient, acoustic Born, OT4, free surface -- on small grids, from the CPU oracle in float64 (stored as
float32 where the values are records).  They pin the oracle against accidental change: the reference holds
no output for these variants (TTI, Born and OT4 are parity-unpinned; the free surface has the 369.955 norm).

    python tests/golden/make_variants_golden.py      # rewrites tests/golden/variants_small.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]
from oracle import oracle as orc                           # noqa: E402
from helpers import smooth_random_vp                       # noqa: E402


def cases():
    """name -> callable returning a dict of arrays (everything float64 inside)."""
    out = {}
    for nd, shape, so in ((2, (44, 38), 8), (3, (20, 18, 22), 4)):
        tag = "%dd" % nd
        vp = smooth_random_vp(shape, seed=11)
        dom = np.array([(n - 1) * 10. for n in shape])
        src = (0.45 * dom + 2.7)[None, :]
        rec = np.stack([np.linspace(0.2, 0.8, 6) * dom[d] + 1.3 * d for d in range(nd)], axis=1)
        nt = 80 if nd == 2 else 40
        eps, delta, theta = .3 * (vp - 1.5), .2 * (vp - 1.5), .5 * (vp - 1.5)
        phi = .25 * (vp - 1.5) if nd == 3 else None
        tti = orc.TTIProblem(vp, (10.,) * nd, 6, so, eps, delta, theta, phi, dtype=np.float64)
        dt = float(tti.critical_dt)
        w = orc.ricker(0.03, dt * np.arange(nt)).reshape(nt, 1)
        dm = np.zeros(tti.shape_pad)
        dm[tuple(slice(9, -9) for _ in shape)] = 0.02 * np.sin(np.arange(np.prod([s - 18 for s in tti.shape_pad]))).reshape(
            [s - 18 for s in tti.shape_pad])
        d = tti.forward(src, w, rec, dt=dt)
        out["tti_forward_" + tag] = d
        out["tti_adjoint_" + tag] = tti.adjoint(src, rec, d, dt=dt)
        out["tti_born_" + tag] = tti.born(src, w, rec, dm, dt=dt)
        out["tti_gradient_" + tag] = tti.gradient(src, w, rec, d, dt=dt)
        iso = orc.Problem(vp, (10.,) * nd, 6, so, dtype=np.float64)
        dti = float(iso.critical_dt)
        wi = orc.ricker(0.03, dti * np.arange(nt)).reshape(nt, 1)
        out["born_" + tag] = iso.born(src, wi, rec, dm, dt=dti)
        fsp = orc.Problem(vp, (10.,) * nd, 6, so, dtype=np.float64, fs=True)
        out["free_surface_forward_" + tag] = fsp.forward(src, wi, rec, dt=dti)[0]
        ot4 = orc.Problem(vp, (10.,) * nd, 6, so, dtype=np.float64, kernel='OT4')
        d4 = float(ot4.solver_dt)
        w4 = orc.ricker(0.03, d4 * np.arange(nt)).reshape(nt, 1)
        r4, u4 = ot4.forward(src, w4, rec, dt=d4, save=True)
        out["ot4_forward_" + tag] = r4
        out["ot4_gradient_" + tag] = ot4.gradient(rec, r4, u4, dt=d4)
    return out


if __name__ == "__main__":
    arrays = {k: np.asarray(v, dtype=np.float32) for k, v in cases().items()}
    np.savez_compressed(os.path.join(HERE, "variants_small.npz"), **arrays)
    for k, v in arrays.items():
        print("%-28s %-16s |x| = %.6e" % (k, v.shape, np.linalg.norm(v.astype(np.float64))))
