#!/usr/bin/env python
"""
Generates the golden vectors tests/golden/*.npz with the CPU oracle (oracle/), after the oracle
has been pinned against the reference's known-answer tests (tests/test_oracle_kat.py).

    python tests/golden/make_golden.py

Cases (inputs follow the benchmark configurations; Marmousi rasters are the reference's
model_data/SMARMN and model_data/SMARM2 files kept under tests/golden/model_data):
  c1_constant      demo_model('constant-isotropic') 101^2, space_order 4, setup_geometry(tn=1000), 1 shot
  c2_marmousi      SMARMN 300x106, space_order 8, dt 2.95, tn 4000, shot 3 of 8 (marmousi_fwi.py:62-117)
  c3_marmousi2     SMARM2 340x140, space_order 16, dt 3.0, tn 4500, shot 5 of 8 (marmousi2_fwi.py:61-118)
  c4_small3d       40x36x32 synthetic 3-D, space_order 8, 1 shot (small stand-in for the 512^3 stress grid)
Each file holds the inputs that are not trivially rebuilt (source/receiver positions, dt, nt) and the
expected outputs: shot record (every k-th trace), objective, cropped gradient and illumination of
the start model against data of the true model.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import oracle as orc  # noqa: E402


def marmousi(name, shape, so, dt, tn, nshots, shot, stride):
    d = os.path.join(HERE, "model_data", name)
    true_vp = np.fromfile(os.path.join(d, "vp.true"), dtype=np.float32).reshape(shape) / 1000
    init_vp = np.fromfile(os.path.join(d, "vp.smooth_20"), dtype=np.float32).reshape(shape) / 1000
    spacing, nbl = (30., 30.), 40
    dom = (shape[0] - 1) * 30.
    src = np.array([np.linspace(0, dom, nshots)[shot], 60.])
    rec = np.empty((shape[0], 2))
    rec[:, 0] = np.linspace(30., dom - 30., shape[0])
    rec[:, 1] = 60.
    return case(true_vp, init_vp, spacing, nbl, so, dt, tn, 0.007, src, rec, stride)


def case(true_vp, init_vp, spacing, nbl, so, dt, tn, f0, src, rec, stride, model_so=None):
    p_true = orc.Problem(true_vp, spacing, nbl, so, dt=dt, model_space_order=model_so)
    p_init = orc.Problem(init_vp, spacing, nbl, so, dt=dt, model_space_order=model_so)
    dtv = float(p_true.critical_dt)
    nt, tv = orc.time_axis(0.0, tn, dtv)
    w = orc.ricker(f0, tv).astype(np.float32).reshape(nt, 1)
    obs, _ = p_true.forward(src, w, rec, dt=dtv)
    syn, u = p_init.forward(src, w, rec, dt=dtv, save=True)
    fval, res = orc.least_square(syn, obs)
    grad = p_init.gradient(rec, res, u, dt=dtv)
    crop = tuple(slice(nbl, -nbl) for _ in range(true_vp.ndim))
    illum = (u * u).sum(axis=0)
    return dict(src=np.asarray(src, np.float64), rec=np.asarray(rec, np.float64), dt=np.float64(dtv), nt=np.int64(nt),
                f0=np.float64(f0), stride=np.int64(stride), obs=obs[:, ::stride].astype(np.float32),
                syn=syn[:, ::stride].astype(np.float32), fval=np.float64(fval), grad=grad[crop].astype(np.float32),
                illum=illum[crop].astype(np.float32), u_last_norm=np.float64(np.linalg.norm(u[-1].astype(np.float64))))


def main():
    out = {}
    # C1: constant model; the "true" model carries a 10 % faster inclusion so that a residual exists
    shape = (101, 101)
    vp0 = np.full(shape, 1.5, dtype=np.float32)
    vp1 = vp0.copy()
    vp1[40:60, 50:70] = 1.65
    rec = np.empty((101, 2))
    rec[:, 0] = np.linspace(0., 1000., 101)
    rec[:, 1] = 20.
    out["c1_constant"] = case(vp1, vp0, (10., 10.), 10, 4, None, 1000., 0.010, np.array([500., 10.]), rec, 5)
    out["c2_marmousi"] = marmousi("SMARMN", (300, 106), 8, 2.95, 4000., 8, 3, 10)
    out["c3_marmousi2"] = marmousi("SMARM2", (340, 140), 16, 3.0, 4500., 8, 5, 10)
    rng = np.random.default_rng(1234)
    shape3 = (40, 36, 32)
    v = rng.random(shape3)
    for _ in range(3):
        for ax in range(3):
            v = (np.roll(v, 1, ax) + v + np.roll(v, -1, ax)) / 3
    v = (1.5 + 3.0 * (v - v.min()) / (v.max() - v.min())).astype(np.float32)
    v0 = np.full(shape3, float(v.mean()), dtype=np.float32)
    g = np.linspace(20., 330., 6)
    rec3 = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    rec3 = np.concatenate([rec3, np.full((rec3.shape[0], 1), 20.)], 1)
    out["c4_small3d"] = case(v, v0, (10., 10., 10.), 8, 8, None, 150., 0.020, np.array([195., 175., 20.]), rec3, 1)
    for name, d in out.items():
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **d)
        print(name, "nt", int(d["nt"]), "dt", float(d["dt"]), "fval %.6e" % float(d["fval"]),
              "%.0f KiB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
