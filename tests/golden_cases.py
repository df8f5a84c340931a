"""Rebuilds the inputs of the golden cases (see tests/golden/make_golden.py) for the tests."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")

# name -> (velocity builder, spacing, nbl, space_order, user dt, model_space_order)


def _marm(name, shape):
    d = os.path.join(GOLD, "model_data", name)
    t = np.fromfile(os.path.join(d, "vp.true"), dtype=np.float32).reshape(shape) / 1000
    s = np.fromfile(os.path.join(d, "vp.smooth_20"), dtype=np.float32).reshape(shape) / 1000
    return t, s


def _c1():
    vp0 = np.full((101, 101), 1.5, dtype=np.float32)
    vp1 = vp0.copy()
    vp1[40:60, 50:70] = 1.65
    return vp1, vp0


def _c4():
    rng = np.random.default_rng(1234)
    v = rng.random((40, 36, 32))
    for _ in range(3):
        for ax in range(3):
            v = (np.roll(v, 1, ax) + v + np.roll(v, -1, ax)) / 3
    v = (1.5 + 3.0 * (v - v.min()) / (v.max() - v.min())).astype(np.float32)
    return v, np.full(v.shape, float(v.mean()), dtype=np.float32)


CASES = {
    "c1_constant": dict(models=_c1, spacing=(10., 10.), nbl=10, so=4, dt=None),
    "c2_marmousi": dict(models=lambda: _marm("SMARMN", (300, 106)), spacing=(30., 30.), nbl=40, so=8, dt=2.95),
    "c3_marmousi2": dict(models=lambda: _marm("SMARM2", (340, 140)), spacing=(30., 30.), nbl=40, so=16, dt=3.0),
    "c4_small3d": dict(models=_c4, spacing=(10., 10., 10.), nbl=8, so=8, dt=None),
}


def load(name):
    g = dict(np.load(os.path.join(GOLD, name + ".npz")))
    c = dict(CASES[name])
    c["true_vp"], c["init_vp"] = c.pop("models")()
    return c, g
