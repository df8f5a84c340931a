"""The oracle reproduces the committed golden vectors (guards the checker itself against drift)."""
import numpy as np
import pytest

from golden_cases import CASES, load
from helpers import rel_l2
from oracle import oracle as orc


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    c, g = load(name)
    p_true = orc.Problem(c["true_vp"], c["spacing"], c["nbl"], c["so"], dt=c["dt"])
    p_init = orc.Problem(c["init_vp"], c["spacing"], c["nbl"], c["so"], dt=c["dt"])
    dt = float(p_true.critical_dt)
    assert dt == float(g["dt"])
    nt, tv = orc.time_axis(0.0, {"c1_constant": 1000., "c2_marmousi": 4000., "c3_marmousi2": 4500.,
                                 "c4_small3d": 150.}[name], dt)
    assert nt == int(g["nt"])
    w = orc.ricker(float(g["f0"]), tv).astype(np.float32).reshape(nt, 1)
    k = int(g["stride"])
    obs, _ = p_true.forward(g["src"], w, g["rec"], dt=dt)
    assert rel_l2(obs[:, ::k], g["obs"]) < 1e-6
    if name in ("c1_constant", "c4_small3d"):     # full objective + gradient only for the quick cases
        syn, u = p_init.forward(g["src"], w, g["rec"], dt=dt, save=True)
        fval, res = orc.least_square(syn, obs)
        assert np.isclose(fval, float(g["fval"]), rtol=1e-5)
        grad = p_init.gradient(g["rec"], res, u, dt=dt)
        crop = tuple(slice(c["nbl"], -c["nbl"]) for _ in range(c["true_vp"].ndim))
        assert rel_l2(grad[crop], g["grad"]) < 1e-5
