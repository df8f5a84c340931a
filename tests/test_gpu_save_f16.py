"""MIFWI_FLAG_SAVE_F16: the saved forward wavefield kept in half precision on the device (the device-side
counterpart of the compressed wavefield serialisation of seismic/tutorials/12_time_blocking.ipynb
cells 15-37).  The time stepping stays float32, so shot records are unchanged; the gradient and the
illumination see the saved levels rounded to 11 bits (relative error <= 4.9e-4 per value).  Tolerances
(rel-L2 against the oracle): 2e-3 on the gradient with the white-noise residuals and short runs used
here, where the correlation sum cancels most (measured 4e-4 in 2-D, 1.1e-3 in 3-D; 5e-5 with the data
residual of a Marmousi-II shot, where the oracle itself was fed rounded levels), 3e-4 on the
illumination, 1e-3 per saved level read back."""
import numpy as np
import pytest

from helpers import Case, rel_l2
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mifwi():
    import mifwi as m
    return m


@pytest.mark.parametrize("shape,so,nshots", [((70, 50), 8, 3), ((61, 83), 16, 2), ((30, 28, 40), 8, 2), ((26, 24, 70), 16, 1)])
def test_half_precision_saved_levels(mifwi, shape, so, nshots):
    nd = len(shape)
    case = Case(shape, (10.,) * nd, 10, so, nt=120 if nd == 2 else 60, nshots=nshots, nrec=11, seed=5, f0=0.03)
    prob, w, src, rec = case.prob, case.wavelet, case.src, case.rec
    dt, nt = case.dt, case.nt
    rng = np.random.default_rng(2)
    res = rng.standard_normal((nshots, nt, rec.shape[0])).astype(np.float32)
    out = {}
    # both with a kernel per step (plans with half-precision levels never run a loop as one launch)
    for name, flags in (("f32", mifwi.FLAG_NO_GRAPH), ("f16", mifwi.FLAG_SAVE_F16)):
        plan = case.plan(mifwi, flags=flags)
        d = plan.forward(w, save=True)
        g, il = plan.gradient(res, want_illum=True)
        u = plan.read_wavefield(0, shot=nshots - 1, first=nt // 2, count=3)
        out[name] = (d, g, il, u)
        if name == "f16":
            with pytest.raises(RuntimeError):
                plan.write_wavefield(np.zeros((nt,) + prob.shape_pad, np.float32))
        plan.close()
    print("half-precision levels: grad %.2e illum %.2e levels %.2e" % (
        rel_l2(out["f16"][1], out["f32"][1]), rel_l2(out["f16"][2], out["f32"][2]), rel_l2(out["f16"][3], out["f32"][3])))
    assert np.array_equal(out["f32"][0], out["f16"][0])            # records: same float32 time stepping
    assert rel_l2(out["f16"][1], out["f32"][1]) <= 2e-3 and not np.array_equal(out["f16"][1], out["f32"][1])
    assert rel_l2(out["f16"][2], out["f32"][2]) <= 3e-4
    assert rel_l2(out["f16"][3], out["f32"][3]) <= 1e-3
    for s in range(nshots):                                        # and against the oracle
        syn, u = prob.forward(src[s], w, rec, dt=dt, save=True)
        assert rel_l2(out["f16"][0][s], syn) <= 1e-5
        gref = prob.gradient(rec, res[s], u, dt=dt)
        assert rel_l2(out["f16"][1][s], gref) <= 2e-3


def test_half_precision_scale_follows_the_wavelet(mifwi):
    """A wavelet 1e-9 or 1e+9 times the usual amplitude would underflow / overflow a fixed-scale half
    format; the per-shot scale is taken from the injected terms, so the relative accuracy is the same."""
    case = Case((64, 48), (10., 10.), 10, 8, nt=100, nshots=2, nrec=9, seed=7, f0=0.03)
    prob, src, rec, dt, nt = case.prob, case.src, case.rec, case.dt, case.nt
    rng = np.random.default_rng(3)
    res = rng.standard_normal((2, nt, rec.shape[0])).astype(np.float32)
    ref = None
    for amp in (1.0, 1e-9, 1e9):
        w = np.stack([case.wavelet * amp, case.wavelet * amp * 1e-3])      # shots of very different strength
        plan = case.plan(mifwi, flags=mifwi.FLAG_SAVE_F16)
        plan.forward(w, save=True, wavelet_per_shot=True)
        g = plan.gradient(res) / amp
        plan.close()
        assert np.isfinite(g).all() and np.abs(g[1]).max() > 0
        if ref is None:
            ref = g
        else:
            assert rel_l2(g, ref) <= 1e-3
    syn, u = prob.forward(src[1], case.wavelet * np.float32(1e-3), rec, dt=dt, save=True)
    assert rel_l2(ref[1], prob.gradient(rec, res[1], u, dt=dt)) <= 1e-3
