"""CPU checks of the TTI restatement (oracle/acoustic_oracle_impl.h, `TTIProblem`).

The reference holds no TTI numbers: its only TTI test asserts finite records on a tiny grid
(seismic/tti/tti_example.py:42-49), reproduced here.  The restatement is further constrained by two
exact identities: with epsilon = delta = 0 the sum u + v equals twice the isotropic acoustic field for
any tilt, and the forward / adjoint operators (seismic/tti/operators.py:234-238,306-311) are an
adjoint pair."""
import numpy as np
import pytest

from oracle import oracle as orc
from helpers import smooth_random_vp


def tti_case(shape, so=8, nbl=8, seed=0, dtype=np.float64, aniso=True):
    nd = len(shape)
    vp = smooth_random_vp(shape, seed=seed)
    k = 1.0 if aniso else 0.0
    eps, delta = k * .3 * (vp - 1.5), k * .2 * (vp - 1.5)
    theta = .5 * (vp - 1.5)
    phi = .25 * (vp - 1.5) if nd == 3 else None
    prob = orc.TTIProblem(vp, (10.,) * nd, nbl, so, eps, delta, theta, phi, dtype=dtype)
    rng = np.random.default_rng(seed + 1)
    dom = np.array([(n - 1) * 10. for n in shape])
    src = (0.5 * dom + 3.3)[None, :]
    rec = (rng.random((7, nd)) * 0.6 + 0.2) * dom
    return prob, vp, src, rec, rng


@pytest.mark.parametrize("shape", [(41, 37), (19, 17, 21)])
def test_tti_adjoint_pair_and_isotropic_limit(shape):
    prob, vp, src, rec, rng = tti_case(shape)
    dt = float(prob.critical_dt)
    nt = 110 if len(shape) == 2 else 50
    s = rng.standard_normal((nt, 1))
    fs = prob.forward(src, s, rec, dt=dt)
    r = rng.standard_normal(fs.shape)
    ftr = prob.adjoint(src, rec, r, dt=dt)
    lhs, rhs = float(np.sum(fs * r)), float(np.sum(s * ftr))
    assert np.abs(fs).max() > 0
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))
    # isotropic limit, tilt kept
    iso, _, _, _, _ = tti_case(shape, aniso=False)
    w = orc.ricker(0.03, dt * np.arange(nt)).reshape(nt, 1)
    d_tti = iso.forward(src, w, rec, dt=dt)
    d_ac, _ = orc.Problem(vp, prob.spacing, prob.nbl, prob.space_order, dtype=np.float64).forward(src, w, rec, dt=dt)
    assert np.linalg.norm(d_tti - 2 * d_ac) <= 1e-12 * np.linalg.norm(d_ac)
    # anisotropy changes the data
    d_an = prob.forward(src, w, rec, dt=dt)
    assert np.linalg.norm(d_an - d_tti) > 1e-2 * np.linalg.norm(d_tti)


@pytest.mark.parametrize("ndim", [2, 3])
def test_tti_stability_kat(ndim):
    """tti_example.py:42-49: 11^ndim points, spacing 20, no absorbing layer, tn = 16000, 'layers-tti'."""
    shape = (11,) * ndim
    v = np.full(shape, 1.5)
    levels = np.linspace(1.5, 3.5, 3)
    for i in range(1, 3):
        v[..., i * int(11 / 3):] = levels[i]
    phi = .25 * (v - 1.5) if ndim == 3 else None
    prob = orc.TTIProblem(v, (20.,) * ndim, 0, 4, .3 * (v - 1.5), .2 * (v - 1.5), .5 * (v - 1.5), phi,
                          dtype=np.float32)
    dt = float(prob.critical_dt)
    # CFL with the Thomsen scale sqrt(1 + 2 max(eps)) (model.py:324-329,364)
    assert np.isclose(dt, float(np.float32("%.3e" % (orc.cfl_coeff(4, ndim) * 20. / (np.sqrt(2.2) * 3.5)))))
    nt, tv = orc.time_axis(0., 16000., dt)
    w = orc.ricker(0.010, tv).reshape(nt, 1)
    src = np.full((1, ndim), 100.)
    src[0, -1] = 20.
    rec = np.stack([np.linspace(0, 200, 11)] + [np.full(11, 100.)] * (ndim - 2) + [np.full(11, 40.)], axis=1)
    d = prob.forward(src, w, rec, dt=dt)
    assert np.isfinite(np.linalg.norm(d.astype(np.float64))) and np.abs(d).max() > 0


def test_tti_model_and_presets():
    from seismic import demo_model
    m = demo_model('layers-tti', shape=(30, 20, 25), spacing=(10., 10., 10.), nbl=5, space_order=4)
    assert m.epsilon.data.shape == m.grid.shape == (40, 30, 35)
    assert np.isclose(m.epsilon.data.max(), .3 * 2.0) and np.isclose(m.phi.data.max(), .25 * 2.0)
    iso = demo_model('layers-isotropic', shape=(30, 20, 25), spacing=(10., 10., 10.), nbl=5, space_order=4)
    assert np.isclose(float(m.critical_dt) * np.sqrt(1 + 2 * .6), float(iso.critical_dt), rtol=2e-3)
    m2 = demo_model('layers-tti-noazimuth', shape=(30, 20, 25), nbl=5, space_order=4)
    assert not hasattr(m2, 'phi') or m2.phi is None or not hasattr(m2.phi, 'data')
    c = demo_model('constant-tti', shape=(20, 20), nbl=4, space_order=4)
    assert np.allclose(c.theta.data, .7) and np.allclose(c.delta.data, .2)


@pytest.mark.parametrize("shape", [(37, 33), (15, 14, 17)])
def test_tti_jacobian_pair_is_adjoint_and_linearises_forward(shape):
    """seismic/tti/operators.py:529-639: <J dm, r> = <dm, J^T r> for the Born / gradient pair, and
    J dm is the derivative of the forward map with respect to the squared slowness."""
    prob, vp, src, rec, rng = tti_case(shape, nbl=6)
    dt = float(prob.critical_dt)
    nt = 90 if len(shape) == 2 else 40
    w = orc.ricker(0.03, dt * np.arange(nt)).reshape(nt, 1)
    dm = np.zeros(prob.shape_pad)
    inner = tuple(slice(8, -8) for _ in shape)
    dm[inner] = 0.02 * rng.standard_normal(dm[inner].shape)
    jd = prob.born(src, w, rec, dm, dt=dt)
    assert np.abs(jd).max() > 0
    r = rng.standard_normal(jd.shape)
    g = prob.gradient(src, w, rec, r, dt=dt)
    lhs, rhs = float(np.sum(jd * r)), float(np.sum(dm * g))
    assert abs(lhs - rhs) <= 1e-12 * max(abs(lhs), abs(rhs))
    # gradient accumulates
    g2 = prob.gradient(src, w, rec, r, dt=dt, grad=g.copy())
    np.testing.assert_allclose(g2, 2 * g, rtol=1e-12, atol=0)
    # finite difference of the forward map in m = 1/vp^2
    d0 = prob.forward(src, w, rec, dt=dt)
    errs = []
    for h in (1e-2, 1e-3):
        pert = copy_with_m(prob, prob.vp ** -2.0 + h * dm)
        errs.append(np.linalg.norm((pert.forward(src, w, rec, dt=dt) - d0) / h - jd) / np.linalg.norm(jd))
    assert errs[1] < 2e-3 and errs[1] < 0.2 * errs[0]       # first-order convergence


def copy_with_m(prob, m):
    import copy
    q = copy.copy(prob)
    q.vp = np.ascontiguousarray(m ** -0.5)
    return q
