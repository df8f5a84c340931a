"""
Size-independent properties at the full benchmark sizes, where the CPU oracle is too slow to be
the checker: the discrete forward and adjoint propagators are an adjoint pair
(<F s, r> = <s, F^T r>, reference seismic/acoustic/operators.py:98-180), forward modelling is
linear in the wavelet, and the checkpointed gradient equals the fully saved one bit for bit.

Sizes: the 3-D north-star grid 512^3 + 2*40 = 592^3 at space order 16 (and 8), and Marmousi-II
(420 x 220 padded, nt = 1501, 8 shots, space order 16).
Tolerance of the float32 adjoint test: 2e-5 relative in 3-D (measured 7.7e-6 at order 16, 1.4e-6 at
order 8).  For Marmousi-II (full nt, white-noise test vectors) the inner products cancel to ~1e-5 of
the product of the norms, so the difference is bounded against that product instead:
|<a,b> - <c,d>| <= tol * max(|a||b|, |c||d|), tol = 5e-8 for the acoustic pair (measured 1e-9 .. 9e-9)
and 3e-6 for the TTI propagator and Jacobian pairs (measured 3e-7 .. 8e-7, tiled and untiled kernels
alike: Gxx = laplace - Gzz subtracts two large float32 terms, and white noise is the worst case for
it).  In float64 the same full-size TTI pair agrees to 2e-15.
"""
import os

import numpy as np
import pytest

from helpers import rel_l2

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mifwi():
    import mifwi as m
    return m


def _dot(a, b):
    return float(np.sum(a.astype(np.float64) * b.astype(np.float64)))


def _pair_ok(a, b, c, d, tol):
    """<a, b> against <c, d>, bounded relative to the Cauchy-Schwarz scale of the two products."""
    n = lambda x: float(np.linalg.norm(x.astype(np.float64).ravel()))
    lhs, rhs = _dot(a, b), _dot(c, d)
    scale = max(n(a) * n(b), n(c) * n(d))
    print("<a,b> = %.9e  <c,d> = %.9e  diff/scale = %.2e" % (lhs, rhs, abs(lhs - rhs) / scale))
    return abs(lhs - rhs) <= tol * scale and abs(lhs) > 1e-12 * scale


@pytest.mark.parametrize("so", [16, 8])
def test_3d_full_grid_adjoint_pair_and_linearity(mifwi, so):
    from seismic import Model
    n, nbl, nt = 512, 40, 48
    vp = np.empty((n, n, n), dtype=np.float32)
    vp[:] = np.linspace(1.5, 4.5, n, dtype=np.float32)[None, None, :]
    vp += 0.2 * np.sin(np.arange(n, dtype=np.float32) * 0.05)[:, None, None]
    model = Model(origin=(0., 0., 0.), spacing=(10., 10., 10.), shape=vp.shape, space_order=so, vp=vp, nbl=nbl,
                  bcs="damp")
    del vp
    dt = float(model.critical_dt)
    sp = model.grid.shape
    assert sp == (592, 592, 592)
    rng = np.random.default_rng(7)
    nsrc, nrec = 3, 200
    src = (rng.random((1, nsrc, 3)) * 0.5 + 0.25).astype(np.float32) * 5110.0
    rec = (rng.random((nrec, 3)) * 0.5 + 0.25).astype(np.float32) * 5110.0
    # keep sources and receivers within reach of each other in nt steps
    rec = (src[0, 0] + (rec - 2555.0) * 0.06).astype(np.float32)
    src[0, 1:] = src[0, 0] + np.array([[55.0, -32.0, 41.0], [-61.0, 17.0, -23.0]], dtype=np.float32)
    plan = mifwi.Plan(sp, model.spacing, model.grid.origin, so, nt, dt, nsrc, nrec, max_shots=1)
    plan.set_model(model.vp.data, model.damp.data)
    plan.set_geometry(src, rec)
    s = rng.standard_normal((nt, nsrc)).astype(np.float32)
    r = rng.standard_normal((1, nt, nrec)).astype(np.float32)
    fs = plan.forward(s)
    assert np.abs(fs).max() > 0
    ftr = plan.adjoint(r)
    lhs, rhs = _dot(fs, r), _dot(s, ftr[0])
    print("3-D so=%d: <Fs,r> = %.9e  <s,F^T r> = %.9e" % (so, lhs, rhs))
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), abs(rhs))
    # linearity in the wavelet
    f2 = plan.forward(2.0 * s)
    assert np.linalg.norm((f2 - 2.0 * fs).ravel()) <= 1e-6 * np.linalg.norm(fs.ravel())
    plan.close()


def test_3d_512_gradient_checkpoint_identity_and_jacobian_pair(mifwi):
    """BASELINE config 4's grid (512^3 + sponge = 592^3, space order 8) with a time axis short enough for the
    full save to fit (40 levels = 33 GB): the checkpointed gradient equals the fully saved one bit for bit, and
    Born modelling and the gradient are an adjoint pair, <J dm, r> = <dm, J^T r>
    (seismic/acoustic/operators.py:183-273).  Device tensors carry the 830 MB grid arrays."""
    import torch
    from seismic import Model
    n, nbl, so, nt = 512, 40, 8, 40
    vp = np.empty((n, n, n), dtype=np.float32)
    vp[:] = np.linspace(1.5, 4.5, n, dtype=np.float32)[None, None, :]
    vp += 0.2 * np.sin(np.arange(n, dtype=np.float32) * 0.05)[:, None, None]
    model = Model(origin=(0., 0., 0.), spacing=(10., 10., 10.), shape=vp.shape, space_order=so, vp=vp, nbl=nbl,
                  bcs="damp")
    del vp
    dt = float(model.critical_dt)
    sp = model.grid.shape
    rng = np.random.default_rng(17)
    nrec = 256
    src = np.array([[[2555.0, 2555.0, 2555.0]]], dtype=np.float32) + rng.random((1, 1, 3)).astype(np.float32) * 7
    rec = (src[0, 0] + (rng.random((nrec, 3)) - 0.5) * 300.0).astype(np.float32)
    s = rng.standard_normal((nt, 1)).astype(np.float32)
    r = rng.standard_normal((1, nt, nrec)).astype(np.float32)
    dev = torch.device("cuda")
    grads = {}
    # 3-D loops run a kernel per step either way: the fully saved and the checkpointed plan share the schedule
    for name, flags in (("full", 0), ("ckpt", mifwi.FLAG_CHECKPOINT)):
        plan = mifwi.Plan(sp, model.spacing, model.grid.origin, so, nt, dt, 1, nrec, max_shots=1, flags=flags)
        plan.set_model(model.vp.data, model.damp.data)
        plan.set_geometry(src, rec)
        plan.forward(s, save=True, want_rec=False)
        g = torch.zeros((1,) + sp, dtype=torch.float32, device=dev)
        il = torch.zeros_like(g)
        plan.gradient(r, grad=g, illum=il)
        grads[name] = (g, il)
        if name == "full":
            # dm: smooth-free random perturbation around the source, where both fields live within nt steps
            dm = torch.zeros(sp, dtype=torch.float32, device=dev)
            c = slice(296 - 24, 296 + 24)
            dm[c, c, c] = 0.01 * torch.from_numpy(rng.standard_normal((48, 48, 48)).astype(np.float32)).to(dev)
            jd = plan.born(s, dm)
            assert np.abs(jd).max() > 0
            lhs = _dot(jd, r)
            rhs = float((dm.double() * g[0].double()).sum().item())
            scale = max(float(np.linalg.norm(jd.astype(np.float64))) * float(np.linalg.norm(r.astype(np.float64))),
                        float(dm.double().norm().item()) * float(g.double().norm().item()))
            print("592^3 so=8: <J dm, r> = %.9e  <dm, J^T r> = %.9e  diff/scale = %.2e" %
                  (lhs, rhs, abs(lhs - rhs) / scale))
            assert abs(lhs - rhs) <= 1e-5 * scale and abs(lhs) > 1e-9 * scale
            del dm
        plan.close()
    assert torch.equal(grads["full"][0], grads["ckpt"][0])
    assert torch.equal(grads["full"][1], grads["ckpt"][1])
    assert float(grads["full"][0].abs().max().item()) > 0


def test_marmousi2_full_adjoint_pair_and_checkpoint_identity(mifwi):
    from seismic import Model
    shape, nbl, so = (340, 140), 40, 16
    data = os.path.join(ROOT, "tests", "golden", "model_data", "SMARM2")
    vp = np.fromfile(os.path.join(data, "vp.smooth_20"), dtype=np.float32).reshape(shape) / 1000
    model = Model(origin=(0, 0), spacing=(25., 25.), shape=shape, space_order=so, vp=vp, nbl=nbl, dt=3.0,
                  bcs="damp")
    dt, nt, nshots, nrec = 3.0, 1501, 8, 340
    src = np.empty((nshots, 1, 2), dtype=np.float32)
    src[:, 0, 0] = np.linspace(0, model.domain_size[0], nshots)
    src[:, 0, 1] = 50.0
    rec = np.empty((nrec, 2), dtype=np.float32)
    rec[:, 0] = np.linspace(25.0, model.domain_size[0] - 25.0, nrec)
    rec[:, 1] = 50.0
    rng = np.random.default_rng(11)
    s = rng.standard_normal((nshots, nt, 1)).astype(np.float32)
    r = rng.standard_normal((nshots, nt, nrec)).astype(np.float32)
    out = {}
    # "full" runs each loop of these 8 shots as one launch, "steps" and "ckpt" a kernel per step
    for name, flags in (("full", 0), ("steps", mifwi.FLAG_NO_GRAPH), ("ckpt", mifwi.FLAG_CHECKPOINT)):
        plan = mifwi.Plan(model.grid.shape, model.spacing, model.grid.origin, so, nt, dt, 1, nrec, max_shots=nshots,
                          flags=flags)
        plan.set_model(model.vp.data, model.damp.data)
        plan.set_geometry(src, rec)
        if name == "full":
            fs = plan.forward(s, wavelet_per_shot=True)
            ftr = plan.adjoint(r)
            for k in range(nshots):
                assert _pair_ok(fs[k], r[k], s[k], ftr[k], 5e-8), k
        plan.forward(s, wavelet_per_shot=True, save=True)
        out[name] = plan.gradient(r, want_illum=True)
        plan.close()
    assert np.array_equal(out["steps"][0], out["ckpt"][0])
    assert np.array_equal(out["steps"][1], out["ckpt"][1])
    # the whole-loop launch sums the same terms in the same order; the field update rounds differently
    assert rel_l2(out["full"][0], out["steps"][0]) < 5e-5
    assert rel_l2(out["full"][1], out["steps"][1]) < 5e-6


def test_marmousi2_tti_full_size_pairs(mifwi):
    """2-D TTI on Marmousi-II at space order 8, two shots (the per-GPU share of 16 shots on 8 GPUs), full
    nt: <F s, r> = <s, F^T r> and <J dm, r> = <dm, J^T r> (seismic/tti/operators.py:430-639) in float32.
    The reference ships no Marmousi-II anisotropy rasters: epsilon, delta, theta follow the recipe of its
    `layers-tti` preset (preset_models.py:212-214) applied to vp, scaled to the velocity range."""
    from seismic import Model
    shape, nbl, so = (340, 140), 40, 8
    data = os.path.join(ROOT, "tests", "golden", "model_data", "SMARM2")
    vp = np.fromfile(os.path.join(data, "vp.smooth_20"), dtype=np.float32).reshape(shape) / 1000
    eps, delta, theta = 0.3 * (vp - 1.5) / 3.2, 0.2 * (vp - 1.5) / 3.2, 0.5 * (vp - 1.5) / 3.2
    model = Model(origin=(0, 0), spacing=(25., 25.), shape=shape, space_order=so, vp=vp, nbl=nbl,
                  epsilon=eps, delta=delta, theta=theta, bcs="damp")
    dt, nt, nshots, nrec = float(model.critical_dt), 1501, 2, 340
    src = np.empty((nshots, 1, 2), dtype=np.float32)
    src[:, 0, 0] = np.array([0.3, 0.7], dtype=np.float32) * model.domain_size[0]
    src[:, 0, 1] = 50.0
    rec = np.empty((nrec, 2), dtype=np.float32)
    rec[:, 0] = np.linspace(25.0, model.domain_size[0] - 25.0, nrec)
    rec[:, 1] = 50.0
    rng = np.random.default_rng(13)
    s = rng.standard_normal((nshots, nt, 1)).astype(np.float32)
    r = rng.standard_normal((nshots, nt, nrec)).astype(np.float32)
    plan = mifwi.Plan(model.grid.shape, model.spacing, model.grid.origin, so, nt, dt, 1, nrec, max_shots=nshots)
    plan.set_model(model.vp.data, model.damp.data)
    plan.tti_set_model(model.epsilon.data, model.delta.data, model.theta.data)
    plan.set_geometry(src, rec)
    fs = plan.tti_forward(s, wavelet_per_shot=True)
    ftr = plan.tti_adjoint(r)
    assert np.isfinite(fs).all() and np.abs(fs).max() > 0
    # 98 tiles per shot: three to four share a CU.  The loops must repeat bit for bit (they did not while the kernel
    # overwrote the data registers of its 16-byte stores too early, tests/test_gpu_tti_whole_loop.py)
    assert plan.stats()["bwd_kernel"].startswith("persist_tti2d_kernel")
    assert np.array_equal(fs, plan.tti_forward(s, wavelet_per_shot=True))
    assert np.array_equal(ftr, plan.tti_adjoint(r))
    for k in range(nshots):
        assert _pair_ok(fs[k], r[k], s[k], ftr[k], 3e-6), k
    dm = np.zeros(model.grid.shape, np.float32)
    dm[60:-60, 60:-60] = 0.01 * rng.standard_normal((300, 100)).astype(np.float32)
    jd = plan.tti_born(s, dm, wavelet_per_shot=True)
    plan.tti_forward(s, wavelet_per_shot=True, save=True)
    g = plan.tti_gradient(r)
    assert np.abs(jd).max() > 0
    for k in range(nshots):
        assert _pair_ok(jd[k], r[k], dm, g[k], 3e-6), k
    plan.close()
