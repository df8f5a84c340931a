"""
Pins the CPU oracle against the reference's own known-answer tests:
  * seismic/inversion/fwi.py:95-97,121     objective 39113, min/max grad -821/2442, 5th GD objective 3828 (atol 10)
  * seismic/acoustic/acoustic_example.py:75-79   ||rec|| = 459.1678 (f64, no free surface, rtol 1e-3)
  * seismic/tutorials/03_fwi.ipynb cell 24     objective history 57285.10 ... 5786.19 of five descent steps
  * seismic/acoustic/accuracy.ipynb cells 14, 16, 23   extrema of the fine-grid trace, its RMS distance from the
    analytic (Hankel function) solution, and the RMS errors of space orders 2 .. 10 on three coarser grids
plus unit pins of the constants (FD weights vs sympy, CFL dt, sponge centre value of
seismic/test_seismic_utils.py:17-36).
"""
import numpy as np
import pytest

from oracle import oracle as orc
from kat_setups import circle_kat, circle_notebook_kat, drp_notebook_kat, hankel_kat, layers3d_kat, modelling_notebook_kat


def test_fd_weights_match_sympy():
    from sympy import finite_diff_weights
    for R in range(1, 9):
        ref = np.array([float(x) for x in finite_diff_weights(2, range(-R, R + 1), 0)[-1][-1]])[R:]
        np.testing.assert_allclose(orc.fd2_weights(R), ref, rtol=1e-13, atol=0)


def test_cfl_dt_values():
    # seismic/model.py:351-365; the 2.041 value is what makes nt = 491 in the circle KAT
    assert orc.critical_dt(3.0, (10., 10.), 2) == np.float32(2.041)
    from sympy import finite_diff_weights
    c = finite_diff_weights(2, range(-8, 9), 0)[-1][-1]
    ref = np.sqrt(4 / float(2 * sum(abs(x) for x in c)))
    assert np.isclose(orc.cfl_coeff(8, 2), ref, rtol=1e-12)


@pytest.mark.parametrize("nbl", [20, 0])
def test_sponge_centre_and_shape(nbl):
    # seismic/test_seismic_utils.py:17-36, "damp" flavour: centre value 0, padded shape
    d = orc.sponge((21 + 2 * nbl, 21 + 2 * nbl), nbl, (10., 10.))
    assert d.shape == (21 + 2 * nbl, 21 + 2 * nbl)
    assert d[d.shape[0] // 2, d.shape[1] // 2] == 0
    if nbl:
        assert d[0, 0] > d[nbl - 1, nbl - 1] > 0
        # corner = sum of the two 1-D profiles, edge midpoint = one of them
        assert np.isclose(d[0, 0], 2 * d[0, d.shape[1] // 2], rtol=1e-6)


def test_time_axis_and_ricker():
    # seismic/test_seismic_utils.py:44-56: tn=250, dt=1 -> 251 samples
    nt, tv = orc.time_axis(0.0, 250.0, 1.0)
    assert nt == 251 and tv[-1] == 250.0
    w = orc.ricker(0.010, tv)
    assert np.isclose(w[100], 1.0)        # peak at t = 1/f0


def _circle_fwi_gradient(c, prob, vp_in, wavelet, dt):
    grad = np.zeros(prob.shape_pad, dtype=np.float32)
    objective = 0.0
    for i in range(c["src_positions"].shape[0]):
        d_obs, _ = prob.forward(c["src_positions"][i], wavelet, c["rec_positions"], dt=dt)
        d_syn, u0 = prob.forward(c["src_positions"][i], wavelet, c["rec_positions"], dt=dt, save=True, vp=vp_in)
        residual = d_syn - d_obs
        objective += .5 * np.linalg.norm(residual.ravel().astype(np.float64)) ** 2
        prob.gradient(c["rec_positions"], residual, u0, dt=dt, vp=vp_in, grad=grad)
    return objective, grad


def test_kat_circle_fwi():
    """seismic/inversion/fwi.py:59-121."""
    c = circle_kat()
    prob = orc.Problem(c["vp_true"], c["spacing"], c["nbl"], c["space_order"],
                       model_space_order=c["model_space_order"])
    dt = prob.critical_dt
    assert dt == np.float32(2.041)
    nt, tv = orc.time_axis(c["t0"], c["tn"], float(dt))
    assert nt == 491
    wavelet = orc.ricker(c["f0"], tv).astype(np.float32).reshape(nt, 1)
    vp0 = orc.pad_vp(c["vp_init"], c["nbl"]).astype(np.float32)

    ff, g = _circle_fwi_gradient(c, prob, vp0, wavelet, dt)
    k = c["kat"]
    print("circle KAT:", ff, g.min(), g.max())
    assert np.isclose(ff, k["objective"], atol=k["atol"], rtol=0)
    assert np.isclose(g.min(), k["gmin"], atol=k["atol"], rtol=0)
    assert np.isclose(g.max(), k["gmax"], atol=k["atol"], rtol=0)

    history = []
    for it in range(5):
        phi, direction = _circle_fwi_gradient(c, prob, vp0, wavelet, dt)
        history.append(phi)
        alpha = np.float32(.05 / direction.max())
        # seismic/inversion/inversion_utils.py:28-39, on the padded grid
        vp0 = np.clip(vp0 + alpha * direction, c["vmin"], c["vmax"]).astype(np.float32)
    print("circle KAT history:", history)
    assert np.isclose(history[-1], k["objective_it5"], atol=k["atol"], rtol=0)


def test_kat_circle_notebook_history():
    """seismic/tutorials/03_fwi.ipynb cell 24: objective of five gradient-descent steps, shots at 30 m."""
    c = circle_notebook_kat()
    prob = orc.Problem(c["vp_true"], c["spacing"], c["nbl"], c["space_order"], dt=c["dt"])
    dt = prob.critical_dt
    assert dt == np.float32(1.4)
    nt, tv = orc.time_axis(c["t0"], c["tn"], float(dt))
    assert nt == 716
    wavelet = orc.ricker(c["f0"], tv).astype(np.float32).reshape(nt, 1)
    vp0 = orc.pad_vp(c["vp_init"], c["nbl"]).astype(np.float32)
    history = []
    for it in range(5):
        phi, direction = _circle_fwi_gradient(c, prob, vp0, wavelet, dt)
        history.append(phi)
        alpha = np.float32(.05 / direction.max())
        vp0 = np.clip(vp0 + alpha * direction, c["vmin"], c["vmax"]).astype(np.float32)
    print("notebook KAT history:", history)
    k = c["kat"]
    assert np.isclose(history[0], k["history"][0], atol=k["atol_first"], rtol=0)
    np.testing.assert_allclose(history[1:], k["history"][1:], rtol=k["rtol_rest"], atol=0)


@pytest.mark.slow
def test_kat_layers3d_forward_f64():
    """seismic/acoustic/acoustic_example.py:75-79, (fs=False, 459.1678, float64)."""
    c = layers3d_kat()
    prob = orc.Problem(c["vp"], c["spacing"], c["nbl"], c["space_order"], dtype=np.float64)
    dt = prob.critical_dt
    nt, tv = orc.time_axis(c["t0"], c["tn"], float(dt))
    wavelet = orc.ricker(c["f0"], tv).reshape(nt, 1)
    rec, _ = prob.forward(c["src_positions"], wavelet, c["rec_positions"], dt=dt)
    nrm = np.linalg.norm(rec.ravel())
    print("layers3d KAT: dt", dt, "nt", nt, "norm", nrm)
    assert np.isclose(nrm, c["kat"]["norm_rec"], rtol=c["kat"]["rtol"], atol=0)


@pytest.mark.slow
def test_kat_layers3d_forward_f32_free_surface():
    """seismic/acoustic/acoustic_example.py:75-79, (fs=True, 369.955, float32)."""
    c = layers3d_kat()
    prob = orc.Problem(c["vp"].astype(np.float32), c["spacing"], c["nbl"], c["space_order"], dtype=np.float32,
                       fs=True)
    assert prob.shape_pad == (130, 130, 90) and prob.origin_pad[-1] == 0.0
    dt = prob.critical_dt
    nt, tv = orc.time_axis(c["t0"], c["tn"], float(dt))
    wavelet = orc.ricker(c["f0"], tv).reshape(nt, 1)
    rec, _ = prob.forward(c["src_positions"], wavelet, c["rec_positions"], dt=dt)
    nrm = np.linalg.norm(rec.ravel().astype(np.float64))
    print("layers3d free-surface KAT: dt", dt, "nt", nt, "norm", nrm)
    assert np.isclose(nrm, 369.955, rtol=1e-3, atol=0)


def test_born_is_the_derivative_of_forward_and_adjoint_of_gradient():
    """Linearisation check of the Born restatement (operators.py:228-273): F(m + eps dm) - F(m)
    -> eps J dm; and <J dm, r> against <dm, J^T r> with J^T = the gradient operator
    (operators.py:183-225), which the reference's solver pairs as jacobian / jacobian_adjoint
    (wavesolver.py:153-246)."""
    rng = np.random.default_rng(5)
    shape, spacing, nbl, so = (41, 37), (10.0, 10.0), 12, 4
    vp = (1.8 + 0.4 * rng.random(shape)).astype(np.float64)
    from scipy.ndimage import gaussian_filter
    vp = gaussian_filter(vp, 3)
    prob = orc.Problem(vp, spacing, nbl, so, dtype=np.float64)
    dt = float(prob.critical_dt)
    nt, tv = orc.time_axis(0.0, 400.0, dt)
    w = orc.ricker(0.02, tv).reshape(nt, 1)
    src = np.array([[203.0, 31.0]])
    rec = np.stack([np.linspace(15.0, 385.0, 19), np.full(19, 27.0)], axis=1)
    dm_phys = gaussian_filter(rng.standard_normal(shape), 2) * 0.02
    dm = np.zeros(prob.shape_pad)
    dm[nbl:-nbl, nbl:-nbl] = dm_phys
    d0, u = prob.forward(src, w, rec, dt=dt, save=True)
    lin = prob.born(src, w, rec, dm, dt=dt)
    # finite difference in m = 1/vp^2
    m = 1.0 / prob.vp ** 2
    errs = []
    for eps in (1e-2, 1e-3):
        d1, _ = prob.forward(src, w, rec, dt=dt, vp=1.0 / np.sqrt(m + eps * dm))
        errs.append(np.linalg.norm((d1 - d0) / eps - lin) / np.linalg.norm(lin))
    print("born linearisation errors", errs)
    assert errs[1] < 0.2 * errs[0] and errs[1] < 5e-3
    # adjoint pair
    r = rng.standard_normal(lin.shape) * (np.arange(nt)[:, None] > 2)
    g = prob.gradient(rec, r, u, dt=dt)
    lhs, rhs = float(np.sum(lin * r)), float(np.sum(g * dm))
    print("<J dm, r> =", lhs, " <dm, J^T r> =", rhs)
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), abs(rhs))


def _hankel_trace(k, shape, spacing, so, dtype):
    prob = orc.Problem(np.full(shape, k["c0"], dtype=dtype), spacing, k["nbl"], so, dtype=dtype, dt=k["dt"])
    tv = np.linspace(0.0, k["tn"], k["nt"])
    w = orc.ricker(k["f0"], tv, t0=k["t0w"]).astype(dtype).reshape(k["nt"], 1)
    return prob.forward(k["src"], w, k["rec"], dt=k["dt"])[0][:, 0].astype(np.float64)


@pytest.mark.slow
def test_kat_analytic_hankel_fine_grid():
    """accuracy.ipynb cells 7-16: 801 x 801 points of 0.5 m, float64, space order 8."""
    k = hankel_kat()
    a, kat = k["analytic"], k["kat"]
    assert np.isclose(a.min(), kat["analytic_min"], rtol=1e-6) and np.isclose(a.max(), kat["analytic_max"], rtol=1e-6)
    f = k["fine"]
    r = _hankel_trace(k, f["shape"], f["spacing"], f["space_order"], f["dtype"])
    print("fine grid: min %.6e max %.6e" % (r.min(), r.max()))
    assert np.isclose(r.min(), kat["numeric_min"], rtol=1e-5) and np.isclose(r.max(), kat["numeric_max"], rtol=1e-5)
    err = np.linalg.norm(a[:-1] - r[:-1], 2) / np.sqrt(k["nt"])          # cell 16
    print("rms error", err)
    assert np.isclose(err, kat["rms_error"], rtol=kat["rtol"])


def test_kat_analytic_hankel_space_orders():
    """accuracy.ipynb cell 23: float32, space orders 2 .. 10 on grids of 2, 2.5 and 4 m."""
    k = hankel_kat()
    a = k["analytic"] / k["fine"]["spacing"][0] ** 2
    for (so, nn, h), want in sorted(k["kat"]["errors"].items()):
        r = _hankel_trace(k, (nn, nn), (h, h), so, np.float32)
        err = np.linalg.norm(r[:-1] / h ** 2 - a[:-1], 2) / np.sqrt(k["nt"] - 1)
        assert np.isclose(err, want, rtol=k["kat"]["rtol"]), (so, nn, err, want)


def test_kat_drp_notebook_wavefield_norm():
    """07_DRP_schemes.ipynb cell 40: the norm of the last wavefield of the standard space-order-20 run, 139.108 to
    1e-4."""
    k = drp_notebook_kat()
    prob = orc.Problem(k["vp"], k["spacing"], k["nbl"], k["space_order"], dtype=np.float32, dt=k["dt"])
    tv = np.linspace(k["t0"], k["tn"], k["nt"])
    w = orc.ricker(k["f0"], tv).astype(np.float32).reshape(k["nt"], 1)
    levels = prob.forward(k["src"], w, np.zeros((0, 2)), dt=k["dt"], return_last=True)[-1]
    norm = float(np.linalg.norm(levels[(k["nt"] - 1) % 3]))
    print("norm of the level of time index %d: %.4f" % (k["nt"] - 1, norm))
    assert np.isclose(norm, k["kat"]["norm_last"], atol=0, rtol=k["kat"]["rtol"])


def test_kat_modelling_notebook_record_norm():
    """01_modelling.ipynb cell 23: `np.isclose(np.linalg.norm(rec.data), 370, rtol=1)`.  The assertion only fixes the
    order of magnitude; the restatement gives 369.97, so the 370 is the rounded value and is checked to 1e-3 here."""
    k = modelling_notebook_kat()
    prob = orc.Problem(k["vp"], k["spacing"], k["nbl"], k["space_order"], dtype=np.float32)
    dt = prob.critical_dt
    nt = int(np.ceil((k["tn"] - k["t0"] + dt) / dt))
    tv = np.linspace(k["t0"], k["t0"] + dt * (nt - 1), nt)
    w = orc.ricker(k["f0"], tv).astype(np.float32).reshape(nt, 1)
    norm = float(np.linalg.norm(prob.forward(k["src"], w, k["rec"], dt=dt)[0]))
    print("norm of the record: %.4f" % norm)
    assert np.isclose(norm, k["kat"]["norm_rec"], rtol=k["kat"]["rtol"])
    assert np.isclose(norm, k["kat"]["norm_rec"], rtol=1e-3)
