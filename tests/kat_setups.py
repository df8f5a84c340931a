"""
Shared problem set-ups for the known-answer tests (KATs) of the reference.

Each builder restates, with numpy only, the inputs of one of the reference's own regression
scripts; nothing here computes wave propagation.
"""
import numpy as np


def circle_kat():
    """
    Inputs of seismic/inversion/fwi.py:13-56 (2-D circle model, 9 shots, 101 receivers).

    demo_model('circle-isotropic') is seismic/preset_models.py:231-251 with the default
    space_order=2 (preset_models.py:43), which only enters through critical_dt; the solver
    itself runs at space_order=4 (seismic/inversion/fwi.py:51).
    """
    shape = (101, 101)
    spacing = (10.0, 10.0)
    nbl = 40
    r = 15

    def circle(vp_circle, vp_background):
        v = np.empty(shape, dtype=np.float32)
        v[:] = vp_background
        a, b = shape[0] / 2, shape[1] / 2
        y, x = np.ogrid[-a:shape[0] - a, -b:shape[1] - b]
        v[x * x + y * y <= r * r] = vp_circle
        return v

    nshots, nrec = 9, 101
    src = np.empty((nshots, 2), dtype=np.float32)
    src[:, 0] = 20.0
    src[:, 1] = np.linspace(0.0, 1000, num=nshots)
    rec = np.empty((nrec, 2))
    rec[:, 1] = np.linspace(0, 1000.0, num=nrec)
    rec[:, 0] = 980.0
    return dict(shape=shape, spacing=spacing, nbl=nbl, space_order=4, model_space_order=2,
                vp_true=circle(3.0, 2.5), vp_init=circle(2.5, 2.5), t0=0.0, tn=1000.0, f0=0.010,
                src_positions=src, rec_positions=rec, vmin=2.0, vmax=3.5,
                kat=dict(objective=39113.0, gmin=-821.0, gmax=2442.0, objective_it5=3828.0, atol=10.0))


def circle_notebook_kat():
    """
    Inputs and printed output of seismic/tutorials/03_fwi.ipynb (cells 3-24): the set-up of circle_kat() with the
    shots at 30 m instead of 20 m (cell 16) and five gradient-descent steps whose objective values cell 24 prints.

    The printed values were recorded with a Devito release whose `critical_dt` was 0.42 * h / vp_max = 1.4 ms for
    this model (the 2-D constant of that era) rather than today's 2.041 ms from the stencil weights
    (seismic/model.py:338-370); the objective is a sum over time samples, so it scales with 1/dt.  With dt = 1.4
    (nt = 716) the restatement gives 57284.98, 35766.02, 21002.24, 11547.93, 5785.97 against the printed
    57285.10, 35766.90, 21002.93, 11548.23, 5786.19; with dt = 2.041 it gives 39293.  The time step is therefore
    passed as `dt=`, which the reference's model honours (seismic/model.py:355-370).
    """
    c = circle_kat()
    c["src_positions"][:, 0] = 30.0
    c["dt"] = np.float32(1.4)
    c["kat"] = dict(history=(57285.103209, 35766.902505, 21002.932951, 11548.234192, 5786.186187), atol_first=10.0,
                    rtol_rest=3e-4)
    return c


def layers3d_kat():
    """
    Inputs of seismic/acoustic/acoustic_example.py:26-40,75-79 (3-D layered model, f64, no
    free surface): demo_model('layers-isotropic') is seismic/preset_models.py:109-126 and
    setup_geometry is seismic/utils.py:12-47.
    """
    shape = (50, 50, 50)
    spacing = (20.0, 20.0, 20.0)
    nbl = 40
    nlayers = 3
    v = np.empty(shape, dtype=np.float64)
    v[:] = 1.5
    vp_i = np.linspace(1.5, 3.5, nlayers)
    for i in range(1, nlayers):
        v[..., i * int(shape[-1] / nlayers):] = vp_i[i]
    domain = tuple((n - 1) * s for n, s in zip(shape, spacing))
    src = np.empty((1, 3))
    src[0, :] = np.array(domain) * .5
    src[0, -1] = 0.0 + spacing[-1]
    recx = np.linspace(0.0, domain[0], shape[0])
    recy = np.linspace(0.0, domain[1], shape[1])
    rec = np.empty((shape[0] * shape[1], 3))
    rec[:, 0] = np.array([recx[i] for i in range(shape[0]) for j in range(shape[1])])
    rec[:, 1] = np.array([recy[j] for i in range(shape[0]) for j in range(shape[1])])
    rec[:, -1] = 0.0 + 2 * spacing[-1]
    return dict(shape=shape, spacing=spacing, nbl=nbl, space_order=4, vp=v, t0=0.0, tn=1000.0, f0=0.010,
                src_positions=src, rec_positions=rec, kat=dict(norm_rec=459.1678, rtol=1e-3))


def hankel_kat():
    """
    The analytic comparison of seismic/acoustic/accuracy.ipynb (cells 2-16, 23): constant 1.5 km/s medium, source
    at (200, 200) m, one receiver at (260, 260) m, Ricker of 90 Hz delayed by 1.5 / f0, dt = 0.1 ms, nt = 1501.
    Returns the inputs, the analytic trace of cells 12-13 (2-D Green's function -i pi H0^(2)(k r) times the source
    spectrum, scaled by the square of the fine grid's spacing) and the numbers the notebook prints.
    """
    from scipy.special import hankel2
    c0, f0, nt, dt = 1.5, 0.09, 1501, 0.1
    src, rec = np.array([[200.0, 200.0]]), np.array([[260.0, 260.0]])
    r = float(np.sqrt(((rec - src) ** 2).sum()))
    # cell 13: the transform runs over 0 .. 3000 ms in 30001 samples, of which the first 1501 are kept
    n_long, t_long = 30001, 3000.0
    dt_long = t_long / (n_long - 1)
    t = np.linspace(-1.5 / f0, t_long - 1.5 / f0, int(t_long / dt_long))          # cell 12, ricker()
    tt = (np.pi ** 2) * (f0 ** 2) * (t ** 2)
    spectrum = np.fft.fft((1.0 - 2.0 * tt) * np.exp(-tt))
    nf = int(n_long / 2 + 1)
    freq = np.arange(nf) / t_long
    u_f = np.zeros(nf, dtype=complex)
    a = np.arange(1, nf - 1)
    u_f[a] = -1j * np.pi * hankel2(0.0, 2 * np.pi * freq[a] / c0 * r) * spectrum[a]
    fine_spacing = 0.5
    analytic = (1.0 / (2.0 * np.pi) * np.real(np.fft.ifft(u_f, n_long)))[:nt] * fine_spacing ** 2
    printed_errors = {   # cell 23 output: (space order, points per side, spacing) -> RMS error against the analytic trace
        (2, 201, 2.0): 0.0037682796393557704, (2, 161, 2.5): 0.005812459176798742, (2, 101, 4.0): 0.013011149171741464,
        (4, 201, 2.0): 0.0003400355796236851, (4, 161, 2.5): 0.0009414704657502754, (4, 101, 4.0): 0.00634686261474598,
        (6, 201, 2.0): 4.5490650020112585e-05, (6, 161, 2.5): 0.0002068743812472065, (6, 101, 4.0): 0.003742867493048817,
        (8, 201, 2.0): 4.1758297084644546e-05, (8, 161, 2.5): 7.16374344942322e-05, (8, 101, 4.0): 0.0025733941924053825,
        (10, 201, 2.0): 4.4878701635970325e-05, (10, 161, 2.5): 4.998218488476813e-05, (10, 101, 4.0): 0.0019484834503388081}
    return dict(c0=c0, f0=f0, t0w=1.5 / f0, nt=nt, dt=dt, tn=dt * (nt - 1), src=src, rec=rec, nbl=40, analytic=analytic,
                fine=dict(shape=(801, 801), spacing=(fine_spacing, fine_spacing), space_order=8, dtype=np.float64),
                kat=dict(numeric_min=-5.349877e-03, numeric_max=+8.529867e-03,        # cell 14 output
                         analytic_min=-5.322232e-03, analytic_max=+8.543911e-03,
                         rms_error=1.1265077536675204e-05,                            # cell 16 output
                         errors=printed_errors, rtol=0.05))


def pml_notebook_kat():
    """Inputs of seismic/abc_methods/03_pml.ipynb (cells 12-38, 42-43, 58-63): two layers of 1.5 and 2.5 km/s on
    101 x 101 nodes at 10 m, a 20-node layer left, right and below (none at the top): 141 x 121 nodes from (-200, 0) m;
    CFL 0.4 time step, 627 samples over 1 s; Ricker of 10 Hz at (500, 10) m, 141 receivers on the nodes of the row at
    10 m; float64, space order 2.  Cell 96 asserts `np.isclose(np.linalg.norm(rec.data), 990, rtol=1)`."""
    nx0, nz0, npml, h = 101, 101, 20, 10.0
    n1, n2 = nx0 + 2 * npml, nz0 + npml
    x0, x1, z0, z1 = -npml * h, 1000.0 + npml * h, 0.0, 1000.0 + npml * h
    pzm = 50                                        # node of the interface, z = 500 m (cell 26)
    v0 = np.empty((n1, n2))
    v0[:, :pzm], v0[:, pzm:] = 1.5, 2.5
    v1 = np.empty((n1, n2))                         # at the staggered points; last row / column repeat (cells 51-53)
    v1[:, :pzm], v1[:, pzm:] = 1.5, 2.5
    tn, cfl = 1000.0, 0.4
    dtmax = h * cfl / v0.max()
    ntmax = int(tn / dtmax) + 1
    dt = tn / ntmax
    nt = ntmax + 1                                  # samples of the time axis (cell 33)
    rec = np.empty((n1, 2))
    rec[:, 0] = np.linspace(x0, x1, n1)
    rec[:, 1] = h
    return dict(shape=(n1, n2), spacing=(h, h), origin=(x0, z0), vp=v0, vp_staggered=v1, npml=npml, quibar=0.05,
                sides=(True, True, False, True), dt=dt, nt=nt, f0=0.01, src=np.array([[500.0, h]]), rec=rec,
                space_order=2, kat=dict(norm_rec=990.0, rtol=1.0))


def drp_notebook_kat():
    """Inputs of the standard (Taylor-weight) run of seismic/tutorials/07_DRP_schemes.ipynb (cells 15-23): 201 x 201
    nodes at 10 m, 1.5 km/s above node 121 in z and 4.0 km/s below, 10 absorbing nodes, space order 20, float32;
    501 samples of 1 ms, Ricker of 15 Hz at (1000, 800) m, no receivers.  Cell 40 asserts
    `np.isclose(np.linalg.norm(u.data[-1]), 139.108, atol=0, rtol=1e-4)`; with three circular time levels
    `u.data[-1]` is the level of time index 500 (500 % 3 == 2), which the one further step the notebook's operator
    takes (to time index 501, buffer 0) leaves as it was."""
    shape, h = (201, 201), 10.0
    v = np.empty(shape, dtype=np.float32)
    v[:, :121], v[:, 121:] = 1.5, 4.0
    return dict(shape=shape, spacing=(h, h), origin=(0.0, 0.0), nbl=10, space_order=20, vp=v, dt=1.0, t0=0.0, tn=500.0,
                nt=501, f0=0.015, src=np.array([[1000.0, 800.0]]), kat=dict(norm_last=139.108, rtol=1e-4))


def modelling_notebook_kat():
    """Inputs of seismic/tutorials/01_modelling.ipynb (cells 5-20): 101 x 101 nodes at 10 m, 1.5 km/s above node 51
    in z and 2.5 km/s below, 10 absorbing nodes, space order 2, float32; the model's own time step over 1 s, Ricker
    of 10 Hz at (500, 20) m, 101 receivers at 20 m depth.  Cell 23 asserts
    `np.isclose(np.linalg.norm(rec.data), 370, rtol=1)`, i.e. a norm in (0, 740]: the order of magnitude only.  The
    notebook's operator records `u.forward`, one sample later than the solver's `u`; the norm does not see that."""
    shape, h = (101, 101), 10.0
    v = np.empty(shape, dtype=np.float32)
    v[:, :51], v[:, 51:] = 1.5, 2.5
    rec = np.empty((101, 2))
    rec[:, 0] = np.linspace(0.0, 1000.0, 101)
    rec[:, 1] = 20.0
    return dict(shape=shape, spacing=(h, h), origin=(0.0, 0.0), nbl=10, space_order=2, vp=v, t0=0.0, tn=1000.0,
                f0=0.010, src=np.array([[500.0, 20.0]]), rec=rec, kat=dict(norm_rec=370.0, rtol=1.0))
