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
