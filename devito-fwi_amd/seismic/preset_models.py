"""
Preset velocity models for demonstrations and tests: the isotropic acoustic presets of the
reference's seismic/preset_models.py (`constant-isotropic` :76-81, `layers-isotropic` :109-126,
`circle-isotropic` :231-251) with the same defaults (:43-51).
"""
import numpy as np

from .model import SeismicModel

__all__ = ['demo_model']


def demo_model(preset, **kwargs):
    space_order = kwargs.pop('space_order', 2)
    shape = kwargs.pop('shape', (101, 101))
    spacing = kwargs.pop('spacing', tuple([10. for _ in shape]))
    origin = kwargs.pop('origin', tuple([0. for _ in shape]))
    nbl = kwargs.pop('nbl', 10)
    dtype = kwargs.pop('dtype', np.float32)
    vp = kwargs.pop('vp', 1.5)
    nlayers = kwargs.pop('nlayers', 3)
    fs = kwargs.pop('fs', False)
    # The reference's presets build upstream Devito's SeismicModel (preset_models.py:5), which has no `dt`
    # argument: a `dt=` given to demo_model (circle_fwi.py:70-75) does not change the CFL time step there.
    kwargs.pop('dt', None)
    name = preset.lower()

    if name == 'constant-isotropic':
        return SeismicModel(space_order=space_order, vp=vp, origin=origin, shape=shape, dtype=dtype,
                            spacing=spacing, nbl=nbl, fs=fs, **kwargs)

    if name == 'layers-isotropic':
        vp_top = kwargs.pop('vp_top', 1.5)
        vp_bottom = kwargs.pop('vp_bottom', 3.5)
        v = np.full(shape, vp_top, dtype=dtype)
        levels = np.linspace(vp_top, vp_bottom, nlayers)
        thickness = int(shape[-1] / nlayers)
        for i in range(1, nlayers):
            v[..., i * thickness:] = levels[i]
        return SeismicModel(space_order=space_order, vp=v, origin=origin, shape=shape, dtype=dtype,
                            spacing=spacing, nbl=nbl, bcs="damp", fs=fs, **kwargs)

    if name == 'circle-isotropic':
        vp_circle = kwargs.pop('vp_circle', 3.0)
        vp_background = kwargs.pop('vp_background', 2.5)
        r = kwargs.pop('r', 15)
        assert len(shape) == 2
        v = np.full(shape, vp_background, dtype=dtype)
        a, b = shape[0] / 2, shape[1] / 2
        y, x = np.ogrid[-a:shape[0] - a, -b:shape[1] - b]
        v[x * x + y * y <= r * r] = vp_circle
        return SeismicModel(space_order=space_order, vp=v, origin=origin, shape=shape, dtype=dtype,
                            spacing=spacing, nbl=nbl, bcs="damp", fs=fs, **kwargs)

    if name == 'constant-tti':
        # preset_models.py:93-107 of the reference
        v = np.full(shape, 1.5, dtype=dtype)
        phi = .35 * np.ones(shape, dtype=dtype) if len(shape) > 2 else None
        return SeismicModel(space_order=space_order, vp=v, origin=origin, shape=shape, dtype=dtype,
                            spacing=spacing, nbl=nbl, epsilon=.3 * np.ones(shape, dtype=dtype),
                            delta=.2 * np.ones(shape, dtype=dtype), theta=.7 * np.ones(shape, dtype=dtype),
                            phi=phi, bcs="damp", **kwargs)

    if name in ('layers-tti', 'layers-tti-noazimuth'):
        # preset_models.py:197-229 of the reference: Thomsen parameters and angles scale with vp - 1.5
        vp_top = kwargs.pop('vp_top', 1.5)
        vp_bottom = kwargs.pop('vp_bottom', 3.5)
        smooth = kwargs.pop('smooth', False)
        v = np.full(shape, vp_top, dtype=dtype)
        levels = np.linspace(vp_top, vp_bottom, nlayers)
        thickness = int(shape[-1] / nlayers)
        for i in range(1, nlayers):
            v[..., i * thickness:] = levels[i]
        azimuth = len(shape) > 2 and name != 'layers-tti-noazimuth'
        model = SeismicModel(space_order=space_order, vp=v, origin=origin, shape=shape, dtype=dtype,
                             spacing=spacing, nbl=nbl, epsilon=.3 * (v - 1.5), delta=.2 * (v - 1.5),
                             theta=.5 * (v - 1.5), phi=.25 * (v - 1.5) if azimuth else None, bcs="damp", **kwargs)
        if smooth:
            model.smooth(('epsilon', 'delta', 'theta', 'phi') if azimuth else ('epsilon', 'delta', 'theta'))
        return model

    raise NotImplementedError("preset %r: only the acoustic (isotropic and TTI) presets are available" % preset)
