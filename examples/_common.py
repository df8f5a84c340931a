"""Shared set-up of the example drivers: the two Marmousi configurations of the reference
(marmousi_fwi.py:62-117 / marmousi_fm.py:52-98 for SMARMN, marmousi2_fwi.py:61-118 / marmousi2_fm.py for
SMARM2) as data, and the three acquisition geometries (true, start and water model) built from them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'devito-fwi_amd'))

from seismic import Model, AcquisitionGeometry          # noqa: E402
from fwi import Filter                                   # noqa: E402

CONFIGS = {
    # raster directory, shape, dt (ms), water columns masked, default record length (ms) per driver kind
    'SMARMN': dict(shape=(300, 106), dt=2.95, water=7, tn={'fwi': 4000., 'fm': 4500.}, vmax=5.2),
    'SMARM2': dict(shape=(340, 140), dt=3.0, water=15, tn={'fwi': 4500., 'fm': 4500.}, vmax=5.0),
}
SPACING, ORIGIN, SPACE_ORDER, NBL, F0 = (30., 30.), (0, 0), 8, 40, 0.007


def default_data_dir(name):
    return os.path.join(ROOT, 'tests', 'golden', 'model_data', name)


def load(name, data_dir=None, space_order=SPACE_ORDER):
    """(config, true_vp, smooth_vp, model_of) for one of the Marmousi rasters (km/s, (nx, nz))."""
    cfg = CONFIGS[name]
    d = data_dir or default_data_dir(name)
    true_vp = np.fromfile(os.path.join(d, 'vp.true'), dtype=np.float32).reshape(cfg['shape']) / 1000
    smooth_vp = np.fromfile(os.path.join(d, 'vp.smooth_20'), dtype=np.float32).reshape(cfg['shape']) / 1000

    def model_of(vp):
        return Model(origin=ORIGIN, spacing=SPACING, shape=cfg['shape'], space_order=space_order, vp=vp,
                     nbl=NBL, fs=False, dt=cfg['dt'])
    return cfg, true_vp, smooth_vp, model_of


def acquisition(model, nsrc, shape):
    """Sources spread over the surface, one receiver per grid column, both two cells deep."""
    src = np.empty((nsrc, 2))
    src[:, 0] = np.linspace(0, model.domain_size[0], num=nsrc)
    src[:, 1] = 2 * SPACING[0]
    rec = np.empty((shape[0], 2))
    rec[:, 0] = np.linspace(SPACING[0], model.domain_size[0] - SPACING[0], num=shape[0])
    rec[:, 1] = 2 * SPACING[0]
    return src, rec


def geometries(models, src, rec, tn, dt, use_filter):
    filt = Filter(filter_type='highpass', freqmin=3, corners=6, df=1000 / dt) if use_filter else None
    return [AcquisitionGeometry(m, rec, src, 0., tn, f0=F0, src_type='Ricker', filter=filt) for m in models]
