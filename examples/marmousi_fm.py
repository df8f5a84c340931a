"""Forward modelling of the Marmousi / Marmousi-II shot gathers on the GPU: the workflow of the
reference's marmousi_fm.py:30-113 and marmousi2_fm.py (observed data in the true model, synthetic data
in the smoothed model, direct wave in a water model; one raw float32 file (nt, nrec) per shot under
<odir>/data), without the plotting.

    python examples/marmousi_fm.py --model SMARM2 --nsrc 21 --odir /tmp/smarm2_fm
"""
import argparse
import os
import sys
from time import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _common as common                                 # noqa: E402
from fwi import fm_multi                                  # noqa: E402

parser = argparse.ArgumentParser(description='Forward modelling of shot gathers')
parser.add_argument('--model', default='SMARMN', choices=sorted(common.CONFIGS))
parser.add_argument('--data', default=None, help='directory holding vp.true and vp.smooth_20 (default: the fixtures)')
parser.add_argument('--odir', type=str, default=None, help='default ./result/<model>_fm')
parser.add_argument('--filter', type=int, default=0)
parser.add_argument('--nsrc', type=int, default=21)
parser.add_argument('--tn', type=float, default=None, help='record length in ms (default 4500)')


def main(argv=None):
    args = parser.parse_args(argv)
    odir = args.odir or os.path.join('.', 'result', args.model + '_fm')
    os.makedirs(os.path.join(odir, 'data'), exist_ok=True)
    cfg, true_vp, smooth_vp, model_of = common.load(args.model, args.data)
    shape = cfg['shape']
    models = (model_of(true_vp), model_of(smooth_vp), model_of(np.ones(shape) * 1.5))
    src, rec = common.acquisition(models[0], args.nsrc, shape)
    geoms = common.geometries(models, src, rec, args.tn or cfg['tn']['fm'], cfg['dt'], args.filter)
    tic = time()
    gathers = {name: fm_multi(g, save=False) for name, g in zip(('obs', 'syn', 'direct'), geoms)}
    toc = time()
    for name, shots in gathers.items():
        for i, shot in enumerate(shots):
            np.asarray(shot.data, dtype=np.float32).tofile(os.path.join(odir, 'data', name + str(i)))
    nt, nrec = gathers['obs'][0].data.shape
    print('%d shots x 3 models, (nt, nrec) = (%d, %d), modelled in %.2f s -> %s'
          % (args.nsrc, nt, nrec, toc - tic, os.path.join(odir, 'data')))
    return gathers


if __name__ == '__main__':
    main()
