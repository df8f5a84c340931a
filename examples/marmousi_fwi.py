"""Marmousi FWI with L-BFGS on the GPU: the workflow of the reference's marmousi_fwi.py:36-190
(true/smooth/water models, observed and direct-wave data, bathymetry mask, bounds, minimize.run),
without the plotting. Start one process per GPU with torch.distributed to shard the shots.

    python examples/marmousi_fwi.py --maxiter 5 --nsrc 8 --odir /tmp/smarmn
"""
import argparse
import os
import sys
from time import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'devito-fwi_amd'))

from seismic import Model, AcquisitionGeometry          # noqa: E402
from fwi import Filter, fm_multi, fwi_obj_multi          # noqa: E402
from misfit import least_square, qWasserstein            # noqa: E402
from optimize import LBFGS, NLCG, SteepestDescent        # noqa: E402
from minimize import minimize                            # noqa: E402

parser = argparse.ArgumentParser(description='Full waveform inversion')
parser.add_argument('--data', default=os.path.join(ROOT, 'tests', 'golden', 'model_data', 'SMARMN'))
parser.add_argument('--misfit', type=int, default=0, choices=[0, 1], help='0=least squares, 1=trace-wise W2')
parser.add_argument('--method', default='LBFGS', choices=['LBFGS', 'NLCG', 'SteepestDescent'])
parser.add_argument('--precond', type=int, default=1)
parser.add_argument('--odir', type=str, default='./result/SMARMN')
parser.add_argument('--bathy', type=int, default=1)
parser.add_argument('--filter', type=int, default=0)
parser.add_argument('--resample', type=float, default=0.)
parser.add_argument('--ftol', type=float, default=1e-5)
parser.add_argument('--gtol', type=float, default=1e-10)
parser.add_argument('--nsrc', type=int, default=29)
parser.add_argument('--maxiter', type=int, default=200)
parser.add_argument('--steplen', type=float, default=0.1)
parser.add_argument('--maxls', type=int, default=5)
parser.add_argument('--tn', type=float, default=4000.)
parser.add_argument('--secant-pairs', type=int, default=0,
                    help='1: store g - g_previous in the L-BFGS memory instead of the reference\'s choice')


def main(argv=None):
    args = parser.parse_args(argv)
    os.makedirs(args.odir, exist_ok=True)
    shape, spacing, origin = (300, 106), (30., 30.), (0, 0)
    space_order, nbl, dt = 8, 40, 2.95
    true_vp = np.fromfile(os.path.join(args.data, 'vp.true'), dtype=np.float32).reshape(shape) / 1000
    smooth_vp = np.fromfile(os.path.join(args.data, 'vp.smooth_20'), dtype=np.float32).reshape(shape) / 1000
    bathy_mask = np.ones(shape, dtype=np.float32)
    bathy_mask[:, :7] = 0
    if not args.bathy:
        bathy_mask = None

    def model_of(vp):
        return Model(origin=origin, spacing=spacing, shape=shape, space_order=space_order, vp=vp,
                     nbl=nbl, fs=False, dt=dt)
    true_model, init_model, water_model = model_of(true_vp), model_of(smooth_vp), model_of(np.ones(shape) * 1.5)

    src = np.empty((args.nsrc, 2))
    src[:, 0] = np.linspace(0, true_model.domain_size[0], num=args.nsrc)
    src[:, 1] = 2 * spacing[0]
    rec = np.empty((shape[0], 2))
    rec[:, 0] = np.linspace(spacing[0], true_model.domain_size[0] - spacing[0], num=shape[0])
    rec[:, 1] = 2 * spacing[0]
    filt = Filter(filter_type='highpass', freqmin=3, corners=6, df=1000 / dt) if args.filter else None

    def geometry_of(model):
        return AcquisitionGeometry(model, rec, src, 0., args.tn, f0=0.007, src_type='Ricker', filter=filt)
    geometry1, geometry0, geometry2 = geometry_of(true_model), geometry_of(init_model), geometry_of(water_model)
    geometry0.resample(args.resample if args.resample else dt)

    obs = fm_multi(geometry1, save=False)
    direct_wave = fm_multi(geometry2, save=False)
    misfit_func = least_square if args.misfit == 0 else qWasserstein(gamma=1.01, method='1d')

    bounds = [1.0 / 5.2 ** 2, 1.0 / 1.5 ** 2]
    m0 = 1. / (smooth_vp.reshape(-1).astype(np.float64)) ** 2
    log = os.path.join(args.odir, 'log' + str(args.misfit))
    kw = dict(ls_method='Bracket', step_len_init=args.steplen, max_ls=args.maxls, log_path=log)
    if args.method == 'LBFGS':
        optimizer = LBFGS(memory=10, pairs='secant' if args.secant_pairs else 'reference', **kw)
    else:
        optimizer = {'NLCG': NLCG, 'SteepestDescent': SteepestDescent}[args.method](**kw)
    minimizer = minimize(optimizer, maxIter=args.maxiter, ftol=args.ftol, gtol=args.gtol, log_path=log)
    tic = time()
    m = minimizer.run(m0, geometry0, obs, misfit_func, direct_wave, bathy_mask, args.precond, bounds)
    toc = time()
    vp = 1.0 / np.sqrt(m.reshape(shape))
    vp.tofile(os.path.join(args.odir, 'marmousi_result_misfit_' + str(args.misfit)))
    err0 = np.linalg.norm(smooth_vp - true_vp) / np.linalg.norm(true_vp)
    err1 = np.linalg.norm(vp - true_vp) / np.linalg.norm(true_vp)
    print('\n Elapsed time: %.2fs   relative model error %.4f -> %.4f' % (toc - tic, err0, err1))
    return vp, err0, err1


if __name__ == '__main__':
    main()
