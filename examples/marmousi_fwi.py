"""Marmousi / Marmousi-II FWI on the GPU: the workflow of the reference's marmousi_fwi.py:36-190 and
marmousi2_fwi.py:36-207 (true/smooth/water models, observed and direct-wave data, bathymetry mask,
bounds, optional first-gradient dump, minimize.run), without the plotting.  Start one process per GPU with
torch.distributed to shard the shots.

    python examples/marmousi_fwi.py --maxiter 5 --nsrc 8 --odir /tmp/smarmn
    python examples/marmousi_fwi.py --model SMARM2 --maxiter 5 --nsrc 8 --odir /tmp/smarm2
"""
import argparse
import os
import sys
from time import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _common as common                                 # noqa: E402
from _common import ROOT                                  # noqa: F401  (also puts the package on sys.path)
from fwi import fm_multi, fwi_obj_multi                   # noqa: E402
from misfit import least_square, qWasserstein             # noqa: E402
from optimize import LBFGS, NLCG, SteepestDescent         # noqa: E402
from minimize import minimize                             # noqa: E402

parser = argparse.ArgumentParser(description='Full waveform inversion')
parser.add_argument('--model', default='SMARMN', choices=sorted(common.CONFIGS))
parser.add_argument('--data', default=None, help='directory holding vp.true and vp.smooth_20 (default: the fixtures)')
parser.add_argument('--check-gradient', type=int, default=0, help='1: also write the first gradient to odir')
parser.add_argument('--misfit', type=int, default=0, choices=[0, 1, 2],
                    help='0=least squares, 1=trace-wise W2, 2=2-D W2 of the whole gather (back-and-forth method)')
parser.add_argument('--method', default='LBFGS', choices=['LBFGS', 'NLCG', 'SteepestDescent'])
parser.add_argument('--precond', type=int, default=1)
parser.add_argument('--odir', type=str, default=None, help='default ./result/<model>')
parser.add_argument('--bathy', type=int, default=1)
parser.add_argument('--filter', type=int, default=0)
parser.add_argument('--resample', type=float, default=0.)
parser.add_argument('--ftol', type=float, default=1e-5)
parser.add_argument('--gtol', type=float, default=1e-10)
parser.add_argument('--nsrc', type=int, default=29)
parser.add_argument('--maxiter', type=int, default=200)
parser.add_argument('--steplen', type=float, default=0.1)
parser.add_argument('--maxls', type=int, default=5)
parser.add_argument('--tn', type=float, default=None, help='record length in ms (default: the reference driver\'s)')
parser.add_argument('--secant-pairs', type=int, default=0,
                    help='1: store g - g_previous in the L-BFGS memory instead of the reference\'s choice')


def main(argv=None):
    args = parser.parse_args(argv)
    args.odir = args.odir or os.path.join('.', 'result', args.model)
    os.makedirs(args.odir, exist_ok=True)
    cfg, true_vp, smooth_vp, model_of = common.load(args.model, args.data)
    shape, dt = cfg['shape'], cfg['dt']
    bathy_mask = None
    if args.bathy:
        bathy_mask = np.ones(shape, dtype=np.float32)
        bathy_mask[:, :cfg['water']] = 0
    true_model, init_model, water_model = model_of(true_vp), model_of(smooth_vp), model_of(np.ones(shape) * 1.5)
    src, rec = common.acquisition(true_model, args.nsrc, shape)
    geometry1, geometry0, geometry2 = common.geometries((true_model, init_model, water_model), src, rec,
                                                        args.tn or cfg['tn']['fwi'], dt, args.filter)
    geometry0.resample(args.resample if args.resample else dt)

    obs = fm_multi(geometry1, save=False)
    direct_wave = fm_multi(geometry2, save=False)
    misfit_func = (least_square, qWasserstein(gamma=1.01, method='1d'),
                   qWasserstein(gamma=1.01, method='2d', num_steps=10, step_scale=1.))[args.misfit]

    if args.check_gradient:
        f, g, _ = fwi_obj_multi(geometry0, obs, misfit_func, direct_wave, bathy_mask, args.precond, True)
        g.tofile(os.path.join(args.odir, '1st_grad_' + str(args.misfit)))
        print('objective %.6e   |gradient| %.6e' % (f, np.linalg.norm(g)))
    bounds = [1.0 / cfg['vmax'] ** 2, 1.0 / 1.5 ** 2]
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
    vp.tofile(os.path.join(args.odir, args.model.lower() + '_result_misfit_' + str(args.misfit)))
    err0 = np.linalg.norm(smooth_vp - true_vp) / np.linalg.norm(true_vp)
    err1 = np.linalg.norm(vp - true_vp) / np.linalg.norm(true_vp)
    print('\n Elapsed time: %.2fs   relative model error %.4f -> %.4f' % (toc - tic, err0, err1))
    return vp, err0, err1


if __name__ == '__main__':
    main()
