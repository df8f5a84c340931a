"""Transmission FWI of a circular anomaly on the GPU: the workflow of the reference's circle_fwi.py:36-198
(201 x 201 grid, 11 sources on one side, 201 receivers on the other, resampled traces, L-BFGS with
box constraints), without the plotting.

    python examples/circle_fwi.py --maxiter 5 --odir /tmp/circle
"""
import argparse
import os
import sys
from time import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import ROOT                                  # noqa: E402,F401  (puts the package on sys.path)
from seismic import demo_model, AcquisitionGeometry      # noqa: E402
from fwi import Filter, fm_multi, fwi_obj_multi          # noqa: E402
from misfit import least_square, qWasserstein            # noqa: E402
from optimize import LBFGS                               # noqa: E402
from minimize import minimize                            # noqa: E402

parser = argparse.ArgumentParser(description='Full waveform inversion of a circular anomaly')
parser.add_argument('--misfit', type=int, default=0, choices=[0, 1, 2],
                    help='0=least squares, 1=trace-wise W2, 2=2-D W2 of the whole gather (back-and-forth method)')
parser.add_argument('--precond', type=int, default=1)
parser.add_argument('--odir', type=str, default='./result/circle')
parser.add_argument('--bathy', type=int, default=1)
parser.add_argument('--check-gradient', type=int, default=0)
parser.add_argument('--filter', type=int, default=0)
parser.add_argument('--resample', type=float, default=0.)
parser.add_argument('--ftol', type=float, default=1e-2)
parser.add_argument('--gtol', type=float, default=1e-4)
parser.add_argument('--nsrc', type=int, default=11)
parser.add_argument('--maxiter', type=int, default=50)
parser.add_argument('--steplen', type=float, default=0.05)
parser.add_argument('--maxls', type=int, default=5)


def main(argv=None):
    args = parser.parse_args(argv)
    os.makedirs(args.odir, exist_ok=True)
    shape, spacing, origin, space_order, nbl, dt, radius = (201, 201), (10., 10.), (0, 0), 6, 40, 1., 60
    kw = dict(r=radius, origin=origin, shape=shape, spacing=spacing, space_order=space_order, nbl=nbl, dt=dt)
    true_model = demo_model('circle-isotropic', vp_circle=3.6, vp_background=3, **kw)
    init_model = demo_model('circle-isotropic', vp_circle=3, vp_background=3, **kw)
    mask = np.ones(shape, dtype=np.float32) if args.bathy else None
    src = np.empty((args.nsrc, 2))
    src[:, 1] = np.linspace(0, true_model.domain_size[0], num=args.nsrc)
    src[:, 0] = 20.
    rec = np.empty((shape[0], 2))
    rec[:, 1] = np.linspace(spacing[0], true_model.domain_size[0] - spacing[0], num=shape[0])
    rec[:, 0] = 1980.
    filt = Filter(filter_type='highpass', freqmin=2, corners=6, df=1000 / dt) if args.filter else None
    geometry1 = AcquisitionGeometry(true_model, rec, src, 0., 1000., f0=0.010, src_type='Ricker', filter=filt)
    geometry0 = AcquisitionGeometry(init_model, rec, src, 0., 1000., f0=0.010, src_type='Ricker', filter=filt)
    resample_dt = args.resample if args.resample else dt
    geometry1.resample(resample_dt)
    geometry0.resample(resample_dt)
    obs = fm_multi(geometry1, save=False)
    misfit_func = (least_square, qWasserstein(gamma=1.01, method='1d'),
                   qWasserstein(gamma=1.01, method='2d', num_steps=10, step_scale=1.))[args.misfit]
    if args.check_gradient:
        f, g, _ = fwi_obj_multi(geometry0, obs, misfit_func, None, mask, args.precond, True)
        g.tofile(os.path.join(args.odir, 'circle_1st_grad_' + str(args.misfit)))
        print('objective %.6e   |gradient| %.6e' % (f, np.linalg.norm(g)))
    bounds = [1.0 / 4.0 ** 2, 1.0 / 2.5 ** 2]
    v0 = init_model.vp.data[nbl:-nbl, nbl:-nbl]
    m0 = 1.0 / (v0.reshape(-1).astype(np.float64)) ** 2
    log = os.path.join(args.odir, 'log' + str(args.misfit))
    optimizer = LBFGS(memory=10, ls_method='Bracket', step_len_init=args.steplen, max_ls=args.maxls, log_path=log)
    minimizer = minimize(optimizer, maxIter=args.maxiter, ftol=args.ftol, gtol=args.gtol, log_path=log)
    tic = time()
    m = minimizer.run(m0, geometry0, obs, misfit_func, None, mask, args.precond, bounds)
    toc = time()
    vp = 1.0 / np.sqrt(m.reshape(shape))
    vp.tofile(os.path.join(args.odir, 'circle_result_misfit_' + str(args.misfit)))
    true_vp = true_model.vp.data[nbl:-nbl, nbl:-nbl]
    err0 = np.linalg.norm(v0 - true_vp) / np.linalg.norm(true_vp)
    err1 = np.linalg.norm(vp - true_vp) / np.linalg.norm(true_vp)
    print('\n Elapsed time: %.2fs   relative model error %.4f -> %.4f' % (toc - tic, err0, err1))
    return vp, err0, err1


if __name__ == '__main__':
    main()
