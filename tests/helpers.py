"""Synthetic problem builders and comparison helpers shared by the parity tests."""
import numpy as np

from oracle import oracle as orc


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b.ravel())
    return np.linalg.norm((a - b).ravel()) / (den if den > 0 else 1.0)


def smooth_random_vp(shape, vmin=1.5, vmax=3.5, seed=0, passes=4):
    """Seeded smooth velocity model in [vmin, vmax] (box-smoothed white noise)."""
    rng = np.random.default_rng(seed)
    v = rng.random(shape)
    for _ in range(passes):
        for ax in range(len(shape)):
            v = (np.roll(v, 1, ax) + v + np.roll(v, -1, ax)) / 3.0
    v = (v - v.min()) / (v.max() - v.min() + 1e-30)
    return (vmin + (vmax - vmin) * v).astype(np.float64)


class Case(object):
    """A small acquisition: oracle problem + geometry + wavelet, all seeded."""

    def __init__(self, shape, spacing, nbl, space_order, nt=None, tn=None, nshots=2, nrec=11, nsrc=1,
                 dtype=np.float32, seed=0, f0=0.015, vp=None, fs=False, kernel='OT2'):
        self.dtype = np.dtype(dtype)
        self.ndim = len(shape)
        vp = smooth_random_vp(shape, seed=seed) if vp is None else vp
        self.fs = bool(fs)
        self.prob = orc.Problem(vp.astype(self.dtype), spacing, nbl, space_order, dtype=self.dtype, fs=fs,
                                kernel=kernel)
        self.dt = float(self.prob.solver_dt)
        if nt is None:
            nt, tv = orc.time_axis(0.0, tn, self.dt)
        else:
            tv = self.dt * np.arange(nt)
        self.nt = nt
        self.f0 = f0
        self.wavelet = orc.ricker(f0, tv).astype(self.dtype).reshape(nt, 1).repeat(nsrc, axis=1)
        if nsrc > 1:  # make the sources differ
            self.wavelet = self.wavelet * (1.0 + 0.25 * np.arange(nsrc, dtype=self.dtype))[None, :]
        rng = np.random.default_rng(seed + 1)
        dom = np.array([(n - 1) * s for n, s in zip(shape, spacing)])
        # off-grid sources in the upper part, receivers spread over the domain at shallow depth
        self.src = (rng.random((nshots, nsrc, self.ndim)) * 0.6 + 0.2) * dom
        self.src[..., -1] = (0.13 + 0.1 * rng.random((nshots, nsrc))) * dom[-1]
        self.rec = rng.random((nrec, self.ndim)) * dom
        self.rec[:, -1] = 0.23 * dom[-1] + rng.random(nrec) * spacing[-1]
        self.src = self.src.astype(self.dtype)
        self.rec = self.rec.astype(self.dtype)
        self.nshots, self.nrec, self.nsrc = nshots, nrec, nsrc

    def plan(self, mifwi, max_shots=None, flags=0):
        p = self.prob
        plan = mifwi.Plan(p.shape_pad, p.spacing, p.origin_pad, p.space_order, self.nt, self.dt, self.nsrc,
                          self.nrec, max_shots=max_shots or self.nshots, dtype=self.dtype,
                          flags=flags | (16 if self.fs else 0) | (32 if self.prob.ot4 else 0))
        plan.set_model(p.vp, p.damp)
        plan.set_geometry(self.src, self.rec)
        return plan

    # oracle results, per shot
    def oracle_forward(self, shot, save=False):
        return self.prob.forward(self.src[shot], self.wavelet, self.rec, dt=self.dt, save=save)

    def oracle_adjoint(self, shot, rec):
        return self.prob.adjoint(self.src[shot], self.rec, rec, dt=self.dt)

    def oracle_gradient(self, residual, u):
        return self.prob.gradient(self.rec, residual, u, dt=self.dt)

    def oracle_born(self, shot, dm):
        return self.prob.born(self.src[shot], self.wavelet, self.rec, dm, dt=self.dt)
