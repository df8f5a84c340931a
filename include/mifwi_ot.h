/*
 * Host-side helpers of the 2-D quadratic Wasserstein misfit (back-and-forth method).  Plain C ABI, no GPU.
 *
 * They replace the two grid kernels of the reference's optimal-transport solver, which it builds as the
 * separate executable misfit/QW2D/bin/bfm2d (misfit/QW2D/src/fot2d.c, called through files by
 * misfit/bfm.py:133-192): the Legendre transform of a potential and the push-forward of a density by the
 * gradient of a potential.  The outer iteration (Poisson solves by DCT, step-size rule, objective) is
 * host Python in devito-fwi_amd/misfit/bfm.py, as in misfit/bfm.py:16-131 of the reference.
 *
 * Arrays are row-major [n2][n1] doubles on the unit square, cell centres ((j + 1/2) / n1, (i + 1/2) / n2).
 */
#ifndef MIFWI_OT_H
#define MIFWI_OT_H

#define MIFWI_OT_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces: compute_2d_dual (misfit/QW2D/src/fot2d.c:65-172): dual(s, t) = max over the grid of
 * s x + t y - u(x, y), evaluated at the cell centres, by two passes of the 1-D transform through the lower
 * convex hull.  Returns 0, or -1 for a bad argument. */
MIFWI_OT_API int mifwi_ot_legendre2d(int n1, int n2, const double *u, double *dual);

/* Replaces: calc_pushforward_map + sampling_pushforward (fot2d.c:268-325,373-457): rho = (grad potential)_# mu.
 * The map is the centred difference of the bilinearly interpolated potential at the cell corners; every
 * cell's mass is spread over 2 max(n stretch, 1) samples per direction and deposited bilinearly; cells
 * stretched by more than n^(-1/3) are skipped; the result is rescaled to mean `total_mass`.
 * Returns 0, or -1 for a bad argument. */
MIFWI_OT_API int mifwi_ot_pushforward2d(int n1, int n2, const double *potential, const double *mu, double *rho,
                                        double total_mass);

#ifdef __cplusplus
}
#endif
#endif
