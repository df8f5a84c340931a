/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the isotropic acoustic forward / adjoint /
 * gradient propagators (see acoustic_oracle_impl.h for the restated equations and the
 * reference file:line each part follows).  Built into oracle/liboracle.so by oracle/Makefile
 * and loaded by oracle/oracle.py.  It is the parity checker used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg; the product never calls it.
 *
 * Parity status: PINNED by the reference's own known-answer tests
 * (seismic/inversion/fwi.py:95-97,121 and seismic/acoustic/acoustic_example.py:75-79),
 * reproduced in tests/test_oracle_kat.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_MAXR 16

/*
 * Centred finite-difference weights for the 2nd derivative on the 2R+1 points -R..R
 * (Fornberg 1988 recurrence; what sympy.finite_diff_weights(2, range(-R,R+1), 0)[-1][-1]
 * returns, seismic/model.py:352 and Devito's `.laplace`).  w[k] for k=0..R (symmetric).
 */
void oracle_fd2_weights(int R, double *w)
{
    const int n = 2 * R + 1, M = 2;
    double *x = (double *)malloc(sizeof(double) * (size_t)n);
    double *c = (double *)calloc((size_t)n * (M + 1), sizeof(double)); /* c[j*(M+1)+m] */
    /* sympy orders the points as given: -R..R */
    for (int j = 0; j < n; ++j) x[j] = (double)(j - R);
    const double x0 = 0.0;
    double c1 = 1.0, c4 = x[0] - x0;
    c[0] = 1.0;
    for (int i = 1; i < n; ++i) {
        int mn = i < M ? i : M;
        double c2 = 1.0, c5 = c4;
        c4 = x[i] - x0;
        for (int j = 0; j < i; ++j) {
            double c3 = x[i] - x[j];
            c2 *= c3;
            if (j == i - 1) {
                for (int m = mn; m >= 1; --m)
                    c[i * (M + 1) + m] = c1 * (m * c[(i - 1) * (M + 1) + m - 1] - c5 * c[(i - 1) * (M + 1) + m]) / c2;
                c[i * (M + 1)] = -c1 * c5 * c[(i - 1) * (M + 1)] / c2;
            }
            for (int m = mn; m >= 1; --m)
                c[j * (M + 1) + m] = (c4 * c[j * (M + 1) + m] - m * c[j * (M + 1) + m - 1]) / c3;
            c[j * (M + 1)] = c4 * c[j * (M + 1)] / c3;
        }
        c1 = c2;
    }
    for (int k = 0; k <= R; ++k) w[k] = c[(R + k) * (M + 1) + 2];
    free(x);
    free(c);
}

/*
 * CFL coefficient of seismic/model.py:338-353 (acoustic branch): sqrt(4 / (ndim * sum|c|)) with
 * c the 2nd-derivative weights on -so..so (radius = space_order, not space_order/2).
 */
double oracle_cfl_coeff(int space_order, int ndim)
{
    double *w = (double *)malloc(sizeof(double) * (size_t)(space_order + 1));
    oracle_fd2_weights(space_order, w);
    double s = fabs(w[0]);
    for (int k = 1; k <= space_order; ++k) s += 2.0 * fabs(w[k]);
    free(w);
    return sqrt(4.0 / ((double)ndim * s));
}

#define REAL float
#define SFX f32
#include "acoustic_oracle_impl.h"
#undef REAL
#undef SFX

#define REAL double
#define SFX f64
#include "acoustic_oracle_impl.h"
#undef REAL
#undef SFX
