// Host-side grid kernels of the back-and-forth optimal-transport solver (include/mifwi_ot.h).
// Restated from the algorithm of the reference's misfit/QW2D/src/fot2d.c in double precision.
#include <algorithm>
#include <cmath>
#include <vector>

#include "mifwi_ot.h"

namespace {

// 1-D discrete Legendre transform on the unit interval: dual[k] = max_i ( s_k x_i - u_i ), x_i = (i + 1/2) / n,
// s_k = (k + 1/2) / n, through the lower convex hull of (i, u_i) (fot2d.c:65-140).  `hull` is scratch of n ints.
void legendre1d(const double *u, double *dual, int n, int *hull)
{
    if (n == 1) {
        dual[0] = 0.5 * 0.5 - u[0];
        return;
    }
    // lower convex hull by a monotone chain: keep slopes non-decreasing
    int hc = 0;
    hull[hc++] = 0;
    hull[hc++] = 1;
    for (int i = 2; i < n; ++i) {
        while (hc >= 2) {
            const int a = hull[hc - 1], b = hull[hc - 2];
            const double old_slope = (u[a] - u[b]) / (a - b), slope = (u[i] - u[a]) / (i - a);
            if (slope >= old_slope) break;
            --hc;
        }
        hull[hc++] = i;
    }
    // walk the hull edges: for slope s the maximiser is the hull vertex where the edge slope first reaches s
    int counter = 1;
    for (int k = 0; k < n; ++k) {
        const double s = (k + 0.5) / n;
        auto edge_slope = [&](int c) { return n * (u[hull[c]] - u[hull[c - 1]]) / (hull[c] - hull[c - 1]); };
        while (s > edge_slope(counter) && counter < hc - 1) ++counter;
        const int idx = hull[counter - 1];
        const double v1 = s * (idx + 0.5) / n - u[idx];
        const double v2 = s * (n - 0.5) / n - u[n - 1];
        dual[k] = std::max(v1, v2);
    }
}

// bilinear interpolation of a cell-centred function with constant extension (fot2d.c:268-292)
double sample(const double *f, double x, double y, int n1, int n2)
{
    const int xi = (int)std::min(std::max(x * n1 - 0.5, 0.0), (double)(n1 - 1));
    const int yi = (int)std::min(std::max(y * n2 - 0.5, 0.0), (double)(n2 - 1));
    const double xf = x * n1 - xi - 0.5, yf = y * n2 - yi - 0.5;
    const int xo = std::max(std::min(xi + ((xf > 0) - (xf < 0)), n1 - 1), 0);
    const int yo = std::max(std::min(yi + ((yf > 0) - (yf < 0)), n2 - 1), 0);
    const double ax = std::fabs(xf), ay = std::fabs(yf);
    return (1 - ax) * (1 - ay) * f[(long)yi * n1 + xi] + ax * (1 - ay) * f[(long)yi * n1 + xo] +
           (1 - ax) * ay * f[(long)yo * n1 + xi] + ax * ay * f[(long)yo * n1 + xo];
}

}  // namespace

extern "C" int mifwi_ot_legendre2d(int n1, int n2, const double *u, double *dual)
{
    if (n1 < 1 || n2 < 1 || !u || !dual) return -1;
    const long pc = (long)n1 * n2;
    std::vector<double> a(pc), b(pc);
    std::vector<int> hull(std::max(n1, n2));
    // along x for every row, then along y of the negated result (max over y of t y - (-L_x u)(s, y))
    for (int i = 0; i < n2; ++i) legendre1d(u + (long)i * n1, a.data() + (long)i * n1, n1, hull.data());
    for (int i = 0; i < n2; ++i)
        for (int j = 0; j < n1; ++j) b[(long)j * n2 + i] = -a[(long)i * n1 + j];
    for (int j = 0; j < n1; ++j) legendre1d(b.data() + (long)j * n2, a.data() + (long)j * n2, n2, hull.data());
    for (int j = 0; j < n1; ++j)
        for (int i = 0; i < n2; ++i) dual[(long)i * n1 + j] = a[(long)j * n2 + i];
    return 0;
}

extern "C" int mifwi_ot_pushforward2d(int n1, int n2, const double *potential, const double *mu, double *rho,
                                      double total_mass)
{
    if (n1 < 1 || n2 < 1 || !potential || !mu || !rho) return -1;
    const long pc = (long)n1 * n2;
    const int w = n1 + 1;
    std::vector<double> xm((long)w * (n2 + 1)), ym((long)w * (n2 + 1));
    const double xs = 1.0 / n1, ys = 1.0 / n2;
    for (int i = 0; i <= n2; ++i)
        for (int j = 0; j <= n1; ++j) {
            const double x = j * xs, y = i * ys;
            xm[(long)i * w + j] = 0.5 * n1 * (sample(potential, x + xs, y, n1, n2) - sample(potential, x - xs, y, n1, n2));
            ym[(long)i * w + j] = 0.5 * n2 * (sample(potential, x, y + ys, n1, n2) - sample(potential, x, y - ys, n1, n2));
        }
    std::fill(rho, rho + pc, 0.0);
    const double xcut = std::pow(1.0 / n1, 1.0 / 3), ycut = std::pow(1.0 / n2, 1.0 / 3);
    for (int i = 0; i < n2; ++i)
        for (int j = 0; j < n1; ++j) {
            const double mass = mu[(long)i * n1 + j];
            if (!(mass > 0)) continue;
            const double x00 = xm[(long)i * w + j], x01 = xm[(long)i * w + j + 1];
            const double x10 = xm[(long)(i + 1) * w + j], x11 = xm[(long)(i + 1) * w + j + 1];
            const double y00 = ym[(long)i * w + j], y01 = ym[(long)i * w + j + 1];
            const double y10 = ym[(long)(i + 1) * w + j], y11 = ym[(long)(i + 1) * w + j + 1];
            const double xstretch = std::max(std::fabs(x01 - x00), std::fabs(x11 - x10));
            const double ystretch = std::max(std::fabs(y10 - y00), std::fabs(y11 - y01));
            if (!(xstretch < xcut && ystretch < ycut)) continue;
            const int nxs = 2 * (int)std::max(n1 * xstretch, 1.0), nys = 2 * (int)std::max(n2 * ystretch, 1.0);
            const double part = mass / ((double)nxs * nys);
            for (int l = 0; l < nys; ++l)
                for (int k = 0; k < nxs; ++k) {
                    const double a = (k + 0.5) / nxs, b = (l + 0.5) / nys;
                    const double xp = (1 - b) * (1 - a) * x00 + (1 - b) * a * x01 + b * (1 - a) * x10 + a * b * x11;
                    const double yp = (1 - b) * (1 - a) * y00 + (1 - b) * a * y01 + b * (1 - a) * y10 + a * b * y11;
                    const double X = xp * n1 - 0.5, Y = yp * n2 - 0.5;
                    int xi = (int)X, yi = (int)Y;            // truncation towards zero, as the reference
                    const double xf = X - xi, yf = Y - yi;
                    int xo = xi + 1, yo = yi + 1;
                    xi = std::min(std::max(xi, 0), n1 - 1), xo = std::min(std::max(xo, 0), n1 - 1);
                    yi = std::min(std::max(yi, 0), n2 - 1), yo = std::min(std::max(yo, 0), n2 - 1);
                    rho[(long)yi * n1 + xi] += (1 - xf) * (1 - yf) * part;
                    rho[(long)yo * n1 + xi] += (1 - xf) * yf * part;
                    rho[(long)yi * n1 + xo] += xf * (1 - yf) * part;
                    rho[(long)yo * n1 + xo] += xf * yf * part;
                }
        }
    double mean = 0;
    for (long i = 0; i < pc; ++i) mean += rho[i] / pc;
    if (mean > 0)
        for (long i = 0; i < pc; ++i) rho[i] *= total_mass / mean;
    return 0;
}
