// TTI ('centered' kernel) propagators: two coupled fields advanced with rotated second derivatives
// built from centred first derivatives (seismic/tti/operators.py:7-28,52-117,120-313 of the
// reference).  One thread per grid point; three launches per time step (rotated first derivative,
// update of both fields, sparse terms of both fields).  Notation, with et = 1 + 2 epsilon, dl = sqrt(1 + 2 delta):
//     Gz(f)  = -( st cp D_x f + st sp D_y f + ct D_z f )            st/ct = sin/cos(theta), sp/cp = sin/cos(phi)
//     Gzz(f) = D_x^T( st cp Gz ) + D_y^T( st sp Gz ) + D_z^T( ct Gz ),   D^T = -D
//     X = laplace(A) - Gzz(A),   Y = Gzz(B)
//     forward:  (A, B) = (u, v):                       H0 = et X + dl Y,  Hz = dl X + Y
//     adjoint:  (A, B) = (et p + dl r, dl p + r):      H0 = X,            Hz = Y
//     out = ( dt^2 H + 2 m cur - m old + eta dt cur ) / ( m + eta dt )   for each of the two fields
// Gz is evaluated p = space_order/4 points into the halo, where the parameter fields hold
// theta = phi = 0 (ct = cp = 1, st = sp = 0) and et = dl = 1, as Devito's zero-filled halos give.
#include "mifwi_internal.hpp"
#include <type_traits>

namespace mifwi {

template <typename T>
__global__ __launch_bounds__(256) void tti_fill_kernel(T *dst, long n, T value)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += 256L * gridDim.x) dst[i] = value;
}

// mode 1: 1 + 2x, 2: sqrt(1 + 2x), 3: cos x, 4: sin x     (dense grid -> haloed field)
template <typename T>
__global__ __launch_bounds__(256) void tti_prep_kernel(StepConsts<T> k, const T *dense, T *haloed, int mode)
{
    const long ng = (long)k.n0 * k.n1 * k.n2;
    for (long g = blockIdx.x * 256L + threadIdx.x; g < ng; g += 256L * gridDim.x) {
        const int i2 = (int)(g % k.n2);
        const long r = g / k.n2;
        const int i1 = (int)(r % k.n1), i0 = (int)(r / k.n1);
        const T x = dense[g];
        T y = x;
        if (mode == 1) y = T(1) + T(2) * x;
        else if (mode == 2) y = sqrt(T(1) + T(2) * x);
        else if (mode == 3) y = cos(x);
        else if (mode == 4) y = sin(x);
        haloed[k.off0 + (long)i0 * k.s0 + (long)i1 * k.s1 + i2] = y;
    }
}

template <typename T>
int launch_tti_param(const Plan &p, const StepConsts<T> &k, const T *dense, T *haloed, int mode, T halo_value)
{
    hipLaunchKernelGGL((tti_fill_kernel<T>), dim3(1024), dim3(256), 0, p.stream, haloed, (long)p.L.field, halo_value);
    if (dense)
        hipLaunchKernelGGL((tti_prep_kernel<T>), dim3(1024), dim3(256), 0, p.stream, k, dense, haloed, mode);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template int launch_tti_param<float>(const Plan &, const StepConsts<float> &, const float *, float *, int, float);
template int launch_tti_param<double>(const Plan &, const StepConsts<double> &, const double *, double *, int, double);

// adjoint only: wA = et p + dl r, wB = dl p + r on the grid points
template <typename T>
__global__ __launch_bounds__(256) void tti_combine_kernel(StepConsts<T> k, TtiParams<T> q, const T *pc, const T *rc,
                                                          long shot_stride, T *wA, T *wB)
{
    const int i2 = blockIdx.x * 64 + threadIdx.x;
    const int i1 = blockIdx.y * 4 + threadIdx.y;
    const int shot = blockIdx.z / k.n0;
    const int i0 = blockIdx.z - shot * k.n0;
    if (i2 >= k.n2 || i1 >= k.n1) return;
    const long i = k.off0 + (long)i0 * k.s0 + (long)i1 * k.s1 + i2;
    const T pv = pc[(long)shot * shot_stride + i], rv = rc[(long)shot * shot_stride + i];
    wA[(long)shot * k.field + i] = q.et[i] * pv + q.dl[i] * rv;
    wB[(long)shot * k.field + i] = q.dl[i] * pv + rv;
}

template <typename T>
__device__ inline T tti_d1(const TtiParams<T> &q, const T *f, long stride, T ih)
{
    T s = 0;
    for (int j = 1; j <= q.p; ++j) s += q.a[j] * (f[j * stride] - f[-j * stride]);
    return s * ih;
}

// gzA = Gz(A), gzB = Gz(B) on the grid extended by q.p points per active dimension
template <typename T, int ND>
__global__ __launch_bounds__(256) void tti_gz_kernel(StepConsts<T> k, TtiParams<T> q, const T *A, const T *B,
                                                     long in_stride, T *gzA, T *gzB)
{
    const int e = q.p;
    const int e0 = ND >= 3 ? e : 0;
    const int n0e = k.n0 + 2 * e0;
    const int i2 = (int)(blockIdx.x * 64 + threadIdx.x) - e;
    const int i1 = (int)(blockIdx.y * 4 + threadIdx.y) - e;
    const int shot = blockIdx.z / n0e;
    const int i0 = (int)(blockIdx.z - shot * n0e) - e0;
    if (i2 >= k.n2 + e || i1 >= k.n1 + e) return;
    const long i = k.off0 + (long)i0 * k.s0 + (long)i1 * k.s1 + i2;
    const T *a = A + (long)shot * in_stride + i, *b = B + (long)shot * in_stride + i;
    const T ct = q.ct[i], st = q.st[i];
    T ga, gb;
    if (ND >= 3) {
        const T cp = q.cp[i], sp = q.sp[i];
        ga = -(st * cp * tti_d1(q, a, k.s0, q.ih[0]) + st * sp * tti_d1(q, a, k.s1, q.ih[1]) + ct * tti_d1(q, a, 1L, q.ih[2]));
        gb = -(st * cp * tti_d1(q, b, k.s0, q.ih[0]) + st * sp * tti_d1(q, b, k.s1, q.ih[1]) + ct * tti_d1(q, b, 1L, q.ih[2]));
    } else {
        ga = -(st * tti_d1(q, a, k.s1, q.ih[1]) + ct * tti_d1(q, a, 1L, q.ih[2]));
        gb = -(st * tti_d1(q, b, k.s1, q.ih[1]) + ct * tti_d1(q, b, 1L, q.ih[2]));
    }
    gzA[(long)shot * k.field + i] = ga;
    gzB[(long)shot * k.field + i] = gb;
}

template <typename T, int ND>
__device__ inline T tti_gzz(const StepConsts<T> &k, const TtiParams<T> &q, const T *gz, long i)
{
    T s = 0;
    for (int j = 1; j <= q.p; ++j) {
        T t = q.a[j] * (q.ct[i + j] * gz[i + j] - q.ct[i - j] * gz[i - j]) * q.ih[2];
        if (ND >= 3) {
            const long jx = j * k.s0, jy = j * k.s1;
            t += q.a[j] * (q.st[i + jx] * q.cp[i + jx] * gz[i + jx] - q.st[i - jx] * q.cp[i - jx] * gz[i - jx]) * q.ih[0];
            t += q.a[j] * (q.st[i + jy] * q.sp[i + jy] * gz[i + jy] - q.st[i - jy] * q.sp[i - jy] * gz[i - jy]) * q.ih[1];
        } else {
            const long jx = j * k.s1;
            t += q.a[j] * (q.st[i + jx] * gz[i + jx] - q.st[i - jx] * gz[i - jx]) * q.ih[1];
        }
        s += t;
    }
    return -s;
}

// update of both fields
template <typename T, int R, int ND>
__global__ __launch_bounds__(256) void tti_step_kernel(StepConsts<T> k, TtiParams<T> q, int fwd, const T *A, const T *B,
                                                       long ab_stride, const T *gzA, const T *gzB, TtiLevels<T> lv,
                                                       const T *m, const T *eta)
{
    const int i2 = blockIdx.x * 64 + threadIdx.x;
    const int i1 = blockIdx.y * 4 + threadIdx.y;
    const int shot = blockIdx.z / k.n0;
    const int i0 = blockIdx.z - shot * k.n0;
    if (i2 >= k.n2 || i1 >= k.n1) return;
    const long i = k.off0 + (long)i0 * k.s0 + (long)i1 * k.s1 + i2;
    const T *a = A + (long)shot * ab_stride + i;
    T lap = k.c0 * a[0];
#pragma unroll
    for (int j = 1; j <= R; ++j) lap += k.c[2][j] * (a[-j] + a[j]);
#pragma unroll
    for (int j = 1; j <= R; ++j) lap += k.c[1][j] * (a[-j * k.s1] + a[j * k.s1]);
    if (ND >= 3) {
#pragma unroll
        for (int j = 1; j <= R; ++j) lap += k.c[0][j] * (a[-j * k.s0] + a[j * k.s0]);
    }
    const T X = lap - tti_gzz<T, ND>(k, q, gzA + (long)shot * k.field, i);
    const T Y = tti_gzz<T, ND>(k, q, gzB + (long)shot * k.field, i);
    const T H0 = fwd ? q.et[i] * X + q.dl[i] * Y : X;
    const T Hz = fwd ? q.dl[i] * X + Y : Y;
    const long s = (long)shot * lv.shot_stride + i;
    const T uc = lv.ucur[s], vc = lv.vcur[s];
    const T mz = m[i], ed = eta[i] * k.dt;
    lv.uout[s] = (k.dt2 * H0 + T(2) * mz * uc - mz * lv.uold[s] + ed * uc) / (mz + ed);
    lv.vout[s] = (k.dt2 * Hz + T(2) * mz * vc - mz * lv.vold[s] + ed * vc) / (mz + ed);
}

// ------------------------------------------------------------------------------------------------
// Tiled 2-D TTI step (float32): one launch advances both fields of every shot.  A 32 x 32 tile of the
// two stencilled fields is staged in LDS with a halo of R, the angle fields with a halo of p = R/2;
// Gz of both fields is formed on the tile extended by p and kept in LDS, then every thread updates four
// points (rows ty, ty+8, ty+16, ty+24 of column tz) from LDS.  Same arithmetic, in the same order, as
// tti_gz_kernel + tti_step_kernel.
// ------------------------------------------------------------------------------------------------
struct Tti2DArgs {
    const float *ucur, *uold, *vcur, *vold;
    float *uout, *vout;
    long shot_stride;
    const float *m, *eta, *et, *dl, *ct, *st;
    float c0, cx[kMaxR + 1], cz[kMaxR + 1], a1[kMaxR / 2 + 1];
    float ihx, ihz, dt, dt2;
    int n1, n2;
    long s1, off0;
    int ntx, ntz, nshots;
};

template <int R, bool FWD>
__global__ __launch_bounds__(256) void tti2d_kernel(Tti2DArgs a)
{
    constexpr int TX = 32, TZ = 32, P = R / 2;
    constexpr int WA = TZ + 2 * R + 1, HA = TX + 2 * R;      // +1: odd pitch spreads the rows over the banks
    constexpr int WE = TZ + 2 * P + 1, HE = TX + 2 * P;
    __shared__ float sA[HA * WA], sB[HA * WA], sgA[HE * WE], sgB[HE * WE], sct[HE * WE], sst[HE * WE];
    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    const int shot = bid % a.nshots;
    const int tile = bid / a.nshots;
    const int tile_x = tile / a.ntz, tile_z = tile - tile_x * a.ntz;
    const int x0 = tile_x * TX, z0 = tile_z * TZ;
    const float *ucur = a.ucur + (long)shot * a.shot_stride;
    const float *vcur = a.vcur + (long)shot * a.shot_stride;

    // ---- stage the two fields (halo R) and the angle fields (halo P) ------------------------------
    for (int e = tid; e < HA * (WA - 1); e += 256) {
        const int r = e / (WA - 1), c = e - r * (WA - 1);
        const int gx = x0 - R + r, gz = z0 - R + c;
        float fa = 0.f, fb = 0.f;
        if (gx < a.n1 + R && gz < a.n2 + R) {
            const long i = a.off0 + (long)gx * a.s1 + gz;
            const float uu = ucur[i], vv = vcur[i];
            if (FWD) {
                fa = uu, fb = vv;
            } else {
                const float et = a.et[i], dl = a.dl[i];
                fa = et * uu + dl * vv;
                fb = dl * uu + vv;
            }
        }
        sA[r * WA + c] = fa;
        sB[r * WA + c] = fb;
    }
    for (int e = tid; e < HE * (WE - 1); e += 256) {
        const int r = e / (WE - 1), c = e - r * (WE - 1);
        const int gx = x0 - P + r, gz = z0 - P + c;
        float ct = 1.f, st = 0.f;
        if (gx < a.n1 + P && gz < a.n2 + P) {
            const long i = a.off0 + (long)gx * a.s1 + gz;
            ct = a.ct[i], st = a.st[i];
        }
        sct[r * WE + c] = ct;
        sst[r * WE + c] = st;
    }
    __syncthreads();

    // ---- Gz of both fields on the tile extended by P -----------------------------------------------
    for (int e = tid; e < HE * (WE - 1); e += 256) {
        const int r = e / (WE - 1), c = e - r * (WE - 1);
        const int ia = (r + R - P) * WA + (c + R - P);
        float dxa = 0.f, dza = 0.f, dxb = 0.f, dzb = 0.f;
#pragma unroll
        for (int j = 1; j <= P; ++j) {
            dxa += a.a1[j] * (sA[ia + j * WA] - sA[ia - j * WA]);
            dza += a.a1[j] * (sA[ia + j] - sA[ia - j]);
            dxb += a.a1[j] * (sB[ia + j * WA] - sB[ia - j * WA]);
            dzb += a.a1[j] * (sB[ia + j] - sB[ia - j]);
        }
        const float ct = sct[r * WE + c], st = sst[r * WE + c];
        sgA[r * WE + c] = -(st * (dxa * a.ihx) + ct * (dza * a.ihz));
        sgB[r * WE + c] = -(st * (dxb * a.ihx) + ct * (dzb * a.ihz));
    }
    __syncthreads();

    // ---- update four points per thread ----------------------------------------------------------------
    const int tz = tid & 31, ty = tid >> 5;
    const int z = z0 + tz;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = ty + 8 * k;
        const int x = x0 + row;
        if (x >= a.n1 || z >= a.n2) continue;
        const int ia = (row + R) * WA + tz + R;
        const int ie = (row + P) * WE + tz + P;
        float lap = a.c0 * sA[ia];
#pragma unroll
        for (int j = 1; j <= R; ++j) lap += a.cz[j] * (sA[ia - j] + sA[ia + j]);
#pragma unroll
        for (int j = 1; j <= R; ++j) lap += a.cx[j] * (sA[ia - j * WA] + sA[ia + j * WA]);
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int j = 1; j <= P; ++j) {
            float ta = a.a1[j] * (sct[ie + j] * sgA[ie + j] - sct[ie - j] * sgA[ie - j]) * a.ihz;
            ta += a.a1[j] * (sst[ie + j * WE] * sgA[ie + j * WE] - sst[ie - j * WE] * sgA[ie - j * WE]) * a.ihx;
            sa += ta;
            float tb = a.a1[j] * (sct[ie + j] * sgB[ie + j] - sct[ie - j] * sgB[ie - j]) * a.ihz;
            tb += a.a1[j] * (sst[ie + j * WE] * sgB[ie + j * WE] - sst[ie - j * WE] * sgB[ie - j * WE]) * a.ihx;
            sb += tb;
        }
        const float X = lap - (-sa), Y = -sb;
        const long i = a.off0 + (long)x * a.s1 + z;
        const long si = (long)shot * a.shot_stride + i;
        float H0 = X, Hz = Y;
        if (FWD) {
            const float et = a.et[i], dl = a.dl[i];
            H0 = et * X + dl * Y;
            Hz = dl * X + Y;
        }
        const float uc = a.ucur[si], vc = a.vcur[si];
        const float mz = a.m[i], ed = a.eta[i] * a.dt;
        a.uout[si] = (a.dt2 * H0 + 2.0f * mz * uc - mz * a.uold[si] + ed * uc) / (mz + ed);
        a.vout[si] = (a.dt2 * Hz + 2.0f * mz * vc - mz * a.vold[si] + ed * vc) / (mz + ed);
    }
}

template <int R>
static int launch_tti2d_r(const Plan &p, const Tti2DArgs &a, int fwd)
{
    const int nblocks = a.ntx * a.ntz * a.nshots;
    if (fwd)
        hipLaunchKernelGGL((tti2d_kernel<R, true>), dim3(nblocks), dim3(256), 0, p.stream, a);
    else
        hipLaunchKernelGGL((tti2d_kernel<R, false>), dim3(nblocks), dim3(256), 0, p.stream, a);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

static int launch_tti2d(const Plan &p, const StepConsts<float> &k, const TtiParams<float> &q, int fwd,
                        const TtiLevels<float> &lv, int nshots)
{
    Tti2DArgs a{};
    a.ucur = lv.ucur, a.uold = lv.uold, a.vcur = lv.vcur, a.vold = lv.vold, a.uout = lv.uout, a.vout = lv.vout;
    a.shot_stride = lv.shot_stride;
    a.m = (const float *)p.m, a.eta = (const float *)p.eta;
    a.et = q.et, a.dl = q.dl, a.ct = q.ct, a.st = q.st;
    a.c0 = k.c0;
    for (int j = 0; j <= p.L.R; ++j) a.cx[j] = k.c[1][j], a.cz[j] = k.c[2][j];
    for (int j = 1; j <= q.p; ++j) a.a1[j] = q.a[j];
    a.ihx = q.ih[1], a.ihz = q.ih[2];
    a.dt = k.dt, a.dt2 = k.dt2;
    a.n1 = k.n1, a.n2 = k.n2, a.s1 = k.s1, a.off0 = k.off0;
    a.ntx = (k.n1 + 31) / 32, a.ntz = (k.n2 + 31) / 32, a.nshots = nshots;
    switch (p.L.R) {
        case 2: return launch_tti2d_r<2>(p, a, fwd);
        case 4: return launch_tti2d_r<4>(p, a, fwd);
        case 6: return launch_tti2d_r<6>(p, a, fwd);
        case 8: return launch_tti2d_r<8>(p, a, fwd);
    }
    set_error("TTI needs space_order divisible by 4");
    return MIFWI_EINVAL;
}

template <typename T, int R>
static int launch_tti_step_r(const Plan &p, const StepConsts<T> &k, const TtiParams<T> &q, int fwd, const T *A, const T *B,
                             long ab_stride, const T *gzA, const T *gzB, const TtiLevels<T> &lv, int nshots)
{
    dim3 block(64, 4, 1);
    dim3 grid((k.n2 + 63) / 64, (k.n1 + 3) / 4, (unsigned)(nshots * k.n0));
    if (p.L.ndim == 3)
        hipLaunchKernelGGL((tti_step_kernel<T, R, 3>), grid, block, 0, p.stream, k, q, fwd, A, B, ab_stride, gzA, gzB, lv,
                           (const T *)p.m, (const T *)p.eta);
    else
        hipLaunchKernelGGL((tti_step_kernel<T, R, 2>), grid, block, 0, p.stream, k, q, fwd, A, B, ab_stride, gzA, gzB, lv,
                           (const T *)p.m, (const T *)p.eta);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

// One TTI time step without the sparse terms.  fwd: A, B = current u, v.  adjoint: A, B are formed
// into wA, wB first.  Launch count is added to `launches`.
template <typename T>
int launch_tti_step(const Plan &p, const StepConsts<T> &k, const TtiParams<T> &q, int fwd, const TtiLevels<T> &lv,
                    T *wA, T *wB, T *gzA, T *gzB, int nshots, int &launches)
{
    if constexpr (std::is_same<T, float>::value) {
        if (p.kind == 1) {   // tiled 2-D path: one launch for both fields
            launches += 1;
            return launch_tti2d(p, k, q, fwd, lv, nshots);
        }
    }
    const int e = q.p, e0 = p.L.ndim == 3 ? e : 0;
    if ((long)nshots * (k.n0 + 2 * e0) > 65535) {
        set_error("TTI kernels: nshots * shape[0] exceeds the grid z limit");
        return MIFWI_EINVAL;
    }
    dim3 block(64, 4, 1);
    const T *A = lv.ucur, *B = lv.vcur;
    long ab_stride = lv.shot_stride;
    if (!fwd) {
        dim3 grid((k.n2 + 63) / 64, (k.n1 + 3) / 4, (unsigned)(nshots * k.n0));
        hipLaunchKernelGGL((tti_combine_kernel<T>), grid, block, 0, p.stream, k, q, lv.ucur, lv.vcur, lv.shot_stride, wA, wB);
        A = wA, B = wB, ab_stride = k.field;
        launches += 1;
    }
    dim3 gridz((k.n2 + 2 * e + 63) / 64, (k.n1 + 2 * e + 3) / 4, (unsigned)(nshots * (k.n0 + 2 * e0)));
    if (p.L.ndim == 3)
        hipLaunchKernelGGL((tti_gz_kernel<T, 3>), gridz, block, 0, p.stream, k, q, A, B, ab_stride, gzA, gzB);
    else
        hipLaunchKernelGGL((tti_gz_kernel<T, 2>), gridz, block, 0, p.stream, k, q, A, B, ab_stride, gzA, gzB);
    MIFWI_HIP(hipGetLastError());
    launches += 2;
    switch (p.L.R) {
        case 2: return launch_tti_step_r<T, 2>(p, k, q, fwd, A, B, ab_stride, gzA, gzB, lv, nshots);
        case 4: return launch_tti_step_r<T, 4>(p, k, q, fwd, A, B, ab_stride, gzA, gzB, lv, nshots);
        case 6: return launch_tti_step_r<T, 6>(p, k, q, fwd, A, B, ab_stride, gzA, gzB, lv, nshots);
        case 8: return launch_tti_step_r<T, 8>(p, k, q, fwd, A, B, ab_stride, gzA, gzB, lv, nshots);
    }
    set_error("TTI needs space_order divisible by 4");
    return MIFWI_EINVAL;
}
template int launch_tti_step<float>(const Plan &, const StepConsts<float> &, const TtiParams<float> &, int,
                                    const TtiLevels<float> &, float *, float *, float *, float *, int, int &);
template int launch_tti_step<double>(const Plan &, const StepConsts<double> &, const TtiParams<double> &, int,
                                     const TtiLevels<double> &, double *, double *, double *, double *, int, int &);

}  // namespace mifwi
