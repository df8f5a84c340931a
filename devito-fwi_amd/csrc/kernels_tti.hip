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
// Tiled 2-D TTI step (float32): one launch advances both fields of every shot, adds the injected
// terms and samples the receivers.  A TX x TZ tile of the two stencilled fields is staged in LDS with a
// halo of R; Gz of both fields is formed on the tile extended by p = R/2 and its products with cos and
// sin of the tilt are kept in LDS (the angle fields themselves go through registers only); then every
// thread updates TX*TZ/256 points from LDS.  Same arithmetic, in the same order, as tti_gz_kernel +
// tti_step_kernel.
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
    // injection into both new fields (slot_count == nullptr: none): terms winj[shot][time][slot_base[tile] + j]
    const int32_t *slot_count, *slot_local, *slot_base;
    int inj_cap, inj_per_shot;
    const float *winj;
    long winj_stride;
    // interpolation of ucur + vcur (pt_idx == nullptr: none)
    const int32_t *pt_idx;
    const float *pt_w;
    int itp_npoint, itp_per_shot;
    float *vals_out;
    int time, nt;
    float inv_dt2;
    TtiExtra<float> x;   // MODE 1: gradient members, MODE 2: Born members
};

// LDS row pitch.  With TZ = 32 a wave covers rows ty and ty + 4 of the tile, so the two half-waves hit
// disjoint halves of the 64 banks when 4 * pitch = 32 (mod 64), i.e. pitch = 8 (mod 16).  With TZ = 64 a
// wave is one row and any pitch is conflict free.
__host__ __device__ constexpr int tti_pitch(int n, int tz)
{
    return tz == 32 ? (n + 7) / 16 * 16 + 8 : n;
}

// MODE 0: plain step, 1: adjoint step + gradient correlation, 2: forward step + Born source term
template <int R, int TX, int TZ, bool FWD, int MODE>
__global__ __launch_bounds__(256) void tti2d_kernel(Tti2DArgs a)
{
    constexpr int P = R / 2, NT = 256;
    constexpr int LA = TZ + 2 * R, HA = TX + 2 * R;   // staged fields: logical row length, rows
    constexpr int LE = TZ + 2 * P, HE = TX + 2 * P;   // extended tile of Gz
    constexpr int WA = tti_pitch(LA, TZ), WE = tti_pitch(LE, TZ);
    constexpr int NPT = TX * TZ / NT;                 // points per thread
    constexpr int RP = NT / TZ;                       // tile rows covered by one pass of the block
    constexpr int NEXT = (HE * LE + NT - 1) / NT;     // Gz points per thread
    constexpr int NSTG = (HA * LA + NT - 1) / NT;     // staged points per thread
    static_assert(WA >= LA && WE >= LE, "pitch");
    __shared__ float sA[HA * WA], sB[HA * WA];
    __shared__ float cgA[HE * WE], sgA[HE * WE], cgB[HE * WE], sgB[HE * WE];
    __shared__ float acc[TX * TZ];

    // Everything the first loads of a block need is fetched from the kernel arguments in one batch: left to be taken
    // where they are used, behind the branches of the block's role and of the tile decoding, the arguments came in
    // seven dependent steps, each with a wait of its own, before the first vector load went out.
    asm volatile("" ::"s"(a.pt_idx), "s"(a.pt_w), "s"(a.itp_npoint), "s"(a.itp_per_shot), "s"(a.nshots), "s"(a.ntz),
                 "s"(a.ntx), "s"(a.shot_stride), "s"(a.ucur), "s"(a.vcur), "s"(a.uold), "s"(a.vold), "s"(a.inj_per_shot),
                 "s"(a.slot_count));
    asm volatile("" ::"s"(a.n1), "s"(a.n2), "s"(a.off0), "s"(a.s1), "s"(a.m), "s"(a.eta), "s"(a.et), "s"(a.dl), "s"(a.ct),
                 "s"(a.st));
    const int tid = threadIdx.x;
    const int nitp_blocks = a.pt_idx ? ((a.itp_npoint + NT - 1) / NT) * a.nshots : 0;
    int bid = blockIdx.x;
    if (bid < nitp_blocks) {
        // ---- interpolation role: the sum of the two current fields at the sparse points ----------
        const int shot = bid % a.nshots;
        const int pnt = (bid / a.nshots) * NT + tid;
        if (pnt >= a.itp_npoint) return;
        const int tab = a.itp_per_shot ? shot : 0;
        const int4 idx = *reinterpret_cast<const int4 *>(a.pt_idx + ((long)tab * a.itp_npoint + pnt) * 4);
        const float4 w = *reinterpret_cast<const float4 *>(a.pt_w + ((long)tab * a.itp_npoint + pnt) * 4);
        const float *f = a.ucur + (long)shot * a.shot_stride;
        const float *g = a.vcur + (long)shot * a.shot_stride;
        const float sum = w.x * (f[idx.x] + g[idx.x]) + w.y * (f[idx.y] + g[idx.y]) + w.z * (f[idx.z] + g[idx.z]) +
                          w.w * (f[idx.w] + g[idx.w]);
        a.vals_out[((long)shot * a.nt + a.time) * a.itp_npoint + pnt] = sum;
        return;
    }
    bid -= nitp_blocks;

    const int shot = bid % a.nshots;
    const int tile = bid / a.nshots;
    const int tile_x = tile / a.ntz, tile_z = tile - tile_x * a.ntz;
    const int x0 = tile_x * TX, z0 = tile_z * TZ;
    const long sbase = (long)shot * a.shot_stride;
    const float *ucur = a.ucur + sbase, *vcur = a.vcur + sbase;
    const int inj_tab = a.inj_per_shot ? shot : 0;
    // read through the constant address space: a scalar load, whose wait does not hold vector loads back (the table is
    // written before the launch; behind the `asm` above the compiler no longer proves that by itself)
    typedef const __attribute__((address_space(4))) int32_t *ConstI32;
    const int ninj = a.slot_count ? ((ConstI32)(unsigned long)a.slot_count)[(long)inj_tab * (a.ntx * a.ntz) + tile] : 0;   // block-uniform

    // ---- this thread's points: start their loads first ----------------------------------------------
    const int tz = tid & (TZ - 1);
    const int ty = TZ == 32 ? (tid >> 6) + 4 * ((tid >> 5) & 1) : (tid >> 6);
    const int z = z0 + tz;
    // Every load of the block is issued before the first wait: a point outside the grid reads grid point (0, 0)
    // instead of branching around its load (behind a branch each load waited for the one before it: NSTG + 2 round
    // trips to memory per block instead of one), and its value is replaced afterwards.
    float pm[NPT], pe[NPT], pet[NPT], pdl[NPT], puo[NPT], pvo[NPT], puc[NPT], pvc[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int x = x0 + ty + RP * k;
        const bool ok = x < a.n1 && z < a.n2;
        const long i = ok ? a.off0 + (long)x * a.s1 + z : a.off0;
        pm[k] = a.m[i], pe[k] = a.eta[i];
        puo[k] = a.uold[sbase + i], pvo[k] = a.vold[sbase + i];
        pet[k] = 1.f, pdl[k] = 1.f, puc[k] = 0.f, pvc[k] = 0.f;
        if (FWD) {
            pet[k] = a.et[i], pdl[k] = a.dl[i];
        } else {
            puc[k] = ucur[i], pvc[k] = vcur[i];
        }
    }
    // angle fields at this thread's points of the extended tile
    float ect[NEXT], est[NEXT];
#pragma unroll
    for (int n = 0; n < NEXT; ++n) {
        const int e = tid + NT * n;
        const int r = e / LE, c = e - r * LE;
        const int gx = x0 - P + r, gz = z0 - P + c;
        const bool ok = e < HE * LE && gx < a.n1 + P && gz < a.n2 + P;
        const long i = ok ? a.off0 + (long)gx * a.s1 + gz : a.off0;
        const float cv = a.ct[i], sv = a.st[i];
        ect[n] = ok ? cv : 1.f, est[n] = ok ? sv : 0.f;
    }

    // ---- stage the two stencilled fields (halo R) --------------------------------------------------
    {
        float su[NSTG], sv[NSTG], set[NSTG], sdl[NSTG];
#pragma unroll
        for (int n = 0; n < NSTG; ++n) {
            const int e = tid + NT * n;
            const int r = e / LA, c = e - r * LA;
            const int gx = x0 - R + r, gz = z0 - R + c;
            const bool ok = e < HA * LA && gx < a.n1 + R && gz < a.n2 + R;
            const long i = ok ? a.off0 + (long)gx * a.s1 + gz : a.off0;
            su[n] = ucur[i], sv[n] = vcur[i];
            if (!FWD) set[n] = a.et[i], sdl[n] = a.dl[i];
        }
#pragma unroll
        for (int n = 0; n < NSTG; ++n) {
            const int e = tid + NT * n;
            const int r = e / LA, c = e - r * LA;
            const int gx = x0 - R + r, gz = z0 - R + c;
            if (e < HA * LA) {
                const bool ok = gx < a.n1 + R && gz < a.n2 + R;
                float fa = su[n], fb = sv[n];
                if (!FWD) {
                    fa = set[n] * su[n] + sdl[n] * sv[n];
                    fb = sdl[n] * su[n] + sv[n];
                }
                sA[r * WA + c] = ok ? fa : 0.f;
                sB[r * WA + c] = ok ? fb : 0.f;
            }
        }
    }
    if (ninj > 0) {
        for (int j = tid; j < TX * TZ; j += NT) acc[j] = 0.f;
        __syncthreads();
        const long tslot = (long)inj_tab * (a.ntx * a.ntz) + tile;
        const float *w = a.winj + ((long)shot * a.nt + a.time) * a.winj_stride + a.slot_base[tslot];
        for (int j = tid; j < ninj; j += NT) acc[a.slot_local[tslot * a.inj_cap + j]] = w[j];
    }
    __syncthreads();

    // ---- Gz of both fields on the extended tile, times cos and sin of the tilt ------------------------
#pragma unroll
    for (int n = 0; n < NEXT; ++n) {
        const int e = tid + NT * n;
        const int r = e / LE, c = e - r * LE;
        if (e < HE * LE) {
            const int ia = (r + R - P) * WA + (c + R - P);
            float dxa = 0.f, dza = 0.f, dxb = 0.f, dzb = 0.f;
#pragma unroll
            for (int j = 1; j <= P; ++j) {
                dxa += a.a1[j] * (sA[ia + j * WA] - sA[ia - j * WA]);
                dza += a.a1[j] * (sA[ia + j] - sA[ia - j]);
                dxb += a.a1[j] * (sB[ia + j * WA] - sB[ia - j * WA]);
                dzb += a.a1[j] * (sB[ia + j] - sB[ia - j]);
            }
            const float ga = -(est[n] * (dxa * a.ihx) + ect[n] * (dza * a.ihz));
            const float gb = -(est[n] * (dxb * a.ihx) + ect[n] * (dzb * a.ihz));
            cgA[r * WE + c] = ect[n] * ga, sgA[r * WE + c] = est[n] * ga;
            cgB[r * WE + c] = ect[n] * gb, sgB[r * WE + c] = est[n] * gb;
        }
    }
    __syncthreads();

    // ---- update this thread's points ----------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int row = ty + RP * k;
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
            float ta = a.a1[j] * (cgA[ie + j] - cgA[ie - j]) * a.ihz;
            ta += a.a1[j] * (sgA[ie + j * WE] - sgA[ie - j * WE]) * a.ihx;
            sa += ta;
            float tb = a.a1[j] * (cgB[ie + j] - cgB[ie - j]) * a.ihz;
            tb += a.a1[j] * (sgB[ie + j * WE] - sgB[ie - j * WE]) * a.ihx;
            sb += tb;
        }
        const float X = lap + sa, Y = -sb;   // Gxx = laplace - Gzz with Gzz = -sa
        float H0 = X, Hz = Y;
        float uc = puc[k], vc = pvc[k];
        if (FWD) {
            H0 = pet[k] * X + pdl[k] * Y;
            Hz = pdl[k] * X + Y;
            uc = sA[ia], vc = sB[ia];
        }
        const float mz = pm[k], ed = pe[k] * a.dt;
        float un = (a.dt2 * H0 + 2.0f * mz * uc - mz * puo[k] + ed * uc) / (mz + ed);
        float vn = (a.dt2 * Hz + 2.0f * mz * vc - mz * pvo[k] + ed * vc) / (mz + ed);
        const long i = a.off0 + (long)x * a.s1 + z;
        if (MODE == 2) {
            const long bi = (long)shot * a.x.b_shot_stride + i;
            const float dmv = a.x.b_dm[i];
            const float utt = (a.x.b_un[bi] - 2.0f * a.x.b_uc[bi] + a.x.b_uo[bi]) * a.inv_dt2;
            const float vtt = (a.x.b_vn[bi] - 2.0f * a.x.b_vc[bi] + a.x.b_vo[bi]) * a.inv_dt2;
            un += a.dt2 * (-dmv * utt) / (mz + ed);
            vn += a.dt2 * (-dmv * vtt) / (mz + ed);
        }
        if (ninj > 0) {
            const float add = acc[row * TZ + tz];
            un += add, vn += add;
        }
        a.uout[sbase + i] = un;
        a.vout[sbase + i] = vn;
        if (MODE == 1) {
            // summed by parts over time: the second time difference is taken of the adjoint fields (all
            // three levels are in registers) instead of the saved ones, so one saved level of each
            // background field is read per step instead of three
            const long gi = (long)shot * a.x.s_shot_stride + i;
            const float ptt = (puo[k] - 2.0f * uc + un) * a.inv_dt2;
            const float rtt = (pvo[k] - 2.0f * vc + vn) * a.inv_dt2;
            float *g = a.x.grad + (long)shot * a.x.g_shot_stride + i;
            *g += -(ptt * a.x.su[gi] + rtt * a.x.sv[gi]);
        }
    }
}

template <int R, int TX, int TZ>
static void launch_tti2d_rt(const Plan &p, const Tti2DArgs &a, int fwd, int nblocks)
{
    if (a.x.grad)
        hipLaunchKernelGGL((tti2d_kernel<R, TX, TZ, false, 1>), dim3(nblocks), dim3(256), 0, p.stream, a);
    else if (a.x.b_un)
        hipLaunchKernelGGL((tti2d_kernel<R, TX, TZ, true, 2>), dim3(nblocks), dim3(256), 0, p.stream, a);
    else if (fwd)
        hipLaunchKernelGGL((tti2d_kernel<R, TX, TZ, true, 0>), dim3(nblocks), dim3(256), 0, p.stream, a);
    else
        hipLaunchKernelGGL((tti2d_kernel<R, TX, TZ, false, 0>), dim3(nblocks), dim3(256), 0, p.stream, a);
}

template <int R>
static int launch_tti2d_r(const Plan &p, const Tti2DArgs &a, int fwd)
{
    const int nblocks = a.ntx * a.ntz * a.nshots + (a.pt_idx ? ((a.itp_npoint + 255) / 256) * a.nshots : 0);
    if ((a.x.grad && fwd) || (a.x.b_un && !fwd)) {
        set_error("TTI step: the gradient term belongs to the adjoint step, the Born term to the forward step");
        return MIFWI_EINVAL;
    }
    if (p.tile_tz == 64) launch_tti2d_rt<R, 16, 64>(p, a, fwd, nblocks);
    else if (p.tile_tx == 16) launch_tti2d_rt<R, 16, 32>(p, a, fwd, nblocks);
    else if (p.tile_tx == 8) launch_tti2d_rt<R, 8, 32>(p, a, fwd, nblocks);
    else launch_tti2d_rt<R, 32, 32>(p, a, fwd, nblocks);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

bool tti2d_available(const Plan &p)
{
    return p.kind == 1 && ((p.tile_tx == 32 && p.tile_tz == 32) || (p.tile_tx == 16 && p.tile_tz == 64) ||
                           ((p.tile_tx == 16 || p.tile_tx == 8) && p.tile_tz == 32));
}

static int launch_tti2d(const Plan &p, const StepConsts<float> &k, const TtiParams<float> &q, int fwd,
                        const TtiLevels<float> &lv, int nshots, const Step2DArgs &sp, const TtiExtra<float> *extra)
{
    Tti2DArgs a{};
    if (extra) a.x = *extra;
    a.inv_dt2 = k.inv_dt2;
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
    a.ntx = p.ntx, a.ntz = p.ntz, a.nshots = nshots;
    a.slot_count = sp.slot_count, a.slot_local = sp.slot_local, a.slot_base = sp.slot_base;
    a.inj_cap = sp.inj_cap, a.inj_per_shot = sp.inj_per_shot, a.winj = sp.winj, a.winj_stride = sp.winj_stride;
    a.pt_idx = sp.pt_idx, a.pt_w = sp.pt_w, a.itp_npoint = sp.itp_npoint, a.itp_per_shot = sp.itp_per_shot;
    a.vals_out = sp.vals_out, a.time = sp.time, a.nt = sp.nt;
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

// ------------------------------------------------------------------------------------------------
// 3-D float32 TTI step as marches along x (the arithmetic of tti_gz_kernel + tti_step_kernel).  A 256-thread
// block owns a 16 x 64 tile of the (y, z) plane and a chunk of planes; a thread owns four consecutive z points of
// one row and keeps the planes it needs again along x in registers.  The y and z neighbours come straight from
// memory as aligned 16-byte loads -- rows y +- j, and the neighbouring pieces of the z line -- which neighbouring
// threads of the block request at the same time.
//   Two marches (tti3d_gz_kernel, then tti3d_step_kernel or tti3d_step_wide_kernel; space orders 12 and 16, and 4
// and 8 on request): no LDS, no barrier.  Stores cost about twice what loads cost on this part, so the first march
// stores Gz of the two fields only and the second forms the products with the direction cosines (wx = st cp and
// wy = st sp, tabulated once per model; wz = ct) at the neighbouring points itself.
//   One march (tti3d_fused_kernel; space orders 4 and 8): Gz is formed a few planes ahead of the update and handed to
// the neighbouring threads through LDS, never stored.
// ------------------------------------------------------------------------------------------------
#ifndef TTI3D_TY
#define TTI3D_TY 16   // rows of a tile of the marching kernels (16 threads of four points per row)
#endif
typedef float tf4 __attribute__((ext_vector_type(4)));

// The loads of the two marches are issued by hand, in batches, each followed by one wait: left to the compiler's
// scheduler (which, short of registers, moves every load down to its first use) the loop made some twenty round trips
// to memory per plane.  gld issues a 16-byte load at sbase (uniform) + voff bytes + OFF; vm_wait<N> waits until at
// most N of the loads issued since are outstanding (loads return in order); arrived() marks a register as written
// at that point, so that nothing that reads it is moved above the wait.
template <int OFF>
__device__ inline void gld(tf4 &v, unsigned voff, const float *sbase)
{
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// the same load marked non-temporal: a stream that is read once and should not displace the lines tiles share
template <int OFF>
__device__ inline void gld_nt(tf4 &v, unsigned voff, const float *sbase)
{
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(v) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int N>
__device__ inline void vm_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ inline void arrived(tf4 &v) { asm volatile("" : "+v"(v)); }

// Workgroups go to the eight XCDs in turn and every XCD has an L2 of its own.  Tiles that are neighbours read each
// other's edge rows and edge cache lines, so an XCD is given a contiguous run of the tile order: block b is the
// (b / 8)-th block of XCD b % 8.
__device__ inline int xcd_contiguous_block()
{
    const int nb = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nb >> 3, r = nb & 7;
    return xcd * q + min(xcd, r) + (b >> 3);
}

__global__ __launch_bounds__(256) void tti3d_weights_kernel(const float *st, const float *cp, const float *sp, float *wx,
                                                            float *wy, long n)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += 256L * gridDim.x) wx[i] = st[i] * cp[i], wy[i] = st[i] * sp[i];
}

// First march: Gz of A and of B on the grid extended by p points per dimension (gz: [2][shots][field]).  The tile
// starts at the left end of the row padding, so every thread's four points are one aligned 16-byte piece; the pieces
// that lie outside the extended grid are computed from padding and stored, and nothing reads them.
struct Tti3DGzArgs {
    const float *A, *B;
    long ab_stride;
    float *gz;
    long gz_stride;            // elements between Gz(A) and Gz(B)
    const float *wx, *wy, *wz;
    float a[kMaxR / 2 + 1], ih[3];
    int n0, n1, n2, e, zl;
    long s0, s1, off0, field;
    int nty, ntz, nchunk, xchunk, nshots;
};

template <int P>
__global__ __launch_bounds__(16 * TTI3D_TY) void tti3d_gz_kernel(Tti3DGzArgs a)
{
    constexpr int TY = TTI3D_TY, TZ = 64, V = 4;
    int t = xcd_contiguous_block();
    const int tile_z = t % a.ntz;
    t /= a.ntz;
    const int tile_y = t % a.nty;
    t /= a.nty;
    const int chunk = t % a.nchunk;
    const int shot = t / a.nchunk;
    const int xa = -a.e + chunk * a.xchunk, xb = min(a.n0 + a.e, xa + a.xchunk);
    const int tid = threadIdx.x;
    const int yy = tile_y * TY + tid / (TZ / V), zz = tile_z * TZ + V * (tid % (TZ / V));   // from (-e, -zl)
    if (yy >= a.n1 + 2 * a.e || zz >= a.n2 + a.e + a.zl || xa >= xb) return;
    const unsigned voff = (unsigned)((long)yy * a.s1 + zz) * 4u;   // bytes from the first point of the extended plane
    const long org = a.off0 - (long)a.e * a.s1 - a.zl;             // that point, in plane 0
    const float *A = a.A + (long)shot * a.ab_stride + org, *B = a.B + (long)shot * a.ab_stride + org;
    float *o = a.gz + (long)shot * a.field + org + voff / 4;

    tf4 qa[2 * P + 1], qb[2 * P + 1];   // planes x - P .. x + P (after the shift)
#pragma unroll
    for (int j = 0; j < 2 * P; ++j) {
        gld<0>(qa[j + 1], voff, A + (long)(xa - P + j) * a.s0);
        gld<0>(qb[j + 1], voff, B + (long)(xa - P + j) * a.s0);
    }
    vm_wait<0>();
#pragma unroll
    for (int j = 1; j <= 2 * P; ++j) arrived(qa[j]), arrived(qb[j]);
    for (int x = xa; x < xb; ++x) {
        const long px = (long)x * a.s0;
        const float *Ap = A + px, *Bp = B + px;
        tf4 na, nb, wx, wy, wz, ya[2 * P], yb[2 * P], za[2], zb[2];
        gld<0>(na, voff, Ap + (long)P * a.s0);
        gld<0>(nb, voff, Bp + (long)P * a.s0);
#pragma unroll
        for (int j = 1; j <= P; ++j) {
            gld<0>(ya[2 * j - 2], voff, Ap - (long)j * a.s1);
            gld<0>(ya[2 * j - 1], voff, Ap + (long)j * a.s1);
            gld<0>(yb[2 * j - 2], voff, Bp - (long)j * a.s1);
            gld<0>(yb[2 * j - 1], voff, Bp + (long)j * a.s1);
        }
        gld<-16>(za[0], voff, Ap);
        gld<16>(za[1], voff, Ap);
        gld<-16>(zb[0], voff, Bp);
        gld<16>(zb[1], voff, Bp);
        gld<0>(wx, voff, a.wx + org + px);
        gld<0>(wy, voff, a.wy + org + px);
        gld<0>(wz, voff, a.wz + org + px);
        vm_wait<0>();
#pragma unroll
        for (int j = 0; j < 2 * P; ++j) qa[j] = qa[j + 1], qb[j] = qb[j + 1];
#pragma unroll
        for (int j = 0; j < 2 * P; ++j) arrived(ya[j]), arrived(yb[j]);
        arrived(na), arrived(nb), arrived(za[0]), arrived(za[1]), arrived(zb[0]), arrived(zb[1]);
        arrived(wx), arrived(wy), arrived(wz);
        qa[2 * P] = na, qb[2 * P] = nb;
        tf4 dxa = {0.f, 0.f, 0.f, 0.f}, dxb = dxa, dya = dxa, dyb = dxa, dza = dxa, dzb = dxa;
#pragma unroll
        for (int j = 1; j <= P; ++j) {
            dxa += a.a[j] * (qa[P + j] - qa[P - j]);
            dxb += a.a[j] * (qb[P + j] - qb[P - j]);
            dya += a.a[j] * (ya[2 * j - 1] - ya[2 * j - 2]);
            dyb += a.a[j] * (yb[2 * j - 1] - yb[2 * j - 2]);
        }
        {
            float la[V + 8], lb[V + 8];   // z - 4 .. z + 7
#pragma unroll
            for (int i = 0; i < V; ++i) {
                la[i] = za[0][i], la[4 + i] = qa[P][i], la[8 + i] = za[1][i];
                lb[i] = zb[0][i], lb[4 + i] = qb[P][i], lb[8 + i] = zb[1][i];
            }
#pragma unroll
            for (int j = 1; j <= P; ++j)
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    dza[i] += a.a[j] * (la[4 + i + j] - la[4 + i - j]);
                    dzb[i] += a.a[j] * (lb[4 + i + j] - lb[4 + i - j]);
                }
        }
        *reinterpret_cast<tf4 *>(o + px) = -(wx * (dxa * a.ih[0]) + wy * (dya * a.ih[1]) + wz * (dza * a.ih[2]));
        *reinterpret_cast<tf4 *>(o + px + a.gz_stride) = -(wx * (dxb * a.ih[0]) + wy * (dyb * a.ih[1]) + wz * (dzb * a.ih[2]));
    }
}

// Second march: X = laplace(A) - Gzz(A), Y = Gzz(B) and the update of both fields.  Registers hold the planes
// x - R .. x + R of A and the planes x - p .. x + p of the x products wx Gz of both fields.
struct Tti3DArgs {
    const float *A, *B;        // stencilled fields of shot 0 (forward: u, v; adjoint: et p + dl r, dl p + r)
    long a_stride;
    const float *gz;           // [2][shots][field] (two marches)
    long gz_stride;
    const float *wx, *wy, *wz;
    const float *et, *dl, *m, *eta;
    const float *ucur, *uold, *vcur, *vold;
    float *uout, *vout;
    long lv_stride;
    float c0, cx[kMaxR + 1], cy[kMaxR + 1], cz[kMaxR + 1];
    float a1[kMaxR / 2 + 1], ih[3];
    float dt, dt2;
    int n0, n1, n2;
    long s0, s1, off0, field;
    int nty, ntz, nchunk, xchunk, nshots, fwd;
};

template <int R>
__global__ __launch_bounds__(16 * TTI3D_TY) void tti3d_step_kernel(Tti3DArgs a)
{
    constexpr int P = R / 2, TY = TTI3D_TY, TZ = 64, V = 4;
    constexpr int NZ4 = (R + 3) / 4;   // 16-byte pieces of the z line of A on either side
    int t = xcd_contiguous_block();
    const int tile_z = t % a.ntz;
    t /= a.ntz;
    const int tile_y = t % a.nty;
    t /= a.nty;
    const int chunk = t % a.nchunk;
    const int shot = t / a.nchunk;
    const int xa = chunk * a.xchunk, xb = min(a.n0, xa + a.xchunk);
    const int tid = threadIdx.x;
    const int y = tile_y * TY + tid / (TZ / V), z = tile_z * TZ + V * (tid % (TZ / V));
    if (y >= a.n1 || z >= a.n2 || xa >= xb) return;
    const unsigned voff = (unsigned)((long)y * a.s1 + z) * 4u;   // bytes from point (x, 0, 0) of a plane
    const float *A = a.A + (long)shot * a.a_stride + a.off0;
    const float *gA = a.gz + (long)shot * a.field + a.off0, *gB = gA + a.gz_stride;
    const float *wx = a.wx + a.off0, *wy = a.wy + a.off0, *wz = a.wz + a.off0;
    const long lv = (long)shot * a.lv_stride + a.off0;

    // qa[j]: A at plane x - R + j; qxa / qxb[j]: wx Gz of the two fields at plane x - P + j (after the shift)
    tf4 qa[2 * R + 1], qxa[2 * P + 1], qxb[2 * P + 1];
    {
        tf4 w[2 * P];
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) gld<0>(qa[j + 1], voff, A + (long)(xa - R + j) * a.s0);
#pragma unroll
        for (int j = 0; j < 2 * P; ++j) {
            const long o = (long)(xa - P + j) * a.s0;
            gld<0>(w[j], voff, wx + o);
            gld<0>(qxa[j + 1], voff, gA + o);
            gld<0>(qxb[j + 1], voff, gB + o);
        }
        vm_wait<0>();
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) arrived(qa[j + 1]);
#pragma unroll
        for (int j = 0; j < 2 * P; ++j) {
            arrived(w[j]), arrived(qxa[j + 1]), arrived(qxb[j + 1]);
            qxa[j + 1] *= w[j], qxb[j + 1] *= w[j];
        }
    }
    // batch 1 of a plane: the new planes of the queues, rows and line of A.  (Issued a plane ahead it would overlap
    // the update, but its registers then live across the loop edge and the compiler copies them there, in flight.)
    tf4 na, nw, nga, ngb, yA[2 * R], zA[2 * NZ4];
    auto issue_b1 = [&](long px) {
        const float *Ap = A + px;
        gld<0>(na, voff, Ap + (long)R * a.s0);
        gld<0>(nw, voff, wx + px + (long)P * a.s0);
        gld<0>(nga, voff, gA + px + (long)P * a.s0);
        gld<0>(ngb, voff, gB + px + (long)P * a.s0);
#pragma unroll
        for (int k = 1; k <= R; ++k) {
            gld<0>(yA[2 * k - 2], voff, Ap - (long)k * a.s1);
            gld<0>(yA[2 * k - 1], voff, Ap + (long)k * a.s1);
        }
        gld<-16>(zA[0], voff, Ap);
        gld<16>(zA[1], voff, Ap);
        if constexpr (NZ4 == 2) {
            gld<-32>(zA[2], voff, Ap);
            gld<32>(zA[3], voff, Ap);
        }
    };
    auto b1_arrived = [&]() {
        arrived(na), arrived(nw), arrived(nga), arrived(ngb);
#pragma unroll
        for (int k = 0; k < 2 * R; ++k) arrived(yA[k]);
#pragma unroll
        for (int k = 0; k < 2 * NZ4; ++k) arrived(zA[k]);
    };
    for (int x = xa; x < xb; ++x) {
        const long px = (long)x * a.s0;
        const float *gAp = gA + px, *gBp = gB + px, *wyp = wy + px, *wzp = wz + px;
        issue_b1(px);
        // batch 2: rows of wy, Gz(A), Gz(B)
        tf4 yw[2 * P], yga[2 * P], ygb[2 * P];
#pragma unroll
        for (int j = 1; j <= P; ++j) {
            const long jy = (long)j * a.s1;
            gld<0>(yw[2 * j - 2], voff, wyp - jy);
            gld<0>(yw[2 * j - 1], voff, wyp + jy);
            gld<0>(yga[2 * j - 2], voff, gAp - jy);
            gld<0>(yga[2 * j - 1], voff, gAp + jy);
            gld<0>(ygb[2 * j - 2], voff, gBp - jy);
            gld<0>(ygb[2 * j - 1], voff, gBp + jy);
        }
        vm_wait<6 * P>();   // batch 1 has arrived
        b1_arrived();
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) qa[j] = qa[j + 1];
        qa[2 * R] = na;
#pragma unroll
        for (int j = 0; j < 2 * P; ++j) qxa[j] = qxa[j + 1], qxb[j] = qxb[j + 1];
        qxa[2 * P] = nw * nga, qxb[2 * P] = nw * ngb;
        // Laplacian of A in the order z, y, x
        const tf4 ac = qa[R];
        tf4 lap = a.c0 * ac;
        {
            float line[V + 8 * NZ4];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                line[4 * NZ4 + i] = ac[i];
                line[4 * NZ4 - 4 + i] = zA[0][i], line[4 * NZ4 + 4 + i] = zA[1][i];
                if constexpr (NZ4 == 2) line[i] = zA[2][i], line[16 + i] = zA[3][i];
            }
#pragma unroll
            for (int k = 1; k <= R; ++k)
#pragma unroll
                for (int i = 0; i < V; ++i) lap[i] += a.cz[k] * (line[4 * NZ4 + i - k] + line[4 * NZ4 + i + k]);
        }
#pragma unroll
        for (int k = 1; k <= R; ++k) lap += a.cy[k] * (yA[2 * k - 2] + yA[2 * k - 1]);
#pragma unroll
        for (int k = 1; k <= R; ++k) lap += a.cx[k] * (qa[R - k] + qa[R + k]);

        // batch 3: the z line of wz, Gz(A), Gz(B); batch 4: the pointwise streams of this plane
        tf4 zw[3], zga[3], zgb[3];
        gld<-16>(zw[0], voff, wzp);
        gld<0>(zw[1], voff, wzp);
        gld<16>(zw[2], voff, wzp);
        gld<-16>(zga[0], voff, gAp);
        gld<0>(zga[1], voff, gAp);
        gld<16>(zga[2], voff, gAp);
        gld<-16>(zgb[0], voff, gBp);
        gld<0>(zgb[1], voff, gBp);
        gld<16>(zgb[2], voff, gBp);
        tf4 et, dl, mm, ee, uc, uo, vc, vo;
        gld<0>(et, voff, a.et + a.off0 + px);
        gld<0>(dl, voff, a.dl + a.off0 + px);
        gld<0>(mm, voff, a.m + a.off0 + px);
        gld<0>(ee, voff, a.eta + a.off0 + px);
        gld<0>(uc, voff, a.ucur + lv + px);
        gld<0>(uo, voff, a.uold + lv + px);
        gld<0>(vc, voff, a.vcur + lv + px);
        gld<0>(vo, voff, a.vold + lv + px);
        vm_wait<17>();   // batch 2 has arrived
        tf4 tyA[P], tyB[P];
#pragma unroll
        for (int j = 0; j < 2 * P; ++j) arrived(yw[j]), arrived(yga[j]), arrived(ygb[j]);
#pragma unroll
        for (int j = 1; j <= P; ++j) {
            tyA[j - 1] = a.a1[j] * (yw[2 * j - 1] * yga[2 * j - 1] - yw[2 * j - 2] * yga[2 * j - 2]) * a.ih[1];
            tyB[j - 1] = a.a1[j] * (yw[2 * j - 1] * ygb[2 * j - 1] - yw[2 * j - 2] * ygb[2 * j - 2]) * a.ih[1];
        }
        vm_wait<8>();   // batch 3 has arrived
        float za[V + 8], zb[V + 8];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            arrived(zw[c]), arrived(zga[c]), arrived(zgb[c]);
            const tf4 va = zw[c] * zga[c], vb = zw[c] * zgb[c];
#pragma unroll
            for (int i = 0; i < V; ++i) za[4 * c + i] = va[i], zb[4 * c + i] = vb[i];
        }
        tf4 sA = {0.f, 0.f, 0.f, 0.f}, sB = sA;
#pragma unroll
        for (int j = 1; j <= P; ++j) {   // per j: z term, then x, then y, as in tti_gzz
            tf4 tA, tB;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                tA[i] = a.a1[j] * (za[4 + i + j] - za[4 + i - j]) * a.ih[2];
                tB[i] = a.a1[j] * (zb[4 + i + j] - zb[4 + i - j]) * a.ih[2];
            }
            tA += a.a1[j] * (qxa[P + j] - qxa[P - j]) * a.ih[0];
            tB += a.a1[j] * (qxb[P + j] - qxb[P - j]) * a.ih[0];
            tA += tyA[j - 1], tB += tyB[j - 1];
            sA += tA, sB += tB;
        }
        const tf4 X = lap + sA, Y = -sB;   // Gzz = -s
        vm_wait<0>();   // batch 4 has arrived
        arrived(et), arrived(dl), arrived(mm), arrived(ee), arrived(uc), arrived(uo), arrived(vc), arrived(vo);
        const tf4 H0 = a.fwd ? et * X + dl * Y : X;
        const tf4 Hz = a.fwd ? dl * X + Y : Y;
        tf4 un, vn;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float ed = ee[i] * a.dt;
            un[i] = (a.dt2 * H0[i] + 2.0f * mm[i] * uc[i] - mm[i] * uo[i] + ed * uc[i]) / (mm[i] + ed);
            vn[i] = (a.dt2 * Hz[i] + 2.0f * mm[i] * vc[i] - mm[i] * vo[i] + ed * vc[i]) / (mm[i] + ed);
        }
        float *uo_p = a.uout + lv + px + voff / 4, *vo_p = a.vout + lv + px + voff / 4;
        if (z + V - 1 < a.n2) {
            *reinterpret_cast<tf4 *>(uo_p) = un;
            *reinterpret_cast<tf4 *>(vo_p) = vn;
        } else {
            for (int i = 0; i < V; ++i)
                if (z + i < a.n2) uo_p[i] = un[i], vo_p[i] = vn[i];
        }
    }
}

// The same step at radius 6 and 8, where the plane queues alone take 108 and 140 registers: the loads of a plane go
// out in five smaller groups (at most 17 in flight), each consumed before the next is issued, and the y terms of Gzz
// are folded in two rows at a time.  Same arithmetic, in the same order.
template <int R>
__global__ __launch_bounds__(16 * TTI3D_TY) void tti3d_step_wide_kernel(Tti3DArgs a)
{
    constexpr int P = R / 2, TY = TTI3D_TY, TZ = 64, V = 4, NZ4 = 2, H = R / 2;
    static_assert(R == 6 || R == 8, "radius");
    int t = xcd_contiguous_block();
    const int tile_z = t % a.ntz;
    t /= a.ntz;
    const int tile_y = t % a.nty;
    t /= a.nty;
    const int chunk = t % a.nchunk;
    const int shot = t / a.nchunk;
    const int xa = chunk * a.xchunk, xb = min(a.n0, xa + a.xchunk);
    const int tid = threadIdx.x;
    const int y = tile_y * TY + tid / (TZ / V), z = tile_z * TZ + V * (tid % (TZ / V));
    if (y >= a.n1 || z >= a.n2 || xa >= xb) return;
    const unsigned voff = (unsigned)((long)y * a.s1 + z) * 4u;
    const float *A = a.A + (long)shot * a.a_stride + a.off0;
    const float *gA = a.gz + (long)shot * a.field + a.off0, *gB = gA + a.gz_stride;
    const float *wx = a.wx + a.off0, *wy = a.wy + a.off0, *wz = a.wz + a.off0;
    const long lv = (long)shot * a.lv_stride + a.off0;

    tf4 qa[2 * R + 1], qxa[2 * P + 1], qxb[2 * P + 1];
#pragma unroll
    for (int j = 0; j < 2 * R; ++j) gld<0>(qa[j + 1], voff, A + (long)(xa - R + j) * a.s0);
    vm_wait<0>();
#pragma unroll
    for (int j = 0; j < 2 * R; ++j) arrived(qa[j + 1]);
#pragma unroll
    for (int j = 0; j < 2 * P; ++j) {
        const long o = (long)(xa - P + j) * a.s0;
        tf4 w;
        gld<0>(w, voff, wx + o);
        gld<0>(qxa[j + 1], voff, gA + o);
        gld<0>(qxb[j + 1], voff, gB + o);
        vm_wait<0>();
        arrived(w), arrived(qxa[j + 1]), arrived(qxb[j + 1]);
        qxa[j + 1] *= w, qxb[j + 1] *= w;
    }
    for (int x = xa; x < xb; ++x) {
        const long px = (long)x * a.s0;
        const float *Ap = A + px, *gAp = gA + px, *gBp = gB + px, *wyp = wy + px, *wzp = wz + px;
        tf4 lap;
        {
            // group 1: the new planes of the queues, the z line of A and its nearer rows
            tf4 na, nw, nga, ngb, zA[4], yA[2 * H];
            gld<0>(na, voff, Ap + (long)R * a.s0);
            gld<0>(nw, voff, wx + px + (long)P * a.s0);
            gld<0>(nga, voff, gAp + (long)P * a.s0);
            gld<0>(ngb, voff, gBp + (long)P * a.s0);
            gld<-16>(zA[0], voff, Ap);
            gld<16>(zA[1], voff, Ap);
            gld<-32>(zA[2], voff, Ap);
            gld<32>(zA[3], voff, Ap);
#pragma unroll
            for (int k = 1; k <= H; ++k) {
                gld<0>(yA[2 * k - 2], voff, Ap - (long)k * a.s1);
                gld<0>(yA[2 * k - 1], voff, Ap + (long)k * a.s1);
            }
            vm_wait<0>();
            arrived(na), arrived(nw), arrived(nga), arrived(ngb);
#pragma unroll
            for (int k = 0; k < 4; ++k) arrived(zA[k]);
#pragma unroll
            for (int k = 0; k < 2 * H; ++k) arrived(yA[k]);
#pragma unroll
            for (int j = 0; j < 2 * R; ++j) qa[j] = qa[j + 1];
            qa[2 * R] = na;
#pragma unroll
            for (int j = 0; j < 2 * P; ++j) qxa[j] = qxa[j + 1], qxb[j] = qxb[j + 1];
            qxa[2 * P] = nw * nga, qxb[2 * P] = nw * ngb;
            const tf4 ac = qa[R];
            lap = a.c0 * ac;
            float line[V + 8 * NZ4];
#pragma unroll
            for (int i = 0; i < V; ++i)
                line[i] = zA[2][i], line[4 + i] = zA[0][i], line[8 + i] = ac[i], line[12 + i] = zA[1][i], line[16 + i] = zA[3][i];
#pragma unroll
            for (int k = 1; k <= R; ++k)
#pragma unroll
                for (int i = 0; i < V; ++i) lap[i] += a.cz[k] * (line[8 + i - k] + line[8 + i + k]);
#pragma unroll
            for (int k = 1; k <= H; ++k) lap += a.cy[k] * (yA[2 * k - 2] + yA[2 * k - 1]);
        }
        float za[V + 8], zb[V + 8];
        {
            // group 2: the farther rows of A and the z line of wz, Gz(A), Gz(B)
            tf4 yA[2 * (R - H)], zw[3], zga[3], zgb[3];
#pragma unroll
            for (int k = H + 1; k <= R; ++k) {
                gld<0>(yA[2 * (k - H) - 2], voff, Ap - (long)k * a.s1);
                gld<0>(yA[2 * (k - H) - 1], voff, Ap + (long)k * a.s1);
            }
            gld<-16>(zw[0], voff, wzp);
            gld<0>(zw[1], voff, wzp);
            gld<16>(zw[2], voff, wzp);
            gld<-16>(zga[0], voff, gAp);
            gld<0>(zga[1], voff, gAp);
            gld<16>(zga[2], voff, gAp);
            gld<-16>(zgb[0], voff, gBp);
            gld<0>(zgb[1], voff, gBp);
            gld<16>(zgb[2], voff, gBp);
            vm_wait<9>();
#pragma unroll
            for (int k = 0; k < 2 * (R - H); ++k) arrived(yA[k]);
#pragma unroll
            for (int k = H + 1; k <= R; ++k) lap += a.cy[k] * (yA[2 * (k - H) - 2] + yA[2 * (k - H) - 1]);
#pragma unroll
            for (int k = 1; k <= R; ++k) lap += a.cx[k] * (qa[R - k] + qa[R + k]);
            vm_wait<0>();
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                arrived(zw[c]), arrived(zga[c]), arrived(zgb[c]);
                const tf4 va = zw[c] * zga[c], vb = zw[c] * zgb[c];
#pragma unroll
                for (int i = 0; i < V; ++i) za[4 * c + i] = va[i], zb[4 * c + i] = vb[i];
            }
        }
        // groups 3: rows of wy, Gz(A), Gz(B), two distances at a time; per j: z term, then x, then y, as in tti_gzz
        tf4 sA = {0.f, 0.f, 0.f, 0.f}, sB = sA;
#pragma unroll
        for (int j0 = 1; j0 <= P; j0 += 2) {
            constexpr int NJ = 2;
            tf4 yw[2 * NJ], yga[2 * NJ], ygb[2 * NJ];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = j0 + jj;
                if (j > P) continue;
                const long jy = (long)j * a.s1;
                gld<0>(yw[2 * jj], voff, wyp - jy);
                gld<0>(yw[2 * jj + 1], voff, wyp + jy);
                gld<0>(yga[2 * jj], voff, gAp - jy);
                gld<0>(yga[2 * jj + 1], voff, gAp + jy);
                gld<0>(ygb[2 * jj], voff, gBp - jy);
                gld<0>(ygb[2 * jj + 1], voff, gBp + jy);
            }
            vm_wait<0>();
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = j0 + jj;
                if (j > P) continue;
                arrived(yw[2 * jj]), arrived(yw[2 * jj + 1]), arrived(yga[2 * jj]), arrived(yga[2 * jj + 1]);
                arrived(ygb[2 * jj]), arrived(ygb[2 * jj + 1]);
                tf4 tA, tB;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    tA[i] = a.a1[j] * (za[4 + i + j] - za[4 + i - j]) * a.ih[2];
                    tB[i] = a.a1[j] * (zb[4 + i + j] - zb[4 + i - j]) * a.ih[2];
                }
                tA += a.a1[j] * (qxa[P + j] - qxa[P - j]) * a.ih[0];
                tB += a.a1[j] * (qxb[P + j] - qxb[P - j]) * a.ih[0];
                tA += a.a1[j] * (yw[2 * jj + 1] * yga[2 * jj + 1] - yw[2 * jj] * yga[2 * jj]) * a.ih[1];
                tB += a.a1[j] * (yw[2 * jj + 1] * ygb[2 * jj + 1] - yw[2 * jj] * ygb[2 * jj]) * a.ih[1];
                sA += tA, sB += tB;
            }
        }
        const tf4 X = lap + sA, Y = -sB;   // Gzz = -s
        // group 4: the pointwise streams of this plane
        tf4 et, dl, mm, ee, uc, uo, vc, vo;
        gld<0>(et, voff, a.et + a.off0 + px);
        gld<0>(dl, voff, a.dl + a.off0 + px);
        gld<0>(mm, voff, a.m + a.off0 + px);
        gld<0>(ee, voff, a.eta + a.off0 + px);
        gld<0>(uc, voff, a.ucur + lv + px);
        gld<0>(uo, voff, a.uold + lv + px);
        gld<0>(vc, voff, a.vcur + lv + px);
        gld<0>(vo, voff, a.vold + lv + px);
        vm_wait<0>();
        arrived(et), arrived(dl), arrived(mm), arrived(ee), arrived(uc), arrived(uo), arrived(vc), arrived(vo);
        const tf4 H0 = a.fwd ? et * X + dl * Y : X;
        const tf4 Hz = a.fwd ? dl * X + Y : Y;
        tf4 un, vn;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float ed = ee[i] * a.dt;
            un[i] = (a.dt2 * H0[i] + 2.0f * mm[i] * uc[i] - mm[i] * uo[i] + ed * uc[i]) / (mm[i] + ed);
            vn[i] = (a.dt2 * Hz[i] + 2.0f * mm[i] * vc[i] - mm[i] * vo[i] + ed * vc[i]) / (mm[i] + ed);
        }
        float *uo_p = a.uout + lv + px + voff / 4, *vo_p = a.vout + lv + px + voff / 4;
        if (z + V - 1 < a.n2) {
            *reinterpret_cast<tf4 *>(uo_p) = un;
            *reinterpret_cast<tf4 *>(vo_p) = vn;
        } else {
            for (int i = 0; i < V; ++i)
                if (z + i < a.n2) uo_p[i] = un[i], vo_p[i] = vn[i];
        }
    }
}

// One march for radius 2 and 4: Gz is not stored at all.  In the iteration of plane x a thread forms Gz of both
// fields at plane x + p for its own column (the x part of Gzz at plane x needs the products up to there; A's queue
// reaches x + R = x + 2p, B gets a queue of 2p + 1 planes), pushes wx Gz into the x queues, and hands wy Gz and wz Gz
// to its neighbours through LDS; 2p (TZ/4) + 2 TY threads of the block also form them for one column each of the halo
// (p rows above and below the tile, one 16-byte piece left and right of every row), with their neighbours along x
// loaded directly.  After the barrier every thread takes the y and z parts of Gzz at plane x + p from LDS into a
// queue of p + 1 planes, and then updates plane x.  A chunk starts 2p planes early (no update there) so that the
// queues are full at its first plane.  11 streams read and 2 written per step, against 19 and 4 with two marches.
template <int R>
__global__ __launch_bounds__(256) void tti3d_fused_kernel(Tti3DArgs a)
{
    constexpr int P = R / 2, TY = 16, TZ = 64, V = 4, TQ = TZ / V;
    constexpr int HY = TY + 2 * P, WZ = TZ + 8;
    constexpr int NHY = 2 * P * TQ, NHZ = 2 * TY, NH = NHY + NHZ;
    static_assert(R == 2 || R == 4, "radius");
    static_assert(NH <= 256, "halo columns");
    __shared__ __attribute__((aligned(16))) float syA[2][HY * TZ], syB[2][HY * TZ], szA[2][TY * WZ], szB[2][TY * WZ];
    int t = xcd_contiguous_block();
    const int tile_z = t % a.ntz;
    t /= a.ntz;
    const int tile_y = t % a.nty;
    t /= a.nty;
    const int chunk = t % a.nchunk;
    const int shot = t / a.nchunk;
    const int xa = chunk * a.xchunk, xb = min(a.n0, xa + a.xchunk);
    if (xa >= xb) return;
    const int tid = threadIdx.x, ty = tid / TQ, tq = tid % TQ;
    const int y0 = tile_y * TY, z0 = tile_z * TZ;
    const int y = y0 + ty, z = z0 + V * tq;
    const bool live = y < a.n1 && z < a.n2;              // points this thread updates
    const bool liveg = y < a.n1 + P && z < a.n2 + P;     // columns whose Gz somebody needs
    // lane offsets in bytes from point (x, -P, -4) of a plane; columns nobody needs read the tile's first column
    const long s1 = a.s1;
    const unsigned voff0 = (unsigned)(((long)(y0 + P) * s1 + z0 + 4) * 4);
    const unsigned voff = liveg ? (unsigned)(((long)(y + P) * s1 + z + 4) * 4) : voff0;
    const long org = a.off0 - (long)P * s1 - 4;
    // the halo column of this thread (tid < NH): rows above / below the tile, then pieces left / right of each row
    const bool hisy = tid < NHY;
    int hy, hz, hl;   // grid row, first grid z, float index in sy (rows) or sz (pieces)
    if (hisy) {
        const int r = tid / TQ;
        hy = r < P ? y0 - P + r : y0 + TY + (r - P);
        hz = z0 + V * tq;
        hl = (r < P ? r : TY + r) * TZ + V * tq;
    } else {
        const int k = tid - NHY, row = k >> 1, side = k & 1;
        hy = y0 + row;
        hz = side ? z0 + TZ : z0 - 4;
        hl = row * WZ + (side ? TZ + 4 : 0);
    }
    const bool hrole = tid < NH;
    const bool hlive = hrole && hy < a.n1 + P && hz < a.n2 + P;
    const unsigned hvoff = hlive ? (unsigned)(((long)(hy + P) * s1 + hz + 4) * 4) : voff0;

    const float *A = a.A + (long)shot * a.a_stride + org, *B = a.B + (long)shot * a.a_stride + org;
    const float *wx = a.wx + org, *wy = a.wy + org, *wz = a.wz + org;
    const long lv = (long)shot * a.lv_stride + org;

    // qa[j]: A at plane x - R + j; qb[j]: B at plane x + j; qxa / qxb[j]: wx Gz at plane x - P + j;
    // pa / pb[j]: y and z parts of the divergence of the products at plane x + j   (all after the shift)
    tf4 qa[2 * R + 1], qb[2 * P + 1], qxa[2 * P + 1], qxb[2 * P + 1], pa[P + 1], pb[P + 1];
    const int x0 = xa - 2 * P;
#pragma unroll
    for (int j = 0; j < 2 * R; ++j) gld<0>(qa[j + 1], voff, A + (long)max(x0 - R + j, -R) * a.s0);
#pragma unroll
    for (int j = 0; j < 2 * P; ++j) gld<0>(qb[j + 1], voff, B + (long)(x0 + j) * a.s0);
    vm_wait<0>();
#pragma unroll
    for (int j = 1; j <= 2 * R; ++j) arrived(qa[j]);
#pragma unroll
    for (int j = 1; j <= 2 * P; ++j) arrived(qb[j]);
#pragma unroll
    for (int j = 0; j <= 2 * P; ++j) qxa[j] = qxb[j] = tf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j <= P; ++j) pa[j] = pb[j] = tf4{0.f, 0.f, 0.f, 0.f};

    for (int x = x0; x < xb; ++x) {
        const int buf = (x - x0) & 1;
        const long pg = (long)(x + P) * a.s0;   // the plane Gz is formed at
        const float *Ag = A + pg, *Bg = B + pg;
        {
            // own column: new planes of the queues, rows and line of both fields at plane x + P, the three weights
            tf4 na, nb, ya[2 * P], yb[2 * P], za[2], zb[2], w0, w1, w2;
            gld<0>(na, voff, Ag + (long)P * a.s0);
            gld<0>(nb, voff, Bg + (long)P * a.s0);
#pragma unroll
            for (int j = 1; j <= P; ++j) {
                gld<0>(ya[2 * j - 2], voff, Ag - (long)j * s1);
                gld<0>(ya[2 * j - 1], voff, Ag + (long)j * s1);
                gld<0>(yb[2 * j - 2], voff, Bg - (long)j * s1);
                gld<0>(yb[2 * j - 1], voff, Bg + (long)j * s1);
            }
            gld<-16>(za[0], voff, Ag);
            gld<16>(za[1], voff, Ag);
            gld<-16>(zb[0], voff, Bg);
            gld<16>(zb[1], voff, Bg);
            gld<0>(w0, voff, wx + pg);
            gld<0>(w1, voff, wy + pg);
            gld<0>(w2, voff, wz + pg);
            vm_wait<0>();
            arrived(na), arrived(nb), arrived(za[0]), arrived(za[1]), arrived(zb[0]), arrived(zb[1]);
            arrived(w0), arrived(w1), arrived(w2);
#pragma unroll
            for (int j = 0; j < 2 * P; ++j) arrived(ya[j]), arrived(yb[j]);
#pragma unroll
            for (int j = 0; j < 2 * R; ++j) qa[j] = qa[j + 1];
            qa[2 * R] = na;
#pragma unroll
            for (int j = 0; j < 2 * P; ++j) qb[j] = qb[j + 1];
            qb[2 * P] = nb;
            tf4 dxa = {0.f, 0.f, 0.f, 0.f}, dxb = dxa, dya = dxa, dyb = dxa, dza = dxa, dzb = dxa;
#pragma unroll
            for (int j = 1; j <= P; ++j) {
                dxa += a.a1[j] * (qa[R + P + j] - qa[R + P - j]);
                dxb += a.a1[j] * (qb[P + j] - qb[P - j]);
                dya += a.a1[j] * (ya[2 * j - 1] - ya[2 * j - 2]);
                dyb += a.a1[j] * (yb[2 * j - 1] - yb[2 * j - 2]);
            }
            float la[V + 8], lb[V + 8];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                la[i] = za[0][i], la[4 + i] = qa[R + P][i], la[8 + i] = za[1][i];
                lb[i] = zb[0][i], lb[4 + i] = qb[P][i], lb[8 + i] = zb[1][i];
            }
#pragma unroll
            for (int j = 1; j <= P; ++j)
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    dza[i] += a.a1[j] * (la[4 + i + j] - la[4 + i - j]);
                    dzb[i] += a.a1[j] * (lb[4 + i + j] - lb[4 + i - j]);
                }
            const tf4 ga = -(w0 * (dxa * a.ih[0]) + w1 * (dya * a.ih[1]) + w2 * (dza * a.ih[2]));
            const tf4 gb = -(w0 * (dxb * a.ih[0]) + w1 * (dyb * a.ih[1]) + w2 * (dzb * a.ih[2]));
#pragma unroll
            for (int j = 0; j < 2 * P; ++j) qxa[j] = qxa[j + 1], qxb[j] = qxb[j + 1];
            qxa[2 * P] = w0 * ga, qxb[2 * P] = w0 * gb;
            *reinterpret_cast<tf4 *>(&syA[buf][(ty + P) * TZ + V * tq]) = w1 * ga;
            *reinterpret_cast<tf4 *>(&syB[buf][(ty + P) * TZ + V * tq]) = w1 * gb;
            *reinterpret_cast<tf4 *>(&szA[buf][ty * WZ + 4 + V * tq]) = w2 * ga;
            *reinterpret_cast<tf4 *>(&szB[buf][ty * WZ + 4 + V * tq]) = w2 * gb;
        }
        if (hrole) {
            // halo column: nothing of it is in registers
            tf4 xa4[2 * P], xb4[2 * P], ya[2 * P], yb[2 * P], za[3], zb[3], w0, w1, w2;
#pragma unroll
            for (int j = 1; j <= P; ++j) {
                gld<0>(xa4[2 * j - 2], hvoff, Ag - (long)j * a.s0);
                gld<0>(xa4[2 * j - 1], hvoff, Ag + (long)j * a.s0);
                gld<0>(xb4[2 * j - 2], hvoff, Bg - (long)j * a.s0);
                gld<0>(xb4[2 * j - 1], hvoff, Bg + (long)j * a.s0);
            }
#pragma unroll
            for (int j = 1; j <= P; ++j) {
                gld<0>(ya[2 * j - 2], hvoff, Ag - (long)j * s1);
                gld<0>(ya[2 * j - 1], hvoff, Ag + (long)j * s1);
                gld<0>(yb[2 * j - 2], hvoff, Bg - (long)j * s1);
                gld<0>(yb[2 * j - 1], hvoff, Bg + (long)j * s1);
            }
            gld<-16>(za[0], hvoff, Ag);
            gld<0>(za[1], hvoff, Ag);
            gld<16>(za[2], hvoff, Ag);
            gld<-16>(zb[0], hvoff, Bg);
            gld<0>(zb[1], hvoff, Bg);
            gld<16>(zb[2], hvoff, Bg);
            gld<0>(w0, hvoff, wx + pg);
            gld<0>(w1, hvoff, wy + pg);
            gld<0>(w2, hvoff, wz + pg);
            vm_wait<0>();
#pragma unroll
            for (int j = 0; j < 2 * P; ++j) arrived(xa4[j]), arrived(xb4[j]), arrived(ya[j]), arrived(yb[j]);
#pragma unroll
            for (int c = 0; c < 3; ++c) arrived(za[c]), arrived(zb[c]);
            arrived(w0), arrived(w1), arrived(w2);
            tf4 dxa = {0.f, 0.f, 0.f, 0.f}, dxb = dxa, dya = dxa, dyb = dxa, dza = dxa, dzb = dxa;
#pragma unroll
            for (int j = 1; j <= P; ++j) {
                dxa += a.a1[j] * (xa4[2 * j - 1] - xa4[2 * j - 2]);
                dxb += a.a1[j] * (xb4[2 * j - 1] - xb4[2 * j - 2]);
                dya += a.a1[j] * (ya[2 * j - 1] - ya[2 * j - 2]);
                dyb += a.a1[j] * (yb[2 * j - 1] - yb[2 * j - 2]);
            }
            float la[V + 8], lb[V + 8];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < V; ++i) la[4 * c + i] = za[c][i], lb[4 * c + i] = zb[c][i];
#pragma unroll
            for (int j = 1; j <= P; ++j)
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    dza[i] += a.a1[j] * (la[4 + i + j] - la[4 + i - j]);
                    dzb[i] += a.a1[j] * (lb[4 + i + j] - lb[4 + i - j]);
                }
            const tf4 ga = -(w0 * (dxa * a.ih[0]) + w1 * (dya * a.ih[1]) + w2 * (dza * a.ih[2]));
            const tf4 gb = -(w0 * (dxb * a.ih[0]) + w1 * (dyb * a.ih[1]) + w2 * (dzb * a.ih[2]));
            const tf4 w = hisy ? w1 : w2;
            float *da = hisy ? &syA[buf][hl] : &szA[buf][hl], *db = hisy ? &syB[buf][hl] : &szB[buf][hl];
            *reinterpret_cast<tf4 *>(da) = w * ga;
            *reinterpret_cast<tf4 *>(db) = w * gb;
        }
        __syncthreads();
        {
            // y and z parts of the divergence of the products at plane x + P
            float la[V + 8], lb[V + 8];
            const tf4 l0 = *reinterpret_cast<const tf4 *>(&szA[buf][ty * WZ + V * tq]);
            const tf4 l1 = *reinterpret_cast<const tf4 *>(&szA[buf][ty * WZ + V * tq + 4]);
            const tf4 l2 = *reinterpret_cast<const tf4 *>(&szA[buf][ty * WZ + V * tq + 8]);
            const tf4 m0 = *reinterpret_cast<const tf4 *>(&szB[buf][ty * WZ + V * tq]);
            const tf4 m1 = *reinterpret_cast<const tf4 *>(&szB[buf][ty * WZ + V * tq + 4]);
            const tf4 m2 = *reinterpret_cast<const tf4 *>(&szB[buf][ty * WZ + V * tq + 8]);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                la[i] = l0[i], la[4 + i] = l1[i], la[8 + i] = l2[i];
                lb[i] = m0[i], lb[4 + i] = m1[i], lb[8 + i] = m2[i];
            }
            tf4 sa = {0.f, 0.f, 0.f, 0.f}, sb = sa;
#pragma unroll
            for (int j = 1; j <= P; ++j) {
                tf4 ta, tb;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    ta[i] = a.a1[j] * (la[4 + i + j] - la[4 + i - j]) * a.ih[2];
                    tb[i] = a.a1[j] * (lb[4 + i + j] - lb[4 + i - j]) * a.ih[2];
                }
                const tf4 yap = *reinterpret_cast<const tf4 *>(&syA[buf][(ty + P + j) * TZ + V * tq]);
                const tf4 yam = *reinterpret_cast<const tf4 *>(&syA[buf][(ty + P - j) * TZ + V * tq]);
                const tf4 ybp = *reinterpret_cast<const tf4 *>(&syB[buf][(ty + P + j) * TZ + V * tq]);
                const tf4 ybm = *reinterpret_cast<const tf4 *>(&syB[buf][(ty + P - j) * TZ + V * tq]);
                ta += a.a1[j] * (yap - yam) * a.ih[1];
                tb += a.a1[j] * (ybp - ybm) * a.ih[1];
                sa += ta, sb += tb;
            }
#pragma unroll
            for (int j = 0; j < P; ++j) pa[j] = pa[j + 1], pb[j] = pb[j + 1];
            pa[P] = sa, pb[P] = sb;
        }
        if (x < xa) continue;   // planes before the chunk only fill the queues
        const long px = (long)x * a.s0;
        const float *Ap = A + px;
        tf4 lap;
        tf4 et, dl, mm, ee, uc, uo, vc, vo;
        {
            tf4 yA[2 * R], zA[2];
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                gld<0>(yA[2 * k - 2], voff, Ap - (long)k * s1);
                gld<0>(yA[2 * k - 1], voff, Ap + (long)k * s1);
            }
            gld<-16>(zA[0], voff, Ap);
            gld<16>(zA[1], voff, Ap);
            gld_nt<0>(et, voff, a.et + org + px);
            gld_nt<0>(dl, voff, a.dl + org + px);
            gld_nt<0>(mm, voff, a.m + org + px);
            gld_nt<0>(ee, voff, a.eta + org + px);
            gld<0>(uc, voff, a.ucur + lv + px);
            gld_nt<0>(uo, voff, a.uold + lv + px);
            gld<0>(vc, voff, a.vcur + lv + px);
            gld_nt<0>(vo, voff, a.vold + lv + px);
            vm_wait<8>();
#pragma unroll
            for (int k = 0; k < 2 * R; ++k) arrived(yA[k]);
            arrived(zA[0]), arrived(zA[1]);
            const tf4 ac = qa[R];
            lap = a.c0 * ac;
            float line[V + 8];
#pragma unroll
            for (int i = 0; i < V; ++i) line[i] = zA[0][i], line[4 + i] = ac[i], line[8 + i] = zA[1][i];
#pragma unroll
            for (int k = 1; k <= R; ++k)
#pragma unroll
                for (int i = 0; i < V; ++i) lap[i] += a.cz[k] * (line[4 + i - k] + line[4 + i + k]);
#pragma unroll
            for (int k = 1; k <= R; ++k) lap += a.cy[k] * (yA[2 * k - 2] + yA[2 * k - 1]);
#pragma unroll
            for (int k = 1; k <= R; ++k) lap += a.cx[k] * (qa[R - k] + qa[R + k]);
        }
        tf4 sA = pa[0], sB = pb[0];
#pragma unroll
        for (int j = 1; j <= P; ++j) {
            sA += a.a1[j] * (qxa[P + j] - qxa[P - j]) * a.ih[0];
            sB += a.a1[j] * (qxb[P + j] - qxb[P - j]) * a.ih[0];
        }
        const tf4 X = lap + sA, Y = -sB;   // Gzz = -s
        vm_wait<0>();
        arrived(et), arrived(dl), arrived(mm), arrived(ee), arrived(uc), arrived(uo), arrived(vc), arrived(vo);
        const tf4 H0 = a.fwd ? et * X + dl * Y : X;
        const tf4 Hz = a.fwd ? dl * X + Y : Y;
        tf4 un, vn;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float ed = ee[i] * a.dt;
            un[i] = (a.dt2 * H0[i] + 2.0f * mm[i] * uc[i] - mm[i] * uo[i] + ed * uc[i]) / (mm[i] + ed);
            vn[i] = (a.dt2 * Hz[i] + 2.0f * mm[i] * vc[i] - mm[i] * vo[i] + ed * vc[i]) / (mm[i] + ed);
        }
        if (live) {
            float *uo_p = a.uout + lv + px + voff / 4, *vo_p = a.vout + lv + px + voff / 4;
            if (z + V - 1 < a.n2) {
                __builtin_nontemporal_store(un, reinterpret_cast<tf4 *>(uo_p));
                __builtin_nontemporal_store(vn, reinterpret_cast<tf4 *>(vo_p));
            } else {
                for (int i = 0; i < V; ++i)
                    if (z + i < a.n2) uo_p[i] = un[i], vo_p[i] = vn[i];
            }
        }
    }
}

// float32 3-D plans that could allocate the tables of the marches (p.tti_gz3, p.tti_w) take them.  (At radius 6
// and 8 the batches of tti3d_step_kernel and the plane queues do not fit in 256 registers together; the compiler then
// moves registers whose loads are still in flight and the results are wrong -- hence tti3d_step_wide_kernel.)
static bool tti3d_available(const Plan &p)
{
    const bool one_march = p.L.R <= 4 && !p.tti_two_pass;   // needs no stored Gz
    return p.L.ndim == 3 && p.esize == 4 && p.tti_w && (one_march || p.tti_gz3) && !(p.cfg.flags & MIFWI_FLAG_GENERIC) &&
           p.L.R % 2 == 0 && p.L.R <= 8;
}

// wx, wy of the parameter fields of the plan (order in p.tti_par: et, dl, ct, st, cp, sp)
int launch_tti3d_weights(const Plan &p)
{
    float *w = (float *)p.tti_w;
    const float *par = (const float *)p.tti_par;
    const long f = p.L.field;
    hipLaunchKernelGGL(tti3d_weights_kernel, dim3(2048), dim3(256), 0, p.stream, par + 3 * f, par + 4 * f, par + 5 * f, w, w + f, f);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

// chunks of planes per tile: about `rounds` blocks per CU, chunks no shorter than twice the planes a chunk primes
static void tti3d_chunks(const Plan &p, int tiles, int planes, int prime, int rounds, int &nchunk, int &xchunk)
{
    if (const char *env = getenv("MIFWI_TTI3D_ROUNDS")) rounds = std::max(1, atoi(env));   // tuning experiments
    int n = (rounds * p.ncu + tiles - 1) / tiles;
    n = std::max(1, std::min(n, planes / (2 * prime)));
    xchunk = (planes + n - 1) / n;
    nchunk = (planes + xchunk - 1) / xchunk;
}

static int launch_tti3d(const Plan &p, const StepConsts<float> &k, const TtiParams<float> &q, int fwd, const float *A,
                        const float *B, long ab_stride, const TtiLevels<float> &lv, int nshots, int &launches)
{
    float *gz = (float *)p.tti_gz3;
    const float *w = (const float *)p.tti_w;
    const long gstride = (long)p.cfg.max_shots * k.field;
    const bool fused = p.L.R <= 4 && !p.tti_two_pass;
    if (!fused) {
        Tti3DGzArgs g{};
        g.A = A, g.B = B, g.ab_stride = ab_stride, g.gz = gz, g.gz_stride = gstride;
        g.wx = w, g.wy = w + k.field, g.wz = q.ct;
        for (int j = 1; j <= q.p; ++j) g.a[j] = q.a[j];
        g.ih[0] = q.ih[0], g.ih[1] = q.ih[1], g.ih[2] = q.ih[2];
        g.n0 = k.n0, g.n1 = k.n1, g.n2 = k.n2, g.e = q.p, g.zl = p.L.zl;
        g.s0 = k.s0, g.s1 = k.s1, g.off0 = k.off0, g.field = k.field;
        g.nty = (k.n1 + 2 * q.p + TTI3D_TY - 1) / TTI3D_TY, g.ntz = (p.L.zl + k.n2 + q.p + 63) / 64;
        tti3d_chunks(p, g.nty * g.ntz * nshots, k.n0 + 2 * q.p, 2 * q.p, 8, g.nchunk, g.xchunk);
        g.nshots = nshots;
        const int nb = g.nty * g.ntz * g.nchunk * nshots;
        switch (q.p) {
            case 1: hipLaunchKernelGGL((tti3d_gz_kernel<1>), dim3(nb), dim3(16 * TTI3D_TY), 0, p.stream, g); break;
            case 2: hipLaunchKernelGGL((tti3d_gz_kernel<2>), dim3(nb), dim3(16 * TTI3D_TY), 0, p.stream, g); break;
            case 3: hipLaunchKernelGGL((tti3d_gz_kernel<3>), dim3(nb), dim3(16 * TTI3D_TY), 0, p.stream, g); break;
            default: hipLaunchKernelGGL((tti3d_gz_kernel<4>), dim3(nb), dim3(16 * TTI3D_TY), 0, p.stream, g);
        }
        MIFWI_HIP(hipGetLastError());
    }
    Tti3DArgs a{};
    a.A = A, a.B = B, a.a_stride = ab_stride, a.gz = gz, a.gz_stride = gstride;
    a.wx = w, a.wy = w + k.field, a.wz = q.ct;
    a.et = q.et, a.dl = q.dl, a.m = (const float *)p.m, a.eta = (const float *)p.eta;
    a.ucur = lv.ucur, a.uold = lv.uold, a.vcur = lv.vcur, a.vold = lv.vold, a.uout = lv.uout, a.vout = lv.vout;
    a.lv_stride = lv.shot_stride;
    a.c0 = k.c0;
    for (int j = 0; j <= p.L.R; ++j) a.cx[j] = k.c[0][j], a.cy[j] = k.c[1][j], a.cz[j] = k.c[2][j];
    for (int j = 1; j <= q.p; ++j) a.a1[j] = q.a[j];
    a.ih[0] = q.ih[0], a.ih[1] = q.ih[1], a.ih[2] = q.ih[2];
    a.dt = k.dt, a.dt2 = k.dt2;
    a.n0 = k.n0, a.n1 = k.n1, a.n2 = k.n2, a.s0 = k.s0, a.s1 = k.s1, a.off0 = k.off0, a.field = k.field;
    a.nty = (k.n1 + TTI3D_TY - 1) / TTI3D_TY, a.ntz = (k.n2 + 63) / 64;
    tti3d_chunks(p, a.nty * a.ntz * nshots, k.n0, 2 * p.L.R, 8, a.nchunk, a.xchunk);
    a.nshots = nshots, a.fwd = fwd;
    const int nb = a.nty * a.ntz * a.nchunk * nshots;
    if (fused) {
        if (p.L.R == 2)
            hipLaunchKernelGGL((tti3d_fused_kernel<2>), dim3(nb), dim3(256), 0, p.stream, a);
        else
            hipLaunchKernelGGL((tti3d_fused_kernel<4>), dim3(nb), dim3(256), 0, p.stream, a);
        MIFWI_HIP(hipGetLastError());
        launches += 1;
        return MIFWI_OK;
    }
    switch (p.L.R) {
        case 2: hipLaunchKernelGGL((tti3d_step_kernel<2>), dim3(nb), dim3(16 * TTI3D_TY), 0, p.stream, a); break;
        case 4: hipLaunchKernelGGL((tti3d_step_kernel<4>), dim3(nb), dim3(16 * TTI3D_TY), 0, p.stream, a); break;
        case 6: hipLaunchKernelGGL((tti3d_step_wide_kernel<6>), dim3(nb), dim3(16 * TTI3D_TY), 0, p.stream, a); break;
        default: hipLaunchKernelGGL((tti3d_step_wide_kernel<8>), dim3(nb), dim3(16 * TTI3D_TY), 0, p.stream, a);
    }
    MIFWI_HIP(hipGetLastError());
    launches += 2;
    return MIFWI_OK;
}

// One TTI time step without the sparse terms.  fwd: A, B = current u, v.  adjoint: A, B are formed
// into wA, wB first.  Launch count is added to `launches`.
template <typename T>
int launch_tti_step(const Plan &p, const StepConsts<T> &k, const TtiParams<T> &q, int fwd, const TtiLevels<T> &lv,
                    T *wA, T *wB, T *gzA, T *gzB, int nshots, int &launches, const Step2DArgs *sparse2d,
                    const TtiExtra<T> *extra)
{
    if constexpr (std::is_same<T, float>::value) {
        if (sparse2d) {   // tiled 2-D path: one launch for both fields, the sparse terms and the extra term
            launches += 1;
            return launch_tti2d(p, k, q, fwd, lv, nshots, *sparse2d, extra);
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
    if constexpr (std::is_same<T, float>::value) {
        if (tti3d_available(p)) return launch_tti3d(p, k, q, fwd, A, B, ab_stride, lv, nshots, launches);
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
// ------------------------------------------------------------------------------------------------
// Zero-lag correlation of the TTI gradient (seismic/tti/operators.py:631), one thread per point.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void tti_grad_kernel(StepConsts<T> k, const T *su, const T *sv, long level_stride,
                                                       long s_shot_stride, const T *pc, const T *rc,
                                                       long a_shot_stride, T *grad, long g_shot_stride)
{
    const int i2 = blockIdx.x * 64 + threadIdx.x;
    const int i1 = blockIdx.y * 4 + threadIdx.y;
    const int shot = blockIdx.z / k.n0;
    const int i0 = blockIdx.z - shot * k.n0;
    if (i2 >= k.n2 || i1 >= k.n1) return;
    const long i = k.off0 + (long)i0 * k.s0 + (long)i1 * k.s1 + i2;
    const T *u = su + (long)shot * s_shot_stride + i, *v = sv + (long)shot * s_shot_stride + i;
    const T utt = (u[level_stride] - T(2) * u[0] + u[-level_stride]) * k.inv_dt2;
    const T vtt = (v[level_stride] - T(2) * v[0] + v[-level_stride]) * k.inv_dt2;
    const long a = (long)shot * a_shot_stride + i;
    grad[(long)shot * g_shot_stride + i] += -(utt * pc[a] + vtt * rc[a]);
}

template <typename T>
int launch_tti_grad(const Plan &p, const StepConsts<T> &k, const T *su, const T *sv, long level_stride, long s_shot_stride,
                    const T *pc, const T *rc, long a_shot_stride, T *grad, long g_shot_stride, int nshots)
{
    if ((long)nshots * k.n0 > 65535) {
        set_error("TTI gradient kernel: nshots * shape[0] exceeds the grid z limit");
        return MIFWI_EINVAL;
    }
    dim3 block(64, 4, 1);
    dim3 grid((k.n2 + 63) / 64, (k.n1 + 3) / 4, (unsigned)(nshots * k.n0));
    hipLaunchKernelGGL((tti_grad_kernel<T>), grid, block, 0, p.stream, k, su, sv, level_stride, s_shot_stride, pc, rc,
                       a_shot_stride, grad, g_shot_stride);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template int launch_tti_grad<float>(const Plan &, const StepConsts<float> &, const float *, const float *, long, long,
                                    const float *, const float *, long, float *, long, int);
template int launch_tti_grad<double>(const Plan &, const StepConsts<double> &, const double *, const double *, long, long,
                                     const double *, const double *, long, double *, long, int);

template int launch_tti_step<float>(const Plan &, const StepConsts<float> &, const TtiParams<float> &, int,
                                    const TtiLevels<float> &, float *, float *, float *, float *, int, int &, const Step2DArgs *,
                                    const TtiExtra<float> *);
template int launch_tti_step<double>(const Plan &, const StepConsts<double> &, const TtiParams<double> &, int,
                                     const TtiLevels<double> &, double *, double *, double *, double *, int, int &, const Step2DArgs *,
                                     const TtiExtra<double> *);

}  // namespace mifwi
