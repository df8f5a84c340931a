// Untiled reference kernels of libmifwi: one thread per grid point, neighbours read through the
// vector L1/L2.  They cover every case the ABI accepts (1-3 dims, space orders 2..16, f32 and
// f64) and are the fallback of the tiled fast paths.  Also here: the sparse source/receiver
// kernel, model preparation, halo pack/unpack and the L2-misfit reduction.
//
// Discrete equations (seismic/acoustic/operators.py:59-95 of the reference, with the one-sided
// damping term that its known-answer tests pin):
//     out = ( dt^2 L(cur) + 2 m cur - m old + eta dt cur ) / ( m + eta dt )
//     L(u) = c0 u + sum_d sum_{k=1..R} c[d][k] (u[-k e_d] + u[+k e_d])
// The same update serves forward (old = u[t-1], out = u[t+1]) and backward propagation
// (old = v[t+1], out = v[t-1]).  The gradient variant additionally accumulates
//     grad += -(u[t-1] - 2 u[t] + u[t+1]) / dt^2 * v[t]      (operators.py:216-217)
// and, optionally, illum += u[t]^2 (fwi.py:170).
#include "mifwi_internal.hpp"

namespace mifwi {

// plain star stencil L(f) at the point f points to
template <typename T, int R, int ND>
__device__ inline T star(const StepConsts<T> &k, const T *f)
{
    T lap = k.c0 * f[0];
#pragma unroll
    for (int j = 1; j <= R; ++j) lap += k.c[2][j] * (f[-j] + f[j]);
    if (ND >= 2) {
#pragma unroll
        for (int j = 1; j <= R; ++j) lap += k.c[1][j] * (f[-j * k.s1] + f[j * k.s1]);
    }
    if (ND >= 3) {
#pragma unroll
        for (int j = 1; j <= R; ++j) lap += k.c[0][j] * (f[-j * k.s0] + f[j * k.s0]);
    }
    return lap;
}

// The same at a point of row i2 (last dimension) below a free surface (operators.py:8-35): in the rows
// z < space_order a term f[z-j] with a negative offset becomes sign(z-j) * f[|z-j|].
template <typename T, int R, int ND>
__device__ inline T star_fs(const StepConsts<T> &k, const T *f, int i2)
{
    if (i2 >= 2 * R) return star<T, R, ND>(k, f);
    T lap = k.c0 * f[0];
#pragma unroll
    for (int j = 1; j <= R; ++j) {
        const int zz = i2 - j;
        const T lo = zz > 0 ? f[-j] : (zz < 0 ? -f[-i2 - zz] : T(0));
        lap += k.c[2][j] * (lo + f[j]);
    }
    if (ND >= 2) {
#pragma unroll
        for (int j = 1; j <= R; ++j) lap += k.c[1][j] * (f[-j * k.s1] + f[j * k.s1]);
    }
    if (ND >= 3) {
#pragma unroll
        for (int j = 1; j <= R; ++j) lap += k.c[0][j] * (f[-j * k.s0] + f[j * k.s0]);
    }
    return lap;
}

template <typename T, int R, int ND>
__global__ __launch_bounds__(256) void lap_scale_kernel(StepConsts<T> k, const T *in, long in_shot_stride,
                                                        const T *m, int power, T *out, long out_shot_stride)
{
    const int i2 = blockIdx.x * 64 + threadIdx.x;
    const int i1 = blockIdx.y * 4 + threadIdx.y;
    const int shot = blockIdx.z / k.n0;
    const int i0 = blockIdx.z - shot * k.n0;
    if (i2 >= k.n2 || i1 >= k.n1) return;
    const long p = k.off0 + (long)i0 * k.s0 + (long)i1 * k.s1 + i2;
    // power 1 is the inner half of the OT4 stencil term, which the free-surface rewrite of the update reaches into
    // (every field read through a negative offset is mirrored); power 2 belongs to the gradient's correlation term,
    // which the reference leaves as it is (operators.py:219 is not passed through freesurface())
    const T *f = in + (long)shot * in_shot_stride + p;
    const T lap = (k.fs && power == 1) ? star_fs<T, R, ND>(k, f, i2) : star<T, R, ND>(k, f);
    const T mz = m[p];
    out[(long)shot * out_shot_stride + p] = power == 1 ? lap / mz : lap / (mz * mz);
}

template <typename T, int R, int ND, bool GRAD>
__global__ __launch_bounds__(256) void step_generic_kernel(StepConsts<T> k, StepPtrs<T> sp, GradPtrs<T> gp)
{
    const int i2 = blockIdx.x * 64 + threadIdx.x;
    const int i1 = blockIdx.y * 4 + threadIdx.y;
    const int shot = blockIdx.z / k.n0;
    const int i0 = blockIdx.z - shot * k.n0;
    if (i2 >= k.n2 || i1 >= k.n1) return;
    const long p = k.off0 + (long)i0 * k.s0 + (long)i1 * k.s1 + i2;
    const T *cur = sp.cur + (long)shot * sp.shot_stride + p;
    const T uc = cur[0];
    T lap = k.c0 * uc;
    if (k.fs && i2 < 2 * R) {
        // free surface (operators.py:8-35): in the rows z < space_order a term u[z-j] with a negative
        // offset becomes sign(z-j) * u[|z-j|]
#pragma unroll
        for (int j = 1; j <= R; ++j) {
            const int zz = i2 - j;
            const T lo = zz > 0 ? cur[-j] : (zz < 0 ? -cur[-i2 - zz] : T(0));
            lap += k.c[2][j] * (lo + cur[j]);
        }
    } else {
#pragma unroll
        for (int j = 1; j <= R; ++j) lap += k.c[2][j] * (cur[-j] + cur[j]);
    }
    if (ND >= 2) {
#pragma unroll
        for (int j = 1; j <= R; ++j) lap += k.c[1][j] * (cur[-j * k.s1] + cur[j * k.s1]);
    }
    if (ND >= 3) {
#pragma unroll
        for (int j = 1; j <= R; ++j) lap += k.c[0][j] * (cur[-j * k.s0] + cur[j * k.s0]);
    }
    if (sp.ot4) {
        // laplace(u)/m read through a negative offset is mirrored like u itself: 1/m = vp^2 keeps its sign (the two
        // signs of vp cancel) and the inner Laplacian of the antisymmetric field changes it
        const T *w = sp.ot4 + (long)shot * k.field + p;
        lap += k.dt2 / T(12) * (k.fs ? star_fs<T, R, ND>(k, w, i2) : star<T, R, ND>(k, w));
    }
    const T mz = sp.m[p];
    const T ed = sp.eta[p] * k.dt;
    const T uo = sp.old[(long)shot * sp.shot_stride + p];
    sp.out[(long)shot * sp.shot_stride + p] = (k.dt2 * lap + T(2) * mz * uc - mz * uo + ed * uc) / (mz + ed);
    if (GRAD) {
        const long q = (long)shot * gp.u_shot_stride + p;
        const T um = gp.um[q], u0 = gp.uc[q], up = gp.up[q];
        const long g = (long)shot * gp.g_shot_stride + p;
        T utt = (um - T(2) * u0 + up) * k.inv_dt2;
        if (gp.bih) utt += k.dt2 / T(12) * star<T, R, ND>(k, gp.bih + (long)shot * k.field + p);
        gp.grad[g] += -utt * uc;
        if (gp.illum) {
            T add = u0 * u0;
            if (gp.first) add += up * up;
            gp.illum[g] += add;
        }
    }
}

template <typename T, int R, int ND>
static int launch_step_rnd(const Plan &p, const StepConsts<T> &k, const StepPtrs<T> &sp, const GradPtrs<T> &gp,
                           int nshots)
{
    dim3 block(64, 4, 1);
    dim3 grid((k.n2 + 63) / 64, (k.n1 + 3) / 4, (unsigned)(nshots * k.n0));
    if (gp.um)
        hipLaunchKernelGGL((step_generic_kernel<T, R, ND, true>), grid, block, 0, p.stream, k, sp, gp);
    else
        hipLaunchKernelGGL((step_generic_kernel<T, R, ND, false>), grid, block, 0, p.stream, k, sp, gp);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

template <typename T, int R>
static int launch_lap_r(const Plan &p, const StepConsts<T> &k, const T *in, long is, int power, T *out, long os,
                        int nshots)
{
    dim3 block(64, 4, 1);
    dim3 grid((k.n2 + 63) / 64, (k.n1 + 3) / 4, (unsigned)(nshots * k.n0));
    const T *m = (const T *)p.m;
    switch (p.L.ndim) {
        case 1: hipLaunchKernelGGL((lap_scale_kernel<T, R, 1>), grid, block, 0, p.stream, k, in, is, m, power, out, os); break;
        case 2: hipLaunchKernelGGL((lap_scale_kernel<T, R, 2>), grid, block, 0, p.stream, k, in, is, m, power, out, os); break;
        default: hipLaunchKernelGGL((lap_scale_kernel<T, R, 3>), grid, block, 0, p.stream, k, in, is, m, power, out, os);
    }
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

template <typename T>
int launch_lap_scale(const Plan &p, const StepConsts<T> &k, const T *in, long in_shot_stride, int power, T *out,
                     long out_shot_stride, int nshots)
{
    if ((long)nshots * k.n0 > 65535) {
        set_error("generic kernel: nshots * shape[0] exceeds the grid z limit");
        return MIFWI_EINVAL;
    }
    switch (p.L.R) {
        case 1: return launch_lap_r<T, 1>(p, k, in, in_shot_stride, power, out, out_shot_stride, nshots);
        case 2: return launch_lap_r<T, 2>(p, k, in, in_shot_stride, power, out, out_shot_stride, nshots);
        case 3: return launch_lap_r<T, 3>(p, k, in, in_shot_stride, power, out, out_shot_stride, nshots);
        case 4: return launch_lap_r<T, 4>(p, k, in, in_shot_stride, power, out, out_shot_stride, nshots);
        case 5: return launch_lap_r<T, 5>(p, k, in, in_shot_stride, power, out, out_shot_stride, nshots);
        case 6: return launch_lap_r<T, 6>(p, k, in, in_shot_stride, power, out, out_shot_stride, nshots);
        case 7: return launch_lap_r<T, 7>(p, k, in, in_shot_stride, power, out, out_shot_stride, nshots);
        case 8: return launch_lap_r<T, 8>(p, k, in, in_shot_stride, power, out, out_shot_stride, nshots);
        case 9: return launch_lap_r<T, 9>(p, k, in, in_shot_stride, power, out, out_shot_stride, nshots);
        case 10: return launch_lap_r<T, 10>(p, k, in, in_shot_stride, power, out, out_shot_stride, nshots);
    }
    set_error("unsupported stencil radius");
    return MIFWI_EINVAL;
}
template int launch_lap_scale<float>(const Plan &, const StepConsts<float> &, const float *, long, int, float *, long,
                                     int);
template int launch_lap_scale<double>(const Plan &, const StepConsts<double> &, const double *, long, int, double *,
                                      long, int);

template <typename T, int R>
static int launch_step_r(const Plan &p, const StepConsts<T> &k, const StepPtrs<T> &sp, const GradPtrs<T> &gp,
                         int nshots)
{
    switch (p.L.ndim) {
        case 1: return launch_step_rnd<T, R, 1>(p, k, sp, gp, nshots);
        case 2: return launch_step_rnd<T, R, 2>(p, k, sp, gp, nshots);
        default: return launch_step_rnd<T, R, 3>(p, k, sp, gp, nshots);
    }
}

template <typename T>
int launch_step_generic(const Plan &p, const StepConsts<T> &k, const StepPtrs<T> &sp, const GradPtrs<T> &gp,
                        int nshots)
{
    if ((long)nshots * k.n0 > 65535) {
        set_error("generic kernel: nshots * shape[0] exceeds the grid z limit");
        return MIFWI_EINVAL;
    }
    switch (p.L.R) {
        case 1: return launch_step_r<T, 1>(p, k, sp, gp, nshots);
        case 2: return launch_step_r<T, 2>(p, k, sp, gp, nshots);
        case 3: return launch_step_r<T, 3>(p, k, sp, gp, nshots);
        case 4: return launch_step_r<T, 4>(p, k, sp, gp, nshots);
        case 5: return launch_step_r<T, 5>(p, k, sp, gp, nshots);
        case 6: return launch_step_r<T, 6>(p, k, sp, gp, nshots);
        case 7: return launch_step_r<T, 7>(p, k, sp, gp, nshots);
        case 8: return launch_step_r<T, 8>(p, k, sp, gp, nshots);
        case 9: return launch_step_r<T, 9>(p, k, sp, gp, nshots);
        case 10: return launch_step_r<T, 10>(p, k, sp, gp, nshots);
    }
    set_error("unsupported stencil radius");
    return MIFWI_EINVAL;
}
template int launch_step_generic<float>(const Plan &, const StepConsts<float> &, const StepPtrs<float> &,
                                        const GradPtrs<float> &, int);
template int launch_step_generic<double>(const Plan &, const StepConsts<double> &, const StepPtrs<double> &,
                                         const GradPtrs<double> &, int);

// ------------------------------------------------------------------------------------------
// Born source term (operators.py:263), one thread per point.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void born_source_kernel(StepConsts<T> k, const T *u_new, const T *u_cur,
                                                          const T *u_old, long shot_stride, const T *dm,
                                                          const T *m, const T *eta, T *U_out, long U_shot_stride,
                                                          int R, int ghost)
{
    const int i2 = blockIdx.x * 64 + threadIdx.x;
    const int i1 = blockIdx.y * 4 + threadIdx.y;
    const int shot = blockIdx.z / k.n0;
    const int i0 = blockIdx.z - shot * k.n0;
    if (i2 >= k.n2 || i1 >= k.n1) return;
    const long p = k.off0 + (long)i0 * k.s0 + (long)i1 * k.s1 + i2;
    const long q = (long)shot * shot_stride + p;
    const T utt = (u_new[q] - T(2) * u_cur[q] + u_old[q]) * k.inv_dt2;
    const long o = (long)shot * U_shot_stride + p;
    const T val = U_out[o] + k.dt2 * (-dm[p] * utt) / (m[p] + eta[p] * k.dt);
    U_out[o] = val;
    if (ghost && i2 >= 1 && i2 <= R) U_out[o - 2 * i2] = -val;
}

template <typename T>
int launch_born_source(const Plan &p, const StepConsts<T> &k, const T *u_new, const T *u_cur, const T *u_old,
                       long shot_stride, const T *dm, T *U_out, long U_shot_stride, int nshots, int ghost)
{
    if ((long)nshots * k.n0 > 65535) {
        set_error("born source kernel: nshots * shape[0] exceeds the grid z limit");
        return MIFWI_EINVAL;
    }
    dim3 block(64, 4, 1);
    dim3 grid((k.n2 + 63) / 64, (k.n1 + 3) / 4, (unsigned)(nshots * k.n0));
    hipLaunchKernelGGL((born_source_kernel<T>), grid, block, 0, p.stream, k, u_new, u_cur, u_old, shot_stride, dm,
                       (const T *)p.m, (const T *)p.eta, U_out, U_shot_stride, p.L.R, ghost);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template int launch_born_source<float>(const Plan &, const StepConsts<float> &, const float *, const float *,
                                       const float *, long, const float *, float *, long, int, int);
template int launch_born_source<double>(const Plan &, const StepConsts<double> &, const double *, const double *,
                                        const double *, long, const double *, double *, long, int, int);

// ------------------------------------------------------------------------------------------
// Sparse operations of one time step (seismic/acoustic/operators.py:134,137,173,176,221).
// blockIdx.y = shot.  Blocks [0, nb_inj) inject, blocks [nb_inj, nb_inj + nb_itp) interpolate.
// ------------------------------------------------------------------------------------------
template <typename T>
struct SparseDev {
    // injection
    const int32_t *uc_cell, *uc_ptr, *uc_count, *e_point;
    const T *e_w;
    int inj_kmax, inj_npoint, inj_per_shot, nb_inj;
    T *field_inj, *field_inj2;
    const T *vals_inj;
    // interpolation
    const int32_t *pt_idx;
    const T *pt_w;
    int itp_npoint, itp_ncorner, itp_per_shot;
    const T *field_int, *field_int2;
    T *vals_out;
    long shot_stride;
    const T *m;
    int time, nt;
    T dt2;
};

template <typename T>
__global__ __launch_bounds__(256) void sparse_kernel(SparseDev<T> d)
{
    const int shot = blockIdx.y;
    if ((int)blockIdx.x < d.nb_inj) {
        const int tab = d.inj_per_shot ? shot : 0;
        const int kk = blockIdx.x * 256 + threadIdx.x;
        if (kk >= d.uc_count[tab]) return;
        const int32_t *ptr = d.uc_ptr + (long)tab * (d.inj_kmax + 1);
        const int32_t *ep = d.e_point + (long)tab * d.inj_kmax;
        const T *ew = d.e_w + (long)tab * d.inj_kmax;
        const T *vals = d.vals_inj + ((long)shot * d.nt + d.time) * d.inj_npoint;
        T sum = 0;
        for (int e = ptr[kk]; e < ptr[kk + 1]; ++e) sum += ew[e] * vals[ep[e]];
        const long cell = d.uc_cell[(long)tab * d.inj_kmax + kk];
        const T term = sum * d.dt2 / d.m[cell];
        d.field_inj[(long)shot * d.shot_stride + cell] += term;
        if (d.field_inj2) d.field_inj2[(long)shot * d.shot_stride + cell] += term;
    } else {
        const int tab = d.itp_per_shot ? shot : 0;
        const int pnt = (blockIdx.x - d.nb_inj) * 256 + threadIdx.x;
        if (pnt >= d.itp_npoint) return;
        const int32_t *idx = d.pt_idx + ((long)tab * d.itp_npoint + pnt) * d.itp_ncorner;
        const T *w = d.pt_w + ((long)tab * d.itp_npoint + pnt) * d.itp_ncorner;
        const T *f = d.field_int + (long)shot * d.shot_stride;
        T sum = 0;
        if (d.field_int2) {
            const T *f2 = d.field_int2 + (long)shot * d.shot_stride;
            for (int c = 0; c < d.itp_ncorner; ++c) sum += w[c] * (f[idx[c]] + f2[idx[c]]);
        } else {
            for (int c = 0; c < d.itp_ncorner; ++c) sum += w[c] * f[idx[c]];
        }
        d.vals_out[((long)shot * d.nt + d.time) * d.itp_npoint + pnt] = sum;
    }
}

// Injected terms of all time steps (see Plan::d_winj).  blockIdx = (cell chunk, time chunk, shot).
constexpr int kPrepTimes = 16;
__global__ __launch_bounds__(256) void inject_prepare_kernel(const int32_t *g_u, const int32_t *uc_cell,
                                                             const int32_t *uc_ptr, const int32_t *uc_count,
                                                             const int32_t *e_point, const float *e_w, int kmax,
                                                             int npoint, int per_shot, const float *vals, const float *m,
                                                             int nt, float dt2, float *winj)
{
    const int shot = blockIdx.z;
    const int tab = per_shot ? shot : 0;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= uc_count[tab]) return;
    const int u = g_u[(long)tab * kmax + g];
    const int e0 = uc_ptr[(long)tab * (kmax + 1) + u], e1 = uc_ptr[(long)tab * (kmax + 1) + u + 1];
    const float scale = dt2 / m[uc_cell[(long)tab * kmax + u]];
    const int32_t *ep = e_point + (long)tab * kmax;
    const float *ew = e_w + (long)tab * kmax;
    const int t0 = blockIdx.y * kPrepTimes, t1 = min(nt, t0 + kPrepTimes);
    for (int time = t0; time < t1; ++time) {
        const float *v = vals + ((long)shot * nt + time) * npoint;
        float sum = 0.f;
        for (int e = e0; e < e1; ++e) sum += ew[e] * v[ep[e]];
        winj[((long)shot * nt + time) * kmax + g] = sum * scale;
    }
}

int launch_inject_prepare(const Plan &p, const SparseSet &S, const int32_t *g_u, const float *vals, int nshots,
                          float dt2, float *winj)
{
    if (S.npoint <= 0 || S.max_count <= 0) return MIFWI_OK;
    dim3 grid((S.max_count + 255) / 256, (p.cfg.nt + kPrepTimes - 1) / kPrepTimes, nshots);
    hipLaunchKernelGGL(inject_prepare_kernel, grid, dim3(256), 0, p.stream, g_u, S.uc_cell, S.uc_ptr, S.uc_count,
                       S.e_point, (const float *)S.e_w, S.kmax, S.npoint, S.per_shot, vals, (const float *)p.m, p.cfg.nt,
                       dt2, winj);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

template <typename T>
int launch_sparse(const Plan &p, const SparsePtrs<T> &sp, int nshots)
{
    SparseDev<T> d{};
    d.nb_inj = 0;
    int nb_itp = 0;
    if (sp.inj && sp.inj->npoint > 0 && sp.inj->max_count > 0) {
        d.uc_cell = sp.inj->uc_cell;
        d.uc_ptr = sp.inj->uc_ptr;
        d.uc_count = sp.inj->uc_count;
        d.e_point = sp.inj->e_point;
        d.e_w = (const T *)sp.inj->e_w;
        d.inj_kmax = sp.inj->kmax;
        d.inj_npoint = sp.inj->npoint;
        d.inj_per_shot = sp.inj->per_shot;
        d.nb_inj = (sp.inj->max_count + 255) / 256;
        d.field_inj = sp.field_inj;
        d.field_inj2 = sp.field_inj2;
        d.vals_inj = sp.vals_inj;
    }
    if (sp.itp && sp.itp->npoint > 0) {
        d.pt_idx = sp.itp->pt_idx;
        d.pt_w = (const T *)sp.itp->pt_w;
        d.itp_npoint = sp.itp->npoint;
        d.itp_ncorner = sp.itp->ncorner;
        d.itp_per_shot = sp.itp->per_shot;
        d.field_int = sp.field_int;
        d.field_int2 = sp.field_int2;
        d.vals_out = sp.vals_out;
        nb_itp = (sp.itp->npoint + 255) / 256;
    }
    if (d.nb_inj + nb_itp == 0) return MIFWI_OK;
    d.shot_stride = sp.shot_stride;
    d.m = sp.m;
    d.time = sp.time;
    d.nt = sp.nt;
    d.dt2 = sp.dt2;
    hipLaunchKernelGGL((sparse_kernel<T>), dim3(d.nb_inj + nb_itp, nshots), dim3(256), 0, p.stream, d);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template int launch_sparse<float>(const Plan &, const SparsePtrs<float> &, int);
template int launch_sparse<double>(const Plan &, const SparsePtrs<double> &, int);

// ------------------------------------------------------------------------------------------
// Perfectly matched layer, 2-D (Grote and Sim; discretised as seismic/abc_methods/03_pml.ipynb cells 75-76
// of the reference).  u lives on the nodes, phi1 and phi2 on the points half a cell further in x and z
// (array index (x, z) = point (x + 1/2, z + 1/2)).  With s = zeta_x + zeta_z:
//   u[t+1] (1 + s dt/2) = 2 u[t] - u[t-1] (1 - s dt/2)
//                         + dt^2 ( L(u[t]) / m - zeta_x zeta_z u[t] - P1 - P2 )
//   P1 = -(1/2hx) (phi1[x,z-1] + phi1[x,z] - phi1[x-1,z-1] - phi1[x-1,z])
//   P2 = -(1/2hz) (phi2[x-1,z] + phi2[x,z] - phi2[x-1,z-1] - phi2[x,z-1])
//   phi1[t+1] (1/dt + zx/2) = phi1[t] (1/dt - zx/2) + (zz - zx) c^2 (1/4hx) (a(u[t+1]) + a(u[t]))
//   phi2[t+1] (1/dt + zz/2) = phi2[t] (1/dt - zz/2) + (zx - zz) c^2 (1/4hz) (b(u[t+1]) + b(u[t]))
//   a(u) = u[x+1,z] + u[x+1,z+1] - u[x,z] - u[x,z+1],  b(u) = u[x,z+1] + u[x+1,z+1] - u[x,z] - u[x+1,z]
// with zx, zz, c^2 of the phi equations taken at the staggered point (c^2: mean of 1/m over the cell's
// corners inside the grid).  Fields are zero outside the grid (halo), as everywhere in this library.
// ------------------------------------------------------------------------------------------
// node (i1, i2) lies in a layer region: the subdomains d1-d3 of the notebook (cells 20-22), for any choice of sides
template <typename T>
__device__ inline bool pml_in_layer(const PmlPtrs<T> &q, const StepConsts<T> &k, int i1, int i2)
{
    return ((q.sides & 1) && i1 < q.npml) || ((q.sides & 2) && i1 >= k.n1 - q.npml) || ((q.sides & 4) && i2 < q.npml) ||
           ((q.sides & 8) && i2 >= k.n2 - q.npml);
}

template <typename T, int R>
__global__ __launch_bounds__(256) void pml_u_kernel(StepConsts<T> k, PmlPtrs<T> q)
{
    const int i2 = blockIdx.x * 64 + threadIdx.x;
    const int i1 = blockIdx.y * 4 + threadIdx.y;
    const int shot = blockIdx.z;
    if (i2 >= k.n2 || i1 >= k.n1) return;
    const long p = k.off0 + (long)i1 * k.s1 + i2;
    const T *cur = q.cur + (long)shot * q.shot_stride + p;
    const T lap = star<T, R, 2>(k, cur);
    const T uc = cur[0], uo = q.old[(long)shot * q.shot_stride + p];
    const T *f1 = q.phi1 + (long)shot * q.phi_shot_stride + p, *f2 = q.phi2 + (long)shot * q.phi_shot_stride + p;
    const T P1 = -T(0.5) * q.ihx * (f1[-1] + f1[0] - f1[-k.s1 - 1] - f1[-k.s1]);
    const T P2 = -T(0.5) * q.ihz * (f2[-k.s1] + f2[0] - f2[-k.s1 - 1] - f2[-1]);
    const T zx = q.zx0[i1], zz = q.zz0[i2];
    const T h = T(0.5) * (zx + zz) * k.dt;
    const T rhs = T(2) * uc - uo * (T(1) - h) + k.dt2 * (lap / q.m[p] - zx * zz * uc - P1 - P2);
    T un = rhs / (T(1) + h);
    // the notebook's layout: outside the layer regions the plain wave equation, without the auxiliary terms
    if (q.edges && !pml_in_layer(q, k, i1, i2)) un = T(2) * uc - uo + k.dt2 * (lap / q.m[p]);
    q.out[(long)shot * q.shot_stride + p] = un;
}

// Boundary rows of the notebook (03_pml.ipynb cell 83), after the source term: u = 0 on the first and last x node and on
// the last z node, u[x, 0] = u[x, 1].
template <typename T>
__global__ __launch_bounds__(256) void pml_edges_kernel(StepConsts<T> k, T *out, long shot_stride)
{
    const int i2 = blockIdx.x * 64 + threadIdx.x;
    const int i1 = blockIdx.y * 4 + threadIdx.y;
    if (i2 >= k.n2 || i1 >= k.n1) return;
    T *u = out + (long)blockIdx.z * shot_stride + k.off0 + (long)i1 * k.s1;
    if (i1 == 0 || i1 == k.n1 - 1 || i2 == k.n2 - 1) u[i2] = T(0);
    else if (i2 == 0) u[0] = u[1];
}

template <typename T>
__global__ __launch_bounds__(256) void pml_phi_kernel(StepConsts<T> k, PmlPtrs<T> q)
{
    const int i2 = blockIdx.x * 64 + threadIdx.x;
    const int i1 = blockIdx.y * 4 + threadIdx.y;
    const int shot = blockIdx.z;
    if (i2 >= k.n2 || i1 >= k.n1) return;
    const long p = k.off0 + (long)i1 * k.s1 + i2;
    const T zx = q.zx1[i1], zz = q.zz1[i2];
    const long ps = (long)shot * q.phi_shot_stride + p;
    if (q.edges ? !pml_in_layer(q, k, i1, i2) : (zx == T(0) && zz == T(0))) {   // outside the layer the auxiliary fields stay zero
        q.phi1_out[ps] = T(0), q.phi2_out[ps] = T(0);
        return;
    }
    const T *un = q.out + (long)shot * q.shot_stride + p, *uc = q.cur + (long)shot * q.shot_stride + p;
    const long sx = k.s1;
    const T a = (un[sx] + un[sx + 1] - un[0] - un[1]) + (uc[sx] + uc[sx + 1] - uc[0] - uc[1]);
    const T b = (un[1] + un[sx + 1] - un[0] - un[sx]) + (uc[1] + uc[sx + 1] - uc[0] - uc[sx]);
    // c^2 at the staggered point: mean over the corners of the cell that lie inside the grid
    const bool xin = i1 + 1 < k.n1, zin = i2 + 1 < k.n2;
    T c2 = T(1) / q.m[p];
    int nc = 1;
    if (xin) c2 += T(1) / q.m[p + sx], ++nc;
    if (zin) c2 += T(1) / q.m[p + 1], ++nc;
    if (xin && zin) c2 += T(1) / q.m[p + sx + 1], ++nc;
    c2 /= T(nc);
    if (q.c2s) c2 = q.c2s[p];   // the caller's own velocity at the staggered points (the notebook's vel1), squared
    const T idt = T(1) / k.dt;
    q.phi1_out[ps] = (q.phi1[ps] * (idt - T(0.5) * zx) + (zz - zx) * c2 * T(0.25) * q.ihx * a) / (idt + T(0.5) * zx);
    q.phi2_out[ps] = (q.phi2[ps] * (idt - T(0.5) * zz) + (zx - zz) * c2 * T(0.25) * q.ihz * b) / (idt + T(0.5) * zz);
}

template <typename T>
int launch_pml_u(const Plan &p, const StepConsts<T> &k, const PmlPtrs<T> &q, int nshots)
{
    const dim3 block(64, 4, 1), grid((k.n2 + 63) / 64, (k.n1 + 3) / 4, (unsigned)nshots);
    switch (p.L.R) {
        case 1: hipLaunchKernelGGL((pml_u_kernel<T, 1>), grid, block, 0, p.stream, k, q); break;
        case 2: hipLaunchKernelGGL((pml_u_kernel<T, 2>), grid, block, 0, p.stream, k, q); break;
        case 3: hipLaunchKernelGGL((pml_u_kernel<T, 3>), grid, block, 0, p.stream, k, q); break;
        case 4: hipLaunchKernelGGL((pml_u_kernel<T, 4>), grid, block, 0, p.stream, k, q); break;
        case 5: hipLaunchKernelGGL((pml_u_kernel<T, 5>), grid, block, 0, p.stream, k, q); break;
        case 6: hipLaunchKernelGGL((pml_u_kernel<T, 6>), grid, block, 0, p.stream, k, q); break;
        case 7: hipLaunchKernelGGL((pml_u_kernel<T, 7>), grid, block, 0, p.stream, k, q); break;
        default: hipLaunchKernelGGL((pml_u_kernel<T, 8>), grid, block, 0, p.stream, k, q);
    }
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template <typename T>
int launch_pml_phi(const Plan &p, const StepConsts<T> &k, const PmlPtrs<T> &q, int nshots)
{
    const dim3 block(64, 4, 1), grid((k.n2 + 63) / 64, (k.n1 + 3) / 4, (unsigned)nshots);
    hipLaunchKernelGGL((pml_phi_kernel<T>), grid, block, 0, p.stream, k, q);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template <typename T>
int launch_pml_edges(const Plan &p, const StepConsts<T> &k, T *out, long shot_stride, int nshots)
{
    const dim3 block(64, 4, 1), grid((k.n2 + 63) / 64, (k.n1 + 3) / 4, (unsigned)nshots);
    hipLaunchKernelGGL((pml_edges_kernel<T>), grid, block, 0, p.stream, k, out, shot_stride);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template int launch_pml_edges<float>(const Plan &, const StepConsts<float> &, float *, long, int);
template int launch_pml_edges<double>(const Plan &, const StepConsts<double> &, double *, long, int);
template int launch_pml_u<float>(const Plan &, const StepConsts<float> &, const PmlPtrs<float> &, int);
template int launch_pml_u<double>(const Plan &, const StepConsts<double> &, const PmlPtrs<double> &, int);
template int launch_pml_phi<float>(const Plan &, const StepConsts<float> &, const PmlPtrs<float> &, int);
template int launch_pml_phi<double>(const Plan &, const StepConsts<double> &, const PmlPtrs<double> &, int);

// ------------------------------------------------------------------------------------------
// dense <-> haloed copies and model preparation
// ------------------------------------------------------------------------------------------
struct DenseMap {
    int n0, n1, n2;
    long s0, s1, off0, ngrid;
};
static DenseMap dense_map(const Layout &L)
{
    return DenseMap{L.n[0], L.n[1], L.n[2], L.s0, L.s1, L.off0, L.ngrid};
}
__device__ inline long halo_index(const DenseMap &d, long g)
{
    const int i2 = (int)(g % d.n2);
    const long r = g / d.n2;
    const int i1 = (int)(r % d.n1);
    const int i0 = (int)(r / d.n1);
    return d.off0 + (long)i0 * d.s0 + (long)i1 * d.s1 + i2;
}

template <typename T>
__global__ void prep_model_kernel(DenseMap d, const T *vp, const T *damp, T *m, T *eta)
{
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < d.ngrid; g += (long)gridDim.x * blockDim.x) {
        const long h = halo_index(d, g);
        const T v = vp[g];
        m[h] = T(1) / (v * v);  // seismic/model.py:395-400
        eta[h] = damp ? damp[g] : T(0);
    }
}
template <typename T>
int launch_prep_model(const Plan &p, const T *vp, const T *damp, T *m, T *eta)
{
    const int nb = (int)((p.L.ngrid + 255) / 256 < 4096 ? (p.L.ngrid + 255) / 256 : 4096);
    hipLaunchKernelGGL((prep_model_kernel<T>), dim3(nb), dim3(256), 0, p.stream, dense_map(p.L), vp, damp, m, eta);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template int launch_prep_model<float>(const Plan &, const float *, const float *, float *, float *);
template int launch_prep_model<double>(const Plan &, const double *, const double *, double *, double *);

template <typename T>
__global__ void square_kernel(T *a, long n)
{
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < n; g += (long)gridDim.x * blockDim.x) a[g] = a[g] * a[g];
}
template <typename T>
int launch_square(const Plan &p, T *a, long n)
{
    hipLaunchKernelGGL((square_kernel<T>), dim3((int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, p.stream, a, n);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template int launch_square<float>(const Plan &, float *, long);
template int launch_square<double>(const Plan &, double *, long);

template <typename T, int MODE>
__global__ void unpack_kernel(DenseMap d, const T *haloed, long shot_stride, T *dense)
{
    const int shot = blockIdx.y;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < d.ngrid; g += (long)gridDim.x * blockDim.x) {
        const T v = haloed[(long)shot * shot_stride + halo_index(d, g)];
        if (MODE == 0) dense[(long)shot * d.ngrid + g] = v;
        else dense[(long)shot * d.ngrid + g] += v;
    }
}
template <typename T>
int launch_unpack(const Plan &p, const T *haloed, long shot_stride, T *dense, int nshots, int mode)
{
    const int nb = (int)((p.L.ngrid + 255) / 256 < 4096 ? (p.L.ngrid + 255) / 256 : 4096);
    if (mode == 0)
        hipLaunchKernelGGL((unpack_kernel<T, 0>), dim3(nb, nshots), dim3(256), 0, p.stream, dense_map(p.L), haloed,
                           shot_stride, dense);
    else
        hipLaunchKernelGGL((unpack_kernel<T, 1>), dim3(nb, nshots), dim3(256), 0, p.stream, dense_map(p.L), haloed,
                           shot_stride, dense);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template int launch_unpack<float>(const Plan &, const float *, long, float *, int, int);
template int launch_unpack<double>(const Plan &, const double *, long, double *, int, int);

template <typename T>
__global__ void pack_kernel(DenseMap d, const T *dense, T *haloed, long shot_stride)
{
    const int shot = blockIdx.y;
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < d.ngrid; g += (long)gridDim.x * blockDim.x)
        haloed[(long)shot * shot_stride + halo_index(d, g)] = dense[(long)shot * d.ngrid + g];
}
template <typename T>
int launch_pack(const Plan &p, const T *dense, T *haloed, long shot_stride, int nshots)
{
    const int nb = (int)((p.L.ngrid + 255) / 256 < 4096 ? (p.L.ngrid + 255) / 256 : 4096);
    hipLaunchKernelGGL((pack_kernel<T>), dim3(nb, nshots), dim3(256), 0, p.stream, dense_map(p.L), dense, haloed,
                       shot_stride);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template int launch_pack<float>(const Plan &, const float *, float *, long, int);
template int launch_pack<double>(const Plan &, const double *, double *, long, int);

// ------------------------------------------------------------------------------------------
// L2 misfit (misfit/misfit.py:5-9): residual = syn - obs, fval = 0.5 * ||residual||^2.
// Deterministic: fixed block partition, tree reduction in f64, then one block per shot sums the
// partials in index order.
// ------------------------------------------------------------------------------------------
constexpr int kMisfitBlocks = kMisfitPartials;

template <typename T>
__global__ __launch_bounds__(256) void misfit_partial_kernel(const T *syn, const T *obs, T *res, long n, double *partial)
{
    __shared__ double red[256];
    const int shot = blockIdx.y;
    const T *s = syn + (long)shot * n;
    const T *o = obs + (long)shot * n;
    T *r = res + (long)shot * n;
    double acc = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += 256L * kMisfitBlocks) {
        const T d = s[i] - o[i];
        r[i] = d;
        acc += (double)d * (double)d;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[shot * kMisfitBlocks + blockIdx.x] = red[0];
}
__global__ void misfit_final_kernel(const double *partial, double *fval)
{
    const int shot = blockIdx.x;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < kMisfitBlocks; ++b) s += partial[shot * kMisfitBlocks + b];
        fval[shot] = 0.5 * s;
    }
}
template <typename T>
int launch_l2_misfit_on(hipStream_t stream, const T *syn, const T *obs, T *res, long n, int nshots, double *partial,
                        double *fval)
{
    hipLaunchKernelGGL((misfit_partial_kernel<T>), dim3(kMisfitBlocks, nshots), dim3(256), 0, stream, syn, obs, res, n,
                       partial);
    MIFWI_HIP(hipGetLastError());
    hipLaunchKernelGGL(misfit_final_kernel, dim3(nshots), dim3(64), 0, stream, partial, fval);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template int launch_l2_misfit_on<float>(hipStream_t, const float *, const float *, float *, long, int, double *, double *);
template int launch_l2_misfit_on<double>(hipStream_t, const double *, const double *, double *, long, int, double *,
                                         double *);

template <typename T>
int launch_l2_misfit(const Plan &p, const T *syn, const T *obs, T *res, long n, int nshots, double *partial,
                     double *fval)
{
    return launch_l2_misfit_on<T>(p.stream, syn, obs, res, n, nshots, partial, fval);
}
template int launch_l2_misfit<float>(const Plan &, const float *, const float *, float *, long, int, double *,
                                     double *);
template int launch_l2_misfit<double>(const Plan &, const double *, const double *, double *, long, int, double *,
                                      double *);

// ------------------------------------------------------------------------------------------
// Tail of the multi-shot objective.  mute_sum: crop each shot's padded-grid gradient and illumination to the
// physical grid, scale by that shot's source/receiver mute factor and add the shots up in float64, in shot order
// and without contracting the product into the sum -- the arithmetic of `grad += crop(g) * mute` on the host
// (fwi.py:164-171,190-193).  precondition: g / sqrt(illum + 1e-30) and the mask (fwi.py:200-204).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mute_sum_kernel(CropMap c, int nshots, const T *g, const T *il, const double *mute,
                                                       double *gs, double *ils)
{
#pragma clang fp contract(off)   // product and sum round separately, as they do on the host
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= c.nphys) return;
    long rem = i, off = 0;
    for (int d = c.nd - 1; d >= 0; --d) {
        const long q = rem / c.phys[d];
        off += (rem - q * c.phys[d] + c.lo[d]) * c.pad_stride[d];
        rem = q;
    }
    double a = 0.0, b = 0.0;
    for (int s = 0; s < nshots; ++s) {
        const double m = mute ? mute[(long)s * c.nphys + i] : 1.0;
        a += (double)g[(long)s * c.field + off] * m;
        if (il) b += (double)il[(long)s * c.field + off] * m;
    }
    gs[i] = a;
    if (ils) ils[i] = b;
}
template <typename T>
int launch_mute_sum(hipStream_t stream, const CropMap &c, int nshots, const T *g, const T *il, const double *mute,
                    double *gs, double *ils)
{
    const int nb = (int)((c.nphys + 255) / 256);
    hipLaunchKernelGGL((mute_sum_kernel<T>), dim3(nb), dim3(256), 0, stream, c, nshots, g, il, mute, gs, ils);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template int launch_mute_sum<float>(hipStream_t, const CropMap &, int, const float *, const float *, const double *,
                                    double *, double *);
template int launch_mute_sum<double>(hipStream_t, const CropMap &, int, const double *, const double *, const double *,
                                     double *, double *);

__global__ __launch_bounds__(256) void precondition_kernel(long n, const double *g, const double *il, const double *mask,
                                                           double *out)
{
#pragma clang fp contract(off)
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n) return;
    double v = g[i];
    if (il) v = v / sqrt(il[i] + 1e-30);
    if (mask) v = v * mask[i];
    out[i] = v;
}
int launch_precondition(hipStream_t stream, long n, const double *g, const double *il, const double *mask, double *out)
{
    hipLaunchKernelGGL(precondition_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, stream, n, g, il, mask, out);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

// ------------------------------------------------------------------------------------------
// Half-precision saved levels.  Scale of one shot: 64 / max |injected source term| over the whole run
// (terms are dt^2 * wavelet * weight / m, the per-step increments of the field at the source cells).
// The field's peak is about (samples per wavelet period) / 4 such increments in 2-D (12 on Marmousi-II)
// and less in 3-D, so it lands near 8e2 of the half format's 65504 -- a factor ~80 of headroom, values
// beyond it saturate -- with 11 significant bits down to 8e-8 of the peak (the error is the mantissa
// rounding either way: scales of 16 and 256 gave the same gradients to three digits).  A silent shot gets scale 1.
// ------------------------------------------------------------------------------------------
// Only the slots g < uc_count[table] of a (shot, time) row hold terms (inject_prepare_kernel writes no others);
// the rest of the row is whatever the buffer held before and is skipped.
__global__ __launch_bounds__(256) void save_scale_kernel(const float *winj, int nt, int kmax, const int32_t *uc_count,
                                                         int per_shot_tables, float *scale)
{
    __shared__ float red[256];
    const int shot = blockIdx.x;
    float m = 0.f;
    if (winj) {
        const int count = uc_count[per_shot_tables ? shot : 0];
        const float *w = winj + (long)shot * nt * kmax;
        for (long i = threadIdx.x; i < (long)nt * count; i += 256) {
            const long time = i / count;
            m = fmaxf(m, fabsf(w[time * kmax + (i - time * count)]));
        }
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) scale[shot] = (red[0] > 0.f && isfinite(red[0])) ? 64.0f / red[0] : 1.0f;
}

int launch_save_scale(const Plan &p, const float *winj, const SparseSet &S, int nshots, float *scale)
{
    hipLaunchKernelGGL(save_scale_kernel, dim3(nshots), dim3(256), 0, p.stream, winj, p.cfg.nt, S.kmax, S.uc_count,
                       S.per_shot, scale);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

__global__ __launch_bounds__(256) void half_to_float_kernel(const __half *src, float *dst, long n, const float *scale,
                                                            int shot)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = __half2float(src[i]) / scale[shot];
}

int launch_half_to_float(const Plan &p, const void *src, float *dst, long n, const float *scale, int shot)
{
    hipLaunchKernelGGL(half_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.stream,
                       (const __half *)src, dst, n, scale, shot);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

}  // namespace mifwi
