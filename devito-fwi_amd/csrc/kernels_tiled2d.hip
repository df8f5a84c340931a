// Tiled 2-D time-stepping kernel (float32): the fast path for 2-D acquisitions.
//
// One launch advances every shot of the batch by one time step and also does that step's sparse
// work, so a time loop is exactly one kernel per step:
//   * tile blocks: each block of TX*TZ/4 threads owns a TX x TZ tile of one shot.  The tile of the field
//     read through the stencil (plus a halo of R rows / round_up(R,4) columns) is staged in LDS with
//     16-byte loads; every thread then produces 4 consecutive z points from ds_read_b128 reads (the
//     z line once, 2R row neighbours), reads old/m/eta with one aligned 16-byte load each and
//     stores 16 bytes.  The gradient variants also read one saved forward level per step they account
//     for and read-modify-write grad (and illum); a saving forward step also streams the new level (and,
//     every second step, the level it read) into the saved array.
//   * a tile block that owns cells touched by sources/receivers (per-tile slot table built on the
//     host) forms their injected terms in LDS while its stencil loads are in flight -- one thread per
//     cell, fixed summation order, no atomics -- and the owning threads add them before storing.
//   * extra blocks at the start of the grid interpolate the field being read at the receiver (or
//     source) positions.
// Block ids are dealt shot-fastest: with a multiple of 8 shots every block of a shot carries the
// same id modulo 8, i.e. runs on one XCD, whose L2 then holds that shot's working set.
//
// Equations: see kernels_generic.hip (same update and illumination term).  The gradient is the same
// sum rearranged by parts over time,
//     sum_t (u[t-1] - 2u[t] + u[t+1]) v[t]  =  sum_t u[t] (v[t+1] - 2v[t] + v[t-1]),
// which is exact because u[0] = u[1] = 0 and the backward field starts from zero: no boundary
// terms survive.  v[t-1] is the value this step produces, injected receiver term included.
#include "mifwi_internal.hpp"

namespace mifwi {

__device__ inline float fast_div(float a, float b)
{
    // a / b without the IEEE division sequence: reciprocal, one Newton step on it, then one
    // residual correction of the quotient (q += r * (a - b q)), which leaves it within 1 ulp
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}

typedef float vfloat4_2d __attribute__((ext_vector_type(4)));
__device__ inline void st4_out(float *p, float4 v)
{
#ifdef MIFWI_2D_NT_STORE
    vfloat4_2d t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vfloat4_2d *>(p));
#else
    *reinterpret_cast<float4 *>(p) = v;
#endif
}
// 16-byte accesses addressed as a block-uniform (SGPR) base pointer plus a 32-bit per-lane byte offset:
// the cheapest global addressing mode, no 64-bit vector address arithmetic.  One field level stays
// below 4 GB (checked when the plan is created).
__device__ inline float4 ld4_at(const char *base, unsigned off)
{
    return *reinterpret_cast<const float4 *>(base + off);
}
__device__ inline float4 ld4_stream_at(const char *base, unsigned off)
{
#ifdef MIFWI_2D_NT_LOAD
    const vfloat4_2d v = __builtin_nontemporal_load(reinterpret_cast<const vfloat4_2d *>(base + off));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4 *>(base + off);
#endif
}
__device__ inline void st4_at(char *base, unsigned off, float4 v)
{
    *reinterpret_cast<float4 *>(base + off) = v;
}
__device__ inline void st4_stream_at(char *base, unsigned off, float4 v)
{
    vfloat4_2d t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vfloat4_2d *>(base + off));
}
__device__ inline void st4_stream(float *p, float4 v)
{
    vfloat4_2d t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vfloat4_2d *>(p));
}
__device__ inline float4 ld4_saved(const float *p)
{
#ifdef MIFWI_2D_NT_LOAD
    const vfloat4_2d v = __builtin_nontemporal_load(reinterpret_cast<const vfloat4_2d *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4 *>(p);
#endif
}

// Four consecutive points of a half-precision saved level: `off` is the byte offset the same points have
// in a float field, so the halves sit at off / 2 (8-byte aligned).
typedef unsigned vuint2_2d __attribute__((ext_vector_type(2)));
__device__ inline float4 ld4_half_stream_at(const char *base, unsigned off, float inv_scale)
{
    const vuint2_2d v = __builtin_nontemporal_load(reinterpret_cast<const vuint2_2d *>(base + (off >> 1)));
    float4 r;
    unpack_half2(v.x, r.x, r.y);
    unpack_half2(v.y, r.z, r.w);
    return make_float4(r.x * inv_scale, r.y * inv_scale, r.z * inv_scale, r.w * inv_scale);
}
__device__ inline void st4_half_stream_at(char *base, unsigned off, float4 v, float scale)
{
    vuint2_2d t;
    t.x = pack_half2(v.x * scale, v.y * scale);
    t.y = pack_half2(v.z * scale, v.w * scale);
    __builtin_nontemporal_store(t, reinterpret_cast<vuint2_2d *>(base + (off >> 1)));
}

// MODE 0: plain propagation, 1: gradient, 2: scattered field of Born modelling, 4: gradient of two steps
// (a.pair): the step before ran as MODE 0 and its terms are added here, which halves the read-modify-write
// traffic of the accumulators.
template <int R, int TX, int TZ, int MODE>
// Register budget of the gradient variant: 128 VGPRs, i.e. four blocks per CU (unconstrained it takes
// 136 and three).  A batch of 8 shots is ~3.1 tiles per CU, so with three resident blocks the last
// tiles of a launch wait for a free slot: 7.6 -> 6.6 us per step.  The plain variant fits anyway and
// schedules better without the bound.
__global__ __launch_bounds__(TX *TZ / 4, (MODE == 1 || MODE == 4) ? 4 : 1) void step2d_kernel(Step2DArgs a)
{
    constexpr bool GRAD = (MODE == 1 || MODE == 4), PAIR = (MODE == 4), BORN = (MODE == 2);
    constexpr int RZ = (R + 3) / 4 * 4;          // z halo in LDS, multiple of 4
    constexpr int ROWF4 = (TZ + 2 * RZ) / 4;     // float4 per staged row
    constexpr int P = tile2d_lds_pitch(TZ, R);   // LDS row pitch in floats
    constexpr int TQ = TZ / 4;                   // threads per tile row
    constexpr int NT = TX * TQ;                   // threads per block: one float4 of the tile each
    constexpr int NSTAGE = ((TX + 2 * R) * ROWF4 + NT - 1) / NT;  // staged float4 per thread
    static_assert(NT % 64 == 0 && NT <= 1024, "whole wavefronts");
    __shared__ __attribute__((aligned(16))) float lds[(TX + 2 * R) * P];
    __shared__ __attribute__((aligned(16))) float acc[TX * TZ];   // injected terms of this tile

    const int tid = threadIdx.x;
    const int nitp_blocks = a.pt_idx ? ((a.itp_npoint + NT - 1) / NT) * a.nshots : 0;
    int bid = blockIdx.x;

    if (bid < nitp_blocks) {
        // ---- interpolation role (first in the grid: two dependent loads, start them early) -----
        const int shot = a.shot0 + bid % a.nshots;
        const int pnt = (bid / a.nshots) * NT + tid;
        if (pnt >= a.itp_npoint) return;
        const int tab = a.itp_per_shot ? shot : 0;
        const int4 idx = *reinterpret_cast<const int4 *>(a.pt_idx + ((long)tab * a.itp_npoint + pnt) * 4);
        const float4 w = *reinterpret_cast<const float4 *>(a.pt_w + ((long)tab * a.itp_npoint + pnt) * 4);
        const float *f = a.cur + (long)shot * a.shot_stride;
        const float sum = w.x * f[idx.x] + w.y * f[idx.y] + w.z * f[idx.z] + w.w * f[idx.w];
        a.vals_out[((long)shot * a.nt + a.time) * a.itp_npoint + pnt] = sum;
        return;
    }
    bid -= nitp_blocks;

    const int shot = a.shot0 + bid % a.nshots;
    const int tile = bid / a.nshots;
    const int tile_x = tile / a.ntz, tile_z = tile - tile_x * a.ntz;
    const int x0 = tile_x * TX, z0 = tile_z * TZ;
    const int tx = tid / TQ, tq = tid - tx * TQ;
    const int x = x0 + tx, z = z0 + 4 * tq;
    const bool live = (x < a.n1) && (z < a.n2);
    const long p = a.off0 + (long)x * a.s1 + z;
    const long ps = (long)shot * a.shot_stride + p;
    const unsigned pb = ((unsigned)a.off0 + (unsigned)x * (unsigned)a.s1 + (unsigned)z) * 4u;   // byte offset of (x, z)
    const char *curb = reinterpret_cast<const char *>(a.cur + (long)shot * a.shot_stride);
    const char *oldb = reinterpret_cast<const char *>(a.old + (long)shot * a.shot_stride);
    char *outb = reinterpret_cast<char *>(a.out + (long)shot * a.shot_stride);

    // ---- issue every global load up front: they share one memory round trip -------------------
    const int inj_tab = a.inj_per_shot ? shot : 0;
    const int ninj = a.slot_count ? a.slot_count[(long)inj_tab * (a.ntx * a.ntz) + tile] : 0;  // block-uniform
    float4 stage[NSTAGE];
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) {
        const int c = tid + i * NT;
        const int row = c / ROWF4, q = c - row * ROWF4;
        const int gx = x0 - R + row;
        stage[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < (TX + 2 * R) * ROWF4 && gx < a.n1 + R)
            stage[i] = ld4_at(curb, (unsigned)((int)a.off0 + gx * (int)a.s1 + (z0 - RZ + 4 * q)) * 4u);
    }
    float4 m4 = make_float4(1.f, 1.f, 1.f, 1.f), e4 = make_float4(0.f, 0.f, 0.f, 0.f), o4 = e4;
    float4 u04 = e4, up4 = e4, gv = e4, iv = e4, po4 = e4;
    float4 bn4 = e4, bc4 = e4, bo4 = e4, bd4 = e4;
    const long gs = (long)shot * a.g_shot_stride + p;
    if (live) {
        m4 = ld4_at(reinterpret_cast<const char *>(a.m), pb);
        e4 = ld4_at(reinterpret_cast<const char *>(a.eta), pb);
        o4 = ld4_at(oldb, pb);
        if (GRAD) {
            if (a.save_scale) {   // half-precision saved levels (block-uniform branch)
                const float inv = 1.0f / a.save_scale[shot];
                const long hs = (long)shot * a.u_shot_stride * 2;
                u04 = ld4_half_stream_at(reinterpret_cast<const char *>(a.uc) + hs, pb, inv);
                if (PAIR || (a.first && a.illum))
                    up4 = ld4_half_stream_at(reinterpret_cast<const char *>(a.up) + hs, pb, inv);
            } else {
                u04 = ld4_stream_at(reinterpret_cast<const char *>(a.uc + (long)shot * a.u_shot_stride), pb);
                if (PAIR)
                    up4 = ld4_stream_at(reinterpret_cast<const char *>(a.up + (long)shot * a.u_shot_stride), pb);
                else if (a.first && a.illum)
                    up4 = ld4_at(reinterpret_cast<const char *>(a.up + (long)shot * a.u_shot_stride), pb);
            }
            if (PAIR) po4 = ld4_at(outb, pb);   // backward level time + 2, about to be overwritten
            gv = ld4_at(reinterpret_cast<const char *>(a.grad + (long)shot * a.g_shot_stride), pb);
            if (a.illum) iv = ld4_at(reinterpret_cast<const char *>(a.illum + (long)shot * a.g_shot_stride), pb);
        }
        if (BORN) {
            bn4 = ld4_at(reinterpret_cast<const char *>(a.b_un + (long)shot * a.b_shot_stride), pb);
            bc4 = ld4_at(reinterpret_cast<const char *>(a.b_uc + (long)shot * a.b_shot_stride), pb);
            bo4 = ld4_at(reinterpret_cast<const char *>(a.b_uo + (long)shot * a.b_shot_stride), pb);
            bd4 = ld4_at(reinterpret_cast<const char *>(a.b_dm), pb);
        }
    }

    // ---- injected terms of the cells this tile owns (rare: block-uniform branch): the terms were
    // formed before the time loop, so both loads are independent of everything else in flight
    if (ninj > 0) {
        *reinterpret_cast<float4 *>(&acc[4 * tid]) = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        const long tslot = (long)inj_tab * (a.ntx * a.ntz) + tile;
        const float *w = a.winj + ((long)shot * a.nt + a.time) * a.winj_stride + a.slot_base[tslot];
        for (int j = tid; j < ninj; j += NT) acc[a.slot_local[tslot * a.inj_cap + j]] = w[j];
    }

    // ---- stage the tile of `cur` (with halo) in LDS --------------------------------------------
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) {
        const int c = tid + i * NT;
        const int row = c / ROWF4, q = c - row * ROWF4;
        if (c < (TX + 2 * R) * ROWF4) *reinterpret_cast<float4 *>(&lds[row * P + 4 * q]) = stage[i];
    }
    __syncthreads();

    if (live) {
        const float *ctr = &lds[(tx + R) * P + RZ + 4 * tq];
        // z line: floats -RZ .. 4+RZ-1 around the 4 centre points
        float f[4 + 2 * RZ];
#pragma unroll
        for (int q = 0; q < (4 + 2 * RZ) / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(ctr - RZ + 4 * q);
            f[4 * q + 0] = v.x, f[4 * q + 1] = v.y, f[4 * q + 2] = v.z, f[4 * q + 3] = v.w;
        }
        float lap[4], uc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uc[i] = f[RZ + i];
            lap[i] = a.c0 * uc[i];
        }
#pragma unroll
        for (int k = 1; k <= R; ++k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lap[i] += a.cz[k] * (f[RZ + i - k] + f[RZ + i + k]);
        }
#pragma unroll
        for (int k = 1; k <= R; ++k) {
            const float4 lo = *reinterpret_cast<const float4 *>(ctr - k * P);
            const float4 hi = *reinterpret_cast<const float4 *>(ctr + k * P);
            lap[0] += a.cx[k] * (lo.x + hi.x);
            lap[1] += a.cx[k] * (lo.y + hi.y);
            lap[2] += a.cx[k] * (lo.z + hi.z);
            lap[3] += a.cx[k] * (lo.w + hi.w);
        }
        const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ee[4] = {e4.x, e4.y, e4.z, e4.w},
                    oo[4] = {o4.x, o4.y, o4.z, o4.w};
        float res[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ed = ee[i] * a.dt;
            res[i] = fast_div(a.dt2 * lap[i] + 2.0f * mm[i] * uc[i] - mm[i] * oo[i] + ed * uc[i], mm[i] + ed);
        }
        if (BORN) {
            const float bn[4] = {bn4.x, bn4.y, bn4.z, bn4.w}, bc[4] = {bc4.x, bc4.y, bc4.z, bc4.w},
                        bo[4] = {bo4.x, bo4.y, bo4.z, bo4.w}, bd[4] = {bd4.x, bd4.y, bd4.z, bd4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float utt = (bn[i] - 2.0f * bc[i] + bo[i]) * a.inv_dt2;
                res[i] += fast_div(a.dt2 * (-bd[i] * utt), mm[i] + ee[i] * a.dt);
            }
        }
        if (ninj > 0) {
            const float4 add = *reinterpret_cast<const float4 *>(&acc[4 * tid]);
            res[0] += add.x, res[1] += add.y, res[2] += add.z, res[3] += add.w;
        }
        const bool full = (z + 3 < a.n2);
        if (full) {
            st4_at(outb, pb, make_float4(res[0], res[1], res[2], res[3]));
        } else {
            for (int i = 0; i < 4; ++i)
                if (z + i < a.n2) a.out[ps + i] = res[i];
        }
        if (a.copy) {
            // the saved forward levels are write-only streams here: the time loop itself runs on the three
            // rotating levels, which stay cache resident.  `copy_prev` takes the level just read.
            float *dsts[2] = {a.copy, a.copy_prev};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (!dsts[c]) continue;   // block-uniform
                const float *val = c ? uc : res;
                if (a.save_scale) {
                    const float sc = a.save_scale[shot];
                    char *cb = reinterpret_cast<char *>(dsts[c]) + (long)shot * a.copy_shot_stride * 2;
                    if (full) {
                        st4_half_stream_at(cb, pb, make_float4(val[0], val[1], val[2], val[3]), sc);
                    } else {
                        __half *ch = reinterpret_cast<__half *>(cb) + p;
                        for (int i = 0; i < 4; ++i)
                            if (z + i < a.n2) ch[i] = __float2half_rn(fminf(fmaxf(val[i] * sc, -65504.0f), 65504.0f));
                    }
                } else {
                    float *cp = dsts[c] + (long)shot * a.copy_shot_stride + p;
                    if (full) {
                        st4_stream_at(reinterpret_cast<char *>(dsts[c] + (long)shot * a.copy_shot_stride), pb,
                                      make_float4(val[0], val[1], val[2], val[3]));
                    } else {
                        for (int i = 0; i < 4; ++i)
                            if (z + i < a.n2) cp[i] = val[i];
                    }
                }
            }
        }
        if (a.fs && z <= R) {
            // free surface: row -zi of the z halo mirrors row zi with the opposite sign, so that the
            // next step's stencil reads the antisymmetric extension (operators.py:8-35)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int zi = z + i;
                if (zi >= 1 && zi <= R && zi < a.n2) a.out[ps - z - zi] = -res[i];
            }
        }
        if (GRAD) {
            const float u0[4] = {u04.x, u04.y, u04.z, u04.w}, up[4] = {up4.x, up4.y, up4.z, up4.w};
            float g[4] = {gv.x, gv.y, gv.z, gv.w}, il[4] = {iv.x, iv.y, iv.z, iv.w};
            if (PAIR) {
                // terms of the step before (time + 1), added first: the sums keep their order in time
                const float po[4] = {po4.x, po4.y, po4.z, po4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    g[i] = grad_accumulate(g[i], po[i], oo[i], uc[i], up[i], a.inv_dt2);
                    il[i] = illum_accumulate(il[i], up[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // zero-lag correlation, summed by parts over time: the second time difference is taken
                // of the backward field (all three levels are in registers here) instead of the saved
                // forward field, so one saved level is read per step instead of three
                g[i] = grad_accumulate(g[i], oo[i], uc[i], res[i], u0[i], a.inv_dt2);
                if (!PAIR && a.first)
                    il[i] = illum_accumulate(illum_accumulate(il[i], up[i]), u0[i]);   // levels nt-1, nt-2
                else
                    il[i] = illum_accumulate(il[i], u0[i]);
            }
            if (full) {
                st4_at(reinterpret_cast<char *>(a.grad + (long)shot * a.g_shot_stride), pb, make_float4(g[0], g[1], g[2], g[3]));
                if (a.illum)
                    st4_at(reinterpret_cast<char *>(a.illum + (long)shot * a.g_shot_stride), pb,
                           make_float4(il[0], il[1], il[2], il[3]));
            } else {
                for (int i = 0; i < 4; ++i)
                    if (z + i < a.n2) {
                        a.grad[gs + i] = g[i];
                        if (a.illum) a.illum[gs + i] = il[i];
                    }
            }
        }
    }
}

template <int R, int TX, int TZ>
static int launch_rt(const Plan &p, const Step2DArgs &a, int nblocks, hipStream_t stream)
{
    if (a.um && a.pair)
        hipLaunchKernelGGL((step2d_kernel<R, TX, TZ, 4>), dim3(nblocks), dim3(TX * TZ / 4), 0, stream, a);
    else if (a.um)
        hipLaunchKernelGGL((step2d_kernel<R, TX, TZ, 1>), dim3(nblocks), dim3(TX * TZ / 4), 0, stream, a);
    else if (a.b_un)
        hipLaunchKernelGGL((step2d_kernel<R, TX, TZ, 2>), dim3(nblocks), dim3(TX * TZ / 4), 0, stream, a);
    else
        hipLaunchKernelGGL((step2d_kernel<R, TX, TZ, 0>), dim3(nblocks), dim3(TX * TZ / 4), 0, stream, a);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

template <int R>
static int launch_r(const Plan &p, const Step2DArgs &a, int nblocks, hipStream_t stream)
{
    if (p.tile_tz == 64) return launch_rt<R, 16, 64>(p, a, nblocks, stream);
    if (p.tile_tx == 48) return launch_rt<R, 48, 32>(p, a, nblocks, stream);
    if (p.tile_tx == 24) return launch_rt<R, 24, 32>(p, a, nblocks, stream);
    if (p.tile_tx == 16) return launch_rt<R, 16, 32>(p, a, nblocks, stream);
    if (p.tile_tx == 8) return launch_rt<R, 8, 32>(p, a, nblocks, stream);
    return launch_rt<R, 32, 32>(p, a, nblocks, stream);
}

int launch_step2d(const Plan &p, const Step2DArgs &a, hipStream_t stream)
{
    const int nthreads = p.tile_tx * p.tile_tz / 4;
    const int nblocks = a.ntx * a.ntz * a.nshots + (a.pt_idx ? ((a.itp_npoint + nthreads - 1) / nthreads) * a.nshots : 0);
    switch (p.L.R) {
        case 1: return launch_r<1>(p, a, nblocks, stream);
        case 2: return launch_r<2>(p, a, nblocks, stream);
        case 3: return launch_r<3>(p, a, nblocks, stream);
        case 4: return launch_r<4>(p, a, nblocks, stream);
        case 5: return launch_r<5>(p, a, nblocks, stream);
        case 6: return launch_r<6>(p, a, nblocks, stream);
        case 7: return launch_r<7>(p, a, nblocks, stream);
        case 8: return launch_r<8>(p, a, nblocks, stream);
    }
    set_error("unsupported stencil radius");
    return MIFWI_EINVAL;
}

}  // namespace mifwi
