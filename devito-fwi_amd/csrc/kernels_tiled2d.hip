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
typedef unsigned vuint2_2d __attribute__((ext_vector_type(2)));
// 16-byte accesses addressed as a block-uniform (SGPR) base pointer plus a 32-bit per-lane byte offset:
// the cheapest global addressing mode, no 64-bit vector address arithmetic.  One field level stays
// below 4 GB (checked when the plan is created).
__device__ inline float4 ld4_at(const char *base, unsigned off)
{
    return *reinterpret_cast<const float4 *>(base + off);
}

#ifndef MIFWI_2D_V2_BLOCKS
#define MIFWI_2D_V2_BLOCKS 8   /* waves per SIMD: four 512-thread blocks per CU, 64 VGPRs */
#endif
// V consecutive points (V = 2 or 4) as one aligned 8- or 16-byte access
template <int V>
struct PV {
    float v[V];
};
typedef float vfloat2_2d __attribute__((ext_vector_type(2)));
template <int V>
struct VT2;
template <>
struct VT2<2> {
    typedef vfloat2_2d type;
};
template <>
struct VT2<4> {
    typedef vfloat4_2d type;
};
template <int V>
__device__ inline PV<V> pv_fill(float x)
{
    PV<V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = x;
    return r;
}
template <int V>
__device__ inline PV<V> pv_ld(const char *base, unsigned off)
{
    const typename VT2<V>::type t = *reinterpret_cast<const typename VT2<V>::type *>(base + off);
    PV<V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = t[i];
    return r;
}
template <int V>
__device__ inline PV<V> pv_ld_stream(const char *base, unsigned off)
{
#ifdef MIFWI_2D_NT_LOAD
    const typename VT2<V>::type t = __builtin_nontemporal_load(reinterpret_cast<const typename VT2<V>::type *>(base + off));
#else
    const typename VT2<V>::type t = *reinterpret_cast<const typename VT2<V>::type *>(base + off);
#endif
    PV<V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = t[i];
    return r;
}
template <int V>
__device__ inline void pv_st(char *base, unsigned off, const float (&d)[V])
{
    typename VT2<V>::type t;
#pragma unroll
    for (int i = 0; i < V; ++i) t[i] = d[i];
    *reinterpret_cast<typename VT2<V>::type *>(base + off) = t;
}
template <int V>
__device__ inline void pv_st_stream(char *base, unsigned off, const float (&d)[V])
{
    typename VT2<V>::type t;
#pragma unroll
    for (int i = 0; i < V; ++i) t[i] = d[i];
    __builtin_nontemporal_store(t, reinterpret_cast<typename VT2<V>::type *>(base + off));
}
// half-precision saved levels: `off` is the byte offset the same points have in a float field
template <int V>
__device__ inline PV<V> pv_ld_half_stream(const char *base, unsigned off, float inv_scale)
{
    PV<V> r;
    if (V == 4) {
        const vuint2_2d t = __builtin_nontemporal_load(reinterpret_cast<const vuint2_2d *>(base + (off >> 1)));
        unpack_half2(t.x, r.v[0], r.v[1]);
        unpack_half2(t.y, r.v[V - 2], r.v[V - 1]);
    } else {
        const unsigned t = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(base + (off >> 1)));
        unpack_half2(t, r.v[0], r.v[1]);
    }
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] *= inv_scale;
    return r;
}
template <int V>
__device__ inline void pv_st_half_stream(char *base, unsigned off, const float (&d)[V], float scale)
{
    if (V == 4) {
        vuint2_2d t;
        t.x = pack_half2(d[0] * scale, d[1] * scale);
        t.y = pack_half2(d[V - 2] * scale, d[V - 1] * scale);
        __builtin_nontemporal_store(t, reinterpret_cast<vuint2_2d *>(base + (off >> 1)));
    } else {
        __builtin_nontemporal_store(pack_half2(d[0] * scale, d[1] * scale), reinterpret_cast<unsigned *>(base + (off >> 1)));
    }
}
template <int V>
__device__ inline void pv_lds_ld(const float *p, float (&d)[V])
{
    const typename VT2<V>::type t = *reinterpret_cast<const typename VT2<V>::type *>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) d[i] = t[i];
}

// MODE 0: plain propagation, 1: gradient, 2: scattered field of Born modelling, 4: gradient of two steps
// (a.pair): the step before ran as MODE 0 and its terms are added here, which halves the read-modify-write
// traffic of the accumulators.  V: points per thread (TX*TZ/V threads per block).
// The fourth-order-in-time kernel (Step2DArgs::pt, pre_mode, bih) has variants of its own, so that the common ones
// carry none of it: XMODE 5 = mode 0, 6 = mode 1, 7 = mode 2 with those arguments honoured.
template <int R, int TX, int TZ, int XMODE, int V>
// Register budget of the gradient variant: 128 VGPRs, i.e. four blocks per CU (unconstrained it takes
// 136 and three).  A batch of 8 shots is ~3.1 tiles per CU, so with three resident blocks the last
// tiles of a launch wait for a free slot: 7.6 -> 6.6 us per step.  The plain variant fits anyway and
// schedules better without the bound.
__global__ __launch_bounds__(TX *TZ / V, V == 2 ? MIFWI_2D_V2_BLOCKS : ((XMODE == 1 || XMODE == 4) ? 4 : 1)) void step2d_kernel(Step2DArgs a)
{
    constexpr bool OT4 = XMODE >= 5;
    constexpr int MODE = XMODE == 5 ? 0 : (XMODE == 6 ? 1 : (XMODE == 7 ? 2 : XMODE));
    constexpr bool GRAD = (MODE == 1 || MODE == 4), PAIR = (MODE == 4), BORN = (MODE == 2);
    constexpr int RZ = (R + 3) / 4 * 4;          // z halo in LDS, multiple of 4
    constexpr int ROWF4 = (TZ + 2 * RZ) / 4;     // float4 per staged row
    constexpr int P = tile2d_lds_pitch(TZ, R);   // LDS row pitch in floats
    constexpr int TQ = TZ / V;                   // threads per tile row
    constexpr int NT = TX * TQ;                   // threads per block: V points of the tile each
    constexpr int NSTAGE = ((TX + 2 * R) * ROWF4 + NT - 1) / NT;  // staged float4 per thread
    static_assert(NT % 64 == 0 && NT <= 1024, "whole wavefronts");
    __shared__ __attribute__((aligned(16))) float lds[(TX + 2 * R) * P];
    __shared__ __attribute__((aligned(16))) float acc[TX * TZ];   // injected terms of this tile

    // Everything the first loads of a block need is fetched from the kernel arguments in one batch: left to be taken
    // where they are used, behind the branches of the block's role and of the tile decoding, the arguments come in
    // half a dozen dependent steps, each with a wait of its own, before the first vector load goes out.
    asm volatile("" ::"s"(a.pt_idx), "s"(a.pt_w), "s"(a.itp_npoint), "s"(a.itp_per_shot), "s"(a.nshots), "s"(a.shot0),
                 "s"(a.ntz), "s"(a.ntx), "s"(a.shot_stride), "s"(a.cur), "s"(a.old), "s"(a.out), "s"(a.inj_per_shot),
                 "s"(a.slot_count));
    asm volatile("" ::"s"(a.n1), "s"(a.n2), "s"(a.off0), "s"(a.s1), "s"(a.m), "s"(a.eta));
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
        const float *f = (OT4 && a.pt ? a.pt : a.cur) + (long)shot * a.shot_stride;
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
    const int x = x0 + tx, z = z0 + V * tq;
    const bool live = (x < a.n1) && (z < a.n2);
    const long p = a.off0 + (long)x * a.s1 + z;
    const long cur_ss = OT4 && a.cur_shot_stride ? a.cur_shot_stride : a.shot_stride;
    const long out_ss = OT4 && a.out_shot_stride ? a.out_shot_stride : a.shot_stride;
    const long ps = (long)shot * out_ss + p;   // of `out`
    const unsigned pb = ((unsigned)a.off0 + (unsigned)x * (unsigned)a.s1 + (unsigned)z) * 4u;   // byte offset of (x, z)
    const char *curb = reinterpret_cast<const char *>(a.cur + (long)shot * cur_ss);
    const char *oldb = reinterpret_cast<const char *>(a.old + (long)shot * a.shot_stride);
    char *outb = reinterpret_cast<char *>(a.out + (long)shot * out_ss);

    // ---- issue every global load up front: they share one memory round trip -------------------
    const int inj_tab = a.inj_per_shot ? shot : 0;
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
    PV<V> m4 = pv_fill<V>(1.f), e4 = pv_fill<V>(0.f), o4 = e4;
    PV<V> u04 = e4, up4 = e4, gv = e4, iv = e4, po4 = e4;
    PV<V> bn4 = e4, bc4 = e4, bo4 = e4, bd4 = e4, pt4 = e4, bh4 = e4;
    const long gs = (long)shot * a.g_shot_stride + p;
    if (live) {
        m4 = pv_ld<V>(reinterpret_cast<const char *>(a.m), pb);
        e4 = pv_ld<V>(reinterpret_cast<const char *>(a.eta), pb);
        o4 = pv_ld<V>(oldb, pb);
        if (OT4 && a.pt) pt4 = pv_ld<V>(reinterpret_cast<const char *>(a.pt + (long)shot * a.shot_stride), pb);   // block-uniform
        if (OT4 && GRAD && a.bih) bh4 = pv_ld<V>(reinterpret_cast<const char *>(a.bih + (long)shot * a.g_shot_stride), pb);
        if (GRAD) {
            if (a.save_scale) {   // half-precision saved levels (block-uniform branch)
                const float inv = 1.0f / a.save_scale[shot];
                const long hs = (long)shot * a.u_shot_stride * 2;
                u04 = pv_ld_half_stream<V>(reinterpret_cast<const char *>(a.uc) + hs, pb, inv);
                if (PAIR || (a.first && a.illum))
                    up4 = pv_ld_half_stream<V>(reinterpret_cast<const char *>(a.up) + hs, pb, inv);
            } else {
                u04 = pv_ld_stream<V>(reinterpret_cast<const char *>(a.uc + (long)shot * a.u_shot_stride), pb);
                if (PAIR)
                    up4 = pv_ld_stream<V>(reinterpret_cast<const char *>(a.up + (long)shot * a.u_shot_stride), pb);
                else if (a.first && a.illum)
                    up4 = pv_ld<V>(reinterpret_cast<const char *>(a.up + (long)shot * a.u_shot_stride), pb);
            }
            if (PAIR) po4 = pv_ld<V>(outb, pb);   // backward level time + 2, about to be overwritten
            gv = pv_ld<V>(reinterpret_cast<const char *>(a.grad + (long)shot * a.g_shot_stride), pb);
            if (a.illum) iv = pv_ld<V>(reinterpret_cast<const char *>(a.illum + (long)shot * a.g_shot_stride), pb);
        }
        if (BORN) {
            bn4 = pv_ld<V>(reinterpret_cast<const char *>(a.b_un + (long)shot * a.b_shot_stride), pb);
            bc4 = pv_ld<V>(reinterpret_cast<const char *>(a.b_uc + (long)shot * a.b_shot_stride), pb);
            bo4 = pv_ld<V>(reinterpret_cast<const char *>(a.b_uo + (long)shot * a.b_shot_stride), pb);
            bd4 = pv_ld<V>(reinterpret_cast<const char *>(a.b_dm), pb);
        }
    }

    // ---- injected terms of the cells this tile owns (rare: block-uniform branch): the terms were
    // formed before the time loop, so both loads are independent of everything else in flight.  The count is the
    // last load issued: it comes back as a vector load whose value every lane waits for, and ahead of the others
    // that wait held them back.
    // read through the constant address space: a scalar load, whose wait does not hold vector loads back (the table is
    // written before the launch; behind the `asm` above the compiler no longer proves that by itself)
    typedef const __attribute__((address_space(4))) int32_t *ConstI32;
    const int ninj = a.slot_count ? ((ConstI32)(unsigned long)a.slot_count)[(long)inj_tab * (a.ntx * a.ntz) + tile] : 0;   // block-uniform
    if (ninj > 0) {
#pragma unroll
        for (int i = 0; i < V; ++i) acc[V * tid + i] = 0.f;
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
        const float *ctr = &lds[(tx + R) * P + RZ + V * tq];
        // z line: floats -RZ .. V+RZ-1 around the V centre points
        float f[V + 2 * RZ];
#pragma unroll
        for (int q = 0; q < (V + 2 * RZ) / V; ++q) {
            float t[V];
            pv_lds_ld<V>(ctr - RZ + V * q, t);
#pragma unroll
            for (int i = 0; i < V; ++i) f[V * q + i] = t[i];
        }
        float lap[V], uc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            uc[i] = f[RZ + i];
            lap[i] = a.c0 * uc[i];
        }
#pragma unroll
        for (int k = 1; k <= R; ++k) {
#pragma unroll
            for (int i = 0; i < V; ++i) lap[i] += a.cz[k] * (f[RZ + i - k] + f[RZ + i + k]);
        }
#pragma unroll
        for (int k = 1; k <= R; ++k) {
            float lo[V], hi[V];
            pv_lds_ld<V>(ctr - k * P, lo);
            pv_lds_ld<V>(ctr + k * P, hi);
#pragma unroll
            for (int i = 0; i < V; ++i) lap[i] += a.cx[k] * (lo[i] + hi[i]);
        }
        const float(&mm)[V] = m4.v, (&ee)[V] = e4.v, (&oo)[V] = o4.v;
        float res[V];
        if (OT4 && MODE == 0 && a.pre_mode) {   // block-uniform: a field derived from `cur`, not a time step (see Step2DArgs)
#pragma unroll
            for (int i = 0; i < V; ++i)
                res[i] = a.pre_mode == 1 ? uc[i] + fast_div(a.pre_c * lap[i], mm[i])
                                         : (a.pre_mode == 2 ? fast_div(lap[i], mm[i] * mm[i]) : lap[i]);
            if (z + V - 1 < a.n2) {
                pv_st<V>(outb, pb, res);
            } else {
                for (int i = 0; i < V; ++i)
                    if (z + i < a.n2) a.out[ps + i] = res[i];
            }
            return;
        }
        if (OT4 && a.pt) {   // the stencilled field was z; the pointwise terms take the level itself
#pragma unroll
            for (int i = 0; i < V; ++i) uc[i] = pt4.v[i];
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float ed = ee[i] * a.dt;
            res[i] = fast_div(a.dt2 * lap[i] + 2.0f * mm[i] * uc[i] - mm[i] * oo[i] + ed * uc[i], mm[i] + ed);
        }
        if (BORN) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float utt = (bn4.v[i] - 2.0f * bc4.v[i] + bo4.v[i]) * a.inv_dt2;
                res[i] += fast_div(a.dt2 * (-bd4.v[i] * utt), mm[i] + ee[i] * a.dt);
            }
        }
        if (ninj > 0) {
#pragma unroll
            for (int i = 0; i < V; ++i) res[i] += acc[V * tid + i];
        }
        const bool full = (z + V - 1 < a.n2);
        if (full) {
            pv_st<V>(outb, pb, res);
        } else {
            for (int i = 0; i < V; ++i)
                if (z + i < a.n2) a.out[ps + i] = res[i];
        }
        if (a.copy) {
            // the saved forward levels are write-only streams here: the time loop itself runs on the three
            // rotating levels, which stay cache resident.  `copy_prev` takes the level just read.
            float *dsts[2] = {a.copy, a.copy_prev};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (!dsts[c]) continue;   // block-uniform
                const float(&val)[V] = c ? uc : res;
                if (a.save_scale) {
                    const float sc = a.save_scale[shot];
                    char *cb = reinterpret_cast<char *>(dsts[c]) + (long)shot * a.copy_shot_stride * 2;
                    if (full) {
                        pv_st_half_stream<V>(cb, pb, val, sc);
                    } else {
                        __half *ch = reinterpret_cast<__half *>(cb) + p;
                        for (int i = 0; i < V; ++i)
                            if (z + i < a.n2) ch[i] = __float2half_rn(fminf(fmaxf(val[i] * sc, -65504.0f), 65504.0f));
                    }
                } else {
                    float *cp = dsts[c] + (long)shot * a.copy_shot_stride + p;
                    if (full) {
                        pv_st_stream<V>(reinterpret_cast<char *>(dsts[c] + (long)shot * a.copy_shot_stride), pb, val);
                    } else {
                        for (int i = 0; i < V; ++i)
                            if (z + i < a.n2) cp[i] = val[i];
                    }
                }
            }
        }
        if (a.fs && z <= R) {
            // free surface: row -zi of the z halo mirrors row zi with the opposite sign, so that the
            // next step's stencil reads the antisymmetric extension (operators.py:8-35)
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int zi = z + i;
                if (zi >= 1 && zi <= R && zi < a.n2) a.out[ps - z - zi] = -res[i];
            }
        }
        if (GRAD) {
            const float(&u0)[V] = u04.v, (&up)[V] = up4.v;
            float g[V], il[V];
#pragma unroll
            for (int i = 0; i < V; ++i) g[i] = gv.v[i], il[i] = iv.v[i];
            if (PAIR) {
                // terms of the step before (time + 1), added first: the sums keep their order in time
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    g[i] = grad_accumulate(g[i], po4.v[i], oo[i], uc[i], up[i], a.inv_dt2);
                    il[i] = illum_accumulate(il[i], up[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                // zero-lag correlation, summed by parts over time: the second time difference is taken
                // of the backward field (all three levels are in registers here) instead of the saved
                // forward field, so one saved level is read per step instead of three
                g[i] = grad_accumulate(g[i], oo[i], uc[i], res[i], u0[i], a.inv_dt2);
                if (!PAIR && a.first)
                    il[i] = illum_accumulate(illum_accumulate(il[i], up[i]), u0[i]);   // levels nt-1, nt-2
                else
                    il[i] = illum_accumulate(il[i], u0[i]);
            }
            if (OT4 && a.bih) {   // fourth-order-in-time kernel: the biharmonic term of this step's saved level
#pragma unroll
                for (int i = 0; i < V; ++i) g[i] = __builtin_fmaf(-(a.pre_c * bh4.v[i]), uc[i], g[i]);
            }
            if (full) {
                pv_st<V>(reinterpret_cast<char *>(a.grad + (long)shot * a.g_shot_stride), pb, g);
                if (a.illum) pv_st<V>(reinterpret_cast<char *>(a.illum + (long)shot * a.g_shot_stride), pb, il);
            } else {
                for (int i = 0; i < V; ++i)
                    if (z + i < a.n2) {
                        a.grad[gs + i] = g[i];
                        if (a.illum) a.illum[gs + i] = il[i];
                    }
            }
        }
    }
}

template <int R, int TX, int TZ, int V>
static int launch_rtv(const Plan &p, const Step2DArgs &a, hipStream_t stream)
{
    constexpr int NT = TX * TZ / V;
    const int nblocks = a.ntx * a.ntz * a.nshots + (a.pt_idx ? ((a.itp_npoint + NT - 1) / NT) * a.nshots : 0);
    const bool ot4 = a.pt || a.pre_mode || a.bih;
    if (ot4 && a.um)
        hipLaunchKernelGGL((step2d_kernel<R, TX, TZ, 6, V>), dim3(nblocks), dim3(NT), 0, stream, a);
    else if (ot4 && a.b_un)
        hipLaunchKernelGGL((step2d_kernel<R, TX, TZ, 7, V>), dim3(nblocks), dim3(NT), 0, stream, a);
    else if (ot4)
        hipLaunchKernelGGL((step2d_kernel<R, TX, TZ, 5, V>), dim3(nblocks), dim3(NT), 0, stream, a);
    else if (a.um && a.pair)
        hipLaunchKernelGGL((step2d_kernel<R, TX, TZ, 4, V>), dim3(nblocks), dim3(NT), 0, stream, a);
    else if (a.um)
        hipLaunchKernelGGL((step2d_kernel<R, TX, TZ, 1, V>), dim3(nblocks), dim3(NT), 0, stream, a);
    else if (a.b_un)
        hipLaunchKernelGGL((step2d_kernel<R, TX, TZ, 2, V>), dim3(nblocks), dim3(NT), 0, stream, a);
    else
        hipLaunchKernelGGL((step2d_kernel<R, TX, TZ, 0, V>), dim3(nblocks), dim3(NT), 0, stream, a);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

template <int R, int TX, int TZ>
static int launch_rt(const Plan &p, const Step2DArgs &a, int, hipStream_t stream)
{
    // two points per thread (twice the waves per tile) is an experiment of the 32 x 32 tile only
    if constexpr (TX == 32 && TZ == 32) {
        if (p.vec2d == 2) return launch_rtv<R, TX, TZ, 2>(p, a, stream);
    }
    return launch_rtv<R, TX, TZ, 4>(p, a, stream);
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
