// Whole time loops of small 2-D batches in ONE launch (float32, 32 x 32 tiles, at most one shot per XCD).
//
// A time step of a Marmousi-sized batch moves a few megabytes; launched as one kernel per step it costs
// what a dependent kernel boundary plus one load -> LDS -> store round trip costs, not what its bytes
// cost.  Here every tile of a shot keeps a block resident for the whole loop:
//   * the block holds m, eta and its own points of the two newest levels in registers (and, in the
//     gradient variant, its points of the gradient and illumination sums), so per step it reads only the
//     halo ring of the newest level (written by its neighbours) and one saved forward level, and writes its
//     tile of the new level (plus the saved copy in a saving forward loop);
//   * neighbours synchronise through one counter per tile: a block publishes "step s done" after its
//     stores have reached the L2, and waits for its eight neighbours' counters before it reads their
//     halos.  All tiles of a shot run on ONE XCD -- each block reads the XCC id of the hardware it landed
//     on and takes the next free tile of the shot dealt to that XCD -- so the L2 they share is the
//     point of coherence: the counter is stored with group scope, counters and halos are read with loads
//     that miss the per-CU L1, and no cache is flushed or invalidated.  Measured cost of the exchange
//     alone on MI355X: 1.6 us per step against 4.3 us for a kernel per step.
//   * every wait is bounded.  A block that waits too long raises the abort flag and every block leaves;
//     the host then repeats the loop with one kernel per step.  Nothing depends on how the hardware deals
//     blocks to XCDs except speed: an uneven deal only makes the abort path run.
// Arithmetic, sparse terms and summation orders are those of step2d_kernel (kernels_tiled2d.hip).
#include "mifwi_internal.hpp"

// The exchange below is written against the cache hierarchy of gfx942 / gfx950 and nothing else: a per-CU L1
// that writes through, one L2 per XCD that serves every CU of that XCD, device-scope (sc1) loads that miss
// the L1.  On any other target the counter store with group scope promises nothing to another CU.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "kernels_persist2d.hip relies on the L1 / per-XCD L2 behaviour of gfx942 and gfx950"
#endif

namespace mifwi {

typedef int vi4 __attribute__((ext_vector_type(4)));
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
__device__ vf4 raw_buffer_load_f4(vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ int raw_buffer_load_i1(vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
__device__ void raw_buffer_store_f4(vf4 v, vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
__device__ inline vi4 buf_of(const void *p)
{
    union {
        struct {
            const void *p;
            unsigned n, f;
        } s;
        vi4 v;
    } r;
    r.s.p = p;
    r.s.n = 0xffffffffu;
    r.s.f = 0x00020000u;
    return r.v;
}
constexpr int kPlain = 0, kStream = 2 /* non-temporal */, kMissL1 = 16 /* device scope: misses the per-CU L1, served by the L2 */;

__device__ inline float fast_div_r(float a, float b, float r)
{
    // a / b with r = refined reciprocal of b (see fast_div in kernels_tiled2d.hip): same result, the
    // reciprocal is formed once per loop because b = m + eta dt does not change
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
__device__ inline float refined_rcp(float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
}

// MODE 0: propagation (forward or backward in time), 1: backward with gradient and illumination sums
// ISO: equal spacing in x and z (one coefficient set in scalar registers instead of two).
template <int R, int TX, int TZ, int MODE, bool ISO>
__global__ __launch_bounds__(256, 4) void persist2d_kernel(Persist2DArgs q)
{
    constexpr bool GRAD = (MODE == 1);
    constexpr int V = 4, NT = TX * TZ / V;
    static_assert(NT == 256, "tiles of 1024 points");
    constexpr int RZ = (R + 3) / 4 * 4, RZ4 = RZ / 4;
    constexpr int ROWF4 = (TZ + 2 * RZ) / 4;
    constexpr int P = tile2d_lds_pitch(TZ, R);
    constexpr int TQ = TZ / V;
    constexpr int NBAND = R * ROWF4;             // float4 in the band above (below) the tile
    constexpr int NH = 2 * NBAND + 2 * TX * RZ4; // float4 in the halo ring
    constexpr int NHT = (NH + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float lds[(TX + 2 * R) * P];
    __shared__ __attribute__((aligned(16))) float acc[TX * TZ];
    __shared__ int s_list[NT];
    __shared__ short s_own_off[4][NT];   // staged-tile offsets of the four corners of the point a thread samples
    __shared__ float s_own_w[4][NT];
    // gradient and illumination sums of a thread's points: one private slot per thread (no register is free
    // for them at four blocks per CU)
    __shared__ __attribute__((aligned(16))) float s_sums[GRAD ? 2 * V * NT : 4];
    __shared__ int s_misc[4];

    const Step2DArgs &a = q.a;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;   // XCC_ID, bits 3:0
        s_misc[0] = xcc;
        s_misc[1] = atomicAdd(&q.counter[xcc], 1);
        s_misc[2] = 0;
        s_misc[3] = 0;   // set by a thread whose wait ran out
    }
    __syncthreads();
    // this launch advances shots shot0 .. shot0 + nshots - 1 (at most 8), one per XCD
    const int group_slot = s_misc[0], tile = s_misc[1];
    const int ntiles = a.ntx * a.ntz;
    if (group_slot >= a.nshots || tile >= ntiles) return;   // block-uniform: nothing to do on this XCD / surplus block
    if (group_slot == q.drop_slot) return;                  // tests: an XCD that takes no part (its shot stays undone)
    const int shot = a.shot0 + group_slot;

    const int tile_x = tile / a.ntz, tile_z = tile - tile_x * a.ntz;
    const int x0 = tile_x * TX, z0 = tile_z * TZ;
    const int tx = tid / TQ, tq = tid - tx * TQ;
    const int x = x0 + tx, z = z0 + V * tq;
    const bool live = (x < a.n1) && (z < a.n2);
    const bool full = (z + V - 1 < a.n2);
    const unsigned pb = ((unsigned)a.off0 + (unsigned)x * (unsigned)a.s1 + (unsigned)z) * 4u;
    const unsigned lvl_bytes = (unsigned)q.field * 4u;
    float *rot = q.rot + (long)shot * 3 * q.field;
    const vi4 rotB = buf_of(rot);

    // eight neighbours: lanes 0..7 of the block watch one counter each
    int nbr = -1;
    if (tid < 8) {
        const int k = tid < 4 ? tid : tid + 1;   // skip the centre of the 3 x 3 neighbourhood
        const int nx = tile_x + k / 3 - 1, nz = tile_z + k % 3 - 1;
        if (nx >= 0 && nx < a.ntx && nz >= 0 && nz < a.ntz) nbr = nx * a.ntz + nz;
    }
    int *done = q.done + group_slot * ntiles;
    const vi4 doneB = buf_of(done), abortB = buf_of(q.abort_flag);

    // halo-ring pieces (float4) of this thread
    int h_lds[NHT];
    unsigned h_glb[NHT];
    bool h_ok[NHT];
#pragma unroll
    for (int i = 0; i < NHT; ++i) {
        const int h = tid + i * NT;
        int row, c4;
        if (h < 2 * NBAND) {
            const int band = h / NBAND, r = (h - band * NBAND) / ROWF4;
            c4 = h - band * NBAND - r * ROWF4;
            row = band ? (R + TX + r) : r;
        } else {
            const int hh = h - 2 * NBAND;
            const int r = hh / (2 * RZ4), cc = hh - r * (2 * RZ4);
            row = R + r;
            c4 = cc < RZ4 ? cc : (TZ / 4 + cc);
        }
        const int gx = x0 - R + row;
        h_ok[i] = (h < NH) && (gx < a.n1 + R);
        h_lds[i] = row * P + 4 * c4;
        h_glb[i] = (unsigned)((int)a.off0 + gx * (int)a.s1 + (z0 - RZ + 4 * c4)) * 4u;
    }

    // sampled points whose first corner with a weight lies in this tile: at most one per thread
    // (the host checks the capacity before choosing this kernel)
    int own_id = -1;
    if (a.pt_idx) {
        const int tab = a.itp_per_shot ? shot : 0;
        for (int pnt = tid; pnt < a.itp_npoint; pnt += NT) {
            const int4 idx = *reinterpret_cast<const int4 *>(a.pt_idx + ((long)tab * a.itp_npoint + pnt) * 4);
            const float4 w = *reinterpret_cast<const float4 *>(a.pt_w + ((long)tab * a.itp_npoint + pnt) * 4);
            const int first = w.x != 0.f ? idx.x : w.y != 0.f ? idx.y : w.z != 0.f ? idx.z : w.w != 0.f ? idx.w : -1;
            if (first < 0) continue;
            const int rel = first - (int)a.off0;
            const int gx = rel / (int)a.s1, gz = rel - gx * (int)a.s1;
            if (gx / TX == tile_x && gz / TZ == tile_z) {
                const int k = atomicAdd(&s_misc[2], 1);
                if (k < NT) s_list[k] = pnt;
            }
        }
        __syncthreads();
        if (tid < s_misc[2] && tid < NT) {
            own_id = s_list[tid];
            const int4 idx = *reinterpret_cast<const int4 *>(a.pt_idx + ((long)tab * a.itp_npoint + own_id) * 4);
            const float4 w = *reinterpret_cast<const float4 *>(a.pt_w + ((long)tab * a.itp_npoint + own_id) * 4);
            const int ii[4] = {idx.x, idx.y, idx.z, idx.w};
            const float ww[4] = {w.x, w.y, w.z, w.w};
            int own_off[4], fallback = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int rel = ii[c] - (int)a.off0;
                const int gx = rel / (int)a.s1, gz = rel - gx * (int)a.s1;
                own_off[c] = (gx - x0 + R) * P + (gz - z0 + RZ);
                if (ww[c] != 0.f && fallback == 0) fallback = own_off[c];
            }
            // corners without weight point anywhere inside the staged tile
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s_own_off[c][tid] = (short)(ww[c] == 0.f ? fallback : own_off[c]);
                s_own_w[c][tid] = ww[c];
            }
        }
    }

    const int inj_tab = a.inj_per_shot ? shot : 0;
    const long tslot = (long)inj_tab * ntiles + tile;
    const int ninj = a.slot_count ? a.slot_count[tslot] : 0;   // block-uniform
    const int inj_base = ninj > 0 ? a.slot_base[tslot] : 0;
    const int wj_cell = (ninj > 0 && tid < ninj) ? a.slot_local[tslot * a.inj_cap + tid] : -1;   // the cell this thread feeds

    // state kept in registers for the whole loop
    const int time0 = q.forward ? 1 : a.nt - 2;
    const int dir = q.forward ? 1 : -1;
    float mm[V], ed[V], ucur[V], uold[V];
#pragma unroll
    for (int i = 0; i < V; ++i) mm[i] = 1.f, ed[i] = 0.f, ucur[i] = 0.f, uold[i] = 0.f;
    vf4 *const sum_g = reinterpret_cast<vf4 *>(s_sums) + tid, *const sum_il = reinterpret_cast<vf4 *>(s_sums) + (GRAD ? NT : 0) + tid;
    if (GRAD) *sum_g = vf4{0.f, 0.f, 0.f, 0.f}, *sum_il = vf4{0.f, 0.f, 0.f, 0.f};
    const char *usv = GRAD ? reinterpret_cast<const char *>(q.usaved + (long)shot * q.field) : nullptr;
    const long usv_l = q.usaved_lstride * 4;   // bytes between saved levels
    if (live) {
        const float4 m4 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.m) + pb);
        const float4 e4 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.eta) + pb);
        const float4 c4 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(rot) + (long)(time0 % 3) * lvl_bytes + pb);
        const float4 o4 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(rot) + (long)((time0 - dir + 3) % 3) * lvl_bytes + pb);
        mm[0] = m4.x, mm[1] = m4.y, mm[2] = m4.z, mm[3] = m4.w;
        ed[0] = e4.x * a.dt, ed[1] = e4.y * a.dt, ed[2] = e4.z * a.dt, ed[3] = e4.w * a.dt;
        ucur[0] = c4.x, ucur[1] = c4.y, ucur[2] = c4.z, ucur[3] = c4.w;
        uold[0] = o4.x, uold[1] = o4.y, uold[2] = o4.z, uold[3] = o4.w;
        if (GRAD) {
            const float4 g4 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.grad + (long)shot * a.g_shot_stride) + pb);
            *sum_g = vf4{g4.x, g4.y, g4.z, g4.w};
            if (a.illum) {
                float il[V];
                const float4 i4 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.illum + (long)shot * a.g_shot_stride) + pb);
                il[0] = i4.x, il[1] = i4.y, il[2] = i4.z, il[3] = i4.w;
                // the level after the first one enters the illumination once, with the first step
                const float4 p4 = *reinterpret_cast<const float4 *>(usv + (long)(time0 + 1) * usv_l + pb);
                const float pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
                for (int i = 0; i < V; ++i) il[i] = illum_accumulate(il[i], pp[i]);
                *sum_il = vf4{il[0], il[1], il[2], il[3]};
            }
        }
    }

    // the divisor m + eta dt of the update and its refined reciprocal do not change during the loop (the terms of
    // the numerator keep the order of step2d_kernel: regrouping them moved the Marmousi record by 2e-6 of its norm)
    float bb[V], rr[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        bb[i] = mm[i] + ed[i];
        rr[i] = refined_rcp(bb[i]);
    }
    for (int s = 0; s < q.nsteps; ++s) {
        const int time = time0 + dir * s;
        const unsigned cur_off = (unsigned)(time % 3) * lvl_bytes;
        const unsigned out_off = (unsigned)((time + dir + 3) % 3) * lvl_bytes;

        // loads that depend on nothing this step produces: injected terms, the next saved level
        float wj = 0.f;
        if (wj_cell >= 0) wj = a.winj[((long)shot * a.nt + time) * a.winj_stride + inj_base + tid];
        // own points of the newest level: from registers
        *reinterpret_cast<vf4 *>(&lds[(tx + R) * P + RZ + V * tq]) = vf4{ucur[0], ucur[1], ucur[2], ucur[3]};
        if (ninj > 0) {
#pragma unroll
            for (int i = 0; i < V; ++i) acc[V * tid + i] = 0.f;
        }

        // wait until the neighbours have published the level this step reads
        int fail = 0;
        if (s > 0 && nbr >= 0) {
            int n = 0;
            while (raw_buffer_load_i1(doneB, nbr * 4, 0, kMissL1) < s) {
                if (++n > q.spin_limit || ((n & 255) == 0 && raw_buffer_load_i1(abortB, 0, 0, kMissL1) != 0)) {
                    fail = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        // (a flag in LDS and one barrier: the block-wide `or` of the library costs three barriers and a reduction)
        if (fail) s_misc[3] = 1;
        __syncthreads();   // also orders the zeroing of acc before its scatter
        if (s_misc[3]) {
            if (tid == 0) __hip_atomic_store(q.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        vf4 hv[NHT];
#pragma unroll
        for (int i = 0; i < NHT; ++i) {
            hv[i] = vf4{0.f, 0.f, 0.f, 0.f};
            if (h_ok[i]) hv[i] = raw_buffer_load_f4(rotB, (int)h_glb[i], (int)cur_off, kMissL1);
        }
        if (ninj > 0) {
            if (wj_cell >= 0) acc[wj_cell] = wj;
            for (int j = tid + NT; j < ninj; j += NT)   // more owned cells than threads (rare)
                acc[a.slot_local[tslot * a.inj_cap + j]] = a.winj[((long)shot * a.nt + time) * a.winj_stride + inj_base + j];
        }
#pragma unroll
        for (int i = 0; i < NHT; ++i)
            if (tid + i * NT < NH) *reinterpret_cast<vf4 *>(&lds[h_lds[i]]) = hv[i];
        __syncthreads();
        // The saved forward level of this step comes from memory and is needed last.  It is requested only now, when
        // the halo ring has arrived: loads return in the order of issue, so requested first (at the top of the step:
        // 2.71 us per step) it held the ring, which the L2 has ready, back until it had arrived itself, and requested
        // together with the ring (2.63) it competed with the ring of every tile of the shot for the same burst out
        // of the L2; here 2.42.  The gradient step still costs 0.5 us more than the adjoint step, all of it this
        // stream (without the load: 2.07 us); requesting it a step ahead changes nothing -- the wait for the tile
        // store's acknowledgement at the end of the step covers every older load of the wave -- nor does any cache
        // policy of the load, nor blocks of their own that pull the levels into the L2 ahead of the tiles (2.8-3.2).
        vf4 u0 = vf4{0.f, 0.f, 0.f, 0.f};
        if (GRAD && live) u0 = raw_buffer_load_f4(buf_of(usv + (long)time * usv_l), (int)pb, 0, kStream);

        // sampled points of the level being read (stored after this step's level has been published)
        float sampled = 0.f;
        if (own_id >= 0)
            sampled = s_own_w[0][tid] * lds[s_own_off[0][tid]] + s_own_w[1][tid] * lds[s_own_off[1][tid]] +
                      s_own_w[2][tid] * lds[s_own_off[2][tid]] + s_own_w[3][tid] * lds[s_own_off[3][tid]];

        float res[V];
#pragma unroll
        for (int i = 0; i < V; ++i) res[i] = 0.f;
        if (live) {
            const float *ctr = &lds[(tx + R) * P + RZ + V * tq];
            float f[V + 2 * RZ];
#pragma unroll
            for (int k = 0; k < (V + 2 * RZ) / V; ++k) {
                const vf4 t = *reinterpret_cast<const vf4 *>(ctr - RZ + V * k);
#pragma unroll
                for (int i = 0; i < V; ++i) f[V * k + i] = t[i];
            }
            // The sums are formed for pairs of neighbouring outputs with packed two-float multiply-adds, one
            // coefficient per multiply-add (see step3d_kernel): every instruction of this phase is paid for by all
            // the waves that share a SIMD, in every tile of the shot, once per time step.
            float uc[V];
            vf2 lap2[V / 2];
#pragma unroll
            for (int i = 0; i < V; ++i) uc[i] = ucur[i];   // what this thread staged, f[RZ + i]
#pragma unroll
            for (int p = 0; p < V / 2; ++p) lap2[p] = vf2{a.c0, a.c0} * vf2{uc[2 * p], uc[2 * p + 1]};
#pragma unroll
            for (int k = 0; k + 1 < V + 2 * RZ; ++k) {
                const vf2 pk = vf2{f[k], f[k + 1]};
#pragma unroll
                for (int p = 0; p < V / 2; ++p) {
                    const int d = k - RZ - 2 * p, ad = d < 0 ? -d : d;   // distance of the pair from output pair p
                    if (ad >= 1 && ad <= R) lap2[p] = __builtin_elementwise_fma(vf2{a.cz[ad], a.cz[ad]}, pk, lap2[p]);
                }
            }
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                const float cxk = ISO ? a.cz[k] : a.cx[k];
                const vf4 lo = *reinterpret_cast<const vf4 *>(ctr - k * P);
                const vf4 hi = *reinterpret_cast<const vf4 *>(ctr + k * P);
#pragma unroll
                for (int p = 0; p < V / 2; ++p) {
                    lap2[p] = __builtin_elementwise_fma(vf2{cxk, cxk}, vf2{lo[2 * p], lo[2 * p + 1]}, lap2[p]);
                    lap2[p] = __builtin_elementwise_fma(vf2{cxk, cxk}, vf2{hi[2 * p], hi[2 * p + 1]}, lap2[p]);
                }
            }
            float lap[V];
#pragma unroll
            for (int i = 0; i < V; ++i) lap[i] = lap2[i / 2][i % 2];
#pragma unroll
            for (int i = 0; i < V; ++i)
                res[i] = fast_div_r(a.dt2 * lap[i] + 2.0f * mm[i] * uc[i] - mm[i] * uold[i] + ed[i] * uc[i], bb[i], rr[i]);
            if (ninj > 0) {
#pragma unroll
                for (int i = 0; i < V; ++i) res[i] += acc[V * tid + i];
            }
            if (!full) {
                // points past the end of a row are halo: they stay zero in the registers as they do in memory
#pragma unroll
                for (int i = 0; i < V; ++i)
                    if (z + i >= a.n2) res[i] = 0.f;
            }
            if (full) {
                raw_buffer_store_f4(vf4{res[0], res[1], res[2], res[3]}, rotB, (int)pb, (int)out_off, kPlain);
            } else {
                float *po = reinterpret_cast<float *>(reinterpret_cast<char *>(rot) + out_off + pb);
                for (int i = 0; i < V; ++i)
                    if (z + i < a.n2) po[i] = res[i];
            }
            if (a.fs && z <= R) {
                // free surface: row -zi of the z halo mirrors row zi with the opposite sign (see step2d_kernel); the
                // tiles along the surface read it back with their halo ring
                float *row0 = reinterpret_cast<float *>(reinterpret_cast<char *>(rot) + out_off + pb) - z;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const int zi = z + i;
                    if (zi >= 1 && zi <= R && zi < a.n2) row0[-zi] = -res[i];
                }
            }
            if (GRAD) {
                vf4 g = *sum_g, il = *sum_il;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    g[i] = grad_accumulate(g[i], uold[i], uc[i], res[i], u0[i], a.inv_dt2);
                    il[i] = illum_accumulate(il[i], u0[i]);
                }
                *sum_g = g, *sum_il = il;
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) uold[i] = ucur[i], ucur[i] = res[i];

        // publish: this wave's stores have reached the L2, then every wave's, then the counter moves
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&done[tile], s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

        // write-only streams nobody waits for: they drain while the next step waits for its neighbours
        if (own_id >= 0) a.vals_out[((long)shot * a.nt + time) * a.itp_npoint + own_id] = sampled;
        if (q.saved && live) {
            float *sp = q.saved + (long)(time + 1) * q.saved_lstride + (long)shot * q.field;
            if (full) {
                raw_buffer_store_f4(vf4{ucur[0], ucur[1], ucur[2], ucur[3]}, buf_of(sp), (int)pb, 0, kStream);
            } else {
                float *pp = reinterpret_cast<float *>(reinterpret_cast<char *>(sp) + pb);
                for (int i = 0; i < V; ++i)
                    if (z + i < a.n2) pp[i] = ucur[i];
            }
        }
    }

    if (GRAD && live) {
        float *gp = reinterpret_cast<float *>(reinterpret_cast<char *>(a.grad + (long)shot * a.g_shot_stride) + pb);
        float *ip = a.illum ? reinterpret_cast<float *>(reinterpret_cast<char *>(a.illum + (long)shot * a.g_shot_stride) + pb) : nullptr;
        const vf4 g = *sum_g, il = *sum_il;
        for (int i = 0; i < V; ++i)
            if (z + i < a.n2) {
                gp[i] = g[i];
                if (ip) ip[i] = il[i];
            }
    }
}

// After a whole-loop launch: every tile of every shot of the group must have counted all its steps.  A shot whose
// XCD received no block (a device with fewer than 8 XCDs, a partition mode) has waited for nobody and raised
// nothing; this is what notices it.  One block.
__global__ void persist2d_check_kernel(const int *done, int ncounters, int nsteps, int *abort_flag)
{
    int bad = 0;
    for (int i = threadIdx.x; i < ncounters; i += blockDim.x) bad |= (done[i] != nsteps);
    if (bad) *abort_flag = 1;
}

int launch_persist2d_check(const int *done, int ncounters, int nsteps, int *abort_flag, hipStream_t stream)
{
    hipLaunchKernelGGL(persist2d_check_kernel, dim3(1), dim3(256), 0, stream, done, ncounters, nsteps, abort_flag);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

// Which XCC ids do the blocks of a launch land on?  One bit per id seen.
__global__ void xcc_census_kernel(unsigned *mask)
{
    if (threadIdx.x == 0) {
        const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;
        atomicOr(mask, 1u << xcc);
    }
}

// Number of XCDs that blocks of a launch on the current device are dealt to when exactly the ids 0 .. n-1 are
// seen, else 0.  The whole-loop kernel deals one shot to each of 8 ids; a device that shows another set (a
// partition of the GPU, a part with fewer dies) must not run it.
int persist2d_xcd_count(hipStream_t stream)
{
    unsigned *d = nullptr, h = 0;
    if (hipMalloc((void **)&d, sizeof(unsigned)) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    bool ok = hipMemsetAsync(d, 0, sizeof(unsigned), stream) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(xcc_census_kernel, dim3(512), dim3(64), 0, stream, d);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost, stream) == hipSuccess &&
             hipStreamSynchronize(stream) == hipSuccess;
    }
    (void)hipFree(d);
    if (!ok) {
        (void)hipGetLastError();
        return 0;
    }
    int n = 0;
    while (n < 16 && (h >> n & 1u)) ++n;
    return (h >> n) == 0 ? n : 0;
}

template <int R, int TX, int TZ>
static int occupancy_persist_t(bool grad)
{
    int nb = 0;
    // (of the variant with two coefficient sets, which needs no fewer registers than the other)
    hipError_t e = grad ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, persist2d_kernel<R, TX, TZ, 1, false>, 256, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, persist2d_kernel<R, TX, TZ, 0, false>, 256, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return nb;
}
template <int R>
static int occupancy_persist(bool grad, int tile_tz)
{
    return tile_tz == 64 ? occupancy_persist_t<R, 16, 64>(grad) : occupancy_persist_t<R, 32, 32>(grad);
}

// resident blocks per CU of the persistent kernel for this radius and tile shape (32 x 32 or 16 x 64); 0: not available
int persist2d_blocks_per_cu(int R, int tile_tz, bool grad)
{
    switch (R) {
        case 1: return occupancy_persist<1>(grad, tile_tz);
        case 2: return occupancy_persist<2>(grad, tile_tz);
        case 3: return occupancy_persist<3>(grad, tile_tz);
        case 4: return occupancy_persist<4>(grad, tile_tz);
        case 5: return occupancy_persist<5>(grad, tile_tz);
        case 6: return occupancy_persist<6>(grad, tile_tz);
        case 7: return occupancy_persist<7>(grad, tile_tz);
        case 8: return occupancy_persist<8>(grad, tile_tz);
    }
    return 0;
}

template <int R, int TX, int TZ>
static int launch_prt(const Persist2DArgs &q, bool grad, int nblocks, hipStream_t stream)
{
    bool iso = true;
    for (int k = 1; k <= R; ++k) iso = iso && q.a.cx[k] == q.a.cz[k];
    if (grad && iso)
        hipLaunchKernelGGL((persist2d_kernel<R, TX, TZ, 1, true>), dim3(nblocks), dim3(256), 0, stream, q);
    else if (grad)
        hipLaunchKernelGGL((persist2d_kernel<R, TX, TZ, 1, false>), dim3(nblocks), dim3(256), 0, stream, q);
    else if (iso)
        hipLaunchKernelGGL((persist2d_kernel<R, TX, TZ, 0, true>), dim3(nblocks), dim3(256), 0, stream, q);
    else
        hipLaunchKernelGGL((persist2d_kernel<R, TX, TZ, 0, false>), dim3(nblocks), dim3(256), 0, stream, q);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}
template <int R>
static int launch_pr(const Plan &p, const Persist2DArgs &q, bool grad, int nblocks, hipStream_t stream)
{
    if (p.tile_tz == 64) return launch_prt<R, 16, 64>(q, grad, nblocks, stream);
    return launch_prt<R, 32, 32>(q, grad, nblocks, stream);
}

// One launch = the whole time loop.  blocks_per_xcd blocks are dealt to each of the 8 XCDs (ids modulo 8);
// those beyond the tiles of a shot leave at once.
int launch_persist2d(const Plan &p, const Persist2DArgs &q, bool grad, int blocks_per_xcd, hipStream_t stream)
{
    const int nblocks = 8 * blocks_per_xcd;
    switch (p.L.R) {
        case 1: return launch_pr<1>(p, q, grad, nblocks, stream);
        case 2: return launch_pr<2>(p, q, grad, nblocks, stream);
        case 3: return launch_pr<3>(p, q, grad, nblocks, stream);
        case 4: return launch_pr<4>(p, q, grad, nblocks, stream);
        case 5: return launch_pr<5>(p, q, grad, nblocks, stream);
        case 6: return launch_pr<6>(p, q, grad, nblocks, stream);
        case 7: return launch_pr<7>(p, q, grad, nblocks, stream);
        case 8: return launch_pr<8>(p, q, grad, nblocks, stream);
    }
    set_error("unsupported stencil radius");
    return MIFWI_EINVAL;
}

}  // namespace mifwi
