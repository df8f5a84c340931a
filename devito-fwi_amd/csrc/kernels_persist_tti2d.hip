// Whole time loops of small 2-D TTI batches in ONE launch (float32, 32 x 32 tiles, space orders 4 and 8, at most one
// shot per XCD): the two coupled fields of the 'centered' kernel of the reference (seismic/tti/operators.py:180-243,
// 430-481 forward, 484-527 adjoint, 590-639 Jacobian adjoint) on the schedule of kernels_persist2d.hip.
//
// Every tile of a shot keeps a 128-thread block resident for the whole loop; a thread owns two quads of four consecutive
// z points of one row (quads l and l + 4 of the row's eight: a wave's 16-byte LDS reads are then conflict free with a
// row pitch of 80 floats, which is exactly the two fields' rows side by side).  Every per-point value of a thread is a
// pair {quad l, quad l + 4}: written that way all the arithmetic compiles to packed two-float instructions with one
// coefficient each (packing along z would need operands shifted by an odd number of points).  Two waves per tile leave 256 registers
// to a thread at the four tiles per CU that a Marmousi-sized shot needs on its XCD: the state of eight points of two
// fields does not fit 128.  m, eta dt, 1 + 2 eps, sqrt(1 + 2 delta), cos and sin of the tilt and the two newest levels of both fields at
// those points stay in registers; per step a block reads only the halo ring of the two fields' newest level (written
// by its neighbours into three rotating levels), and in the gradient loop one saved level of each background field.
// Neighbours synchronise through one step counter per tile in the L2 of the XCD that holds the shot, exactly as in
// kernels_persist2d.hip (same hardware assumptions, same bounded waits, same abort path).
//
// One step of a block:
//   1. the two stencilled fields A, B (forward: u, v; backward: et p + dl r, dl p + r) of the tile and its ring go
//      to LDS, rows of A and B side by side in one array (row pitch 80 floats, B 40 floats into the row);
//   2. every thread forms Gz = -(sin dA/dx + cos dA/dz) of both fields at its four points: the z neighbours come from
//      the thread's own line of 12 values, the x neighbours as 16-byte LDS reads.  sin * Gz goes to LDS (the x
//      derivative of the second pass needs it from other rows); cos * Gz stays in registers, and the two values each
//      side that the z derivative needs from the neighbouring quads come from the neighbouring LANES (DPP quad
//      permutations: a row of the tile is four consecutive lanes): no LDS traffic, no barrier.  Only the quads at the
//      tile's left and right edge have no such neighbour: 64 threads of the block also form cos * Gz on the two
//      columns outside each edge, and 32 threads sin * Gz on the two rows above and below the tile, one small task
//      each, and pass them through LDS;
//   3. after one barrier every thread updates its four points of both fields, adds the injected terms, stores the new
//      levels, and -- gradient loop -- adds the correlation with the saved background levels to sums it keeps in
//      registers for the whole loop.
// Sums run in the order of tti2d_kernel (kernels_tti.hip); the first-derivative weights carry 1 / h here, which that
// kernel multiplies in afterwards (a rounding's difference).
#include "mifwi_internal.hpp"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "kernels_persist_tti2d.hip relies on the L1 / per-XCD L2 behaviour of gfx942 and gfx950"
#endif

namespace mifwi {

typedef int vi4 __attribute__((ext_vector_type(4)));
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
__device__ vf4 raw_buffer_load_f4(vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float raw_buffer_load_f1(vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ int raw_buffer_load_i1(vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
__device__ void raw_buffer_store_f4(vf4 v, vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
// `size` bytes per lane from a buffer straight into LDS, lane i to lds + i * size (lds is wave-uniform); no registers
// are held while the data is on its way
__device__ void raw_buffer_load_lds(vi4 rsrc, __attribute__((address_space(3))) void *lds, int size, int voffset, int soffset,
                                    int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");
namespace {
__device__ inline vi4 tbuf_of(const void *p)
{
    union {
        struct {
            const void *p;
            unsigned n, f;
        } s;
        vi4 v;
    } r;
    r.s.p = p;
    r.s.n = 0x7fffffffu;   // range of 2 GB - 1 bytes: a lane whose offset is kDead loads zeros and stores nothing
    r.s.f = 0x00020000u;
    return r.v;
}
// 16-byte buffer store whose data registers are then left alone for eight cycles.  On gfx950 a vector instruction
// that overwrites the data registers of a `buffer_store_dwordx4 ... sN offen` in the very next issue slot can reach
// the store: with other waves busy on the same SIMD, lanes 12-15 of each row of 16 stored the NEW content of the
// first data register (seen here as the first word of a quad of u holding v's, run-to-run different results as
// soon as two tiles shared a CU).  The compiler guards the form with a constant offset only (1-2 wait states);
// the empty use below keeps the registers live across the wait.
template <int AUX>
__device__ inline void store_f4_held(vf4 v, vi4 rsrc, int voffset, int soffset)
{
    raw_buffer_store_f4(v, rsrc, voffset, soffset, AUX);
    asm volatile("s_nop 7" : : "v"(v));
}
constexpr unsigned kDead = 0x80000000u;
constexpr int kPlain = 0, kStream = 2 /* non-temporal */, kMissL1 = 16 /* device scope: served by the XCD's L2 */;

__device__ inline float tti_rcp(float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
}
__device__ inline float tti_div(float a, float b, float r)
{
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
// value of the previous (PREV) or next lane of the same group of four lanes, cyclically: quad_perm [3,0,1,2] / [1,2,3,0]
template <bool PREV>
__device__ inline float quad_rotate(float v)
{
    const int i = __builtin_amdgcn_update_dpp(0, __float_as_int(v), PREV ? 0x93 : 0x39, 0xf, 0xf, true);
    return __int_as_float(i);
}
}  // namespace

// MODE 0: forward, 1: backward (adjoint), 2: backward with the gradient correlation
template <int R, int MODE>
__global__ __launch_bounds__(128, 2) void persist_tti2d_kernel(PersistTtiArgs q)
{
    constexpr bool FWD = (MODE == 0), GRAD = (MODE == 2);
    constexpr int TX = 32, TZ = 32, V = 4, H = 2, NT = 128, P = R / 2;   // NT = kPttiThreads (launch_persist_tti2d)
    constexpr int RZ = 4, RZ4 = 1;                 // z halo of the staged fields (floats, quads): >= R, multiple of 4
    constexpr int ROWF4 = (TZ + 2 * RZ) / 4;       // quads per staged row
    constexpr int OFFB = TZ + 2 * RZ, PA = 2 * OFFB;   // B sits behind A in a row of the image: pitch 80 = 16 (mod 64)
    constexpr int OFFG = TZ, PG = 80;              // sin * Gz of A and of B side by side, row pitch 16 (mod 64) as well
    constexpr int TQ = TZ / V, TL = TQ / H;        // quads per row, threads per row
    constexpr int NBAND = R * ROWF4, NH = 2 * NBAND + 2 * TX * RZ4;
    constexpr int NHT = (NH + NT - 1) / NT;
    constexpr int NOWN = 64;                       // sampled points a tile can own
    static_assert(R == 2 || R == 4, "space order 4 or 8");
    static_assert(RZ >= 2 * P && PA % 64 == 16 && PG % 64 == 16 && 2 * P * TQ + 2 * TX <= NT, "layout");
    __shared__ __attribute__((aligned(16))) float sAB[(TX + 2 * R) * PA];
    __shared__ __attribute__((aligned(16))) float sgA[(TX + 2 * P) * PG];
    float *const sgB = sgA + OFFG;
    __shared__ __attribute__((aligned(16))) float acc[TX * TZ];
    __shared__ float zh[2][2][TX][P];              // cos * Gz outside the left / right edge: [field][side][row][column]
    __shared__ float xh_ct[2 * P * TQ][V], xh_st[2 * P * TQ][V];   // tilt at the points of the row tasks
    __shared__ float zt_ct[2 * TX][P], zt_st[2 * TX][P];           // and of the column tasks
    constexpr int NOWN_T = GRAD ? 1 : NOWN;        // the gradient loop samples nothing
    __shared__ int s_own_off[4][NOWN_T];
    __shared__ float s_own_w[4][NOWN_T];
    __shared__ int s_list[NOWN_T];
    // gradient loop: the saved background levels of a step travel from memory straight into LDS (16 bytes per lane,
    // a wave's lanes side by side) and wait there for the update at the end of the step
    __shared__ __attribute__((aligned(16))) float ssave[GRAD ? 2 * H * NT * V : 4];
    __shared__ int s_misc[4];

    const Step2DArgs &a = q.a;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;   // XCC_ID
        s_misc[0] = xcc;
        s_misc[1] = atomicAdd(&q.counter[xcc], 1);
        s_misc[2] = 0;
        s_misc[3] = 0;
    }
    __syncthreads();
    const int group_slot = s_misc[0], tile = s_misc[1];
    const int ntiles = a.ntx * a.ntz;
    if (group_slot >= a.nshots || tile >= ntiles) return;   // block-uniform
    if (group_slot == q.drop_slot) return;                  // tests
    const int shot = a.shot0 + group_slot;

    const int tile_x = tile / a.ntz, tile_z = tile - tile_x * a.ntz;
    const int x0 = tile_x * TX, z0 = tile_z * TZ;
    const int tx = tid / TL, tl = tid - tx * TL;
    const int x = x0 + tx;
    // quad h of this thread: columns zq[h] .. zq[h] + 3.  A quad outside the grid (a ragged last tile) takes part in
    // everything with zeros: its byte offset kDead makes its loads return zero and its stores vanish, and its
    // dt^2 = 0 keeps its points at zero; no branch, no lane mask held in scalar registers for the whole loop.
    int zq[H];
    unsigned pb[H];
    float dt2l[H];
    int flags = 0;   // per-lane predicates as bits of one register (a lane mask each would cost two scalar registers)
    constexpr int F_RAGGED = 1, F_RING0 = 2, F_RING1 = 4, F_LEFT = 8, F_RIGHT = 16, F_ZTASK = 32, F_XTASK = 64, F_OWN = 128,
                  F_FEED = 256, F_POLL = 512, F_LIVE0 = 1024, F_LIVE1 = 2048;
#pragma unroll
    for (int h = 0; h < H; ++h) {
        zq[h] = z0 + V * (tl + TL * h);
        const bool lv = (x < a.n1) && (zq[h] < a.n2);
        if (lv && zq[h] + V - 1 >= a.n2) flags |= F_RAGGED;   // the row ends inside this quad
        if (lv) flags |= h ? F_LIVE1 : F_LIVE0;
        pb[h] = lv ? ((unsigned)a.off0 + (unsigned)x * (unsigned)a.s1 + (unsigned)zq[h]) * 4u : kDead;
        dt2l[h] = lv ? a.dt2 : 0.f;
    }
    if (tl == 0) flags |= F_LEFT;
    if (tl == TL - 1) flags |= F_RIGHT;
    const unsigned lvl_bytes = (unsigned)q.field * 4u;
    float *rotu = q.rot_u + (long)shot * 3 * q.field, *rotv = q.rot_v + (long)shot * 3 * q.field;
    const vi4 rotuB = tbuf_of(rotu), rotvB = tbuf_of(rotv);

    int nbr = -1;
    if (tid < 8) {
        const int k = tid < 4 ? tid : tid + 1;
        const int nx = tile_x + k / 3 - 1, nz = tile_z + k % 3 - 1;
        if (nx >= 0 && nx < a.ntx && nz >= 0 && nz < a.ntz) nbr = nx * a.ntz + nz;
    }
    if (nbr >= 0) flags |= F_POLL;
    int *done = q.done + group_slot * ntiles;
    const vi4 doneB = tbuf_of(done), abortB = tbuf_of(q.abort_flag);

    // element index of the haloed arrays for grid point (gx, gz), and whether that point is inside the grid (outside,
    // the parameter fields hold their halo values: 1 for the cosine and the Thomsen factors, 0 for the sine)
    auto inside = [&](int gx, int gz) { return gx >= 0 && gx < a.n1 && gz >= 0 && gz < a.n2; };
    auto at = [&](int gx, int gz) { return (long)a.off0 + (long)gx * a.s1 + gz; };

    // halo-ring pieces (one float4 per field each) of this thread
    int h_lds[NHT];
    unsigned h_glb[NHT];
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
        const int gx = x0 - R + row, gz = z0 - RZ + 4 * c4;
        const bool ok = (h < NH) && (gx < a.n1 + R);   // else: nothing to load, zeros are staged
        if (h < NH) flags |= i ? F_RING1 : F_RING0;
        h_lds[i] = row * PA + 4 * c4;
        h_glb[i] = ok ? (unsigned)((int)a.off0 + gx * (int)a.s1 + gz) * 4u : kDead;
    }
    // The fields whose halo ring a step reads are the two STENCILLED ones.  Forward they are the levels themselves
    // (three rotating slots).  Backward they are the combinations et p + dl r and dl p + r: every tile publishes
    // them for its own points next to the levels, in two alternating slots (a neighbour is never more than one step
    // behind, so the slot being overwritten is one nobody reads any more), and no tile needs the Thomsen factors of
    // another tile's points.
    float *exa = FWD ? rotu : q.comb_a + (long)shot * 2 * q.field, *exb = FWD ? rotv : q.comb_b + (long)shot * 2 * q.field;
    const vi4 exaB = tbuf_of(exa), exbB = tbuf_of(exb);

    // sampled points whose first weighted corner lies in this tile: at most one per thread of the first NOWN
    // (forward loop: the staged image is sampled; backward loop: the levels themselves, which the image does not hold)
    int own_id = -1;
    if (!GRAD && a.pt_idx) {
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
                if (k < NOWN) s_list[k] = pnt;
            }
        }
        __syncthreads();
        if (tid < s_misc[2] && tid < NOWN) {
            own_id = s_list[tid];
            const int4 idx = *reinterpret_cast<const int4 *>(a.pt_idx + ((long)tab * a.itp_npoint + own_id) * 4);
            const float4 w = *reinterpret_cast<const float4 *>(a.pt_w + ((long)tab * a.itp_npoint + own_id) * 4);
            const int ii[4] = {idx.x, idx.y, idx.z, idx.w};
            const float ww[4] = {w.x, w.y, w.z, w.w};
            int own_off[4], fallback = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (FWD) {
                    const int rel = ii[c] - (int)a.off0;
                    const int gx = rel / (int)a.s1, gz = rel - gx * (int)a.s1;
                    own_off[c] = (gx - x0 + R) * PA + (gz - z0 + RZ);
                } else {
                    own_off[c] = ii[c] * 4;   // byte offset into a level
                }
                if (ww[c] != 0.f && fallback == 0) fallback = own_off[c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s_own_off[c][tid] = ww[c] == 0.f ? fallback : own_off[c];
                s_own_w[c][tid] = ww[c];
            }
        }
    }

    const int inj_tab = a.inj_per_shot ? shot : 0;
    const long tslot = (long)inj_tab * ntiles + tile;
    const int ninj = a.slot_count ? a.slot_count[tslot] : 0;   // block-uniform
    const int inj_base = ninj > 0 ? a.slot_base[tslot] : 0;
    const int wj_cell = (ninj > 0 && tid < ninj) ? a.slot_local[tslot * a.inj_cap + tid] : -1;
    if (wj_cell >= 0) flags |= F_FEED;
    if (own_id >= 0) flags |= F_OWN;
    if (tid < 2 * TX) flags |= F_ZTASK;
    else if (tid < 2 * TX + 2 * P * TQ) flags |= F_XTASK;

    // tilt at the points of the small tasks: columns left and right of the tile (threads 0 .. 2 TX - 1, P points each),
    // rows above and below it (the 2 P TQ threads after them, one quad each)
    constexpr int XT0 = 2 * TX;   // first thread with a row task
    if (tid >= XT0 && tid < XT0 + 2 * P * TQ) {
        const int t = tid - XT0;
        const int hrow = t / TQ, hq = t - hrow * TQ;
        const int gx = x0 + (hrow < P ? hrow - P : TX + hrow - P), gz = z0 + V * hq;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const bool in = inside(gx, gz + i);
            xh_ct[t][i] = in ? q.ct[at(gx, gz + i)] : 1.f;
            xh_st[t][i] = in ? q.st[at(gx, gz + i)] : 0.f;
        }
    }
    if (tid < 2 * TX) {
        const int side = tid / TX, r = tid - side * TX;
        const int gx = x0 + r, gz = side ? z0 + TZ : z0 - P;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const bool in = inside(gx, gz + i);
            zt_ct[tid][i] = in ? q.ct[at(gx, gz + i)] : 1.f;
            zt_st[tid][i] = in ? q.st[at(gx, gz + i)] : 0.f;
        }
    }

    // State kept in registers for the whole loop.  The thread's two quads are the two components of a pair: every
    // per-point value is a vf2 {quad l, quad l + 4}, so that one packed instruction (v_pk_fma_f32 and friends) serves
    // both quads with the same coefficient -- no operand has to be shifted by an odd number of points, which is what
    // keeps the compiler from packing along z.
    const int time0 = q.forward ? 1 : a.nt - 2;
    const int dir = q.forward ? 1 : -1;
    vf2 mm[V], ed[V], et[V], dl[V], ct[V], st[V], ucur[V], uold[V], vcur[V], vold[V], gsum[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        mm[i] = vf2{1.f, 1.f}, ed[i] = vf2{0.f, 0.f}, et[i] = vf2{1.f, 1.f}, dl[i] = vf2{1.f, 1.f}, ct[i] = vf2{1.f, 1.f};
        st[i] = vf2{0.f, 0.f}, ucur[i] = vf2{0.f, 0.f}, uold[i] = vf2{0.f, 0.f}, vcur[i] = vf2{0.f, 0.f}, vold[i] = vf2{0.f, 0.f};
        gsum[i] = vf2{0.f, 0.f};
#pragma unroll
        for (int h = 0; h < H; ++h)
            if (x < a.n1 && zq[h] + i < a.n2) {   // (set-up only)
                const long e = at(x, zq[h] + i);
                mm[i][h] = a.m[e], ed[i][h] = a.eta[e] * a.dt, et[i][h] = q.et[e], dl[i][h] = q.dl[e], ct[i][h] = q.ct[e];
                st[i][h] = q.st[e];
                const long c = (long)(time0 % 3) * q.field + e, o = (long)((time0 - dir + 3) % 3) * q.field + e;
                ucur[i][h] = rotu[c], uold[i][h] = rotu[o], vcur[i][h] = rotv[c], vold[i][h] = rotv[o];
                if (GRAD) gsum[i][h] = q.grad[(long)shot * q.g_shot_stride + e];
            }
    }
    const vf2 dt2p = vf2{dt2l[0], dt2l[1]};
    const char *su_base = GRAD ? reinterpret_cast<const char *>(q.su + (long)shot * q.s_shot_stride) : nullptr;
    const char *sv_base = GRAD ? reinterpret_cast<const char *>(q.sv + (long)shot * q.s_shot_stride) : nullptr;

    // first-derivative weights with 1 / h folded in (one rounding more per weight, two multiplications fewer per term)
    float a1x[P + 1], a1z[P + 1];
#pragma unroll
    for (int j = 0; j <= P; ++j) a1x[j] = q.a1[j] * q.ihx, a1z[j] = q.a1[j] * q.ihz;

    int ctr[H];   // this thread's quads in the image (A; B is OFFB further)
#pragma unroll
    for (int h = 0; h < H; ++h) ctr[h] = (tx + R) * PA + RZ + V * (tl + TL * h);
    // a quad of each of the thread's two: 16-byte LDS reads at the two places, combined point by point
    auto ld2 = [&](const float *base, int o0, int o1, vf2 *out) {
        const vf4 x0 = *reinterpret_cast<const vf4 *>(base + o0), x1 = *reinterpret_cast<const vf4 *>(base + o1);
#pragma unroll
        for (int i = 0; i < V; ++i) out[i] = vf2{x0[i], x1[i]};
    };
    auto st2 = [&](float *base, int o0, int o1, const vf2 *in) {
        *reinterpret_cast<vf4 *>(base + o0) = vf4{in[0][0], in[1][0], in[2][0], in[3][0]};
        *reinterpret_cast<vf4 *>(base + o1) = vf4{in[0][1], in[1][1], in[2][1], in[3][1]};
    };
    __syncthreads();   // the task tables are complete

    for (int s = 0; s < q.nsteps; ++s) {
        const int time = time0 + dir * s;
        const unsigned cur_off = (unsigned)(time % 3) * lvl_bytes;
        const unsigned out_off = (unsigned)((time + dir + 3) % 3) * lvl_bytes;
        // slot of the exchanged fields that holds the level being read, and the one the new level goes to
        const unsigned ex_cur = FWD ? cur_off : (unsigned)(s & 1) * lvl_bytes, ex_out = FWD ? out_off : (unsigned)((s + 1) & 1) * lvl_bytes;

        // Values that do not change during the loop but must not be expanded by the compiler into what it can derive from
        // them -- a lane mask per predicate, the reciprocal of m + eta dt and 2 m per point, all kept for the whole loop:
        // that is what made it spill.  Behind these empty statements it has to form them where they are used.
        int fl = flags;
        asm volatile("" : "+v"(fl));
#pragma unroll
        for (int i = 0; i < V; ++i) asm volatile("" : "+v"(mm[i]), "+v"(ed[i]));

        float wj = 0.f;
        if (fl & F_FEED) wj = a.winj[((long)shot * a.nt + time) * a.winj_stride + inj_base + tid];
        // own points of the two stencilled fields: from registers (backward loops: the combinations formed at the
        // end of the previous step are in place already)
        if (FWD || s == 0) {
            vf2 fa[V], fb[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                fa[i] = FWD ? ucur[i] : et[i] * ucur[i] + dl[i] * vcur[i];
                fb[i] = FWD ? vcur[i] : dl[i] * ucur[i] + vcur[i];
            }
            st2(sAB, ctr[0], ctr[1], fa);
            st2(sAB, ctr[0] + OFFB, ctr[1] + OFFB, fb);
        }
        if (ninj > 0) {
#pragma unroll
            for (int h = 0; h < H; ++h) *reinterpret_cast<vf4 *>(&acc[tx * TZ + V * (tl + TL * h)]) = vf4{0.f, 0.f, 0.f, 0.f};
        }

        // wait until the neighbours have published the level this step reads
        int fail = 0;
        if (s > 0 && (fl & F_POLL)) {
            int n = 0;
            while (raw_buffer_load_i1(doneB, nbr * 4, 0, kMissL1) < s) {
                if (++n > q.spin_limit || ((n & 255) == 0 && raw_buffer_load_i1(abortB, 0, 0, kMissL1) != 0)) {
                    fail = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (fail) s_misc[3] = 1;
        __syncthreads();
        if (s_misc[3]) {
            if (tid == 0) __hip_atomic_store(q.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        vf4 hu[NHT], hv[NHT];
#pragma unroll
        for (int i = 0; i < NHT; ++i) {
            hu[i] = raw_buffer_load_f4(exaB, (int)h_glb[i], (int)ex_cur, kMissL1);
            hv[i] = raw_buffer_load_f4(exbB, (int)h_glb[i], (int)ex_cur, kMissL1);
        }
        // backward loop without gradient: the adjoint fields are sampled at the source positions, from the levels
        float sampled = 0.f;
        if (!FWD && (fl & F_OWN)) {
            float sm = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int off = s_own_off[c][tid];
                const float f = raw_buffer_load_f1(rotuB, off, (int)cur_off, kMissL1);
                const float g = raw_buffer_load_f1(rotvB, off, (int)cur_off, kMissL1);
                sm += s_own_w[c][tid] * (f + g);
            }
            sampled = sm;
        }
        if (ninj > 0) {
            if (fl & F_FEED) acc[wj_cell] = wj;
            if (__builtin_expect(ninj > NT, 0)) {   // more owned cells than threads (rare)
#pragma unroll 1
                for (int j = tid + NT; j < q.a.slot_count[tslot]; j += NT)
                    acc[q.a.slot_local[tslot * q.a.inj_cap + j]] =
                        q.a.winj[((long)shot * q.a.nt + time) * q.a.winj_stride + q.a.slot_base[tslot] + j];
            }
        }
#pragma unroll
        for (int i = 0; i < NHT; ++i)
            if (fl & (i ? F_RING1 : F_RING0)) {
                *reinterpret_cast<vf4 *>(&sAB[h_lds[i]]) = hu[i];
                *reinterpret_cast<vf4 *>(&sAB[h_lds[i] + OFFB]) = hv[i];
            }
        __syncthreads();
        // saved background levels of this step: requested when the ring has arrived (see kernels_persist2d.hip)
        if (GRAD) {
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            const vi4 suB = tbuf_of(su_base + (long)time * lvl_bytes), svB = tbuf_of(sv_base + (long)time * lvl_bytes);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                raw_buffer_load_lds(suB, (__attribute__((address_space(3))) void *)&ssave[((0 * H + h) * NT + wave * 64) * V], 16,
                                    (int)pb[h], 0, 0, kStream);
                raw_buffer_load_lds(svB, (__attribute__((address_space(3))) void *)&ssave[((1 * H + h) * NT + wave * 64) * V], 16,
                                    (int)pb[h], 0, 0, kStream);
            }
        }
        if (FWD && (fl & F_OWN)) {
            float sm = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int off = s_own_off[c][tid];
                sm += s_own_w[c][tid] * (sAB[off] + sAB[off + OFFB]);
            }
            sampled = sm;
        }

        // ---- Gz of both fields at this thread's points; Laplacian of A -----------------------------------------
        vf2 lap[V], cgA[V], cgB[V];
        __builtin_amdgcn_sched_barrier(0);   // one field at a time: the phases are long enough as they are
        {
            // field A: its line of 12 values serves the z part of the Laplacian and the z derivative, the rows above
            // and below the x parts
            vf2 la[V + 2 * RZ];
#pragma unroll
            for (int k = 0; k < (V + 2 * RZ) / V; ++k) ld2(sAB, ctr[0] - RZ + V * k, ctr[1] - RZ + V * k, &la[V * k]);
            vf2 dx[V], dz[V];
#pragma unroll
            for (int i = 0; i < V; ++i) lap[i] = a.c0 * la[RZ + i], dx[i] = vf2{0.f, 0.f}, dz[i] = vf2{0.f, 0.f};
#pragma unroll
            for (int j = 1; j <= R; ++j)
#pragma unroll
                for (int i = 0; i < V; ++i) lap[i] += a.cz[j] * (la[RZ + i - j] + la[RZ + i + j]);
#pragma unroll
            for (int j = 1; j <= P; ++j)
#pragma unroll
                for (int i = 0; i < V; ++i) dz[i] += a1z[j] * (la[RZ + i + j] - la[RZ + i - j]);
#pragma unroll
            for (int j = 1; j <= R; ++j) {
                vf2 lo[V], hi[V];
                ld2(sAB, ctr[0] - j * PA, ctr[1] - j * PA, lo);
                ld2(sAB, ctr[0] + j * PA, ctr[1] + j * PA, hi);
#pragma unroll
                for (int i = 0; i < V; ++i) lap[i] += a.cx[j] * (lo[i] + hi[i]);
                if (j <= P) {
#pragma unroll
                    for (int i = 0; i < V; ++i) dx[i] += a1x[j] * (hi[i] - lo[i]);
                }
            }
            vf2 sg[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const vf2 g = -(st[i] * dx[i] + ct[i] * dz[i]);
                cgA[i] = ct[i] * g, sg[i] = st[i] * g;
            }
            st2(sgA, (tx + P) * PG + V * tl, (tx + P) * PG + V * (tl + TL), sg);
            // the sums are complete HERE: left to itself the compiler moves them down to their use in the update and
            // keeps the quads they are formed from alive until then (registers spilled)
#pragma unroll
            for (int i = 0; i < V; ++i) asm volatile("" : "+v"(lap[i]), "+v"(cgA[i]));
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            vf2 lb[V + 2 * P];   // z - P .. z + 3 + P
            {
                const vf2 l0 = *reinterpret_cast<const vf2 *>(&sAB[ctr[0] + OFFB - 2]), l1 = *reinterpret_cast<const vf2 *>(&sAB[ctr[1] + OFFB - 2]);
                const vf2 r0 = *reinterpret_cast<const vf2 *>(&sAB[ctr[0] + OFFB + V]), r1 = *reinterpret_cast<const vf2 *>(&sAB[ctr[1] + OFFB + V]);
                if (P == 2)
                    lb[0] = vf2{l0[0], l1[0]}, lb[1] = vf2{l0[1], l1[1]}, lb[V + 2] = vf2{r0[0], r1[0]}, lb[V + 3] = vf2{r0[1], r1[1]};
                else
                    lb[0] = vf2{l0[1], l1[1]}, lb[V + 1] = vf2{r0[0], r1[0]};
                ld2(sAB, ctr[0] + OFFB, ctr[1] + OFFB, &lb[P]);
            }
            vf2 dx[V], dz[V];
#pragma unroll
            for (int i = 0; i < V; ++i) dx[i] = vf2{0.f, 0.f}, dz[i] = vf2{0.f, 0.f};
#pragma unroll
            for (int j = 1; j <= P; ++j) {
                vf2 lo[V], hi[V];
                ld2(sAB, ctr[0] + OFFB - j * PA, ctr[1] + OFFB - j * PA, lo);
                ld2(sAB, ctr[0] + OFFB + j * PA, ctr[1] + OFFB + j * PA, hi);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    dx[i] += a1x[j] * (hi[i] - lo[i]);
                    dz[i] += a1z[j] * (lb[P + i + j] - lb[P + i - j]);
                }
            }
            vf2 sg[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const vf2 g = -(st[i] * dx[i] + ct[i] * dz[i]);
                cgB[i] = ct[i] * g, sg[i] = st[i] * g;
            }
            st2(sgB, (tx + P) * PG + V * tl, (tx + P) * PG + V * (tl + TL), sg);
#pragma unroll
            for (int i = 0; i < V; ++i) asm volatile("" : "+v"(cgB[i]));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- the small tasks: cos * Gz on the columns beside the tile, sin * Gz on the rows above and below it --------
        if (fl & F_ZTASK) {
            const int side = tid / TX, r = tid - side * TX;
            const int c0i = (r + R) * PA + RZ + (side ? TZ : -P);   // first of the P columns in the image
#pragma unroll
            for (int i = 0; i < P; ++i) {
                float dxa = 0.f, dza = 0.f, dxb = 0.f, dzb = 0.f;
#pragma unroll
                for (int j = 1; j <= P; ++j) {
                    dxa += a1x[j] * (sAB[c0i + i + j * PA] - sAB[c0i + i - j * PA]);
                    dza += a1z[j] * (sAB[c0i + i + j] - sAB[c0i + i - j]);
                    dxb += a1x[j] * (sAB[c0i + OFFB + i + j * PA] - sAB[c0i + OFFB + i - j * PA]);
                    dzb += a1z[j] * (sAB[c0i + OFFB + i + j] - sAB[c0i + OFFB + i - j]);
                }
                const float cv = zt_ct[tid][i], sv = zt_st[tid][i];
                const float ga = -(sv * dxa + cv * dza);
                const float gb = -(sv * dxb + cv * dzb);
                zh[0][side][r][i] = cv * ga;
                zh[1][side][r][i] = cv * gb;
            }
        } else if (fl & F_XTASK) {
            const int t = tid - XT0;
            const int hrow = t / TQ, hq = t - hrow * TQ;
            const int er = hrow < P ? hrow : TX + hrow;              // row of the sin * Gz arrays
            const int c0i = (er - P + R) * PA + RZ + V * hq;        // the quad in the image
            float la[V + 2 * P], lb[V + 2 * P];
#pragma unroll
            for (int i = 0; i < V + 2 * P; ++i) la[i] = sAB[c0i - P + i], lb[i] = sAB[c0i + OFFB - P + i];
            vf4 sa4, sb4;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float dxa = 0.f, dza = 0.f, dxb = 0.f, dzb = 0.f;
#pragma unroll
                for (int j = 1; j <= P; ++j) {
                    dxa += a1x[j] * (sAB[c0i + i + j * PA] - sAB[c0i + i - j * PA]);
                    dza += a1z[j] * (la[P + i + j] - la[P + i - j]);
                    dxb += a1x[j] * (sAB[c0i + OFFB + i + j * PA] - sAB[c0i + OFFB + i - j * PA]);
                    dzb += a1z[j] * (lb[P + i + j] - lb[P + i - j]);
                }
                const float cv = xh_ct[t][i], sv = xh_st[t][i];
                const float ga = -(sv * dxa + cv * dza);
                const float gb = -(sv * dxb + cv * dzb);
                sa4[i] = sv * ga, sb4[i] = sv * gb;
            }
            *reinterpret_cast<vf4 *>(&sgA[er * PG + V * hq]) = sa4;
            *reinterpret_cast<vf4 *>(&sgB[er * PG + V * hq]) = sb4;
        }
        __syncthreads();
        // the saved levels are in LDS (vmcnt(0); waited for here, before this step's stores join the queue)
        if (GRAD) __builtin_amdgcn_s_waitcnt(0x0f70);

        // ---- update of both fields -------------------------------------------------------------------------------------
        const bool ragged = (fl & F_RAGGED) != 0;   // a row of the grid ends inside one of this lane's quads (rare)
        __builtin_amdgcn_sched_barrier(0);
        vf2 un[V], vn[V];
        {
            // the z lines of cos * Gz through the two quads.  A row of the tile is four consecutive lanes, and the
            // permutations are cyclic within them.  Left of quad l lies the previous lane's first quad (lane 0: outside
            // the tile, from the column tasks); right of it the next lane's first quad, except for lane 3, where it is
            // quad 4 = lane 0's second quad = the cyclically next lane's SECOND quad.  Left of quad l + 4 lies the
            // previous lane's second quad, except for lane 0, where it is quad 3 = the cyclically previous lane's FIRST
            // quad; right of it the next lane's second quad (lane 3: outside the tile).
            vf2 eA[V + 2 * P], eB[V + 2 * P];
#pragma unroll
            for (int i = 0; i < V; ++i) eA[P + i] = cgA[i], eB[P + i] = cgB[i];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const float pa0 = quad_rotate<true>(cgA[V - P + i][0]), pa1 = quad_rotate<true>(cgA[V - P + i][1]);
                const float pv0 = quad_rotate<true>(cgB[V - P + i][0]), pv1 = quad_rotate<true>(cgB[V - P + i][1]);
                const float na0 = quad_rotate<false>(cgA[i][0]), na1 = quad_rotate<false>(cgA[i][1]);
                const float nv0 = quad_rotate<false>(cgB[i][0]), nv1 = quad_rotate<false>(cgB[i][1]);
                eA[i] = vf2{pa0, (fl & F_LEFT) ? pa0 : pa1}, eB[i] = vf2{pv0, (fl & F_LEFT) ? pv0 : pv1};
                eA[P + V + i] = vf2{(fl & F_RIGHT) ? na1 : na0, na1}, eB[P + V + i] = vf2{(fl & F_RIGHT) ? nv1 : nv0, nv1};
            }
            if (fl & F_LEFT) {
#pragma unroll
                for (int i = 0; i < P; ++i) eA[i][0] = zh[0][0][tx][i], eB[i][0] = zh[1][0][tx][i];
            }
            if (fl & F_RIGHT) {
#pragma unroll
                for (int i = 0; i < P; ++i) eA[P + V + i][1] = zh[0][1][tx][i], eB[P + V + i][1] = zh[1][1][tx][i];
            }
            const int gq0 = (tx + P) * PG + V * tl, gq1 = gq0 + V * TL;
            vf2 sa[V], sb[V];
#pragma unroll
            for (int i = 0; i < V; ++i) sa[i] = vf2{0.f, 0.f}, sb[i] = vf2{0.f, 0.f};
#pragma unroll
            for (int j = 1; j <= P; ++j) {
                vf2 alo[V], ahi[V];
                ld2(sgA, gq0 - j * PG, gq1 - j * PG, alo);
                ld2(sgA, gq0 + j * PG, gq1 + j * PG, ahi);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    sa[i] += a1z[j] * (eA[P + i + j] - eA[P + i - j]);
                    sa[i] += a1x[j] * (ahi[i] - alo[i]);
                }
            }
#pragma unroll
            for (int j = 1; j <= P; ++j) {
                vf2 blo[V], bhi[V];
                ld2(sgB, gq0 - j * PG, gq1 - j * PG, blo);
                ld2(sgB, gq0 + j * PG, gq1 + j * PG, bhi);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    sb[i] += a1z[j] * (eB[P + i + j] - eB[P + i - j]);
                    sb[i] += a1x[j] * (bhi[i] - blo[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const vf2 X = lap[i] + sa[i], Y = -sb[i];   // Gxx = laplace - Gzz with Gzz = -sa
                const vf2 H0 = FWD ? et[i] * X + dl[i] * Y : X;
                const vf2 Hz = FWD ? dl[i] * X + Y : Y;
                const vf2 uc = ucur[i], vc = vcur[i], mz = mm[i], e = ed[i], b = mz + e;
                vf2 rb = vf2{__builtin_amdgcn_rcpf(b[0]), __builtin_amdgcn_rcpf(b[1])};
                rb = __builtin_elementwise_fma(__builtin_elementwise_fma(-b, rb, vf2{1.0f, 1.0f}), rb, rb);
                const vf2 nu = dt2p * H0 + 2.0f * mz * uc - mz * uold[i] + e * uc;
                const vf2 nv = dt2p * Hz + 2.0f * mz * vc - mz * vold[i] + e * vc;
                const vf2 qu = nu * rb, qv = nv * rb;
                un[i] = __builtin_elementwise_fma(__builtin_elementwise_fma(-b, qu, nu), rb, qu);
                vn[i] = __builtin_elementwise_fma(__builtin_elementwise_fma(-b, qv, nv), rb, qv);
            }
        }
        if (ninj > 0) {
            vf2 add[V];
            ld2(acc, tx * TZ + V * tl, tx * TZ + V * (tl + TL), add);
#pragma unroll
            for (int i = 0; i < V; ++i) un[i] += add[i], vn[i] += add[i];
        }
        if (__builtin_expect(ragged, 0)) {
            // points past the end of a row are halo: they stay zero in the registers as they do in memory
            const int n2 = q.a.n2;
#pragma unroll
            for (int i = 0; i < V; ++i)
#pragma unroll
                for (int h = 0; h < H; ++h)
                    if (zq[h] + i >= n2) un[i][h] = 0.f, vn[i][h] = 0.f;
        }
        vf2 ca[V], cb[V];   // backward loops: the stencilled combinations of the new levels, for the neighbours
#pragma unroll
        for (int i = 0; i < V; ++i) ca[i] = et[i] * un[i] + dl[i] * vn[i], cb[i] = dl[i] * un[i] + vn[i];
        if (!FWD) {
            // they are the next step's stencilled fields at this thread's points: nobody reads the image any more
            // in this step (its last readers finished before the barrier above)
            st2(sAB, ctr[0], ctr[1], ca);
            st2(sAB, ctr[0] + OFFB, ctr[1] + OFFB, cb);
        }
        if (__builtin_expect(!ragged, 1)) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                store_f4_held<kPlain>(vf4{un[0][h], un[1][h], un[2][h], un[3][h]}, rotuB, (int)pb[h], (int)out_off);
                store_f4_held<kPlain>(vf4{vn[0][h], vn[1][h], vn[2][h], vn[3][h]}, rotvB, (int)pb[h], (int)out_off);
                if (!FWD) {
                    store_f4_held<kPlain>(vf4{ca[0][h], ca[1][h], ca[2][h], ca[3][h]}, exaB, (int)pb[h], (int)ex_out);
                    store_f4_held<kPlain>(vf4{cb[0][h], cb[1][h], cb[2][h], cb[3][h]}, exbB, (int)pb[h], (int)ex_out);
                }
            }
        } else {
            const int n2 = q.a.n2;
#pragma unroll 1
            for (int h = 0; h < H; ++h) {
                const long e0 = (long)q.a.off0 + (long)x * q.a.s1 + zq[h];
                float *pu = q.rot_u + ((long)shot * 3 + (time + dir + 3) % 3) * q.field + e0;
                float *pv = q.rot_v + ((long)shot * 3 + (time + dir + 3) % 3) * q.field + e0;
                float *pa = FWD ? nullptr : q.comb_a + ((long)shot * 2 + ((s + 1) & 1)) * q.field + e0;
                float *pc = FWD ? nullptr : q.comb_b + ((long)shot * 2 + ((s + 1) & 1)) * q.field + e0;
                for (int i = 0; i < V; ++i)
                    if (zq[h] + i < n2) {
                        pu[i] = h ? un[i][1] : un[i][0], pv[i] = h ? vn[i][1] : vn[i][0];
                        if (!FWD) pa[i] = h ? ca[i][1] : ca[i][0], pc[i] = h ? cb[i][1] : cb[i][0];
                    }
            }
        }
        if (GRAD) {
            // summed by parts over time (see tti2d_kernel): the second time difference is taken of the adjoint fields
            vf2 su2[V], sv2[V];
            ld2(ssave, ((0 * H + 0) * NT + tid) * V, ((0 * H + 1) * NT + tid) * V, su2);
            ld2(ssave, ((1 * H + 0) * NT + tid) * V, ((1 * H + 1) * NT + tid) * V, sv2);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const vf2 ptt = (uold[i] - 2.0f * ucur[i] + un[i]) * a.inv_dt2;
                const vf2 rtt = (vold[i] - 2.0f * vcur[i] + vn[i]) * a.inv_dt2;
                gsum[i] += -(ptt * su2[i] + rtt * sv2[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) uold[i] = ucur[i], ucur[i] = un[i], vold[i] = vcur[i], vcur[i] = vn[i];

        // publish: this wave's stores have reached the L2, then every wave's, then the counter moves
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&done[tile], s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

        // write-only streams nobody waits for
        if ((fl & F_OWN) && q.a.vals_out) q.a.vals_out[((long)shot * q.a.nt + time) * q.a.itp_npoint + own_id] = sampled;
        if (q.save_u) {
            float *pu = q.save_u + (long)shot * q.save_shot_stride + (long)(time + 1) * q.field;
            float *pv = q.save_v + (long)shot * q.save_shot_stride + (long)(time + 1) * q.field;
            if (__builtin_expect(!ragged, 1)) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    store_f4_held<kStream>(vf4{ucur[0][h], ucur[1][h], ucur[2][h], ucur[3][h]}, tbuf_of(pu), (int)pb[h], 0);
                    store_f4_held<kStream>(vf4{vcur[0][h], vcur[1][h], vcur[2][h], vcur[3][h]}, tbuf_of(pv), (int)pb[h], 0);
                }
            } else {
#pragma unroll 1
                for (int h = 0; h < H; ++h) {
                    const long e0 = (long)q.a.off0 + (long)x * q.a.s1 + zq[h];
                    for (int i = 0; i < V; ++i)
                        if (zq[h] + i < q.a.n2) pu[e0 + i] = h ? ucur[i][1] : ucur[i][0], pv[e0 + i] = h ? vcur[i][1] : vcur[i][0];
                }
            }
        }
    }

    if (GRAD) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            if (!(flags & (h ? F_LIVE1 : F_LIVE0))) continue;
            float *gp = q.grad + (long)shot * q.g_shot_stride + at(x, zq[h]);
            for (int i = 0; i < V; ++i)
                if (zq[h] + i < a.n2) gp[i] = h ? gsum[i][1] : gsum[i][0];
        }
    }
}

constexpr int kPttiThreads = 128;   // two waves per 32 x 32 tile

template <int R>
static int occupancy_ptti(int mode)
{
    int nb = 0;
    hipError_t e = mode == 0   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, persist_tti2d_kernel<R, 0>, kPttiThreads, 0)
                   : mode == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, persist_tti2d_kernel<R, 1>, kPttiThreads, 0)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, persist_tti2d_kernel<R, 2>, kPttiThreads, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return nb;
}

// resident blocks per CU of the whole-loop TTI kernel (mode 0 forward, 1 adjoint, 2 gradient); 0: not available
int persist_tti2d_blocks_per_cu(int R, int mode)
{
    if (R == 2) return occupancy_ptti<2>(mode);
    if (R == 4) return occupancy_ptti<4>(mode);
    return 0;
}

template <int R>
static void launch_ptti_r(const PersistTtiArgs &q, int mode, int nblocks, hipStream_t stream)
{
    if (mode == 0) hipLaunchKernelGGL((persist_tti2d_kernel<R, 0>), dim3(nblocks), dim3(kPttiThreads), 0, stream, q);
    else if (mode == 1) hipLaunchKernelGGL((persist_tti2d_kernel<R, 1>), dim3(nblocks), dim3(kPttiThreads), 0, stream, q);
    else hipLaunchKernelGGL((persist_tti2d_kernel<R, 2>), dim3(nblocks), dim3(kPttiThreads), 0, stream, q);
}

int launch_persist_tti2d(const Plan &p, const PersistTtiArgs &q, int mode, int blocks_per_xcd, hipStream_t stream)
{
    const int nblocks = 8 * blocks_per_xcd;
    if (p.L.R == 2) launch_ptti_r<2>(q, mode, nblocks, stream);
    else if (p.L.R == 4) launch_ptti_r<4>(q, mode, nblocks, stream);
    else {
        set_error("whole-loop TTI launches exist for space orders 4 and 8");
        return MIFWI_EINVAL;
    }
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

}  // namespace mifwi
