// Whole time loops of small 2-D TTI batches in ONE launch (float32, 32 x 32 tiles, space orders 4 and 8, at most one
// shot per XCD): the two coupled fields of the 'centered' kernel of the reference (seismic/tti/operators.py:180-243,
// 430-481 forward, 484-527 adjoint, 590-639 Jacobian adjoint) on the schedule of kernels_persist2d.hip.
//
// Every tile of a shot keeps a 128-thread block resident for the whole loop; a thread owns two quads of four consecutive
// z points of one row (quads l and l + 4 of the row's eight: a wave's 16-byte LDS reads are then conflict free with a
// row pitch of 80 floats, which is exactly the two fields' rows side by side).  Two waves per tile leave 256 registers
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

    // state kept in registers for the whole loop
    const int time0 = q.forward ? 1 : a.nt - 2;
    const int dir = q.forward ? 1 : -1;
    float mm[H][V], ed[H][V], et[H][V], dl[H][V], ct[H][V], st[H][V], ucur[H][V], uold[H][V], vcur[H][V], vold[H][V], gsum[H][V];
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int i = 0; i < V; ++i) {
            mm[h][i] = 1.f, ed[h][i] = 0.f, et[h][i] = 1.f, dl[h][i] = 1.f, ct[h][i] = 1.f, st[h][i] = 0.f;
            ucur[h][i] = 0.f, uold[h][i] = 0.f, vcur[h][i] = 0.f, vold[h][i] = 0.f, gsum[h][i] = 0.f;
            if (x < a.n1 && zq[h] + i < a.n2) {   // (set-up only)
                const long e = at(x, zq[h] + i);
                mm[h][i] = a.m[e], ed[h][i] = a.eta[e] * a.dt, et[h][i] = q.et[e], dl[h][i] = q.dl[e], ct[h][i] = q.ct[e],
                st[h][i] = q.st[e];
                const long c = (long)(time0 % 3) * q.field + e, o = (long)((time0 - dir + 3) % 3) * q.field + e;
                ucur[h][i] = rotu[c], uold[h][i] = rotu[o], vcur[h][i] = rotv[c], vold[h][i] = rotv[o];
                if (GRAD) gsum[h][i] = q.grad[(long)shot * q.g_shot_stride + e];
            }
        }
    const char *su_base = GRAD ? reinterpret_cast<const char *>(q.su + (long)shot * q.s_shot_stride) : nullptr;
    const char *sv_base = GRAD ? reinterpret_cast<const char *>(q.sv + (long)shot * q.s_shot_stride) : nullptr;

    // first-derivative weights with 1 / h folded in (one rounding more per weight, two multiplications fewer per term)
    float a1x[P + 1], a1z[P + 1];
#pragma unroll
    for (int j = 0; j <= P; ++j) a1x[j] = q.a1[j] * q.ihx, a1z[j] = q.a1[j] * q.ihz;

    int ctr[H];   // this thread's quads in the image (A; B is OFFB further)
#pragma unroll
    for (int h = 0; h < H; ++h) ctr[h] = (tx + R) * PA + RZ + V * (tl + TL * h);
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
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int i = 0; i < V; ++i) asm volatile("" : "+v"(mm[h][i]), "+v"(ed[h][i]));

        float wj = 0.f;
        if (fl & F_FEED) wj = a.winj[((long)shot * a.nt + time) * a.winj_stride + inj_base + tid];
        // own points of the two stencilled fields: from registers (backward loops: the combinations formed at the
        // end of the previous step are in place already)
        if (FWD || s == 0) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                vf4 fa, fb;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    fa[i] = FWD ? ucur[h][i] : et[h][i] * ucur[h][i] + dl[h][i] * vcur[h][i];
                    fb[i] = FWD ? vcur[h][i] : dl[h][i] * ucur[h][i] + vcur[h][i];
                }
                *reinterpret_cast<vf4 *>(&sAB[ctr[h]]) = fa;
                *reinterpret_cast<vf4 *>(&sAB[ctr[h] + OFFB]) = fb;
            }
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
        float lap[H][V], cgA[H][V], cgB[H][V];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            __builtin_amdgcn_sched_barrier(0);   // one quad and one field at a time: the phases are long enough as they are
            {
                // field A: its line of 12 values serves the z part of the Laplacian and the z derivative, the rows above
                // and below the x parts
                float la[V + 2 * RZ];
#pragma unroll
                for (int k = 0; k < (V + 2 * RZ) / V; ++k) {
                    const vf4 ta = *reinterpret_cast<const vf4 *>(&sAB[ctr[h] - RZ + V * k]);
#pragma unroll
                    for (int i = 0; i < V; ++i) la[V * k + i] = ta[i];
                }
                float dx[V], dz[V];
#pragma unroll
                for (int i = 0; i < V; ++i) lap[h][i] = a.c0 * la[RZ + i], dx[i] = 0.f, dz[i] = 0.f;
#pragma unroll
                for (int j = 1; j <= R; ++j)
#pragma unroll
                    for (int i = 0; i < V; ++i) lap[h][i] += a.cz[j] * (la[RZ + i - j] + la[RZ + i + j]);
#pragma unroll
                for (int j = 1; j <= P; ++j)
#pragma unroll
                    for (int i = 0; i < V; ++i) dz[i] += a1z[j] * (la[RZ + i + j] - la[RZ + i - j]);
#pragma unroll
                for (int j = 1; j <= R; ++j) {
                    const vf4 lo = *reinterpret_cast<const vf4 *>(&sAB[ctr[h] - j * PA]);
                    const vf4 hi = *reinterpret_cast<const vf4 *>(&sAB[ctr[h] + j * PA]);
#pragma unroll
                    for (int i = 0; i < V; ++i) lap[h][i] += a.cx[j] * (lo[i] + hi[i]);
                    if (j <= P) {
#pragma unroll
                        for (int i = 0; i < V; ++i) dx[i] += a1x[j] * (hi[i] - lo[i]);
                    }
                }
                vf4 sg4;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float g = -(st[h][i] * dx[i] + ct[h][i] * dz[i]);
                    cgA[h][i] = ct[h][i] * g, sg4[i] = st[h][i] * g;
                }
                *reinterpret_cast<vf4 *>(&sgA[(tx + P) * PG + V * (tl + TL * h)]) = sg4;
                // the sums are complete HERE: left to itself the compiler moves them down to their use in the update and
                // keeps the eleven quads they are formed from alive until then (registers spilled)
#pragma unroll
                for (int i = 0; i < V; ++i) asm volatile("" : "+v"(lap[h][i]), "+v"(cgA[h][i]));
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                float lb[V + 2 * P];   // z - P .. z + 3 + P
                {
                    const vf2 l2 = *reinterpret_cast<const vf2 *>(&sAB[ctr[h] + OFFB - 2]);
                    const vf4 m4 = *reinterpret_cast<const vf4 *>(&sAB[ctr[h] + OFFB]);
                    const vf2 r2 = *reinterpret_cast<const vf2 *>(&sAB[ctr[h] + OFFB + V]);
                    if (P == 2) lb[0] = l2[0], lb[1] = l2[1], lb[V + 2] = r2[0], lb[V + 3] = r2[1];
                    else lb[0] = l2[1], lb[V + 1] = r2[0];
#pragma unroll
                    for (int i = 0; i < V; ++i) lb[P + i] = m4[i];
                }
                float dx[V], dz[V];
#pragma unroll
                for (int i = 0; i < V; ++i) dx[i] = 0.f, dz[i] = 0.f;
#pragma unroll
                for (int j = 1; j <= P; ++j) {
                    const vf4 lo = *reinterpret_cast<const vf4 *>(&sAB[ctr[h] + OFFB - j * PA]);
                    const vf4 hi = *reinterpret_cast<const vf4 *>(&sAB[ctr[h] + OFFB + j * PA]);
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        dx[i] += a1x[j] * (hi[i] - lo[i]);
                        dz[i] += a1z[j] * (lb[P + i + j] - lb[P + i - j]);
                    }
                }
                vf4 sg4;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float g = -(st[h][i] * dx[i] + ct[h][i] * dz[i]);
                    cgB[h][i] = ct[h][i] * g, sg4[i] = st[h][i] * g;
                }
                *reinterpret_cast<vf4 *>(&sgB[(tx + P) * PG + V * (tl + TL * h)]) = sg4;
#pragma unroll
                for (int i = 0; i < V; ++i) asm volatile("" : "+v"(cgB[h][i]));
            }
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
        // cos * Gz of the quads beside this thread's two.  A row of the tile is four consecutive lanes and the
        // permutations are cyclic within them: the quad left of quad l is the previous lane's first quad, except for
        // lane 0, where it lies outside the tile; the quad right of quad l is the next lane's first quad, except for
        // lane 3, where it is quad 4 = lane 0's second quad = the cyclically next lane's SECOND quad; likewise quad
        // l + 4 has the previous lane's second quad to its left (lane 0: quad 3 = lane 3's first quad) and the next
        // lane's second quad to its right (lane 3: outside the tile).  The two outside cases are filled in after the
        // barrier from what the column tasks left in LDS.
        __syncthreads();
        // the saved levels are in LDS (vmcnt(0); waited for here, before this step's stores join the queue)
        if (GRAD) __builtin_amdgcn_s_waitcnt(0x0f70);

        // ---- update of both fields -------------------------------------------------------------------------------------
        const bool ragged = (fl & F_RAGGED) != 0;   // a row of the grid ends inside one of this lane's quads (rare)
#pragma unroll
        for (int h = 0; h < H; ++h) {
            __builtin_amdgcn_sched_barrier(0);
            float eA[V + 2 * P], eB[V + 2 * P];   // the z line of cos * Gz through quad h
#pragma unroll
            for (int i = 0; i < V; ++i) eA[P + i] = cgA[h][i], eB[P + i] = cgB[h][i];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                // A row of the tile is four consecutive lanes, and the permutations are cyclic within them.  Left of
                // quad l lies the previous lane's first quad (lane 0: outside the tile, filled in below); right of it
                // the next lane's first quad, except for lane 3, where it is quad 4 = lane 0's second quad = the
                // cyclically next lane's SECOND quad.  Left of quad l + 4 lies the previous lane's second quad, except
                // for lane 0, where it is quad 3 = the cyclically previous lane's FIRST quad; right of it the next
                // lane's second quad (lane 3: outside the tile).
                if (h == 0) {
                    eA[i] = quad_rotate<true>(cgA[0][V - P + i]), eB[i] = quad_rotate<true>(cgB[0][V - P + i]);
                    const float na0 = quad_rotate<false>(cgA[0][i]), na1 = quad_rotate<false>(cgA[1][i]);
                    const float nv0 = quad_rotate<false>(cgB[0][i]), nv1 = quad_rotate<false>(cgB[1][i]);
                    eA[P + V + i] = (fl & F_RIGHT) ? na1 : na0, eB[P + V + i] = (fl & F_RIGHT) ? nv1 : nv0;
                } else {
                    const float pa0 = quad_rotate<true>(cgA[0][V - P + i]), pa1 = quad_rotate<true>(cgA[1][V - P + i]);
                    const float pv0 = quad_rotate<true>(cgB[0][V - P + i]), pv1 = quad_rotate<true>(cgB[1][V - P + i]);
                    eA[i] = (fl & F_LEFT) ? pa0 : pa1, eB[i] = (fl & F_LEFT) ? pv0 : pv1;
                    eA[P + V + i] = quad_rotate<false>(cgA[1][i]), eB[P + V + i] = quad_rotate<false>(cgB[1][i]);
                }
            }
            if (h == 0 && (fl & F_LEFT)) {
#pragma unroll
                for (int i = 0; i < P; ++i) eA[i] = zh[0][0][tx][i], eB[i] = zh[1][0][tx][i];
            }
            if (h == 1 && (fl & F_RIGHT)) {
#pragma unroll
                for (int i = 0; i < P; ++i) eA[P + V + i] = zh[0][1][tx][i], eB[P + V + i] = zh[1][1][tx][i];
            }
            float un[V], vn[V];
            const int gq = (tx + P) * PG + V * (tl + TL * h);
            float sa[V], sb[V];
#pragma unroll
            for (int i = 0; i < V; ++i) sa[i] = 0.f, sb[i] = 0.f;
#pragma unroll
            for (int j = 1; j <= P; ++j) {
                const vf4 alo = *reinterpret_cast<const vf4 *>(&sgA[gq - j * PG]);
                const vf4 ahi = *reinterpret_cast<const vf4 *>(&sgA[gq + j * PG]);
                const vf4 blo = *reinterpret_cast<const vf4 *>(&sgB[gq - j * PG]);
                const vf4 bhi = *reinterpret_cast<const vf4 *>(&sgB[gq + j * PG]);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    sa[i] += a1z[j] * (eA[P + i + j] - eA[P + i - j]);
                    sa[i] += a1x[j] * (ahi[i] - alo[i]);
                    sb[i] += a1z[j] * (eB[P + i + j] - eB[P + i - j]);
                    sb[i] += a1x[j] * (bhi[i] - blo[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float X = lap[h][i] + sa[i], Y = -sb[i];   // Gxx = laplace - Gzz with Gzz = -sa
                const float H0 = FWD ? et[h][i] * X + dl[h][i] * Y : X;
                const float Hz = FWD ? dl[h][i] * X + Y : Y;
                const float uc = ucur[h][i], vc = vcur[h][i], mz = mm[h][i], e = ed[h][i], b = mz + e, rb = tti_rcp(b);
                un[i] = tti_div(dt2l[h] * H0 + 2.0f * mz * uc - mz * uold[h][i] + e * uc, b, rb);
                vn[i] = tti_div(dt2l[h] * Hz + 2.0f * mz * vc - mz * vold[h][i] + e * vc, b, rb);
            }
            if (ninj > 0) {
                const vf4 add = *reinterpret_cast<const vf4 *>(&acc[tx * TZ + V * (tl + TL * h)]);
#pragma unroll
                for (int i = 0; i < V; ++i) un[i] += add[i], vn[i] += add[i];
            }
            if (__builtin_expect(ragged, 0)) {
                // points past the end of a row are halo: they stay zero in the registers as they do in memory
                const int n2 = q.a.n2;
#pragma unroll
                for (int i = 0; i < V; ++i)
                    if (zq[h] + i >= n2) un[i] = 0.f, vn[i] = 0.f;
            }
            vf4 ca, cb;   // backward loops: the stencilled combinations of the new levels, for the neighbours
#pragma unroll
            for (int i = 0; i < V; ++i) ca[i] = et[h][i] * un[i] + dl[h][i] * vn[i], cb[i] = dl[h][i] * un[i] + vn[i];
            if (!FWD) {
                // they are the next step's stencilled fields at this thread's points: nobody reads the image any more
                // in this step (its last readers finished before the barrier above)
                *reinterpret_cast<vf4 *>(&sAB[ctr[h]]) = ca;
                *reinterpret_cast<vf4 *>(&sAB[ctr[h] + OFFB]) = cb;
            }
            if (__builtin_expect(!ragged, 1)) {
                store_f4_held<kPlain>(vf4{un[0], un[1], un[2], un[3]}, rotuB, (int)pb[h], (int)out_off);
                store_f4_held<kPlain>(vf4{vn[0], vn[1], vn[2], vn[3]}, rotvB, (int)pb[h], (int)out_off);
                if (!FWD) {
                    store_f4_held<kPlain>(ca, exaB, (int)pb[h], (int)ex_out);
                    store_f4_held<kPlain>(cb, exbB, (int)pb[h], (int)ex_out);
                }
            } else {
                const int n2 = q.a.n2;
                const long e0 = (long)q.a.off0 + (long)x * q.a.s1 + zq[h];
                float *pu = q.rot_u + ((long)shot * 3 + (time + dir + 3) % 3) * q.field + e0;
                float *pv = q.rot_v + ((long)shot * 3 + (time + dir + 3) % 3) * q.field + e0;
                float *pa = FWD ? nullptr : q.comb_a + ((long)shot * 2 + ((s + 1) & 1)) * q.field + e0;
                float *pc = FWD ? nullptr : q.comb_b + ((long)shot * 2 + ((s + 1) & 1)) * q.field + e0;
#pragma unroll
                for (int i = 0; i < V; ++i)
                    if (zq[h] + i < n2) {
                        pu[i] = un[i], pv[i] = vn[i];
                        if (!FWD) pa[i] = ca[i], pc[i] = cb[i];
                    }
            }
            if (GRAD) {
                // summed by parts over time (see tti2d_kernel): the second time difference is taken of the adjoint fields
                const vf4 su4 = *reinterpret_cast<const vf4 *>(&ssave[((0 * H + h) * NT + tid) * V]);
                const vf4 sv4 = *reinterpret_cast<const vf4 *>(&ssave[((1 * H + h) * NT + tid) * V]);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float ptt = (uold[h][i] - 2.0f * ucur[h][i] + un[i]) * a.inv_dt2;
                    const float rtt = (vold[h][i] - 2.0f * vcur[h][i] + vn[i]) * a.inv_dt2;
                    gsum[h][i] += -(ptt * su4[i] + rtt * sv4[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < V; ++i) uold[h][i] = ucur[h][i], ucur[h][i] = un[i], vold[h][i] = vcur[h][i], vcur[h][i] = vn[i];
        }

        // publish: this wave's stores have reached the L2, then every wave's, then the counter moves
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&done[tile], s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

        // write-only streams nobody waits for
        if ((fl & F_OWN) && q.a.vals_out) q.a.vals_out[((long)shot * q.a.nt + time) * q.a.itp_npoint + own_id] = sampled;
        if (q.save_u) {
            float *pu = q.save_u + (long)shot * q.save_shot_stride + (long)(time + 1) * q.field;
            float *pv = q.save_v + (long)shot * q.save_shot_stride + (long)(time + 1) * q.field;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                if (__builtin_expect(!ragged, 1)) {
                    store_f4_held<kStream>(vf4{ucur[h][0], ucur[h][1], ucur[h][2], ucur[h][3]}, tbuf_of(pu), (int)pb[h], 0);
                    store_f4_held<kStream>(vf4{vcur[h][0], vcur[h][1], vcur[h][2], vcur[h][3]}, tbuf_of(pv), (int)pb[h], 0);
                } else {
                    const long e0 = (long)q.a.off0 + (long)x * q.a.s1 + zq[h];
                    for (int i = 0; i < V; ++i)
                        if (zq[h] + i < q.a.n2) pu[e0 + i] = ucur[h][i], pv[e0 + i] = vcur[h][i];
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
                if (zq[h] + i < a.n2) gp[i] = gsum[h][i];
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
