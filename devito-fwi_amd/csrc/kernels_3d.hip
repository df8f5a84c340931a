// 2.5-D time-stepping kernel for 3-D grids (float32): the fast path for 3-D acquisitions.
//
// Layout reminder: grid (x, y, z), z contiguous, zero halos of width R around every dimension.
// A 256-thread block owns a TY x TZ tile of the (y, z) plane and marches along x over a chunk of
// planes.  Each thread owns 4 consecutive z points of one y row:
//   * y and z directions: the NEWEST plane of the own column (x + R when output plane x is finished) plus its
//     halo ring is staged in LDS (the centre from the registers, the ring with 16-byte loads that the
//     neighbouring tiles' blocks, which request the same plane for their own columns in the same
//     iteration, keep warm in L2) and read back with ds_read_b128: the z line once, 2R row neighbours;
//   * x direction: the R planes behind the staged one are gathered from registers into its sum, and its own
//     values are scattered into the R sums still open (R + 1 planes of values + R + 1 sums in registers;
//     one aligned 16-byte load per thread and plane feeds them).
// LDS is double-buffered so that a plane costs one barrier, and the loads of plane i+1 are issued
// before the arithmetic of plane i.  old / m / eta are read and the result is written with one
// aligned 16-byte access each: 20 bytes per point and step of compulsory traffic.
// Block ids are remapped so that each XCD (ids equal modulo 8) gets a contiguous run of tiles:
// halo rings are then mostly served by that XCD's own L2.
// Source/receiver injection uses a CSR of owned cells per (tile, plane); interpolation of the
// field being read runs in extra blocks at the start of the grid.  Equations: kernels_generic.hip.
#include "mifwi_internal.hpp"

namespace mifwi {

__device__ inline float fast_div3(float a, float b)
{
    // a / b without the IEEE division sequence: reciprocal, one Newton step on it, then one
    // residual correction of the quotient (q += r * (a - b q)), which leaves it within 1 ulp
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}

// V consecutive floats (V = 2 or 4) moved with one aligned 8- or 16-byte access.
// Addresses are a uniform (SGPR) base pointer plus a 32-bit per-lane byte offset, the cheapest
// global addressing mode; one field level stays below 4 GB (checked when the plan is created).
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));
template <int V>
struct VecT;
template <>
struct VecT<2> {
    typedef vf2 type;
};
template <>
struct VecT<4> {
    typedef vf4 type;
};

template <int V>
__device__ inline void ldv(const char *base, unsigned off, float (&d)[V])
{
    const typename VecT<V>::type v = *reinterpret_cast<const typename VecT<V>::type *>(base + off);
#pragma unroll
    for (int i = 0; i < V; ++i) d[i] = v[i];
}
// streaming accesses (each byte used once per step) are non-temporal so that they do not evict the
// planes of the stencilled field that neighbouring tiles re-read from L2
template <int V>
__device__ inline void ldv_stream(const char *base, unsigned off, float (&d)[V])
{
    const typename VecT<V>::type v =
        __builtin_nontemporal_load(reinterpret_cast<const typename VecT<V>::type *>(base + off));
#pragma unroll
    for (int i = 0; i < V; ++i) d[i] = v[i];
}
template <int V>
__device__ inline void stv_stream(char *base, unsigned off, const float (&d)[V])
{
    typename VecT<V>::type v;
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = d[i];
    __builtin_nontemporal_store(v, reinterpret_cast<typename VecT<V>::type *>(base + off));
}
// The same accesses through a buffer descriptor: the instruction takes the descriptor and a plane offset from
// scalar registers and a 32-bit byte offset per lane, so that no 64-bit address is formed per lane (flat
// global addressing costs one 64-bit add and a register pair per access when the base moves with the loop).
typedef int vi4 __attribute__((ext_vector_type(4)));
__device__ vf4 raw_buffer_load_f4(vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ vf2 raw_buffer_load_f2(vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ void raw_buffer_store_f4(vf4 v, vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
__device__ void raw_buffer_store_f2(vf2 v, vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
__device__ void raw_buffer_store_f1(float v, vi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
// Descriptor of the memory that starts at p (p is uniform over the block): no stride, and a range of 2 GB - 1
// bytes.  A lane whose offset is kDead lies outside it: its loads return zeros and its stores are dropped without
// touching memory (probed on the hardware, tools/micro/bufoob.hip).  Lanes that have nothing to load or store
// are switched off through their offset, not through a branch around the instruction, so that every plane
// issues the same sequence of memory instructions and the waits in front of their uses can count the younger
// ones that may stay in flight (a branch around a load makes the compiler wait for everything).
constexpr unsigned kDead = 0x80000000u;
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
    r.s.n = 0x7fffffffu;
    r.s.f = 0x00020000u;
    return r.v;
}
constexpr int kBufStream = 2;   // non-temporal
template <int V, int AUX = 0>
__device__ inline void bldv(vi4 b, unsigned soff, unsigned off, float (&d)[V])
{
    if (V == 4) {
        const vf4 v = raw_buffer_load_f4(b, (int)off, (int)soff, AUX);
#pragma unroll
        for (int i = 0; i < V; ++i) d[i] = v[i];
    } else {
        const vf2 v = raw_buffer_load_f2(b, (int)off, (int)soff, AUX);
#pragma unroll
        for (int i = 0; i < V; ++i) d[i] = v[i];
    }
}
// the points z + i < n2 of a thread's V, one predicated 4-byte store each (the last tile of a row when n2 is not a
// multiple of V; same descriptor and offsets as the vector store, so the common path keeps no pointer for it)
template <int V>
__device__ inline void bstv_ragged(vi4 b, unsigned soff, unsigned off, const float (&d)[V], int nlive)
{
#pragma unroll
    for (int i = 0; i < V; ++i)
        if (i < nlive) raw_buffer_store_f1(d[i], b, (int)(off + 4 * i), (int)soff, 0);
}
// The kernel's own argument block, for branches that are rarely taken: what they need is read from it where
// they run instead of being held in scalar registers across the march along x (the compiler cannot see that
// the pointer is the one it already has, so it does not hoist the loads).
typedef const Step3DArgs __attribute__((address_space(4))) *RareArgs;
__device__ inline RareArgs rare_args()
{
    unsigned long v = (unsigned long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(v));
    return (RareArgs)v;
}
template <int V>
__device__ inline void bstv_stream(vi4 b, unsigned soff, unsigned off, const float (&d)[V])
{
    if (V == 4) {
        vf4 v;
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = d[i];
        raw_buffer_store_f4(v, b, (int)off, (int)soff, kBufStream);
    } else {
        vf2 v;
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = d[i];
        raw_buffer_store_f2(v, b, (int)off, (int)soff, kBufStream);
    }
}
// V consecutive points of a half-precision saved level; `off` is their byte offset in a float field
typedef unsigned vu2 __attribute__((ext_vector_type(2)));
template <int V>
__device__ inline void ldv_half_stream(const char *base, unsigned off, float inv_scale, float (&d)[V])
{
    if (V == 4) {
        const vu2 v = __builtin_nontemporal_load(reinterpret_cast<const vu2 *>(base + (off >> 1)));
        unpack_half2(v.x, d[0], d[1]);
        unpack_half2(v.y, d[V - 2], d[V - 1]);
    } else {
        const unsigned v = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(base + (off >> 1)));
        unpack_half2(v, d[0], d[1]);
    }
#pragma unroll
    for (int i = 0; i < V; ++i) d[i] *= inv_scale;
}
template <int V>
__device__ inline void stv_half_stream(char *base, unsigned off, const float (&d)[V], float scale)
{
    if (V == 4) {
        vu2 v;
        v.x = pack_half2(d[0] * scale, d[1] * scale);
        v.y = pack_half2(d[V - 2] * scale, d[V - 1] * scale);
        __builtin_nontemporal_store(v, reinterpret_cast<vu2 *>(base + (off >> 1)));
    } else {
        __builtin_nontemporal_store(pack_half2(d[0] * scale, d[1] * scale), reinterpret_cast<unsigned *>(base + (off >> 1)));
    }
}
template <int V>
__device__ inline void lds_ldv(const float *p, float (&d)[V])
{
    const typename VecT<V>::type v = *reinterpret_cast<const typename VecT<V>::type *>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) d[i] = v[i];
}
template <int V>
__device__ inline void lds_stv(float *p, const float (&d)[V])
{
    typename VecT<V>::type v;
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = d[i];
    *reinterpret_cast<typename VecT<V>::type *>(p) = v;
}

// R: stencil radius.  TY x TZ: tile of the (y, z) plane.  V: points per thread (TY*TZ/V threads).
// Register budget: the propagation variants are held to 168 VGPRs (3 blocks of 256 threads per CU; they need
// 139 at radius 8); the gradient variants, which carry five more vectors (163 at radius 8), are launched for 2
// blocks per CU: a third measured no faster, they are bound by their bytes.
// GRAD: also accumulate the FWI gradient.  ISO: equal spacing in x, y and z (one coefficient set).
#ifndef MIFWI_3D_FWD_WAVES
#define MIFWI_3D_FWD_WAVES 3
#endif
#ifndef MIFWI_3D_GRAD_WAVES
#define MIFWI_3D_GRAD_WAVES 2
#endif
// MODE 0: plain propagation, 1: gradient, 2: scattered field of Born modelling, 3: plain propagation that
// also streams a half-precision copy of the new level into the saved array, 4: gradient of two steps
// (a.pair: the step before ran as MODE 0 and its terms are added here, see Step2DArgs::pair).
// The fourth-order-in-time kernel (Step2DArgs::pt, pre_mode, bih) has variants of its own, so that the common ones
// carry none of it: 5 = mode 0, 6 = mode 1, 7 = mode 2 with those arguments honoured.
template <int R, int TY, int TZ, int V, int XMODE, bool ISO>
__global__ __launch_bounds__(TY *TZ / V, (XMODE == 1 || XMODE == 2 || XMODE >= 4 ? MIFWI_3D_GRAD_WAVES : MIFWI_3D_FWD_WAVES) * (V == 2 ? 2 : 1)) void step3d_kernel(Step3DArgs a)
{
    constexpr bool OT4 = XMODE >= 5;
    constexpr int MODE = XMODE == 5 ? 0 : (XMODE == 6 ? 1 : (XMODE == 7 ? 2 : XMODE));
    constexpr bool GRAD = (MODE == 1 || MODE == 4), PAIR = (MODE == 4), BORN = (MODE == 2);
    constexpr int NT = TY * TZ / V;                     // threads per block
    constexpr int RZ = (R + 3) / 4 * 4;
    constexpr int RZ4 = RZ / 4;
    constexpr int ROWF4 = (TZ + 2 * RZ) / 4;
    constexpr int P = tile2d_lds_pitch(TZ, R);
    constexpr int TQ = TZ / V;                          // threads per tile row
    constexpr int NBAND = R * ROWF4;                    // float4 in the top (or bottom) halo band
    constexpr int NH = 2 * NBAND + 2 * TY * RZ4;        // float4 in the halo ring
    constexpr int NHT = (NH + NT - 1) / NT;             // ring float4 per thread
    __shared__ __attribute__((aligned(16))) float lds[2][(TY + 2 * R) * P];
    __shared__ __attribute__((aligned(16))) float acc[TY * TZ];

    const int tid = threadIdx.x;
    // interpolation blocks come first; their count is padded to a multiple of 8 so that the tile
    // blocks keep id % 8 == XCD label
    const int nitp_real = a.pt_idx ? ((a.itp_npoint + NT - 1) / NT) * a.nshots : 0;
    const int nitp_blocks = (nitp_real + 7) / 8 * 8;
    int bid = blockIdx.x;
    if (bid < nitp_blocks) {
        const int shot = bid % a.nshots;
        const int pnt = (bid / a.nshots) * NT + tid;
        if (pnt >= a.itp_npoint || bid >= nitp_real) return;
        const int tab = a.itp_per_shot ? shot : 0;
        const int32_t *idx = a.pt_idx + ((long)tab * a.itp_npoint + pnt) * 8;
        const float *w = a.pt_w + ((long)tab * a.itp_npoint + pnt) * 8;
        const float *f = (OT4 && a.pt ? a.pt : a.cur) + (long)shot * a.shot_stride;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) sum += w[c] * f[idx[c]];
        a.vals_out[((long)shot * a.nt + a.time) * a.itp_npoint + pnt] = sum;
        return;
    }
    bid -= nitp_blocks;
    // XCD-aware remap: ids equal modulo 8 share an XCD; give each a contiguous run of work items
    const int per_xcd = (a.nwork + 7) / 8;
    const int work = (bid % 8) * per_xcd + bid / 8;
    if (work >= a.nwork) return;
    int t = work;
    const int tile_z = t % a.ntz;
    t /= a.ntz;
    const int tile_y = t % a.nty;
    t /= a.nty;
    const int chunk = t % a.nchunk;
    const int shot = t / a.nchunk;
    const int tile = tile_y * a.ntz + tile_z;
    const int oa = a.x_lo + chunk * a.xchunk;   // this block's share of the planes of the launch
    const int ob = min(a.x_hi, oa + a.xchunk);
    const int y0 = tile_y * TY, z0 = tile_z * TZ;

    const int ty = tid / TQ, tq = tid - ty * TQ;
    const int y = y0 + ty, z = z0 + V * tq;
    const bool live = (y < a.n1) && (z < a.n2);
    const bool full = (z + V - 1 < a.n2);
    const int nlive = live ? min(V, a.n2 - z) : 0;   // points of this thread inside the grid
    const bool fs_rows = a.fs && z <= R;              // rows whose mirror image this thread keeps (free surface)
    // own column: byte offset of (plane 0, y, z) from a level base; planes advance by sb bytes
    const unsigned col = (unsigned)((a.off0 + (long)y * a.s1 + z) * 4);
    const unsigned col_l = live ? col : kDead;            // as a load offset
    const unsigned col_s = (live && full) ? col : kDead;  // as the offset of a V-point store
    const long sb = a.s0 * 4;
    const char *cur = reinterpret_cast<const char *>(a.cur + (long)shot * (OT4 && a.cur_shot_stride ? a.cur_shot_stride : a.shot_stride));
    const char *old = reinterpret_cast<const char *>(a.old + (long)shot * a.shot_stride);
    char *out = reinterpret_cast<char *>(a.out + (long)shot * (OT4 && a.out_shot_stride ? a.out_shot_stride : a.shot_stride));
    const char *mb = reinterpret_cast<const char *>(a.m);
    const char *eb = reinterpret_cast<const char *>(a.eta);
    // descriptors: the stencilled level starts R planes early so that plane offsets are never negative
    const unsigned sbu = (unsigned)sb;
    const vi4 curB = buf_of(cur - (long)R * sb), oldB = buf_of(old), outB = buf_of(out), mB = buf_of(mb), eB = buf_of(eb);

    // halo-ring pieces (float4) of this thread: LDS offset and byte offset from a plane base
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
            row = band ? (R + TY + r) : r;
        } else {
            const int hh = h - 2 * NBAND;
            const int r = hh / (2 * RZ4), cc = hh - r * (2 * RZ4);
            row = R + r;
            c4 = cc < RZ4 ? cc : (TZ / 4 + cc);
        }
        const int gy = y0 - R + row;
        h_ok[i] = (h < NH) && (gy < a.n1 + R);
        h_lds[i] = row * P + 4 * c4;
        h_glb[i] = h_ok[i] ? (unsigned)((a.off0 + (long)gy * a.s1 + (z0 - RZ + 4 * c4)) * 4) : kDead;
    }

    // Time loops (forward, and backward without illumination sums: where the backward field and its two neighbours in
    // time are zero a step's gradient term is zero, the squares of the forward field are not) keep, per shot, the box
    // of grid points (a range per axis) in which any of their steps so far has
    // produced a value other than zero (a.box_lo / a.box_hi: x, y, z ranges a.box_stride apart, updated below).  R
    // points further out the two levels this step reads are still exactly zero and so is its result, unless something
    // is injected there: those planes, and all planes of a tile the box does not reach, are written, not computed,
    // and the planes that can hold something are shared out anew among the chunks.  (A box read while other blocks
    // of the launch widen it is a wider one: still correct.)
    constexpr bool BOX = (XMODE == 0 || XMODE == 1 || XMODE == 3 || XMODE == 4);
    int xa = oa, xb = ob;
    if (BOX && a.box_lo) {
        const int bs = a.box_stride;
        int lo = min(a.box_lo[shot] - R, a.inj_lo[0]), hi = max(a.box_hi[shot] + R, a.inj_hi[0]);
        lo = max(lo, a.x_lo), hi = min(hi, a.x_hi);
        const int ylo = min(a.box_lo[bs + shot] - R, a.inj_lo[1]), yhi = max(a.box_hi[bs + shot] + R, a.inj_hi[1]);
        const int zlo = min(a.box_lo[2 * bs + shot] - R, a.inj_lo[2]), zhi = max(a.box_hi[2 * bs + shot] + R, a.inj_hi[2]);
        const bool tile_on = ylo < y0 + TY && yhi > y0 && zlo < z0 + TZ && zhi > z0;
        if (hi < lo || !tile_on) hi = lo;
        const int xc = (hi - lo + a.nchunk - 1) / a.nchunk;
        xa = min(hi, lo + chunk * xc), xb = min(hi, xa + xc);
        const float zero[V] = {};
        const bool ragged0 = !full && nlive > 0;
        for (int x = oa; (a.box_fill || MODE == 3) && x < ob; ++x) {
            if (x >= lo && x < hi) {
                x = hi - 1;
                continue;
            }
            if (a.box_fill) {   // (not into a level that was cleared when the loop began and has stayed zero here)
                bstv_stream<V>(outB, (unsigned)x * sbu, col_s, zero);
                if (__builtin_expect(ragged0, 0)) bstv_ragged<V>(outB, (unsigned)x * sbu, col, zero, nlive);
            }
            if (MODE == 3 && live) {
                char *cb = reinterpret_cast<char *>(a.copy16) + (((long)shot * a.copy_shot_stride * 4 + (long)x * sb) >> 1);
                if (full) {
                    stv_half_stream<V>(cb, col, zero, 1.0f);
                } else {
                    __half *ch = reinterpret_cast<__half *>(cb + (col >> 1));
                    for (int i = 0; i < V; ++i)
                        if (z + i < a.n2) ch[i] = __float2half_rn(0.0f);
                }
            }
        }
        if (xa >= xb) return;   // (block-uniform)
    }
    int nz_lo = 0x7fffffff, nz_hi = -0x7fffffff;   // planes in which this thread has produced a nonzero value

    // planes ef_lo <= x < ef_hi of this tile are free of the sponge: they skip its 4 bytes per point
    const int ef_lo = a.eta_free ? a.eta_free[2 * tile] : 0, ef_hi = a.eta_free ? a.eta_free[2 * tile + 1] : 0;

    // injection work of this block: CSR over (tile, plane), ranges for the whole chunk
    const int inj_tab = a.inj_per_shot ? shot : 0;
    const int32_t *cptr = a.cell_ptr ? a.cell_ptr + ((long)inj_tab * a.nty * a.ntz + tile) * (a.n0 + 1) : nullptr;
    const bool any_inj = cptr ? (cptr[xa] < cptr[xb]) : false;   // block-uniform

    // The x part of a plane's sum is split around the plane that is being staged.  In the iteration of output plane
    // x the plane STAGED in LDS is x + R, the newest one of the own column: its in-plane (y, z) part and the x
    // terms of the R planes behind it (gathered from the own values kept in registers) open the sum of output
    // x + R, and its own values are scattered into the R sums that are still open (outputs x .. x + R - 1); the
    // sum of output x is complete then.  The ring of a plane is thereby requested in the same iteration in which
    // the neighbouring tiles request that plane for their own columns, so that it is still in the XCD's L2
    // (staging the OUTPUT plane asked for it R planes later, when it was gone again).  A chunk takes R iterations
    // more: the first R only stage, the last R only finish.
    // Registers: R + 1 planes of own values and R + 1 open sums instead of a queue of 2R + 1 planes.
    constexpr int NP = V / 2;
    float pq[R + 1][V], qn[V];   // own values of planes x .. x+R (pq[0]: the output plane, pq[R]: the staged one); x+R+1 in flight
    vf2 pa[R + 1][NP];           // open sums of outputs x .. x+R
#pragma unroll
    for (int j = 0; j < R; ++j) bldv<V>(curB, (unsigned)(xa + j) * sbu, col_l, pq[j + 1]);   // planes xa-R .. xa-1
    bldv<V>(curB, (unsigned)(xa + R) * sbu, col_l, qn);                                       // plane xa
    float hv[NHT][4];
#pragma unroll
    for (int i = 0; i < NHT; ++i) bldv<4>(curB, (unsigned)(xa + R) * sbu, h_glb[i], hv[i]);
#pragma unroll
    for (int j = 0; j <= R; ++j) {
#pragma unroll
        for (int p = 0; p < NP; ++p) pa[j][p] = vf2{0.f, 0.f};
    }

    // The result of a plane is stored at the start of the NEXT plane, ahead of that plane's loads: a wait for a
    // load then never has a younger store in front of it (the compiler's waits let the youngest loads stay in
    // flight but count no store among them, so a store issued just before the wait would have to complete first).
    float res_p[V];
#pragma unroll
    for (int i = 0; i < V; ++i) res_p[i] = 0.f;
    unsigned soff_p = 0, col_p = kDead;

    int buf = 0;
    for (int x = xa - R; x < xb; ++x) {
        const bool fin = x >= xa;        // output plane x is completed and written in this iteration
        const bool stg = x + R < xb;     // plane x + R is staged: it is an output plane of this chunk
        bstv_stream<V>(outB, soff_p, col_p, res_p);
        col_p = kDead;
        // advance: plane x+R arrived during the previous iteration, x+R+1 is requested now; the sums move up by one
#pragma unroll
        for (int j = 0; j < R; ++j) {
#pragma unroll
            for (int i = 0; i < V; ++i) pq[j][i] = pq[j + 1][i];
#pragma unroll
            for (int p = 0; p < NP; ++p) pa[j][p] = pa[j + 1][p];
        }
#pragma unroll
        for (int i = 0; i < V; ++i) pq[R][i] = qn[i];
        if (x + 1 < xb) bldv<V>(curB, (unsigned)(x + 2 * R + 1) * sbu, col_l, qn);
        float(&cl)[V] = pq[R];
        float(&own)[V] = pq[0];
        // this plane's coefficients are only needed once the stencil sum is complete
        float mm[V], ee[V], oo[V];
#pragma unroll
        for (int i = 0; i < V; ++i) mm[i] = ee[i] = oo[i] = 0.f;
        if (fin) {
            bldv<V, kBufStream>(mB, (unsigned)x * sbu, col_l, mm);
            if (!(x >= ef_lo && x < ef_hi)) bldv<V, kBufStream>(eB, (unsigned)x * sbu, col_l, ee);
            bldv<V, kBufStream>(oldB, (unsigned)x * sbu, col_l, oo);
        }
        // fourth-order-in-time kernel (see Step2DArgs): the level itself when the stencilled field is z, and the
        // biharmonic term of the saved level in gradient launches (both block-uniform)
        float ptv[V], bhv[V];
#pragma unroll
        for (int i = 0; i < V; ++i) ptv[i] = bhv[i] = 0.f;
        if (OT4 && a.pt && fin) bldv<V>(buf_of(reinterpret_cast<const char *>(a.pt + (long)shot * a.shot_stride)), (unsigned)x * sbu, col_l, ptv);
        if (OT4 && GRAD && a.bih && fin)
            bldv<V, kBufStream>(buf_of(reinterpret_cast<const char *>(a.bih + (long)shot * a.g_shot_stride)), (unsigned)x * sbu, col_l, bhv);

        // accumulators and saved level(s) of this plane: issued with the coefficient loads, used last
        float u0[V], up[V], g[V], il[V], po[V];
        char *gradb = nullptr, *illumb = nullptr;
        if (GRAD) {
#pragma unroll
            for (int i = 0; i < V; ++i) u0[i] = up[i] = g[i] = il[i] = po[i] = 0.f;
        }
        if (GRAD && fin) {
            gradb = reinterpret_cast<char *>(a.grad + (long)shot * a.g_shot_stride) + (long)x * sb;
            illumb = a.illum ? reinterpret_cast<char *>(a.illum + (long)shot * a.g_shot_stride) + (long)x * sb : nullptr;
            if (a.save_scale) {   // half-precision saved levels (uniform branch)
                if (live) {
                    const long us = (long)shot * a.u_shot_stride * 4 + (long)x * sb;
                    const float inv = 1.0f / a.save_scale[shot];
                    ldv_half_stream<V>(reinterpret_cast<const char *>(a.uc) + (us >> 1), col, inv, u0);
                    if (PAIR || (a.first && illumb))
                        ldv_half_stream<V>(reinterpret_cast<const char *>(a.up) + (us >> 1), col, inv, up);
                }
            } else {
                const long ub = (long)shot * a.u_shot_stride * 4;
                bldv<V, kBufStream>(buf_of(reinterpret_cast<const char *>(a.uc) + ub), (unsigned)x * sbu, col_l, u0);
                if (PAIR || (a.first && illumb))
                    bldv<V, kBufStream>(buf_of(reinterpret_cast<const char *>(a.up) + ub), (unsigned)x * sbu, col_l, up);
            }
            if (PAIR) bldv<V, kBufStream>(outB, (unsigned)x * sbu, col_l, po);   // backward level time + 2, about to be overwritten
            bldv<V, kBufStream>(buf_of(gradb - (long)x * sb), (unsigned)x * sbu, col_l, g);
            if (illumb) bldv<V, kBufStream>(buf_of(illumb - (long)x * sb), (unsigned)x * sbu, col_l, il);
        }

        // stage plane x + R: centre from the own column, ring from the registers loaded one plane ago
        float *L = lds[buf];
        if (stg) {
            lds_stv<V>(&L[(ty + R) * P + RZ + V * tq], cl);
#pragma unroll
            for (int i = 0; i < NHT; ++i)
                if (tid + i * NT < NH) lds_stv<4>(&L[h_lds[i]], hv[i]);
        }

        // injected terms of the cells this tile owns in plane x (rare)
        int inj_beg = 0, inj_end = 0;
        if (__builtin_expect(any_inj && fin, 0)) {
            inj_beg = cptr[x];
            inj_end = cptr[x + 1];
            if (__builtin_expect(inj_beg < inj_end, 0)) {
                const auto *ra = rare_args();
                for (int j = tid; j < TY * TZ; j += NT) acc[j] = 0.f;
                __syncthreads();
                const float *w = ra->winj + ((long)shot * ra->nt + ra->time) * ra->winj_stride;
                const int32_t *cloc = ra->cell_local + (long)(ra->inj_per_shot ? shot : 0) * ra->inj_kmax;
                for (int j = inj_beg + tid; j < inj_end; j += NT) acc[cloc[j]] = w[j];
            }
        }
        if (stg || inj_beg < inj_end) __syncthreads();   // block-uniform

        // request the next plane's ring before the arithmetic
        if (x + R + 1 < xb) {
#pragma unroll
            for (int i = 0; i < NHT; ++i) bldv<4>(curB, (unsigned)(x + 1 + 2 * R) * sbu, h_glb[i], hv[i]);
        }
        float bn[V], bc[V], bo[V], bd[V];
        if (BORN) {
#pragma unroll
            for (int i = 0; i < V; ++i) bn[i] = bc[i] = bo[i] = bd[i] = 0.f;
            if (live && fin) {
                const long bs = (long)shot * a.b_shot_stride * 4 + (long)x * sb;
                ldv_stream<V>(reinterpret_cast<const char *>(a.b_un) + bs, col, bn);
                ldv_stream<V>(reinterpret_cast<const char *>(a.b_uc) + bs, col, bc);
                ldv_stream<V>(reinterpret_cast<const char *>(a.b_uo) + bs, col, bo);
                ldv_stream<V>(reinterpret_cast<const char *>(a.b_dm) + (long)x * sb, col, bd);
            }
        }
        float res[V], uc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) res[i] = 0.f, uc[i] = own[i];
        if (stg && live) {
            const float *ctr = &L[(ty + R) * P + RZ + V * tq];
            // The sums are formed for PAIRS of neighbouring outputs (packed two-float multiply-adds), and every
            // multiply-add gives both outputs of a pair the SAME coefficient: the z line is walked as pairs of
            // consecutive values (line[k], line[k+1]), each of which serves output pair p at distance
            // d = k - RZ - 2p.  (Broadcasting one value to a pair of outputs instead needs every ordered pair of
            // coefficients in scalar registers: about sixty of them at radius 8, most of them spilled and read
            // back for every plane.)  Values are folded into the sums as soon as they are read.
            vf2 lap2[NP];
            float line[V + 2 * RZ];
#pragma unroll
            for (int p = 0; p < NP; ++p) lap2[p] = vf2{0.f, 0.f};
#pragma unroll
            for (int s = 0; s < (V + 2 * RZ) / V; ++s) {
                float fv[V];
                lds_ldv<V>(ctr - RZ + V * s, fv);
#pragma unroll
                for (int e = 0; e < V; ++e) line[V * s + e] = fv[e];
#pragma unroll
                for (int k = V * s - 1; k <= V * s + V - 2; ++k) {
                    if (k < 0) continue;
                    const vf2 pk = vf2{line[k], line[k + 1]};
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const int d = k - RZ - 2 * p;        // neighbour distance for both outputs of pair p
                        const int ad = d < 0 ? -d : d;
                        if (ad >= 1 && ad <= R) lap2[p] = __builtin_elementwise_fma(vf2{a.cz[ad], a.cz[ad]}, pk, lap2[p]);
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) lap2[p] = __builtin_elementwise_fma(vf2{a.c0, a.c0}, vf2{cl[2 * p], cl[2 * p + 1]}, lap2[p]);
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                const float cyk = ISO ? a.cz[k] : a.cy[k];
                float lo[V], hi[V];
                lds_ldv<V>(ctr - k * P, lo);
                lds_ldv<V>(ctr + k * P, hi);
#pragma unroll
                for (int p = 0; p < NP; ++p) lap2[p] = __builtin_elementwise_fma(vf2{cyk, cyk}, vf2{lo[2 * p], lo[2 * p + 1]}, lap2[p]);
#pragma unroll
                for (int p = 0; p < NP; ++p) lap2[p] = __builtin_elementwise_fma(vf2{cyk, cyk}, vf2{hi[2 * p], hi[2 * p + 1]}, lap2[p]);
            }
            // x terms of the planes behind the staged one
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                const float cxk = ISO ? a.cz[k] : a.cx[k];
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    lap2[p] = __builtin_elementwise_fma(vf2{cxk, cxk}, vf2{pq[R - k][2 * p], pq[R - k][2 * p + 1]}, lap2[p]);
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) pa[R][p] = lap2[p];
        }
        // the staged plane's own values are the x term at distance k of the output k planes behind it
#pragma unroll
        for (int k = 1; k <= R; ++k) {
            const float cxk = ISO ? a.cz[k] : a.cx[k];
#pragma unroll
            for (int p = 0; p < NP; ++p)
                pa[R - k][p] = __builtin_elementwise_fma(vf2{cxk, cxk}, vf2{cl[2 * p], cl[2 * p + 1]}, pa[R - k][p]);
        }
        if (fin && live) {
            float lap[V];
#pragma unroll
            for (int i = 0; i < V; ++i) lap[i] = pa[0][i / 2][i % 2];
            if (OT4 && a.pt) {   // the stencilled field was z; the pointwise terms take the level itself
#pragma unroll
                for (int i = 0; i < V; ++i) uc[i] = ptv[i];
            }
            if (OT4 && MODE == 0 && a.pre_mode) {   // block-uniform: a field derived from `cur`, not a time step
#pragma unroll
                for (int i = 0; i < V; ++i)
                    res[i] = a.pre_mode == 1 ? uc[i] + fast_div3(a.pre_c * lap[i], mm[i])
                                             : (a.pre_mode == 2 ? fast_div3(lap[i], mm[i] * mm[i]) : lap[i]);
            } else {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float ed = ee[i] * a.dt;
                    res[i] = fast_div3(a.dt2 * lap[i] + 2.0f * mm[i] * uc[i] - mm[i] * oo[i] + ed * uc[i], mm[i] + ed);
                }
            }
            if (BORN) {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float utt = (bn[i] - 2.0f * bc[i] + bo[i]) * a.inv_dt2;
                    res[i] += fast_div3(a.dt2 * (-bd[i] * utt), mm[i] + ee[i] * a.dt);
                }
            }
            if (__builtin_expect(inj_beg < inj_end, 0)) {
#pragma unroll
                for (int i = 0; i < V; ++i) res[i] += acc[ty * TZ + V * tq + i];
            }
            if (GRAD) {
                if (PAIR) {
                    // terms of the step before (time + 1), added first: the sums keep their order in time
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        g[i] = grad_accumulate(g[i], po[i], oo[i], uc[i], up[i], a.inv_dt2);
                        il[i] = illum_accumulate(il[i], up[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    // correlation summed by parts over time (see kernels_tiled2d.hip): second time
                    // difference of the backward field, one saved forward level per step
                    g[i] = grad_accumulate(g[i], oo[i], uc[i], res[i], u0[i], a.inv_dt2);
                    if (!PAIR && a.first)
                        il[i] = illum_accumulate(illum_accumulate(il[i], up[i]), u0[i]);   // levels nt-1, nt-2
                    else
                        il[i] = illum_accumulate(il[i], u0[i]);
                }
                if (OT4 && a.bih) {
#pragma unroll
                    for (int i = 0; i < V; ++i) g[i] = __builtin_fmaf(-(a.pre_c * bhv[i]), uc[i], g[i]);
                }
            }
        }
        if (BOX && a.box_lo && fin) {
            unsigned bits = 0;
#pragma unroll
            for (int i = 0; i < V; ++i)
                if (i < nlive) bits |= __float_as_uint(res[i]);
            if (bits << 1) nz_lo = min(nz_lo, x), nz_hi = max(nz_hi, x + 1);
        }
        // results: one V-point store per thread (dropped for lanes outside the grid); a thread that straddles
        // the end of a row stores its points one by one
        const bool ragged = fin && !full && nlive > 0;
#pragma unroll
        for (int i = 0; i < V; ++i) res_p[i] = res[i];
        soff_p = fin ? (unsigned)x * sbu : 0u, col_p = fin ? col_s : kDead;
        if (__builtin_expect(ragged, 0)) bstv_ragged<V>(outB, (unsigned)x * sbu, col, res, nlive);
        if (MODE == 3 && live && fin) {
            // half-precision copy of the new level into the saved array
            const float sc = a.save_scale[shot];
            char *cb = reinterpret_cast<char *>(a.copy16) + (((long)shot * a.copy_shot_stride * 4 + (long)x * sb) >> 1);
            if (full) {
                stv_half_stream<V>(cb, col, res, sc);
            } else {
                __half *ch = reinterpret_cast<__half *>(cb + (col >> 1));
                for (int i = 0; i < V; ++i)
                    if (z + i < a.n2) ch[i] = __float2half_rn(fminf(fmaxf(res[i] * sc, -65504.0f), 65504.0f));
            }
        }
        if (__builtin_expect(fs_rows && live && fin, 0)) {
            // free surface: antisymmetric mirror of rows 1..R into the z halo (see kernels_tiled2d.hip)
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int zi = z + i;
                if (zi >= 1 && zi <= R && i < nlive)
                    raw_buffer_store_f1(-res[i], outB, (int)(col - 4u * (unsigned)(z + zi)), (int)((unsigned)x * sbu), 0);
            }
        }
        if (GRAD && fin) {
            bstv_stream<V>(buf_of(gradb - (long)x * sb), (unsigned)x * sbu, col_s, g);
            if (illumb) bstv_stream<V>(buf_of(illumb - (long)x * sb), (unsigned)x * sbu, col_s, il);
            if (__builtin_expect(ragged, 0)) {
                bstv_ragged<V>(buf_of(gradb - (long)x * sb), (unsigned)x * sbu, col, g, nlive);
                if (illumb) bstv_ragged<V>(buf_of(illumb - (long)x * sb), (unsigned)x * sbu, col, il, nlive);
            }
        }
        if (__builtin_expect(inj_beg < inj_end, 0)) __syncthreads();   // acc is rewritten by the next plane that injects
        buf ^= 1;
    }
    bstv_stream<V>(outB, soff_p, col_p, res_p);
    if (BOX && a.box_lo) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            nz_lo = min(nz_lo, __shfl_xor(nz_lo, o));
            nz_hi = max(nz_hi, __shfl_xor(nz_hi, o));
        }
        // a thread that has seen a value owns row y and points z .. z + nlive - 1
        const bool any = nz_lo < nz_hi;
        int ny_lo = any ? y : 0x7fffffff, ny_hi = any ? y + 1 : -0x7fffffff;
        int nq_lo = any ? z : 0x7fffffff, nq_hi = any ? z + nlive : -0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ny_lo = min(ny_lo, __shfl_xor(ny_lo, o)), ny_hi = max(ny_hi, __shfl_xor(ny_hi, o));
            nq_lo = min(nq_lo, __shfl_xor(nq_lo, o)), nq_hi = max(nq_hi, __shfl_xor(nq_hi, o));
        }
        if ((tid & 63) == 0 && nz_lo < nz_hi) {
            const int bs = a.box_stride;
            // only a wave that widens the box writes: thousands of atomics on one address take their turns in the L2
            // (measured: 18 ns each, five times the step of a 256^3 grid the wave has filled)
            const int mine_lo[3] = {nz_lo, ny_lo, nq_lo}, mine_hi[3] = {nz_hi, ny_hi, nq_hi};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                int *L = &a.box_lo[d * bs + shot], *H = &a.box_hi[d * bs + shot];
                if (mine_lo[d] < __hip_atomic_load(L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(L, mine_lo[d]);
                if (mine_hi[d] > __hip_atomic_load(H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(H, mine_hi[d]);
            }
        }
    }
}

// eta_zero[tile][x] = 1 when the sponge field vanishes on the whole (tile, plane) patch
__global__ void eta_zero_kernel(const float *eta, unsigned char *flags, int n0, int n1, int n2, long s0, long s1,
                                long off0, int ntz, int TY, int TZ)
{
    const int tile = blockIdx.x / n0, x = blockIdx.x - tile * n0;
    const int y0 = (tile / ntz) * TY, z0 = (tile % ntz) * TZ;
    int nonzero = 0;
    for (int i = threadIdx.x; i < TY * TZ; i += blockDim.x) {
        const int y = y0 + i / TZ, z = z0 + i % TZ;
        if (y < n1 && z < n2 && eta[off0 + (long)x * s0 + (long)y * s1 + z] != 0.f) nonzero = 1;
    }
    nonzero = __syncthreads_or(nonzero);
    if (threadIdx.x == 0) flags[blockIdx.x] = nonzero ? 0 : 1;
}

// free_range[tile] = the longest run lo <= x < hi of planes whose flag is set (one thread per tile)
__global__ void eta_free_kernel(const unsigned char *flags, int *free_range, int ntiles, int n0)
{
    const int tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= ntiles) return;
    int best_lo = 0, best_hi = 0, lo = 0;
    for (int x = 0; x <= n0; ++x) {
        if (x == n0 || !flags[(long)tile * n0 + x]) {
            if (x - lo > best_hi - best_lo) best_lo = lo, best_hi = x;
            lo = x + 1;
        }
    }
    free_range[2 * tile] = best_lo;
    free_range[2 * tile + 1] = best_hi;
}

int launch_eta_zero(const Plan &p, unsigned char *flags, int *free_range)
{
    const int ntiles = p.ntx * p.ntz;
    const int nblocks = ntiles * p.L.n[0];
    hipLaunchKernelGGL(eta_zero_kernel, dim3(nblocks), dim3(256), 0, p.stream, (const float *)p.eta, flags, p.L.n[0],
                       p.L.n[1], p.L.n[2], p.L.s0, p.L.s1, p.L.off0, p.ntz, p.tile_tx, p.tile_tz);
    MIFWI_HIP(hipGetLastError());
    hipLaunchKernelGGL(eta_free_kernel, dim3((ntiles + 63) / 64), dim3(64), 0, p.stream, (const unsigned char *)flags,
                       free_range, ntiles, p.L.n[0]);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

template <int R, int TY, int V, bool ISO>
static int launch_r3v(const Plan &p, const Step3DArgs &a, hipStream_t stream)
{
    constexpr int NT = TY * 64 / V;
    const int nitp = a.pt_idx ? (((a.itp_npoint + NT - 1) / NT) * a.nshots + 7) / 8 * 8 : 0;
    const int nblocks = (a.nwork + 7) / 8 * 8 + nitp;
    const bool ot4 = a.pt || a.pre_mode || a.bih;
    if (ot4 && V == 4) {
        if (a.um)
            hipLaunchKernelGGL((step3d_kernel<R, TY, 64, 4, 6, ISO>), dim3(nblocks), dim3(NT), 0, stream, a);
        else if (a.b_un)
            hipLaunchKernelGGL((step3d_kernel<R, TY, 64, 4, 7, ISO>), dim3(nblocks), dim3(NT), 0, stream, a);
        else
            hipLaunchKernelGGL((step3d_kernel<R, TY, 64, 4, 5, ISO>), dim3(nblocks), dim3(NT), 0, stream, a);
    } else if (ot4) {
        set_error("the fourth-order-in-time kernel runs with four points per thread");
        return MIFWI_EINVAL;
    } else if (a.um && a.pair)
        hipLaunchKernelGGL((step3d_kernel<R, TY, 64, V, 4, ISO>), dim3(nblocks), dim3(NT), 0, stream, a);
    else if (a.um)
        hipLaunchKernelGGL((step3d_kernel<R, TY, 64, V, 1, ISO>), dim3(nblocks), dim3(NT), 0, stream, a);
    else if (a.b_un)   // always four points per thread
        hipLaunchKernelGGL((step3d_kernel<R, TY, 64, 4, 2, ISO>), dim3((a.nwork + 7) / 8 * 8 +
                               (a.pt_idx ? (((a.itp_npoint + TY * 16 - 1) / (TY * 16)) * a.nshots + 7) / 8 * 8 : 0)),
                           dim3(TY * 16), 0, stream, a);
    else if (a.copy16)
        hipLaunchKernelGGL((step3d_kernel<R, TY, 64, V, 3, ISO>), dim3(nblocks), dim3(NT), 0, stream, a);
    else
        hipLaunchKernelGGL((step3d_kernel<R, TY, 64, V, 0, ISO>), dim3(nblocks), dim3(NT), 0, stream, a);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

template <int R, int TY>
static int occupancy_r3(bool grad)
{
    int nb = 0;
    hipError_t e = grad ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, step3d_kernel<R, TY, 64, 4, 1, true>, TY * 16, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, step3d_kernel<R, TY, 64, 4, 0, true>, TY * 16, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 2;
    }
    return nb < 1 ? 1 : nb;
}

// resident blocks per CU of the 2.5-D kernel for this radius (used to size the x chunks)
int step3d_blocks_per_cu(int R, bool grad)
{
    switch (R) {
        case 1: return occupancy_r3<1, 16>(grad);
        case 2: return occupancy_r3<2, 16>(grad);
        case 3: return occupancy_r3<3, 16>(grad);
        case 4: return occupancy_r3<4, 16>(grad);
        case 5: return occupancy_r3<5, 16>(grad);
        case 6: return occupancy_r3<6, 16>(grad);
        case 7: return occupancy_r3<7, 16>(grad);
        default: return occupancy_r3<8, 16>(grad);
    }
}

template <int R>
static int launch_r3(const Plan &p, const Step3DArgs &a, hipStream_t stream)
{
    // equal spacing in all three dimensions: one coefficient set instead of three (fewer SGPRs)
    bool iso = true;
    for (int k = 0; k <= R; ++k) iso = iso && a.cx[k] == a.cz[k] && a.cy[k] == a.cz[k];
    // four points per thread (256 threads); two points per thread (512 threads, twice the waves at half the
    // registers) measured slower at every order
    return iso ? launch_r3v<R, 16, 4, true>(p, a, stream) : launch_r3v<R, 16, 4, false>(p, a, stream);
}

int launch_step3d(const Plan &p, const Step3DArgs &a, hipStream_t stream)
{
    switch (p.L.R) {
        case 1: return launch_r3<1>(p, a, stream);
        case 2: return launch_r3<2>(p, a, stream);
        case 3: return launch_r3<3>(p, a, stream);
        case 4: return launch_r3<4>(p, a, stream);
        case 5: return launch_r3<5>(p, a, stream);
        case 6: return launch_r3<6>(p, a, stream);
        case 7: return launch_r3<7>(p, a, stream);
        case 8: return launch_r3<8>(p, a, stream);
    }
    set_error("unsupported stencil radius");
    return MIFWI_EINVAL;
}

}  // namespace mifwi
