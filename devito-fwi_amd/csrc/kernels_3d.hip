// 2.5-D time-stepping kernel for 3-D grids (float32): the fast path for 3-D acquisitions.
//
// Layout reminder: grid (x, y, z), z contiguous, zero halos of width R around every dimension.
// A 256-thread block owns a TY x TZ tile of the (y, z) plane and marches along x over a chunk of
// planes.  Each thread owns 4 consecutive z points of one y row:
//   * x direction: a register queue of the 2R+1 planes around the current one (one aligned 16-byte
//     load per thread and plane feeds it);
//   * y and z directions: the current plane of the tile plus its halo ring is staged in LDS (the
//     centre from the register queue, the ring with 16-byte loads that neighbouring tiles' blocks
//     keep warm in L2) and read back with ds_read_b128: the z line once, 2R row neighbours.
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
    float r = __builtin_amdgcn_rcpf(b);
    r = r * (2.0f - b * r);
    return a * r;
}

__device__ inline float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

template <int R, int TY, int TZ, bool GRAD>
__global__ __launch_bounds__(256) void step3d_kernel(Step3DArgs a)
{
    constexpr int RZ = (R + 3) / 4 * 4;
    constexpr int RZ4 = RZ / 4;
    constexpr int ROWF4 = (TZ + 2 * RZ) / 4;
    constexpr int P = tile2d_lds_pitch(TZ, R);
    constexpr int TQ = TZ / 4;
    constexpr int NBAND = R * ROWF4;                    // float4 in the top (or bottom) halo band
    constexpr int NH = 2 * NBAND + 2 * TY * RZ4;        // float4 in the halo ring
    constexpr int NHT = (NH + 255) / 256;               // ring float4 per thread
    static_assert(TQ * TY == 256, "one float4 per thread and plane");
    __shared__ __attribute__((aligned(16))) float lds[2][(TY + 2 * R) * P];
    __shared__ __attribute__((aligned(16))) float acc[TY * TZ];

    const int tid = threadIdx.x;
    // interpolation blocks come first; their count is padded to a multiple of 8 so that the tile
    // blocks keep id % 8 == XCD label
    const int nitp_blocks = a.pt_idx ? (((a.itp_npoint + 255) / 256) * a.nshots + 7) / 8 * 8 : 0;
    int bid = blockIdx.x;
    if (bid < nitp_blocks) {
        const int shot = bid % a.nshots;
        const int pnt = (bid / a.nshots) * 256 + tid;
        if (pnt >= a.itp_npoint || bid >= ((a.itp_npoint + 255) / 256) * a.nshots) return;
        const int tab = a.itp_per_shot ? shot : 0;
        const int32_t *idx = a.pt_idx + ((long)tab * a.itp_npoint + pnt) * 8;
        const float *w = a.pt_w + ((long)tab * a.itp_npoint + pnt) * 8;
        const float *f = a.cur + (long)shot * a.shot_stride;
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
    const int xa = chunk * a.xchunk;
    const int xb = min(a.n0, xa + a.xchunk);
    const int y0 = tile_y * TY, z0 = tile_z * TZ;

    const int ty = tid / TQ, tq = tid - ty * TQ;
    const int y = y0 + ty, z = z0 + 4 * tq;
    const bool live = (y < a.n1) && (z < a.n2);
    const bool full = (z + 3 < a.n2);
    const long col = a.off0 + (long)y * a.s1 + z;                 // own column, plane 0
    const float *cur = a.cur + (long)shot * a.shot_stride;
    const float *old = a.old + (long)shot * a.shot_stride;
    float *out = a.out + (long)shot * a.shot_stride;

    // halo-ring pieces of this thread: LDS offset and offset from the plane's tile origin
    int h_lds[NHT];
    long h_glb[NHT];
    bool h_ok[NHT];
#pragma unroll
    for (int i = 0; i < NHT; ++i) {
        const int h = tid + i * 256;
        int row, c4;
        if (h < 2 * NBAND) {
            const int band = h / NBAND, r = (h - band * NBAND) / ROWF4;
            c4 = h - band * NBAND - r * ROWF4;
            row = band ? (R + TY + r) : r;
        } else {
            const int hh = h - 2 * NBAND;
            const int r = hh / (2 * RZ4), cc = hh - r * (2 * RZ4);
            row = R + r;
            c4 = cc < RZ4 ? cc : (TQ + cc);
        }
        const int gy = y0 - R + row;
        h_ok[i] = (h < NH) && (gy < a.n1 + R);
        h_lds[i] = row * P + 4 * c4;
        h_glb[i] = a.off0 + (long)gy * a.s1 + (z0 - RZ + 4 * c4);
    }

    // injection work of this block: CSR over (tile, plane), ranges for the whole chunk
    const int inj_tab = a.inj_per_shot ? shot : 0;
    const int32_t *cptr = a.cell_ptr ? a.cell_ptr + ((long)inj_tab * a.nty * a.ntz + tile) * (a.n0 + 1) : nullptr;
    const bool any_inj = cptr ? (cptr[xa] < cptr[xb]) : false;   // block-uniform

    // register queue: planes i-R .. i+R of the own column; prime it for i = xa
    float4 q[2 * R + 1];
#pragma unroll
    for (int j = 0; j < 2 * R; ++j)
        q[j + 1] = live ? ld4(cur + col + (long)(xa - R + j) * a.s0) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 hv[NHT];
#pragma unroll
    for (int i = 0; i < NHT; ++i)
        hv[i] = h_ok[i] ? ld4(cur + h_glb[i] + (long)xa * a.s0) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 m4 = make_float4(1.f, 1.f, 1.f, 1.f), e4 = make_float4(0.f, 0.f, 0.f, 0.f), o4 = e4;
    if (live) {
        m4 = ld4(a.m + col + (long)xa * a.s0);
        e4 = ld4(a.eta + col + (long)xa * a.s0);
        o4 = ld4(old + col + (long)xa * a.s0);
    }

    int buf = 0;
    for (int x = xa; x < xb; ++x) {
        // shift the queue and fetch plane x+R of the own column
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) q[j] = q[j + 1];
        q[2 * R] = live ? ld4(cur + col + (long)(x + R) * a.s0) : make_float4(0.f, 0.f, 0.f, 0.f);

        // stage plane x: centre from the queue, ring from the registers loaded one plane ago
        float *L = lds[buf];
        *reinterpret_cast<float4 *>(&L[(ty + R) * P + RZ + 4 * tq]) = q[R];
#pragma unroll
        for (int i = 0; i < NHT; ++i)
            if (tid + i * 256 < NH) *reinterpret_cast<float4 *>(&L[h_lds[i]]) = hv[i];

        // injected terms of the cells this tile owns in plane x (rare)
        int inj_beg = 0, inj_end = 0;
        if (any_inj) {
            inj_beg = cptr[x];
            inj_end = cptr[x + 1];
            if (inj_beg < inj_end) {
                *reinterpret_cast<float4 *>(&acc[4 * tid]) = make_float4(0.f, 0.f, 0.f, 0.f);
                __syncthreads();
                const float *vals = a.vals_inj + ((long)shot * a.nt + a.time) * a.inj_npoint;
                const int32_t *uptr = a.uc_ptr + (long)inj_tab * (a.inj_kmax + 1);
                const int32_t *ep = a.e_point + (long)inj_tab * a.inj_kmax;
                const float *ew = a.e_w + (long)inj_tab * a.inj_kmax;
                for (int j = inj_beg + tid; j < inj_end; j += 256) {
                    const long cj = (long)inj_tab * a.inj_kmax + j;
                    const int local = a.cell_local[cj];
                    const int u = a.cell_uc[cj];
                    float sum = 0.f;
                    for (int e = uptr[u]; e < uptr[u + 1]; ++e) sum += ew[e] * vals[ep[e]];
                    const int ly = local / TZ, lz = local - ly * TZ;
                    const float mc = a.m[a.off0 + (long)x * a.s0 + (long)(y0 + ly) * a.s1 + (z0 + lz)];
                    acc[local] = sum * a.dt2 / mc;
                }
            }
        }
        __syncthreads();

        // keep this plane's coefficients, then issue the next plane's loads before the arithmetic
        const float4 mc4 = m4, ec4 = e4, oc4 = o4;
        float4 um4, u04, up4, gv, iv;
        long gs = 0;
        if (GRAD) {
            um4 = u04 = up4 = gv = iv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                const long qs = (long)shot * a.u_shot_stride + col + (long)x * a.s0;
                gs = (long)shot * a.g_shot_stride + col + (long)x * a.s0;
                um4 = ld4(a.um + qs);
                u04 = ld4(a.uc + qs);
                up4 = ld4(a.up + qs);
                gv = ld4(a.grad + gs);
                if (a.illum) iv = ld4(a.illum + gs);
            }
        }
        if (x + 1 < xb) {
#pragma unroll
            for (int i = 0; i < NHT; ++i)
                hv[i] = h_ok[i] ? ld4(cur + h_glb[i] + (long)(x + 1) * a.s0) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                m4 = ld4(a.m + col + (long)(x + 1) * a.s0);
                e4 = ld4(a.eta + col + (long)(x + 1) * a.s0);
                o4 = ld4(old + col + (long)(x + 1) * a.s0);
            }
        }

        if (live) {
            const float *ctr = &L[(ty + R) * P + RZ + 4 * tq];
            float f[4 + 2 * RZ];
#pragma unroll
            for (int s = 0; s < (4 + 2 * RZ) / 4; ++s) {
                const float4 v = *reinterpret_cast<const float4 *>(ctr - RZ + 4 * s);
                f[4 * s + 0] = v.x, f[4 * s + 1] = v.y, f[4 * s + 2] = v.z, f[4 * s + 3] = v.w;
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
                lap[0] += a.cy[k] * (lo.x + hi.x);
                lap[1] += a.cy[k] * (lo.y + hi.y);
                lap[2] += a.cy[k] * (lo.z + hi.z);
                lap[3] += a.cy[k] * (lo.w + hi.w);
            }
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                const float4 lo = q[R - k], hi = q[R + k];
                lap[0] += a.cx[k] * (lo.x + hi.x);
                lap[1] += a.cx[k] * (lo.y + hi.y);
                lap[2] += a.cx[k] * (lo.z + hi.z);
                lap[3] += a.cx[k] * (lo.w + hi.w);
            }
            const float mm[4] = {mc4.x, mc4.y, mc4.z, mc4.w}, ee[4] = {ec4.x, ec4.y, ec4.z, ec4.w},
                        oo[4] = {oc4.x, oc4.y, oc4.z, oc4.w};
            float res[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ed = ee[i] * a.dt;
                res[i] = fast_div3(a.dt2 * lap[i] + 2.0f * mm[i] * uc[i] - mm[i] * oo[i] + ed * uc[i], mm[i] + ed);
            }
            if (inj_beg < inj_end) {
                const float4 add = *reinterpret_cast<const float4 *>(&acc[4 * tid]);
                res[0] += add.x, res[1] += add.y, res[2] += add.z, res[3] += add.w;
            }
            const long ps = col + (long)x * a.s0;
            if (full) {
                *reinterpret_cast<float4 *>(out + ps) = make_float4(res[0], res[1], res[2], res[3]);
            } else {
                for (int i = 0; i < 4; ++i)
                    if (z + i < a.n2) out[ps + i] = res[i];
            }
            if (GRAD) {
                const float um[4] = {um4.x, um4.y, um4.z, um4.w}, u0[4] = {u04.x, u04.y, u04.z, u04.w},
                            up[4] = {up4.x, up4.y, up4.z, up4.w};
                float g[4] = {gv.x, gv.y, gv.z, gv.w}, il[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    g[i] += -((um[i] - 2.0f * u0[i] + up[i]) * a.inv_dt2) * uc[i];
                    il[i] += u0[i] * u0[i] + (a.first ? up[i] * up[i] : 0.0f);
                }
                if (full) {
                    *reinterpret_cast<float4 *>(a.grad + gs) = make_float4(g[0], g[1], g[2], g[3]);
                    if (a.illum) *reinterpret_cast<float4 *>(a.illum + gs) = make_float4(il[0], il[1], il[2], il[3]);
                } else {
                    for (int i = 0; i < 4; ++i)
                        if (z + i < a.n2) {
                            a.grad[gs + i] = g[i];
                            if (a.illum) a.illum[gs + i] = il[i];
                        }
                }
            }
        }
        if (inj_beg < inj_end) __syncthreads();   // acc is rewritten by the next plane that injects
        buf ^= 1;
    }
}

template <int R>
static int launch_r3(const Plan &p, const Step3DArgs &a, int nblocks, hipStream_t stream)
{
    if (a.um)
        hipLaunchKernelGGL((step3d_kernel<R, 16, 64, true>), dim3(nblocks), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((step3d_kernel<R, 16, 64, false>), dim3(nblocks), dim3(256), 0, stream, a);
    MIFWI_HIP(hipGetLastError());
    return MIFWI_OK;
}

int launch_step3d(const Plan &p, const Step3DArgs &a, hipStream_t stream)
{
    const int nblocks = (a.nwork + 7) / 8 * 8 + (a.pt_idx ? (((a.itp_npoint + 255) / 256) * a.nshots + 7) / 8 * 8 : 0);
    switch (p.L.R) {
        case 1: return launch_r3<1>(p, a, nblocks, stream);
        case 2: return launch_r3<2>(p, a, nblocks, stream);
        case 3: return launch_r3<3>(p, a, nblocks, stream);
        case 4: return launch_r3<4>(p, a, nblocks, stream);
        case 5: return launch_r3<5>(p, a, nblocks, stream);
        case 6: return launch_r3<6>(p, a, nblocks, stream);
        case 7: return launch_r3<7>(p, a, nblocks, stream);
        case 8: return launch_r3<8>(p, a, nblocks, stream);
    }
    set_error("unsupported stencil radius");
    return MIFWI_EINVAL;
}

}  // namespace mifwi
