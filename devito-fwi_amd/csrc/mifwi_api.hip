// Host side of libmifwi: the extern "C" entry points of include/mifwi.h, plan/device-memory
// management, sparse geometry tables and the time loops that drive the HIP kernels.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "mifwi_internal.hpp"

namespace mifwi {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }

static int fail(int code, const std::string &msg)
{
    set_error(msg);
    return code;
}

// Centred 2nd-derivative weights on -R..R in closed form:
//   w_k = 2 (-1)^(k+1) (R!)^2 / (k^2 (R-k)! (R+k)!),  w_0 = -2 sum_k w_k
// (the values of sympy.finite_diff_weights(2, range(-R,R+1), 0)[-1][-1] that Devito's `.laplace`
// uses, seismic/acoustic/operators.py:56).
static void fd2_weights(int R, double *w)
{
    double s = 0.0;
    for (int k = 1; k <= R; ++k) {
        // (R!)^2 / ((R-k)! (R+k)!) = prod_{j=1..k} (R-k+j)/(R+j)
        double q = 1.0;
        for (int j = 1; j <= k; ++j) q *= (double)(R - k + j) / (double)(R + j);
        w[k] = ((k & 1) ? 2.0 : -2.0) * q / ((double)k * (double)k);
        s += w[k];
    }
    w[0] = -2.0 * s;
}

static long round_up(long v, long m) { return (v + m - 1) / m * m; }

static int make_layout(const mifwi_config &c, Layout &L)
{
    L = Layout{};
    L.ndim = c.ndim;
    L.R = c.space_order / 2;
    for (int d = 0; d < c.ndim; ++d) {
        const int k = 3 - c.ndim + d;
        L.n[k] = c.shape[d];
        L.act[k] = 1;
        L.halo[k] = L.R;
    }
    L.zl = (int)round_up(L.R, 4);
    const long pitch = round_up((long)L.zl + round_up(L.n[2], 4) + L.zl, 32);
    L.s1 = pitch;
    L.s0 = pitch * (L.n[1] + 2 * L.halo[1]);
    const long body = L.s0 * (L.n[0] + 2 * L.halo[0]);
    const long first = (long)L.halo[0] * L.s0 + (long)L.halo[1] * L.s1 + L.zl;  // unshifted index of (0,0,0)
    const long shift = (32 - first % 32) % 32;
    L.off0 = first + shift;
    // slack so that tile staging reads that run past the end of the last row (into what is the
    // zero left halo of the next row everywhere else) stay inside the allocation
    L.field = round_up(body + shift + 256, 32);
    L.ngrid = (long)L.n[0] * L.n[1] * L.n[2];
    if (L.field > 0x7fffffffL) return -1;  // sparse tables use 32-bit offsets
    return 0;
}

template <typename T>
static StepConsts<T> make_consts(const Plan &p)
{
    StepConsts<T> k{};
    double w[kMaxRUntiled + 1];
    fd2_weights(p.L.R, w);
    double c0 = 0.0;
    for (int d = 0; d < p.cfg.ndim; ++d) {
        const int kd = 3 - p.cfg.ndim + d;
        const double ih2 = 1.0 / (p.cfg.spacing[d] * p.cfg.spacing[d]);
        c0 += w[0] * ih2;
        for (int j = 0; j <= p.L.R; ++j) k.c[kd][j] = (T)(w[j] * ih2);
    }
    k.c0 = (T)c0;
    k.dt = (T)p.cfg.dt;
    k.dt2 = k.dt * k.dt;
    k.inv_dt2 = T(1) / k.dt2;
    k.n0 = p.L.n[0];
    k.n1 = p.L.n[1];
    k.n2 = p.L.n[2];
    k.s0 = p.L.s0;
    k.s1 = p.L.s1;
    k.off0 = p.L.off0;
    k.field = p.L.field;
    k.fs = (p.cfg.flags & MIFWI_FLAG_FREE_SURFACE) ? 1 : 0;
    return k;
}

// ---- device memory helpers ------------------------------------------------------------------
static int dev_alloc(Plan &p, void **ptr, size_t bytes, bool zero)
{
    *ptr = nullptr;
    if (bytes == 0) bytes = 256;
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc of ") + std::to_string(bytes) + " bytes: " + hipGetErrorString(e));
        (void)hipGetLastError();
        return MIFWI_ENOMEM;
    }
    p.device_bytes += (int64_t)bytes;
    if (zero) MIFWI_HIP(hipMemsetAsync(*ptr, 0, bytes, p.stream));
    return MIFWI_OK;
}
static void dev_free(void *ptr)
{
    if (ptr) (void)hipFree(ptr);
}

static void free_sparse(SparseSet &s)
{
    dev_free(s.pt_idx);
    dev_free(s.pt_w);
    dev_free(s.uc_cell);
    dev_free(s.uc_ptr);
    dev_free(s.uc_count);
    dev_free(s.e_point);
    dev_free(s.e_w);
    dev_free(s.slot_count);
    dev_free(s.slot_local);
    dev_free(s.slot_ep);
    dev_free(s.slot_ew);
    dev_free(s.slot_base);
    dev_free(s.g_u);
    dev_free(s.cell_ptr);
    dev_free(s.cell_local);
    dev_free(s.cell_uc);
    s = SparseSet{};
}

static void clear_graphs(Plan &p)
{
    for (auto &kv : p.graphs) (void)hipGraphExecDestroy(kv.second.first);
    p.graphs.clear();
}

static void destroy(Plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    clear_graphs(*p);
    dev_free(p->m);
    dev_free(p->eta);
    dev_free(p->rot_u);
    dev_free(p->rot_v);
    dev_free(p->saved);
    dev_free(p->ckpt);
    dev_free(p->d_box);
    dev_free(p->d_ck_box);
    dev_free(p->d_boxv);
    dev_free(p->seg);
    free_sparse(p->src);
    free_sparse(p->rec);
    dev_free(p->d_wavelet);
    dev_free(p->d_wavelet_ck);
    dev_free(p->d_rec);
    dev_free(p->d_res);
    dev_free(p->d_srca);
    dev_free(p->d_grad);
    dev_free(p->d_illum);
    dev_free(p->d_dm);
    dev_free(p->pml_phi);
    dev_free(p->pml_zeta);
    dev_free(p->pml_c2s);
    dev_free(p->d_winj[0]);
    dev_free(p->d_winj[1]);
    dev_free(p->d_psync);
    if (p->h_abort) (void)hipHostFree(p->h_abort);
    dev_free(p->tti_par);
    dev_free(p->tti_gz3);
    dev_free(p->tti_w);
    dev_free(p->tti_fwd);
    dev_free(p->tti_adj);
    dev_free(p->tti_tmp);
    dev_free(p->tti_save);
    dev_free(p->tti_ckpt);
    dev_free(p->tti_seg);
    dev_free(p->d_save_scale);
    dev_free(p->d_t4);
    dev_free(p->d_w2);
    dev_free(p->d_bih);
    dev_free(p->d_dense);
    dev_free(p->d_partial);
    dev_free(p->d_fval);
    dev_free(p->d_reduce);
    dev_free(p->eta_zero);
    dev_free(p->eta_free);
    for (auto &pair : p->ev)
        for (hipEvent_t e : pair)
            if (e) (void)hipEventDestroy(e);
    if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

static int finish(Plan &p)
{
    if (!(p.cfg.flags & MIFWI_FLAG_ASYNC)) MIFWI_HIP(hipStreamSynchronize(p.stream));
    return MIFWI_OK;
}

// copy `bytes` between any two of {host, device} pointers on the plan's stream
static int copy_any(Plan &p, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return MIFWI_OK;
    MIFWI_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, p.stream));
    return MIFWI_OK;
}

static bool is_device_ptr(const void *ptr)
{
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

// device ordinal that owns a device pointer, -1 for anything else
static int device_of_ptr(const void *ptr)
{
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return a.type == hipMemoryTypeDevice ? a.device : -1;
}

// ---- the entry point that allocates nothing ------------------------------------------------------
// Partial sums of mifwi_l2_misfit_noalloc live in this array instead of in memory from hipMalloc.  Every GPU
// the library runs on holds its own copy, at its own address, until the process ends (128 KiB each).  In host
// code the array's name is only a handle for hipGetSymbolAddress: the address kernels are given is looked up
// once per device, with that device current, and kept in g_scratch_ptr.
constexpr int kNoAllocMaxShots = 256;
constexpr int kMaxDevices = 64;
static __device__ __align__(256) double g_misfit_scratch[kNoAllocMaxShots * kMisfitPartials];
static std::mutex g_scratch_mutex;
static double *g_scratch_ptr[kMaxDevices] = {};

// device address of g_misfit_scratch on device `dev`, which must be the current device
static int scratch_of_current_device(int dev, double **out)
{
    if (dev < 0 || dev >= kMaxDevices) return fail(MIFWI_EINVAL, "device ordinal beyond the library's table");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        MIFWI_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_misfit_scratch)));
        if (!ptr) return fail(MIFWI_EHIP, "hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = static_cast<double *>(ptr);
    }
    *out = g_scratch_ptr[dev];
    return MIFWI_OK;
}

// ---- sparse tables --------------------------------------------------------------------------
// Multilinear support of one point, in the arithmetic of the code Devito generates for
// SparseTimeFunction.inject/interpolate (seismic/self_adjoint/sa_01_iso_implementation1.ipynb:
// 1255-1316 of the reference): i = floor((x - o)/h), p = x - o - h*i, weights prod (1-p/h | p/h),
// all in the grid dtype.  Corners outside the padded grid only ever meet zero field values (halo)
// and zero velocity, so they are dropped here.
template <typename T>
static int build_sparse(Plan &p, SparseSet &S, const T *coords /* [tables][npoint][ndim] host */, int npoint,
                        int tables, int per_shot)
{
    free_sparse(S);
    const Layout &L = p.L;
    const int nd = L.ndim, nc = 1 << nd;
    S.npoint = npoint;
    S.ncorner = nc;
    S.per_shot = per_shot;
    S.nshots = tables;
    S.kmax = std::max(1, npoint * nc);
    std::vector<int32_t> pt_idx((size_t)tables * npoint * nc, (int32_t)L.off0);
    std::vector<T> pt_w((size_t)tables * npoint * nc, T(0));
    std::vector<int32_t> uc_cell((size_t)tables * S.kmax, (int32_t)L.off0), uc_ptr((size_t)tables * (S.kmax + 1), 0),
        uc_count(tables, 0), e_point((size_t)tables * S.kmax, 0);
    std::vector<T> e_w((size_t)tables * S.kmax, T(0));
    const int ntiles = p.kind == 1 ? p.ntx * p.ntz : 0;
    S.ntiles = ntiles;
    const int ntiles3 = p.kind == 2 ? p.ntx * p.ntz : 0;
    std::vector<int32_t> cell_ptr((size_t)tables * ntiles3 * (L.n[0] + 1), 0), cell_local, cell_uc;
    if (ntiles3) {
        cell_local.assign((size_t)tables * S.kmax, 0);
        cell_uc.assign((size_t)tables * S.kmax, 0);
    }
    // per table and tile: (tile-local cell, [(point, weight)...]) lists, flattened after the loop
    std::vector<std::vector<std::vector<std::pair<int32_t, std::vector<std::pair<int32_t, T>>>>>> tiles(
        ntiles ? tables : 0);
    std::vector<std::vector<std::vector<int32_t>>> tile_u(ntiles ? tables : 0);  // unique-cell index per (table, tile, slot)
    struct Entry {
        int32_t cell;
        int32_t point;
        int32_t corner;
        T w;
    };
    std::vector<Entry> ent;
    S.max_count = 0;
    for (int d = 0; d < 3; ++d) S.c_lo[d] = 0x7fffffff, S.c_hi[d] = -0x7fffffff;
    for (int t = 0; t < tables; ++t) {
        ent.clear();
        for (int q = 0; q < npoint; ++q) {
            int i[3] = {0, 0, 0};
            T f[3] = {0, 0, 0};
            for (int d = 0; d < nd; ++d) {
                const int k = 3 - nd + d;
                const T h = (T)p.cfg.spacing[d], o = (T)p.cfg.origin[d];
                const T pos = coords[((size_t)t * npoint + q) * nd + d] - o;
                const T fl = std::floor(pos / h);
                if (!(std::fabs((double)fl) < 1e9)) return fail(MIFWI_EINVAL, "non-finite or absurd sparse coordinate");
                i[k] = (int)fl;
                f[k] = (pos - h * fl) / h;
            }
            for (int b = 0; b < nc; ++b) {
                int j[3] = {0, 0, 0};
                T w = 1;
                bool ok = true;
                for (int d = 0; d < nd; ++d) {
                    const int k = 3 - nd + d;
                    const int up = (b >> (nd - 1 - d)) & 1;
                    j[k] = i[k] + up;
                    w *= up ? f[k] : (T(1) - f[k]);
                    if (j[k] < 0 || j[k] >= L.n[k]) ok = false;
                }
                if (!ok) continue;
                const long cell = L.off0 + (long)j[0] * L.s0 + (long)j[1] * L.s1 + j[2];
                pt_idx[((size_t)t * npoint + q) * nc + b] = (int32_t)cell;
                pt_w[((size_t)t * npoint + q) * nc + b] = w;
                ent.push_back(Entry{(int32_t)cell, q, b, w});
            }
        }
        std::sort(ent.begin(), ent.end(), [](const Entry &a, const Entry &b) {
            if (a.cell != b.cell) return a.cell < b.cell;
            if (a.point != b.point) return a.point < b.point;
            return a.corner < b.corner;
        });
        int nu = 0;
        int32_t *cellp = &uc_cell[(size_t)t * S.kmax];
        int32_t *ptrp = &uc_ptr[(size_t)t * (S.kmax + 1)];
        for (size_t e = 0; e < ent.size(); ++e) {
            if (e == 0 || ent[e].cell != ent[e - 1].cell) {
                cellp[nu] = ent[e].cell;
                ptrp[nu] = (int32_t)e;
                ++nu;
            }
            e_point[(size_t)t * S.kmax + e] = ent[e].point;
            e_w[(size_t)t * S.kmax + e] = ent[e].w;
        }
        ptrp[nu] = (int32_t)ent.size();
        uc_count[t] = nu;
        S.max_count = std::max(S.max_count, nu);
        if (ntiles3) {
            // order the owned cells by ((y,z) tile, x plane, position in the tile)
            struct C3 {
                int32_t tile, x, local, u;
            };
            std::vector<C3> cs(nu);
            for (int u = 0; u < nu; ++u) {
                const long rel = (long)cellp[u] - L.off0;
                const int i0 = (int)(rel / L.s0);
                const long r1 = rel - (long)i0 * L.s0;
                const int i1 = (int)(r1 / L.s1), i2 = (int)(r1 % L.s1);
                cs[u] = C3{(i1 / p.tile_tx) * p.ntz + (i2 / p.tile_tz), i0, (i1 % p.tile_tx) * p.tile_tz + (i2 % p.tile_tz), u};
                const int ii[3] = {i0, i1, i2};
                for (int d = 0; d < 3; ++d) S.c_lo[d] = std::min(S.c_lo[d], ii[d]), S.c_hi[d] = std::max(S.c_hi[d], ii[d] + 1);
            }
            std::sort(cs.begin(), cs.end(), [](const C3 &a, const C3 &b) {
                if (a.tile != b.tile) return a.tile < b.tile;
                if (a.x != b.x) return a.x < b.x;
                return a.local < b.local;
            });
            int32_t *cp = &cell_ptr[(size_t)t * ntiles3 * (L.n[0] + 1)];
            int j = 0;
            for (int tl = 0; tl < ntiles3; ++tl)
                for (int x = 0; x <= L.n[0]; ++x) {
                    cp[(size_t)tl * (L.n[0] + 1) + x] = j;
                    while (x < L.n[0] && j < nu && cs[j].tile == tl && cs[j].x == x) {
                        cell_local[(size_t)t * S.kmax + j] = cs[j].local;
                        cell_uc[(size_t)t * S.kmax + j] = cs[j].u;
                        ++j;
                    }
                }
        }
        if (ntiles) {
            // points sampled per tile, a point counted for the tile of its first corner that carries a weight
            std::vector<int> own(ntiles, 0);
            for (int q = 0; q < npoint; ++q)
                for (int b = 0; b < nc; ++b) {
                    const size_t e = ((size_t)t * npoint + q) * nc + b;
                    if (pt_w[e] == T(0)) continue;
                    const long rel = (long)pt_idx[e] - L.off0;
                    const int i1 = (int)(rel / L.s1), i2 = (int)(rel % L.s1);
                    S.max_own = std::max(S.max_own, ++own[(i1 / p.tile_tx) * p.ntz + (i2 / p.tile_tz)]);
                    break;
                }
            auto &tt = tiles[t];
            tt.resize(ntiles);
            tile_u[t].resize(ntiles);
            for (int u = 0; u < nu; ++u) {
                const long rel = (long)cellp[u] - L.off0;
                const int i1 = (int)(rel / L.s1), i2 = (int)(rel % L.s1);
                const int tile = (i1 / p.tile_tx) * p.ntz + (i2 / p.tile_tz);
                const int local = (i1 % p.tile_tx) * p.tile_tz + (i2 % p.tile_tz);
                std::vector<std::pair<int32_t, T>> es;
                for (int e = ptrp[u]; e < ptrp[u + 1]; ++e) es.push_back({ent[e].point, ent[e].w});
                S.maxe = std::max(S.maxe, (int)es.size());
                tt[tile].push_back({local, std::move(es)});
                tile_u[t][tile].push_back(u);
                S.cap = std::max(S.cap, (int)tt[tile].size());
            }
        }
    }
    std::vector<int32_t> slot_count, slot_local, slot_ep, slot_base, g_u;
    std::vector<T> slot_ew;
    if (ntiles) {
        slot_base.assign((size_t)tables * ntiles, 0);
        g_u.assign((size_t)tables * S.kmax, 0);
        S.cap = std::max(S.cap, 1);
        S.maxe = std::max(S.maxe, 1);
        slot_count.assign((size_t)tables * ntiles, 0);
        slot_local.assign((size_t)tables * ntiles * S.cap, -1);
        slot_ep.assign((size_t)tables * ntiles * S.cap * S.maxe, 0);
        slot_ew.assign((size_t)tables * ntiles * S.cap * S.maxe, T(0));
        for (int t = 0; t < tables; ++t)
            for (int tl = 0; tl < ntiles; ++tl) {
                const auto &cells = tiles[t][tl];
                slot_count[(size_t)t * ntiles + tl] = (int32_t)cells.size();
                slot_base[(size_t)t * ntiles + tl] = tl ? slot_base[(size_t)t * ntiles + tl - 1] +
                                                              slot_count[(size_t)t * ntiles + tl - 1]
                                                        : 0;
                for (size_t c = 0; c < cells.size(); ++c) {
                    const size_t sl = ((size_t)t * ntiles + tl) * S.cap + c;
                    slot_local[sl] = cells[c].first;
                    g_u[(size_t)t * S.kmax + slot_base[(size_t)t * ntiles + tl] + c] = tile_u[t][tl][c];
                    for (size_t e = 0; e < cells[c].second.size(); ++e) {
                        slot_ep[sl * S.maxe + e] = cells[c].second[e].first;
                        slot_ew[sl * S.maxe + e] = cells[c].second[e].second;
                    }
                }
            }
    }
    int rc;
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        if ((rc = dev_alloc(p, dst, bytes, false))) return rc;
        MIFWI_HIP(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, p.stream));
        return MIFWI_OK;
    };
    if ((rc = up((void **)&S.pt_idx, pt_idx.data(), pt_idx.size() * 4))) return rc;
    if ((rc = up(&S.pt_w, pt_w.data(), pt_w.size() * sizeof(T)))) return rc;
    if ((rc = up((void **)&S.uc_cell, uc_cell.data(), uc_cell.size() * 4))) return rc;
    if ((rc = up((void **)&S.uc_ptr, uc_ptr.data(), uc_ptr.size() * 4))) return rc;
    if ((rc = up((void **)&S.uc_count, uc_count.data(), uc_count.size() * 4))) return rc;
    if ((rc = up((void **)&S.e_point, e_point.data(), e_point.size() * 4))) return rc;
    if ((rc = up(&S.e_w, e_w.data(), e_w.size() * sizeof(T)))) return rc;
    if (ntiles3) {
        if ((rc = up((void **)&S.cell_ptr, cell_ptr.data(), cell_ptr.size() * 4))) return rc;
        if ((rc = up((void **)&S.cell_local, cell_local.data(), cell_local.size() * 4))) return rc;
        if ((rc = up((void **)&S.cell_uc, cell_uc.data(), cell_uc.size() * 4))) return rc;
    }
    if (ntiles) {
        if ((rc = up((void **)&S.slot_count, slot_count.data(), slot_count.size() * 4))) return rc;
        if ((rc = up((void **)&S.slot_local, slot_local.data(), slot_local.size() * 4))) return rc;
        if ((rc = up((void **)&S.slot_ep, slot_ep.data(), slot_ep.size() * 4))) return rc;
        if ((rc = up(&S.slot_ew, slot_ew.data(), slot_ew.size() * sizeof(T)))) return rc;
        if ((rc = up((void **)&S.slot_base, slot_base.data(), slot_base.size() * 4))) return rc;
        if ((rc = up((void **)&S.g_u, g_u.data(), g_u.size() * 4))) return rc;
    }
    // the host vectors die at return: make sure the uploads have been consumed
    MIFWI_HIP(hipStreamSynchronize(p.stream));
    return MIFWI_OK;
}

// ---- plan -------------------------------------------------------------------------------------
static int validate(const mifwi_config &c)
{
    if (c.ndim < 1 || c.ndim > 3) return fail(MIFWI_EINVAL, "ndim must be 1, 2 or 3");
    if (c.space_order < 2 || c.space_order > 2 * kMaxRUntiled || (c.space_order & 1))
        return fail(MIFWI_EINVAL, "space_order must be even and in 2..20");
    if ((c.flags & MIFWI_FLAG_PML) && c.space_order > 2 * kMaxR)
        return fail(MIFWI_EINVAL, "the perfectly matched layer is available up to space order 16");
    for (int d = 0; d < c.ndim; ++d) {
        if (c.shape[d] < 1) return fail(MIFWI_EINVAL, "shape entries must be positive");
        if (!(c.spacing[d] > 0.0)) return fail(MIFWI_EINVAL, "spacing entries must be positive");
    }
    if (c.nt < 3) return fail(MIFWI_EINVAL, "nt must be at least 3");
    if (!(c.dt > 0.0)) return fail(MIFWI_EINVAL, "dt must be positive");
    if (c.dtype != MIFWI_F32 && c.dtype != MIFWI_F64) return fail(MIFWI_EINVAL, "dtype must be MIFWI_F32 or MIFWI_F64");
    if (c.nsrc < 0 || c.nrec < 0) return fail(MIFWI_EINVAL, "nsrc/nrec must be non-negative");
    if (c.max_shots < 1) return fail(MIFWI_EINVAL, "max_shots must be at least 1");
    if (c.flags & MIFWI_FLAG_PML) {
        if (c.ndim != 2) return fail(MIFWI_EINVAL, "the perfectly matched layer is available for 2-D grids only");
        if (c.flags & (MIFWI_FLAG_OT4 | MIFWI_FLAG_FREE_SURFACE))
            return fail(MIFWI_EINVAL, "the perfectly matched layer is not combinable with OT4 or a free surface");
    }
    return MIFWI_OK;
}

static int create(const mifwi_config *cfg, mifwi_plan **out)
{
    if (!cfg || !out) return fail(MIFWI_EINVAL, "null argument");
    *out = nullptr;
    int rc = validate(*cfg);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        return fail(MIFWI_ENODEV, "no HIP device available (this library has no CPU path)");
    }
    int dev = cfg->device;
    if (dev < 0) MIFWI_HIP(hipGetDevice(&dev));
    if (dev >= ndev) return fail(MIFWI_EINVAL, "device ordinal out of range");
    MIFWI_HIP(hipSetDevice(dev));
    {
        double *unused = nullptr;   // first use of this device: look up the address of the static scratch array
        if ((rc = scratch_of_current_device(dev, &unused))) return rc;
    }
    Plan *p = new (std::nothrow) Plan();
    if (!p) return fail(MIFWI_ENOMEM, "host allocation failed");
    p->cfg = *cfg;
    p->device = dev;
    p->esize = cfg->dtype == MIFWI_F64 ? 8 : 4;
    if (make_layout(*cfg, p->L)) {
        delete p;
        return fail(MIFWI_EINVAL, "grid too large for 32-bit sparse offsets");
    }
    if (const char *env = getenv("MIFWI_PAIR_GRAD")) p->pair_grad = atoi(env);  // tuning experiments
    if (const char *env = getenv("MIFWI_PAIR_SAVE")) p->pair_save = atoi(env);  // tuning experiments
    if (const char *env = getenv("MIFWI_2D_VEC")) p->vec2d = atoi(env);  // tuning experiments
    // the fourth-order-in-time kernel runs on the tiled kernels too (see Step2DArgs::pt), except with
    // half-precision saved levels (its gradient stencils the saved level)
    // ... and with a free surface (the mirrored reads reach into laplace(u)/m as well: untiled kernels only)
    const bool ot4_untiled = (cfg->flags & MIFWI_FLAG_OT4) && ((cfg->flags & (MIFWI_FLAG_SAVE_F16 | MIFWI_FLAG_FREE_SURFACE)) || getenv("MIFWI_OT4_UNTILED"));
    // (space orders 18 and 20 exist on the untiled kernels only)
    const bool tiled_order = cfg->space_order <= 2 * kMaxR;
    if (cfg->ndim == 2 && cfg->dtype == MIFWI_F32 && !(cfg->flags & (MIFWI_FLAG_GENERIC | MIFWI_FLAG_PML)) && !ot4_untiled && tiled_order) {
        // tiled fast path: pick the tile shape that wastes fewer threads on this grid
        const int n1 = p->L.n[1], n2 = p->L.n[2];
        auto waste = [&](int tx, int tz) { return (double)((n1 + tx - 1) / tx * tx) * ((n2 + tz - 1) / tz * tz); };
        p->kind = 1;
        if (waste(16, 64) <= waste(32, 32)) p->tile_tx = 16, p->tile_tz = 64;
        else p->tile_tx = 32, p->tile_tz = 32;
        if (const char *env = getenv("MIFWI_2D_TX")) p->tile_tx = atoi(env), p->tile_tz = 32;  // tuning experiments
        p->ntx = (n1 + p->tile_tx - 1) / p->tile_tx;
        p->ntz = (n2 + p->tile_tz - 1) / p->tile_tz;
        hipDeviceProp_t prop{};
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) p->ncu = prop.multiProcessorCount;
        if (const char *env = getenv("MIFWI_PERSIST")) p->persist = atoi(env);   // 0: one kernel per step always
        if (const char *env = getenv("MIFWI_PERSIST_SPIN")) p->persist_spin = atoi(env);   // tests: 0 abandons at once
        if (const char *env = getenv("MIFWI_PERSIST_NOCHECK")) p->persist_check = atoi(env) == 0;   // tests only
    }
    if (cfg->ndim == 3 && cfg->dtype == MIFWI_F32 && !(cfg->flags & MIFWI_FLAG_GENERIC) && !ot4_untiled && tiled_order) {
        // 2.5-D fast path: 16 x 64 tiles of the (y, z) plane, x split into chunks so that one launch
        // offers several blocks per CU
        p->kind = 2;
        p->tile_tx = 16, p->tile_tz = 64;
        p->ntx = (p->L.n[1] + p->tile_tx - 1) / p->tile_tx;
        p->ntz = (p->L.n[2] + 63) / 64;
        hipDeviceProp_t prop{};
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) p->ncu = prop.multiProcessorCount;
        p->bpc_fwd = step3d_blocks_per_cu(p->L.R, false);
        p->bpc_grad = step3d_blocks_per_cu(p->L.R, true);
        // forward loops skip the planes the wave has not reached (see Step3DArgs::box_lo); not with the mirror rows of a
        // free surface, which live outside the planes' own rows
        p->box_enabled = !(cfg->flags & (MIFWI_FLAG_FREE_SURFACE | MIFWI_FLAG_OT4));
        if (const char *env = getenv("MIFWI_3D_BOX")) p->box_enabled = p->box_enabled && atoi(env) != 0;
        if (const char *env = getenv("MIFWI_3D_NCHUNK")) p->nchunk_env = std::max(1, atoi(env));  // tuning experiments
        if (const char *env = getenv("MIFWI_3D_NCHUNK_GRAD")) p->nchunk_grad_env = std::max(1, atoi(env));
        if ((cfg->flags & MIFWI_FLAG_TIME_BLOCK) && !(cfg->flags & MIFWI_FLAG_OT4)) p->slab_w = 64, p->slab_chunks = 2;   // the best measured at 592^3
        if (const char *env = getenv("MIFWI_3D_SLAB")) p->slab_w = atoi(env);                      // tuning experiments
        if (const char *env = getenv("MIFWI_3D_SLAB_CHUNKS")) p->slab_chunks = std::max(1, atoi(env));
        if (p->slab_w > 0 && p->slab_w < 2 * p->L.R) p->slab_w = 2 * p->L.R;
    }
    // the tiled kernels address a field level with 32-bit byte offsets; the 2.5-D kernel's buffer descriptors
    // span 2 GB - 1 bytes (offsets beyond that switch a lane off)
    if (p->kind != 0 && (unsigned long long)p->L.field * 4ull >= (p->kind == 2 ? 0x7fff0000ull : (1ull << 32))) p->kind = 0;
    auto bail = [&](int code) {
        destroy(p);
        return code;
    };
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(MIFWI_EHIP, "hipStreamCreate failed"));
    p->own_stream = true;
    if (p->kind == 1 && p->persist) {
        // the whole-loop kernel deals one shot to each of 8 XCDs: ask the device, once, where its blocks land
        static std::mutex census_mutex;
        static std::map<int, int> census;
        std::lock_guard<std::mutex> lock(census_mutex);
        auto it = census.find(dev);
        if (it == census.end()) it = census.emplace(dev, persist2d_xcd_count(p->stream)).first;
        p->xcds = it->second;
        if (const char *env = getenv("MIFWI_FAKE_XCDS")) p->xcds = atoi(env);   // tests: pretend another device
    }
    for (auto &pair : p->ev)
        for (hipEvent_t &e : pair)
            if (hipEventCreate(&e) != hipSuccess) return bail(fail(MIFWI_EHIP, "hipEventCreate failed"));
    const size_t es = p->esize;
    const size_t field = (size_t)p->L.field;
    const size_t S = (size_t)cfg->max_shots;
    const size_t nt = (size_t)cfg->nt;
    if ((rc = dev_alloc(*p, &p->m, field * es, true))) return bail(rc);
    if ((rc = dev_alloc(*p, &p->eta, field * es, true))) return bail(rc);
    if ((rc = dev_alloc(*p, &p->rot_u, S * 3 * field * es, true))) return bail(rc);
    if ((rc = dev_alloc(*p, &p->rot_v, S * 3 * field * es, true))) return bail(rc);
    if (cfg->flags & MIFWI_FLAG_PML) {
        if ((rc = dev_alloc(*p, &p->pml_phi, S * 4 * field * es, true))) return bail(rc);
        if ((rc = dev_alloc(*p, &p->pml_zeta, 2 * (size_t)(p->L.n[1] + p->L.n[2]) * es, true))) return bail(rc);
    }
    if (cfg->flags & MIFWI_FLAG_OT4) {
        if ((rc = dev_alloc(*p, &p->d_t4, S * field * es, true))) return bail(rc);
        if ((rc = dev_alloc(*p, &p->d_w2, S * field * es, true))) return bail(rc);
        if (p->kind != 0 && (rc = dev_alloc(*p, &p->d_bih, S * field * es, true))) return bail(rc);
    }
    if ((rc = dev_alloc(*p, &p->d_wavelet, S * nt * std::max(1, cfg->nsrc) * es, true))) return bail(rc);
    if ((rc = dev_alloc(*p, &p->d_srca, S * nt * std::max(1, cfg->nsrc) * es, true))) return bail(rc);
    if ((rc = dev_alloc(*p, &p->d_rec, S * nt * std::max(1, cfg->nrec) * es, true))) return bail(rc);
    if ((rc = dev_alloc(*p, &p->d_res, S * nt * std::max(1, cfg->nrec) * es, true))) return bail(rc);
    p->dense_elems = std::max<size_t>(2, S) * (size_t)p->L.ngrid;
    if ((rc = dev_alloc(*p, &p->d_dense, p->dense_elems * es, false))) return bail(rc);
    if ((rc = dev_alloc(*p, (void **)&p->d_partial, S * 64 * sizeof(double), true))) return bail(rc);
    if ((rc = dev_alloc(*p, (void **)&p->d_fval, S * sizeof(double), true))) return bail(rc);
    if (hipStreamSynchronize(p->stream) != hipSuccess) return bail(fail(MIFWI_EHIP, "stream synchronize failed"));
    *out = reinterpret_cast<mifwi_plan *>(p);
    return MIFWI_OK;
}

template <typename T>
static int set_model(Plan &p, const void *vp, const void *damp)
{
    const size_t bytes = (size_t)p.L.ngrid * sizeof(T);
    T *dvp = (T *)p.d_dense;
    T *ddamp = dvp + p.L.ngrid;
    int rc;
    const T *vp_dev = (const T *)vp, *damp_dev = (const T *)damp;
    bool from_host = false;
    if (!is_device_ptr(vp)) {
        if ((rc = copy_any(p, dvp, vp, bytes))) return rc;
        vp_dev = dvp;
        from_host = true;
    }
    if (damp && !is_device_ptr(damp)) {
        if ((rc = copy_any(p, ddamp, damp, bytes))) return rc;
        damp_dev = ddamp;
        from_host = true;
    }
    // pageable host sources must not be reused by the caller before the copy has happened; the kernels below are
    // ordered behind it on the plan's stream and need no waiting for
    if (from_host) MIFWI_HIP(hipStreamSynchronize(p.stream));
    if ((rc = launch_prep_model<T>(p, vp_dev, damp_dev, (T *)p.m, (T *)p.eta))) return rc;
    if (p.ck_K > 0 && p.saved_valid) {
        // checkpoints are re-propagated with the model at hand: those of the previous model are of no use now
        p.saved_valid = false;
        p.ck_model_changed = true;
    }
    if (p.tti_ck_K > 0) p.tti_save_valid = false;
    if (p.kind == 2) {
        if (!p.eta_zero && (rc = dev_alloc(p, (void **)&p.eta_zero, (size_t)p.ntx * p.ntz * p.L.n[0], true))) return rc;
        if (!p.eta_free && (rc = dev_alloc(p, (void **)&p.eta_free, (size_t)p.ntx * p.ntz * 2 * sizeof(int), true))) return rc;
        if ((rc = launch_eta_zero(p, p.eta_zero, p.eta_free))) return rc;
    }
    p.have_model = true;
    return MIFWI_OK;
}

template <typename T>
static int set_geometry(Plan &p, int nshots, const void *src_coords, const void *rec_coords, int rec_per_shot)
{
    const int nd = p.cfg.ndim;
    std::vector<T> sc((size_t)nshots * p.cfg.nsrc * nd), rc_((size_t)(rec_per_shot ? nshots : 1) * p.cfg.nrec * nd);
    if (!sc.empty()) MIFWI_HIP(hipMemcpy(sc.data(), src_coords, sc.size() * sizeof(T), hipMemcpyDefault));
    if (!rc_.empty()) MIFWI_HIP(hipMemcpy(rc_.data(), rec_coords, rc_.size() * sizeof(T), hipMemcpyDefault));
    int rc;
    clear_graphs(p);  // captured loops hold pointers into the geometry tables
    p.geo_src.assign((const unsigned char *)sc.data(), (const unsigned char *)(sc.data() + sc.size()));
    p.geo_rec.assign((const unsigned char *)rc_.data(), (const unsigned char *)(rc_.data() + rc_.size()));
    p.geo_rec_per_shot = rec_per_shot;
    if ((rc = build_sparse<T>(p, p.src, sc.data(), p.cfg.nsrc, nshots, 1))) return rc;
    if ((rc = build_sparse<T>(p, p.rec, rc_.data(), p.cfg.nrec, rec_per_shot ? nshots : 1, rec_per_shot ? 1 : 0)))
        return rc;
    p.have_geometry = true;
    p.geom_shots = nshots;
    return MIFWI_OK;
}

static int no_pml(const Plan &p)
{
    if (p.cfg.flags & MIFWI_FLAG_PML)
        return fail(MIFWI_EINVAL, "only mifwi_forward is available on a perfectly matched layer plan");
    return MIFWI_OK;
}

static int check_run(Plan &p, int nshots)
{
    if (!p.have_model) return fail(MIFWI_ESTATE, "mifwi_set_model has not been called");
    if (!p.have_geometry) return fail(MIFWI_ESTATE, "mifwi_set_geometry has not been called");
    if (nshots < 1 || nshots > p.cfg.max_shots) return fail(MIFWI_EINVAL, "nshots must be in 1..max_shots");
    if (nshots > p.geom_shots) return fail(MIFWI_EINVAL, "nshots exceeds the shots given to mifwi_set_geometry");
    return MIFWI_OK;
}

static void fill_stats(Plan &p, double ms, double steps, int nshots, double bytes_per_point, int launches,
                       bool backward)
{
    if (backward) {
        p.stats.bwd_ms = ms;
        p.stats.bwd_launches = launches;
    } else {
        p.stats.fwd_ms = ms;
        p.stats.fwd_launches = launches;
    }
    p.stats.kernel_ms = ms;
    p.stats.points = steps * (double)p.L.ngrid * nshots;
    p.stats.gpointss = ms > 0 ? p.stats.points / (ms * 1e-3) / 1e9 : 0.0;
    p.stats.algorithmic_bytes = p.stats.points * bytes_per_point * (p.esize / 4.0);
    p.stats.launches = launches;
    p.stats.kernel_kind = p.loop_persist ? 3 : p.loop_persist_tti ? 5 : p.kind;
    (backward ? p.stats.bwd_kernel_kind : p.stats.fwd_kernel_kind) = p.stats.kernel_kind;
    p.loop_persist = false;
    p.loop_persist_tti = false;
    p.stats.device_bytes = p.device_bytes;
    p.stats.ckpt_steps = p.ck_K > 0 ? p.ck_nc : 0;
    p.stats.ckpt_segment = p.ck_K;
    // MIFWI_FLAG_ASYNC: the loop has only been enqueued and ms is unknown; resolve_stats() fills the times in
    p.timing_pending[backward ? 1 : 0] = (p.cfg.flags & MIFWI_FLAG_ASYNC) != 0;
    p.timing_last = backward ? 1 : 0;
    p.timing_sum = false;
}

// Times of loops that were enqueued without waiting (MIFWI_FLAG_ASYNC): read the events once they have
// completed.  Blocks until they have; called by mifwi_get_stats and mifwi_synchronize.
static int resolve_stats(Plan &p)
{
    if (!p.timing_pending[0] && !p.timing_pending[1]) return MIFWI_OK;
    double sum = 0.0;
    for (int b = 0; b < 2; ++b) {
        if (!p.timing_pending[b]) continue;
        float ms = 0;
        MIFWI_HIP(hipEventSynchronize(p.ev[b][1]));
        MIFWI_HIP(hipEventElapsedTime(&ms, p.ev[b][0], p.ev[b][1]));
        (b ? p.stats.bwd_ms : p.stats.fwd_ms) = ms;
        sum += ms;
        if (!p.timing_sum && b == p.timing_last) p.stats.kernel_ms = ms;
        p.timing_pending[b] = false;
    }
    if (p.timing_sum) p.stats.kernel_ms = sum;
    p.stats.gpointss = p.stats.kernel_ms > 0 ? p.stats.points / (p.stats.kernel_ms * 1e-3) / 1e9 : 0.0;
    return MIFWI_OK;
}

template <typename T>
static int upload_wavelet(Plan &p, int nshots, const void *wavelet, int per_shot)
{
    const size_t one = (size_t)p.cfg.nt * p.cfg.nsrc * sizeof(T);
    if (one == 0) return MIFWI_OK;
    if (!wavelet) return fail(MIFWI_EINVAL, "wavelet is null");
    int rc;
    if (per_shot) return copy_any(p, p.d_wavelet, wavelet, one * nshots);
    for (int s = 0; s < nshots; ++s)
        if ((rc = copy_any(p, (char *)p.d_wavelet + s * one, wavelet, one))) return rc;
    return MIFWI_OK;
}

// constants of the tiled 2-D kernel
static Step2DArgs base2d(const Plan &p, int nshots)
{
    Step2DArgs a{};
    double w[kMaxR + 1];
    fd2_weights(p.L.R, w);
    const double ihx = 1.0 / (p.cfg.spacing[0] * p.cfg.spacing[0]), ihz = 1.0 / (p.cfg.spacing[1] * p.cfg.spacing[1]);
    a.c0 = (float)(w[0] * ihx + w[0] * ihz);
    for (int j = 0; j <= p.L.R; ++j) {
        a.cx[j] = (float)(w[j] * ihx);
        a.cz[j] = (float)(w[j] * ihz);
    }
    a.dt = (float)p.cfg.dt;
    a.dt2 = a.dt * a.dt;
    a.inv_dt2 = 1.0f / a.dt2;
    a.n1 = p.L.n[1];
    a.n2 = p.L.n[2];
    a.s1 = p.L.s1;
    a.off0 = p.L.off0;
    a.ntx = p.ntx;
    a.ntz = p.ntz;
    a.nshots = nshots;
    a.m = (const float *)p.m;
    a.eta = (const float *)p.eta;
    a.nt = p.cfg.nt;
    a.fs = (p.cfg.flags & MIFWI_FLAG_FREE_SURFACE) ? 1 : 0;
    return a;
}
// Form the injected terms of every time step of the coming loop (tiled kernels only).
template <typename T>
static int prepare_injection(Plan &p, const SparseSet &S, const void *vals, int nshots)
{
    if (p.kind == 0 || S.npoint <= 0 || S.max_count <= 0) return MIFWI_OK;
    const size_t need = (size_t)p.cfg.max_shots * p.cfg.nt * S.kmax;
    const int w = (&S == &p.rec) ? 1 : 0;
    int rc;
    if (need > p.winj_elems[w]) {
        clear_graphs(p);   // captured loops hold the old address
        dev_free(p.d_winj[w]);
        p.d_winj[w] = nullptr, p.winj_elems[w] = 0;
        if ((rc = dev_alloc(p, &p.d_winj[w], need * sizeof(float), false))) return rc;
        p.winj_elems[w] = need;
    }
    const float dt = (float)p.cfg.dt;
    return launch_inject_prepare(p, S, p.kind == 1 ? S.g_u : S.cell_uc, (const float *)vals, nshots, dt * dt,
                                 (float *)p.d_winj[w]);
}

static void set_inject(Step2DArgs &a, const SparseSet &S, const Plan &p)
{
    if (S.npoint <= 0 || S.max_count <= 0) return;
    a.slot_count = S.slot_count;
    a.slot_local = S.slot_local;
    a.slot_base = S.slot_base;
    a.inj_cap = S.cap;
    a.inj_per_shot = S.per_shot;
    a.winj = (const float *)p.d_winj[&S == &p.rec ? 1 : 0];
    a.winj_stride = S.kmax;
}
static void set_interp(Step2DArgs &a, const SparseSet &S, void *vals_out)
{
    if (S.npoint <= 0) return;
    a.pt_idx = S.pt_idx;
    a.pt_w = (const float *)S.pt_w;
    a.itp_npoint = S.npoint;
    a.itp_per_shot = S.per_shot;
    a.vals_out = (float *)vals_out;
}

// constants of the 2.5-D kernel
// Split x into chunks so that one launch fills the GPU an almost whole number of times.
// Every block primes its register queue with 2R extra planes, so few chunks would be cheapest, but
// blocks differ in cost (sponge-free planes skip a load, edge tiles are partly empty) and only a
// launch of several rounds of blocks balances that.  Measured on 592^3 (MI355X, 3 blocks/CU):
// tail efficiency (rounds / ceil(rounds)) explains the dips (3 chunks: 1.45 rounds, -15 %), and
// among well-filled choices 3-4 rounds ran 5-9 % faster than 1-2.
static void choose_chunks(const Plan &p, int nshots, bool grad, int &nchunk, int &xchunk)
{
    const int n0 = p.L.n[0];
    const int min_chunk = std::max(8, 2 * p.L.R);
    const int max_chunks = std::max(1, std::min(16, n0 / min_chunk));
    const double slots = (double)p.ncu * (grad ? p.bpc_grad : p.bpc_fwd);
    int best = 1;
    double best_score = -1.0;
    for (int nc = 1; nc <= max_chunks; ++nc) {
        const int xc = (n0 + nc - 1) / nc;
        const int real_nc = (n0 + xc - 1) / xc;
        const double rounds = (double)p.ntx * p.ntz * real_nc * nshots / slots;
        const double tail = rounds / std::ceil(rounds);
        const double prime = (double)xc / (xc + 2 * p.L.R);
        const double balance = std::min(1.0, 0.85 + 0.05 * rounds);   // saturates at 3 rounds
        const double score = tail * balance * std::pow(prime, 0.35);
        if (score > best_score) best_score = score, best = nc;
    }
    // Since the kernel stages the newest plane of a column, blocks that start together stay together and share
    // their halo lines in the L2: up to radius 6 ONE round that fills at least 65 % of the slots beats several
    // (592^3, forward step, 2 chunks = 0.72 of a round against the choice above: order 4 0.716 / 0.747 ms,
    // order 8 0.737 / 0.780, order 12 0.778 / 0.824; order 16: 0.847 / 0.811, which keeps the rule above).
    if (!grad && p.L.R <= 6 && !getenv("MIFWI_3D_CHUNKS_BY_ROUNDS")) {
        int one = 0;
        double fill = 0.0;
        for (int nc = 1; nc <= max_chunks; ++nc) {
            const int xc = (n0 + nc - 1) / nc;
            const int real_nc = (n0 + xc - 1) / xc;
            const double rounds = (double)p.ntx * p.ntz * real_nc * nshots / slots;
            if (rounds <= 1.0 && rounds > fill) fill = rounds, one = nc;
        }
        if (one && fill >= 0.65) best = one;
    }
    if (p.nchunk_env) best = std::min(p.nchunk_env, max_chunks);
    if (grad && p.nchunk_grad_env) best = std::min(p.nchunk_grad_env, max_chunks);
    xchunk = (n0 + best - 1) / best;
    nchunk = (n0 + xchunk - 1) / xchunk;
}

static Step3DArgs base3d(const Plan &p, int nshots, bool grad)
{
    Step3DArgs a{};
    double w[kMaxR + 1];
    fd2_weights(p.L.R, w);
    double ih[3];
    for (int d = 0; d < 3; ++d) ih[d] = 1.0 / (p.cfg.spacing[d] * p.cfg.spacing[d]);
    a.c0 = (float)(w[0] * ih[0] + w[0] * ih[1] + w[0] * ih[2]);
    for (int j = 0; j <= p.L.R; ++j) {
        a.cx[j] = (float)(w[j] * ih[0]);
        a.cy[j] = (float)(w[j] * ih[1]);
        a.cz[j] = (float)(w[j] * ih[2]);
    }
    a.dt = (float)p.cfg.dt;
    a.dt2 = a.dt * a.dt;
    a.inv_dt2 = 1.0f / a.dt2;
    a.n0 = p.L.n[0];
    a.n1 = p.L.n[1];
    a.n2 = p.L.n[2];
    a.s0 = p.L.s0;
    a.s1 = p.L.s1;
    a.off0 = p.L.off0;
    a.nty = p.ntx;
    a.ntz = p.ntz;
    choose_chunks(p, nshots, grad, a.nchunk, a.xchunk);
    a.x_lo = 0, a.x_hi = p.L.n[0];
    a.nshots = nshots;
    a.nwork = p.ntx * p.ntz * a.nchunk * nshots;
    a.eta_free = p.eta_free;
    a.m = (const float *)p.m;
    a.eta = (const float *)p.eta;
    a.nt = p.cfg.nt;
    a.fs = (p.cfg.flags & MIFWI_FLAG_FREE_SURFACE) ? 1 : 0;
    return a;
}
static void set_inject3(Step3DArgs &a, const SparseSet &S, const Plan &p)
{
    if (S.npoint <= 0 || S.max_count <= 0) return;
    a.cell_ptr = S.cell_ptr;
    a.cell_local = S.cell_local;
    a.inj_kmax = S.kmax;
    a.inj_per_shot = S.per_shot;
    a.winj = (const float *)p.d_winj[&S == &p.rec ? 1 : 0];
    a.winj_stride = S.kmax;
}
static void set_interp3(Step3DArgs &a, const SparseSet &S, void *vals_out)
{
    if (S.npoint <= 0) return;
    a.pt_idx = S.pt_idx;
    a.pt_w = (const float *)S.pt_w;
    a.itp_npoint = S.npoint;
    a.itp_per_shot = S.per_shot;
    a.vals_out = (float *)vals_out;
}

// ---- time loops ------------------------------------------------------------------------------
// Everything a loop needs that does not change from step to step.
template <typename T>
struct LoopCtx {
    int nshots;
    StepConsts<T> k;   // generic kernels
    Step2DArgs a2;     // tiled 2-D kernel (float only)
    Step3DArgs a3;     // 2.5-D kernel (float only)
};

template <typename T>
static LoopCtx<T> make_ctx(const Plan &p, int nshots, bool grad)
{
    LoopCtx<T> c{};
    c.nshots = nshots;
    c.k = make_consts<T>(p);
    if (p.kind == 1) c.a2 = base2d(p, nshots);
    if (p.kind == 2) c.a3 = base3d(p, nshots, grad);
    return c;
}

// Born source term of the scattered field (operators.py:263): background levels and dm
template <typename T>
struct BornTerm {
    const T *un, *uc, *uo;   // u[time+1] (source already injected), u[time], u[time-1]; shot 0
    long shot_stride;
    const T *dm;             // haloed, shared by the shots
};

// ---- fourth-order-in-time kernel on the tiled kernels (operators.py:52-56, 216-219) ---------------
// Before the step proper, a launch of the same kernel writes z = cur + dt^2/12 laplace(cur)/m into the scratch
// field d_t4 (one field per shot); the step then stencils z and takes its pointwise terms from the level itself
// (Step2DArgs::pt).  `base` carries the constants only: a derived field has nothing injected or sampled.
template <typename Args>
static void ot4_derive_args(const Plan &p, const Args &base, Args &q, const float *in, long in_stride, float *out, int mode)
{
    q = base;
    q.cur = in, q.old = in, q.out = out;   // `old` is read and not used
    q.shot_stride = in_stride;
    q.out_shot_stride = p.L.field;
    q.pre_mode = mode;
    q.pre_c = base.dt2 / 12.0f;
}
static int ot4_launch(Plan &p, const Step2DArgs &q) { return launch_step2d(p, q, p.stream); }
static int ot4_launch(Plan &p, const Step3DArgs &q) { return launch_step3d(p, q, p.stream); }
// z of a.cur; on return a stencils z
template <typename Args>
static int ot4_prestep(Plan &p, const Args &base, Args &a, int &launches)
{
    if (!(p.cfg.flags & MIFWI_FLAG_OT4)) return MIFWI_OK;
    Args q;
    ot4_derive_args(p, base, q, a.cur, a.shot_stride, (float *)p.d_t4, 1);
    int rc;
    if ((rc = ot4_launch(p, q))) return rc;
    launches += 1;
    a.pt = a.cur;
    a.cur = (const float *)p.d_t4;
    a.cur_shot_stride = p.L.field;
    a.pre_c = q.pre_c;
    return MIFWI_OK;
}
// gradient launches: a.bih = laplace(laplace(u0)/m^2) of the saved forward level u0, through d_w2
template <typename Args>
static int ot4_bih(Plan &p, const Args &base, Args &a, const float *u0, long u_stride, int &launches)
{
    if (!(p.cfg.flags & MIFWI_FLAG_OT4)) return MIFWI_OK;
    Args q;
    int rc;
    ot4_derive_args(p, base, q, u0, u_stride, (float *)p.d_w2, 2);
    if ((rc = ot4_launch(p, q))) return rc;
    ot4_derive_args(p, base, q, (const float *)p.d_w2, p.L.field, (float *)p.d_bih, 3);
    if ((rc = ot4_launch(p, q))) return rc;
    launches += 2;
    a.bih = (const float *)p.d_bih;
    return MIFWI_OK;
}

// One forward step: out = step(cur, old) + injected source term of `time`; with `interp` the
// receivers sample `cur` into d_rec[time].
template <typename T>
static int forward_step(Plan &p, const LoopCtx<T> &c, const T *cur, const T *old, T *out, long shot_stride, int time,
                        bool interp, int &launches, bool inject = true, T *copy = nullptr, long copy_stride = 0,
                        const BornTerm<T> *born = nullptr, T *copy_prev = nullptr, int x_lo = -1, int x_hi = -1,
                        int x_chunks = 1)
{
    int rc;
    if (p.kind == 1) {
        Step2DArgs a = c.a2;
        a.shot_stride = shot_stride;
        if (inject) set_inject(a, p.src, p);
        if (interp) set_interp(a, p.rec, p.d_rec);
        a.cur = (const float *)cur, a.old = (const float *)old, a.out = (float *)out, a.time = time;
        if ((rc = ot4_prestep(p, c.a2, a, launches))) return rc;
        a.copy = (float *)copy, a.copy_shot_stride = copy_stride, a.copy_prev = (float *)copy_prev;
        if (copy && p.save_f16) a.save_scale = (const float *)p.d_save_scale;
        if (born) {
            a.b_un = (const float *)born->un, a.b_uc = (const float *)born->uc, a.b_uo = (const float *)born->uo;
            a.b_dm = (const float *)born->dm, a.b_shot_stride = born->shot_stride;
        }
        if ((rc = launch_step2d(p, a, p.stream))) return rc;
        launches += 1;
        return MIFWI_OK;
    }
    if (p.kind == 2) {
        Step3DArgs a = c.a3;
        a.shot_stride = shot_stride;
        if (x_lo >= 0) {   // one slab of planes only (2.5-D kernel)
            a.x_lo = x_lo, a.x_hi = x_hi;
            a.xchunk = (x_hi - x_lo + x_chunks - 1) / x_chunks;
            a.nchunk = (x_hi - x_lo + a.xchunk - 1) / a.xchunk;
            a.nwork = a.nty * a.ntz * a.nchunk * a.nshots;
        }
        if (inject) set_inject3(a, p.src, p);
        if (interp) set_interp3(a, p.rec, p.d_rec);
        if (p.box_active && inject && !born && x_lo < 0) {
            a.box_lo = p.d_box, a.box_hi = p.d_box + 3 * p.cfg.max_shots, a.box_stride = p.cfg.max_shots;
            // the rotating levels are cleared when a forward loop begins and the box only grows: what lies outside it
            // there is still zero; the saved array and the segment buffers hold what earlier loops left
            const char *ob = (const char *)out, *rb = (const char *)p.rot_u;
            a.box_fill = !(rb && ob >= rb && ob < rb + 3 * (size_t)p.cfg.max_shots * p.L.field * sizeof(T));
            for (int d = 0; d < 3; ++d)
                a.inj_lo[d] = a.cell_ptr ? p.src.c_lo[d] : 0x7fffffff, a.inj_hi[d] = a.cell_ptr ? p.src.c_hi[d] : -0x7fffffff;
        }
        a.cur = (const float *)cur, a.old = (const float *)old, a.out = (float *)out, a.time = time;
        if ((rc = ot4_prestep(p, c.a3, a, launches))) return rc;
        if (copy) {
            if (!p.save_f16) return fail(MIFWI_EINVAL, "internal: the 3-D kernel copies half-precision levels only");
            a.copy16 = copy, a.copy_shot_stride = copy_stride, a.save_scale = (const float *)p.d_save_scale;
        }
        if (born) {
            a.b_un = (const float *)born->un, a.b_uc = (const float *)born->uc, a.b_uo = (const float *)born->uo;
            a.b_dm = (const float *)born->dm, a.b_shot_stride = born->shot_stride;
        }
        if ((rc = launch_step3d(p, a, p.stream))) return rc;
        launches += 1;
        return MIFWI_OK;
    }
    StepPtrs<T> sp{cur, old, out, shot_stride, (const T *)p.m, (const T *)p.eta, nullptr};
    GradPtrs<T> gp{};
    if (p.cfg.flags & MIFWI_FLAG_OT4) {
        if ((rc = launch_lap_scale<T>(p, c.k, cur, shot_stride, 1, (T *)p.d_t4, p.L.field, c.nshots))) return rc;
        sp.ot4 = (const T *)p.d_t4;
        launches += 1;
    }
    if ((rc = launch_step_generic<T>(p, c.k, sp, gp, c.nshots))) return rc;
    if (born) {   // untiled path: the source term is its own pointwise launch
        if ((rc = launch_born_source<T>(p, c.k, born->un, born->uc, born->uo, born->shot_stride, born->dm, out,
                                        shot_stride, c.nshots, 0)))
            return rc;
        launches += 1;
    }
    SparsePtrs<T> q{};
    if (inject) {
        q.inj = &p.src;
        q.field_inj = out;
        q.vals_inj = (const T *)(p.wavelet_override ? p.wavelet_override : p.d_wavelet);
    }
    if (interp) {
        q.itp = &p.rec;
        q.field_int = cur;
        q.vals_out = (T *)p.d_rec;
    }
    q.shot_stride = shot_stride;
    q.m = (const T *)p.m;
    q.time = time;
    q.nt = p.cfg.nt;
    q.dt2 = c.k.dt2;
    if ((rc = launch_sparse<T>(p, q, c.nshots))) return rc;
    launches += 2;
    return MIFWI_OK;
}

// One backward step: out = step(cur, old) + injected adjoint source of `time`.  With u0 != null
// the gradient (and illumination) accumulators are updated from the saved forward levels
// um/u0/up = u[time-1], u[time], u[time+1]; without, the field is sampled at the source positions.
// pair_role (tiled kernels only) 1: plain step of a gradient loop whose terms the next step adds
// (u0 == null, nothing sampled); 2: that next step.
template <typename T>
static int backward_step(Plan &p, const LoopCtx<T> &c, const T *cur, const T *old, T *out, long shot_stride, int time,
                         const T *um, const T *u0, const T *up, long u_stride, bool with_illum, bool first,
                         int &launches, int pair_role = 0)
{
    int rc;
    const bool with_grad = (u0 != nullptr);
    if (p.kind == 1) {
        Step2DArgs a = c.a2;
        a.shot_stride = shot_stride;
        set_inject(a, p.rec, p);
        if (!with_grad && pair_role != 1) set_interp(a, p.src, p.d_srca);
        a.pair = (pair_role == 2);
        a.cur = (const float *)cur, a.old = (const float *)old, a.out = (float *)out, a.time = time;
        if ((rc = ot4_prestep(p, c.a2, a, launches))) return rc;
        if (with_grad) {
            a.um = (const float *)um, a.uc = (const float *)u0, a.up = (const float *)up;
            a.u_shot_stride = u_stride;
            if (p.save_f16) a.save_scale = (const float *)p.d_save_scale;
            a.grad = (float *)p.d_grad;
            a.illum = with_illum ? (float *)p.d_illum : nullptr;
            a.g_shot_stride = p.L.field;
            a.first = first;
            if ((rc = ot4_bih(p, c.a2, a, (const float *)u0, u_stride, launches))) return rc;
        }
        if ((rc = launch_step2d(p, a, p.stream))) return rc;
        launches += 1;
        return MIFWI_OK;
    }
    if (p.kind == 2) {
        Step3DArgs a = c.a3;
        a.shot_stride = shot_stride;
        set_inject3(a, p.rec, p);
        if (!with_grad && pair_role != 1) set_interp3(a, p.src, p.d_srca);
        if (p.box_active_v) {   // `out` is one of the rotating levels, cleared when the loop began
            a.box_lo = p.d_boxv, a.box_hi = p.d_boxv + 3 * p.cfg.max_shots, a.box_stride = p.cfg.max_shots, a.box_fill = 0;
            for (int d = 0; d < 3; ++d)
                a.inj_lo[d] = a.cell_ptr ? p.rec.c_lo[d] : 0x7fffffff, a.inj_hi[d] = a.cell_ptr ? p.rec.c_hi[d] : -0x7fffffff;
        }
        a.pair = (pair_role == 2);
        a.cur = (const float *)cur, a.old = (const float *)old, a.out = (float *)out, a.time = time;
        if ((rc = ot4_prestep(p, c.a3, a, launches))) return rc;
        if (with_grad) {
            a.um = (const float *)um, a.uc = (const float *)u0, a.up = (const float *)up;
            a.u_shot_stride = u_stride;
            if (p.save_f16) a.save_scale = (const float *)p.d_save_scale;
            a.grad = (float *)p.d_grad;
            a.illum = with_illum ? (float *)p.d_illum : nullptr;
            a.g_shot_stride = p.L.field;
            a.first = first;
            if ((rc = ot4_bih(p, c.a3, a, (const float *)u0, u_stride, launches))) return rc;
        }
        if ((rc = launch_step3d(p, a, p.stream))) return rc;
        launches += 1;
        return MIFWI_OK;
    }
    StepPtrs<T> sp{cur, old, out, shot_stride, (const T *)p.m, (const T *)p.eta, nullptr};
    GradPtrs<T> gp{};
    if (with_grad) {
        gp.um = um, gp.uc = u0, gp.up = up;
        gp.u_shot_stride = u_stride;
        gp.grad = (T *)p.d_grad;
        gp.illum = with_illum ? (T *)p.d_illum : nullptr;
        gp.g_shot_stride = p.L.field;
        gp.first = first;
    }
    if (p.cfg.flags & MIFWI_FLAG_OT4) {
        if ((rc = launch_lap_scale<T>(p, c.k, cur, shot_stride, 1, (T *)p.d_t4, p.L.field, c.nshots))) return rc;
        sp.ot4 = (const T *)p.d_t4;
        launches += 1;
        if (with_grad) {
            if ((rc = launch_lap_scale<T>(p, c.k, u0, u_stride, 2, (T *)p.d_w2, p.L.field, c.nshots))) return rc;
            gp.bih = (const T *)p.d_w2;
            launches += 1;
        }
    }
    if ((rc = launch_step_generic<T>(p, c.k, sp, gp, c.nshots))) return rc;
    SparsePtrs<T> q{};
    q.inj = &p.rec;
    q.field_inj = out;
    q.vals_inj = (const T *)p.d_res;
    if (!with_grad) {
        q.itp = &p.src;
        q.field_int = cur;
        q.vals_out = (T *)p.d_srca;
    }
    q.shot_stride = shot_stride;
    q.m = (const T *)p.m;
    q.time = time;
    q.nt = p.cfg.nt;
    q.dt2 = c.k.dt2;
    if ((rc = launch_sparse<T>(p, q, c.nshots))) return rc;
    launches += 2;
    return MIFWI_OK;
}

// Run one time loop (`enqueue(launches)` launches its kernels on p.stream) bracketed by the plan's
// events.  With `key` the loop is captured once into a HIP graph, keyed by everything that changes
// its launches, and replayed afterwards: the host then issues one graph launch instead of nt-2
// kernel launches, which keeps small 2-D grids from being bound by the host's launch rate.
// (Splitting the batch into shot groups on separate streams was measured and rejected: chains on
// different hardware queues did not overlap on MI355X, 8 shots in 2/4/8 groups ran 1.2-3x slower
// than one chain.)
template <typename F>
static int timed_loop(Plan &p, const std::tuple<int, int, int, int, int> *key, F &&enqueue, int &launches, float &ms,
                      bool backward = false)
{
    int rc = MIFWI_OK;
    launches = 0;
    hipEvent_t e0 = p.ev[backward ? 1 : 0][0], e1 = p.ev[backward ? 1 : 0][1];
    MIFWI_HIP(hipEventRecord(e0, p.stream));
    // the legacy default stream (a null handle) cannot be captured: launch directly on it
    if (!key || (p.cfg.flags & MIFWI_FLAG_NO_GRAPH) || p.stream == nullptr) {
        if ((rc = enqueue(launches))) return rc;
    } else {
        auto it = p.graphs.find(*key);
        if (it == p.graphs.end()) {
            hipGraph_t graph = nullptr;
            MIFWI_HIP(hipStreamBeginCapture(p.stream, hipStreamCaptureModeThreadLocal));
            rc = enqueue(launches);
            hipError_t e = hipStreamEndCapture(p.stream, &graph);
            if (rc) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc;
            }
            if (e != hipSuccess) return fail(MIFWI_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
            hipGraphExec_t exec = nullptr;
            e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) return fail(MIFWI_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
            it = p.graphs.emplace(*key, std::make_pair(exec, launches)).first;
            // the capture only recorded the loop; it has not run yet
        }
        launches = it->second.second;
        MIFWI_HIP(hipGraphLaunch(it->second.first, p.stream));
    }
    MIFWI_HIP(hipEventRecord(e1, p.stream));
    ms = 0;
    if (p.cfg.flags & MIFWI_FLAG_ASYNC) return MIFWI_OK;   // the caller's fill_stats marks the time as pending
    MIFWI_HIP(hipEventSynchronize(e1));
    MIFWI_HIP(hipEventElapsedTime(&ms, e0, e1));
    return MIFWI_OK;
}

// copy `nlev` consecutive levels of every shot between two [shot][levels][field] arrays
template <typename T>
static int copy_levels(Plan &p, T *dst, long dst_stride, const T *src, long src_stride, int nlev, int nshots)
{
    MIFWI_HIP(hipMemcpy2DAsync(dst, (size_t)dst_stride * sizeof(T), src, (size_t)src_stride * sizeof(T),
                               (size_t)nlev * p.L.field * sizeof(T), (size_t)nshots, hipMemcpyDeviceToDevice, p.stream));
    return MIFWI_OK;
}

// Decide, on the first saving forward run, how the forward wavefield is kept for the gradient:
// every level (ck_K = 0) when that fits in free device memory, else two-level checkpointing of as
// few of the first steps as memory demands: the levels entering every ck_K-th step are kept and each
// segment is recomputed into a segment buffer right before the backward pass crosses it, while the
// levels of the remaining steps are stored one by one.  The gradient is bit-identical in every case
// (same kernels, same order); the price is a second propagation of the checkpointed steps.
template <typename T>
static int prepare_save(Plan &p)
{
    if (p.saved || p.ckpt) return MIFWI_OK;
    const int nt = p.cfg.nt;
    const size_t level = (size_t)p.cfg.max_shots * p.L.field * sizeof(T);
    size_t free_b = 0, total_b = 0;
    MIFWI_HIP(hipMemGetInfo(&free_b, &total_b));
    if (const char *env = getenv("MIFWI_SAVE_BUDGET_MB"))   // tests: pretend that less memory is free
        free_b = std::min(free_b, (size_t)std::max(1L, atol(env)) << 20);
    const bool force = (p.cfg.flags & MIFWI_FLAG_CHECKPOINT) != 0;
    // half-precision levels: float32 plans on the tiled kernels only
    const bool want_f16 = (p.cfg.flags & MIFWI_FLAG_SAVE_F16) && std::is_same<T, float>::value && p.kind != 0;
    const size_t slevel = want_f16 ? level / 2 : level;
    int rc;
    if (!force && (double)slevel * nt <= 0.92 * (double)free_b) {
        p.ck_K = 0;
        p.save_f16 = want_f16;
        if (want_f16 && !p.d_save_scale &&
            (rc = dev_alloc(p, &p.d_save_scale, (size_t)p.cfg.max_shots * sizeof(float), true)))
            return rc;
        return dev_alloc(p, &p.saved, slevel * nt, true);
    }
    p.save_f16 = false;   // checkpoints restart the propagation and stay float32
    const int nsteps = nt - 2;
    // Checkpoint the first ck_nc steps (segments of ck_K steps, ck_K even so that segments start at odd times and
    // never split a pair of gradient steps) and store every level from ck_nc on.  Memory in levels:
    // (nt - nc) stored + 2 per segment + ck_K + 2 for the segment being recomputed.  Of all (nc, K) that fit,
    // the smallest nc wins: only those steps are propagated a second time.  The tiled 2-D kernel and a forced
    // MIFWI_FLAG_CHECKPOINT checkpoint everything.
    const double budget = 0.95 * (double)free_b / (double)level;
    const bool all = force || p.kind == 1;
    int best_nc = -1, best_K = 0;
    for (int K = 2; !all && K <= nsteps; K += 2) {
        for (int nseg = 1; nseg * K < nsteps; ++nseg) {
            const int nc = nseg * K;
            if ((double)(nt - nc) + 2.0 * nseg + K + 2 > budget) continue;
            if (best_nc < 0 || nc < best_nc) best_nc = nc, best_K = K;
            break;   // more segments of this K only recompute more
        }
    }
    if (best_nc < 0) {   // everything checkpointed: K ~ sqrt(2 nsteps) minimises the memory
        int K = std::max(2, (int)std::ceil(std::sqrt(2.0 * nsteps)));
        K += K & 1;
        const int nseg = (nsteps + K - 1) / K;
        if (2.0 * nseg + K + 2 <= 0.95 * (double)free_b / (double)level) best_nc = nsteps, best_K = K;
    }
    if (best_nc < 0)
        return fail(MIFWI_ENOMEM, "not enough device memory even for the checkpointed wavefield (" +
                                      std::to_string((size_t)free_b >> 20) + " MiB free)");
    p.ck_K = best_K;
    p.ck_nc = best_nc;
    p.ck_nseg = (best_nc + best_K - 1) / best_K;
    if ((rc = dev_alloc(p, &p.ckpt, level * 2 * p.ck_nseg, true))) return rc;
    // the recomputation must see the wavelet of the run that took the checkpoints, whatever is modelled between
    if ((rc = dev_alloc(p, &p.d_wavelet_ck, (size_t)p.cfg.max_shots * nt * std::max(1, p.cfg.nsrc) * sizeof(T), true)))
        return rc;
    if ((rc = dev_alloc(p, &p.seg, level * (p.ck_K + 2), true))) return rc;
    if (p.ck_nc < nsteps) return dev_alloc(p, &p.saved, level * (size_t)(nt - p.ck_nc), true);   // levels ck_nc .. nt-1
    return MIFWI_OK;
}

// forward time loop on data already resident on the device (d_wavelet -> d_rec)
// ---- whole time loops in one launch (kernels_persist2d.hip) ---------------------------------------
// `sampled`: the point set the loop interpolates (its points must fit one per thread and tile), or null.
static bool persist_eligible(Plan &p, int nshots, const SparseSet *sampled, bool grad)
{
    if (!p.persist || p.kind != 1 || p.esize != 4 || p.save_f16 || (p.cfg.flags & MIFWI_FLAG_OT4)) return false;
    if (p.cfg.flags & (MIFWI_FLAG_ASYNC | MIFWI_FLAG_NO_GRAPH)) return false;
    if (!((p.tile_tx == 32 && p.tile_tz == 32) || (p.tile_tx == 16 && p.tile_tz == 64)) || p.vec2d != 4) return false;
    if (p.cfg.nt < 3) return false;
    // one shot per XCD, ids 0 .. 7, each with an equal share of the CUs: the full GPU in its single-partition mode
    if (p.xcds != 8 || p.ncu % 8 != 0) return false;
    if ((unsigned long long)p.L.field * 12ull >= (1ull << 31)) return false;   // a shot's three levels behind one descriptor
    if (sampled && sampled->max_own > 256) return false;
    int &bpc = p.persist_bpc[grad ? 1 : 0];
    if (bpc < 0) bpc = persist2d_blocks_per_cu(p.L.R, p.tile_tz, grad);
    // every tile of a shot needs a resident block on the shot's XCD for the whole loop
    if (bpc * (p.ncu / 8) < p.ntx * p.ntz) return false;
    if (p.persist_skip > 0) {   // a launch was abandoned not long ago (the GPU was busy with something else)
        --p.persist_skip;
        return false;
    }
    return true;
}

// Queue the launches that run the whole loop -- one per group of 8 shots (one shot per XCD), one group after
// the other -- and the read-back of the abort flag.
static int enqueue_persist(Plan &p, int nshots, bool forward, bool with_grad, bool with_illum, bool save, int &launches)
{
    const Layout &L = p.L;
    const int ntiles = p.ntx * p.ntz;
    const size_t nsync = (size_t)8 * ntiles + 16 + 4;
    int rc;
    if (!p.d_psync) {
        if ((rc = dev_alloc(p, (void **)&p.d_psync, nsync * sizeof(int), true))) return rc;
        MIFWI_HIP(hipHostMalloc((void **)&p.h_abort, sizeof(int), hipHostMallocDefault));
    }
    MIFWI_HIP(hipMemsetAsync(p.d_psync, 0, nsync * sizeof(int), p.stream));
    const size_t ngroup_words = (size_t)8 * ntiles + 16;   // counters a launch uses; the abort flag after them is shared
    Persist2DArgs q{};
    q.a = base2d(p, nshots);
    if (forward) {
        set_inject(q.a, p.src, p);
        set_interp(q.a, p.rec, p.d_rec);
        q.rot = (float *)p.rot_u;
    } else {
        set_inject(q.a, p.rec, p);
        if (!with_grad) set_interp(q.a, p.src, p.d_srca);
        q.rot = (float *)p.rot_v;
    }
    if (with_grad) {
        q.a.grad = (float *)p.d_grad;
        q.a.illum = with_illum ? (float *)p.d_illum : nullptr;
        q.a.g_shot_stride = L.field;
        q.usaved = (const float *)p.saved;
        q.usaved_lstride = (long)p.cfg.max_shots * L.field;
    }
    q.field = L.field;
    q.forward = forward ? 1 : 0;
    q.nsteps = p.cfg.nt - 2;
    q.saved = (forward && save) ? (float *)p.saved : nullptr;
    q.saved_lstride = (long)p.cfg.max_shots * L.field;
    q.done = p.d_psync;
    q.counter = p.d_psync + 8 * ntiles;
    q.abort_flag = q.counter + 16;
    q.spin_limit = p.persist_spin;
    q.drop_slot = -1;
    if (const char *env = getenv("MIFWI_PERSIST_DROP_SLOT")) q.drop_slot = atoi(env);   // tests
    const int resident = p.persist_bpc[with_grad ? 1 : 0] * (p.ncu / 8);
    launches = 0;
    for (int s0 = 0; s0 < nshots; s0 += 8) {
        if (s0 > 0) MIFWI_HIP(hipMemsetAsync(p.d_psync, 0, ngroup_words * sizeof(int), p.stream));
        q.a.shot0 = s0;
        q.a.nshots = std::min(8, nshots - s0);
        if ((rc = launch_persist2d(p, q, with_grad, std::min(resident, ntiles + 16), p.stream))) return rc;
        // completion is explicit: a shot whose tiles did not all count nsteps steps makes the loop run anew
        if (p.persist_check && (rc = launch_persist2d_check(q.done, q.a.nshots * ntiles, q.nsteps, q.abort_flag, p.stream))) return rc;
        launches += 1;
    }
    MIFWI_HIP(hipMemcpyAsync(p.h_abort, q.abort_flag, sizeof(int), hipMemcpyDeviceToHost, p.stream));
    return MIFWI_OK;
}

template <typename T>
static int run_forward(Plan &p, int nshots, int save)
{
    const Layout &L = p.L;
    const int nt = p.cfg.nt;
    int rc;
    if (save && (rc = prepare_save<T>(p))) return rc;
    const bool full_save = save && p.ck_K == 0;
    // Tiled 2-D kernel: the loop runs on the three rotating levels and streams a copy of every new
    // level into the saved array; measured 1.2-1.4 us per step faster than stepping through the
    // saved array itself, whose freshly written levels are not cache resident when read back.
    // Half-precision levels are always written as a copy (2-D and 3-D): stepping needs float32 levels.
    const bool copy_save = full_save && (p.kind == 1 || p.save_f16);
    const size_t es = p.save_f16 ? 2 : sizeof(T);   // bytes per saved element
    T *base;
    long shot_stride;
    if (full_save && !copy_save) {
        // saved field is level-major, [nt][max_shots][field]: one step reads and writes one
        // contiguous run of memory for the whole batch
        base = (T *)p.saved;
        shot_stride = L.field;
        MIFWI_HIP(hipMemsetAsync(base, 0, 2 * (size_t)p.cfg.max_shots * L.field * sizeof(T), p.stream));  // u[0] = u[1] = 0
    } else {
        base = (T *)p.rot_u;
        shot_stride = 3L * L.field;
        MIFWI_HIP(hipMemsetAsync(base, 0, (size_t)nshots * shot_stride * sizeof(T), p.stream));
        p.rot_u_valid = false;
        if (copy_save)   // u[0] = u[1] = 0
            MIFWI_HIP(hipMemsetAsync(p.saved, 0, 2 * (size_t)p.cfg.max_shots * L.field * es, p.stream));
    }
    if (save) p.saved_valid = false;
    MIFWI_HIP(hipMemsetAsync(p.d_rec, 0, (size_t)nshots * nt * p.cfg.nrec * sizeof(T), p.stream));
    if ((rc = prepare_injection<T>(p, p.src, p.d_wavelet, nshots))) return rc;
    if (copy_save && p.save_f16) {
        const bool have = p.src.npoint > 0 && p.src.max_count > 0;
        if ((rc = launch_save_scale(p, have ? (const float *)p.d_winj[0] : nullptr, p.src, nshots,
                                    (float *)p.d_save_scale)))
            return rc;
    }
    const LoopCtx<T> ctx = make_ctx<T>(p, nshots, false);
    const long sv_lstride = (long)p.cfg.max_shots * L.field;
    const bool in_saved = full_save && !copy_save;
    auto lvl = [&](int t) { return in_saved ? base + (long)t * sv_lstride : base + (long)(t % 3) * L.field; };
    const bool ckpt = save && !full_save;
    // steps after ck_nc of a checkpointed plan run inside the stored levels (`saved` holds levels ck_nc .. nt-1)
    const bool mixed = ckpt && p.ck_nc < nt - 2;
    auto stored = [&](int t) { return (T *)p.saved + (long)(t - p.ck_nc) * sv_lstride; };
    // Two steps per sweep over the grid (2.5-D kernel, nothing saved): x is cut into slabs of slab_w planes and
    // the second step follows the first one slab behind, so that what it reads -- the level the first step has
    // just written, the level before it and the coefficients -- is still in the 256 MB last-level cache.
    const int slab_w = (p.kind == 2 && !save) ? p.slab_w : 0;
    if (ckpt) {
        MIFWI_HIP(hipMemcpyAsync(p.d_wavelet_ck, p.d_wavelet, (size_t)nshots * nt * std::max(1, p.cfg.nsrc) * sizeof(T),
                                 hipMemcpyDeviceToDevice, p.stream));
        p.ck_model_changed = false;
    }
    auto enqueue_slabs = [&](int &launches) -> int {
        const int n0 = L.n[0], R = L.R, W = slab_w, S = (n0 + W - 1) / W;
        int r;
        for (int time = 1; time <= nt - 2; time += 2) {
            if (time + 1 > nt - 2) {   // odd number of steps: the last one alone
                if ((r = forward_step<T>(p, ctx, lvl(time), lvl(time - 1), lvl(time + 1), shot_stride, time, true, launches)))
                    return r;
                break;
            }
            for (int sl = 0; sl < S; ++sl) {
                // first step on planes [sl W, (sl+1) W): receivers sample u[time], complete since the last sweep
                if ((r = forward_step<T>(p, ctx, lvl(time), lvl(time - 1), lvl(time + 1), shot_stride, time, sl == 0,
                                         launches, true, nullptr, 0, nullptr, nullptr, sl * W, std::min(n0, (sl + 1) * W),
                                         p.slab_chunks)))
                    return r;
                // second step R planes behind; it overwrites u[time-1] only where the first step is done with it.
                // Receivers sample u[time+1], which is complete once the last slab of the first step has run.
                const int lo = std::max(0, sl * W - R), hi = (sl == S - 1) ? n0 : (sl + 1) * W - R;
                if (hi > lo &&
                    (r = forward_step<T>(p, ctx, lvl(time + 1), lvl(time), lvl(time + 2), shot_stride, time + 1, sl == S - 1,
                                         launches, true, nullptr, 0, nullptr, nullptr, lo, hi, p.slab_chunks)))
                    return r;
            }
        }
        return MIFWI_OK;
    };
    // planes the wave has not reached are written, not computed (2.5-D kernel, one step per launch)
    const bool box = p.kind == 2 && p.box_enabled && slab_w == 0;
    if (box && !p.d_box && (rc = dev_alloc(p, (void **)&p.d_box, 6 * (size_t)p.cfg.max_shots * sizeof(int), true))) return rc;
    if (box && ckpt && !p.d_ck_box &&
        (rc = dev_alloc(p, (void **)&p.d_ck_box, 6 * (size_t)p.cfg.max_shots * p.ck_nseg * sizeof(int), true)))
        return rc;
    auto enqueue = [&](int &launches) -> int {
        if (slab_w > 0) return enqueue_slabs(launches);
        if (box) {   // empty range: lo = 0x7f7f7f7f, hi = 0x80808080 (negative)
            MIFWI_HIP(hipMemsetAsync(p.d_box, 0x7f, 3 * (size_t)p.cfg.max_shots * sizeof(int), p.stream));
            MIFWI_HIP(hipMemsetAsync(p.d_box + 3 * p.cfg.max_shots, 0x80, 3 * (size_t)p.cfg.max_shots * sizeof(int), p.stream));
        }
        for (int time = 1; time <= nt - 2; ++time) {
            if (ckpt && time <= p.ck_nc && (time - 1) % p.ck_K == 0) {
                // keep the two levels that enter this segment: u[time-1], u[time]
                const int j = (time - 1) / p.ck_K;
                T *dst = (T *)p.ckpt + (long)j * 2 * L.field;
                const long cstride = (long)p.ck_nseg * 2 * L.field;
                int r;
                if ((r = copy_levels<T>(p, dst, cstride, lvl(time - 1), shot_stride, 1, nshots))) return r;
                if ((r = copy_levels<T>(p, dst + L.field, cstride, lvl(time), shot_stride, 1, nshots))) return r;
                if (box)   // and the range the two levels lie in
                    MIFWI_HIP(hipMemcpyAsync(p.d_ck_box + (size_t)j * 6 * p.cfg.max_shots, p.d_box,
                                             6 * (size_t)p.cfg.max_shots * sizeof(int), hipMemcpyDeviceToDevice, p.stream));
            }
            if (mixed && time > p.ck_nc) {
                int r;
                if (time == p.ck_nc + 1) {   // hand the two levels entering the stored part over to it
                    if ((r = copy_levels<T>(p, stored(time - 1), L.field, lvl(time - 1), shot_stride, 1, nshots))) return r;
                    if ((r = copy_levels<T>(p, stored(time), L.field, lvl(time), shot_stride, 1, nshots))) return r;
                }
                if ((r = forward_step<T>(p, ctx, stored(time), stored(time - 1), stored(time + 1), L.field, time, true,
                                         launches)))
                    return r;
                continue;
            }
            auto saved_at = [&](int level) { return (T *)((char *)p.saved + (size_t)level * sv_lstride * es); };
            T *copy = copy_save ? saved_at(time + 1) : nullptr, *copy_prev = nullptr;
            if (copy_save && p.kind == 1 && p.pair_save) {
                // steps at even times write the saved levels time and time + 1, the others none (the last
                // step always writes its own)
                if (time % 2 == 0) copy_prev = saved_at(time);
                else if (time != nt - 2) copy = nullptr;
            }
            int r = forward_step<T>(p, ctx, lvl(time), lvl(time - 1), lvl(time + 1), shot_stride, time, true, launches,
                                    true, copy, L.field, nullptr, copy_prev);
            if (r) return r;
        }
        return MIFWI_OK;
    };
    int launches = 0;
    float ms = 0;
    const auto key = std::make_tuple(0, nshots, save ? 1 : 0, slab_w, 0);
    struct BoxScope {   // forward_step hands the range to the kernel while this loop is being enqueued
        Plan &p;
        BoxScope(Plan &q, bool on) : p(q) { p.box_active = on; }
        ~BoxScope() { p.box_active = false; }
    } box_scope(p, box);
    if (!ckpt && !in_saved && persist_eligible(p, nshots, &p.rec, false)) {
        auto whole_loop = [&](int &n) -> int { return enqueue_persist(p, nshots, true, false, false, copy_save, n); };
        if ((rc = timed_loop(p, nullptr, whole_loop, launches, ms))) return rc;
        if (*p.h_abort) {   // a block waited too long for a neighbour: this loop anew, and the next ones, a kernel per step
            p.persist_skip = p.persist_backoff;
            p.persist_backoff = std::min(4096, 2 * p.persist_backoff);
            p.stats.persist_fallbacks += 1;
            return run_forward<T>(p, nshots, save);
        }
        p.loop_persist = true;
    } else if ((rc = timed_loop(p, ckpt ? nullptr : &key, enqueue, launches, ms))) {
        return rc;
    }
    fill_stats(p, ms, nt - 2, nshots, 20.0, launches, false);
    if (save) {
        p.saved_valid = true;
        p.saved_shots = nshots;
    }
    if (!in_saved && !mixed) p.rot_u_valid = true;
    return MIFWI_OK;
}

// backward time loop: adjoint source in d_res; with_grad accumulates into d_grad/d_illum,
// otherwise samples the field at the source positions into d_srca.
template <typename T>
static int run_backward(Plan &p, int nshots, bool with_grad, bool with_illum)
{
    const Layout &L = p.L;
    const int nt = p.cfg.nt;
    int rc;
    T *base = (T *)p.rot_v;
    const long shot_stride = 3L * L.field;
    MIFWI_HIP(hipMemsetAsync(base, 0, (size_t)nshots * shot_stride * sizeof(T), p.stream));
    p.rot_v_valid = false;
    if (with_grad) {
        if (!p.saved_valid && p.ck_model_changed)
            return fail(MIFWI_ESTATE, "mifwi_set_model was called after the checkpoints of the forward wavefield were "
                                      "taken: run forward(save=1) again with the new model");
        if (!p.saved_valid || p.saved_shots < nshots)
            return fail(MIFWI_ESTATE, "gradient needs the wavefield of a preceding forward(save=1) for these shots");
        if (!p.d_grad && (rc = dev_alloc(p, &p.d_grad, (size_t)p.cfg.max_shots * L.field * sizeof(T), true))) return rc;
        MIFWI_HIP(hipMemsetAsync(p.d_grad, 0, (size_t)nshots * L.field * sizeof(T), p.stream));
        if (with_illum) {
            if (!p.d_illum && (rc = dev_alloc(p, &p.d_illum, (size_t)p.cfg.max_shots * L.field * sizeof(T), true)))
                return rc;
            MIFWI_HIP(hipMemsetAsync(p.d_illum, 0, (size_t)nshots * L.field * sizeof(T), p.stream));
        }
    } else {
        MIFWI_HIP(hipMemsetAsync(p.d_srca, 0, (size_t)nshots * nt * p.cfg.nsrc * sizeof(T), p.stream));
    }
    if ((rc = prepare_injection<T>(p, p.rec, p.d_res, nshots))) return rc;
    if (with_grad && p.ck_K > 0 && (rc = prepare_injection<T>(p, p.src, p.d_wavelet_ck, nshots))) return rc;  // recomputation
    // untiled kernels read the wavelet itself while recomputing: point them at the kept copy for this loop
    struct Override {
        Plan &p;
        ~Override() { p.wavelet_override = nullptr; }
    } override_guard{p};
    if (with_grad && p.ck_K > 0) p.wavelet_override = p.d_wavelet_ck;
    const LoopCtx<T> ctx = make_ctx<T>(p, nshots, with_grad);
    const LoopCtx<T> fctx = make_ctx<T>(p, nshots, false);
    auto lvl = [&](int t) { return base + (long)((t + 3) % 3) * L.field; };
    const bool ckpt = with_grad && p.ck_K > 0;
    // Tiled kernels: two gradient steps share one update of the accumulators.  The step at an even
    // time runs as a plain propagation (role 1) and the following one, at the odd time below it, adds both
    // steps' terms (role 2).  Roles depend on the time index alone, so a checkpointed loop (segments start
    // at odd times) runs the same kernel at every step as the fully saved one.
    // Measured on 592^3 (2.5-D kernel): backward loop at order 8 with half-precision levels 670 -> 608 ms,
    // checkpointed with float levels 1084 -> 1033 ms (recomputation included); fully saved float levels,
    // nt = 200: order 16 291 -> 272 ms, order 4 256 -> 242 ms.
    const bool pair_grad = with_grad && p.pair_grad && p.kind != 0 && !(p.cfg.flags & MIFWI_FLAG_OT4);
    auto pair_role = [&](int time) {
        if (!pair_grad) return 0;
        if (time % 2 == 0) return (time != nt - 2 && time >= 2) ? 1 : 0;
        return (time + 1 < nt - 2) ? 2 : 0;
    };
    // where the backward field has not arrived nothing is computed (2.5-D kernel); the illumination sums need every point
    const bool boxv = p.kind == 2 && p.box_enabled && !(with_grad && with_illum);
    if (boxv && !p.d_boxv && (rc = dev_alloc(p, (void **)&p.d_boxv, 6 * (size_t)p.cfg.max_shots * sizeof(int), true))) return rc;
    struct BoxScopeV {
        Plan &p;
        BoxScopeV(Plan &q, bool on) : p(q) { p.box_active_v = on; }
        ~BoxScopeV() { p.box_active_v = false; }
    } box_scope_v(p, boxv);
    auto enqueue = [&](int &launches) -> int {
        int r;
        if (boxv) {   // empty box
            MIFWI_HIP(hipMemsetAsync(p.d_boxv, 0x7f, 3 * (size_t)p.cfg.max_shots * sizeof(int), p.stream));
            MIFWI_HIP(hipMemsetAsync(p.d_boxv + 3 * p.cfg.max_shots, 0x80, 3 * (size_t)p.cfg.max_shots * sizeof(int), p.stream));
        }
        if (!ckpt) {
            // saved levels: float, or half precision (then the kernels take the pointers as raw addresses)
            const char *sv = (const char *)p.saved;
            const size_t es = p.save_f16 ? 2 : sizeof(T);
            const long sv_lstride = (long)p.cfg.max_shots * L.field, sv_stride = L.field;
            auto saved_level = [&](int t) { return (const T *)(sv + (size_t)t * sv_lstride * es); };
            for (int time = nt - 2; time >= 1; --time) {
                const T *u0 = with_grad ? saved_level(time) : nullptr;
                const int role = pair_role(time);
                if (role == 1) u0 = nullptr;
                if ((r = backward_step<T>(p, role == 1 ? fctx : ctx, lvl(time), lvl(time + 1), lvl(time - 1), shot_stride, time,
                                          u0 ? saved_level(time - 1) : nullptr, u0, u0 ? saved_level(time + 1) : nullptr,
                                          sv_stride, with_illum, time == nt - 2, launches, role)))
                    return r;
            }
            return MIFWI_OK;
        }
        // checkpointed.  The steps after ck_nc come first: their levels are stored (`saved` holds ck_nc .. nt-1)
        {
            const long sv_lstride = (long)p.cfg.max_shots * L.field;
            auto stored = [&](int t) { return (const T *)p.saved + (long)(t - p.ck_nc) * sv_lstride; };
            for (int time = nt - 2; time > p.ck_nc; --time) {
                const int role = pair_role(time);
                const bool u = role != 1;
                if ((r = backward_step<T>(p, role == 1 ? fctx : ctx, lvl(time), lvl(time + 1), lvl(time - 1), shot_stride, time,
                                          u ? stored(time - 1) : nullptr, u ? stored(time) : nullptr,
                                          u ? stored(time + 1) : nullptr, L.field, with_illum, time == nt - 2, launches, role)))
                    return r;
            }
        }
        // then walk the segments backwards, recomputing each into the segment buffer first
        T *seg = (T *)p.seg;
        const long seg_stride = (long)(p.ck_K + 2) * L.field;
        const long cstride = (long)p.ck_nseg * 2 * L.field;
        for (int j = p.ck_nseg - 1; j >= 0; --j) {
            const int t0 = 1 + j * p.ck_K, t1 = std::min(p.ck_nc, t0 + p.ck_K - 1);
            // seg level i holds u[t0 - 1 + i]
            if ((r = copy_levels<T>(p, seg, seg_stride, (const T *)p.ckpt + (long)j * 2 * L.field, cstride, 2, nshots)))
                return r;
            const bool box = p.kind == 2 && p.box_enabled && p.d_box && p.d_ck_box;
            if (box)   // the range the checkpointed levels lie in
                MIFWI_HIP(hipMemcpyAsync(p.d_box, p.d_ck_box + (size_t)j * 6 * p.cfg.max_shots,
                                         6 * (size_t)p.cfg.max_shots * sizeof(int), hipMemcpyDeviceToDevice, p.stream));
            p.box_active = box;
            for (int time = t0; time <= t1; ++time) {
                T *c = seg + (long)(time - t0 + 1) * L.field;
                if ((r = forward_step<T>(p, fctx, c, c - L.field, c + L.field, seg_stride, time, false, launches))) {
                    p.box_active = false;
                    return r;
                }
            }
            p.box_active = false;
            for (int time = t1; time >= t0; --time) {
                const T *u0 = seg + (long)(time - t0 + 1) * L.field;
                const int role = pair_role(time);
                if ((r = backward_step<T>(p, role == 1 ? fctx : ctx, lvl(time), lvl(time + 1), lvl(time - 1), shot_stride, time,
                                          u0 - L.field, role == 1 ? nullptr : u0, u0 + L.field, seg_stride, with_illum,
                                          time == nt - 2, launches, role)))
                    return r;
            }
        }
        return MIFWI_OK;
    };
    int launches = 0;
    float ms = 0;
    const auto key = std::make_tuple(1, nshots, with_grad ? 1 : 0, with_illum ? 1 : 0, 0);
    if (!ckpt && persist_eligible(p, nshots, with_grad ? nullptr : &p.src, with_grad)) {
        auto whole_loop = [&](int &n) -> int { return enqueue_persist(p, nshots, false, with_grad, with_illum, false, n); };
        if ((rc = timed_loop(p, nullptr, whole_loop, launches, ms, true))) return rc;
        if (*p.h_abort) {
            p.persist_skip = p.persist_backoff;
            p.persist_backoff = std::min(4096, 2 * p.persist_backoff);
            p.stats.persist_fallbacks += 1;
            return run_backward<T>(p, nshots, with_grad, with_illum);
        }
        p.loop_persist = true;
    } else if ((rc = timed_loop(p, ckpt ? nullptr : &key, enqueue, launches, ms, true))) {
        return rc;
    }
    fill_stats(p, ms, nt - 2, nshots, with_grad ? 32.0 : 20.0, launches, true);
    p.rot_v_valid = true;
    return MIFWI_OK;
}

// deliver haloed per-shot fields (d_grad / d_illum) to a dense caller array
template <typename T>
static int deliver_grid(Plan &p, const void *haloed, void *user, int nshots, int accumulate)
{
    if (!user) return MIFWI_OK;
    const size_t bytes = (size_t)nshots * p.L.ngrid * sizeof(T);
    int rc;
    if (is_device_ptr(user)) return launch_unpack<T>(p, (const T *)haloed, p.L.field, (T *)user, nshots, accumulate ? 1 : 0);
    if ((size_t)nshots * p.L.ngrid > p.dense_elems) return fail(MIFWI_EINVAL, "internal staging buffer too small");
    if (accumulate && (rc = copy_any(p, p.d_dense, user, bytes))) return rc;
    if ((rc = launch_unpack<T>(p, (const T *)haloed, p.L.field, (T *)p.d_dense, nshots, accumulate ? 1 : 0))) return rc;
    return copy_any(p, user, p.d_dense, bytes);
}

template <typename T>
static int run_forward_pml(Plan &p, int nshots);

template <typename T>
static int forward(Plan &p, int nshots, const void *wavelet, int per_shot, void *rec_out, int save)
{
    int rc;
    if ((rc = check_run(p, nshots))) return rc;
    if ((rc = upload_wavelet<T>(p, nshots, wavelet, per_shot))) return rc;
    if (p.cfg.flags & MIFWI_FLAG_PML) {
        if (save) return fail(MIFWI_EINVAL, "a perfectly matched layer plan models without saving the wavefield");
        if ((rc = run_forward_pml<T>(p, nshots))) return rc;
    } else if ((rc = run_forward<T>(p, nshots, save))) {
        return rc;
    }
    if (rec_out && (rc = copy_any(p, rec_out, p.d_rec, (size_t)nshots * p.cfg.nt * p.cfg.nrec * sizeof(T)))) return rc;
    return finish(p);
}

// Linearised modelling: u in rot_u, U in rot_v, both advanced in one time loop.
template <typename T>
static int born(Plan &p, int nshots, const void *wavelet, int per_shot, const void *dm, void *rec_out)
{
    int rc;
    if (int e = no_pml(p)) return e;
    if ((rc = check_run(p, nshots))) return rc;
    if (!dm) return fail(MIFWI_EINVAL, "dm is null");
    if ((rc = upload_wavelet<T>(p, nshots, wavelet, per_shot))) return rc;
    const Layout &L = p.L;
    const int nt = p.cfg.nt;
    if (!p.d_dm && (rc = dev_alloc(p, &p.d_dm, (size_t)L.field * sizeof(T), true))) return rc;
    if ((rc = copy_any(p, p.d_dense, dm, (size_t)L.ngrid * sizeof(T)))) return rc;
    if ((rc = launch_pack<T>(p, (const T *)p.d_dense, (T *)p.d_dm, L.field, 1))) return rc;
    T *bu = (T *)p.rot_u, *bU = (T *)p.rot_v;
    const long stride = 3L * L.field;
    MIFWI_HIP(hipMemsetAsync(bu, 0, (size_t)nshots * stride * sizeof(T), p.stream));
    MIFWI_HIP(hipMemsetAsync(bU, 0, (size_t)nshots * stride * sizeof(T), p.stream));
    MIFWI_HIP(hipMemsetAsync(p.d_rec, 0, (size_t)nshots * nt * p.cfg.nrec * sizeof(T), p.stream));
    p.rot_u_valid = p.rot_v_valid = false;
    if ((rc = prepare_injection<T>(p, p.src, p.d_wavelet, nshots))) return rc;
    const LoopCtx<T> ctx = make_ctx<T>(p, nshots, false);
    auto enqueue = [&](int &launches) -> int {
        for (int time = 1; time <= nt - 2; ++time) {
            const long c = (long)(time % 3) * L.field, o = (long)((time + 2) % 3) * L.field,
                       n = (long)((time + 1) % 3) * L.field;
            int r;
            if ((r = forward_step<T>(p, ctx, bu + c, bu + o, bu + n, stride, time, false, launches, true))) return r;
            const BornTerm<T> bt{bu + n, bu + c, bu + o, stride, (const T *)p.d_dm};
            if ((r = forward_step<T>(p, ctx, bU + c, bU + o, bU + n, stride, time, true, launches, false, nullptr, 0, &bt)))
                return r;
        }
        return MIFWI_OK;
    };
    int launches = 0;
    float ms = 0;
    const auto key = std::make_tuple(3, nshots, 0, 0, 0);
    if ((rc = timed_loop(p, &key, enqueue, launches, ms))) return rc;
    // two propagations (20 B/pt each) + the source term's reads (3 levels of u, dm)
    fill_stats(p, ms, nt - 2, nshots, 56.0, launches, false);
    p.rot_u_valid = p.rot_v_valid = true;
    if (rec_out && (rc = copy_any(p, rec_out, p.d_rec, (size_t)nshots * nt * p.cfg.nrec * sizeof(T)))) return rc;
    return finish(p);
}

template <typename T>
static int adjoint(Plan &p, int nshots, const void *rec_in, void *srca_out)
{
    int rc;
    if (int e = no_pml(p)) return e;
    if ((rc = check_run(p, nshots))) return rc;
    if (!rec_in) return fail(MIFWI_EINVAL, "rec_in is null");
    if ((rc = copy_any(p, p.d_res, rec_in, (size_t)nshots * p.cfg.nt * p.cfg.nrec * sizeof(T)))) return rc;
    if ((rc = run_backward<T>(p, nshots, false, false))) return rc;
    if (srca_out && (rc = copy_any(p, srca_out, p.d_srca, (size_t)nshots * p.cfg.nt * p.cfg.nsrc * sizeof(T)))) return rc;
    return finish(p);
}

template <typename T>
static int gradient(Plan &p, int nshots, const void *residual, void *grad, int accumulate, void *illum)
{
    int rc;
    if (int e = no_pml(p)) return e;
    if ((rc = check_run(p, nshots))) return rc;
    if (!residual || !grad) return fail(MIFWI_EINVAL, "residual or grad is null");
    if ((rc = copy_any(p, p.d_res, residual, (size_t)nshots * p.cfg.nt * p.cfg.nrec * sizeof(T)))) return rc;
    if ((rc = run_backward<T>(p, nshots, true, illum != nullptr))) return rc;
    if ((rc = deliver_grid<T>(p, p.d_grad, grad, nshots, accumulate))) return rc;
    if (illum && (rc = deliver_grid<T>(p, p.d_illum, illum, nshots, 0))) return rc;
    return finish(p);
}

template <typename T>
static int l2_misfit_gradient(Plan &p, int nshots, const void *wavelet, int per_shot, const void *obs, double *fval,
                              void *grad, void *illum, void *residual_out, void *syn_out)
{
    int rc;
    if ((rc = no_pml(p))) return rc;
    if ((rc = check_run(p, nshots))) return rc;
    if (!obs) return fail(MIFWI_EINVAL, "obs is null");
    const size_t dbytes = (size_t)nshots * p.cfg.nt * p.cfg.nrec * sizeof(T);
    if ((rc = upload_wavelet<T>(p, nshots, wavelet, per_shot))) return rc;
    double ms_total = 0, pts = 0, abytes = 0;
    int launches = 0;
    if ((rc = run_forward<T>(p, nshots, grad != nullptr))) return rc;
    ms_total += p.stats.kernel_ms, pts += p.stats.points, abytes += p.stats.algorithmic_bytes, launches += p.stats.launches;
    // observed data: use in place when already on the device, else stage it in d_res
    const T *obs_dev = (const T *)obs;
    if (!is_device_ptr(obs)) {
        if ((rc = copy_any(p, p.d_res, obs, dbytes))) return rc;
        obs_dev = (const T *)p.d_res;
    }
    if ((rc = launch_l2_misfit<T>(p, (const T *)p.d_rec, obs_dev, (T *)p.d_res, (long)p.cfg.nt * p.cfg.nrec, nshots,
                                  p.d_partial, p.d_fval)))
        return rc;
    if (fval && (rc = copy_any(p, fval, p.d_fval, (size_t)nshots * sizeof(double)))) return rc;
    if (syn_out && (rc = copy_any(p, syn_out, p.d_rec, dbytes))) return rc;
    if (residual_out && (rc = copy_any(p, residual_out, p.d_res, dbytes))) return rc;
    if (grad) {
        if ((rc = run_backward<T>(p, nshots, true, illum != nullptr))) return rc;
        ms_total += p.stats.kernel_ms, pts += p.stats.points, abytes += p.stats.algorithmic_bytes,
            launches += p.stats.launches;
        if ((rc = deliver_grid<T>(p, p.d_grad, grad, nshots, 0))) return rc;
        if (illum && (rc = deliver_grid<T>(p, p.d_illum, illum, nshots, 0))) return rc;
    }
    p.stats.kernel_ms = ms_total;
    p.stats.points = pts;
    p.stats.algorithmic_bytes = abytes;
    p.stats.gpointss = ms_total > 0 ? pts / (ms_total * 1e-3) / 1e9 : 0;
    p.stats.launches = launches;
    p.timing_sum = true;   // asynchronous plans: kernel_ms = forward + backward once both are known
    return finish(p);
}

// Damping profiles of the perfectly matched layer (fdamp of seismic/abc_methods/03_pml.ipynb cell 60): along a
// dimension of n nodes, a = depth into the npml-cell layer / its width at node i and at the staggered point i + 1/2
template <typename T>
static int set_pml(Plan &p, int npml, double quibar, int sides, int edges, const void *vp_staggered)
{
    if (!(p.cfg.flags & MIFWI_FLAG_PML)) return fail(MIFWI_ESTATE, "the plan was not created with MIFWI_FLAG_PML");
    if (npml < 1 || 2 * npml >= std::min(p.L.n[1], p.L.n[2]) || !(quibar >= 0.0))
        return fail(MIFWI_EINVAL, "npml must be positive and leave an interior; quibar must not be negative");
    if (sides < 0 || sides > 15 || (edges != 0 && edges != 1)) return fail(MIFWI_EINVAL, "sides is a mask of 4 bits, edges 0 or 1");
    const int n1 = p.L.n[1], n2 = p.L.n[2];
    std::vector<T> z(2 * (size_t)(n1 + n2));
    auto zeta = [&](double pos, int n, bool low, bool high) {   // pos: node coordinate in cells
        double a = 0.0;
        if (low && pos <= npml) a = (npml - pos) / npml;
        else if (high && pos >= n - 1 - npml) a = (pos - (n - 1 - npml)) / npml;
        return (T)(quibar * (a - std::sin(2.0 * M_PI * a) / (2.0 * M_PI)));
    };
    for (int i = 0; i < n1; ++i) z[i] = zeta(i, n1, sides & 1, sides & 2), z[n1 + n2 + i] = zeta(i + 0.5, n1, sides & 1, sides & 2);
    for (int i = 0; i < n2; ++i) z[n1 + i] = zeta(i, n2, sides & 4, sides & 8), z[2 * n1 + n2 + i] = zeta(i + 0.5, n2, sides & 4, sides & 8);
    if (edges) {   // the notebook's staggered profiles repeat their last entry (cells 72-74)
        z[n1 + n2 + n1 - 1] = z[n1 + n2 + n1 - 2];
        z[2 * n1 + n2 + n2 - 1] = z[2 * n1 + n2 + n2 - 2];
    }
    MIFWI_HIP(hipMemcpyAsync(p.pml_zeta, z.data(), z.size() * sizeof(T), hipMemcpyHostToDevice, p.stream));
    int rc;
    if (vp_staggered) {
        // squared, in the haloed layout of a field level
        if (!p.pml_c2s && (rc = dev_alloc(p, &p.pml_c2s, (size_t)p.L.field * sizeof(T), true))) return rc;
        if ((rc = copy_any(p, p.d_dense, vp_staggered, (size_t)p.L.ngrid * sizeof(T)))) return rc;
        if ((rc = launch_pack<T>(p, (const T *)p.d_dense, (T *)p.pml_c2s, p.L.field, 1))) return rc;
        if ((rc = launch_square<T>(p, (T *)p.pml_c2s, p.L.field))) return rc;
    } else if (p.pml_c2s) {
        dev_free(p.pml_c2s);
        p.pml_c2s = nullptr;
    }
    MIFWI_HIP(hipStreamSynchronize(p.stream));
    clear_graphs(p);
    p.pml_sides = sides, p.pml_edges = edges, p.pml_npml = npml;
    p.have_pml = true;
    return MIFWI_OK;
}

// forward time loop of a MIFWI_FLAG_PML plan: per step the displacement update, the sparse terms (source into
// u[t+1], receivers sample u[t]) and the auxiliary-field update, in the order the reference's notebook gives its
// operator (03_pml.ipynb cell 85)
template <typename T>
static int run_forward_pml(Plan &p, int nshots)
{
    if (!p.have_pml) return fail(MIFWI_ESTATE, "mifwi_set_pml has not been called");
    const Layout &L = p.L;
    const int nt = p.cfg.nt;
    T *base = (T *)p.rot_u;
    const long shot_stride = 3L * L.field, phi_stride = 4L * L.field;
    MIFWI_HIP(hipMemsetAsync(base, 0, (size_t)nshots * shot_stride * sizeof(T), p.stream));
    MIFWI_HIP(hipMemsetAsync(p.pml_phi, 0, (size_t)nshots * phi_stride * sizeof(T), p.stream));
    MIFWI_HIP(hipMemsetAsync(p.d_rec, 0, (size_t)nshots * nt * p.cfg.nrec * sizeof(T), p.stream));
    p.rot_u_valid = false;
    const StepConsts<T> k = make_consts<T>(p);
    const T *z = (const T *)p.pml_zeta;
    const int n1 = L.n[1], n2 = L.n[2];
    T *phi = (T *)p.pml_phi;
    auto enqueue = [&](int &launches) -> int {
        for (int time = 1; time <= nt - 2; ++time) {
            PmlPtrs<T> q{};
            q.cur = base + (long)(time % 3) * L.field, q.old = base + (long)((time + 2) % 3) * L.field;
            q.out = base + (long)((time + 1) % 3) * L.field;
            q.shot_stride = shot_stride, q.m = (const T *)p.m;
            q.zx0 = z, q.zz0 = z + n1, q.zx1 = z + n1 + n2, q.zz1 = z + 2 * n1 + n2;
            const long in = (long)(time & 1) * L.field, on = (long)((time + 1) & 1) * L.field;
            q.phi1 = phi + in, q.phi2 = phi + 2 * L.field + in;
            q.phi1_out = phi + on, q.phi2_out = phi + 2 * L.field + on;
            q.phi_shot_stride = phi_stride;
            q.ihx = (T)(1.0 / p.cfg.spacing[0]), q.ihz = (T)(1.0 / p.cfg.spacing[1]);
            q.sides = p.pml_sides, q.npml = p.pml_npml, q.edges = p.pml_edges, q.c2s = (const T *)p.pml_c2s;
            int r;
            if ((r = launch_pml_u<T>(p, k, q, nshots))) return r;
            SparsePtrs<T> sq{};
            sq.inj = &p.src, sq.field_inj = q.out, sq.vals_inj = (const T *)p.d_wavelet;
            sq.itp = &p.rec, sq.field_int = q.cur, sq.vals_out = (T *)p.d_rec;
            sq.shot_stride = shot_stride, sq.m = (const T *)p.m, sq.time = time, sq.nt = nt, sq.dt2 = k.dt2;
            if ((r = launch_sparse<T>(p, sq, nshots))) return r;
            if (p.pml_edges) {
                if ((r = launch_pml_edges<T>(p, k, q.out, shot_stride, nshots))) return r;
                launches += 1;
            }
            if ((r = launch_pml_phi<T>(p, k, q, nshots))) return r;
            launches += 3;
        }
        return MIFWI_OK;
    };
    int launches = 0, rc;
    float ms = 0;
    const auto key = std::make_tuple(8, nshots, 0, 0, 0);
    if ((rc = timed_loop(p, &key, enqueue, launches, ms))) return rc;
    // displacement 16 B (u[t], u[t-1], m in, u[t+1] out) + in the layer the two auxiliary fields in and out
    fill_stats(p, ms, nt - 2, nshots, 16.0, launches, false);
    p.rot_u_valid = true;
    return MIFWI_OK;
}

// Least-squares misfit of data already modelled (misfit/misfit.py:5-9): residual = syn - obs,
// fval[shot] = 0.5 * ||residual||^2.  Host arrays are staged through the plan's data buffers.
template <typename T>
static int l2_misfit(Plan &p, int nshots, const void *syn, const void *obs, void *residual_out, double *fval)
{
    if (nshots < 1 || nshots > p.cfg.max_shots) return fail(MIFWI_EINVAL, "nshots must be in 1..max_shots");
    if (!syn || !obs) return fail(MIFWI_EINVAL, "syn or obs is null");
    const long n = (long)p.cfg.nt * p.cfg.nrec;
    const size_t bytes = (size_t)nshots * n * sizeof(T);
    int rc;
    const T *s = (const T *)syn, *o = (const T *)obs;
    if (!is_device_ptr(syn)) {
        if ((rc = copy_any(p, p.d_rec, syn, bytes))) return rc;
        s = (const T *)p.d_rec;
    }
    if (!is_device_ptr(obs)) {
        if ((rc = copy_any(p, p.d_res, obs, bytes))) return rc;
        o = (const T *)p.d_res;
    }
    T *r = (residual_out && is_device_ptr(residual_out)) ? (T *)residual_out : (T *)p.d_res;
    if ((rc = launch_l2_misfit<T>(p, s, o, r, n, nshots, p.d_partial, p.d_fval))) return rc;
    if (residual_out && r != residual_out && (rc = copy_any(p, residual_out, r, bytes))) return rc;
    if (fval && (rc = copy_any(p, fval, p.d_fval, (size_t)nshots * sizeof(double)))) return rc;
    return finish(p);
}

// ---- tail of the multi-shot objective (fwi.py:164-171,190-204) -------------------------------------
static int reduce_buffer(Plan &p)
{
    if (p.d_reduce) return MIFWI_OK;
    long ngrid = 1;
    for (int d = 0; d < p.cfg.ndim; ++d) ngrid *= p.cfg.shape[d];
    return dev_alloc(p, (void **)&p.d_reduce, (size_t)3 * ngrid * sizeof(double), false);
}

template <typename T>
static int sum_shots(Plan &p, int nshots, const void *grad, const void *illum, const double *mute, const int32_t *lo,
                     const int32_t *phys, double *g_sum, double *il_sum)
{
    if (nshots < 1) return fail(MIFWI_EINVAL, "nshots must be at least 1");
    if (!grad || !g_sum || !lo || !phys) return fail(MIFWI_EINVAL, "null argument");
    if ((illum == nullptr) != (il_sum == nullptr)) return fail(MIFWI_EINVAL, "illum and il_sum go together");
    if (!is_device_ptr(grad) || (illum && !is_device_ptr(illum)) || (mute && !is_device_ptr(mute)))
        return fail(MIFWI_EINVAL, "mifwi_sum_shots reads device arrays");
    CropMap c{};
    c.nd = p.cfg.ndim;
    c.field = 1;
    c.nphys = 1;
    for (int d = c.nd - 1; d >= 0; --d) {
        if (lo[d] < 0 || phys[d] < 1 || lo[d] + phys[d] > p.cfg.shape[d])
            return fail(MIFWI_EINVAL, "physical window outside the padded grid");
        c.lo[d] = lo[d];
        c.phys[d] = phys[d];
        c.pad_stride[d] = c.field;
        c.field *= p.cfg.shape[d];
        c.nphys *= phys[d];
    }
    int rc = reduce_buffer(p);
    if (rc) return rc;
    double *gs = is_device_ptr(g_sum) ? g_sum : p.d_reduce;
    double *ils = !il_sum ? nullptr : (is_device_ptr(il_sum) ? il_sum : p.d_reduce + c.field);
    if ((rc = launch_mute_sum<T>(p.stream, c, nshots, (const T *)grad, (const T *)illum, mute, gs, ils))) return rc;
    if (gs != g_sum && (rc = copy_any(p, g_sum, gs, (size_t)c.nphys * sizeof(double)))) return rc;
    if (il_sum && ils != il_sum && (rc = copy_any(p, il_sum, ils, (size_t)c.nphys * sizeof(double)))) return rc;
    return finish(p);
}

static int precondition(Plan &p, long n, const double *g, const double *il, const double *mask, double *out)
{
    if (n < 0 || !g || !out) return fail(MIFWI_EINVAL, "bad argument");
    long ngrid = 1;
    for (int d = 0; d < p.cfg.ndim; ++d) ngrid *= p.cfg.shape[d];
    if (n > ngrid) return fail(MIFWI_EINVAL, "n exceeds the padded grid");
    if (!is_device_ptr(g) || (il && !is_device_ptr(il)) || (mask && !is_device_ptr(mask)))
        return fail(MIFWI_EINVAL, "mifwi_precondition reads device arrays");
    int rc = reduce_buffer(p);
    if (rc) return rc;
    double *o = is_device_ptr(out) ? out : p.d_reduce + 2 * ngrid;
    if (n > 0 && (rc = launch_precondition(p.stream, n, g, il, mask, o))) return rc;
    if (o != out && (rc = copy_any(p, out, o, (size_t)n * sizeof(double)))) return rc;
    return finish(p);
}

// ---- the entry point that allocates nothing: its body (scratch array and address look-up: top of this file) ---
template <typename T>
static int l2_misfit_noalloc(int nshots, long n, const void *syn, const void *obs, void *res, double *fval,
                             hipStream_t stream)
{
    int dev = 0;
    MIFWI_HIP(hipGetDevice(&dev));
    double *partial = nullptr;
    int rc = scratch_of_current_device(dev, &partial);
    if (rc) return rc;
    // every partial sum of the shots of this call is written by the first kernel before the second reads it:
    // what the previous call left in the array needs no clearing
    return launch_l2_misfit_on<T>(stream, (const T *)syn, (const T *)obs, (T *)res, n, nshots, partial, fval);
}

// ---- TTI ---------------------------------------------------------------------------------------
// Can 2-D TTI time loops of this plan run as one launch each (kernels_persist_tti2d.hip)?  Decided once, at the first
// mifwi_tti_set_model, because it fixes the tile shape: float32, space order 4 or 8, the full GPU with its 8 XCDs,
// and every 32 x 32 tile of a shot resident on one XCD for each of the three variants of the kernel.
static bool tti_whole_loop_possible(Plan &p)
{
    if (!p.persist || p.kind != 1 || p.esize != 4 || p.xcds != 8 || p.ncu % 8 != 0) return false;
    if (p.L.R != 2 && p.L.R != 4) return false;
    if (p.cfg.flags & (MIFWI_FLAG_ASYNC | MIFWI_FLAG_NO_GRAPH | MIFWI_FLAG_GENERIC)) return false;
    if (p.cfg.nt < 3 || (unsigned long long)p.L.field * 12ull >= (1ull << 31)) return false;
    if (const char *env = getenv("MIFWI_TTI_PERSIST"))
        if (atoi(env) == 0) return false;
    const int ntiles = ((p.L.n[1] + 31) / 32) * ((p.L.n[2] + 31) / 32);
    for (int mode = 0; mode < 3; ++mode) {
        if (p.persist_tti_bpc[mode] < 0) p.persist_tti_bpc[mode] = persist_tti2d_blocks_per_cu(p.L.R, mode);
        if (p.persist_tti_bpc[mode] * (p.ncu / 8) < ntiles) return false;
    }
    return true;
}

// mode 0 forward, 1 adjoint, 2 gradient; `sampled`: the point set the loop interpolates, or null
static bool tti_persist_eligible(Plan &p, int mode, const SparseSet *sampled)
{
    if (!p.tti_whole_loop || !(p.tile_tx == 32 && p.tile_tz == 32)) return false;
    if (sampled && sampled->max_own > 64) return false;   // a tile samples at most 64 points on this path
    if (p.persist_skip > 0) {
        --p.persist_skip;
        return false;
    }
    (void)mode;
    return true;
}

// Queue the launches that run a whole TTI loop (groups of 8 shots, one shot per XCD), the check of the tiles' step
// counts behind each, and the read-back of the abort flag.  fu / fv: rotating levels [shots][3][field] of the two
// fields; save_u / save_v: [shots][nt][field] or null; su / sv: saved background fields of the gradient loop or null.
template <typename T>
static int enqueue_persist_tti(Plan &p, int nshots, int mode, float *fu, float *fv, float *save_u, float *save_v,
                               const float *su, const float *sv, const TtiParams<T> &tq, int &launches)
{
    const Layout &L = p.L;
    const int ntiles = p.ntx * p.ntz;
    const size_t nsync = (size_t)8 * ntiles + 16 + 4;
    int rc;
    if (!p.d_psync) {
        if ((rc = dev_alloc(p, (void **)&p.d_psync, nsync * sizeof(int), true))) return rc;
        MIFWI_HIP(hipHostMalloc((void **)&p.h_abort, sizeof(int), hipHostMallocDefault));
    }
    MIFWI_HIP(hipMemsetAsync(p.d_psync, 0, nsync * sizeof(int), p.stream));
    const size_t ngroup_words = (size_t)8 * ntiles + 16;
    PersistTtiArgs q{};
    q.a = base2d(p, nshots);
    if (mode == 0) {
        set_inject(q.a, p.src, p);
        set_interp(q.a, p.rec, p.d_rec);
    } else {
        set_inject(q.a, p.rec, p);
        if (mode == 1) set_interp(q.a, p.src, p.d_srca);
    }
    q.et = (const float *)tq.et, q.dl = (const float *)tq.dl, q.ct = (const float *)tq.ct, q.st = (const float *)tq.st;
    for (int j = 1; j <= tq.p; ++j) q.a1[j] = (float)tq.a[j];
    q.ihx = (float)tq.ih[1], q.ihz = (float)tq.ih[2];
    q.rot_u = fu, q.rot_v = fv;
    const long S = p.cfg.max_shots;
    q.comb_a = (float *)p.tti_tmp, q.comb_b = (float *)p.tti_tmp + 2 * S * L.field;
    q.field = L.field;
    q.forward = mode == 0 ? 1 : 0;
    q.nsteps = p.cfg.nt - 2;
    q.save_u = save_u, q.save_v = save_v, q.save_shot_stride = (long)p.cfg.nt * L.field;
    q.su = su, q.sv = sv, q.s_shot_stride = (long)p.cfg.nt * L.field;
    q.grad = (float *)p.d_grad, q.g_shot_stride = L.field;
    q.done = p.d_psync;
    q.counter = p.d_psync + 8 * ntiles;
    q.abort_flag = q.counter + 16;
    q.spin_limit = p.persist_spin;
    q.drop_slot = -1;
    if (const char *env = getenv("MIFWI_PERSIST_DROP_SLOT")) q.drop_slot = atoi(env);   // tests
    const int resident = p.persist_tti_bpc[mode] * (p.ncu / 8);
    launches = 0;
    for (int s0 = 0; s0 < nshots; s0 += 8) {
        if (s0 > 0) MIFWI_HIP(hipMemsetAsync(p.d_psync, 0, ngroup_words * sizeof(int), p.stream));
        q.a.shot0 = s0;
        q.a.nshots = std::min(8, nshots - s0);
        // blocks beyond the tiles leave at once; tests raise their number so that the tiles of a small grid share CUs
        int surplus = 16;
        if (const char *env = getenv("MIFWI_TTI_SURPLUS_BLOCKS")) surplus = std::max(0, atoi(env));
        if ((rc = launch_persist_tti2d(p, q, mode, std::min(resident, ntiles + surplus), p.stream))) return rc;
        if (p.persist_check && (rc = launch_persist2d_check(q.done, q.a.nshots * ntiles, q.nsteps, q.abort_flag, p.stream))) return rc;
        launches += 1;
    }
    MIFWI_HIP(hipMemcpyAsync(p.h_abort, q.abort_flag, sizeof(int), hipMemcpyDeviceToHost, p.stream));
    return MIFWI_OK;
}

template <typename T>
static int tti_set_model(Plan &p, const void *eps, const void *delta, const void *theta, const void *phi)
{
    if (int e = no_pml(p)) return e;
    if (p.cfg.ndim < 2) return fail(MIFWI_EINVAL, "TTI needs a 2-D or 3-D grid");
    if (p.cfg.space_order % 4 || p.cfg.space_order > 2 * kMaxR) return fail(MIFWI_EINVAL, "TTI needs a space_order of 4, 8, 12 or 16");
    if (p.cfg.flags & (MIFWI_FLAG_FREE_SURFACE | MIFWI_FLAG_OT4))
        return fail(MIFWI_EINVAL, "TTI is not available with a free surface or the OT4 kernel");
    if (!eps || !delta || !theta) return fail(MIFWI_EINVAL, "epsilon, delta and theta are required");
    int rc;
    if (p.kind == 1 && !p.tti_par && !getenv("MIFWI_2D_TX") && tti_whole_loop_possible(p)) {
        // whole time loops in one launch (kernels_persist_tti2d.hip): that kernel works on 32 x 32 tiles
        p.tti_whole_loop = true;
        if (!(p.tile_tx == 32 && p.tile_tz == 32)) {
            clear_graphs(p);
            p.tile_tx = 32, p.tile_tz = 32;
            p.ntx = (p.L.n[1] + 31) / 32, p.ntz = (p.L.n[2] + 31) / 32;
            if (p.have_geometry) {   // the sparse tables are per tile: rebuild them from the kept coordinates
                if ((rc = build_sparse<T>(p, p.src, (const T *)p.geo_src.data(), p.cfg.nsrc, p.geom_shots, 1))) return rc;
                if ((rc = build_sparse<T>(p, p.rec, (const T *)p.geo_rec.data(), p.cfg.nrec,
                                          p.geo_rec_per_shot ? p.geom_shots : 1, p.geo_rec_per_shot ? 1 : 0)))
                    return rc;
            }
        }
    } else if (p.kind == 1 && !p.tti_par && !getenv("MIFWI_2D_TX")) {
        // The TTI kernel holds two fields and four product arrays of a tile in LDS, so it runs best on small
        // tiles: 8 x 32 was the fastest shape at space orders up to 8 for every batch size, and at higher
        // orders up to about three 32 x 32 tiles per CU (measured on MI355X, Marmousi-II grid: 2 shots at
        // order 8 10.1 -> 6.4 us per step, at order 16 12.8 -> 8.8; at order 16 8 shots 15.8 -> 15.2, 16 shots
        // 25.1 -> 25.9, 32 shots 41.9 -> 48.1, hence the condition).  The acoustic kernels of the same plan lose
        // <= 3 % on that shape.
        const int t32 = ((p.L.n[1] + 31) / 32) * ((p.L.n[2] + 31) / 32);
        if ((p.L.R <= 4 || 8L * p.cfg.max_shots * t32 <= 25L * p.ncu) && !(p.tile_tx == 8 && p.tile_tz == 32)) {
            clear_graphs(p);
            p.tile_tx = 8, p.tile_tz = 32;
            p.ntx = (p.L.n[1] + 7) / 8, p.ntz = (p.L.n[2] + 31) / 32;
            if (p.have_geometry) {   // the sparse tables are per tile: rebuild them from the kept coordinates
                if ((rc = build_sparse<T>(p, p.src, (const T *)p.geo_src.data(), p.cfg.nsrc, p.geom_shots, 1))) return rc;
                if ((rc = build_sparse<T>(p, p.rec, (const T *)p.geo_rec.data(), p.cfg.nrec,
                                          p.geo_rec_per_shot ? p.geom_shots : 1, p.geo_rec_per_shot ? 1 : 0)))
                    return rc;
            }
        }
    }
    const size_t fb = (size_t)p.L.field * sizeof(T), S = (size_t)p.cfg.max_shots;
    if (!p.tti_par) {
        if ((rc = dev_alloc(p, &p.tti_par, 6 * fb, true))) return rc;
        if ((rc = dev_alloc(p, &p.tti_fwd, 2 * S * 3 * fb, true))) return rc;
        if ((rc = dev_alloc(p, &p.tti_adj, 2 * S * 3 * fb, true))) return rc;
        if ((rc = dev_alloc(p, &p.tti_tmp, 4 * S * fb, true))) return rc;
    }
    // float32 3-D plans step with the marching kernels when their tables fit beside everything else: the direction
    // cosine products (2 fields) and, for the form with two marches, Gz of both fields of every shot
    p.tti_two_pass = getenv("MIFWI_TTI3D_TWOPASS") != nullptr;
    if (p.L.ndim == 3 && sizeof(T) == 4 && p.L.R <= 8 && !(p.cfg.flags & MIFWI_FLAG_GENERIC) && !getenv("MIFWI_TTI3D_UNTILED")) {
        const bool need_gz = p.L.R > 4 || p.tti_two_pass;
        const size_t want = (p.tti_w ? 0 : 2 * fb) + (need_gz && !p.tti_gz3 ? 2 * S * fb : 0);
        size_t free_b = 0, total_b = 0;
        if (want && hipMemGetInfo(&free_b, &total_b) == hipSuccess && want < free_b / 2) {
            if (!p.tti_w && (rc = dev_alloc(p, &p.tti_w, 2 * fb, true))) return rc;
            if (need_gz && !p.tti_gz3 && (rc = dev_alloc(p, &p.tti_gz3, 2 * S * fb, true))) return rc;
        }
    }
    const StepConsts<T> k = make_consts<T>(p);
    T *par = (T *)p.tti_par;
    T *dense = (T *)p.d_dense;
    const size_t gb = (size_t)p.L.ngrid * sizeof(T);
    // field order: et, dl, ct, st, cp, sp
    struct { const void *src; int mode; T halo; } spec[6] = {{eps, 1, T(1)},   {delta, 2, T(1)}, {theta, 3, T(1)},
                                                              {theta, 4, T(0)}, {phi, 3, T(1)},   {phi, 4, T(0)}};
    for (int f = 0; f < 6; ++f) {
        const T *src = nullptr;
        if (spec[f].src) {
            if ((rc = copy_any(p, dense, spec[f].src, gb))) return rc;
            src = dense;
        }
        if ((rc = launch_tti_param<T>(p, k, src, par + (long)f * p.L.field, spec[f].mode, spec[f].halo))) return rc;
    }
    if (p.tti_w && (rc = launch_tti3d_weights(p))) return rc;
    if (p.tti_ck_K > 0) p.tti_save_valid = false;   // checkpoints belong to the parameters they were taken with
    return finish(p);   // wavefields of earlier runs stay readable: they are not touched here
}

template <typename T>
static TtiParams<T> tti_params(const Plan &p)
{
    TtiParams<T> q{};
    const T *par = (const T *)p.tti_par;
    q.et = par, q.dl = par + p.L.field, q.ct = par + 2 * p.L.field, q.st = par + 3 * p.L.field;
    q.cp = par + 4 * p.L.field, q.sp = par + 5 * p.L.field;
    q.p = p.L.R / 2;
    for (int j = 1; j <= q.p; ++j) {   // a_j = (-1)^(j+1) (p!)^2 / (j (p-j)! (p+j)!)
        double num = 1.0, den = (double)j;
        for (int i = 1; i <= q.p; ++i) num *= (double)i * (double)i;
        for (int i = 1; i <= q.p - j; ++i) den *= (double)i;
        for (int i = 1; i <= q.p + j; ++i) den *= (double)i;
        q.a[j] = (T)((j % 2 ? 1.0 : -1.0) * num / den);
    }
    for (int d = 0; d < 3; ++d) q.ih[d] = T(1);
    for (int d = 0; d < p.cfg.ndim; ++d) q.ih[3 - p.cfg.ndim + d] = (T)(1.0 / p.cfg.spacing[d]);
    return q;
}

// One TTI step of both fields of every shot: the step itself, injection of `inj` (values vals_in) into
// both new fields and sampling of the sum of the current fields at `itp` (either may be null).
template <typename T>
struct TtiLoop {
    Plan &p;
    int nshots;
    StepConsts<T> k;
    TtiParams<T> q;
    bool tiled;
    T *wA, *wB, *gzA, *gzB;

    TtiLoop(Plan &plan, int n) : p(plan), nshots(n), k(make_consts<T>(plan)), q(tti_params<T>(plan))
    {
        tiled = std::is_same<T, float>::value && tti2d_available(plan);
        const long S = plan.cfg.max_shots;
        T *tmp = (T *)plan.tti_tmp;
        wA = tmp, wB = tmp + S * plan.L.field, gzA = tmp + 2 * S * plan.L.field, gzB = tmp + 3 * S * plan.L.field;
    }
    // the tiled kernel reads precomputed injection terms: form them before the loop is enqueued
    int prepare(const SparseSet *inj, const T *vals_in)
    {
        MIFWI_HIP(hipMemsetAsync(p.tti_tmp, 0, (size_t)4 * p.cfg.max_shots * p.L.field * sizeof(T), p.stream));
        if (tiled && inj) return prepare_injection<T>(p, *inj, vals_in, nshots);
        return MIFWI_OK;
    }
    int step(bool fwd, const TtiLevels<T> &lv, const SparseSet *inj, const T *vals_in, const SparseSet *itp,
             T *vals_out, int time, int &launches, const TtiExtra<T> *extra = nullptr)
    {
        int r;
        if (tiled) {
            Step2DArgs sa{};
            if (inj) set_inject(sa, *inj, p);
            if (itp) set_interp(sa, *itp, vals_out);
            sa.time = time, sa.nt = p.cfg.nt;
            return launch_tti_step<T>(p, k, q, fwd, lv, wA, wB, gzA, gzB, nshots, launches, &sa, extra);
        }
        if ((r = launch_tti_step<T>(p, k, q, fwd, lv, wA, wB, gzA, gzB, nshots, launches, nullptr))) return r;
        if (extra && extra->b_un) {   // untiled path: the Born source terms are pointwise launches of their own
            if ((r = launch_born_source<T>(p, k, extra->b_un, extra->b_uc, extra->b_uo, extra->b_shot_stride, extra->b_dm,
                                           lv.uout, lv.shot_stride, nshots, 0)))
                return r;
            if ((r = launch_born_source<T>(p, k, extra->b_vn, extra->b_vc, extra->b_vo, extra->b_shot_stride, extra->b_dm,
                                           lv.vout, lv.shot_stride, nshots, 0)))
                return r;
            launches += 2;
        }
        if (extra && extra->grad) {   // literal three-level form of the correlation
            if ((r = launch_tti_grad<T>(p, k, extra->su, extra->sv, extra->level_stride, extra->s_shot_stride, lv.ucur,
                                        lv.vcur, lv.shot_stride, extra->grad, extra->g_shot_stride, nshots)))
                return r;
            launches += 1;
        }
        if (!inj && !itp) return MIFWI_OK;
        SparsePtrs<T> sq{};
        if (inj) sq.inj = inj, sq.field_inj = lv.uout, sq.field_inj2 = lv.vout, sq.vals_inj = vals_in;
        if (itp) sq.itp = itp, sq.field_int = lv.ucur, sq.field_int2 = lv.vcur, sq.vals_out = vals_out;
        sq.shot_stride = lv.shot_stride;
        sq.m = (const T *)p.m, sq.time = time, sq.nt = p.cfg.nt, sq.dt2 = k.dt2;
        if ((r = launch_sparse<T>(p, sq, nshots))) return r;
        launches += 1;
        return MIFWI_OK;
    }
};

// rotating levels of a field pair: level (time + shift) % 3
template <typename T>
static TtiLevels<T> tti_rot(T *fu, T *fv, long field, int time, bool fwd)
{
    const long c = (long)(time % 3) * field;
    const long nx = (long)((time + (fwd ? 1 : 2)) % 3) * field;
    const long od = (long)((time + (fwd ? 2 : 1)) % 3) * field;
    return TtiLevels<T>{fu + c, fu + od, fv + c, fv + od, fu + nx, fv + nx, 3L * field};
}

static int tti_check(Plan &p)
{
    if (!p.tti_par) return fail(MIFWI_ESTATE, "mifwi_tti_set_model has not been called");
    return MIFWI_OK;
}

// mode 0: forward (d_wavelet -> d_rec; save keeps every level of both fields), mode 1: adjoint (d_res -> d_srca)
template <typename T>
static int tti_run(Plan &p, int nshots, int mode, int save)
{
    int rc;
    if ((rc = tti_check(p))) return rc;
    const Layout &L = p.L;
    const int nt = p.cfg.nt;
    const long S = p.cfg.max_shots;
    if (save && !p.tti_save && !p.tti_ckpt) {
        // every level of both fields when that fits, else two-level checkpointing as for the acoustic wavefield
        // (prepare_save): two levels of u and v entering every tti_ck_K-th step, segments recomputed by the gradient
        const size_t need = (size_t)2 * S * nt * L.field * sizeof(T);
        size_t free_b = 0, total_b = 0;
        MIFWI_HIP(hipMemGetInfo(&free_b, &total_b));
        if (const char *env = getenv("MIFWI_SAVE_BUDGET_MB"))   // tests: pretend that less memory is free
            free_b = std::min(free_b, (size_t)std::max(1L, atol(env)) << 20);
        clear_graphs(p);
        if (!(p.cfg.flags & MIFWI_FLAG_CHECKPOINT) && (double)need <= 0.9 * (double)free_b) {
            if ((rc = dev_alloc(p, &p.tti_save, need, true))) return rc;
        } else {
            const int nsteps = nt - 2;
            int K = std::max(2, (int)std::ceil(std::sqrt(2.0 * nsteps)));
            K += K & 1;
            const int nseg = (nsteps + K - 1) / K;
            const size_t fb = (size_t)S * L.field * sizeof(T);
            const size_t need_ck = 2 * fb * (2 * (size_t)nseg + K + 2);
            if ((double)need_ck > 0.95 * (double)free_b)
                return fail(MIFWI_ENOMEM, "not enough device memory even for the checkpointed TTI wavefields (" +
                                              std::to_string(need_ck >> 20) + " MiB needed): use fewer shots per call");
            p.tti_ck_K = K, p.tti_ck_nseg = nseg;
            if ((rc = dev_alloc(p, &p.tti_ckpt, 2 * fb * 2 * nseg, true))) return rc;
            if ((rc = dev_alloc(p, &p.tti_seg, 2 * fb * (K + 2), true))) return rc;
            if (!p.d_wavelet_ck &&
                (rc = dev_alloc(p, &p.d_wavelet_ck, (size_t)S * nt * std::max(1, p.cfg.nsrc) * sizeof(T), true)))
                return rc;
        }
    }
    if (save) p.tti_save_valid = false;
    const bool ck = save && p.tti_ck_K > 0;
    if (ck) {
        MIFWI_HIP(hipMemcpyAsync(p.d_wavelet_ck, p.d_wavelet, (size_t)nshots * nt * std::max(1, p.cfg.nsrc) * sizeof(T),
                                 hipMemcpyDeviceToDevice, p.stream));
        p.tti_fwd_valid = false;
    }
    const bool in_save = save && !ck;
    const long stride = in_save ? (long)nt * L.field : 3L * L.field;
    T *fu = in_save ? (T *)p.tti_save : (T *)(mode == 0 ? p.tti_fwd : p.tti_adj);
    T *fv = fu + S * stride;
    if (!in_save) {
        MIFWI_HIP(hipMemsetAsync(fu, 0, (size_t)nshots * stride * sizeof(T), p.stream));
        MIFWI_HIP(hipMemsetAsync(fv, 0, (size_t)nshots * stride * sizeof(T), p.stream));
    }   // saved levels 0 and 1 and every halo are never written: they keep the zeros of the allocation
    const SparseSet &inj = mode == 0 ? p.src : p.rec, &itp = mode == 0 ? p.rec : p.src;
    const T *vals_in = (const T *)(mode == 0 ? p.d_wavelet : p.d_res);
    T *vals_out = (T *)(mode == 0 ? p.d_rec : p.d_srca);
    MIFWI_HIP(hipMemsetAsync(vals_out, 0, (size_t)nshots * nt * std::max(1, itp.npoint) * sizeof(T), p.stream));
    TtiLoop<T> loop(p, nshots);
    if ((rc = loop.prepare(&inj, vals_in))) return rc;
    if (!ck && loop.tiled && tti_persist_eligible(p, mode, &itp)) {
        // the whole loop as one launch per group of 8 shots: the fields step on their three rotating levels and, when
        // saving, stream a copy of every new level into the saved arrays (whose levels 0 and 1 nobody ever writes)
        T *ru = (T *)(mode == 0 ? p.tti_fwd : p.tti_adj), *rv = ru + S * 3 * L.field;
        if (in_save) {
            MIFWI_HIP(hipMemsetAsync(ru, 0, (size_t)nshots * 3 * L.field * sizeof(T), p.stream));
            MIFWI_HIP(hipMemsetAsync(rv, 0, (size_t)nshots * 3 * L.field * sizeof(T), p.stream));
        }
        float *sv_u = in_save ? (float *)p.tti_save : nullptr, *sv_v = in_save ? (float *)p.tti_save + S * (long)nt * L.field : nullptr;
        auto whole_loop = [&](int &n) -> int {
            return enqueue_persist_tti<T>(p, nshots, mode, (float *)ru, (float *)rv, sv_u, sv_v, nullptr, nullptr, loop.q, n);
        };
        int launches = 0;
        float ms = 0;
        if ((rc = timed_loop(p, nullptr, whole_loop, launches, ms, mode == 1))) return rc;
        if (*p.h_abort) {   // a wait ran out, or a tile had not counted all its steps: this loop anew, a kernel per step
            p.persist_skip = p.persist_backoff;
            p.persist_backoff = std::min(4096, 2 * p.persist_backoff);
            p.stats.persist_fallbacks += 1;
            return tti_run<T>(p, nshots, mode, save);
        }
        p.loop_persist_tti = true;
        fill_stats(p, ms, nt - 2, nshots, 2 * 20.0 + 6 * 4.0, launches, mode == 1);
        if (save) p.tti_save_valid = true, p.tti_save_shots = nshots;
        (mode == 0 ? p.tti_fwd_valid : p.tti_adj_valid) = true;
        return MIFWI_OK;
    }
    auto enqueue = [&](int &launches) -> int {
        for (int step = 1; step <= nt - 2; ++step) {
            const int time = mode == 0 ? step : nt - 1 - step;
            TtiLevels<T> lv = tti_rot<T>(fu, fv, L.field, time, mode == 0);
            int r;
            if (in_save) {
                const long c = (long)time * L.field;
                lv = TtiLevels<T>{fu + c, fu + c - L.field, fv + c, fv + c - L.field, fu + c + L.field, fv + c + L.field, stride};
            }
            if (ck && (time - 1) % p.tti_ck_K == 0) {
                // keep the two levels of u and of v that enter this segment: [field][shot][segment][2]
                const long cstride = (long)p.tti_ck_nseg * 2 * L.field;
                T *cu = (T *)p.tti_ckpt + (long)((time - 1) / p.tti_ck_K) * 2 * L.field, *cv = cu + S * cstride;
                if ((r = copy_levels<T>(p, cu, cstride, lv.uold, stride, 1, nshots))) return r;
                if ((r = copy_levels<T>(p, cu + L.field, cstride, lv.ucur, stride, 1, nshots))) return r;
                if ((r = copy_levels<T>(p, cv, cstride, lv.vold, stride, 1, nshots))) return r;
                if ((r = copy_levels<T>(p, cv + L.field, cstride, lv.vcur, stride, 1, nshots))) return r;
            }
            if ((r = loop.step(mode == 0, lv, &inj, vals_in, &itp, vals_out, time, launches))) return r;
        }
        return MIFWI_OK;
    };
    int launches = 0;
    float ms = 0;
    const auto key = std::make_tuple(4 + mode, nshots, save, 0, 0);
    if ((rc = timed_loop(p, ck ? nullptr : &key, enqueue, launches, ms, mode == 1))) return rc;
    fill_stats(p, ms, nt - 2, nshots, 2 * 20.0 + 6 * 4.0, launches, mode == 1);
    if (save) p.tti_save_valid = true, p.tti_save_shots = nshots;
    if (!in_save) (mode == 0 ? p.tti_fwd_valid : p.tti_adj_valid) = true;
    return MIFWI_OK;
}

template <typename T>
static int tti_forward(Plan &p, int nshots, const void *wavelet, int per_shot, void *rec_out, int save)
{
    int rc;
    if ((rc = check_run(p, nshots))) return rc;
    if ((rc = upload_wavelet<T>(p, nshots, wavelet, per_shot))) return rc;
    if ((rc = tti_run<T>(p, nshots, 0, save))) return rc;
    if (rec_out && (rc = copy_any(p, rec_out, p.d_rec, (size_t)nshots * p.cfg.nt * p.cfg.nrec * sizeof(T)))) return rc;
    return finish(p);
}

template <typename T>
static int tti_adjoint(Plan &p, int nshots, const void *rec_in, void *srca_out)
{
    int rc;
    if ((rc = check_run(p, nshots))) return rc;
    if (!rec_in) return fail(MIFWI_EINVAL, "rec_in is null");
    if ((rc = copy_any(p, p.d_res, rec_in, (size_t)nshots * p.cfg.nt * p.cfg.nrec * sizeof(T)))) return rc;
    if ((rc = tti_run<T>(p, nshots, 1, 0))) return rc;
    if (srca_out && (rc = copy_any(p, srca_out, p.d_srca, (size_t)nshots * p.cfg.nt * p.cfg.nsrc * sizeof(T)))) return rc;
    return finish(p);
}

// Linearised TTI modelling (seismic/tti/operators.py:529-587): background fields in tti_fwd, scattered
// fields in tti_adj, all four advanced in one time loop.
template <typename T>
static int tti_born(Plan &p, int nshots, const void *wavelet, int per_shot, const void *dm, void *rec_out)
{
    int rc;
    if ((rc = check_run(p, nshots))) return rc;
    if ((rc = tti_check(p))) return rc;
    if (!dm) return fail(MIFWI_EINVAL, "dm is null");
    if ((rc = upload_wavelet<T>(p, nshots, wavelet, per_shot))) return rc;
    const Layout &L = p.L;
    const int nt = p.cfg.nt;
    const long S = p.cfg.max_shots, stride = 3L * L.field;
    if (!p.d_dm && (rc = dev_alloc(p, &p.d_dm, (size_t)L.field * sizeof(T), true))) return rc;
    if ((rc = copy_any(p, p.d_dense, dm, (size_t)L.ngrid * sizeof(T)))) return rc;
    if ((rc = launch_pack<T>(p, (const T *)p.d_dense, (T *)p.d_dm, L.field, 1))) return rc;
    T *bu = (T *)p.tti_fwd, *bv = bu + S * stride, *dU = (T *)p.tti_adj, *dV = dU + S * stride;
    for (T *f : {bu, bv, dU, dV}) MIFWI_HIP(hipMemsetAsync(f, 0, (size_t)nshots * stride * sizeof(T), p.stream));
    MIFWI_HIP(hipMemsetAsync(p.d_rec, 0, (size_t)nshots * nt * std::max(1, p.cfg.nrec) * sizeof(T), p.stream));
    p.tti_fwd_valid = p.tti_adj_valid = false;
    TtiLoop<T> loop(p, nshots);
    if ((rc = loop.prepare(&p.src, (const T *)p.d_wavelet))) return rc;
    auto enqueue = [&](int &launches) -> int {
        for (int time = 1; time <= nt - 2; ++time) {
            const TtiLevels<T> b = tti_rot<T>(bu, bv, L.field, time, true), d = tti_rot<T>(dU, dV, L.field, time, true);
            int r;
            if ((r = loop.step(true, b, &p.src, (const T *)p.d_wavelet, nullptr, nullptr, time, launches))) return r;
            // qu = -dm u0.dt2, qv = -dm v0.dt2 (u0.dt2 after the injection into u0[t+1])
            TtiExtra<T> x{};
            x.b_un = b.uout, x.b_uc = b.ucur, x.b_uo = b.uold, x.b_vn = b.vout, x.b_vc = b.vcur, x.b_vo = b.vold;
            x.b_shot_stride = stride, x.b_dm = (const T *)p.d_dm;
            if ((r = loop.step(true, d, nullptr, nullptr, &p.rec, (T *)p.d_rec, time, launches, &x))) return r;
        }
        return MIFWI_OK;
    };
    int launches = 0;
    float ms = 0;
    const auto key = std::make_tuple(6, nshots, 0, 0, 0);
    if ((rc = timed_loop(p, &key, enqueue, launches, ms))) return rc;
    fill_stats(p, ms, nt - 2, nshots, 2 * (2 * 20.0 + 6 * 4.0) + 2 * 16.0, launches, false);
    p.tti_fwd_valid = p.tti_adj_valid = true;
    if (rec_out && (rc = copy_any(p, rec_out, p.d_rec, (size_t)nshots * nt * p.cfg.nrec * sizeof(T)))) return rc;
    return finish(p);
}

// Adjoint of the TTI Jacobian (seismic/tti/operators.py:590-639) on the levels kept by tti_forward(save).
template <typename T>
static int tti_gradient(Plan &p, int nshots, const void *residual, void *grad, int accumulate)
{
    int rc;
    if ((rc = check_run(p, nshots))) return rc;
    if ((rc = tti_check(p))) return rc;
    if (!residual || !grad) return fail(MIFWI_EINVAL, "residual or grad is null");
    if (!p.tti_save_valid || p.tti_save_shots < nshots)
        return fail(MIFWI_ESTATE, "no saved TTI wavefields: call mifwi_tti_forward with save = 1 first");
    const Layout &L = p.L;
    const int nt = p.cfg.nt;
    const long S = p.cfg.max_shots, stride = 3L * L.field, sstride = (long)nt * L.field;
    if ((rc = copy_any(p, p.d_res, residual, (size_t)nshots * nt * p.cfg.nrec * sizeof(T)))) return rc;
    if (!p.d_grad && (rc = dev_alloc(p, &p.d_grad, (size_t)S * L.field * sizeof(T), true))) return rc;
    MIFWI_HIP(hipMemsetAsync(p.d_grad, 0, (size_t)nshots * L.field * sizeof(T), p.stream));
    T *bp = (T *)p.tti_adj, *br = bp + S * stride;
    MIFWI_HIP(hipMemsetAsync(bp, 0, (size_t)nshots * stride * sizeof(T), p.stream));
    MIFWI_HIP(hipMemsetAsync(br, 0, (size_t)nshots * stride * sizeof(T), p.stream));
    p.tti_adj_valid = false;
    const T *su = (const T *)p.tti_save, *sv = su + S * sstride;
    const bool ck = p.tti_ck_K > 0;
    TtiLoop<T> loop(p, nshots);
    if ((rc = loop.prepare(&p.rec, (const T *)p.d_res))) return rc;
    if (ck && loop.tiled && (rc = prepare_injection<T>(p, p.src, p.d_wavelet_ck, nshots))) return rc;   // recomputation
    if (!ck && loop.tiled && tti_persist_eligible(p, 2, nullptr)) {
        auto whole_loop = [&](int &n) -> int {
            return enqueue_persist_tti<T>(p, nshots, 2, (float *)bp, (float *)br, nullptr, nullptr, (const float *)su, (const float *)sv,
                                          loop.q, n);
        };
        int launches = 0;
        float ms = 0;
        if ((rc = timed_loop(p, nullptr, whole_loop, launches, ms, true))) return rc;
        if (*p.h_abort) {
            p.persist_skip = p.persist_backoff;
            p.persist_backoff = std::min(4096, 2 * p.persist_backoff);
            p.stats.persist_fallbacks += 1;
            return tti_gradient<T>(p, nshots, residual, grad, accumulate);
        }
        p.loop_persist_tti = true;
        fill_stats(p, ms, nt - 2, nshots, 2 * 20.0 + 6 * 4.0 + 2 * 4.0 + 8.0, launches, true);
        p.tti_adj_valid = true;
        if ((rc = deliver_grid<T>(p, p.d_grad, grad, nshots, accumulate))) return rc;
        return finish(p);
    }
    auto enqueue = [&](int &launches) -> int {
        int r;
        if (!ck) {
            for (int time = nt - 2; time >= 1; --time) {
                const TtiLevels<T> a = tti_rot<T>(bp, br, L.field, time, false);
                TtiExtra<T> x{};
                x.su = su + (long)time * L.field, x.sv = sv + (long)time * L.field, x.level_stride = L.field;
                x.s_shot_stride = sstride, x.grad = (T *)p.d_grad, x.g_shot_stride = L.field;
                if ((r = loop.step(false, a, &p.rec, (const T *)p.d_res, nullptr, nullptr, time, launches, &x))) return r;
            }
            return MIFWI_OK;
        }
        // checkpointed: walk the segments backwards; each is propagated again into the segment buffers
        // ([field][shot][K + 2 levels], level i of a segment = time t0 - 1 + i) before the backward steps cross it
        const int K = p.tti_ck_K;
        const long seg_stride = (long)(K + 2) * L.field, cstride = (long)p.tti_ck_nseg * 2 * L.field;
        T *gu = (T *)p.tti_seg, *gv = gu + S * seg_stride;
        for (int j = p.tti_ck_nseg - 1; j >= 0; --j) {
            const int t0 = 1 + j * K, t1 = std::min(nt - 2, t0 + K - 1);
            const T *cu = (const T *)p.tti_ckpt + (long)j * 2 * L.field, *cv = cu + S * cstride;
            if ((r = copy_levels<T>(p, gu, seg_stride, cu, cstride, 2, nshots))) return r;
            if ((r = copy_levels<T>(p, gv, seg_stride, cv, cstride, 2, nshots))) return r;
            for (int time = t0; time <= t1; ++time) {
                const long c = (long)(time - t0 + 1) * L.field;
                const TtiLevels<T> lv{gu + c, gu + c - L.field, gv + c, gv + c - L.field, gu + c + L.field, gv + c + L.field,
                                      seg_stride};
                if ((r = loop.step(true, lv, &p.src, (const T *)p.d_wavelet_ck, nullptr, nullptr, time, launches))) return r;
            }
            for (int time = t1; time >= t0; --time) {
                const TtiLevels<T> a = tti_rot<T>(bp, br, L.field, time, false);
                const long c = (long)(time - t0 + 1) * L.field;
                TtiExtra<T> x{};
                x.su = gu + c, x.sv = gv + c, x.level_stride = L.field;
                x.s_shot_stride = seg_stride, x.grad = (T *)p.d_grad, x.g_shot_stride = L.field;
                if ((r = loop.step(false, a, &p.rec, (const T *)p.d_res, nullptr, nullptr, time, launches, &x))) return r;
            }
        }
        return MIFWI_OK;
    };
    int launches = 0;
    float ms = 0;
    const auto key = std::make_tuple(7, nshots, 0, 0, 0);
    if ((rc = timed_loop(p, ck ? nullptr : &key, enqueue, launches, ms, true))) return rc;
    fill_stats(p, ms, nt - 2, nshots, 2 * 20.0 + 6 * 4.0 + 2 * 4.0 + 8.0, launches, true);
    p.tti_adj_valid = true;
    if ((rc = deliver_grid<T>(p, p.d_grad, grad, nshots, accumulate))) return rc;
    return finish(p);
}

// levels first .. first+count-1 of one shot of the half-precision saved wavefield, as float
static int read_half_levels(Plan &p, int shot, int first, int count, float *out)
{
    int rc;
    void *tmp = nullptr;   // one haloed float level
    if ((rc = dev_alloc(p, &tmp, (size_t)p.L.field * sizeof(float), false))) return rc;
    const long lstride = (long)p.cfg.max_shots * p.L.field;
    const bool dev = is_device_ptr(out);
    for (int l = 0; l < count && !rc; ++l) {
        const char *src = (const char *)p.saved + ((size_t)(first + l) * lstride + (size_t)shot * p.L.field) * 2;
        float *dst = out + (long)l * p.L.ngrid;
        rc = launch_half_to_float(p, src, (float *)tmp, p.L.field, (const float *)p.d_save_scale, shot);
        if (!rc) rc = launch_unpack<float>(p, (const float *)tmp, p.L.field, dev ? dst : (float *)p.d_dense, 1, 0);
        if (!rc && !dev) rc = copy_any(p, dst, p.d_dense, (size_t)p.L.ngrid * sizeof(float));
        if (!rc && hipStreamSynchronize(p.stream) != hipSuccess) rc = fail(MIFWI_EHIP, "stream synchronisation failed");
    }
    dev_free(tmp);
    return rc ? rc : finish(p);
}

template <typename T>
static int read_wavefield(Plan &p, int which, int shot, int first, int count, void *out)
{
    if (!out) return fail(MIFWI_EINVAL, "out is null");
    if (shot < 0 || shot >= p.cfg.max_shots) return fail(MIFWI_EINVAL, "shot out of range");
    const T *base;
    int nlev;
    long lstride = p.L.field;   // elements between consecutive levels of one shot
    if (which == 0) {
        if (!p.saved_valid || shot >= p.saved_shots) return fail(MIFWI_ESTATE, "no saved wavefield for this shot");
        if (p.ck_K > 0) return fail(MIFWI_ESTATE, "the forward wavefield is checkpointed, not stored level by level");
        if (first < 0 || count < 0 || first + count > p.cfg.nt) return fail(MIFWI_EINVAL, "level range out of bounds");
        if (p.save_f16) return read_half_levels(p, shot, first, count, (float *)out);
        base = (const T *)p.saved + (long)shot * p.L.field;
        lstride = (long)p.cfg.max_shots * p.L.field;
        nlev = p.cfg.nt;
    } else if (which == 1 || which == 2) {
        if (!(which == 1 ? p.rot_u_valid : p.rot_v_valid)) return fail(MIFWI_ESTATE, "no buffered wavefield available");
        base = (const T *)(which == 1 ? p.rot_u : p.rot_v) + (long)shot * 3 * p.L.field;
        nlev = 3;
    } else if (which >= 3 && which <= 6) {
        const bool fwd = which <= 4;
        if (!(fwd ? p.tti_fwd_valid : p.tti_adj_valid)) return fail(MIFWI_ESTATE, "no TTI wavefield available");
        const T *pair = (const T *)(fwd ? p.tti_fwd : p.tti_adj);
        const int second = (which - 3) & 1;
        base = pair + ((long)second * p.cfg.max_shots + shot) * 3 * p.L.field;
        nlev = 3;
    } else if (which == 7 || which == 8) {
        if (!p.tti_save_valid || shot >= p.tti_save_shots) return fail(MIFWI_ESTATE, "no saved TTI wavefield for this shot");
        if (p.tti_ck_K > 0) return fail(MIFWI_ESTATE, "the TTI wavefields are checkpointed, not stored level by level");
        base = (const T *)p.tti_save + ((long)(which - 7) * p.cfg.max_shots + shot) * p.cfg.nt * p.L.field;
        nlev = p.cfg.nt;
    } else {
        return fail(MIFWI_EINVAL, "which must be 0..8");
    }
    if (first < 0 || count < 0 || first + count > nlev) return fail(MIFWI_EINVAL, "level range out of bounds");
    int rc;
    const bool dev = is_device_ptr(out);
    // levels are evenly spaced fields: treat them as `count` "shots" of one level each
    const int chunk = dev ? count : (int)std::max<size_t>(1, p.dense_elems / (size_t)p.L.ngrid);
    for (int l = 0; l < count; l += chunk) {
        const int n = std::min(chunk, count - l);
        T *dst = (T *)out + (long)l * p.L.ngrid;
        if (dev) {
            if ((rc = launch_unpack<T>(p, base + (long)(first + l) * lstride, lstride, dst, n, 0))) return rc;
        } else {
            if ((rc = launch_unpack<T>(p, base + (long)(first + l) * lstride, lstride, (T *)p.d_dense, n, 0))) return rc;
            if ((rc = copy_any(p, dst, p.d_dense, (size_t)n * p.L.ngrid * sizeof(T)))) return rc;
            MIFWI_HIP(hipStreamSynchronize(p.stream));
        }
    }
    return finish(p);
}

template <typename T>
static int write_wavefield(Plan &p, int shot, int first, int count, const void *in)
{
    if (!in) return fail(MIFWI_EINVAL, "in is null");
    if (shot < 0 || shot >= p.cfg.max_shots) return fail(MIFWI_EINVAL, "shot out of range");
    if (first < 0 || count < 0 || first + count > p.cfg.nt) return fail(MIFWI_EINVAL, "level range out of bounds");
    int rc;
    if ((rc = prepare_save<T>(p))) return rc;
    if (p.ck_K > 0) return fail(MIFWI_ESTATE, "a caller-provided wavefield needs level-by-level storage, which does not fit");
    if (p.save_f16) return fail(MIFWI_ESTATE, "a caller-provided wavefield cannot be stored with MIFWI_FLAG_SAVE_F16");
    T *base = (T *)p.saved + (long)shot * p.L.field;
    const long lstride = (long)p.cfg.max_shots * p.L.field;
    const bool dev = is_device_ptr(in);
    const int chunk = dev ? count : (int)std::max<size_t>(1, p.dense_elems / (size_t)p.L.ngrid);
    for (int l = 0; l < count; l += chunk) {
        const int n = std::min(chunk, count - l);
        const T *src = (const T *)in + (long)l * p.L.ngrid;
        if (!dev) {
            if ((rc = copy_any(p, p.d_dense, src, (size_t)n * p.L.ngrid * sizeof(T)))) return rc;
            src = (const T *)p.d_dense;
        }
        if ((rc = launch_pack<T>(p, src, base + (long)(first + l) * lstride, lstride, n))) return rc;
        if (!dev) MIFWI_HIP(hipStreamSynchronize(p.stream));
    }
    // a caller-provided field makes the shot usable by mifwi_gradient
    p.saved_valid = true;
    p.saved_shots = std::max(p.saved_shots, shot + 1);
    return finish(p);
}

}  // namespace mifwi

// =============================================================================================
// C ABI
// =============================================================================================
using namespace mifwi;

#define PLAN_OR_FAIL(plan)                                                     \
    if (!(plan)) return fail(MIFWI_EINVAL, "plan is null");                    \
    Plan &p = *reinterpret_cast<Plan *>(plan);                                 \
    if (hipSetDevice(p.device) != hipSuccess) return fail(MIFWI_EHIP, "hipSetDevice failed")

#define DISPATCH(expr_f32, expr_f64) (p.cfg.dtype == MIFWI_F64 ? (expr_f64) : (expr_f32))

// Host containers (sparse tables, the graph cache) can throw std::bad_alloc: no exception crosses the C boundary.
#define GUARDED(expr)                                                          \
    try {                                                                      \
        return (expr);                                                         \
    } catch (const std::bad_alloc &) {                                         \
        return fail(MIFWI_ENOMEM, "host allocation failed");                   \
    } catch (const std::exception &e) {                                        \
        return fail(MIFWI_EHIP, std::string("exception: ") + e.what());        \
    } catch (...) {                                                            \
        return fail(MIFWI_EHIP, "unknown exception");                          \
    }

extern "C" {

int mifwi_version(void) { return MIFWI_VERSION; }

const char *mifwi_last_error(void) { return g_error.c_str(); }

int mifwi_plan_create(const mifwi_config *cfg, mifwi_plan **out)
{
    GUARDED(create(cfg, out));
}

int mifwi_plan_destroy(mifwi_plan *plan)
{
    if (!plan) return MIFWI_OK;
    destroy(reinterpret_cast<Plan *>(plan));
    return MIFWI_OK;
}

int mifwi_set_stream(mifwi_plan *plan, void *hip_stream)
{
    PLAN_OR_FAIL(plan);
    MIFWI_HIP(hipStreamSynchronize(p.stream));
    int rc = resolve_stats(p);   // pending loop times belong to the old stream
    if (rc) return rc;
    clear_graphs(p);
    if (p.own_stream && p.stream) (void)hipStreamDestroy(p.stream);
    // a null handle is the legacy default stream: legal, but it cannot be captured, so time loops on it are
    // launched kernel by kernel (see timed_loop)
    p.stream = reinterpret_cast<hipStream_t>(hip_stream);
    p.own_stream = false;
    return MIFWI_OK;
}

int mifwi_synchronize(mifwi_plan *plan)
{
    PLAN_OR_FAIL(plan);
    MIFWI_HIP(hipStreamSynchronize(p.stream));
    return resolve_stats(p);
}

int mifwi_set_model(mifwi_plan *plan, const void *vp, const void *damp)
{
    PLAN_OR_FAIL(plan);
    if (!vp) return fail(MIFWI_EINVAL, "vp is null");
    GUARDED(DISPATCH(set_model<float>(p, vp, damp), set_model<double>(p, vp, damp)));
}

int mifwi_set_geometry(mifwi_plan *plan, int32_t nshots, const void *src_coords, const void *rec_coords,
                       int32_t rec_per_shot)
{
    PLAN_OR_FAIL(plan);
    if (nshots < 1 || nshots > p.cfg.max_shots) return fail(MIFWI_EINVAL, "nshots must be in 1..max_shots");
    if ((p.cfg.nsrc > 0 && !src_coords) || (p.cfg.nrec > 0 && !rec_coords)) return fail(MIFWI_EINVAL, "null coordinates");
    GUARDED(DISPATCH(set_geometry<float>(p, nshots, src_coords, rec_coords, rec_per_shot),
                     set_geometry<double>(p, nshots, src_coords, rec_coords, rec_per_shot)));
}

int mifwi_forward(mifwi_plan *plan, int32_t nshots, const void *wavelet, int32_t wavelet_per_shot, void *rec_out,
                  int32_t save)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(forward<float>(p, nshots, wavelet, wavelet_per_shot, rec_out, save),
                    forward<double>(p, nshots, wavelet, wavelet_per_shot, rec_out, save)));
}

int mifwi_born(mifwi_plan *plan, int32_t nshots, const void *wavelet, int32_t wavelet_per_shot, const void *dm,
               void *rec_out)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(born<float>(p, nshots, wavelet, wavelet_per_shot, dm, rec_out),
                    born<double>(p, nshots, wavelet, wavelet_per_shot, dm, rec_out)));
}

int mifwi_adjoint(mifwi_plan *plan, int32_t nshots, const void *rec_in, void *srca_out)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(adjoint<float>(p, nshots, rec_in, srca_out), adjoint<double>(p, nshots, rec_in, srca_out)));
}

int mifwi_gradient(mifwi_plan *plan, int32_t nshots, const void *residual, void *grad, int32_t accumulate, void *illum)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(gradient<float>(p, nshots, residual, grad, accumulate, illum),
                    gradient<double>(p, nshots, residual, grad, accumulate, illum)));
}

int mifwi_l2_misfit_gradient(mifwi_plan *plan, int32_t nshots, const void *wavelet, int32_t wavelet_per_shot,
                             const void *obs, double *fval, void *grad, void *illum, void *residual_out, void *syn_out)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(
        l2_misfit_gradient<float>(p, nshots, wavelet, wavelet_per_shot, obs, fval, grad, illum, residual_out, syn_out),
        l2_misfit_gradient<double>(p, nshots, wavelet, wavelet_per_shot, obs, fval, grad, illum, residual_out, syn_out)));
}

int mifwi_set_pml(mifwi_plan *plan, int32_t npml, double quibar)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(set_pml<float>(p, npml, quibar, 15, 0, nullptr), set_pml<double>(p, npml, quibar, 15, 0, nullptr)));
}

int mifwi_set_pml_layout(mifwi_plan *plan, int32_t npml, double quibar, int32_t sides, int32_t edges, const void *vp_staggered)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(set_pml<float>(p, npml, quibar, sides, edges, vp_staggered),
                     set_pml<double>(p, npml, quibar, sides, edges, vp_staggered)));
}

int mifwi_l2_misfit(mifwi_plan *plan, int32_t nshots, const void *syn, const void *obs, void *residual_out, double *fval)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(l2_misfit<float>(p, nshots, syn, obs, residual_out, fval),
                     l2_misfit<double>(p, nshots, syn, obs, residual_out, fval)));
}

int mifwi_sum_shots(mifwi_plan *plan, int32_t nshots, const void *grad, const void *illum, const double *mute,
                    const int32_t *lo, const int32_t *phys_shape, double *g_sum, double *il_sum)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(sum_shots<float>(p, nshots, grad, illum, mute, lo, phys_shape, g_sum, il_sum),
                     sum_shots<double>(p, nshots, grad, illum, mute, lo, phys_shape, g_sum, il_sum)));
}

int mifwi_precondition(mifwi_plan *plan, int64_t n, const double *g_sum, const double *il_sum, const double *mask,
                       double *out)
{
    PLAN_OR_FAIL(plan);
    GUARDED(precondition(p, (long)n, g_sum, il_sum, mask, out));
}

int mifwi_l2_misfit_noalloc(int32_t dtype, int32_t nshots, int64_t n_per_shot, const void *syn, const void *obs,
                            void *residual_out, double *fval, void *hip_stream)
{
    if (dtype != MIFWI_F32 && dtype != MIFWI_F64) return fail(MIFWI_EINVAL, "dtype must be MIFWI_F32 or MIFWI_F64");
    if (nshots < 1 || nshots > kNoAllocMaxShots)
        return fail(MIFWI_EINVAL, "mifwi_l2_misfit_noalloc handles 1.." + std::to_string(kNoAllocMaxShots) +
                                      " shots per call (fixed scratch array)");
    if (n_per_shot < 0) return fail(MIFWI_EINVAL, "n_per_shot is negative");
    if (!syn || !obs || !residual_out || !fval) return fail(MIFWI_EINVAL, "null array argument");
    // the scratch array used is the current device's copy: the arrays (and the stream) must be that device's too
    int cur = -1;
    MIFWI_HIP(hipGetDevice(&cur));
    for (const void *ptr : {syn, obs, (const void *)residual_out, (const void *)fval}) {
        const int d = device_of_ptr(ptr);
        if (d < 0) return fail(MIFWI_EINVAL, "mifwi_l2_misfit_noalloc takes device pointers only");
        if (d != cur)
            return fail(MIFWI_EINVAL, "mifwi_l2_misfit_noalloc: an array lives on device " + std::to_string(d) +
                                          " but the current device is " + std::to_string(cur));
    }
    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    GUARDED(dtype == MIFWI_F64 ? l2_misfit_noalloc<double>(nshots, (long)n_per_shot, syn, obs, residual_out, fval, stream)
                               : l2_misfit_noalloc<float>(nshots, (long)n_per_shot, syn, obs, residual_out, fval, stream));
}

int mifwi_tti_set_model(mifwi_plan *plan, const void *epsilon, const void *delta, const void *theta, const void *phi)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(tti_set_model<float>(p, epsilon, delta, theta, phi), tti_set_model<double>(p, epsilon, delta, theta, phi)));
}

int mifwi_tti_forward(mifwi_plan *plan, int32_t nshots, const void *wavelet, int32_t wavelet_per_shot, void *rec_out,
                      int32_t save)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(tti_forward<float>(p, nshots, wavelet, wavelet_per_shot, rec_out, save),
                    tti_forward<double>(p, nshots, wavelet, wavelet_per_shot, rec_out, save)));
}

int mifwi_tti_adjoint(mifwi_plan *plan, int32_t nshots, const void *rec_in, void *srca_out)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(tti_adjoint<float>(p, nshots, rec_in, srca_out), tti_adjoint<double>(p, nshots, rec_in, srca_out)));
}

int mifwi_tti_born(mifwi_plan *plan, int32_t nshots, const void *wavelet, int32_t wavelet_per_shot, const void *dm,
                   void *rec_out)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(tti_born<float>(p, nshots, wavelet, wavelet_per_shot, dm, rec_out),
                    tti_born<double>(p, nshots, wavelet, wavelet_per_shot, dm, rec_out)));
}

int mifwi_tti_gradient(mifwi_plan *plan, int32_t nshots, const void *residual, void *grad, int32_t accumulate)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(tti_gradient<float>(p, nshots, residual, grad, accumulate),
                    tti_gradient<double>(p, nshots, residual, grad, accumulate)));
}

int mifwi_read_wavefield(mifwi_plan *plan, int32_t which, int32_t shot, int32_t first, int32_t count, void *out)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(read_wavefield<float>(p, which, shot, first, count, out),
                    read_wavefield<double>(p, which, shot, first, count, out)));
}

int mifwi_write_wavefield(mifwi_plan *plan, int32_t shot, int32_t first, int32_t count, const void *in)
{
    PLAN_OR_FAIL(plan);
    GUARDED(DISPATCH(write_wavefield<float>(p, shot, first, count, in), write_wavefield<double>(p, shot, first, count, in)));
}

int mifwi_get_stats(mifwi_plan *plan, mifwi_stats *out)
{
    PLAN_OR_FAIL(plan);
    if (!out) return fail(MIFWI_EINVAL, "out is null");
    int rc = resolve_stats(p);
    if (rc) return rc;
    p.stats.device_bytes = p.device_bytes;
    *out = p.stats;
    return MIFWI_OK;
}

}  // extern "C"
