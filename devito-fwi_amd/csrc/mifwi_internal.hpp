// Internal declarations shared by the host side (plan management, geometry tables, time loops)
// and the HIP kernels of libmifwi.  Not part of the public ABI (see include/mifwi.h).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "mifwi.h"

namespace mifwi {

constexpr int kMaxR = 8;        // space_order <= 16 on the tiled, marching and TTI kernels
constexpr int kMaxRUntiled = 10;   // space_order <= 20 on the untiled acoustic kernels (one thread per point)
constexpr int kMaxCorners = 8;  // 2^ndim

// ------------------------------------------------------------------------------------------
// Device memory layout of one field level.
//
// A field is stored as a canonical 3-D array [n0+2h0][n1+2h1][pitch] (unused leading dims have
// size 1 and no halo).  The last dimension is contiguous; grid point (i0,i1,i2) lives at
//     off0 + i0*s0 + i1*s1 + i2
// Halos of width R (zero, never written) surround the grid in every active dimension so that
// stencil reads need no bounds checks.  The z halo on the left is rounded up to a multiple of 4
// elements and the base is shifted so that off0 is a multiple of 32 elements: every grid row
// then starts on a 128-byte boundary (f32) and 16-byte vector accesses along z are aligned.
// ------------------------------------------------------------------------------------------
struct Layout {
    int ndim = 0;
    int R = 0;
    int n[3] = {1, 1, 1};
    int act[3] = {0, 0, 0};
    int halo[3] = {0, 0, 0};
    int zl = 0;           // left z halo (multiple of 4, >= R)
    long s0 = 0, s1 = 0;  // strides of canonical dims 0 and 1 (dim 2 has stride 1)
    long off0 = 0;        // element index of grid point (0,0,0)
    long field = 0;       // elements per field level (multiple of 32)
    long ngrid = 0;       // n0*n1*n2
};

// Per-launch constants of the time-stepping kernels (passed by value).
template <typename T>
struct StepConsts {
    T c0;                  // w_0 * sum_d 1/h_d^2
    T c[3][kMaxRUntiled + 1];     // c[d][k] = w_k / h_d^2, canonical dims
    T dt, dt2, inv_dt2;
    int n0, n1, n2;
    long s0, s1, off0;
    long field;            // elements per level
    int fs;                // free surface at the low end of the last dimension
};

// Sparse point set (sources or receivers) resident on the device.
// Interpolation view: per point, kMaxCorners (index, weight) pairs (weight 0 for unused ones).
// Injection view: CSR over the unique grid cells touched, so that each cell is updated by
// exactly one thread, in a fixed order (deterministic, no atomics).
struct SparseSet {
    int npoint = 0;
    int ncorner = 0;       // 2^ndim
    int per_shot = 0;      // 0: one table shared by all shots
    int nshots = 0;        // tables stored (1 if shared)
    int kmax = 0;          // capacity of unique cells per shot = npoint*ncorner
    // device arrays, [tables][...]
    int32_t *pt_idx = nullptr;   // [tables][npoint][ncorner] element offsets (relative to level base)
    void *pt_w = nullptr;        // [tables][npoint][ncorner] weights (T)
    int32_t *uc_cell = nullptr;  // [tables][kmax]
    int32_t *uc_ptr = nullptr;   // [tables][kmax+1]
    int32_t *uc_count = nullptr; // [tables]
    int32_t *e_point = nullptr;  // [tables][kmax]
    void *e_w = nullptr;         // [tables][kmax] (T)
    int max_count = 0;           // max unique cells over tables (host copy)
    // Injection view of the tiled kernels: per tile a fixed-capacity table of the cells it owns, so
    // that a block finds its work with loads whose addresses depend on nothing but its tile id.
    int ntiles = 0;
    int cap = 0;                   // slots per tile (max cells owned by any tile)
    int maxe = 0;                  // entries per slot (max points touching one cell)
    int32_t *slot_count = nullptr; // [tables][ntiles] cells owned by the tile
    int32_t *slot_local = nullptr; // [tables][ntiles][cap] tile-local cell index, -1 = empty
    int32_t *slot_ep = nullptr;    // [tables][ntiles][cap][maxe] point index (0 when padding)
    void *slot_ew = nullptr;       // [tables][ntiles][cap][maxe] weight (0 when padding)
    int32_t *slot_base = nullptr;  // [tables][ntiles] first position of the tile's cells in tile-major order
    int32_t *g_u = nullptr;        // [tables][kmax] unique-cell index (into uc_*) of tile-major position g
    int max_own = 0;               // most points any tile samples (a point belongs to the tile of its first weighted corner)
    // Injection view of the 2.5-D kernel: owned cells ordered by ((y,z) tile, x plane)
    int32_t *cell_ptr = nullptr;   // [tables][ntiles][n0+1] ranges into cell_local / cell_uc
    int32_t *cell_local = nullptr; // [tables][kmax] tile-local (y,z) index of the cell
    int32_t *cell_uc = nullptr;    // [tables][kmax] index of the cell in uc_ptr
    int c_lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, c_hi[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};   // box of the cells of all tables, 2.5-D kernel (host copy)
};

struct Plan {
    mifwi_config cfg{};
    Layout L{};
    int device = 0;
    size_t esize = 4;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [forward | backward loop][begin | end]
    bool timing_pending[2] = {false, false};   // the loop's events have been recorded but not read (MIFWI_FLAG_ASYNC)
    int timing_last = 0;                       // loop whose time is stats.kernel_ms (0 forward, 1 backward)
    bool timing_sum = false;                   // stats.kernel_ms of the last call = forward + backward loop

    void *m = nullptr, *eta = nullptr;       // haloed model fields
    bool have_model = false;
    bool have_geometry = false;
    int geom_shots = 0;

    void *rot_u = nullptr;                   // [max_shots][3][field] forward rotating buffers
    void *rot_v = nullptr;                   // [max_shots][3][field] adjoint rotating buffers
    void *saved = nullptr;                   // [levels][max_shots][field] saved forward field (level-major);
                                             // levels = nt, or nt - ck_nc when the first steps are checkpointed
    void *ckpt = nullptr;                    // [max_shots][ck_nseg][2][field] levels entering each segment
    void *seg = nullptr;                     // [max_shots][ck_K+2][field] recomputed segment
    int ck_K = 0, ck_nseg = 0;               // segment length (0: every level is stored) and count
    int ck_nc = 0;                           // steps 1 .. ck_nc are checkpointed; `saved` holds levels ck_nc .. nt-1
    bool saved_valid = false;
    int saved_shots = 0;
    bool rot_u_valid = false, rot_v_valid = false;

    SparseSet src, rec;

    void *d_wavelet = nullptr;               // [max_shots][nt][nsrc]
    void *d_wavelet_ck = nullptr;            // wavelets of the run that took the checkpoints (checkpointed plans only)
    const void *wavelet_override = nullptr;  // while segments are recomputed: the wavelet the untiled kernels inject
    bool ck_model_changed = false;           // mifwi_set_model ran after the checkpoints were taken
    void *d_rec = nullptr;                   // [max_shots][nt][nrec]   synthetic data
    void *d_res = nullptr;                   // [max_shots][nt][nrec]   adjoint source / residual
    void *d_srca = nullptr;                  // [max_shots][nt][nsrc]
    void *d_grad = nullptr, *d_illum = nullptr; // [max_shots][field] haloed accumulators
    void *d_t4 = nullptr, *d_w2 = nullptr, *d_bih = nullptr;   // OT4 scratch fields, [max_shots][field] each
    // TTI state: parameter fields [6][field], field pairs [2][max_shots][3][field] (forward u,v / adjoint p,r),
    // scratch [4][max_shots][field] (wA, wB, gzA, gzB)
    void *tti_par = nullptr, *tti_fwd = nullptr, *tti_adj = nullptr, *tti_tmp = nullptr;
    // 3-D float32 plans: Gz of the two stencilled fields, [2][max_shots][field], and the tables st cp, st sp, [2][field]
    // (kernels_tti.hip::tti3d_gz_kernel / tti3d_step_kernel)
    void *tti_gz3 = nullptr, *tti_w = nullptr;
    bool tti_two_pass = false;   // orders 4 and 8 with two marches as well (MIFWI_TTI3D_TWOPASS at tti_set_model)
    bool tti_fwd_valid = false, tti_adj_valid = false;
    // host copies of the coordinates last given to mifwi_set_geometry (the sparse tables depend on the tile
    // shape, which the first mifwi_tti_set_model may change)
    std::vector<unsigned char> geo_src, geo_rec;
    int geo_rec_per_shot = 0;
    bool save_f16 = false;          // saved levels are half precision (MIFWI_FLAG_SAVE_F16 in effect)
    void *d_save_scale = nullptr;   // per shot: stored half = float * scale
    void *tti_save = nullptr;   // every level of both forward fields: [field][shot][nt] (tti_forward with save)
    void *tti_ckpt = nullptr;   // checkpointed instead: [field][shot][segment][2] levels entering each segment
    void *tti_seg = nullptr;    // and [field][shot][tti_ck_K + 2] levels of the segment being propagated again
    int tti_ck_K = 0, tti_ck_nseg = 0;
    bool tti_save_valid = false;
    int tti_save_shots = 0;
    // Injected terms of every time step, formed once before a time loop of the tiled kernels:
    // winj[shot][time][g] = (sum_e w_e vals[shot][time][point_e]) * dt^2 / m[cell], g in the order the
    // kernels walk the cells (tile-major slots in 2-D, cell_local order in 3-D)
    void *d_winj[2] = {nullptr, nullptr};   // [0]: terms of the sources, [1]: of the receivers
    size_t winj_elems[2] = {0, 0};
    // MIFWI_FLAG_PML: auxiliary fields [max_shots][phi1 | phi2][2 levels][field] and damping profiles
    // [zeta_x nodes | zeta_z nodes | zeta_x staggered | zeta_z staggered], n1 + n2 + n1 + n2 values
    void *pml_phi = nullptr, *pml_zeta = nullptr, *pml_c2s = nullptr;
    bool have_pml = false;
    int pml_sides = 15, pml_edges = 0, pml_npml = 0;
    void *d_dm = nullptr;                    // [field] haloed model perturbation of the Born operator
    void *d_dense = nullptr;                 // dense staging, max(max_shots*ngrid, ...) elements
    size_t dense_elems = 0;
    double *d_partial = nullptr;             // misfit partial sums
    double *d_fval = nullptr;                // [max_shots]
    double *d_reduce = nullptr;              // [3 * padded grid] shot sums and preconditioned gradient, on first use

    // fast paths (float32): kind 1 = tiled 2-D kernel, tiles tile_tx x tile_tz over (x, z);
    // kind 2 = 2.5-D kernel, tiles tile_tx x tile_tz over (y, z), marching x in chunks of xchunk
    // planes; kind 0 = generic kernels only
    int kind = 0;
    int tile_tx = 0, tile_tz = 0, ntx = 0, ntz = 0;
    int xchunk = 0, nchunk = 0;
    int ncu = 256, bpc_fwd = 2, bpc_grad = 2;  // CUs and resident 2.5-D blocks per CU (plain / gradient)
    int nchunk_env = 0, nchunk_grad_env = 0;
    // 2.5-D forward loops: per shot the box that holds anything so far (lo[3][max_shots], hi[3][max_shots]), the
    // copies taken with the checkpoints, whether plans use it at all and whether the loop being enqueued does
    int *d_box = nullptr, *d_ck_box = nullptr, *d_boxv = nullptr;   // (d_boxv: the backward field's)
    bool box_enabled = false, box_active = false, box_active_v = false;
    int slab_w = 0, slab_chunks = 1;   // two-step sweeps of the 2.5-D kernel: planes per slab (0 = off), chunks per slab
    unsigned char *eta_zero = nullptr;  // 2.5-D kernel: [tile][n0] flags of sponge-free patches (scratch of the next)
    int *eta_free = nullptr;            // 2.5-D kernel: [tile][2] longest run lo <= x < hi of sponge-free planes
    int vec2d = 4;      // points per thread of the tiled 2-D kernel (2: experiment, 32 x 32 tiles only)
    int pair_save = 1;  // saving forward loops of the tiled 2-D kernel write two saved levels every second step
    int pair_grad = 1;  // gradient loops update the accumulators every second step (tiled kernels)
    // whole-loop launches of the tiled 2-D kernel's work (kernels_persist2d.hip)
    int persist = 1;                   // 0: never (MIFWI_PERSIST=0)
    int persist_skip = 0;              // loops still to run with a kernel per step after a whole-loop launch was abandoned
    int persist_backoff = 64;          // and how many the next abandoned launch will cost (doubles, up to 4096)
    int persist_spin = 1 << 21;        // polls of a neighbour's counter before a loop is abandoned (MIFWI_PERSIST_SPIN)
    int persist_bpc[2] = {-1, -1};     // resident blocks per CU of the plain / gradient variant, -1 = not asked yet
    int persist_tti_bpc[3] = {-1, -1, -1};   // the same for the whole-loop TTI kernel: forward, adjoint, gradient
    bool persist_check = true;         // a one-block kernel behind every whole-loop launch verifies the tiles' step counts
    int xcds = 0;                      // XCDs blocks are dealt to on this device (census at plan creation); the whole-loop kernel needs 8
    int *d_psync = nullptr;            // per-tile step counters [8][tiles], blocks taken per XCD [16], abort flag
    int *h_abort = nullptr;            // pinned host copy of the abort flag
    bool loop_persist = false;         // the loop whose statistics are being filled in ran that way
    bool tti_whole_loop = false;       // 2-D TTI loops of this plan run as one launch each (decided at the first tti_set_model)
    bool loop_persist_tti = false;     // ... and the loop whose statistics are being filled in did

    // captured time loops: (loop kind, nshots, variant a, variant b, group) -> executable graph + launches
    std::map<std::tuple<int, int, int, int, int>, std::pair<hipGraphExec_t, int>> graphs;

    mifwi_stats stats{};
    int64_t device_bytes = 0;
};

void set_error(const std::string &msg);

#define MIFWI_HIP(call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            ::mifwi::set_error(std::string(#call) + ": " + hipGetErrorString(e_));                 \
            return (e_ == hipErrorOutOfMemory) ? MIFWI_ENOMEM : MIFWI_EHIP;                        \
        }                                                                                          \
    } while (0)

// ---- kernel launchers (kernels_generic.hip) ------------------------------------------------
template <typename T>
struct StepPtrs {
    const T *cur;     // level being read with the stencil           (shot 0)
    const T *old;     // level two steps away (centre point only)    (shot 0)
    T *out;           // level being produced                         (shot 0)
    long shot_stride; // elements between consecutive shots
    const T *m, *eta;
    const T *ot4;     // OT4: (1/m) L(cur) per shot, [shots][field]; null for OT2
};

template <typename T>
struct GradPtrs {
    const T *um, *uc, *up; // saved forward levels time-1, time, time+1 (shot 0); null = no gradient
    long u_shot_stride;
    T *grad;               // haloed accumulators (shot 0)
    T *illum;              // or null
    long g_shot_stride;
    int first;             // 1 on the first backward step: also add u[time+1]^2 to illum
    const T *bih;          // OT4: m^-2 L(u[time]) per shot, [shots][field]; null for OT2
};

template <typename T>
int launch_step_generic(const Plan &p, const StepConsts<T> &k, const StepPtrs<T> &sp,
                        const GradPtrs<T> &gp, int nshots);

// One launch doing both sparse operations of a time step:
//   inject:  field_inj[cell] += (sum_e w_e * vals_inj[shot][time][point_e]) * dt2 / m[cell]
//   interp:  vals_out[shot][time][p] = sum_c w_c * field_int[idx_c]
template <typename T>
struct SparsePtrs {
    const SparseSet *inj;      // null: no injection
    T *field_inj;              // shot 0 level base
    T *field_inj2 = nullptr;   // optional second field receiving the same terms (TTI)
    const T *vals_inj;         // [shots][nt][inj->npoint]
    const SparseSet *itp;      // null: no interpolation
    const T *field_int;        // shot 0 level base
    const T *field_int2 = nullptr; // optional second field added to the sampled one (TTI: u + v)
    T *vals_out;               // [shots][nt][itp->npoint]
    long shot_stride;
    const T *m;
    int time, nt;
    T dt2;
};
template <typename T>
int launch_sparse(const Plan &p, const SparsePtrs<T> &sp, int nshots);

// out[shot] = L(in[shot]) / m^power on the grid points (halo untouched, i.e. zero): the inner half of
// the OT4 biharmonic terms (seismic/acoustic/operators.py:55,219).
template <typename T>
int launch_lap_scale(const Plan &p, const StepConsts<T> &k, const T *in, long in_shot_stride, int power, T *out,
                     long out_shot_stride, int nshots);

// Born source term of one step (seismic/acoustic/operators.py:263): for every grid point
//   U_out += dt^2 * ( -dm * (u_new - 2 u_cur + u_old) / dt^2 ) / ( m + eta dt )
// `ghost`: the fields carry the free-surface mirror in their z halo; refresh it for U_out.
template <typename T>
int launch_born_source(const Plan &p, const StepConsts<T> &k, const T *u_new, const T *u_cur, const T *u_old,
                       long shot_stride, const T *dm, T *U_out, long U_shot_stride, int nshots, int ghost);

// winj[shot][time][g] for g < uc_count[table]: see Plan::d_winj.  g_u maps g to the unique cell.
int launch_inject_prepare(const Plan &p, const SparseSet &S, const int32_t *g_u, const float *vals, int nshots,
                          float dt2, float *winj);

template <typename T>
int launch_prep_model(const Plan &p, const T *vp_dense, const T *damp_dense, T *m, T *eta);

// 2-D perfectly matched layer (kernels_generic.hip): displacement step and auxiliary-field step of all shots
template <typename T>
struct PmlPtrs {
    const T *cur, *old;      // u[t], u[t-1] (shot 0)
    T *out;                  // u[t+1]
    long shot_stride;
    const T *m;
    const T *zx0, *zz0, *zx1, *zz1;   // damping profiles along dimension 1 (x) and 2 (z): nodes, staggered points
    const T *phi1, *phi2;    // auxiliary fields at time t (shot 0)
    T *phi1_out, *phi2_out;  // and at t+1
    long phi_shot_stride;
    T ihx, ihz;              // 1 / spacing
    // layout: sides that have a layer (bit 0 low x, 1 high x, 2 low z, 3 high z), its width in nodes, whether the
    // notebook's regions and boundary rows apply, and the caller's c^2 at the staggered points (null: corner mean)
    int sides, npml, edges;
    const T *c2s;
};
template <typename T>
int launch_pml_edges(const Plan &p, const StepConsts<T> &k, T *out, long shot_stride, int nshots);
template <typename T>
int launch_pml_u(const Plan &p, const StepConsts<T> &k, const PmlPtrs<T> &q, int nshots);
template <typename T>
int launch_pml_phi(const Plan &p, const StepConsts<T> &k, const PmlPtrs<T> &q, int nshots);

// dense (nshots, grid) <-> haloed (nshots, field).  mode: 0 dst = src, 1 dst += src
template <typename T>
int launch_unpack(const Plan &p, const T *haloed, long shot_stride, T *dense, int nshots, int mode);
template <typename T>
int launch_pack(const Plan &p, const T *dense, T *haloed, long shot_stride, int nshots);
// a[i] *= a[i] over n elements
template <typename T>
int launch_square(const Plan &p, T *a, long n);

// res = syn - obs ; fval[shot] = 0.5 * sum res^2 (f64, deterministic two-stage reduction)
template <typename T>
int launch_l2_misfit(const Plan &p, const T *syn, const T *obs, T *res, long n_per_shot, int nshots,
                     double *partial, double *fval);
// the same on any stream; `partial` holds kMisfitPartials doubles per shot, each written before it is read
constexpr int kMisfitPartials = 64;
template <typename T>
int launch_l2_misfit_on(hipStream_t stream, const T *syn, const T *obs, T *res, long n_per_shot, int nshots,
                        double *partial, double *fval);

// physical-grid window of a padded-grid array (the crop of fwi.py:164)
struct CropMap {
    int nd;
    int phys[3], lo[3];
    long pad_stride[3];
    long field;   // elements of one shot's padded-grid array
    long nphys;   // elements of the physical grid
};
// gs = sum_s crop(g[s]) * mute[s] (ils likewise from il; il, ils and mute may be null), float64, shot order
template <typename T>
int launch_mute_sum(hipStream_t stream, const CropMap &c, int nshots, const T *g, const T *il, const double *mute,
                    double *gs, double *ils);
// out = g / sqrt(il + 1e-30) * mask (il null: no division; mask null: no mask)
int launch_precondition(hipStream_t stream, long n, const double *g, const double *il, const double *mask, double *out);

// ---- TTI propagators (kernels_tti.hip) ----------------------------------------------------------
template <typename T>
struct TtiParams {
    const T *et, *dl, *ct, *st, *cp, *sp;  // haloed: 1+2eps, sqrt(1+2delta), cos/sin(theta), cos/sin(phi)
    T a[kMaxR / 2 + 1];                    // centred first-derivative weights a_1..a_p
    T ih[3];                               // 1/h per canonical dimension
    int p;                                 // space_order / 4
};
template <typename T>
struct TtiLevels {
    const T *ucur, *uold, *vcur, *vold;    // shot 0
    T *uout, *vout;
    long shot_stride;
};
// fill a haloed parameter field: halo_value everywhere, then f(dense) on the grid points
// (mode 1: 1+2x, 2: sqrt(1+2x), 3: cos, 4: sin); dense == null leaves the constant
template <typename T>
int launch_tti_param(const Plan &p, const StepConsts<T> &k, const T *dense, T *haloed, int mode, T halo_value);
int launch_tti3d_weights(const Plan &p);   // fills p.tti_w from p.tti_par (3-D float32 plans)
// Optional extra term of a TTI step.  Born (b_un != null): the stepped fields are the scattered ones and get
// dt^2 * ( -dm * u0.dt2 ) / ( m + eta dt ) resp. the same with v0 (seismic/tti/operators.py:573-576);
// b_* are the background levels t+1, t, t-1 of shot 0.  Gradient (grad != null): the stepped fields are the
// adjoint ones and grad -= u0.dt2 * p + v0.dt2 * r (operators.py:631); su/sv point at level `time` of the
// saved background fields of shot 0, consecutive levels level_stride apart.
template <typename T>
struct TtiExtra {
    const T *b_un, *b_uc, *b_uo, *b_vn, *b_vc, *b_vo;
    long b_shot_stride;
    const T *b_dm;
    const T *su, *sv;
    long level_stride, s_shot_stride;
    T *grad;
    long g_shot_stride;
};
struct Step2DArgs;
// sparse2d != null (float plans for which tti2d_available): the tiled 2-D kernel, which also adds the
// injected terms and samples the receivers described by the injection/interpolation members of *sparse2d
template <typename T>
int launch_tti_step(const Plan &p, const StepConsts<T> &k, const TtiParams<T> &q, int fwd, const TtiLevels<T> &lv,
                    T *wA, T *wB, T *gzA, T *gzB, int nshots, int &launches, const Step2DArgs *sparse2d,
                    const TtiExtra<T> *extra = nullptr);   // extra is honoured by the tiled kernel only
bool tti2d_available(const Plan &p);
// grad[shot] += -( u0.dt2 * p + v0.dt2 * r ) at one time level (seismic/tti/operators.py:631); su/sv point at
// level `time` of shot 0 of the saved fields, one level = level_stride elements
template <typename T>
int launch_tti_grad(const Plan &p, const StepConsts<T> &k, const T *su, const T *sv, long level_stride, long s_shot_stride,
                    const T *pc, const T *rc, long a_shot_stride, T *grad, long g_shot_stride, int nshots);

// ---- tiled 2-D fast path (kernels_tiled2d.hip) -------------------------------------------------
// LDS row pitch (floats) of a staged tile.  ds_read_b128 serves a wave in four groups of 16 lanes
// over 64 banks; a group spans two tile rows when a row is 8 lanes (TZ = 32) and one row plus parts
// of others when it is 16 lanes (TZ = 64).  Rows must therefore start half a bank row apart for
// TZ = 32 (pitch = 32 mod 64) and a whole bank row apart for TZ = 64 (pitch = 0 mod 64); with a
// pitch of 0 mod 64 the 32-wide tile had every read 2-way conflicted.
constexpr int tile2d_lds_pitch(int TZ, int R)
{
    const int need = TZ + 2 * ((R + 3) / 4 * 4);
    if (TZ == 32) return (need + 31) / 64 * 64 + 32;
    return (need + 63) / 64 * 64;
}

struct Step2DArgs {
    const float *cur, *old;
    float *out;
    long shot_stride;
    const float *m, *eta;
    float c0, cx[kMaxR + 1], cz[kMaxR + 1];
    float dt, dt2, inv_dt2;
    int n1, n2;
    long s1, off0;
    int ntx, ntz;
    int nshots, shot0;  // this launch covers shots shot0 .. shot0+nshots-1 (all pointers are shot 0's)
    // gradient (um == nullptr: plain propagation)
    const float *um, *uc, *up;
    long u_shot_stride;
    float *grad, *illum;
    long g_shot_stride;
    int first;
    // pair != 0: this launch also adds the gradient / illumination terms of the step before it (time + 1),
    // which ran as a plain propagation: its three backward levels are `out`'s old content, `old` and
    // `cur`, and `up` is the saved forward level time + 1
    int pair;
    // injection (slot_count == nullptr: none): per tile the cells it owns, and their precomputed
    // terms winj[shot][time][slot_base[tile] + j]
    const int32_t *slot_count, *slot_local, *slot_base;
    int inj_cap, inj_per_shot;
    const float *winj;
    long winj_stride;   // cells per (shot, time)
    // interpolation of `cur` (pt_idx == nullptr: none)
    const int32_t *pt_idx;
    const float *pt_w;
    int itp_npoint, itp_per_shot;
    float *vals_out;
    int time, nt;
    int fs;   // free surface: keep the antisymmetric mirror of rows z = 1..R in the z halo of `out`
    // second, streamed copy of `out` (the saved forward level), or null
    float *copy;
    long copy_shot_stride;
    // streamed copy of `cur` into the saved level before it, or null: every second step of a saving loop
    // writes two saved levels and the steps between write none
    float *copy_prev;
    // save_scale != nullptr: `copy`, `copy_prev` and um/uc/up are half precision, value * save_scale[shot]; their
    // strides still count elements
    const float *save_scale;
    // Born source term (b_un == nullptr: none): the field being advanced is the scattered one and
    // out += dt^2 * ( -dm * (b_un - 2 b_uc + b_uo) / dt^2 ) / ( m + eta dt )   (operators.py:263)
    const float *b_un, *b_uc, *b_uo, *b_dm;
    long b_shot_stride;
    // Fourth-order-in-time kernel (OT4, operators.py:52-56: laplace(u) + dt^2/12 laplace(laplace(u)/m)).  The
    // Laplacian is linear, so the sum is the Laplacian of z = u + dt^2/12 laplace(u)/m: a launch with pre_mode 1
    // writes z, and the step proper stencils z (`cur`) while its pointwise terms and the sampled receivers use the
    // level itself (`pt`, null when `cur` is the level).  pre_mode != 0 launches compute, instead of a time step,
    // out = 1: cur + pre_c laplace(cur)/m, 2: laplace(cur)/m^2, 3: laplace(cur) (nothing injected or sampled).
    // `bih` (gradient launches, or null) holds laplace(laplace(u0)/m^2) of the saved level: the gradient gains
    // -pre_c * bih * v (operators.py:218).
    const float *pt;
    int pre_mode;
    float pre_c;
    const float *bih;
    // elements between the shots of `cur` and of `out` when they differ from shot_stride (0: they do not); `old`
    // and `pt` always use shot_stride
    long cur_shot_stride, out_shot_stride;
};
int launch_step2d(const Plan &p, const Step2DArgs &a, hipStream_t stream);

// ---- whole time loops of small 2-D batches in one launch (kernels_persist2d.hip) ----------------
struct Persist2DArgs {
    Step2DArgs a;          // constants, model, injection and interpolation tables, gradient sums (cur/old/out/time unused)
    float *rot;            // [shots][3][field] rotating levels of the field being advanced; level t lives in slot t % 3
    long field;
    int forward;           // 1: time runs 1 .. nt-2, 0: nt-2 .. 1
    int nsteps;
    float *saved;          // forward loops: every new level is also streamed to saved[time + 1][shot], or null
    long saved_lstride;    // elements between saved levels
    const float *usaved;   // gradient loops: the saved forward field
    long usaved_lstride;
    int *done;             // [8][tiles] steps completed by each tile
    int *counter;          // [16] blocks that have taken a tile, per XCD
    int *abort_flag;
    int spin_limit;        // polls of a neighbour's counter before the loop is abandoned
    int drop_slot;         // -1; tests: blocks that land on this XCD leave at once, as if the device lacked it
};
int persist2d_blocks_per_cu(int R, int tile_tz, bool grad);
int launch_persist2d(const Plan &p, const Persist2DArgs &q, bool grad, int blocks_per_xcd, hipStream_t stream);
// sets *abort_flag unless done[0 .. ncounters) all equal nsteps (enqueued behind a whole-loop launch)
int launch_persist2d_check(const int *done, int ncounters, int nsteps, int *abort_flag, hipStream_t stream);
// XCDs the current device deals blocks to (ids 0 .. n-1 seen by a census launch), 0 when the ids have gaps
int persist2d_xcd_count(hipStream_t stream);

// ---- whole time loops of small 2-D TTI batches in one launch (kernels_persist_tti2d.hip) -------
struct PersistTtiArgs {
    Step2DArgs a;          // constants, model (m, eta), injection and interpolation tables
    const float *et, *dl, *ct, *st;   // haloed parameter fields: 1 + 2 eps, sqrt(1 + 2 delta), cos / sin of the tilt
    float a1[kMaxR / 2 + 1];          // centred first-derivative weights
    float ihx, ihz;                   // 1 / spacing
    float *rot_u, *rot_v;  // [shots][3][field] rotating levels of the two fields being advanced
    float *comb_a, *comb_b;   // backward loops: [shots][2][field] the stencilled combinations et p + dl r, dl p + r of the
                              // newest level, two alternating slots (zero before the loop)
    long field;
    int forward;           // 1: time runs 1 .. nt-2, 0: nt-2 .. 1
    int nsteps;
    float *save_u, *save_v;   // forward loops: every new level is also streamed to save_*[shot][time + 1], or null
    long save_shot_stride;    // elements between the shots of save_u / save_v (levels are `field` apart)
    const float *su, *sv;     // gradient loops: saved background fields, [shot][time]
    long s_shot_stride;
    float *grad;              // gradient sums [shot][field]
    long g_shot_stride;
    int *done, *counter, *abort_flag;   // as in Persist2DArgs
    int spin_limit, drop_slot;
};
int persist_tti2d_blocks_per_cu(int R, int mode);
int launch_persist_tti2d(const Plan &p, const PersistTtiArgs &q, int mode, int blocks_per_xcd, hipStream_t stream);

// ---- 2.5-D fast path for 3-D grids (kernels_3d.hip) -------------------------------------------
struct Step3DArgs {
    const float *cur, *old;
    float *out;
    long shot_stride;
    const float *m, *eta;
    float c0, cx[kMaxR + 1], cy[kMaxR + 1], cz[kMaxR + 1];
    float dt, dt2, inv_dt2;
    int n0, n1, n2;
    long s0, s1, off0;
    int nty, ntz, nchunk, xchunk, nwork, nshots;
    int x_lo, x_hi;   // planes this launch updates: x_lo <= x < x_hi, split into nchunk chunks of xchunk planes
    const int *eta_free;   // [tile][2]: planes lo <= x < hi of the tile's column are free of the sponge; or null
    // gradient (um == nullptr: plain propagation)
    const float *um, *uc, *up;
    long u_shot_stride;
    float *grad, *illum;
    long g_shot_stride;
    int first;
    int pair;   // see Step2DArgs::pair
    // injection (cell_ptr == nullptr: none): cells ordered by (tile, plane) and their precomputed
    // terms winj[shot][time][position]
    const int32_t *cell_ptr, *cell_local;
    int inj_kmax, inj_per_shot;
    const float *winj;
    long winj_stride;
    // interpolation of `cur` (pt_idx == nullptr: none)
    const int32_t *pt_idx;
    const float *pt_w;
    int itp_npoint, itp_per_shot;
    float *vals_out;
    int time, nt;
    int fs;   // free surface: keep the antisymmetric mirror of rows z = 1..R in the z halo of `out`
    // Born source term (b_un == nullptr: none): the field being advanced is the scattered one and
    // out += dt^2 * ( -dm * (b_un - 2 b_uc + b_uo) / dt^2 ) / ( m + eta dt )   (operators.py:263)
    const float *b_un, *b_uc, *b_uo, *b_dm;
    long b_shot_stride;
    // half-precision copy of `out` (the saved forward level), or null; save_scale != nullptr also makes
    // um/uc/up half precision; strides count elements (see Step2DArgs)
    void *copy16;
    long copy_shot_stride;
    const float *save_scale;
    // fourth-order-in-time kernel: see Step2DArgs
    const float *pt;
    int pre_mode;
    float pre_c;
    const float *bih;
    long cur_shot_stride, out_shot_stride;
    // forward loops and backward loops without illumination sums (null otherwise): per shot the box lo <= i < hi (x, then y, then z, box_stride ints apart) in which
    // the loop's steps so far have produced a value other than zero, read and widened by the kernel; inj_lo / inj_hi:
    // the box of the injected cells
    int *box_lo, *box_hi;
    int box_stride;
    int box_fill;   // store zeros where nothing is computed (0: `out` was cleared when the loop began)
    int inj_lo[3], inj_hi[3];
};
int launch_step3d(const Plan &p, const Step3DArgs &a, hipStream_t stream);
int launch_eta_zero(const Plan &p, unsigned char *flags, int *free_range);
int step3d_blocks_per_cu(int R, bool grad);


// ---- gradient and illumination terms with pinned rounding ----------------------------------------
// g - ((va - 2 vb + vc) / dt^2) u and il + u^2.  Every gradient kernel forms them through these two
// functions, so a step's term has the same bits whether it is added by the launch that advances the
// backward field or, deferred, by the next one.
__device__ inline float grad_accumulate(float g, float va, float vb, float vc, float u, float inv_dt2)
{
#pragma clang fp contract(off)
    const float d = __builtin_fmaf(-2.0f, vb, va) + vc;
    const float s = d * inv_dt2;
    return __builtin_fmaf(-s, u, g);
}
__device__ inline float illum_accumulate(float il, float u)
{
#pragma clang fp contract(off)
    const float s = u * u;
    return il + s;
}

// ---- half-precision storage of saved levels -----------------------------------------------------
// value -> half(value * scale), saturating at the largest finite half instead of overflowing to inf
__device__ inline unsigned pack_half2(float a, float b)
{
    const float lim = 65504.0f;
    const __half2 h = __floats2half2_rn(fminf(fmaxf(a, -lim), lim), fminf(fmaxf(b, -lim), lim));
    return *reinterpret_cast<const unsigned *>(&h);
}
__device__ inline void unpack_half2(unsigned u, float &a, float &b)
{
    const float2 f = __half22float2(*reinterpret_cast<const __half2 *>(&u));
    a = f.x, b = f.y;
}
// per-shot scale of the half-precision saved levels from the injected source terms of the whole run
int launch_save_scale(const Plan &p, const float *winj, const SparseSet &S, int nshots, float *scale);
// haloed half-precision level(s) -> haloed float (divided by the shot's scale)
int launch_half_to_float(const Plan &p, const void *src, float *dst, long n, const float *scale, int shot);

}  // namespace mifwi
