/*
 * mifwi -- MI355X-native isotropic acoustic propagators (forward / adjoint / gradient)
 * for full-waveform inversion.  Plain C ABI: pointers, sizes, ints and doubles only.
 *
 * What this boundary replaces in the reference (LongyanU/devito-fwi).  The reference has no
 * native propagator source of its own: seismic/acoustic/wavesolver.py:112,149,203 call
 * `Operator.apply(...)` of Devito, which JIT-compiles a shared object and calls, through
 * ctypes, one entry point per operator of the form
 *     int Kernel(struct dataobj *damp_vec, const float dt, struct dataobj *vp_vec,
 *                const float o_x, const float o_z, struct dataobj *rec_vec,
 *                struct dataobj *rec_coords_vec, struct dataobj *src_vec,
 *                struct dataobj *src_coords_vec, struct dataobj *u_vec,
 *                const int x_M, const int x_m, ..., const int time_M, const int time_m,
 *                struct profiler *timers)
 * (shape recorded in seismic/self_adjoint/sa_01_iso_implementation1.ipynb:1187-1206).
 * The entry points below are what a ctypes binding for that call site binds instead; each
 * one names the reference call it stands in for.
 *
 * Conventions
 *   - Every function returns 0 on success and a negative MIFWI_E* code on failure; the text of
 *     the last failure on the calling thread is returned by mifwi_last_error().  Nothing
 *     throws across the boundary.
 *   - Grid arrays are C-ordered (first dimension slowest, last contiguous), cover the PADDED
 *     grid (physical shape + 2*nbl per dimension) and carry NO halo: the library adds its own.
 *   - Array arguments may be host pointers or device pointers of the plan's GPU (the library
 *     copies with hipMemcpyDefault); they are borrowed for the duration of the call only.
 *   - A plan owns all device memory it needs, including the saved forward wavefield, which stays
 *     on the GPU between mifwi_forward(save=1) and mifwi_gradient().
 *   - A plan is not re-entrant; calls on one plan are serialised by the caller.  Calls return
 *     after the work has completed (they synchronise the plan's stream) unless
 *     MIFWI_FLAG_ASYNC is set, in which case mifwi_synchronize() must precede any host read.
 *   - There is no CPU fallback: without a usable HIP device mifwi_plan_create() fails.
 */
#ifndef MIFWI_H
#define MIFWI_H

#include <stdint.h>

/* the only symbols the shared library exports */
#define MIFWI_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

#define MIFWI_VERSION 103

enum {
    MIFWI_OK = 0,
    MIFWI_EINVAL = -1,   /* bad argument (shape, order, null pointer, shot count...)      */
    MIFWI_EHIP = -2,     /* a HIP runtime call failed (text in mifwi_last_error)          */
    MIFWI_ENOMEM = -3,   /* device allocation failed                                      */
    MIFWI_ESTATE = -4,   /* call sequence error (e.g. gradient without a saved wavefield) */
    MIFWI_ENODEV = -5    /* no usable GPU                                                 */
};

enum { MIFWI_F32 = 0, MIFWI_F64 = 1 };

enum {
    MIFWI_FLAG_ASYNC = 1,        /* do not synchronise at the end of each call            */
    MIFWI_FLAG_NO_GRAPH = 2,     /* one plain kernel launch per time step: no graph replay, and 2-D float32
                                    time loops are not run as a single launch (which is the default where it
                                    applies; the two schedules agree to rounding, ~1e-5 relative)         */
    MIFWI_FLAG_GENERIC = 4,      /* force the untiled reference kernels                    */
    MIFWI_FLAG_CHECKPOINT = 8,   /* keep the forward wavefield as checkpoints + recomputation
                                    even when every level would fit (automatic otherwise)  */
    MIFWI_FLAG_FREE_SURFACE = 16, /* free surface at the low end of the LAST dimension: the model was
                                    built with fs=True (seismic/model.py:100-110,157: no absorbing
                                    layer there) and the stencil mirrors the field antisymmetrically
                                    about index 0 (seismic/acoustic/operators.py:8-35,92-94)   */
    MIFWI_FLAG_OT4 = 32,         /* 4th-order-in-time kernel: Laplacian + dt^2/12 * biharmonic(1/m)
                                    and the matching gradient term (seismic/acoustic/operators.py:
                                    38-56,216-219).  Float32 plans run it on the tiled kernels: the Laplacian is
                                    linear, so a launch writes z = u + dt^2/12 laplace(u)/m and the step proper
                                    stencils z (with MIFWI_FLAG_SAVE_F16, and in float64, on the untiled
                                    kernels; with MIFWI_FLAG_FREE_SURFACE on the untiled kernels too: the
                                    mirrored reads of operators.py:8-35 then apply to laplace(u)/m as well).  The caller passes
                                    dt = 1.73 * CFL dt as the reference's solver does (wavesolver.py:41-46). */
    MIFWI_FLAG_TIME_BLOCK = 128, /* 3-D float32 forward modelling without saving: two time steps per sweep over the
                                    grid.  x is cut into slabs; the second step follows the first one slab behind, so
                                    what it reads (the level just written, the one before, the coefficients) is still
                                    in the 256 MB last-level cache instead of coming from HBM a second time.  Same
                                    kernels, same arithmetic: results are bit-identical to the plain loop.  Off by
                                    default: on MI355X it measured 3 % slower than one sweep per step at 592^3
                                    (the last-level cache serves reads only ~1.5x faster than HBM and the slab
                                    launches add fill and drain time).                                          */
    MIFWI_FLAG_PML = 256,        /* 2-D forward modelling with a perfectly matched layer instead of the sponge: the
                                    second-order formulation of Grote and Sim with two auxiliary fields on the
                                    staggered grid, discretised as seismic/abc_methods/03_pml.ipynb (cells 2, 60,
                                    75-85) of the reference does it, here with the Laplacian at the plan's space
                                    order and the layer on all four sides (mifwi_set_pml).  `damp` of mifwi_set_model
                                    is ignored.  Untiled kernels; mifwi_forward without saving only (the adjoint and
                                    the gradient of this scheme are not part of the library).  The reference holds
                                    no result of the hot path with this boundary: parity is against the CPU
                                    restatement only.                                                           */
    MIFWI_FLAG_SAVE_F16 = 64     /* keep the saved forward wavefield (mifwi_forward(save=1)) in IEEE
                                    half precision, scaled per shot from the injected source terms:
                                    half the memory (nt levels of a 592^3 grid fit one GPU) and half
                                    the saved-level traffic of the gradient.  float32 plans on the tiled
                                    kernels only (ignored otherwise, and when the levels do not fit
                                    even so and checkpointing takes over).  The time stepping itself
                                    stays float32; the gradient changes by ~5e-5 relative (rounding of
                                    the saved levels to 11 bits).  Device-side counterpart of the
                                    compressed wavefield serialisation of
                                    seismic/tutorials/12_time_blocking.ipynb cells 15-37.          */
};

typedef struct mifwi_plan mifwi_plan; /* opaque */

/*
 * Discretisation of one acquisition: what `SeismicModel` + `AcquisitionGeometry` +
 * `AcousticWaveSolver.__init__` fix in the reference (seismic/model.py:91-124,
 * seismic/utils.py:60-95, seismic/acoustic/wavesolver.py:28-46).
 */
typedef struct mifwi_config {
    int32_t ndim;          /* 1, 2 or 3                                                    */
    int32_t shape[3];      /* padded grid shape, shape[0..ndim)                            */
    double spacing[3];     /* grid spacing per dimension (m)                               */
    double origin[3];      /* coordinates of padded-grid point 0 (origin - nbl*spacing)    */
    int32_t space_order;   /* even, 2..16; 18 and 20 on the untiled kernels (any dtype)    */
    int32_t nt;            /* time samples; steps run time = 1 .. nt-2                     */
    double dt;             /* time step (ms)                                               */
    int32_t dtype;         /* MIFWI_F32 or MIFWI_F64: element type of every array argument */
    int32_t nsrc;          /* point sources per shot                                       */
    int32_t nrec;          /* receivers per shot                                           */
    int32_t max_shots;     /* shots propagated side by side in one call (batch capacity)   */
    int32_t device;        /* HIP device ordinal, or -1 for the current device             */
    int32_t flags;         /* MIFWI_FLAG_*                                                 */
} mifwi_config;

/* Timing/traffic summary of the last propagation call; stands in for Devito's
 * PerformanceSummary returned by Operator.apply (seismic/acoustic/wavesolver.py:112-114). */
typedef struct mifwi_stats {
    double kernel_ms;       /* GPU time of the time loops of the last call (HIP events on the plan's stream) */
    double points;          /* grid-point updates performed                                   */
    double gpointss;        /* points / kernel time / 1e9                                     */
    double algorithmic_bytes; /* points * bytes-per-update of that operator (20 or 32)       */
    double fwd_ms;          /* GPU time of the last forward time loop                         */
    double bwd_ms;          /* GPU time of the last backward (adjoint / gradient) time loop   */
    int32_t launches;       /* kernel launches of the last call                               */
    int32_t fwd_launches;   /* kernel launches of the last forward time loop                  */
    int32_t bwd_launches;   /* kernel launches of the last backward time loop                 */
    int32_t kernel_kind;    /* 0 generic, 1 tiled 2-D, 2 2.5-D 3-D, 3 whole 2-D loop per launch, 5 whole 2-D TTI loop per launch */
    int64_t device_bytes;   /* device memory currently held by the plan                       */
    int32_t ckpt_steps;     /* leading time steps whose wavefield is kept as checkpoints and propagated a second
                               time by the gradient (0: every level is stored)                 */
    int32_t ckpt_segment;   /* steps per checkpoint segment (0: no checkpointing)             */
    int32_t fwd_kernel_kind; /* kernel_kind of the last forward time loop                      */
    int32_t bwd_kernel_kind; /* kernel_kind of the last backward time loop                     */
    int32_t persist_fallbacks; /* time loops of this plan, since its creation, that were started as one launch
                               (kernel_kind 3), did not complete -- a wait ran out, or a tile had not counted all
                               its steps at the end -- and were run again with a kernel per step; 0 means every
                               loop reported as kind 3 really ran that way                         */
    int32_t reserved_;
} mifwi_stats;

MIFWI_API int mifwi_version(void);
MIFWI_API const char *mifwi_last_error(void);

/* Replaces: AcousticWaveSolver.__init__ + the (memoised) Operator construction,
 * seismic/acoustic/wavesolver.py:28-74. */
MIFWI_API int mifwi_plan_create(const mifwi_config *cfg, mifwi_plan **out);
MIFWI_API int mifwi_plan_destroy(mifwi_plan *plan);

/* Use an existing hipStream_t (e.g. the caller's framework stream) for all work of the plan. */
MIFWI_API int mifwi_set_stream(mifwi_plan *plan, void *hip_stream);
MIFWI_API int mifwi_synchronize(mifwi_plan *plan);

/* Replaces: the `vp` and `damp` arguments of every Operator.apply
 * (seismic/acoustic/wavesolver.py:109,112; seismic/model.py:13-51,395-400).  vp in km/s and
 * damp are padded-grid arrays (damp NULL: no damping); m = 1/vp^2 is formed on the device.  Host arrays have been
 * read when the call returns.  DEVICE arrays are read by a kernel queued on the plan's stream: leave them
 * unchanged (and allocated) until the next call on the plan that waits -- any modelling call, or mifwi_synchronize. */
MIFWI_API int mifwi_set_model(mifwi_plan *plan, const void *vp, const void *damp);

/* Replaces: the src_coords / rec_coords arguments (SparseTimeFunction.coordinates,
 * seismic/utils.py:147-186).  src_coords is (nshots, nsrc, ndim).  rec_coords is
 * (nrec, ndim) shared by all shots when rec_per_shot == 0, else (nshots, nrec, ndim).
 * Physical coordinates in the units of `spacing`, element type = plan dtype. */
MIFWI_API int mifwi_set_geometry(mifwi_plan *plan, int32_t nshots, const void *src_coords,
                       const void *rec_coords, int32_t rec_per_shot);

/* Replaces: AcousticWaveSolver.forward -> op_fwd(save).apply(src, rec, u, vp, dt)
 * (seismic/acoustic/wavesolver.py:76-114; seismic/acoustic/operators.py:98-140).
 * wavelet: (nt, nsrc) shared by all shots when wavelet_per_shot == 0, else (nshots, nt, nsrc).
 * rec_out: (nshots, nt, nrec) or NULL.  save != 0 keeps every time level on the device for
 * mifwi_gradient / mifwi_read_wavefield. */
MIFWI_API int mifwi_forward(mifwi_plan *plan, int32_t nshots, const void *wavelet, int32_t wavelet_per_shot,
                  void *rec_out, int32_t save);

/* Replaces: AcousticWaveSolver.jacobian (= born) -> op_born().apply(dm, src, rec, u, U, vp, dt)
 * (seismic/acoustic/wavesolver.py:207-242; seismic/acoustic/operators.py:228-273): linearised
 * modelling.  The background field u is driven by the wavelet, the scattered field U by
 * -dm * u.dt2; rec_out (nshots, nt, nrec) receives the samples of U.  dm: (grid), one model
 * perturbation (of 1/vp^2) shared by all shots.  Afterwards mifwi_read_wavefield(which = 1) gives
 * the last three levels of u and which = 2 those of U. */
MIFWI_API int mifwi_born(mifwi_plan *plan, int32_t nshots, const void *wavelet, int32_t wavelet_per_shot,
               const void *dm, void *rec_out);

/* Replaces: AcousticWaveSolver.adjoint -> op_adj().apply(srca, rec, v, vp, dt)
 * (seismic/acoustic/wavesolver.py:116-151; seismic/acoustic/operators.py:143-180).
 * rec_in (nshots, nt, nrec); srca_out (nshots, nt, nsrc). */
MIFWI_API int mifwi_adjoint(mifwi_plan *plan, int32_t nshots, const void *rec_in, void *srca_out);

/* Replaces: AcousticWaveSolver.jacobian_adjoint (= gradient) -> op_grad().apply(rec, grad, v, u, vp, dt)
 * (seismic/acoustic/wavesolver.py:153-205; seismic/acoustic/operators.py:183-225), using the
 * wavefield saved by the preceding mifwi_forward(save=1) (or mifwi_write_wavefield).
 * residual (nshots, nt, nrec).  grad (nshots, grid): per-shot gradients on the padded grid;
 * with accumulate != 0 the result is ADDED to the caller's array (Devito's `Inc`), else stored.
 * illum: NULL or (nshots, grid), receives sum_t u^2 (fwi.py:170 of the reference). */
MIFWI_API int mifwi_gradient(mifwi_plan *plan, int32_t nshots, const void *residual, void *grad,
                   int32_t accumulate, void *illum);

/* Fused per-shot objective of fwi.py:131-173 with the L2 misfit (misfit/misfit.py:5-9):
 * forward(save) -> residual = syn - obs -> fval = 0.5*||residual||^2 -> gradient, all on the
 * device.  obs (nshots, nt, nrec); fval (nshots) doubles; grad/illum as in mifwi_gradient
 * (stored, not accumulated; either may be NULL, grad == NULL skips the backward pass);
 * residual_out / syn_out NULL or (nshots, nt, nrec). */
MIFWI_API int mifwi_l2_misfit_gradient(mifwi_plan *plan, int32_t nshots, const void *wavelet,
                             int32_t wavelet_per_shot, const void *obs, double *fval, void *grad,
                             void *illum, void *residual_out, void *syn_out);

/* Layer of a MIFWI_FLAG_PML plan: `npml` grid cells on each side of both dimensions, damping functions
 * zeta_d = quibar * (a - sin(2 pi a) / (2 pi)), a = depth into the layer / layer width (fdamp of
 * seismic/abc_methods/03_pml.ipynb cell 60, where quibar = 0.05), evaluated at the nodes and at the
 * half-way (staggered) points.  Must be called before mifwi_forward. */
MIFWI_API int mifwi_set_pml(mifwi_plan *plan, int32_t npml, double quibar);

/* The same layer with the layout of the reference's notebook spelled out (seismic/abc_methods/03_pml.ipynb):
 * sides: bit 0 / 1 = layer at the first / last nodes of x, bit 2 / 3 = of z (the notebook: 1 | 2 | 8, none at the top,
 * cell 58; mifwi_set_pml: all four).  edges = 1: the damped update and the auxiliary fields only at the nodes of the
 * layer regions (its subdomains d1-d3, cells 20-22 and 76-81), the plain wave equation elsewhere, the staggered
 * profiles' last entry repeated (cells 72-74) and, after the source term of every step, the boundary rows of cell 83
 * (u = 0 on the first and last x node and on the last z node, u[x, 0] = u[x, 1]); edges = 0: fields zero outside the
 * grid as everywhere in this library.  vp_staggered: NULL (c^2 at a staggered point = mean of c^2 over the cell's
 * corners inside the grid) or the caller's own velocity at the staggered points over the padded grid (the notebook's
 * vel1, cells 26 and 51-53). */
MIFWI_API int mifwi_set_pml_layout(mifwi_plan *plan, int32_t npml, double quibar, int32_t sides, int32_t edges,
                                   const void *vp_staggered);

/* Replaces: least_square(syn, obs) of misfit/misfit.py:5-9 as fwi.py:151 calls it between the forward and the
 * gradient pass, on gathers that are already on the device: residual_out = syn - obs and
 * fval[shot] = 0.5 * sum residual^2 (float64, fixed summation order: the same bits on every call).
 * syn, obs, residual_out: (nshots, nt, nrec), host or device; residual_out may be NULL; fval (nshots) doubles. */
MIFWI_API int mifwi_l2_misfit(mifwi_plan *plan, int32_t nshots, const void *syn, const void *obs, void *residual_out,
                    double *fval);

/* Replaces: the shot loop of fwi_obj_multi (fwi.py:184-193) over what fwi_obj_single returns per shot
 * (fwi.py:164-171): crop the padded-grid gradient and illumination to the physical grid, scale by the shot's
 * source/receiver mute factor (fix_source_illumination, fwi.py:104-129, a table the caller builds once per
 * survey) and add the shots up -- in float64, in shot order, product rounded before the sum, which is the
 * arithmetic of `grad += crop(g) * mute` on float64 numpy arrays.  grad, illum: DEVICE arrays (nshots, *shape)
 * in the plan's dtype as mifwi_gradient writes them (illum and il_sum may both be NULL); mute: DEVICE doubles
 * (nshots, *phys_shape) or NULL for no muting; lo / phys_shape: first index and extent of the physical grid
 * inside the padded grid per dimension; g_sum, il_sum (*phys_shape) doubles, host or device. */
MIFWI_API int mifwi_sum_shots(mifwi_plan *plan, int32_t nshots, const void *grad, const void *illum,
                              const double *mute, const int32_t *lo, const int32_t *phys_shape, double *g_sum,
                              double *il_sum);

/* Replaces: the illumination preconditioner and the mask of fwi_obj_multi (fwi.py:200-204):
 * out = g_sum / sqrt(il_sum + 1e-30) * mask over n doubles.  g_sum, il_sum, mask: DEVICE arrays; il_sum NULL skips
 * the division, mask NULL the mask; out: host or device (may alias g_sum). */
MIFWI_API int mifwi_precondition(mifwi_plan *plan, int64_t n, const double *g_sum, const double *il_sum,
                                 const double *mask, double *out);

/* No counterpart in the reference.  For callers that cannot allocate device memory: code under HIP graph capture,
 * tight loops that must not touch the allocator, harnesses that forbid hipMalloc.  The same computation and the
 * same bits as mifwi_l2_misfit, without a plan: syn, obs, residual_out are (nshots, n_per_shot) DEVICE arrays of
 * `dtype` (MIFWI_F32 / MIFWI_F64) on the current device, fval (nshots) device doubles; the kernels are enqueued on
 * `hip_stream` (NULL = the default stream) and the call returns without waiting.  It calls no hipMalloc / hipFree:
 * the partial sums live in a fixed array inside the library (128 KiB per GPU, held until the process ends), which
 * limits a call to 256 shots (MIFWI_EINVAL beyond).  All calls on one GPU share that array: calls that overlap in
 * time on one GPU (different streams) race; serialise them.  The array used is the CURRENT device's: every array
 * (and the stream) must belong to the current device, MIFWI_EINVAL otherwise.  The first call on a device looks the
 * array's address up (hipGetSymbolAddress, which may load the code object): that must happen outside stream capture,
 * so before the first captured call on a device make one eager call there, or create any plan on it
 * (mifwi_plan_create does the same look-up). */
MIFWI_API int mifwi_l2_misfit_noalloc(int32_t dtype, int32_t nshots, int64_t n_per_shot, const void *syn,
                            const void *obs, void *residual_out, double *fval, void *hip_stream);

/* ---- TTI media, 'centered' kernel (seismic/tti/operators.py:198-313,430-519 and
 * seismic/tti/wavesolver.py:80-215 of the reference) -------------------------------------------
 * mifwi_tti_set_model replaces the epsilon / delta / theta / phi operator arguments
 * (wavesolver.py:153-159): arrays over the padded grid like vp (mifwi_set_model still supplies vp and
 * damp); phi may be NULL (2-D, or no azimuth).  space_order must be divisible by 4; 2-D or 3-D.
 * mifwi_tti_forward replaces op_fwd('centered').apply (wavesolver.py:160-166): the wavelet drives
 * both fields, rec_out (nshots, nt, nrec) samples u + v; save != 0 keeps every level of u and v on the
 * device (TimeFunction(save=nt), wavesolver.py:127-137) for mifwi_tti_gradient -- 2 * nt fields per shot, or,
 * when those do not fit, pairs of levels every K steps from which the gradient call propagates one segment at a
 * time again (same bits; mifwi_read_wavefield which = 7, 8 is refused then).  mifwi_tti_adjoint replaces op_adj().apply (wavesolver.py:208-215):
 * rec_in (nshots, nt, nrec) -> srca_out (nshots, nt, nsrc).  The last three levels of (u, v) and (p, r)
 * are readable through mifwi_read_wavefield which = 3, 4 and 5, 6, the saved levels of u and v through
 * which = 7 and 8. */
MIFWI_API int mifwi_tti_set_model(mifwi_plan *plan, const void *epsilon, const void *delta, const void *theta,
                        const void *phi);
MIFWI_API int mifwi_tti_forward(mifwi_plan *plan, int32_t nshots, const void *wavelet, int32_t wavelet_per_shot,
                      void *rec_out, int32_t save);
MIFWI_API int mifwi_tti_adjoint(mifwi_plan *plan, int32_t nshots, const void *rec_in, void *srca_out);

/* Replaces: AnisotropicWaveSolver.jacobian -> op_jac().apply(dm, u0, v0, du, dv, src, rec, ...)
 * (seismic/tti/wavesolver.py:219-280; seismic/tti/operators.py:529-587).  dm over the padded grid;
 * rec_out (nshots, nt, nrec) receives the samples of du + dv. */
MIFWI_API int mifwi_tti_born(mifwi_plan *plan, int32_t nshots, const void *wavelet, int32_t wavelet_per_shot,
                   const void *dm, void *rec_out);

/* Replaces: AnisotropicWaveSolver.jacobian_adjoint -> op_jacadj().apply(rec, dm, u0, v0, du, dv, ...)
 * (seismic/tti/wavesolver.py:282-357; seismic/tti/operators.py:590-639), using the levels of u0 and v0
 * kept on the device by the last mifwi_tti_forward(save = 1).  residual (nshots, nt, nrec);
 * grad (nshots, *shape_pad) per shot, overwritten or (accumulate != 0) added to. */
MIFWI_API int mifwi_tti_gradient(mifwi_plan *plan, int32_t nshots, const void *residual, void *grad,
                       int32_t accumulate);

/* Access to wavefield time levels of one shot, `u.data[first:first+count]` in the reference
 * (TimeFunction(save=nt), seismic/acoustic/wavesolver.py:104-106).  which = 0: saved forward
 * field (level index = time).  which = 1: the three rotating buffers of the last non-saving
 * forward run, which = 2: those of the last adjoint/gradient run (level index = time % 3).
 * which = 3..8: TTI fields, see above. */
MIFWI_API int mifwi_read_wavefield(mifwi_plan *plan, int32_t which, int32_t shot, int32_t first, int32_t count,
                         void *out);
MIFWI_API int mifwi_write_wavefield(mifwi_plan *plan, int32_t shot, int32_t first, int32_t count, const void *in);

MIFWI_API int mifwi_get_stats(mifwi_plan *plan, mifwi_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* MIFWI_H */
