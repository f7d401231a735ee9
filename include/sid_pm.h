/*
 * sid_pm.h - C ABI of the MI355X (gfx950) pattern-matching operator.
 *
 * Drop-in boundary for ONE hot path of nansencenter/sea_ice_drift (v0.7.1): the
 * per-grid-point rotated-template maximum-cross-correlation sweep that the reference
 * runs through multiprocessing.Pool.  Everything here is plain pointers and sizes; no
 * torch / HIP types appear in a signature (streams travel as void*).
 *
 * Reference interfaces replaced (file:line in /root/reference/sea_ice_drift/):
 *
 *   sid_pm_batch        pmlib.py:436-448 + :462   Pool(initargs=(c1,r1,c2fg,r2fg,brd,img1,img2,
 *                                                 img_size,alpha0,kwargs)).map(use_mcc_mp, range(N))
 *                                                 -> np.array(results) of shape (N,5)
 *   per point           pmlib.py:176-212 use_mcc, :117-174 rotate_and_match,
 *                       :89-115 get_template, :156 cv2.matchTemplate(TM_CCOEFF_NORMED),
 *                       :36-59 get_hessian   (all fused into one HIP kernel)
 *   sid_pm_create/...   pmlib.py:430-434 _init_pool (shared read-only state of the workers):
 *                       here the state is device-resident and owned by a handle, so the
 *                       operator is re-entrant (the reference's module globals, :33-34, are not)
 *
 * Ownership: every pointer argument is caller-owned and is not retained past the call,
 * except device image pointers given to sid_pm_bind_pair, which must stay valid until
 * the next bind/upload or sid_pm_destroy.  Device buffers created by the library belong
 * to the handle and are freed by sid_pm_destroy.
 *
 * Errors: functions return SID_PM_OK (0) or a negative code; sid_pm_last_error() gives a
 * thread-local message.  Per-point failure is never an error: exactly like the reference
 * (pmlib.py:152-154) a point whose rotated template touches a 0 pixel yields NaN x 5
 * (and -1 in out_ij); so does a search window that is not wholly inside image 2.
 *
 * Numerics: see DESIGN.md "NCC specification".  Integer outputs (peak row/col, angle
 * index) are exact; r is the float32 the specification defines; h is float32 arithmetic
 * as in NumPy, agreeing with the CPU oracle to 1e-5.
 */
#ifndef SID_PM_H
#define SID_PM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SID_PM_ABI_VERSION 4

/* return codes */
#define SID_PM_OK               0
#define SID_PM_ERR_ARG         -1   /* bad argument (null pointer, n_angles < 1, ...)          */
#define SID_PM_ERR_HIP         -2   /* a HIP runtime call failed                               */
#define SID_PM_ERR_NOMEM       -3   /* host or device allocation failed                        */
#define SID_PM_ERR_UNSUPPORTED -4   /* option or size outside what the kernels implement       */
#define SID_PM_ERR_NODEVICE    -5   /* no gfx950 device visible                                */
#define SID_PM_ERR_STATE       -6   /* call order violated (run before set_points, ...)        */

/* flags = the reference's boolean kwargs (pmlib.py:36, :121) */
#define SID_PM_HES_NORM 1u          /* hes_norm=True  (default in the reference)               */
#define SID_PM_HES_SMTH 2u          /* hes_smth=True  (gaussian_filter sigma=1 before Hessian) */
#define SID_PM_MCC_NORM 4u          /* mcc_norm=True                                           */
#define SID_PM_ROT_ORDER1 8u        /* rot_order=1 (pmlib.py:89,112-113): templates sampled bilinearly with scipy's arithmetic
                                     * (float64 taps, uint8 output rounding); default rot_order=0 = nearest neighbour.   */
#define SID_PM_ROT_ORDER(n) (((uint32_t)(n) & 7u) << 3)   /* rot_order = n, 0..5, in flag bits 3..5 (SID_PM_ROT_ORDER(1) == SID_PM_ROT_ORDER1).
                                     * Orders 2..5 are scipy's spline interpolation: the WHOLE image 1 goes through scipy's recursive
                                     * B-spline prefilter once per pair (float64 coefficients, 8 B per pixel x 2 buffers of device
                                     * memory, kept with the handle until the pair changes), then every template sample is the
                                     * tensor product of n + 1 weights per axis - scipy's arithmetic operation for operation.      */

typedef struct sid_pm_ctx sid_pm_ctx;

int         sid_pm_abi_version(void);
const char *sid_pm_strerror(int code);
const char *sid_pm_last_error(void);
int         sid_pm_device_count(int *count);

/*
 * One-shot, host buffers in / host buffers out: the whole Pool.map seam.
 *   img1,img2        uint8 images, row-major, `stride` bytes between rows (>= cols)
 *   c1,r1            [n] template centre on image 1 (float pixel coordinates)
 *   c2fg,r2fg        [n] first-guess centre on image 2
 *   border           [n] search half-width around the first guess (pixels)
 *   img_size         template side s (reference default 35; 34 in the benchmark), 2..255.  Sides up to 64 with a
 *                    search window of at most 257 px run one workgroup per point; larger sides and larger
 *                    windows (any border, any rectangular shape) run the large-window pipeline - the placements of
 *                    ONE point tiled over the whole device, the NCC matrices of all angles in global memory
 *   alpha0           scene rotation in degrees (pmlib.py:428)
 *   angles           [n_angles] trial angles in degrees, in the reference's list order
 *   rot              REQUIRED [n_angles][4] = {cos a, sin a, tcT0, tcT1} with
 *                    a = radians(angle - alpha0), tcT = [tc,tc].dot([[cos,-sin],[sin,cos]]),
 *                    tc = int(s/2.)+1 (pmlib.py:105-110), as the caller's NumPy computed them.  The sample
 *                    positions of a template are floor(x + 0.5) of float64 coordinates built from these four
 *                    numbers; a cos / sin that differs from NumPy's in the last bit can tip one at a rounding
 *                    tie, so the library does not derive them itself: NULL is SID_PM_ERR_ARG (ABI 4; ABI <= 3
 *                    fell back to libm).  INTEGRATION.md section 2 has the three NumPy lines.
 *   out              [n][5] float64: c2, r2, angle, r, h            (pmlib.py:212)
 *   out_ij           optional [n][3] int32: peak row, peak col, angle index (-1 = NaN point)
 */
int sid_pm_batch(const uint8_t *img1, int64_t rows1, int64_t cols1, int64_t stride1,
                 const uint8_t *img2, int64_t rows2, int64_t cols2, int64_t stride2,
                 const double *c1, const double *r1, const double *c2fg, const double *r2fg,
                 const double *border, int64_t n, int img_size, double alpha0,
                 const double *angles, const double *rot, int n_angles, uint32_t flags,
                 double *out, int32_t *out_ij);

/* ---- device-resident form: upload once, run many (bench.py, pair streaming, RCCL gather) ---- */

int  sid_pm_create(int device, sid_pm_ctx **ctx);
void sid_pm_destroy(sid_pm_ctx *ctx);

/* HIP stream (hipStream_t as void*) all later work of this handle is enqueued on; NULL = default */
int sid_pm_set_stream(sid_pm_ctx *ctx, void *hip_stream);

/* Copy a host image pair into handle-owned device buffers.  The copy is enqueued on the
 * handle's own copy stream: it overlaps the kernels of a run in flight (truly asynchronous
 * when the host buffers are pinned - hipHostRegister / hipHostMalloc / torch pin_memory -
 * and staged synchronously otherwise); the host buffers must stay valid until the run that
 * uses the pair has been synchronised.  `slot` 0/1 selects one of two device pairs so that
 * pair k+1 uploads while pair k is matched (BASELINE config 5); sid_pm_select_pair picks the
 * one the next run uses.  Ordering is handled inside: a run waits for the upload of its
 * slot, an upload waits for the runs that still read its slot. */
int sid_pm_upload_pair(sid_pm_ctx *ctx, int slot,
                       const uint8_t *img1, int64_t rows1, int64_t cols1, int64_t stride1,
                       const uint8_t *img2, int64_t rows2, int64_t cols2, int64_t stride2);
int sid_pm_select_pair(sid_pm_ctx *ctx, int slot);

/* Borrow images already in device memory (e.g. torch uint8 tensors). */
int sid_pm_bind_pair(sid_pm_ctx *ctx,
                     const uint8_t *d_img1, int64_t rows1, int64_t cols1, int64_t stride1,
                     const uint8_t *d_img2, int64_t rows2, int64_t cols2, int64_t stride2);

/* Host vectors of the n points + the sweep parameters.  Validates, orders the points by
 * search-window size (largest first, for load balance), groups them by LDS footprint and
 * uploads them; the points stay resident until the next call.
 * Device scratch reserved here (grow-only, freed by sid_pm_destroy): 48 B + 52 B per point for records and results; for the
 * launches that keep per-placement tables in global memory (template sides 34 / 35) 8192 RECYCLED blocks of the largest such
 * table, whatever n - 8 B per placement of the search window (14 KB at border 20, 83 KB at border 50), 16 / 32 B per placement
 * more for angle sets of at most 3 / 7 angles (they keep the sweep's accumulators as well): 0.7 - 2 GB; beyond border 68 a block
 * of its own per point for every table of the point (0.6 MB at border 111).  Points whose launch keeps its sums in LDS get none;
 * points beyond one workgroup's LDS (SID_PM_CLASS_LARGE) share the scratch of the large-window pipeline (sid_pm_rotate_and_match).
 * rot_order 2..5: + 16 B per pixel of image 1 (spline coefficients, at the first run on a pair) + n x n_angles x img_size^2 B
 * (pre-sampled templates).  SID_PM_ERR_NOMEM with the numbers when the device cannot hold it. */
int sid_pm_set_points(sid_pm_ctx *ctx, const double *c1, const double *r1, const double *c2fg,
                      const double *r2fg, const double *border, int64_t n, int img_size,
                      double alpha0, const double *angles, const double *rot, int n_angles,
                      uint32_t flags);

/* Optional: have the kernels write into caller-owned device arrays ([n][5] float64 and
 * [n][3] int32, e.g. torch tensors that an RCCL gather will read) instead of the handle's
 * own.  Call after set_points; the binding is dropped by the next set_points.  NULL, NULL
 * restores the handle's buffers.  Any device-visible pointer serves: bound to pinned host
 * memory (hipHostMalloc) the kernels write their 52 bytes per point straight to the host and
 * sid_pm_run + sid_pm_sync is the whole step (no copy after the kernels; do not call
 * sid_pm_fetch then - it copies FROM the bound arrays). */
int sid_pm_bind_results(sid_pm_ctx *ctx, double *d_out, int32_t *d_out_ij);

/* Enqueue the kernels for the resident points on the resident pair (asynchronous).  The launches of a run - one per launch class
 * - follow each other on the handle's stream; a SHORT run (all launches together at most 16 rounds of workgroups: a small batch, or
 * a rank's shard of an N-GPU run) puts them side by side on three more non-blocking streams the handle owns (created by
 * sid_pm_create) and joins them on the handle's stream before it returns to it, so the caller sees one stream either way. */
int sid_pm_run(sid_pm_ctx *ctx);
/* Wait for the stream; then sid_pm_check. */
int sid_pm_sync(sid_pm_ctx *ctx);
/* Points with a valid search window that a launch of the runs since the last check refused (the LDS layout the kernel
 * computes for the point did not fit the launch the host's classification sized for it - a library bug, not a property of
 * the data): SID_PM_ERR_STATE with the count in sid_pm_last_error(), and the counter is cleared.  Such a point would
 * otherwise be indistinguishable from the reference's legitimate NaN row (zero pixel in the template, pmlib.py:152-154).
 * Reads a pinned host word the kernels write: call it once the launch stream is synchronised, by whatever means
 * (sid_pm_sync and sid_pm_fetch do). */
int sid_pm_check(sid_pm_ctx *ctx);
/* Copy results to the host (waits for the stream; then sid_pm_check).  out_ij may be NULL. */
int sid_pm_fetch(sid_pm_ctx *ctx, double *out, int32_t *out_ij);
/* Device pointers of the result arrays ([n][5] float64, [n][3] int32, original point
 * order) for device-side consumers such as an RCCL gather.  Valid until set_points/destroy. */
int sid_pm_device_results(sid_pm_ctx *ctx, double **d_out, int32_t **d_out_ij);

/* Exchange step of the N-GPU path (what Pool.map's result list is to the reference, pmlib.py:444,462): `world` blocks of m
 * rows [m x 5 float64 | m x 3 int32] - every rank's kernels wrote one in place, one RCCL gather stacked them on the
 * destination rank - are put back into the original point order in one kernel: out[i] / out_ij[i] = row d_perm[i] of the
 * stacked blocks (m even).  `out` / `out_ij` may be pinned host memory (any device-visible pointer): the results then
 * reach the host without a copy after the kernel.  out_ij may be NULL.  Asynchronous on `hip_stream`. */
int sid_pm_unpermute(const void *d_stack, int64_t world, int64_t m, const int32_t *d_perm, int64_t n,
                     double *out, int32_t *out_ij, void *hip_stream);

/* Work accounting of the resident points, for roofline reporting (DESIGN.md "Measurement"):
 *   info[0] = kernel launches per run          info[1] = valid points
 *   info[2] = algorithmic MACs  sum K*Rh*Rw*s*s   info[3] = algorithmic HBM bytes
 *   info[4] = max dynamic LDS bytes per block  info[5] = reserved                           */
int sid_pm_work_info(sid_pm_ctx *ctx, double info[6]);

/* Estimated cost (nanoseconds on one MI355X) of grid points with the given search borders, derived from what the kernel
 * executes for each - matrix instructions of the sweep and of the winner's matrix, placements, residency class of the LDS
 * footprint - for cutting points into shards of equal cost over several GPUs.  `flags` = the flags the run will use (they
 * decide the LDS layout of the Hessian and with it the class borders).  Pure host arithmetic (no device needed).
 * Replaces the role of `threads`-sized chunks of the reference's Pool.map (pmlib.py:442-444).
 * ABI 3: the `flags` argument is new (ABI 2 classified every run as flags = 0).                                        */
int sid_pm_estimate_cost(const double *border, int64_t n, int img_size, int n_angles, uint32_t flags, double *cost_ns);
/* Launch class of a point of that border.  Points of equal value share a launch.  Host arithmetic as well.
 *   value & SID_PM_CLASS_PER_CU  workgroups per CU (1 .. 4) of its launch - a launch has 256 x that many points in flight,
 *                                which prices the tail of a short launch when the points are cut into shards;
 *   SID_PM_CLASS_GS              the launch keeps its per-placement sums in global memory (a launch of its own);
 *   SID_PM_CLASS_BIG             search borders beyond 68 px: every per-placement table in global memory;
 *   SID_PM_CLASS_W3B             reserved (second launch of the three-wavefront class; not produced by this build).
 * ABI 1 returned the workgroups per CU alone; consumers must mask with SID_PM_CLASS_PER_CU.                             */
#define SID_PM_CLASS_PER_CU 15
#define SID_PM_CLASS_GS     16
#define SID_PM_CLASS_BIG    32
#define SID_PM_CLASS_W3B    64
#define SID_PM_CLASS_LARGE  128  /* beyond one workgroup's LDS (border > 111 px at 34 / 35 px, template side > 64): the point runs the
                                  * large-window pipeline, in batches of up to 64 such points behind the launches of the others
                                  * (sid_pm_run then waits for the stream once before it enqueues them)                     */
int sid_pm_estimate_residency(const double *border, int64_t n, int img_size, int n_angles, uint32_t flags, int32_t *launch_class);

/* Estimated kernel time (nanoseconds) of ONE sid_pm_run over points with these borders: the point costs per launch class, the
 * tail of every launch (its last, half-empty round of workgroups), the launcher's own rule for putting the launches of a short
 * run side by side (at most 16 rounds of workgroups together), the large-window points behind them.  What the cuts of an N-GPU
 * split are made with (sea_ice_drift_amd/dist.py): launcher and estimate share the rule, they cannot drift apart.  Host arithmetic. */
int sid_pm_estimate_run_time(const double *border, int64_t n, int img_size, int n_angles, uint32_t flags, double *time_ns);

/* ---- the per-point functions of the reference as calls of their own ---- */

/* rotate_and_match (pmlib.py:117-174): the rotated templates of side img_size around (c1, r1) on image 1 of the handle's
 * current pair against the window rows [win_row0, win_row0 + win_rows) x columns [win_col0, win_col0 + win_cols) of its
 * image 2 - ANY rectangular shape, up to the whole image (the reference's own test passes the whole of image 2,
 * tests.py:336-337).  A caller with the reference's arguments (img1, ..., image2, ...) uploads (img1, image2) as the pair
 * and passes the window (0, 0, rows2, cols2).
 *   out5            dc, dr, best_a, best_r, best_h                    (pmlib.py:174; dc = col - (win_cols - s) / 2.)
 *   ij3             optional: peak row, peak column, index of the winning angle (-1: NaN point)
 *   ccm             optional [win_rows - s + 1][win_cols - s + 1] float32 = best_result, ccm_cap = its capacity in floats
 *   best_template   optional [s][s] uint8
 * A template that touches a 0 pixel yields NaN x 5 in out5 and leaves ccm / best_template untouched (pmlib.py:152-154).
 * Fewer than two placements along an axis is SID_PM_ERR_ARG (np.gradient raises there).  `rot` as in sid_pm_batch (required).
 * Device scratch: 4 B x placements x n_angles + 24 B x placements (kept with the handle). */
int sid_pm_rotate_and_match(sid_pm_ctx *ctx, double c1, double r1, int img_size,
                            int64_t win_row0, int64_t win_col0, int64_t win_rows, int64_t win_cols,
                            double alpha0, const double *angles, const double *rot, int n_angles, uint32_t flags,
                            double out5[5], int32_t ij3[3], float *ccm, int64_t ccm_cap, uint8_t *best_template);

/* get_template (pmlib.py:89-115): the s x s uint8 template around (c, r) of a host image, rot4 = {cos a, sin a, tcT0, tcT1}
 * as above; rot_order 0 or 1.  Only the part of the image the samples can touch travels to the device. */
int sid_pm_get_template(int device, const uint8_t *img, int64_t rows, int64_t cols, int64_t stride, double c, double r,
                        const double rot4[4], int img_size, int rot_order, uint8_t *out);

/* get_hessian (pmlib.py:36-59) of a host float32 matrix [rows][cols]: np.gradient twice along each axis, np.hypot, and - with
 * SID_PM_HES_NORM - (hes - median) / std; SID_PM_HES_SMTH: gaussian_filter(ccm, 1) first.  hes [rows][cols] float32. */
int sid_pm_get_hessian(int device, const float *ccm, int64_t rows, int64_t cols, uint32_t flags, float *hes);

/* ---- diagnostics used by the parity tests ---- */

/* Intermediate results of one point: rotated templates [n_angles][s][s] (uint8), the NCC
 * matrix of the winning angle and its raw Hessian magnitude [rh][rw] (float32, caller
 * provides capacity `cap` floats each), shape in rh_rw[2]; phase_cycles = shader-clock stamps at
 * the kernel's phase boundaries (one workgroup on an idle device).  Any output pointer may be NULL. */
int sid_pm_debug_point(sid_pm_ctx *ctx, double c1, double r1, double c2fg, double r2fg,
                       double border, int img_size, double alpha0, const double *angles,
                       const double *rot, int n_angles, uint32_t flags,
                       uint8_t *templates, float *ccm, float *hes, int64_t cap, int32_t rh_rw[2],
                       double out5[5], int32_t ij3[3], int64_t phase_cycles[32]);

/* y[i] = 1.0 / sqrt(x[i]) evaluated on the device in IEEE double, the one transcendental
 * step of the NCC specification; lets a test pin device vs host rounding. */
int sid_pm_debug_rsqrt(sid_pm_ctx *ctx, const double *x, double *y, int64_t n);

/* The kernel evaluates the NCC matrix of the winning angle through a shortened form of the specification's
 * normalisation (no IEEE square root / division; values whose float32 rounding could differ take the
 * specification's own route).  This runs both forms on `evaluations` pseudo-random sums on the device:
 * counts[0] = evaluations done, counts[1] = results that differ (must be 0), counts[2] = evaluations that
 * needed the specification's route. */
int sid_pm_debug_ncc_selftest(sid_pm_ctx *ctx, uint64_t seed, int64_t evaluations, int img_size, uint64_t counts[3]);

/* The Hessian magnitude hypotf(d2x, d2y) = (float)sqrt((double)x*x + (double)y*y) (NumPy's float32 np.hypot,
 * pmlib.py:55) is evaluated in the kernel through a shortened square root with a guard; this runs both routes on
 * `evaluations` pseudo-random float32 pairs on the device: counts[0] = evaluations, counts[1] = results that differ. */
int sid_pm_debug_hypot_selftest(sid_pm_ctx *ctx, uint64_t seed, int64_t evaluations, uint64_t counts[2]);

#ifdef __cplusplus
}
#endif
#endif /* SID_PM_H */
