// Host interface of the large-window pipeline (pm_large.hip), shared with the C ABI (pm_capi.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace sid {

// Template sides the large-window kernels take: sum w' t' of re-centred bytes stays inside int32 and sum w'^2 inside uint32.
constexpr int kLargeMaxSide = 255;

// Device scratch of the pipeline, owned by a handle (grow-only; freed by lw_workspace_release).
struct LwWorkspace {
    void *buf[16] = {};
    size_t cap[16] = {};
};
void lw_workspace_release(LwWorkspace &W);
// bytes of device scratch one call with these dimensions needs (for the out-of-memory message)
size_t lw_scratch_bytes(int wh, int ww, int s, int K, uint32_t flags);

// One rotate_and_match (reference pmlib.py:117-174): K rotated templates of side s around (c1, r1) on image 1 against the
// window [win_r0, win_r0 + wh) x [win_c0, win_c0 + ww) of image 2.  Everything is enqueued on `stream`; nothing is read back.
// Results: out5 = add_c + dc, add_r + dr, angle, r, h (device-visible, 5 doubles) and ij3 = peak row, peak column, angle
// index (device-visible, 3 int32; may be null); NaN x 5 / -1 when a template touches a zero pixel (pmlib.py:152-154).
struct LargeCall {
    const uint8_t *img1; int64_t rows1, cols1, stride1;
    const double *d_coef = nullptr;                        // rot_order 2..5 (flags bits 3..5): image 1 through lw_spline_prefilter, [rows1][cols1]
    const uint8_t *img2; int64_t stride2;                  // image 2 (the window must lie inside it: the caller checks)
    int64_t win_r0, win_c0; int wh, ww;
    double c1, r1;
    int s, K; uint32_t flags;
    const double *d_rot;                                   // device [K][4]: cos, sin, tcT0, tcT1
    const double *d_angles;                                // device [K]
    double add_c, add_r;                                   // c2fg, r2fg of use_mcc (pmlib.py:209-210); 0 for rotate_and_match itself
    double gauss_w[5];                                     // hes_smth taps (pm_capi.hip gauss_taps)
    double *out5; int32_t *ij3;
};
// returns a hipError_t as int (0 = success); -1: scratch allocation failed
int lw_run(const LargeCall &c, LwWorkspace &W, void *stream);
// n calls of one run (same template side, angles and flags; any windows) in batches of up to 64 points: one launch per phase
// and batch, the point a launch dimension.  Waits for the stream once at entry (the scratch of an earlier run is reused).
int lw_run_batch(const LargeCall *calls, int n, LwWorkspace &W, void *stream);
// After lw_run on the same workspace has completed: device pointers of the NCC matrix [rh][rw] (float32) and the template
// [s][s] (uint8) of angle k (the matrices of ALL candidate angles stay in the workspace until the next lw_run).
const float *lw_ncc_matrix(const LwWorkspace &W, int wh, int ww, int s, int k);
const uint8_t *lw_template(const LwWorkspace &W, int s, int k);
// NaN x 5 / -1 rows for the listed points (points without a valid window when no other kernel of the run writes them)
int lw_write_nan(const int32_t *d_idx, int n, double *out, int32_t *out_ij, void *stream);

// get_template (pmlib.py:89-115) as a call of its own: one template of side s sampled from `d_img` - the rows [row0, row0 + nrows)
// and columns [col0, col0 + ncols) of an image of rows x cols pixels (only that part needs to be on the device) - to d_out [s][s].
// order 2..5: d_coef = the WHOLE image through lw_spline_prefilter (then d_img is not read).
int lw_get_template(const uint8_t *d_img, int64_t stride, int64_t row0, int64_t col0, int64_t nrows, int64_t ncols, int64_t rows, int64_t cols,
                    double c, double r, const double *d_rot4, int s, int order, uint8_t *d_out, void *stream, const double *d_coef = nullptr);
// scipy.ndimage.spline_filter(img, order, output=float64, mode='constant') of a device uint8 image (pmlib.py:112-113 with rot_order =
// 2..5: affine_transform prefilters the WHOLE image): result in buf1 [rows][cols]; buf0 = scratch of the same size.
int lw_spline_prefilter(const uint8_t *d_img, int64_t rows, int64_t cols, int64_t stride, int order, double *buf0, double *buf1, void *stream);
// the K templates of n points (centres d_c1 / d_r1) sampled from the prefiltered image into d_pre [n][K][s][s]
int lw_presample(const double *d_coef, int64_t rows, int64_t cols, const double *d_c1, const double *d_r1, int64_t n, const double *d_rot, int K, int s,
                 int order, uint8_t *d_pre, void *stream);
// get_hessian (pmlib.py:36-59) of a float32 matrix on the device: d_hes [rh][rw] = the (normalised) Hessian magnitudes.
int lw_get_hessian(const float *d_ccm, int rh, int rw, uint32_t flags, const double gauss_w[5], float *d_hes, LwWorkspace &W, void *stream);

}  // namespace sid
