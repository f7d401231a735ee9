/* sid_stage.h - C ABI of the uint8 staging step on MI355X (gfx950): SURVEY.md section 8, row f3.
 *
 * Replaces, in the reference (sea_ice_drift v0.7.1), lib.py:27-59 get_uint8_image():
 *
 *     vmin = np.nanpercentile(image, pmin)          (lib.py:46-48, when vmin is None)
 *     vmax = np.nanpercentile(image, pmax)          (lib.py:49-51, when vmax is None)
 *     uint8Image = 1 + 254 * (image - vmin) / (vmax - vmin)        (lib.py:54)
 *     clip to [1, 255]; pixels whose input is not finite -> 0; astype('uint8')   (lib.py:55-59)
 *
 * The device does the two passes over the 10^8-pixel float32 image that cost seconds on the host:
 *   sid_stage_order_stats  - exact order statistics of the non-NaN pixels (what nanpercentile interpolates
 *                            between; the float32 interpolation itself is three scalar operations and stays
 *                            with the caller, sea_ice_drift_amd/lib.py, in NumPy's own arithmetic);
 *   sid_stage_scale_u8     - the element-wise map, float32 operation for operation as NumPy evaluates it.
 * All pointers are device pointers unless stated; `hip_stream` is a hipStream_t (may be NULL).
 * 0 on success, a negative SID_PM_ERR_* code otherwise (sid_pm.h); sid_stage_last_error() has the message.
 */
#ifndef SID_STAGE_H
#define SID_STAGE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Workspace handle: device histograms + a pinned host copy, created once and reused (no allocation per call). */
typedef struct sid_stage_ws sid_stage_ws;
int  sid_stage_create(int device, sid_stage_ws **ws);
void sid_stage_destroy(sid_stage_ws *ws);

/* First pass over the image: histogram of the leading 11 key bits of every non-NaN pixel -> *n_valid (host) = their
 * number (what np.nanpercentile needs to turn a percentile into ranks).  The histogram stays in the workspace. */
int sid_stage_begin(sid_stage_ws *ws, const float *d_img, int64_t rows, int64_t cols, int64_t stride, int64_t *n_valid,
                    void *hip_stream);

/* The same with a hint (round 4): the order statistics that will be asked for lie near these fractions (numpy percentile p ->
 * p / 100; at most 4) of the non-NaN pixels.  A sample of 8192 pixels brackets every fraction by a key range, and ONE pass
 * over the image counts the non-NaN pixels, the pixels below each range and a 2048-bin histogram inside it; a rank that falls
 * into a range then costs one more pass (none when a bin is a single key) instead of two.  The results are exact either way:
 * a rank outside every range takes the three-pass route of sid_stage_begin (its first pass is then run on demand).
 * SID_STAGE_NO_HINT=1 in the environment makes this entry point sid_stage_begin (A/B runs).                              */
int sid_stage_begin_hint(sid_stage_ws *ws, const float *d_img, int64_t rows, int64_t cols, int64_t stride,
                         const double *fractions, int n_fractions, int64_t *n_valid, void *hip_stream);

/* Exact order statistics of the image of the last sid_stage_begin / sid_stage_begin_hint: values[k] (host) = the ranks[k]-th smallest
 * (0-based) non-NaN pixel; two more passes over the image per group of up to eight ranks (one after a hint that holds). */
int sid_stage_order_stats_ws(sid_stage_ws *ws, const int64_t *ranks, int n_ranks, float *values);

/* One-shot forms of the above (a temporary workspace per call). */
/* Number of non-NaN pixels of a float32 image [rows][cols] with row stride `stride` (in elements) -> *n_valid (host). */
int sid_stage_count_valid(const float *d_img, int64_t rows, int64_t cols, int64_t stride, int64_t *n_valid,
                          void *hip_stream);

/* values[k] (host) = the ranks[k]-th smallest (0-based) non-NaN pixel, ranks ascending not required,
 * 0 <= ranks[k] < n_valid; +-inf take part in the order like any value (as in np.sort). */
int sid_stage_order_stats(const float *d_img, int64_t rows, int64_t cols, int64_t stride,
                          const int64_t *ranks, int n_ranks, float *values, void *hip_stream);

/* out[r][c] = uint8(clip(1 + (254 * (img[r][c] - vmin)) / denom, 1, 255)), 0 where img is not finite (or the
 * result is NaN); every operation rounded to float32 (lib.py:54 with float32 operands). */
int sid_stage_scale_u8(const float *d_img, int64_t rows, int64_t cols, int64_t stride, float vmin, float denom,
                       uint8_t *d_out, int64_t out_stride, void *hip_stream);

const char *sid_stage_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
