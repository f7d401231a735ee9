/* sid_orb.h - C ABI of the key-point detector / descriptor on MI355X (gfx950): SURVEY.md section 8, row f1.
 *
 * Serves the interface of the reference's ftlib.find_key_points (sea_ice_drift v0.7.1, ftlib.py:26-61):
 *
 *     detector = cv2.ORB_create(); setEdgeThreshold(34); setMaxFeatures(100000); setNLevels(7); setPatchSize(34)
 *     keyPoints, descriptors = detector.detectAndCompute(image, None)          (ftlib.py:46-58)
 *
 * i.e. uint8 image in -> key points (x, y) on the full-resolution grid + 256-bit binary descriptors out, which
 * feed the Hamming matcher (sid_ft.h) unchanged.  OpenCV is not a dependency of this library and is not
 * installed where it is built, so OpenCV's own key points are NOT reproduced (parity unpinned, SURVEY 8 f1:
 * "own detector, same interface"); what is implemented is an ORB-family detector with an all-integer
 * specification, restated in NumPy as oracle/orb_oracle.py and matched by the kernels bit for bit:
 *
 *   pyramid      n_levels levels, level l = level 0 resampled by 1 / scale^l (scale 1.2): 16.16 fixed-point source
 *                coordinates, 8-bit bilinear weights, round half up
 *   corners      FAST-9 on the 16-pixel Bresenham ring of radius 3 (threshold fast_threshold), score = the largest
 *                threshold that still passes; 3x3 non-maximum suppression (strictly greater than all 8 neighbours)
 *   ranking      Harris response on a 7x7 block with central differences, 25 (ab - c^2) - (a + b)^2 in int64
 *                (k = 0.04); the best n_l per level, n_l from OpenCV's geometric split of n_features
 *   orientation  intensity centroid (m10, m01) over the disc of radius patch_size / 2, quantised to 32 directions by
 *                integer dot products with a host-built direction table
 *   descriptor   256 intensity comparisons on a 5x5-binomial-smoothed level image; the pair pattern (seeded, within
 *                radius 13) is pre-rotated per direction on the host
 *
 * Host buffers in / host buffers out; 0 on success, a negative SID_PM_ERR_* code otherwise (sid_pm.h).
 */
#ifndef SID_ORB_H
#define SID_ORB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sid_orb_params {
    int32_t edge_threshold;   /* no key point closer than this to the border of its level (reference: 34)   */
    int32_t n_features;       /* upper bound on the number of key points (reference: 100000)               */
    int32_t n_levels;         /* pyramid levels (reference: 7), 1..16                                      */
    int32_t patch_size;       /* orientation disc diameter (reference: 34)                                 */
    int32_t fast_threshold;   /* FAST threshold (OpenCV default 20)                                        */
    float   scale_factor;     /* pyramid scale (OpenCV default 1.2)                                        */
} sid_orb_params;

/* pattern:   [32][256][4] int8 = (ax, ay, bx, by) of every comparison for every direction (host-built, see
 *            sea_ice_drift_amd/orb.py; the same table drives the oracle)
 * dirs:      [32][2] int32 = round(2^14 cos, 2^14 sin) of the 32 directions
 * xy:        [max_out][2] float32 out: key point (x, y) in full-resolution pixels (level coordinate times the scale)
 * meta:      [max_out][4] int32 out: level x, level y, level, direction index (may be NULL)
 * response:  [max_out] int64 out: Harris response (may be NULL)
 * desc:      [max_out][32] uint8 out
 * Key points are ordered by level, then by descending response, then by (y, x).  *n_out <= max_out.
 * Thread-safe: every call takes a free workspace of its device (device buffers ~7 bytes per pixel, a private stream; kept
 * for the life of the process) or creates one, so calls from several host threads run side by side. */
int sid_orb_detect(int device, const uint8_t *img, int64_t rows, int64_t cols, int64_t stride,
                   const sid_orb_params *params, const int8_t *pattern, const int32_t *dirs,
                   float *xy, int32_t *meta, int64_t *response, uint8_t *desc, int64_t max_out, int64_t *n_out);

const char *sid_orb_last_error(void);
/* Free the idle detector workspaces (device buffers of ~7 bytes per pixel, a stream and pinned staging each; at most two idle ones are kept per device) on `device` (every device: -1).  The blocks are kept between calls so that no call pays for
 * hipMalloc / hipFree; a long-lived process that is done with the GPU hands the memory back with this (the Python mirror's
 * pmlib.release_contexts() calls it).  Not to be called while a call on that device is in flight. */
int sid_orb_release(int device);

#ifdef __cplusplus
}
#endif
#endif
