/* sid_ft.h - C ABI of the feature-tracking matcher on MI355X (gfx950): SURVEY.md section 8, row f1.
 *
 * Replaces, in the reference (sea_ice_drift v0.7.1):
 *
 *   sid_ft_knn2 / sid_ft_knn2_device
 *       <- ftlib.py:92-99  _get_matches():  bf = cv2.BFMatcher(cv2.NORM_HAMMING);
 *                                            matches = bf.knnMatch(descriptors1, descriptors2, k=2)
 *          (called from ftlib.py:87 get_match_coords and, through it, ftlib.py:267 feature_tracking)
 *
 * Descriptors are ORB's 32-byte (256-bit) strings, row-major uint8 [n][32] (ftlib.py:56-58).  For every query
 * descriptor the two train descriptors with the smallest Hamming distance are returned, nearest first.
 * Equal distances are ordered by the smaller train index (a brute-force scan in index order; OpenCV's own
 * tie order is not pinned by the reference - cv2 is not available to this build).
 * The Lowe ratio filter that follows (ftlib.py:101-116) is host code in sea_ice_drift_amd/ftlib.py.
 *
 * Plain pointers and sizes; 0 on success, a negative SID_PM_ERR_* code otherwise (sid_pm.h);
 * sid_ft_last_error() returns the message of the calling thread's last failure.
 */
#ifndef SID_FT_H
#define SID_FT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SID_FT_DESC_BYTES 32

/* Host buffers in, host buffers out (upload, match, download on the given device).
 *   desc1 [n1][32], desc2 [n2][32] uint8; idx [n1][2] int32 = train indices (nearest, second nearest),
 *   dist [n1][2] int32 = their Hamming distances; entries that do not exist (n2 < 2) are -1.            */
int sid_ft_knn2(int device, const uint8_t *desc1, int64_t n1, const uint8_t *desc2, int64_t n2,
                int32_t *idx, int32_t *dist);

/* Same on device-resident buffers (16-byte aligned descriptors), asynchronous on `hip_stream`
 * (a hipStream_t, may be NULL); `workspace` (16-byte aligned) must hold sid_ft_workspace_bytes(n1, n2) bytes: the per-chunk
 * candidates and, from 2^24 pairs on, the byte-per-bit copies of both descriptor sets for the int8-MFMA form (256 bytes per
 * descriptor, padded to multiples of 256 descriptors).                                                                  */
int sid_ft_knn2_device(const uint8_t *d_desc1, int64_t n1, const uint8_t *d_desc2, int64_t n2,
                       int32_t *d_idx, int32_t *d_dist, void *d_workspace, void *hip_stream);
int64_t sid_ft_workspace_bytes(int64_t n1, int64_t n2);

const char *sid_ft_last_error(void);
/* Free the grow-only device scratch block of sid_ft_knn2 on `device` (every device: -1).  The blocks are kept between calls so that no call pays for
 * hipMalloc / hipFree; a long-lived process that is done with the GPU hands the memory back with this (the Python mirror's
 * pmlib.release_contexts() calls it).  Not to be called while a call on that device is in flight. */
int sid_ft_release(int device);

#ifdef __cplusplus
}
#endif
#endif
