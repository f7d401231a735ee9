/* sid_fg.h - C ABI of the first-guess evaluation on MI355X (gfx950): SURVEY.md section 8, row f4.
 *
 * Replaces, in the reference's prelude of pattern matching (sea_ice_drift v0.7.1):
 *
 *   sid_fg_interp_linear   the evaluation half of lib.interpolation_near (lib.py:179-201):
 *                              griddata(src, x2, dst, method='linear'), griddata(src, y2, dst, method='linear')
 *                          i.e. point location in the Delaunay triangulation of the key points + barycentric
 *                          interpolation of both components at the N grid points, NaN outside the convex hull.
 *                          The triangulation itself stays SciPy's (Qhull) on the host - the simplices are an input -
 *                          so the triangles, and with them the interpolant, are the reference's.
 *   sid_fg_nearest_dist    pmlib.get_distance_to_nearest_keypoint (pmlib.py:61-77) sampled at the grid points
 *                          (pmlib.py:300-305): distance from each (row, col) to the nearest key-point pixel - the
 *                          exact Euclidean distance the full-image EDT holds there (sqrt of an exact integer).
 *
 * Arithmetic of the interpolation: per simplex the 2x2 system of the barycentric transform is solved by LU with
 * partial pivoting, c = Tinv (x - r), c2 = 1 - c0 - c1, value = c0 v0 + c1 v1 + c2 v2 accumulated in vertex order
 * (scipy.spatial.Delaunay.transform / LinearNDInterpolator); a point belongs to the first simplex (lowest index)
 * with all c >= -eps, eps = 100 * DBL_EPSILON.  SciPy reaches a simplex by a directed walk, so for a query ON an edge,
 * a vertex or the hull it may pick another simplex (or none); and its BLAS may round the 2x2 solve differently in the
 * last bit.  Neither may change what the reference computes next (np.round of the first guess, pmlib.py:285-288), so
 * every query for which that cannot be excluded is FLAGGED in doubt[] and the caller evaluates those with SciPy itself
 * (sea_ice_drift_amd/lib.py interpolation_near).  A query whose smallest barycentric coordinate in some simplex lies
 * within 1e-9 of zero (on an edge, a vertex, the hull) is looked at against every simplex: it stays flagged unless hull
 * membership is clear (no smallest coordinate within 4e-15 of SciPy's threshold -2.2e-14), all containing simplices round to the same
 * integers and no value lies within 1e-6 of a half-integer; every other query lies strictly inside exactly one simplex
 * (flagged only when its value is that close to a half-integer).
 * Limitation: degenerate (flat) simplices are skipped here, whereas SciPy gives them a NaN transform and then accepts
 * queries in their neighbours with a wider tolerance (sqrt(eps)) towards them - a case the flags do not model.  The
 * caller must not use this entry point for a triangulation that holds such a simplex: sea_ice_drift_amd/lib.py checks
 * (condition number of every simplex) and evaluates with SciPy alone then.
 * Host buffers in / host buffers out; 0 on success, a negative SID_PM_ERR_* code otherwise (sid_pm.h).
 */
#ifndef SID_FG_H
#define SID_FG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pts [n_pts][2], simplices [n_simp][3] (indices into pts), values [n_pts][2], q [n_q][2] -> out [n_q][2];
 * optional (may be NULL): simplex [n_q] = index of the simplex a query was located in (-1: none), doubt [n_q] (above) */
int sid_fg_interp_linear(int device, const double *pts, int64_t n_pts, const int32_t *simplices, int64_t n_simp,
                         const double *values, const double *q, int64_t n_q, double *out, int32_t *simplex, int32_t *doubt);

/* seeds [n_seeds][2], q [n_q][2] -> dist [n_q] = min over seeds of the Euclidean distance */
int sid_fg_nearest_dist(int device, const double *seeds, int64_t n_seeds, const double *q, int64_t n_q, double *dist);

/* get_distance_to_nearest_keypoint (pmlib.py:61-77) at full resolution: dist [rows][cols] = distance of every pixel (row, column)
 * to the nearest of seeds [n_seeds][2] = (row, column) - the reference's distance_transform_edt image, evaluated per pixel through
 * the same buckets (sqrt of the exact integer squared distance in float64: the same numbers). */
int sid_fg_distance_image(int device, const double *seeds, int64_t n_seeds, int64_t rows, int64_t cols, double *dist);

const char *sid_fg_last_error(void);
/* Free the grow-only device scratch block of the two entry points on `device` (every device: -1).  The blocks are kept between calls so that no call pays for
 * hipMalloc / hipFree; a long-lived process that is done with the GPU hands the memory back with this (the Python mirror's
 * pmlib.release_contexts() calls it).  Not to be called while a call on that device is in flight. */
int sid_fg_release(int device);

#ifdef __cplusplus
}
#endif
#endif
