// first_guess.hip - evaluation of the first guess on gfx950 (C ABI: include/sid_fg.h; the reference's
// lib.interpolation_near, lib.py:179-201, and pmlib.get_distance_to_nearest_keypoint, pmlib.py:61-77).
// Point location walks a uniform grid of buckets (round 4): every simplex is listed in the cells its bounding box touches, a
// query tests the simplices of its own cell only - a dozen instead of all 6 x 10^4 (rounds 2-3: brute force over (query,
// simplex) pairs, 1.7 ms per call; kept as the fallback for triangulations whose bucket lists would not fit).  The
// nearest-key-point distance is brute force over (query, seed) pairs with the seeds staged through LDS.
#include <hip/hip_runtime.h>
#include <math.h>
#include <float.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>

#include "../../include/sid_fg.h"
#include "../../include/sid_pm.h"

#define SID_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local char g_err[256] = "";
int fail(int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}

struct Simp { double t00, t01, t10, t11, rx, ry; int v0, v1, v2, ok; };

// barycentric transform of every simplex: LU with partial pivoting of M = [[x0 - xr, y0 - yr], [x1 - xr, y1 - yr]],
// Tinv = M^-T (scipy.spatial.Delaunay.transform); degenerate simplices are marked and never match
__global__ void k_transform(const double *pts, const int32_t *simp, int64_t ns, Simp *out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    const int a = simp[3 * k], b = simp[3 * k + 1], c = simp[3 * k + 2];
    const double rx = pts[2 * c], ry = pts[2 * c + 1];
    double m00 = pts[2 * a] - rx, m01 = pts[2 * a + 1] - ry, m10 = pts[2 * b] - rx, m11 = pts[2 * b + 1] - ry;
    const bool swap = fabs(m10) > fabs(m00);
    if (swap) { double t = m00; m00 = m10; m10 = t; t = m01; m01 = m11; m11 = t; }
    Simp s; s.rx = rx; s.ry = ry; s.v0 = a; s.v1 = b; s.v2 = c;
    const double l = m10 * (1.0 / m00), u11 = m11 - l * m01;
    s.ok = (m00 != 0.0 && u11 != 0.0 && isfinite(l) && isfinite(1.0 / u11)) ? 1 : 0;
    const double iu11 = 1.0 / u11, iu00 = 1.0 / m00;
    double x[2][2];
    for (int col = 0; col < 2; ++col) {
        double b0 = col == 0 ? 1.0 : 0.0, b1 = col == 1 ? 1.0 : 0.0;
        if (swap) { const double t = b0; b0 = b1; b1 = t; }
        const double y1 = fma(-l, b0, b1);
        const double x1 = y1 * iu11;
        const double x0 = fma(-m01, x1, b0) * iu00;
        x[0][col] = x0; x[1][col] = x1;
    }
    // M^-1 = x; Tinv (rows = barycentric coordinates, columns = x, y) = (M^-1)^T
    s.t00 = x[0][0]; s.t01 = x[1][0]; s.t10 = x[0][1]; s.t11 = x[1][1];
    out[k] = s;
}

// Point location: block (x, y) tests 256 queries against the y-th chunk of the simplices (staged through LDS in tiles);
// the lowest index of a containing simplex wins (atomicMin), which is what one pass over all of them in index order
// would have found.  Chunking the simplices puts ~8x more wavefronts on the device than one thread per query alone
// (40 000 queries are only 625 wavefronts).
constexpr int kTile = 256;
// kTol: a query whose smallest barycentric coordinate in some simplex lies in [-kTol, kTol) sits on (or within rounding
// of) an edge, a vertex or the hull: which simplex - or whether any - SciPy's directed walk assigns to it cannot be told
// from here.  Such queries are FLAGGED (near[]: unlocated ones, here; located ones in k_eval) and the caller evaluates
// them with SciPy itself; every other query lies strictly inside exactly one simplex.
constexpr double kTol = 1e-9;
__global__ __launch_bounds__(256) void k_locate(const Simp *simp, int64_t ns, int64_t chunk, const double *q, int64_t nq, int32_t *loc, int32_t *near)
{
    __shared__ Simp tile[kTile];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double x = i < nq ? q[2 * i] : 0.0, y = i < nq ? q[2 * i + 1] : 0.0;
    const double eps = 100.0 * DBL_EPSILON;
    const int64_t k0 = (int64_t)blockIdx.y * chunk, k1 = k0 + chunk < ns ? k0 + chunk : ns;
    bool found = false, close = false;
    for (int64_t base = k0; base < k1; base += kTile) {
        __syncthreads();
        if (base + (int64_t)threadIdx.x < k1) tile[threadIdx.x] = simp[base + threadIdx.x];
        __syncthreads();
        const int n = (int)((k1 - base) < kTile ? (k1 - base) : kTile);
        if (!found && i < nq) {
            for (int k = 0; k < n; ++k) {
                const Simp &s = tile[k];
                const double dx = x - s.rx, dy = y - s.ry;
                const double c0 = s.t00 * dx + s.t01 * dy, c1 = s.t10 * dx + s.t11 * dy;
                const double c2 = 1.0 - c0 - c1;
                const double mc = fmin(c0, fmin(c1, c2));
                if (s.ok && mc >= -eps) { atomicMin(&loc[i], (int32_t)(base + k)); found = true; break; }
                close = close || (s.ok && mc >= -kTol);
            }
        }
    }
    if (i < nq && !found && close) atomicOr(&near[i], 1);               // (ignored by k_eval when another chunk located the query)
}

// doubt[i] = 1: the caller must not trust out[i] / simplex[i] (see kTol); also set when a value comes within 1e-6 of a
// half-integer (the reference rounds the first guess next, pmlib.py:285-288: a last-bit difference could tip it)
// The flagged queries are also appended to list[] (count: *nlist; any order) for k_resolve.
__global__ void k_eval(const Simp *simp, const int32_t *loc, const int32_t *near, const double *values, const double *q, int64_t nq,
                       double *out, int32_t *simplex, int32_t *doubt, int32_t *list, int32_t *nlist)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    double o0 = NAN, o1 = NAN;
    const int32_t k = loc[i];
    int32_t dbt = (k == 0x7f7f7f7f) ? near[i] : 0;
    if (k != 0x7f7f7f7f) {
        const Simp s = simp[k];
        const double dx = q[2 * i] - s.rx, dy = q[2 * i + 1] - s.ry;
        const double c0 = s.t00 * dx + s.t01 * dy, c1 = s.t10 * dx + s.t11 * dy;
        const double c2 = 1.0 - c0 - c1;
        const double *v0 = values + 2 * (int64_t)s.v0, *v1 = values + 2 * (int64_t)s.v1, *v2 = values + 2 * (int64_t)s.v2;
        o0 = c0 * v0[0]; o0 += c1 * v1[0]; o0 += c2 * v2[0];
        o1 = c0 * v0[1]; o1 += c1 * v1[1]; o1 += c2 * v2[1];
        const double h0 = fabs(o0 - floor(o0) - 0.5), h1 = fabs(o1 - floor(o1) - 0.5);
        dbt = (fmin(c0, fmin(c1, c2)) < kTol || h0 < 1e-6 || h1 < 1e-6 || !(fabs(o0) < 1e15) || !(fabs(o1) < 1e15)) ? 1 : 0;
    }
    out[2 * i] = o0; out[2 * i + 1] = o1;
    simplex[i] = k == 0x7f7f7f7f ? -1 : k;
    doubt[i] = dbt;
    if (dbt) list[atomicAdd(nlist, 1)] = (int32_t)i;
}

// ---- uniform grid of buckets over the bounding box of the key points ----
// A simplex is listed in every cell its bounding box - widened by far more than the location tolerance kTol can reach -
// touches; cell indices come from ONE monotone function of the coordinate (cell_of), for simplices and queries alike, so a
// query that lies within kTol of a simplex finds it in the list of its own cell.  Queries outside the box are clamped to the
// border cells (they can only be outside the hull, or within kTol of it).
constexpr int kGrid = 128;
struct GridGeo { double x0, y0, ix, iy; };
__device__ __forceinline__ int cell_of(double v, double v0, double inv)
{
    const double c = floor((v - v0) * inv);
    return c < 0.0 ? 0 : (c > (double)(kGrid - 1) ? kGrid - 1 : (int)c);
}
__device__ __forceinline__ void simplex_cells(const double *pts, const int32_t *simp, int64_t k, const GridGeo g, int &cx0, int &cx1, int &cy0, int &cy1)
{
    const int a = simp[3 * k], b = simp[3 * k + 1], c = simp[3 * k + 2];
    const double xa = pts[2 * a], ya = pts[2 * a + 1], xb = pts[2 * b], yb = pts[2 * b + 1], xc = pts[2 * c], yc = pts[2 * c + 1];
    const double x_lo = fmin(xa, fmin(xb, xc)), x_hi = fmax(xa, fmax(xb, xc)), y_lo = fmin(ya, fmin(yb, yc)), y_hi = fmax(ya, fmax(yb, yc));
    // kTol = 1e-9 of a barycentric coordinate moves a point by at most 1e-9 of the simplex's extent; 1e-6 of it (+ an
    // absolute part for the rounding of large coordinates) is far on the safe side
    const double m = 1e-6 * ((x_hi - x_lo) + (y_hi - y_lo)) + 1e-9 * (fabs(x_hi) + fabs(x_lo) + fabs(y_hi) + fabs(y_lo)) + 1e-12;
    cx0 = cell_of(x_lo - m, g.x0, g.ix); cx1 = cell_of(x_hi + m, g.x0, g.ix);
    cy0 = cell_of(y_lo - m, g.y0, g.iy); cy1 = cell_of(y_hi + m, g.y0, g.iy);
}
// pass 0: entries per cell; pass 1: the entries themselves (start[] = exclusive prefix sums of the counts, cursor[] zeroed)
template <int PASS>
__global__ void k_grid_build(const double *pts, const int32_t *simp, const Simp *tr, int64_t ns, GridGeo g, int32_t *count, const int32_t *start,
                             int32_t *ids, int64_t cap)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns || !tr[k].ok) return;                                  // (degenerate simplices never match)
    int cx0, cx1, cy0, cy1;
    simplex_cells(pts, simp, k, g, cx0, cx1, cy0, cy1);
    for (int cy = cy0; cy <= cy1; ++cy)
        for (int cx = cx0; cx <= cx1; ++cx) {
            const int cell = cy * kGrid + cx;
            const int32_t pos = atomicAdd(&count[cell], 1);
            if (PASS == 1) { const int64_t at = (int64_t)start[cell] + pos; if (at < cap) ids[at] = (int32_t)k; }
        }
}
// exclusive prefix sums of the kGrid^2 counts (one workgroup; start[kGrid^2] = total), counts zeroed for the second pass
__global__ __launch_bounds__(1024) void k_grid_scan(int32_t *count, int32_t *start)
{
    __shared__ int32_t part[1024];
    constexpr int per = kGrid * kGrid / 1024;
    const int t = threadIdx.x;
    int32_t loc[per], sum = 0;
    for (int j = 0; j < per; ++j) { loc[j] = sum; sum += count[t * per + j]; count[t * per + j] = 0; }
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const int32_t base = part[t] - sum;
    for (int j = 0; j < per; ++j) start[t * per + j] = base + loc[j];
    if (t == 1023) start[kGrid * kGrid] = part[t];
}
// point location through the buckets: the lowest index of a containing simplex, as the brute-force pass finds it
__global__ __launch_bounds__(256) void k_locate_grid(const Simp *simp, const int32_t *start, const int32_t *ids, GridGeo g,
                                                      const double *q, int64_t nq, int32_t *loc, int32_t *near)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const double x = q[2 * i], y = q[2 * i + 1];
    const double eps = 100.0 * DBL_EPSILON;
    const int cell = cell_of(y, g.y0, g.iy) * kGrid + cell_of(x, g.x0, g.ix);
    int32_t best = 0x7f7f7f7f;
    bool close = false;
    for (int32_t e = start[cell]; e < start[cell + 1]; ++e) {
        const int32_t k = ids[e];
        const Simp s = simp[k];
        const double dx = x - s.rx, dy = y - s.ry;
        const double c0 = s.t00 * dx + s.t01 * dy, c1 = s.t10 * dx + s.t11 * dy;
        const double c2 = 1.0 - c0 - c1;
        const double mc = fmin(c0, fmin(c1, c2));
        if (mc >= -eps) best = k < best ? k : best;
        close = close || mc >= -kTol;
    }
    loc[i] = best;
    near[i] = (best == 0x7f7f7f7f && close) ? 1 : 0;
}

// Second look at the flagged queries (k_eval: on an edge, a vertex or the hull), against EVERY simplex, no early exit.  A
// flagged query is RESOLVED - the caller need not ask SciPy - when whatever simplex SciPy's walk ends in, the ROUNDED result
// is the same: (a) hull membership is clear - the best smallest barycentric coordinate over all simplices is not within
// 4e-15 of SciPy's threshold -eps = -2.2e-14; (b) every simplex that contains the query (min c >= -eps) rounds to the same pair of
// integers (a shared vertex or edge interpolates to the same value from either side, up to the last bits); (c) no such
// value lies within 1e-6 of a half-integer.  Resolved queries get the values of the lowest-index containing simplex (or
// NaN); the others keep their flag.
// start / ids (may be null): the bucket lists of k_grid_build - the flagged query then looks at the simplices of its cell only
__global__ __launch_bounds__(256) void k_resolve(const Simp *simp, int64_t ns, const double *values, const double *q,
                                                 const int32_t *list, const int32_t *nlist, double *out, int32_t *simplex, int32_t *doubt,
                                                 const int32_t *start, const int32_t *ids, GridGeo g)
{
    // one WAVEFRONT per flagged query: its 64 lanes share the simplices (lane l takes l, l + 64, ...: coalesced 64-byte
    // records) and merge what they found - one thread per query walking all 6 x 10^4 simplices took 19 of the 21 ms of the call
    // with the integer-coordinate key points of a detector
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int32_t nf = *nlist;
    const double eps = 100.0 * DBL_EPSILON;
    for (int64_t f = wave; f < nf; f += nwaves) {
        const int64_t i = list[f];
        const double x = q[2 * i], y = q[2 * i + 1];
        bool hull_unclear = false, disagree = false, near_half = false, any = false;
        double R0 = 0.0, R1 = 0.0, V0 = NAN, V1 = NAN;
        int32_t K = 0x7fffffff;                                         // this lane's lowest containing simplex
        int64_t e0 = 0, e1 = ns;
        if (ids) { const int cell = cell_of(y, g.y0, g.iy) * kGrid + cell_of(x, g.x0, g.ix); e0 = start[cell]; e1 = start[cell + 1]; }
        for (int64_t e = e0 + lane; e < e1; e += 64) {
            const int64_t k = ids ? (int64_t)ids[e] : e;
            const Simp s = simp[k];
            if (!s.ok) continue;
            const double dx = x - s.rx, dy = y - s.ry;
            const double c0 = s.t00 * dx + s.t01 * dy, c1 = s.t10 * dx + s.t11 * dy;
            const double c2 = 1.0 - c0 - c1;
            const double mc = fmin(c0, fmin(c1, c2));
            if (!(mc >= -kTol)) continue;
            if (fabs(mc + eps) < 4e-15) hull_unclear = true;           // (eps itself is 2.2e-14; a coordinate carries ~1e-16 of rounding)
            if (mc >= -eps) {
                const double *v0 = values + 2 * (int64_t)s.v0, *v1 = values + 2 * (int64_t)s.v1, *v2 = values + 2 * (int64_t)s.v2;
                double o0 = c0 * v0[0]; o0 += c1 * v1[0]; o0 += c2 * v2[0];
                double o1 = c0 * v0[1]; o1 += c1 * v1[1]; o1 += c2 * v2[1];
                const double r0 = rint(o0), r1 = rint(o1);             // half to even, as np.round
                near_half = near_half || fabs(o0 - floor(o0) - 0.5) < 1e-6 || fabs(o1 - floor(o1) - 0.5) < 1e-6 || !(fabs(o0) < 1e15) || !(fabs(o1) < 1e15);
                if (!any) { any = true; R0 = r0; R1 = r1; V0 = o0; V1 = o1; K = (int32_t)k; }
                else {
                    if (r0 != R0 || r1 != R1) disagree = true;
                    if ((int32_t)k < K) { K = (int32_t)k; V0 = o0; V1 = o1; }   // (bucket lists are not in index order)
                }
            }
        }
        // merge: the lowest containing simplex of all lanes gives the values; every lane's own first one (with which the
        // rest of its share agrees, or `disagree` is set) must round like it
        int32_t kmin = K;
        for (int d = 32; d >= 1; d >>= 1) { const int32_t o = __shfl_xor(kmin, d); kmin = o < kmin ? o : kmin; }
        const unsigned long long owner = __ballot(any && K == kmin);
        const int src = owner ? __ffsll((long long)owner) - 1 : 0;
        const double G0 = __shfl(R0, src), G1 = __shfl(R1, src), W0 = __shfl(V0, src), W1 = __shfl(V1, src);
        if (any && (R0 != G0 || R1 != G1)) disagree = true;
        if (__any(hull_unclear || disagree || near_half)) continue;    // stays flagged
        if (lane == 0) {
            out[2 * i] = owner ? W0 : NAN; out[2 * i + 1] = owner ? W1 : NAN;
            simplex[i] = owner ? kmin : -1;
            doubt[i] = 0;
        }
    }
}

// q == nullptr: the queries are the pixels of an image with `qcols` columns, query i = (row i / qcols, column i % qcols)
// (sid_fg_distance_image: the reference's full-resolution distance image, pmlib.py:61-77)
__global__ __launch_bounds__(256) void k_nearest(const double *seeds, int64_t ns, const double *q, int64_t nq, double *dist, int64_t qcols)
{
    __shared__ double sx[1024], sy[1024];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double x = i >= nq ? 0.0 : q ? q[2 * i] : (double)(i / qcols), y = i >= nq ? 0.0 : q ? q[2 * i + 1] : (double)(i % qcols);
    double best = INFINITY;
    for (int64_t base = 0; base < ns; base += 1024) {
        __syncthreads();
        for (int k = threadIdx.x; k < 1024 && base + k < ns; k += 256) { sx[k] = seeds[2 * (base + k)]; sy[k] = seeds[2 * (base + k) + 1]; }
        __syncthreads();
        const int n = (int)((ns - base) < 1024 ? (ns - base) : 1024);
        for (int k = 0; k < n; ++k) {
            const double dx = x - sx[k], dy = y - sy[k];
            const double d2 = dx * dx + dy * dy;
            best = d2 < best ? d2 : best;
        }
    }
    if (i < nq) dist[i] = sqrt(best);
}

// ---- the same distances through the grid of buckets (round 5) ----
// The seeds are binned into the kGrid x kGrid cells of their bounding box (count, prefix sums, fill: k_grid_scan serves both
// grids); a query looks at the cells in rings of growing Chebyshev radius around its own (clamped) cell and stops when no
// cell further out can hold a closer seed: every seed of a cell outside the block of rings 0 .. r lies beyond one of the block's
// sides that still has grid behind it, i.e. at least `lb` = the smallest distance from the query to such a side away (a side
// the grid ends at has nothing behind it).  The bound is taken with a margin that covers the rounding of cell_of, so the
// minimum is over a superset of the seeds that could attain it, and every candidate distance is the SAME expression the
// brute-force kernel evaluates (dx dx + dy dy, no contraction): the result is bit-identical (test_nearest_keypoint_distance_is_exact,
// fixture G4), at ~25 cells x ~1 seed per query instead of all 20 000.
template <int PASS>
__global__ void k_seed_bin(const double *seeds, int64_t ns, GridGeo g, int32_t *count, const int32_t *start, int32_t *ids)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    const int cell = cell_of(seeds[2 * k + 1], g.y0, g.iy) * kGrid + cell_of(seeds[2 * k], g.x0, g.ix);
    const int32_t pos = atomicAdd(&count[cell], 1);
    if (PASS == 1) ids[start[cell] + pos] = (int32_t)k;
}
__global__ __launch_bounds__(256) void k_nearest_grid(const double *seeds, const int32_t *start, const int32_t *ids, GridGeo g, double wx, double wy,
                                                       double margin, const double *q, int64_t nq, double *dist, int64_t qcols)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const double x = q ? q[2 * i] : (double)(i / qcols), y = q ? q[2 * i + 1] : (double)(i % qcols);
    const int cx = cell_of(x, g.x0, g.ix), cy = cell_of(y, g.y0, g.iy);
    double best = INFINITY;
    for (int r = 0; r < kGrid; ++r) {
        const int xa = cx - r, xb = cx + r, ya = cy - r, yb = cy + r;
        // ring r: the cells of the block [xa, xb] x [ya, yb] that are not in ring r - 1 (clipped to the grid)
        auto visit = [&](int cell) {
            for (int32_t e = start[cell]; e < start[cell + 1]; ++e) {
                const int32_t k = ids[e];
                const double dx = x - seeds[2 * k], dy = y - seeds[2 * k + 1];
                const double d2 = dx * dx + dy * dy;
                best = d2 < best ? d2 : best;
            }
        };
        const int x_lo = xa < 0 ? 0 : xa, x_hi = xb > kGrid - 1 ? kGrid - 1 : xb;
        for (int yy = (ya < 0 ? 0 : ya); yy <= (yb > kGrid - 1 ? kGrid - 1 : yb); ++yy) {
            if (yy == ya || yy == yb) {
                for (int xx = x_lo; xx <= x_hi; ++xx) visit(yy * kGrid + xx);
            } else {
                if (xa >= 0) visit(yy * kGrid + xa);
                if (xb <= kGrid - 1) visit(yy * kGrid + xb);           // (r >= 1 here: xb != xa)
            }
        }
        // nothing closer outside the block?  sides the grid ends at have nothing behind them
        double lb = INFINITY;
        if (xa > 0) lb = fmin(lb, x - (g.x0 + (double)xa * wx));
        if (xb < kGrid - 1) lb = fmin(lb, (g.x0 + (double)(xb + 1) * wx) - x);
        if (ya > 0) lb = fmin(lb, y - (g.y0 + (double)ya * wy));
        if (yb < kGrid - 1) lb = fmin(lb, (g.y0 + (double)(yb + 1) * wy) - y);
        if (lb == INFINITY) break;                                     // the block covers the grid
        lb -= margin;
        if (lb > 0.0 && best <= lb * lb) break;
    }
    dist[i] = sqrt(best);
}

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = fail(SID_PM_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)

// Device scratch of the two entry points: one grow-only block per device, carved per call (no hipMalloc / hipFree per
// call - seven and three of them cost about as much as the kernels).  Calls are serialised by the pool's mutex.
struct Pool { unsigned char *p = nullptr; size_t cap = 0; };
std::mutex g_pool_mu;
Pool g_pool[16];

struct Carver {
    unsigned char *base; size_t off = 0;
    explicit Carver(unsigned char *b) : base(b) {}
    template <typename T> T *take(size_t n) { T *r = reinterpret_cast<T *>(base + off); off += (n * sizeof(T) + 255) / 256 * 256; return r; }
};

int pool_reserve(int device, size_t bytes, unsigned char **out)
{
    if (device < 0 || device >= 16) return fail(SID_PM_ERR_ARG, "device index out of range");
    Pool &pl = g_pool[device];
    if (pl.cap < bytes) {
        if (pl.p) (void)hipFree(pl.p);
        pl.p = nullptr; pl.cap = 0;
        const size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&pl.p), want);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? SID_PM_ERR_NOMEM : SID_PM_ERR_HIP, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        pl.cap = want;
    }
    *out = pl.p;
    return SID_PM_OK;
}

int pick_device(int device, int &prev)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(SID_PM_ERR_NODEVICE, "no such HIP device");
    (void)hipGetDevice(&prev); (void)hipSetDevice(device);
    return SID_PM_OK;
}

}  // namespace

SID_EXPORT int sid_fg_release(int device)
{
    std::lock_guard<std::mutex> lock(g_pool_mu);
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int d = 0; d < 16; ++d) {
        if (device >= 0 && d != device) continue;
        if (g_pool[d].p) { (void)hipSetDevice(d); (void)hipFree(g_pool[d].p); g_pool[d].p = nullptr; g_pool[d].cap = 0; }
    }
    (void)hipSetDevice(prev);
    return SID_PM_OK;
}

SID_EXPORT const char *sid_fg_last_error(void) { return g_err; }

SID_EXPORT int sid_fg_interp_linear(int device, const double *pts, int64_t n_pts, const int32_t *simplices, int64_t n_simp,
                                    const double *values, const double *q, int64_t n_q, double *out, int32_t *simplex, int32_t *doubt)
{
    if (n_q == 0) return SID_PM_OK;
    if (!pts || !simplices || !values || !q || !out || n_pts < 3 || n_simp < 1 || n_q < 0 || n_simp >= 0x7f7f7f7f) return fail(SID_PM_ERR_ARG, "bad argument");
    int prev = 0;
    if (int rc0 = pick_device(device, prev)) return rc0;
    int rc = SID_PM_OK;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    // bucket lists of the uniform grid: room for 8 entries per simplex on average (a Delaunay triangle of scattered points
    // touches 2-4 cells); a triangulation that needs more - long slivers across the whole box - takes the brute-force pass
    const int64_t grid_cap = 8 * n_simp + kGrid * kGrid;
    const bool want_grid = getenv("SID_FG_NO_GRID") == nullptr;
    const size_t need = up(sizeof(double) * 2 * n_pts) * 2 + up(sizeof(double) * 2 * n_q) * 2 + up(sizeof(int32_t) * 3 * n_simp) +
                        up(sizeof(Simp) * n_simp) + up(sizeof(int32_t) * n_q) * 5 + 256 +
                        up(sizeof(int32_t) * (kGrid * kGrid + 1)) * 2 + up(sizeof(int32_t) * grid_cap);
    unsigned char *blk = nullptr;
    double *d_pts = nullptr, *d_val = nullptr, *d_q = nullptr, *d_out = nullptr; int32_t *d_simp = nullptr, *d_loc = nullptr, *d_near = nullptr, *d_sx = nullptr, *d_dbt = nullptr, *d_list = nullptr, *d_nlist = nullptr; Simp *d_t = nullptr;
    int32_t *d_cnt = nullptr, *d_start = nullptr, *d_ids = nullptr;
    bool use_grid = false;
    GridGeo geo{0.0, 0.0, 0.0, 0.0};
    if ((rc = pool_reserve(device, need, &blk))) { (void)hipSetDevice(prev); return rc; }
    {
        Carver cv(blk);
        d_pts = cv.take<double>(2 * n_pts); d_val = cv.take<double>(2 * n_pts); d_q = cv.take<double>(2 * n_q); d_out = cv.take<double>(2 * n_q);
        d_simp = cv.take<int32_t>(3 * n_simp); d_t = cv.take<Simp>(n_simp); d_loc = cv.take<int32_t>(n_q);
        d_near = cv.take<int32_t>(n_q); d_sx = cv.take<int32_t>(n_q); d_dbt = cv.take<int32_t>(n_q); d_list = cv.take<int32_t>(n_q); d_nlist = cv.take<int32_t>(1);
        d_cnt = cv.take<int32_t>(kGrid * kGrid + 1); d_start = cv.take<int32_t>(kGrid * kGrid + 1); d_ids = cv.take<int32_t>(grid_cap);
    }
    {   // grid geometry: the bounding box of the key points
        double x_lo = pts[0], x_hi = pts[0], y_lo = pts[1], y_hi = pts[1];
        for (int64_t k = 1; k < n_pts; ++k) {
            x_lo = fmin(x_lo, pts[2 * k]); x_hi = fmax(x_hi, pts[2 * k]); y_lo = fmin(y_lo, pts[2 * k + 1]); y_hi = fmax(y_hi, pts[2 * k + 1]);
        }
        const double w = x_hi - x_lo, h = y_hi - y_lo;
        if (want_grid && w > 0.0 && h > 0.0 && isfinite(w) && isfinite(h)) { geo = GridGeo{x_lo, y_lo, (double)kGrid / w, (double)kGrid / h}; use_grid = true; }
    }
    HIP_TRY(hipMemcpyAsync(d_pts, pts, sizeof(double) * 2 * n_pts, hipMemcpyHostToDevice, 0));
    HIP_TRY(hipMemcpyAsync(d_val, values, sizeof(double) * 2 * n_pts, hipMemcpyHostToDevice, 0));
    if (q) HIP_TRY(hipMemcpyAsync(d_q, q, sizeof(double) * 2 * n_q, hipMemcpyHostToDevice, 0));
    HIP_TRY(hipMemcpyAsync(d_simp, simplices, sizeof(int32_t) * 3 * n_simp, hipMemcpyHostToDevice, 0));
    hipLaunchKernelGGL(k_transform, dim3((unsigned)((n_simp + 255) / 256)), dim3(256), 0, 0, d_pts, d_simp, n_simp, d_t);
    {
        const unsigned qb = (unsigned)((n_q + 255) / 256), sb = (unsigned)((n_simp + 255) / 256);
        if (use_grid) {
            // bucket lists: count, prefix sums, fill; the total decides whether they fit (one 4-byte read-back)
            int32_t total = 0;
            HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * (kGrid * kGrid + 1), 0));
            hipLaunchKernelGGL(k_grid_build<0>, dim3(sb), dim3(256), 0, 0, d_pts, d_simp, d_t, n_simp, geo, d_cnt, d_start, d_ids, grid_cap);
            hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(1024), 0, 0, d_cnt, d_start);
            HIP_TRY(hipMemcpy(&total, d_start + kGrid * kGrid, sizeof total, hipMemcpyDeviceToHost));
            if ((int64_t)total <= grid_cap) hipLaunchKernelGGL(k_grid_build<1>, dim3(sb), dim3(256), 0, 0, d_pts, d_simp, d_t, n_simp, geo, d_cnt, d_start, d_ids, grid_cap);
            else use_grid = false;
        }
        if (use_grid) {
            hipLaunchKernelGGL(k_locate_grid, dim3(qb), dim3(256), 0, 0, d_t, d_start, d_ids, geo, d_q, n_q, d_loc, d_near);
        } else {
            unsigned ychunks = qb >= 2048 ? 1u : (2048u + qb - 1) / qb;                 // aim at >= 2048 blocks
            const int64_t tiles = (n_simp + kTile - 1) / kTile;
            if ((int64_t)ychunks > tiles) ychunks = (unsigned)tiles;
            const int64_t chunk = ((tiles + ychunks - 1) / ychunks) * kTile;
            ychunks = (unsigned)((n_simp + chunk - 1) / chunk);
            HIP_TRY(hipMemsetAsync(d_loc, 0x7f, sizeof(int32_t) * n_q, 0));             // 0x7f7f7f7f: above any index
            HIP_TRY(hipMemsetAsync(d_near, 0, sizeof(int32_t) * n_q, 0));
            hipLaunchKernelGGL(k_locate, dim3(qb, ychunks), dim3(256), 0, 0, d_t, n_simp, chunk, d_q, n_q, d_loc, d_near);
        }
        HIP_TRY(hipMemsetAsync(d_nlist, 0, sizeof(int32_t), 0));
        hipLaunchKernelGGL(k_eval, dim3(qb), dim3(256), 0, 0, d_t, d_loc, d_near, d_val, d_q, n_q, d_out, d_sx, d_dbt, d_list, d_nlist);
        // (the number of flagged queries is only known on the device: enough wavefronts for a few thousand of them at once,
        // each taking every 4096th entry of the list)
        hipLaunchKernelGGL(k_resolve, dim3(1024), dim3(256), 0, 0, d_t, n_simp, d_val, d_q, d_list, d_nlist, d_out, d_sx, d_dbt,
                           use_grid ? d_start : nullptr, use_grid ? d_ids : nullptr, geo);
    }
    HIP_TRY(hipGetLastError());
    if (simplex) HIP_TRY(hipMemcpyAsync(simplex, d_sx, sizeof(int32_t) * n_q, hipMemcpyDeviceToHost, 0));
    if (doubt) HIP_TRY(hipMemcpyAsync(doubt, d_dbt, sizeof(int32_t) * n_q, hipMemcpyDeviceToHost, 0));
    HIP_TRY(hipMemcpy(out, d_out, sizeof(double) * 2 * n_q, hipMemcpyDeviceToHost));    // (synchronous: the host arrays are the caller's)
    HIP_TRY(hipStreamSynchronize(0));
done:
    (void)hipSetDevice(prev);
    return rc;
}

static int nearest_dist_impl(int device, const double *seeds, int64_t n_seeds, const double *q, int64_t n_q, double *dist, int64_t qcols);

SID_EXPORT int sid_fg_nearest_dist(int device, const double *seeds, int64_t n_seeds, const double *q, int64_t n_q, double *dist)
{
    if (n_q == 0) return SID_PM_OK;
    if (!seeds || !q || !dist || n_seeds < 1 || n_q < 0) return fail(SID_PM_ERR_ARG, "bad argument");
    return nearest_dist_impl(device, seeds, n_seeds, q, n_q, dist, 1);
}

SID_EXPORT int sid_fg_distance_image(int device, const double *seeds, int64_t n_seeds, int64_t rows, int64_t cols, double *dist)
{
    if (rows == 0 || cols == 0) return SID_PM_OK;
    if (!seeds || !dist || n_seeds < 1 || rows < 0 || cols < 0 || rows > (int64_t)1 << 31 || cols > (int64_t)1 << 31 || rows * cols > (int64_t)1 << 36)
        return fail(SID_PM_ERR_ARG, "bad argument");
    return nearest_dist_impl(device, seeds, n_seeds, nullptr, rows * cols, dist, cols);
}

// q == nullptr: the queries are the pixels (row, column) of an image with `qcols` columns
static int nearest_dist_impl(int device, const double *seeds, int64_t n_seeds, const double *q, int64_t n_q, double *dist, int64_t qcols)
{
    int prev = 0;
    if (int rc0 = pick_device(device, prev)) return rc0;
    int rc = SID_PM_OK;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    unsigned char *blk = nullptr;
    double *d_s = nullptr, *d_q = nullptr, *d_d = nullptr;
    int32_t *d_cnt = nullptr, *d_start = nullptr, *d_ids = nullptr;
    if ((rc = pool_reserve(device, up(sizeof(double) * 2 * n_seeds) + up(sizeof(double) * 2 * (q ? n_q : 1)) + up(sizeof(double) * n_q) +
                                   up(sizeof(int32_t) * (kGrid * kGrid + 1)) * 2 + up(sizeof(int32_t) * n_seeds), &blk))) { (void)hipSetDevice(prev); return rc; }
    {
        Carver cv(blk);
        d_s = cv.take<double>(2 * n_seeds); d_q = cv.take<double>(2 * (q ? n_q : 1)); d_d = cv.take<double>(n_q);
        if (!q) d_q = nullptr;
        d_cnt = cv.take<int32_t>(kGrid * kGrid + 1); d_start = cv.take<int32_t>(kGrid * kGrid + 1); d_ids = cv.take<int32_t>(n_seeds);
    }
    HIP_TRY(hipMemcpyAsync(d_s, seeds, sizeof(double) * 2 * n_seeds, hipMemcpyHostToDevice, 0));
    if (q) HIP_TRY(hipMemcpyAsync(d_q, q, sizeof(double) * 2 * n_q, hipMemcpyHostToDevice, 0));
    {
        // bounding box of the seeds (host: they are host arrays); the buckets pay from a few hundred seeds on, and need finite
        // coordinates and a box with an area (SID_FG_NO_GRID=1: brute force always; A/B runs and the parity test of both)
        double x_lo = INFINITY, x_hi = -INFINITY, y_lo = INFINITY, y_hi = -INFINITY;
        bool finite = true;
        for (int64_t k = 0; k < n_seeds; ++k) {
            const double sx = seeds[2 * k], sy = seeds[2 * k + 1];
            finite = finite && isfinite(sx) && isfinite(sy);
            x_lo = fmin(x_lo, sx); x_hi = fmax(x_hi, sx); y_lo = fmin(y_lo, sy); y_hi = fmax(y_hi, sy);
        }
        const double w = x_hi - x_lo, h = y_hi - y_lo;
        if (getenv("SID_FG_NO_GRID") == nullptr && n_seeds >= 256 && finite && w > 0.0 && h > 0.0 && isfinite(w) && isfinite(h)) {
            const GridGeo geo{x_lo, y_lo, (double)kGrid / w, (double)kGrid / h};
            // rounding of (v - v0) * inv in cell_of moves a seed by at most a few ulp of the box across a cell border
            const double margin = 1e-9 * (w + h) + 1e-12 * (fabs(x_lo) + fabs(x_hi) + fabs(y_lo) + fabs(y_hi));
            const unsigned nb = (unsigned)((n_seeds + 255) / 256);
            HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * (kGrid * kGrid + 1), 0));
            hipLaunchKernelGGL(k_seed_bin<0>, dim3(nb), dim3(256), 0, 0, d_s, n_seeds, geo, d_cnt, (const int32_t *)nullptr, (int32_t *)nullptr);
            hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(1024), 0, 0, d_cnt, d_start);
            hipLaunchKernelGGL(k_seed_bin<1>, dim3(nb), dim3(256), 0, 0, d_s, n_seeds, geo, d_cnt, d_start, d_ids);
            hipLaunchKernelGGL(k_nearest_grid, dim3((unsigned)((n_q + 255) / 256)), dim3(256), 0, 0, d_s, d_start, d_ids, geo, w / kGrid, h / kGrid, margin,
                               (const double *)d_q, n_q, d_d, qcols);
        } else {
            hipLaunchKernelGGL(k_nearest, dim3((unsigned)((n_q + 255) / 256)), dim3(256), 0, 0, d_s, n_seeds, (const double *)d_q, n_q, d_d, qcols);
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(dist, d_d, sizeof(double) * n_q, hipMemcpyDeviceToHost));
done:
    (void)hipSetDevice(prev);
    return rc;
}
