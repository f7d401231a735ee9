// pm_large.hip - rotate_and_match (reference pmlib.py:117-174) for search windows and template sides BEYOND the launch classes
// of the one-workgroup-per-point kernels (pm_kernel_mfma.hip / pm_kernel_rp.inc keep a point's whole state in LDS: windows up
// to 257 px, template sides up to 64).  Here the placement space of ONE point is tiled over the whole device:
//
//   lw_templates   K rotated templates (scipy order 0 / 1 arithmetic, pmlib.py:89-115) -> uint8 [K][s][s], their sums, the
//                  zero-pixel flag (pmlib.py:152-154) and the int8 operand table of the matrix instructions
//   lw_rowsums /   exact sum w' and sum w'^2 of every placement (w' = w - 128): running sums along rows, then down columns
//   lw_colsums
//   lw_corr        exact sum w' t' of every placement and angle on v_mfma_i32_16x16x64_i8 - M = 16 angles, N = 16 placements,
//                  K = 64 template columns of one template row; a workgroup owns a tile of 16 x 64 placements staged through
//                  LDS, a wavefront a band of 4 output rows fed by one window fragment per row (the band trick of the
//                  one-point kernels) - normalised at once into the float32 NCC value of the specification (DESIGN.md
//                  section 3; pmlib.py:156) and stored: the NCC matrices of ALL candidate angles live in global memory.
//                  Arg-max per angle: a 64-bit atomic max on (value key, ~flat index) - np.argmax's first maximum.
//   lw_pick        angle pick with the reference's strict > (pmlib.py:158-165)
//   lw_smooth /    get_hessian (pmlib.py:36-59) over the winning matrix: optional sigma-1 Gaussian, np.gradient twice, hypot;
//   lw_hessian     exact median by radix select over the global matrix (lw_hist / lw_hist_scan), std from float64 sums
//   lw_finish      displacement (pmlib.py:168-169), mcc_norm (:171-172), the five results
//
// Nothing is read back by the host between the kernels: the state of a point (LwState) lives in device memory, so a run of
// large points is a stream of launches - and the points of a run go through it in BATCHES: every kernel takes an array of
// per-point parameters and one launch dimension is the point (lw_run_batch; a launch per phase and batch instead of ~25 per
// point, whose fixed cost - 0.15 ms - was most of a point's time).  HBM-bound except lw_corr; sized for 288 GB (K x placements x 4 B of NCC values).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <set>
#include <utility>

#include "pm_kernel.h"
#include "pm_large.h"

namespace sid {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u32;
typedef unsigned long long u64;

constexpr int kTH = 16, kTW = 64;          // placements per workgroup tile of lw_corr (rows x columns)

struct LwState {
    u64 best[kMaxAngles];                  // per angle: (key of the maximum << 32) | (0xffffffff - flat index of its first occurrence)
    double sT[kMaxAngles], rT[kMaxAngles]; // sum t' and 1 / sqrt(N sum t'^2 - (sum t')^2) per template
    int32_t constT[kMaxAngles];            // dT == 0: the NCC matrix is all ones (OpenCV)
    int32_t zero;                          // a template holds a 0 pixel: NaN x 7 (pmlib.py:152-154)
    int32_t valid, kbest, iy, ix;
    float best_r;
    u32 sel_prefix[2][2], sel_rank[2][2];  // radix select of two order statistics (the middles) of [0] the Hessian magnitudes, [1] the NCC matrix
    u32 hist[2][2][256];
    double sums[2][2];                     // sum x, sum x^2 of the same two matrices
};

struct LwParams {
    const uint8_t *img1; long long rows1, cols1, stride1, row0, col0;   // image 1 (row0 / col0: origin of the part that is on the device)
    const double *coef;                                                 // rot_order 2..5: image 1 through scipy's spline prefilter, [rows1][cols1] float64
    const uint8_t *win; long long wstride;                             // first pixel of the window on image 2
    int wh, ww, rh, rw, s, K;
    u32 flags;
    double c1, r1;
    const double *rot, *angles;
    double add_c, add_r;
    double gw[5];
    LwState *st;
    uint8_t *tmpl; uint8_t *opA;
    int32_t *hs1; u32 *hs2; int32_t *si; double *ri;                   // ri: 1 / sqrt(dI) per placement, -1 = OpenCV's low-variance rule (lw_colsums)
    float *ncc, *hes, *tmpa, *tmpb;
    double *partial;
    double *out5; int32_t *ij3;
};

// source / destination matrices of the get_hessian kernels, by name (the pointers are per point): the NCC matrix of the winning
// angle, the two smoothing buffers, the Hessian magnitudes
enum { M_NCC = 0, M_TMPA = 1, M_TMPB = 2, M_HES = 3 };
__device__ __forceinline__ float *lw_matrix(const LwParams &P, int sel)
{
    switch (sel) {
    case M_NCC: return P.ncc + (size_t)P.st->kbest * ((size_t)P.rh * P.rw);
    case M_TMPA: return P.tmpa;
    case M_TMPB: return P.tmpb;
    default: return P.hes;
    }
}

__device__ __forceinline__ u32 f2key(float f) { u32 b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float key2f(u32 k) { u32 b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; return __uint_as_float(b); }

// ---------------------------------------------------------------------------------------------- templates
// One sample of get_template: scipy's NI_GeometricTransform coordinates (matrix = transform.T, float64, the reference's operation
// order), order 0: floor(x + 0.5); order 1: the four taps in float64, uint8 output rounding (oracle: sid_oracle_get_template /
// sid_oracle_get_template1; fixtures G1 / G1b).
__device__ __forceinline__ int lw_sample(const uint8_t *img, long long stride, long long row0, long long col0, long long rows, long long cols,
                                         double c, double r, const double *rot4, int i, int j, bool lin)
{
    const double cosa = rot4[0], sina = rot4[1];
    const double off0 = r - rot4[2], off1 = c - rot4[3];
    double rr = 0.0 + (double)i * cosa;
    rr = rr + (double)j * sina;
    rr = rr + off0;
    double cc = 0.0 + (double)i * (-sina);
    cc = cc + (double)j * cosa;
    cc = cc + off1;
    if (!(rr >= 0.0 && rr <= (double)(rows - 1) && cc >= 0.0 && cc <= (double)(cols - 1))) return 0;
    if (lin) {
        const double fr = floor(rr), fc = floor(cc);
        const double yr = rr - fr, yc = cc - fc;
        const double w0r = 1.0 - yr, w0c = 1.0 - yc;
        const long long r0 = (long long)fr, c0 = (long long)fc;
        long long r1 = r0 + 1, c1 = c0 + 1;                         // the tap behind the last sample is mirrored (its weight is 0)
        if (r1 >= rows) r1 = 2 * rows - 2 - r1;
        if (c1 >= cols) c1 = 2 * cols - 2 - c1;
        if (r1 < 0) r1 = 0;
        if (c1 < 0) c1 = 0;
        const uint8_t *p0 = img + (r0 - row0) * stride - col0, *p1 = img + (r1 - row0) * stride - col0;
        double t = 0.0;
        t = t + ((double)p0[c0] * w0r) * w0c;
        t = t + ((double)p0[c1] * w0r) * yc;
        t = t + ((double)p1[c0] * yr) * w0c;
        t = t + ((double)p1[c1] * yr) * yc;
        t = t > 0.0 ? t + 0.5 : 0.0;
        t = t > 255.0 ? 255.0 : t;
        return (int)t;
    }
    const long long ri = (long long)floor(rr + 0.5), ci = (long long)floor(cc + 0.5);
    return img[(ri - row0) * stride + (ci - col0)];
}

// ---- rot_order 2..5 (pmlib.py:112-113: scipy's affine_transform with order = n): spline weights and taps over the prefiltered
// image (oracle: sid_oracle_get_template_spline; weights and coefficients pinned against scipy itself, fixture G1c against the
// reference's get_template) ----
__device__ __forceinline__ void lw_spline_weights(double x, int order, double *w)
{
    x -= floor((order & 1) ? x : x + 0.5);
    double y = x, z = 1.0 - x, t;
    switch (order) {
    case 2:
        w[1] = 0.75 - x * x;
        y = 0.5 - x;
        w[0] = 0.5 * y * y;
        break;
    case 3:
        w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
        w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
        w[0] = z * z * z / 6.0;
        break;
    case 4:
        t = x * x;
        w[2] = t * (t * 0.25 - 0.625) + 115.0 / 192.0;
        y = 1.0 + x;
        w[1] = y * (y * (y * (5.0 - y) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
        z = 1.0 - x;
        w[3] = z * (z * (z * (5.0 - z) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
        y = 0.5 - x;
        t = y * y;
        w[0] = t * t / 24.0;
        break;
    default:
        t = y * y;
        w[2] = t * (t * (0.25 - y / 12.0) - 0.5) + 0.55;
        t = z * z;
        w[3] = t * (t * (0.25 - z / 12.0) - 0.5) + 0.55;
        y = x + 1.0;
        w[1] = y * (y * (y * (y * (y / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
        y = 2.0 - x;
        w[4] = y * (y * (y * (y * (y / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
        z = 1.0 - x;
        t = z * z;
        w[0] = z * t * t / 120.0;
        break;
    }
    double last = 1.0;
    for (int i = 0; i < order; ++i) last -= w[i];
    w[order] = last;
}

__device__ __forceinline__ long long lw_spline_mirror(long long idx, long long len)
{
    if (len <= 1) return 0;
    const long long s2 = 2 * len - 2;
    if (idx < 0) { idx = s2 * (-idx / s2) + idx; return idx <= 1 - len ? idx + s2 : -idx; }
    if (idx >= len) { idx -= s2 * (idx / s2); if (idx >= len) idx = s2 - idx; }
    return idx;
}

__device__ __forceinline__ int lw_sample_spline(const double *coef, long long rows, long long cols, double c, double r, const double *rot4,
                                                int i, int j, int order)
{
    const double cosa = rot4[0], sina = rot4[1];
    const double off0 = r - rot4[2], off1 = c - rot4[3];
    double rr = 0.0 + (double)i * cosa;
    rr = rr + (double)j * sina;
    rr = rr + off0;
    double cc = 0.0 + (double)i * (-sina);
    cc = cc + (double)j * cosa;
    cc = cc + off1;
    if (!(rr >= 0.0 && rr <= (double)(rows - 1) && cc >= 0.0 && cc <= (double)(cols - 1))) return 0;
    const long long sr = (long long)floor((order & 1) ? rr : rr + 0.5) - order / 2, sc = (long long)floor((order & 1) ? cc : cc + 0.5) - order / 2;
    double wr[6], wc[6];
    lw_spline_weights(rr, order, wr);
    lw_spline_weights(cc, order, wc);
    double t = 0.0;
    for (int a = 0; a <= order; ++a) {
        const double *row = coef + lw_spline_mirror(sr + a, rows) * cols;
        for (int b = 0; b <= order; ++b) t += (row[lw_spline_mirror(sc + b, cols)] * wr[a]) * wc[b];
    }
    t = t > 0.0 ? t + 0.5 : 0.0;
    t = t > 255.0 ? 255.0 : t;
    return (int)t;
}

// scipy's spline prefilter (ni_splines.c apply_filter), one pole, one direction per launch, one LINE per thread (the recursion
// is sequential along a line; the lines - 10^4 of them - are the parallelism).  Element k of line L: base[L * line_stride +
// k * elem_stride].  CAUSAL: c[k] = (src[k] * gain) + z c[k-1] after the mirror initialisation of c[0]; ANTICAUSAL:
// c[k] = z (c[k+1] - src[k]) after the initialisation of c[n-1].  src and dst are different buffers (ping-pong), so that the
// loads of a block of eight elements are issued together ahead of the dependent chain.  Every operation is scipy's, in
// scipy's order (-ffp-contract=off); pow(z, n - 1) comes from the host's libm, as scipy's does.
template <typename TSrc, bool CAUSAL>
__global__ __launch_bounds__(64) void lw_spline_pass_kernel(const TSrc *__restrict__ src, double *__restrict__ dst, long long nlines, long long n,
                                                            long long s_line, long long s_elem, long long d_line, long long d_elem,
                                                            double z, double z_n_1, double gain)
{
    const long long L = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= nlines) return;
    const TSrc *sp = src + L * s_line;
    double *dp = dst + L * d_line;
    auto ld = [&](long long k) { return (double)sp[k * s_elem] * gain; };   // (gain = 1 after the first pass of a line: x * 1.0 is x)
    if (CAUSAL) {
        double z_i = z;
        double c0 = ld(0) + z_n_1 * ld(n - 1);
        for (long long i = 1; i < n - 1; ++i) {
            if (z_i == 0.0) break;                                    // (the remaining terms are +-0)
            c0 += z_i * (ld(i) + (z_n_1 != 0.0 ? z_n_1 * ld(n - 1 - i) : 0.0));
            z_i *= z;
        }
        c0 /= 1 - z_n_1 * z_n_1;
        dp[0] = c0;
        double prev = c0;
        for (long long i = 1; i < n; i += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i + u < n ? ld(i + u) : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i + u < n) { prev = v[u] + z * prev; dp[(i + u) * d_elem] = prev; }
        }
    } else {
        double nxt = (z * ld(n - 2) + ld(n - 1)) * z / (z * z - 1);
        dp[(n - 1) * d_elem] = nxt;
        for (long long i = n - 2; i >= 0; i -= 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i - u >= 0 ? ld(i - u) : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i - u >= 0) { nxt = z * (nxt - v[u]); dp[(i - u) * d_elem] = nxt; }
        }
    }
}

__global__ __launch_bounds__(256) void lw_templates_kernel(const LwParams *Pv)
{
    const LwParams &P = Pv[blockIdx.y];
    __shared__ long long red[256][2];
    __shared__ int redmin[256];
    const int k = blockIdx.x, tid = threadIdx.x, s = P.s;
    const int order = (int)((P.flags >> 3) & 7u);                    // rot_order 0..5 (include/sid_pm.h SID_PM_ROT_ORDER)
    const bool lin = order == 1;
    uint8_t *T = P.tmpl + (size_t)k * s * s;
    long long st = 0, stt = 0;
    int vmin = 255;
    for (int idx = tid; idx < s * s; idx += 256) {
        const int i = idx / s, j = idx - i * s;
        const int v = order >= 2 ? lw_sample_spline(P.coef, P.rows1, P.cols1, P.c1, P.r1, P.rot + 4 * k, i, j, order)
                                 : lw_sample(P.img1, P.stride1, P.row0, P.col0, P.rows1, P.cols1, P.c1, P.r1, P.rot + 4 * k, i, j, lin);
        T[idx] = (uint8_t)v;
        const long long tp = v - 128;
        st += tp; stt += tp * tp;
        vmin = v < vmin ? v : vmin;
    }
    red[tid][0] = st; red[tid][1] = stt; redmin[tid] = vmin;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) {
            red[tid][0] += red[tid + h][0]; red[tid][1] += red[tid + h][1];
            redmin[tid] = redmin[tid + h] < redmin[tid] ? redmin[tid + h] : redmin[tid];
        }
        __syncthreads();
    }
    if (tid == 0 && P.st) {
        const long long n = (long long)s * s, dT = n * red[0][1] - red[0][0] * red[0][0];
        P.st->sT[k] = (double)red[0][0];
        P.st->constT[k] = dT == 0 ? 1 : 0;
        P.st->rT[k] = dT == 0 ? 0.0 : 1.0 / sqrt((double)dT);
        if (redmin[0] == 0) atomicOr(&P.st->zero, 1);
    }
    if (!P.opA) return;
    // operand table of lw_corr: [group of 16 angles][template row -3 .. s + 2][block of 64 columns][k-group of 16 columns]
    // [angle slot][16 B] - a wavefront's A operand of one (row, block) is 1 KB contiguous, lane l at l * 16.  Rows outside the
    // template, columns beyond s and the slots of absent angles are zero (the table is cleared before the launch).
    const int nJB = (s + 63) >> 6, grp = k >> 4, slot = k & 15, nrows = s + 6;
    for (int e = tid; e < s * nJB * 4; e += 256) {
        const int i = e / (nJB * 4), rem = e - i * (nJB * 4), jb = rem >> 2, kg = rem & 3;
        u32 w[4];
        for (int q = 0; q < 4; ++q) {
            u32 x = 0;
            for (int b = 0; b < 4; ++b) {
                const int col = 64 * jb + 16 * kg + 4 * q + b;
                const u32 byte = col < s ? (u32)((int)T[i * s + col] - 128) & 0xffu : 0u;
                x |= byte << (8 * b);
            }
            w[q] = x;
        }
        uint4 *dst = reinterpret_cast<uint4 *>(P.opA + ((((size_t)grp * nrows + (size_t)(i + 3)) * nJB + jb) * 4 + kg) * 256 + slot * 16);
        *dst = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---------------------------------------------------------------------------------------------- box sums
// running sums along the window rows: hs1[rho][x] = sum_{j < s} w'[rho][x + j], hs2 likewise of w'^2.  A thread owns four
// consecutive x of one row: one full sum, three slides.
__global__ __launch_bounds__(256) void lw_rowsums_kernel(const LwParams *Pv)
{
    const LwParams &P = Pv[blockIdx.z];
    const int rho = blockIdx.y, x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= P.rw || rho >= P.wh) return;
    const uint8_t *row = P.win + (long long)rho * P.wstride;
    int a = 0; u32 b = 0;
    for (int j = 0; j < P.s; ++j) { const int v = (int)row[x0 + j] - 128; a += v; b += (u32)(v * v); }
    const size_t o = (size_t)rho * P.rw + x0;
    P.hs1[o] = a; P.hs2[o] = b;
    for (int d = 1; d < 4 && x0 + d < P.rw; ++d) {
        const int vo = (int)row[x0 + d - 1] - 128, vn = (int)row[x0 + d - 1 + P.s] - 128;
        a += vn - vo; b += (u32)(vn * vn) - (u32)(vo * vo);
        P.hs1[o + d] = a; P.hs2[o + d] = b;
    }
}

// ... then down the columns: si[y][x] = sum_{i < s} hs1[y + i][x], and at once the placement's share of the normalisation (DESIGN.md
// section 3): dI = N sum w'^2 - (sum w')^2 exactly, OpenCV's low-variance rule, rI = 1 / sqrt(dI) in IEEE double - once per
// placement here instead of once per placement AND k-group in lw_corr's epilogue (the IEEE square root and division were a
// third of that kernel).  ri = -1 marks a low-variance placement.  A thread owns 32 consecutive y of one column (coalesced in x).
__global__ __launch_bounds__(256) void lw_colsums_kernel(const LwParams *Pv)
{
    const LwParams &P = Pv[blockIdx.z];
    const int x = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * 32;
    if (x >= P.rw || y0 >= P.rh) return;
    const size_t rw = (size_t)P.rw;
    const double nd = (double)P.s * (double)P.s;
    auto put = [&](size_t o, int a, u32 b) {
        const double swd = (double)a, siid = (double)b;
        const double dI = nd * siid - swd * swd;                           // exact
        const double s2 = siid + 256.0 * swd + 16384.0 * nd;               // sum w^2 in the uint8 domain
        const bool lowvar = (2.0 * dI <= nd) && (dI * 8388608.0 <= 10.0 * nd * s2);
        P.si[o] = a;
        P.ri[o] = lowvar ? -1.0 : 1.0 / sqrt(dI);
    };
    int a = 0; u32 b = 0;
    for (int i = 0; i < P.s; ++i) { a += P.hs1[(size_t)(y0 + i) * rw + x]; b += P.hs2[(size_t)(y0 + i) * rw + x]; }
    put((size_t)y0 * rw + x, a, b);
    for (int d = 1; d < 32 && y0 + d < P.rh; ++d) {
        a += P.hs1[(size_t)(y0 + d - 1 + P.s) * rw + x] - P.hs1[(size_t)(y0 + d - 1) * rw + x];
        b += P.hs2[(size_t)(y0 + d - 1 + P.s) * rw + x] - P.hs2[(size_t)(y0 + d - 1) * rw + x];
        put((size_t)(y0 + d) * rw + x, a, b);
    }
}

// ---------------------------------------------------------------------------------------------- correlation + NCC
// The specification's normalisation (oracle match_template_core): exact integers, then IEEE double with
// one rounding per operation (this file is compiled with -ffp-contract=off like the others).
__device__ __forceinline__ float lw_ncc(int acc, double swd, double rI, double nd, double sT, double rT, bool constT)
{
    if (constT) return 1.0f;
    if (rI < 0.0) return 0.0f;                                        // (low-variance placement: lw_colsums)
    const double numer = nd * (double)acc - swd * sT;                 // exact
    double q = numer * rI;
    q = q * rT;
    const double aq = fabs(q);
    return aq < 1.0 ? (float)q : (aq < 1.125 ? (q > 0.0 ? 1.0f : -1.0f) : 0.0f);
}

__host__ __device__ inline int lw_pitch(int s) { return kTW + 64 * ((s + 63) >> 6); }

__global__ __launch_bounds__(256) void lw_corr_kernel(const LwParams *Pv, int ngroups)
{
    const LwParams &P = Pv[blockIdx.z / ngroups];
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int s = P.s, nJB = (s + 63) >> 6, pitch = lw_pitch(s), rowsT = kTH + s - 1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kg = lane >> 4;
    const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH, grp = blockIdx.z % ngroups;
    if (x0 >= P.rw || y0 >= P.rh) return;                            // (the grid is sized for the largest window of the batch)
    // window tile -> LDS as re-centred int8; outside the window: 0 (those products belong to placements that are not stored)
    for (int e = tid; e < rowsT * (pitch >> 2); e += 256) {
        const int tr = e / (pitch >> 2), tc = (e - tr * (pitch >> 2)) << 2;
        const int wy = y0 + tr;
        u32 x = 0;
        if (wy < P.wh) {
            const uint8_t *row = P.win + (long long)wy * P.wstride;
            for (int b = 0; b < 4; ++b) {
                const int wx = x0 + tc + b;
                const u32 byte = wx < P.ww ? (u32)((int)row[wx] - 128) & 0xffu : 0u;
                x |= byte << (8 * b);
            }
        }
        *reinterpret_cast<u32 *>(smem + tr * pitch + tc) = x;
    }
    __syncthreads();
    v4i acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = v4i{0, 0, 0, 0};
    const int nrows = s + 6;
    const uint8_t *opA = P.opA + (size_t)grp * nrows * nJB * 1024 + (size_t)lane * 16;
    const u32 sh = (u32)(n & 3);
    const uint8_t *bbase = smem + (4 * wv) * pitch + 16 * kg + (n & ~3);
    for (int jb = 0; jb < nJB; ++jb) {
        v4i a1 = v4i{0, 0, 0, 0}, a2 = a1, a3 = a1;                   // template rows rho - 1, rho - 2, rho - 3
        // the A operands come from global memory (the table of a 50 px template is 56 KB: L2-resident, not LDS-sized): four steps
        // ahead, so that their round trip is covered by the MFMAs of the steps in between (rows beyond s + 2: the last zero row)
        const int last = s + 5;                                       // table row index of the last zero row
        auto lda = [&](int rho) { const int r = rho + 3 < last ? rho + 3 : last; return *reinterpret_cast<const v4i *>(opA + ((size_t)r * nJB + jb) * 1024); };
        auto step = [&](int rho, const v4i a0) {                      // window row rho of the band; output row yb pairs it with template row rho - yb
            const uint8_t *brow = bbase + rho * pitch + 64 * jb;
#pragma unroll
            for (int xt = 0; xt < 4; ++xt) {
                const u32 *q = reinterpret_cast<const u32 *>(brow + 16 * xt);
                const u32 r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3], r4 = q[4];
                v4i b;
                b[0] = (int)__builtin_amdgcn_alignbyte(r1, r0, sh);
                b[1] = (int)__builtin_amdgcn_alignbyte(r2, r1, sh);
                b[2] = (int)__builtin_amdgcn_alignbyte(r3, r2, sh);
                b[3] = (int)__builtin_amdgcn_alignbyte(r4, r3, sh);
                acc[0][xt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b, acc[0][xt], 0, 0, 0);
                acc[1][xt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b, acc[1][xt], 0, 0, 0);
                acc[2][xt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b, acc[2][xt], 0, 0, 0);
                acc[3][xt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, b, acc[3][xt], 0, 0, 0);
            }
            a3 = a2; a2 = a1; a1 = a0;
        };
        v4i q0 = lda(0), q1 = lda(1), q2 = lda(2), q3 = lda(3);
#ifdef LW_ABLATE_MAIN
        const int nst = 1;
#else
        const int nst = s + 3;
#endif
        for (int rho = 0; rho < nst; rho += 4) {
            step(rho, q0); q0 = lda(rho + 4);
            if (rho + 1 < nst) { step(rho + 1, q1); q1 = lda(rho + 5); }
            if (rho + 2 < nst) { step(rho + 2, q2); q2 = lda(rho + 6); }
            if (rho + 3 < nst) { step(rho + 3, q3); q3 = lda(rho + 7); }
        }
    }
    // accumulator register r of lane (n, kg) = angle slot 4 kg + r at placement column n of its tile
    const double nd = (double)s * (double)s;
    const size_t np = (size_t)P.rh * P.rw;
    double sT[4], rT[4]; bool cT[4], have[4];
    int kk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        kk[r] = grp * 16 + 4 * kg + r;
        have[r] = kk[r] < P.K;
        const int kq = have[r] ? kk[r] : 0;
        sT[r] = P.st->sT[kq]; rT[r] = P.st->rT[kq]; cT[r] = P.st->constT[kq] != 0;
    }
    u64 best[4] = {0ull, 0ull, 0ull, 0ull};
#ifdef LW_ABLATE_EPI
    if (acc[0][0][0] == 0x7fffffff) P.ncc[0] = 1.0f;
    return;
#endif
#pragma unroll
    for (int yb = 0; yb < 4; ++yb) {
        const int y = y0 + 4 * wv + yb;
#pragma unroll
        for (int xt = 0; xt < 4; ++xt) {
            const int x = x0 + 16 * xt + n;
            if (y < P.rh && x < P.rw) {
                const size_t p = (size_t)y * P.rw + x;
                const double swd = (double)P.si[p], rI = P.ri[p];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (have[r]) {
                        const float v = lw_ncc(acc[yb][xt][r], swd, rI, nd, sT[r], rT[r], cT[r]);
                        P.ncc[(size_t)kk[r] * np + p] = v;
                        const u64 key = ((u64)f2key(v) << 32) | (u64)(0xffffffffu - (u32)p);
                        best[r] = key > best[r] ? key : best[r];
                    }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        u64 b = best[r];
        for (int m = 1; m < 16; m <<= 1) {
            const u64 o = (u64)__shfl_xor((unsigned long long)b, m, 64);
            b = o > b ? o : b;
        }
        // (a load first: once the maximum of an angle has been seen, almost no tile beats it - 10^6 atomics on one cache line
        // were two thirds of this kernel's time)
        if (n == 0 && have[r] && b && b > __hip_atomic_load(&P.st->best[kk[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&P.st->best[kk[r]], b);
    }
}

// ---------------------------------------------------------------------------------------------- angle pick
__global__ void lw_pick_kernel(const LwParams *Pv)
{
    const LwParams &P = Pv[blockIdx.x];
    if (threadIdx.x != 0) return;
    LwState *S = P.st;
    const u32 n = (u32)((size_t)P.rh * P.rw);
    for (int w = 0; w < 2; ++w) {                                     // the two middles (equal for an odd count): np.median
        S->sel_prefix[w][0] = S->sel_prefix[w][1] = 0u;
        S->sel_rank[w][0] = (n & 1u) ? n / 2u : n / 2u - 1u;
        S->sel_rank[w][1] = n / 2u;
    }
    if (S->zero) {                                                    // pmlib.py:152-154
        S->valid = 0; S->kbest = 0;
        const double nan = __longlong_as_double(0x7ff8000000000000ll);
        for (int q = 0; q < 5; ++q) P.out5[q] = nan;
        if (P.ij3) { P.ij3[0] = P.ij3[1] = P.ij3[2] = -1; }
        return;
    }
    float bestv = -INFINITY;
    int bk = 0; u32 bidx = 0;
    for (int k = 0; k < P.K; ++k) {
        const u64 key = S->best[k];
        const float v = key2f((u32)(key >> 32));
        if (v > bestv) { bestv = v; bk = k; bidx = 0xffffffffu - (u32)key; }   // strict: the first angle keeps a tie (pmlib.py:160)
    }
    S->valid = 1; S->kbest = bk; S->best_r = bestv;
    S->iy = (int)(bidx / (u32)P.rw); S->ix = (int)(bidx % (u32)P.rw);
}

// ---------------------------------------------------------------------------------------------- get_hessian
// scipy.ndimage.gaussian_filter(ccm, 1) (pmlib.py:46-47): per axis a radius-4 kernel in double, 'reflect' boundary, float32
// after each axis (oracle gaussian_smooth_sigma1: centre first, then the pairs from the outermost in).
__device__ __forceinline__ int lw_reflect(int q, int n)
{
    while (q < 0 || q >= n) { if (q < 0) q = -q - 1; if (q >= n) q = 2 * n - 1 - q; }
    return q;
}

__global__ __launch_bounds__(256) void lw_smooth_kernel(const LwParams *Pv, int src_sel, int dst_sel, int pass)
{
    const LwParams &P = Pv[blockIdx.y];
    const size_t np = (size_t)P.rh * P.rw, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= np || !P.st->valid) return;
    const float *src = lw_matrix(P, src_sel);
    float *dst = lw_matrix(P, dst_sel);
    const int y = (int)(e / P.rw), x = (int)(e - (size_t)y * P.rw);
    const int p = pass == 0 ? y : x, n = pass == 0 ? P.rh : P.rw;
    double acc = (double)src[e] * P.gw[4];
    for (int k = 4; k >= 1; --k) {
        const int ql = lw_reflect(p - k, n), qr = lw_reflect(p + k, n);
        const double l = pass == 0 ? src[(size_t)ql * P.rw + x] : src[(size_t)y * P.rw + ql];
        const double r = pass == 0 ? src[(size_t)qr * P.rw + x] : src[(size_t)y * P.rw + qr];
        acc += (l + r) * P.gw[4 - k];
    }
    dst[e] = (float)acc;
}

__device__ __forceinline__ float lw_grad1(const float *f, size_t stride, int k, int n)
{
    if (k == 0) return f[stride] - f[0];
    if (k == n - 1) return f[(size_t)(n - 1) * stride] - f[(size_t)(n - 2) * stride];
    return (f[(size_t)(k + 1) * stride] - f[(size_t)(k - 1) * stride]) / 2.0f;
}
__device__ __forceinline__ float lw_grad2(const float *f, size_t stride, int k, int n)
{
    if (k == 0) return lw_grad1(f, stride, 1, n) - lw_grad1(f, stride, 0, n);
    if (k == n - 1) return lw_grad1(f, stride, n - 1, n) - lw_grad1(f, stride, n - 2, n);
    return (lw_grad1(f, stride, k + 1, n) - lw_grad1(f, stride, k - 1, n)) / 2.0f;
}

// sum and sum of squares of a block's values in double, in a fixed order -> partial[2 * block]
__device__ __forceinline__ void lw_block_sums(double a, double b, double *partial)
{
    __shared__ double red[256][2];
    const int tid = threadIdx.x;
    red[tid][0] = a; red[tid][1] = b;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) { red[tid][0] += red[tid + h][0]; red[tid][1] += red[tid + h][1]; }
        __syncthreads();
    }
    if (tid == 0) { partial[2 * (size_t)blockIdx.x] = red[0][0]; partial[2 * (size_t)blockIdx.x + 1] = red[0][1]; }
}

// np.gradient twice along each axis, np.hypot (pmlib.py:51-55) - float32 arithmetic as NumPy's; hypotf as glibc evaluates it
__global__ __launch_bounds__(256) void lw_hessian_kernel(const LwParams *Pv, int src_sel)
{
    const LwParams &P = Pv[blockIdx.y];
    const size_t np = (size_t)P.rh * P.rw, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((size_t)blockIdx.x * 256 >= np) return;                      // (block-uniform: the grid is sized for the largest matrix of the batch)
    float *hes = P.hes;
    double *partial = P.partial;
    double a = 0.0, b = 0.0;
    if (e < np && P.st->valid) {
        const float *src = lw_matrix(P, src_sel);
        const int y = (int)(e / P.rw), x = (int)(e - (size_t)y * P.rw);
        const float d2x = lw_grad2(src + (size_t)y * P.rw, 1, x, P.rw);
        const float d2y = lw_grad2(src + x, (size_t)P.rw, y, P.rh);
        const float h = (float)sqrt((double)d2x * (double)d2x + (double)d2y * (double)d2y);
        hes[e] = h;
        a = (double)h; b = (double)h * (double)h;
    }
    lw_block_sums(a, b, partial);
}

__global__ __launch_bounds__(256) void lw_sums_kernel(const LwParams *Pv, int src_sel)
{
    const LwParams &P = Pv[blockIdx.y];
    const size_t np = (size_t)P.rh * P.rw, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((size_t)blockIdx.x * 256 >= np) return;
    double *partial = P.partial;
    double a = 0.0, b = 0.0;
    if (e < np && P.st->valid) {
        const float v = lw_matrix(P, src_sel)[e];
        a = (double)v; b = (double)v * (double)v;
    }
    lw_block_sums(a, b, partial);
}

// second level: the partials of nblk blocks in a fixed order -> st->sums[which]
__global__ __launch_bounds__(256) void lw_sums_final_kernel(const LwParams *Pv, int which)
{
    const LwParams &P = Pv[blockIdx.x];
    const double *partial = P.partial;
    const int nblk = (int)(((size_t)P.rh * P.rw + 255) / 256);
    __shared__ double red[256][2];
    const int tid = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = tid; i < nblk; i += 256) { a += partial[2 * (size_t)i]; b += partial[2 * (size_t)i + 1]; }
    red[tid][0] = a; red[tid][1] = b;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) { red[tid][0] += red[tid + h][0]; red[tid][1] += red[tid + h][1]; }
        __syncthreads();
    }
    if (tid == 0) { P.st->sums[which][0] = red[0][0]; P.st->sums[which][1] = red[0][1]; }
}

// exact order statistics by radix select, 8 bits per pass over the global matrix: histogram of the next byte among the
// elements that match the prefix found so far, for both wanted ranks at once
__global__ __launch_bounds__(256) void lw_hist_kernel(const LwParams *Pv, int src_sel, int pass, int which)
{
    const LwParams &P = Pv[blockIdx.y];
    __shared__ u32 h[2][256];
    const int tid = threadIdx.x;
    h[0][tid] = 0u; h[1][tid] = 0u;
    __syncthreads();
    if (P.st->valid) {
        const float *src = lw_matrix(P, src_sel);
        const size_t np = (size_t)P.rh * P.rw;
        const u32 p0 = P.st->sel_prefix[which][0], p1 = P.st->sel_prefix[which][1];
        const int hi = 32 - 8 * pass, lo = 24 - 8 * pass;
        for (size_t e = (size_t)blockIdx.x * 256 + tid; e < np; e += (size_t)gridDim.x * 256) {
            const u32 key = f2key(src[e]);
            const u32 bin = (key >> lo) & 255u;
            if (pass == 0 || (key >> hi) == (p0 >> hi)) atomicAdd(&h[0][bin], 1u);
            if (pass == 0 || (key >> hi) == (p1 >> hi)) atomicAdd(&h[1][bin], 1u);
        }
    }
    __syncthreads();
    if (h[0][tid]) atomicAdd(&P.st->hist[which][0][tid], h[0][tid]);
    if (h[1][tid]) atomicAdd(&P.st->hist[which][1][tid], h[1][tid]);
}

__global__ void lw_hist_scan_kernel(const LwParams *Pv, int pass, int which)
{
    LwState *S = Pv[blockIdx.x].st;
    const int b = threadIdx.x;
    if (b < 2 && S->valid) {
        u32 cum = 0, rank = S->sel_rank[which][b];
        for (int bin = 0; bin < 256; ++bin) {
            const u32 c = S->hist[which][b][bin];
            if (rank < cum + c) { S->sel_prefix[which][b] |= (u32)bin << (24 - 8 * pass); S->sel_rank[which][b] = rank - cum; break; }
            cum += c;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) S->hist[which][i >> 8][i & 255] = 0u;
}

// np.median (mean of the two middles in float32) and np.std (from float64 sums: DESIGN.md section 3, 1e-5 on h)
__device__ __forceinline__ void lw_med_std(const LwState *S, int which, size_t n, float &med, float &sd)
{
    const float a = key2f(S->sel_prefix[which][0]), b = key2f(S->sel_prefix[which][1]);
    med = (n & 1) ? b : (a + b) / 2.0f;
    const double mean = S->sums[which][0] / (double)n;
    double var = S->sums[which][1] / (double)n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    sd = sqrtf((float)var);
}

__global__ void lw_finish_kernel(const LwParams *Pv)
{
    const LwParams &P = Pv[blockIdx.x];
    if (threadIdx.x != 0) return;
    const LwState *S = P.st;
    if (!S->valid) return;
    const size_t n = (size_t)P.rh * P.rw;
    float h = P.hes[(size_t)S->iy * P.rw + S->ix];
    if (P.flags & 1u) { float med, sd; lw_med_std(S, 0, n, med, sd); h = (h - med) / sd; }
    float r = S->best_r;
    if (P.flags & 4u) { float med, sd; lw_med_std(S, 1, n, med, sd); r = (r - med) / sd; }
    const double dr = (double)S->iy - (double)(P.wh - P.s) / 2., dc = (double)S->ix - (double)(P.ww - P.s) / 2.;   // pmlib.py:168-169
    P.out5[0] = P.add_c + dc; P.out5[1] = P.add_r + dr; P.out5[2] = P.angles[S->kbest];
    P.out5[3] = (double)r; P.out5[4] = (double)h;
    if (P.ij3) { P.ij3[0] = S->iy; P.ij3[1] = S->ix; P.ij3[2] = S->kbest; }
}

// get_hessian as a call of its own: the whole matrix normalised
__global__ __launch_bounds__(256) void lw_normalise_kernel(const LwParams *Pv)
{
    const LwParams &P = Pv[blockIdx.y];
    float *hes = P.hes;
    const size_t np = (size_t)P.rh * P.rw, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= np) return;
    float med, sd;
    lw_med_std(P.st, 0, np, med, sd);
    hes[e] = (hes[e] - med) / sd;
}

__global__ void lw_state_init_kernel(const LwParams *Pv)
{
    const LwParams &P = Pv[blockIdx.x];
    if (threadIdx.x != 0) return;
    LwState *S = P.st;
    const u32 n = (u32)((size_t)P.rh * P.rw);
    S->valid = 1; S->kbest = 0; S->zero = 0;
    for (int w = 0; w < 2; ++w) {
        S->sel_prefix[w][0] = S->sel_prefix[w][1] = 0u;
        S->sel_rank[w][0] = (n & 1u) ? n / 2u : n / 2u - 1u;
        S->sel_rank[w][1] = n / 2u;
    }
}

__global__ void lw_nan_kernel(const int32_t *idx, int n, double *out, int32_t *out_ij)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const size_t i = (size_t)idx[t];
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int q = 0; q < 5; ++q) out[i * 5 + q] = nan;
    if (out_ij) { out_ij[i * 3] = -1; out_ij[i * 3 + 1] = -1; out_ij[i * 3 + 2] = -1; }
}

// ---------------------------------------------------------------------------------------------- host
enum { B_STATE = 0, B_TMPL, B_OPA, B_HS1, B_HS2, B_SI, B_RI, B_NCC, B_HES, B_TMPA, B_TMPB, B_PART, B_PARAMS };
constexpr int kLwBatch = 64;                       // points per batch of launches (one launch dimension)
constexpr size_t kLwBatchBytes = (size_t)3 << 30;  // ... as long as their scratch stays below this

int reserve(LwWorkspace &W, int i, size_t bytes)
{
    if (bytes <= W.cap[i]) return 0;
    if (W.buf[i]) (void)hipFree(W.buf[i]);
    W.buf[i] = nullptr; W.cap[i] = 0;
    if (hipMalloc(&W.buf[i], bytes) != hipSuccess) { (void)hipGetLastError(); return -1; }
    W.cap[i] = bytes;
    return 0;
}

size_t opa_bytes(int s, int K) { return (size_t)((K + 15) / 16) * (size_t)(s + 6) * (size_t)((s + 63) >> 6) * 1024; }
size_t up256(size_t v) { return (v + 255) / 256 * 256; }

hipError_t allow_lds(int bytes)
{
    static std::mutex mu;
    static std::set<int> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(dev)) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(lw_corr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds_bytes());
    if (e == hipSuccess) done.insert(dev);
    (void)bytes;
    return e;
}

// median of one matrix per point of the batch: four histogram passes (two ranks at once) -> st->sel_prefix[which]
void launch_select(const LwParams *Pv, int G, size_t np_max, int src_sel, int which, hipStream_t st)
{
    const unsigned nb = (unsigned)std::min<size_t>((np_max + 255) / 256, 2048);
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(lw_hist_kernel, dim3(nb, (unsigned)G), dim3(256), 0, st, Pv, src_sel, pass, which);
        hipLaunchKernelGGL(lw_hist_scan_kernel, dim3((unsigned)G), dim3(64), 0, st, Pv, pass, which);
    }
}

// per-point scratch of a call, bytes per buffer (256-byte granules)
struct Need { size_t tmpl, opa, hs, s4, ncc, part; };
Need need_of(const LargeCall &c)
{
    const size_t rh = (size_t)(c.wh - c.s + 1), rw = (size_t)(c.ww - c.s + 1), np = rh * rw;
    return Need{up256((size_t)c.K * c.s * c.s), up256(opa_bytes(c.s, c.K)), up256((size_t)c.wh * rw * 4), up256(np * 4), up256((size_t)c.K * np * 4),
                up256(((np + 255) / 256) * 16)};
}
size_t total_of(const Need &n, uint32_t flags) { return n.tmpl + n.opa + 2 * n.hs + 4 * n.s4 + n.ncc + n.part + ((flags & 2u) ? 2 * n.s4 : 0); }   // (si, ri = 2 x, hes)

}  // namespace

void lw_workspace_release(LwWorkspace &W)
{
    for (int i = 0; i < 16; ++i) { if (W.buf[i]) (void)hipFree(W.buf[i]); W.buf[i] = nullptr; W.cap[i] = 0; }
}

size_t lw_scratch_bytes(int wh, int ww, int s, int K, uint32_t flags)
{
    LargeCall c;
    c.wh = wh; c.ww = ww; c.s = s; c.K = K;
    return sizeof(LwState) + total_of(need_of(c), flags);
}

// The calls of one run (same template side, angles and flags; any windows), in batches of up to kLwBatch points: one launch per
// phase and batch, the point = a launch dimension.
int lw_run_batch(const LargeCall *calls, int n, LwWorkspace &W, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n <= 0) return 0;
    const int s = calls[0].s, K = calls[0].K;
    const uint32_t flags = calls[0].flags;
    const int order = (int)((flags >> 3) & 7u);
    if (s < 2 || s > kLargeMaxSide || K < 1 || K > kMaxAngles || order > 5) return (int)hipErrorInvalidValue;
    for (int i = 0; i < n; ++i) {
        const LargeCall &c = calls[i];
        const int rh = c.wh - s + 1, rw = c.ww - s + 1;
        if (c.s != s || c.K != K || c.flags != flags || rh < 2 || rw < 2 || c.wh > 65535 || (size_t)rh * rw >= 0xffffffffull ||
            (order >= 2 && !c.d_coef)) return (int)hipErrorInvalidValue;   // (wh: a launch dimension of lw_rowsums)
    }
    // chunks, the scratch of the largest one, and every point's parameters with chunk-local scratch offsets
    std::vector<int> chunk_end;
    Need big{0, 0, 0, 0, 0, 0};
    size_t big_hs = 0, big_s4 = 0, big_ncc = 0, big_part = 0;
    std::vector<size_t> off_hs((size_t)n), off_s4((size_t)n), off_ncc((size_t)n), off_part((size_t)n);
    {
        size_t bytes = 0, o_hs = 0, o_s4 = 0, o_ncc = 0, o_part = 0;
        int first = 0;
        for (int i = 0; i < n; ++i) {
            const Need nd = need_of(calls[i]);
            const size_t t = total_of(nd, flags);
            if (i > first && (i - first >= kLwBatch || bytes + t > kLwBatchBytes)) {
                chunk_end.push_back(i); first = i; bytes = 0; o_hs = o_s4 = o_ncc = o_part = 0;
            }
            off_hs[(size_t)i] = o_hs; off_s4[(size_t)i] = o_s4; off_ncc[(size_t)i] = o_ncc; off_part[(size_t)i] = o_part;
            o_hs += nd.hs; o_s4 += nd.s4; o_ncc += nd.ncc; o_part += nd.part; bytes += t;
            big_hs = std::max(big_hs, o_hs); big_s4 = std::max(big_s4, o_s4); big_ncc = std::max(big_ncc, o_ncc); big_part = std::max(big_part, o_part);
            big.tmpl = nd.tmpl; big.opa = nd.opa;
        }
        chunk_end.push_back(n);
    }
    const int gmax = std::min(n, kLwBatch);
    // nothing of an earlier run on this stream may still read what is (re)allocated or overwritten below
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int)e;
    if (reserve(W, B_STATE, sizeof(LwState) * (size_t)kLwBatch) || reserve(W, B_TMPL, big.tmpl * gmax) || reserve(W, B_OPA, big.opa * gmax) ||
        reserve(W, B_HS1, big_hs) || reserve(W, B_HS2, big_hs) || reserve(W, B_SI, big_s4) || reserve(W, B_RI, 2 * big_s4) || reserve(W, B_NCC, big_ncc) ||
        reserve(W, B_HES, big_s4) || ((flags & 2u) && (reserve(W, B_TMPA, big_s4) || reserve(W, B_TMPB, big_s4))) || reserve(W, B_PART, big_part) ||
        reserve(W, B_PARAMS, sizeof(LwParams) * (size_t)n))
        return -1;
    std::vector<LwParams> hp((size_t)n);
    {
        int first = 0, ci = 0;
        for (int i = 0; i < n; ++i) {
            if (i == chunk_end[(size_t)ci]) { first = i; ++ci; }
            const LargeCall &c = calls[i];
            const int g = i - first;
            LwParams &P = hp[(size_t)i];
            memset(&P, 0, sizeof P);
            P.img1 = c.img1; P.rows1 = c.rows1; P.cols1 = c.cols1; P.stride1 = c.stride1; P.row0 = 0; P.col0 = 0; P.coef = c.d_coef;
            P.win = c.img2 + c.win_r0 * c.stride2 + c.win_c0; P.wstride = c.stride2;
            P.wh = c.wh; P.ww = c.ww; P.rh = c.wh - s + 1; P.rw = c.ww - s + 1; P.s = s; P.K = K; P.flags = flags;
            P.c1 = c.c1; P.r1 = c.r1; P.rot = c.d_rot; P.angles = c.d_angles; P.add_c = c.add_c; P.add_r = c.add_r;
            for (int q = 0; q < 5; ++q) P.gw[q] = c.gauss_w[q];
            auto at = [&](int b, size_t off) { return static_cast<uint8_t *>(W.buf[b]) + off; };
            P.st = static_cast<LwState *>(W.buf[B_STATE]) + g;
            P.tmpl = at(B_TMPL, big.tmpl * g); P.opA = at(B_OPA, big.opa * g);
            P.hs1 = reinterpret_cast<int32_t *>(at(B_HS1, off_hs[(size_t)i])); P.hs2 = reinterpret_cast<u32 *>(at(B_HS2, off_hs[(size_t)i]));
            P.si = reinterpret_cast<int32_t *>(at(B_SI, off_s4[(size_t)i])); P.ri = reinterpret_cast<double *>(at(B_RI, 2 * off_s4[(size_t)i]));
            P.ncc = reinterpret_cast<float *>(at(B_NCC, off_ncc[(size_t)i])); P.hes = reinterpret_cast<float *>(at(B_HES, off_s4[(size_t)i]));
            if (flags & 2u) { P.tmpa = reinterpret_cast<float *>(at(B_TMPA, off_s4[(size_t)i])); P.tmpb = reinterpret_cast<float *>(at(B_TMPB, off_s4[(size_t)i])); }
            P.partial = reinterpret_cast<double *>(at(B_PART, off_part[(size_t)i]));
            P.out5 = c.out5; P.ij3 = c.ij3;
        }
    }
    if ((e = hipMemcpy(W.buf[B_PARAMS], hp.data(), sizeof(LwParams) * (size_t)n, hipMemcpyHostToDevice)) != hipSuccess) return (int)e;
    if ((e = allow_lds(0)) != hipSuccess) return (int)e;
    const LwParams *dparams = static_cast<const LwParams *>(W.buf[B_PARAMS]);
    const int ngroups = (K + 15) / 16, lds = (kTH + s - 1) * lw_pitch(s);
    int first = 0;
    for (const int end : chunk_end) {
        const int G = end - first;
        const LwParams *Pv = dparams + first;
        int wh_max = 0, rh_max = 0, rw_max = 0;
        size_t np_max = 0;
        for (int i = first; i < end; ++i) {
            wh_max = std::max(wh_max, hp[(size_t)i].wh); rh_max = std::max(rh_max, hp[(size_t)i].rh); rw_max = std::max(rw_max, hp[(size_t)i].rw);
            np_max = std::max(np_max, (size_t)hp[(size_t)i].rh * hp[(size_t)i].rw);
        }
        const unsigned nblk = (unsigned)((np_max + 255) / 256);
        if ((e = hipMemsetAsync(W.buf[B_STATE], 0, sizeof(LwState) * (size_t)G, st)) != hipSuccess) return (int)e;
        if ((e = hipMemsetAsync(W.buf[B_OPA], 0, big.opa * (size_t)G, st)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL(lw_templates_kernel, dim3((unsigned)K, (unsigned)G), dim3(256), 0, st, Pv);
        hipLaunchKernelGGL(lw_rowsums_kernel, dim3((unsigned)((rw_max + 1023) / 1024), (unsigned)wh_max, (unsigned)G), dim3(256), 0, st, Pv);
        hipLaunchKernelGGL(lw_colsums_kernel, dim3((unsigned)((rw_max + 255) / 256), (unsigned)((rh_max + 31) / 32), (unsigned)G), dim3(256), 0, st, Pv);
        hipLaunchKernelGGL(lw_corr_kernel, dim3((unsigned)((rw_max + kTW - 1) / kTW), (unsigned)((rh_max + kTH - 1) / kTH), (unsigned)(ngroups * G)),
                           dim3(256), (size_t)lds, st, Pv, ngroups);
        hipLaunchKernelGGL(lw_pick_kernel, dim3((unsigned)G), dim3(64), 0, st, Pv);
        int hsrc = M_NCC;
        if (flags & 2u) {
            hipLaunchKernelGGL(lw_smooth_kernel, dim3(nblk, (unsigned)G), dim3(256), 0, st, Pv, (int)M_NCC, (int)M_TMPA, 0);
            hipLaunchKernelGGL(lw_smooth_kernel, dim3(nblk, (unsigned)G), dim3(256), 0, st, Pv, (int)M_TMPA, (int)M_TMPB, 1);
            hsrc = M_TMPB;
        }
        hipLaunchKernelGGL(lw_hessian_kernel, dim3(nblk, (unsigned)G), dim3(256), 0, st, Pv, hsrc);
        if (flags & 1u) {
            hipLaunchKernelGGL(lw_sums_final_kernel, dim3((unsigned)G), dim3(256), 0, st, Pv, 0);
            launch_select(Pv, G, np_max, M_HES, 0, st);
        }
        if (flags & 4u) {
            hipLaunchKernelGGL(lw_sums_kernel, dim3(nblk, (unsigned)G), dim3(256), 0, st, Pv, (int)M_NCC);
            hipLaunchKernelGGL(lw_sums_final_kernel, dim3((unsigned)G), dim3(256), 0, st, Pv, 1);
            launch_select(Pv, G, np_max, M_NCC, 1, st);
        }
        hipLaunchKernelGGL(lw_finish_kernel, dim3((unsigned)G), dim3(64), 0, st, Pv);
        first = end;
    }
    return (int)hipGetLastError();
}

int lw_run(const LargeCall &c, LwWorkspace &W, void *stream) { return lw_run_batch(&c, 1, W, stream); }

const float *lw_ncc_matrix(const LwWorkspace &W, int wh, int ww, int s, int k)
{
    return static_cast<const float *>(W.buf[B_NCC]) + (size_t)k * (size_t)(wh - s + 1) * (size_t)(ww - s + 1);
}

const uint8_t *lw_template(const LwWorkspace &W, int s, int k) { return static_cast<const uint8_t *>(W.buf[B_TMPL]) + (size_t)k * s * s; }

namespace {
// rot_order 2..5 in the batch path: the K templates of every point of a run, sampled from the prefiltered image 1 into global
// memory [n][K][s][s]; the one-workgroup-per-point kernels then LOAD their templates (sample_exact with Geo::pre) instead of
// sampling - the float64 coefficient patch of a point would not fit their LDS.  One workgroup per (point, angle).
__global__ __launch_bounds__(256) void lw_presample_kernel(const double *coef, long long rows, long long cols, const double *c1v, const double *r1v,
                                                           const double *rot, int K, int s, int order, uint8_t *pre)
{
    const long long pt = blockIdx.x / K;
    const int k = blockIdx.x - (int)(pt * K);
    const double c1 = c1v[pt], r1 = r1v[pt];
    uint8_t *T = pre + ((size_t)pt * K + k) * (size_t)s * s;
    for (int idx = threadIdx.x; idx < s * s; idx += 256) {
        const int i = idx / s, j = idx - i * s;
        T[idx] = (uint8_t)lw_sample_spline(coef, rows, cols, c1, r1, rot + 4 * k, i, j, order);
    }
}
const double kPoles[6][2] = {{0, 0}, {0, 0}, {-0.171572875253809902396622551581, 0}, {-0.267949192431122706472553658494, 0},
                             {-0.361341225900220177092212841325, -0.013725429297339121360331226939},
                             {-0.430575347099973791851434783493, -0.043096288203264653822712839920}};
}  // namespace

int lw_spline_prefilter(const uint8_t *d_img, int64_t rows, int64_t cols, int64_t stride, int order, double *buf0, double *buf1, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (order < 2 || order > 5 || rows < 1 || cols < 1) return (int)hipErrorInvalidValue;
    const int npoles = order / 2;
    double gain = 1.0;
    for (int p = 0; p < npoles; ++p) { const double z = kPoles[order][p]; gain *= (1.0 - z) * (1.0 - 1.0 / z); }
    // axis 0 (lines = columns: element stride = a row), then axis 1 (lines = rows); scipy skips an axis of length 1
    const double *cur = nullptr;                                       // null: the uint8 image is the source
    auto pass = [&](bool causal, long long nlines, long long n, long long s_line, long long s_elem, long long d_line, long long d_elem,
                    double z, double g, double *dst) {
        const double z_n_1 = pow(z, (double)(n - 1));
        const unsigned nb = (unsigned)((nlines + 63) / 64);
        if (!cur) {
            if (causal) hipLaunchKernelGGL((lw_spline_pass_kernel<uint8_t, true>), dim3(nb), dim3(64), 0, st, d_img, dst, nlines, n, s_line, s_elem, d_line, d_elem, z, z_n_1, g);
            else hipLaunchKernelGGL((lw_spline_pass_kernel<uint8_t, false>), dim3(nb), dim3(64), 0, st, d_img, dst, nlines, n, s_line, s_elem, d_line, d_elem, z, z_n_1, g);
        } else {
            if (causal) hipLaunchKernelGGL((lw_spline_pass_kernel<double, true>), dim3(nb), dim3(64), 0, st, cur, dst, nlines, n, s_line, s_elem, d_line, d_elem, z, z_n_1, g);
            else hipLaunchKernelGGL((lw_spline_pass_kernel<double, false>), dim3(nb), dim3(64), 0, st, cur, dst, nlines, n, s_line, s_elem, d_line, d_elem, z, z_n_1, g);
        }
        cur = dst;
    };
    bool first = true;
    for (int axis = 0; axis < 2; ++axis) {
        const long long n = axis == 0 ? rows : cols, nlines = axis == 0 ? cols : rows;
        if (n <= 1) continue;
        for (int p = 0; p < npoles; ++p) {
            const double z = kPoles[order][p];
            // causal: cur (or the image) -> buf0; anticausal: buf0 -> buf1
            const long long sl = cur ? (axis == 0 ? 1 : cols) : (axis == 0 ? 1 : stride), se = cur ? (axis == 0 ? cols : 1) : (axis == 0 ? stride : 1);
            const long long dl = axis == 0 ? 1 : cols, de = axis == 0 ? cols : 1;
            pass(true, nlines, n, sl, se, dl, de, z, (p == 0) ? gain : 1.0, buf0);      // (apply_filter multiplies the line by the gain first, on every axis)
            pass(false, nlines, n, dl, de, dl, de, z, 1.0, buf1);
            first = false;
        }
    }
    if (first) {                                                       // a 1 x 1 image: the coefficients are the pixels
        return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

int lw_presample(const double *d_coef, int64_t rows, int64_t cols, const double *d_c1, const double *d_r1, int64_t n, const double *d_rot, int K, int s,
                 int order, uint8_t *d_pre, void *stream)
{
    if (n <= 0) return 0;
    if ((int64_t)n * K > 0x7fffffffll) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(lw_presample_kernel, dim3((unsigned)(n * K)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_coef, (long long)rows, (long long)cols,
                       d_c1, d_r1, d_rot, K, s, order, d_pre);
    return (int)hipGetLastError();
}

int lw_write_nan(const int32_t *d_idx, int n, double *out, int32_t *out_ij, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(lw_nan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_idx, n, out, out_ij);
    return (int)hipGetLastError();
}

namespace {
// one parameter record on the device for the single-point calls below (a small allocation of its own, freed by the caller's
// synchronisation point: these entry points are synchronous adaptors)
struct OneParam {
    LwParams *d = nullptr;
    int put(const LwParams &P) { if (hipMalloc(reinterpret_cast<void **>(&d), sizeof(LwParams)) != hipSuccess) return -1; return (int)hipMemcpy(d, &P, sizeof P, hipMemcpyHostToDevice); }
    ~OneParam() { if (d) { (void)hipDeviceSynchronize(); (void)hipFree(d); } }
};
}  // namespace

int lw_get_template(const uint8_t *d_img, int64_t stride, int64_t row0, int64_t col0, int64_t nrows, int64_t ncols, int64_t rows, int64_t cols,
                    double c, double r, const double *d_rot4, int s, int order, uint8_t *d_out, void *stream, const double *d_coef)
{
    (void)nrows; (void)ncols;
    if (s < 1 || s > 4096 || order < 0 || order > 5 || (order >= 2 && !d_coef)) return (int)hipErrorInvalidValue;
    LwParams P;
    memset(&P, 0, sizeof P);
    P.img1 = d_img; P.rows1 = rows; P.cols1 = cols; P.stride1 = stride; P.row0 = row0; P.col0 = col0;
    P.s = s; P.K = 1; P.flags = (uint32_t)order << 3; P.c1 = c; P.r1 = r; P.rot = d_rot4; P.tmpl = d_out; P.coef = d_coef;
    OneParam one;
    if (int rc = one.put(P)) return rc;
    hipLaunchKernelGGL(lw_templates_kernel, dim3(1, 1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const LwParams *)one.d);
    return (int)hipGetLastError();                                     // (~OneParam waits for the kernel)
}

int lw_get_hessian(const float *d_ccm, int rh, int rw, uint32_t flags, const double gauss_w[5], float *d_hes, LwWorkspace &W, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (rh < 2 || rw < 2) return (int)hipErrorInvalidValue;
    const size_t np = (size_t)rh * rw, nblk = (np + 255) / 256;
    if (np >= 0xffffffffull) return (int)hipErrorInvalidValue;
    if (reserve(W, B_STATE, sizeof(LwState)) || reserve(W, B_PART, nblk * 32) ||
        ((flags & 2u) && (reserve(W, B_TMPA, np * 4) || reserve(W, B_TMPB, np * 4)))) return -1;
    LwParams P;
    memset(&P, 0, sizeof P);
    P.rh = rh; P.rw = rw; P.flags = flags; P.K = 1;
    for (int q = 0; q < 5; ++q) P.gw[q] = gauss_w[q];
    P.st = static_cast<LwState *>(W.buf[B_STATE]);
    P.ncc = const_cast<float *>(d_ccm);                                // (read only: M_NCC with kbest = 0)
    P.tmpa = static_cast<float *>(W.buf[B_TMPA]); P.tmpb = static_cast<float *>(W.buf[B_TMPB]);
    P.partial = static_cast<double *>(W.buf[B_PART]); P.hes = d_hes;
    hipError_t e = hipMemsetAsync(P.st, 0, sizeof(LwState), st);
    if (e != hipSuccess) return (int)e;
    OneParam one;
    if (int rc = one.put(P)) return rc;
    const LwParams *Pv = one.d;
    hipLaunchKernelGGL(lw_state_init_kernel, dim3(1), dim3(64), 0, st, Pv);
    int src = M_NCC;
    if (flags & 2u) {
        hipLaunchKernelGGL(lw_smooth_kernel, dim3((unsigned)nblk, 1), dim3(256), 0, st, Pv, (int)M_NCC, (int)M_TMPA, 0);
        hipLaunchKernelGGL(lw_smooth_kernel, dim3((unsigned)nblk, 1), dim3(256), 0, st, Pv, (int)M_TMPA, (int)M_TMPB, 1);
        src = M_TMPB;
    }
    hipLaunchKernelGGL(lw_hessian_kernel, dim3((unsigned)nblk, 1), dim3(256), 0, st, Pv, src);
    if (flags & 1u) {
        hipLaunchKernelGGL(lw_sums_final_kernel, dim3(1), dim3(256), 0, st, Pv, 0);
        launch_select(Pv, 1, np, M_HES, 0, st);
        hipLaunchKernelGGL(lw_normalise_kernel, dim3((unsigned)nblk, 1), dim3(256), 0, st, Pv);
    }
    return (int)hipGetLastError();                                     // (~OneParam waits for the kernels)
}

}  // namespace sid
