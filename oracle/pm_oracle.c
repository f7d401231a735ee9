/*
 * CPU oracle (plain C form) for the sea_ice_drift pattern-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / timed CPU baseline.  Nothing in
 * sea_ice_drift_amd/ links, loads or calls this file.
 *
 * Restates /root/reference/sea_ice_drift/pmlib.py (v0.7.1):
 *   sid_oracle_get_template    pmlib.py:89-115   (scipy affine_transform, order 0)
 *   sid_oracle_get_template1   pmlib.py:89-115   (the same with rot_order=1)
 *   sid_oracle_match_template  pmlib.py:156      (cv2.matchTemplate TM_CCOEFF_NORMED;
 *                                                 OpenCV un-vendored and absent:
 *                                                 PARITY UNPINNED at this call, see
 *                                                 oracle/pm_oracle.py for the spec)
 *   sid_oracle_hessian         pmlib.py:36-59    (np.gradient x2, hypot, median, std)
 *   sid_oracle_use_mcc         pmlib.py:117-212  (rotate_and_match + use_mcc)
 *   sid_oracle_pm_batch        pmlib.py:436-448  (the Pool.map seam; OpenMP over points
 *                                                 stands in for multiprocessing.Pool)
 *
 * Build: see oracle/Makefile (-ffp-contract=off: every double/float operation below
 * is one IEEE rounding, which is what the spec and the HIP kernel assume).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SID_FLAG_HES_NORM 1u
#define SID_FLAG_HES_SMTH 2u
#define SID_FLAG_MCC_NORM 4u
#define SID_FLAG_ROT_ORDER1 8u   /* rot_order=1 (pmlib.py:89,112-113): bilinear template sampling */
#define SID_ROT_ORDER(flags) (((flags) >> 3) & 7u)   /* rot_order 0..5 in flag bits 3..5 (include/sid_pm.h) */

/* ------------------------------------------------------------------ a1 */
/* rot = {cos a, sin a, tcT0, tcT1} with tcT = [tc,tc].dot([[cos,-sin],[sin,cos]])
 * (pmlib.py:105-110).  Returns the minimum sampled value (0 => invalid point). */
int sid_oracle_get_template(const uint8_t *img, int64_t rows, int64_t cols, int64_t stride,
                            double c, double r, const double *rot, int s, uint8_t *out)
{
    const double cosa = rot[0], sina = rot[1];
    const double off0 = r - rot[2], off1 = c - rot[3];
    const double msin = -sina;
    int vmin = 255;
    for (int i = 0; i < s; ++i) {
        for (int j = 0; j < s; ++j) {
            /* scipy NI_GeometricTransform: coordinate = 0.0; += i*M[h][0]; += j*M[h][1]; += shift */
            double rr = 0.0 + (double)i * cosa;
            rr = rr + (double)j * sina;
            rr = rr + off0;
            double cc = 0.0 + (double)i * msin;
            cc = cc + (double)j * cosa;
            cc = cc + off1;
            uint8_t v = 0;
            if (rr >= 0.0 && rr <= (double)(rows - 1) && cc >= 0.0 && cc <= (double)(cols - 1)) {
                int64_t ri = (int64_t)floor(rr + 0.5), ci = (int64_t)floor(cc + 0.5);
                v = img[ri * stride + ci];
            }
            out[i * s + j] = v;
            if (v < vmin) vmin = v;
        }
    }
    return vmin;
}

/* The same with rot_order=1: scipy NI_GeometricTransform at spline order 1, mode='constant', cval=0, uint8 output
 * (oracle/pm_oracle.py get_template_order1 has the derivation; fixture G1b pins both against the reference's own call). */
int sid_oracle_get_template1(const uint8_t *img, int64_t rows, int64_t cols, int64_t stride,
                             double c, double r, const double *rot, int s, uint8_t *out)
{
    const double cosa = rot[0], sina = rot[1];
    const double off0 = r - rot[2], off1 = c - rot[3];
    const double msin = -sina;
    int vmin = 255;
    for (int i = 0; i < s; ++i) {
        for (int j = 0; j < s; ++j) {
            double rr = 0.0 + (double)i * cosa;
            rr = rr + (double)j * sina;
            rr = rr + off0;
            double cc = 0.0 + (double)i * msin;
            cc = cc + (double)j * cosa;
            cc = cc + off1;
            uint8_t v = 0;
            if (rr >= 0.0 && rr <= (double)(rows - 1) && cc >= 0.0 && cc <= (double)(cols - 1)) {
                const double fr = floor(rr), fc = floor(cc);
                const double yr = rr - fr, yc = cc - fc;
                const double w0r = 1.0 - yr, w1r = yr, w0c = 1.0 - yc, w1c = yc;
                const int64_t r0 = (int64_t)fr, c0 = (int64_t)fc;
                int64_t r1 = r0 + 1, c1 = c0 + 1;                /* the tap behind the last sample is mirrored (its weight is 0) */
                if (r1 >= rows) r1 = 2 * rows - 2 - r1;
                if (c1 >= cols) c1 = 2 * cols - 2 - c1;
                if (r1 < 0) r1 = 0;
                if (c1 < 0) c1 = 0;
                double t = 0.0;
                t = t + ((double)img[r0 * stride + c0] * w0r) * w0c;
                t = t + ((double)img[r0 * stride + c1] * w0r) * w1c;
                t = t + ((double)img[r1 * stride + c0] * w1r) * w0c;
                t = t + ((double)img[r1 * stride + c1] * w1r) * w1c;
                t = t > 0.0 ? t + 0.5 : 0.0;                      /* CASE_INTERP_OUT_UINT of ni_interpolation.c */
                t = t > 255.0 ? 255.0 : t;
                v = (uint8_t)t;
            }
            out[i * s + j] = v;
            if (v < vmin) vmin = v;
        }
    }
    return vmin;
}

/* ------------------------------------------------------------------ a1, rot_order 2..5 */
/* scipy.ndimage.affine_transform(..., order = 2..5, mode='constant', cval=0, output=uint8, prefilter=True): the WHOLE image through
 * scipy's recursive B-spline prefilter (ni_splines.c apply_filter: gain, then per pole causal initialisation with the MIRROR
 * formulas - 'constant' takes those -, causal recursion, anticausal initialisation, anticausal recursion; axis 0 first), then the
 * tensor product of order + 1 weights per axis (get_spline_interpolation_weights) over the coefficient image.  The NumPy form
 * (oracle/pm_oracle.py) has the derivation and is pinned against scipy itself; this is the same arithmetic in C. */
static const double kSplinePoles[6][2] = {{0, 0}, {0, 0}, {-0.171572875253809902396622551581, 0}, {-0.267949192431122706472553658494, 0},
                                          {-0.361341225900220177092212841325, -0.013725429297339121360331226939},
                                          {-0.430575347099973791851434783493, -0.043096288203264653822712839920}};

static void spline_filter_line(double *c, int64_t n, int64_t stride, int order)
{
    if (n <= 1) return;
    const int npoles = order / 2;
    double gain = 1.0;
    for (int p = 0; p < npoles; ++p) { const double z = kSplinePoles[order][p]; gain *= (1.0 - z) * (1.0 - 1.0 / z); }
    for (int64_t i = 0; i < n; ++i) c[i * stride] *= gain;
    for (int p = 0; p < npoles; ++p) {
        const double z = kSplinePoles[order][p];
        const double z_n_1 = pow(z, (double)(n - 1));
        double z_i = z;
        double c0 = c[0] + z_n_1 * c[(n - 1) * stride];
        for (int64_t i = 1; i < n - 1; ++i) {
            if (z_i == 0.0) break;                                /* (the remaining terms are +-0) */
            c0 += z_i * (c[i * stride] + z_n_1 * c[(n - 1 - i) * stride]);
            z_i *= z;
        }
        c0 /= 1 - z_n_1 * z_n_1;
        c[0] = c0;
        for (int64_t i = 1; i < n; ++i) c[i * stride] += z * c[(i - 1) * stride];
        c[(n - 1) * stride] = (z * c[(n - 2) * stride] + c[(n - 1) * stride]) * z / (z * z - 1);
        for (int64_t i = n - 2; i >= 0; --i) c[i * stride] = z * (c[(i + 1) * stride] - c[i * stride]);
    }
}

/* spline_filter(img, order, output=float64): coef [rows][cols] */
int sid_oracle_spline_coefficients(const uint8_t *img, int64_t rows, int64_t cols, int64_t stride, int order, double *coef)
{
    if (order < 2 || order > 5) return -1;
    for (int64_t i = 0; i < rows; ++i) for (int64_t j = 0; j < cols; ++j) coef[i * cols + j] = (double)img[i * stride + j];
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < cols; ++j) spline_filter_line(coef + j, rows, cols, order);      /* axis 0 */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < rows; ++i) spline_filter_line(coef + i * cols, cols, 1, order);  /* axis 1 */
    return 0;
}

static void spline_weights(double x, int order, double *w)
{
    x -= floor(order & 1 ? x : x + 0.5);
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
    w[order] = 1.0;
    for (int i = 0; i < order; ++i) w[order] -= w[i];
}

static inline int64_t spline_mirror(int64_t idx, int64_t len)
{
    if (len <= 1) return 0;
    const int64_t s2 = 2 * len - 2;
    if (idx < 0) { idx = s2 * (-idx / s2) + idx; return idx <= 1 - len ? idx + s2 : -idx; }
    if (idx >= len) { idx -= s2 * (idx / s2); if (idx >= len) idx = s2 - idx; }
    return idx;
}

/* get_template(..., rot_order = order) from the prefiltered image coef [rows][cols]; returns the minimum sampled value */
int sid_oracle_get_template_spline(const double *coef, int64_t rows, int64_t cols, double c, double r, const double *rot, int s,
                                   int order, uint8_t *out)
{
    const double cosa = rot[0], sina = rot[1];
    const double off0 = r - rot[2], off1 = c - rot[3];
    const double msin = -sina;
    int vmin = 255;
    for (int i = 0; i < s; ++i) {
        for (int j = 0; j < s; ++j) {
            double rr = 0.0 + (double)i * cosa;
            rr = rr + (double)j * sina;
            rr = rr + off0;
            double cc = 0.0 + (double)i * msin;
            cc = cc + (double)j * cosa;
            cc = cc + off1;
            uint8_t v = 0;
            if (rr >= 0.0 && rr <= (double)(rows - 1) && cc >= 0.0 && cc <= (double)(cols - 1)) {
                const int64_t sr = (int64_t)floor(order & 1 ? rr : rr + 0.5) - order / 2, sc = (int64_t)floor(order & 1 ? cc : cc + 0.5) - order / 2;
                double wr[6], wc[6];
                spline_weights(rr, order, wr);
                spline_weights(cc, order, wc);
                double t = 0.0;
                for (int a = 0; a <= order; ++a) {
                    const int64_t ia = spline_mirror(sr + a, rows);
                    for (int b = 0; b <= order; ++b) {
                        const int64_t ib = spline_mirror(sc + b, cols);
                        t += (coef[ia * cols + ib] * wr[a]) * wc[b];
                    }
                }
                t = t > 0.0 ? t + 0.5 : 0.0;
                t = t > 255.0 ? 255.0 : t;
                v = (uint8_t)t;
            }
            out[i * s + j] = v;
            if (v < vmin) vmin = v;
        }
    }
    return vmin;
}

/* ------------------------------------------------------------------ a3 */
/* Exact integer sums + the double normalisation of the spec.  Scratch: sit (uint32 rh*rw: sum W*T <= 255^2 s^2 fits for s <= 257),
 * si/sii (int64 rh*rw).  out: float32 rh*rw. */
static void match_template_core(const uint8_t *win, int wh, int ww, int64_t wstride,
                                const uint8_t *tmpl, int s, float *out,
                                uint32_t *sit, int64_t *si, int64_t *sii, int have_sums)
{
    const int rh = wh - s + 1, rw = ww - s + 1;
    const int64_t n = (int64_t)s * s;
    if (!have_sums) {
        /* box sums by column-sum sliding (exact) */
        int64_t *cs = (int64_t *)malloc(sizeof(int64_t) * 2 * ww);
        int64_t *cs2 = cs + ww;
        for (int x = 0; x < ww; ++x) {
            int64_t a = 0, b = 0;
            for (int i = 0; i < s; ++i) { int64_t v = win[i * wstride + x]; a += v; b += v * v; }
            cs[x] = a; cs2[x] = b;
        }
        for (int y = 0; y < rh; ++y) {
            if (y > 0) {
                for (int x = 0; x < ww; ++x) {
                    int64_t o = win[(int64_t)(y - 1) * wstride + x], nn = win[(int64_t)(y + s - 1) * wstride + x];
                    cs[x] += nn - o; cs2[x] += nn * nn - o * o;
                }
            }
            int64_t a = 0, b = 0;
            for (int j = 0; j < s; ++j) { a += cs[j]; b += cs2[j]; }
            si[y * rw] = a; sii[y * rw] = b;
            for (int x = 1; x < rw; ++x) {
                a += cs[x + s - 1] - cs[x - 1]; b += cs2[x + s - 1] - cs2[x - 1];
                si[y * rw + x] = a; sii[y * rw + x] = b;
            }
        }
        free(cs);
    }
    int64_t s_t = 0, s_tt = 0;
    for (int k = 0; k < s * s; ++k) { int64_t v = tmpl[k]; s_t += v; s_tt += v * v; }
    const int64_t d_t = n * s_tt - s_t * s_t;
    if (d_t == 0) {
        for (int k = 0; k < rh * rw; ++k) out[k] = 1.0f;
        return;
    }
    memset(sit, 0, sizeof(uint32_t) * (size_t)rh * rw);
    for (int y = 0; y < rh; ++y) {
        uint32_t *acc = sit + (size_t)y * rw;
        for (int i = 0; i < s; ++i) {
            const uint8_t *wrow = win + (int64_t)(y + i) * wstride;
            const uint8_t *trow = tmpl + i * s;
            for (int j = 0; j < s; ++j) {
                const uint32_t t = trow[j];
                const uint8_t *w = wrow + j;
                for (int x = 0; x < rw; ++x) acc[x] += t * (uint32_t)w[x];
            }
        }
    }
    const double r_t = 1.0 / sqrt((double)d_t);
    for (int k = 0; k < rh * rw; ++k) {
        const int64_t numer = n * (int64_t)sit[k] - si[k] * s_t;
        const int64_t d_i = n * sii[k] - si[k] * si[k];
        float res;
        if (2 * d_i <= n && d_i * (int64_t)(1 << 23) <= 10 * n * sii[k]) {
            res = 0.0f;
        } else {
            const double r_i = 1.0 / sqrt((double)d_i);
            double q = (double)numer * r_i;
            q = q * r_t;
            const double aq = fabs(q);
            if (aq < 1.0) res = (float)q;
            else if (aq < 1.125) res = q > 0 ? 1.0f : -1.0f;
            else res = 0.0f;
        }
        out[k] = res;
    }
}

int sid_oracle_match_template(const uint8_t *win, int wh, int ww, int64_t wstride,
                              const uint8_t *tmpl, int s, float *out)
{
    const int rh = wh - s + 1, rw = ww - s + 1;
    if (rh < 1 || rw < 1) return -1;
    uint32_t *sit = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)rh * rw);
    int64_t *si = (int64_t *)malloc(sizeof(int64_t) * 2 * rh * rw);
    match_template_core(win, wh, ww, wstride, tmpl, s, out, sit, si, si + (size_t)rh * rw, 0);
    free(sit); free(si);
    return 0;
}

/* ------------------------------------------------------------------ a5 */
/* NumPy's float32 pairwise summation (numpy/_core/src/umath/loops_utils.h.src), so that
 * std() below rounds like np.std on a contiguous float32 array. */
static float pairwise_sum_f32(const float *a, int64_t n)
{
    if (n < 8) {
        float res = 0.f;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        int64_t i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
    }
}

/* np.add.reduce over a contiguous float32 array: the ufunc machinery feeds the inner loop
 * in chunks of the default buffer size (8192 elements) and accumulates the chunk results
 * sequentially (verified against numpy 2.2 on random lengths). */
static float numpy_sum_f32(const float *a, int64_t n)
{
    float res = 0.f;
    for (int64_t i = 0; i < n; i += 8192) {
        const int64_t m = n - i < 8192 ? n - i : 8192;
        res += pairwise_sum_f32(a + i, m);
    }
    return res;
}

static int cmp_f32(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* np.median and np.std of a float32 array (float32 results). tmp: n floats of scratch. */
static void median_std_f32(const float *v, int64_t n, float *tmp, float *med, float *sd)
{
    memcpy(tmp, v, sizeof(float) * n);
    qsort(tmp, n, sizeof(float), cmp_f32);
    if (n & 1) *med = tmp[n / 2];
    else *med = (tmp[n / 2 - 1] + tmp[n / 2]) / 2.0f;     /* np.mean of the two middles, float32 */
    const float mean = numpy_sum_f32(v, n) / (float)n;
    for (int64_t k = 0; k < n; ++k) { const float x = v[k] - mean; tmp[k] = x * x; }
    const float var = numpy_sum_f32(tmp, n) / (float)n;
    *sd = sqrtf(var);
}

static inline float grad1(const float *f, int64_t stride, int k, int n)
{
    if (k == 0) return f[stride] - f[0];
    if (k == n - 1) return f[(int64_t)(n - 1) * stride] - f[(int64_t)(n - 2) * stride];
    return (f[(int64_t)(k + 1) * stride] - f[(int64_t)(k - 1) * stride]) / 2.0f;
}

/* second application of the same 1-D gradient, evaluated at index k */
static inline float grad2(const float *f, int64_t stride, int k, int n)
{
    if (k == 0) return grad1(f, stride, 1, n) - grad1(f, stride, 0, n);
    if (k == n - 1) return grad1(f, stride, n - 1, n) - grad1(f, stride, n - 2, n);
    return (grad1(f, stride, k + 1, n) - grad1(f, stride, k - 1, n)) / 2.0f;
}

/* raw Hessian magnitude, float32 (pmlib.py:51-55) */
static void raw_hessian(const float *ccm, int rh, int rw, float *hes)
{
    for (int y = 0; y < rh; ++y)
        for (int x = 0; x < rw; ++x) {
            const float d2x = grad2(ccm + (int64_t)y * rw, 1, x, rw);
            const float d2y = grad2(ccm + x, rw, y, rh);
            hes[y * rw + x] = hypotf(d2x, d2y);
        }
}

/* scipy.ndimage.gaussian_filter(ccm, 1) on float32: per axis a radius-4 kernel in double,
 * 'reflect' boundary, result cast to float32 after each axis (pmlib.py:46-47). */
static void gaussian_smooth_sigma1(const float *in, int rh, int rw, float *out, float *tmp)
{
    double w[9];
    for (int k = -4; k <= 4; ++k) w[k + 4] = exp(-0.5 * (double)(k * k));
    /* phi.sum() in NumPy's pairwise order for n = 9 */
    const double sum = (((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]))) + w[8];
    for (int k = 0; k < 9; ++k) w[k] /= sum;
    /* scipy filters axis 0 first ... the kernel is symmetric so correlate == convolve */
    for (int pass = 0; pass < 2; ++pass) {
        const float *src = pass == 0 ? in : tmp;
        float *dst = pass == 0 ? tmp : out;
        const int n = pass == 0 ? rh : rw;            /* length along the filtered axis */
        for (int y = 0; y < rh; ++y)
            for (int x = 0; x < rw; ++x) {
                const int p = pass == 0 ? y : x;
                double acc = 0.0;
                /* scipy's correlate1d for symmetric kernels: centre + pairs (left+right)*w */
                {
                    int q = p;
                    acc = (double)src[pass == 0 ? (int64_t)q * rw + x : (int64_t)y * rw + q] * w[4];
                }
                for (int k = 4; k >= 1; --k) {               /* scipy adds the outermost pair first */
                    int ql = p - k, qr = p + k;
                    /* reflect: (d c b a | a b c d | d c b a) */
                    while (ql < 0 || ql >= n) { if (ql < 0) ql = -ql - 1; if (ql >= n) ql = 2 * n - 1 - ql; }
                    while (qr < 0 || qr >= n) { if (qr < 0) qr = -qr - 1; if (qr >= n) qr = 2 * n - 1 - qr; }
                    const double l = src[pass == 0 ? (int64_t)ql * rw + x : (int64_t)y * rw + ql];
                    const double r = src[pass == 0 ? (int64_t)qr * rw + x : (int64_t)y * rw + qr];
                    acc += (l + r) * w[4 - k];
                }
                dst[(int64_t)y * rw + x] = (float)acc;
            }
    }
}

/* get_hessian(ccm)[iy, ix] (pmlib.py:36-59, :167).  scratch: 3*rh*rw floats. */
float sid_oracle_hessian_at(const float *ccm, int rh, int rw, unsigned flags, int iy, int ix,
                            float *scratch)
{
    const int64_t n = (int64_t)rh * rw;
    float *hes = scratch, *tmp = scratch + n, *smth = scratch + 2 * n;
    const float *src = ccm;
    if (flags & SID_FLAG_HES_SMTH) { gaussian_smooth_sigma1(ccm, rh, rw, smth, tmp); src = smth; }
    raw_hessian(src, rh, rw, hes);
    float h = hes[(int64_t)iy * rw + ix];
    if (flags & SID_FLAG_HES_NORM) {
        float med, sd;
        median_std_f32(hes, n, tmp, &med, &sd);
        h = (h - med) / sd;
    }
    return h;
}

/* whole-matrix form for the unit tests */
int sid_oracle_hessian(const float *ccm, int rh, int rw, unsigned flags, float *out)
{
    const int64_t n = (int64_t)rh * rw;
    float *scratch = (float *)malloc(sizeof(float) * 3 * n);
    float *tmp = scratch + n, *smth = scratch + 2 * n;
    const float *src = ccm;
    if (flags & SID_FLAG_HES_SMTH) { gaussian_smooth_sigma1(ccm, rh, rw, smth, tmp); src = smth; }
    raw_hessian(src, rh, rw, out);
    if (flags & SID_FLAG_HES_NORM) {
        float med, sd;
        median_std_f32(out, n, tmp, &med, &sd);
        for (int64_t k = 0; k < n; ++k) out[k] = (out[k] - med) / sd;
    }
    free(scratch);
    return 0;
}

/* -------------------------------------------------------- a2, a4, a6 */
typedef struct {
    const double *coef;             /* rot_order 2..5: the prefiltered image 1 (owned by the caller of the batch) */
    uint8_t *tmpl, *tmpl_best;
    float *res, *best, *scratch;
    uint32_t *sit;
    int64_t *si;
    size_t cap;                     /* capacity in placements */
    int s;
} sid_ws;

static void ws_reserve(sid_ws *w, size_t nplace, int s)
{
    if (w->cap >= nplace && w->s >= s) return;
    free(w->tmpl); free(w->tmpl_best); free(w->res); free(w->best); free(w->scratch); free(w->sit); free(w->si);
    w->tmpl = (uint8_t *)malloc((size_t)s * s);
    w->tmpl_best = (uint8_t *)malloc((size_t)s * s);
    w->res = (float *)malloc(sizeof(float) * nplace);
    w->best = (float *)malloc(sizeof(float) * nplace);
    w->scratch = (float *)malloc(sizeof(float) * 3 * nplace);
    w->sit = (uint32_t *)malloc(sizeof(uint32_t) * nplace);
    w->si = (int64_t *)malloc(sizeof(int64_t) * 2 * nplace);
    w->cap = nplace; w->s = s;
}

static void ws_free(sid_ws *w)
{
    free(w->tmpl); free(w->tmpl_best); free(w->res); free(w->best); free(w->scratch); free(w->sit); free(w->si);
    memset(w, 0, sizeof(*w));
}

/* rotate_and_match (pmlib.py:117-174) on a search window of ANY rectangular shape: win = image2[0][0], wh x ww, wstride.
 * Returns -1 (NaN x 7 in the reference: a zero pixel in a template, pmlib.py:152-154) or the winning angle's index;
 * ddrc = {dc, dr} (pmlib.py:168-169), *rr = best_r (after mcc_norm), *hh = best_h, iyx = peak row / col; the winning NCC
 * matrix stays in w->best and the winning template in w->tmpl_best. */
static int rotate_and_match_ws(sid_ws *w, const uint8_t *img1, int64_t rows1, int64_t cols1, int64_t stride1,
                               const uint8_t *win, int wh, int ww, int64_t wstride, double c1, double r1,
                               int s, const double *rot, int n_angles, unsigned flags,
                               double *ddrc, float *rr_out, float *hh_out, int *iyx, float *gap)
{
    const int rh = wh - s + 1, rw = ww - s + 1;
    ws_reserve(w, (size_t)rh * rw, s);
    float best_r = -INFINITY;
    float top1 = -INFINITY, top2 = -INFINITY;                   /* two largest values over all angles and placements */
    int best_k = -1; int64_t best_idx = -1;
    for (int k = 0; k < n_angles; ++k) {
        const int order = (int)SID_ROT_ORDER(flags);
        const int vmin = order >= 2 ? sid_oracle_get_template_spline(w->coef, rows1, cols1, c1, r1, rot + 4 * k, s, order, w->tmpl)
                                    : (order == 1 ? sid_oracle_get_template1 : sid_oracle_get_template)(img1, rows1, cols1, stride1, c1, r1, rot + 4 * k, s, w->tmpl);
        if (vmin == 0) return -1;                               /* pmlib.py:152-154 -> NaN */
        match_template_core(win, wh, ww, wstride, w->tmpl, s, w->res, w->sit, w->si,
                            w->si + (size_t)rh * rw, k > 0);
        int64_t idx = 0; float mx = w->res[0];
        for (int64_t p = 1; p < (int64_t)rh * rw; ++p) if (w->res[p] > mx) { mx = w->res[p]; idx = p; }  /* first max */
        if (gap)
            for (int64_t p = 0; p < (int64_t)rh * rw; ++p) {
                const float v = w->res[p];
                if (v > top1) { top2 = top1; top1 = v; } else if (v > top2) top2 = v;
            }
        if (mx > best_r) {                                      /* strict (pmlib.py:160) */
            best_r = mx; best_k = k; best_idx = idx;
            float *t = w->best; w->best = w->res; w->res = t;
            memcpy(w->tmpl_best, w->tmpl, (size_t)s * s);
        }
    }
    if (best_k < 0) return -1;
    const int iy = (int)(best_idx / rw), ix = (int)(best_idx % rw);
    *hh_out = sid_oracle_hessian_at(w->best, rh, rw, flags, iy, ix, w->scratch);
    ddrc[1] = iy - (wh - s) / 2.; ddrc[0] = ix - (ww - s) / 2.;
    float rr = best_r;
    if (flags & SID_FLAG_MCC_NORM) {
        float med, sd;
        median_std_f32(w->best, (int64_t)rh * rw, w->scratch, &med, &sd);
        rr = (best_r - med) / sd;
    }
    *rr_out = rr;
    iyx[0] = iy; iyx[1] = ix;
    /* distance of the peak to the runner-up anywhere in the (angle, row, col) volume: a float32/DFT matcher
     * (cv2) could pick the other one when this is below its noise (~1e-6) */
    if (gap) *gap = top1 - top2;
    return best_k;
}

/* One grid point: use_mcc + rotate_and_match.  out5 = c2, r2, a, r, h ; ij3 = row, col, angle idx */
static void use_mcc_ws(sid_ws *w, const uint8_t *img1, int64_t rows1, int64_t cols1, int64_t stride1,
                       const uint8_t *img2, int64_t rows2, int64_t cols2, int64_t stride2,
                       double c1, double r1, double c2fg, double r2fg, double border,
                       int s, const double *angles, const double *rot, int n_angles,
                       unsigned flags, double *out5, int32_t *ij3, float *gap)
{
    const int hws = (int)((double)s / 2.);
    /* Python int(): truncation toward zero (pmlib.py:201-202) */
    const int64_t r0 = (int64_t)(r2fg - hws - border), r1e = (int64_t)(r2fg + hws + border + 1);
    const int64_t c0 = (int64_t)(c2fg - hws - border), c1e = (int64_t)(c2fg + hws + border + 1);
    for (int k = 0; k < 5; ++k) out5[k] = NAN;
    if (ij3) { ij3[0] = ij3[1] = ij3[2] = -1; }
    if (gap) *gap = NAN;
    if (!(r0 >= 0 && c0 >= 0 && r1e <= rows2 && c1e <= cols2 && r1e - r0 >= s + 1 && c1e - c0 >= s + 1)) return;
    const int wh = (int)(r1e - r0), ww = (int)(c1e - c0);
    double ddrc[2]; float rr, hh; int iyx[2];
    const int best_k = rotate_and_match_ws(w, img1, rows1, cols1, stride1, img2 + r0 * stride2 + c0, wh, ww, stride2, c1, r1,
                                           s, rot, n_angles, flags, ddrc, &rr, &hh, iyx, gap);
    if (best_k < 0) { if (gap) *gap = NAN; return; }
    out5[0] = c2fg + ddrc[0]; out5[1] = r2fg + ddrc[1]; out5[2] = angles[best_k];
    out5[3] = (double)rr; out5[4] = (double)hh;
    if (ij3) { ij3[0] = iyx[0]; ij3[1] = iyx[1]; ij3[2] = best_k; }
}

/* rotate_and_match (pmlib.py:117-174) as a call of its own: the search window is the whole of `image2` (any rectangular
 * shape, e.g. the reference's tests.py:336-337).  out5 = dc, dr, best_a, best_r, best_h; ij3 = peak row, col, angle index;
 * ccm [rh*rw] and tmpl [s*s] may be NULL.  Returns 0, 1 when the point is NaN x 7 (ij3 = -1), -1 on a bad shape. */
int sid_oracle_rotate_and_match(const uint8_t *img1, int64_t rows1, int64_t cols1, int64_t stride1,
                                const uint8_t *image2, int64_t rows2, int64_t cols2, int64_t stride2,
                                double c1, double r1, int img_size, const double *angles, const double *rot, int n_angles,
                                unsigned flags, double *out5, int32_t *ij3, float *ccm, uint8_t *tmpl)
{
    const int s = img_size;
    if (n_angles < 1 || s < 2 || rows2 - s + 1 < 2 || cols2 - s + 1 < 2 || !rot || SID_ROT_ORDER(flags) > 5) return -1;
    sid_ws w; memset(&w, 0, sizeof(w));
    double *coef = NULL;
    if (SID_ROT_ORDER(flags) >= 2) {
        coef = (double *)malloc(sizeof(double) * (size_t)rows1 * (size_t)cols1);
        if (!coef || sid_oracle_spline_coefficients(img1, rows1, cols1, stride1, (int)SID_ROT_ORDER(flags), coef)) { free(coef); return -1; }
        w.coef = coef;
    }
    double ddrc[2]; float rr, hh; int iyx[2];
    for (int k = 0; k < 5; ++k) out5[k] = NAN;
    if (ij3) { ij3[0] = ij3[1] = ij3[2] = -1; }
    const int best_k = rotate_and_match_ws(&w, img1, rows1, cols1, stride1, image2, (int)rows2, (int)cols2, stride2, c1, r1,
                                           s, rot, n_angles, flags, ddrc, &rr, &hh, iyx, NULL);
    if (best_k >= 0) {
        out5[0] = ddrc[0]; out5[1] = ddrc[1]; out5[2] = angles[best_k]; out5[3] = (double)rr; out5[4] = (double)hh;
        if (ij3) { ij3[0] = iyx[0]; ij3[1] = iyx[1]; ij3[2] = best_k; }
        if (ccm) memcpy(ccm, w.best, sizeof(float) * (size_t)(rows2 - s + 1) * (size_t)(cols2 - s + 1));
        if (tmpl) memcpy(tmpl, w.tmpl_best, (size_t)s * s);
    }
    ws_free(&w);
    free(coef);
    return best_k >= 0 ? 0 : 1;
}

/* rot may be NULL: then cos/sin/tcT are derived here with libm (NumPy's own cos/sin can
 * differ from libm in the last bit; tests pass rot computed by NumPy as the reference does). */
static double *make_rot(const double *angles, int n_angles, double alpha0, int s, const double *rot_in)
{
    double *rot = (double *)malloc(sizeof(double) * 4 * n_angles);
    if (rot_in) { memcpy(rot, rot_in, sizeof(double) * 4 * n_angles); return rot; }
    const double tc = (double)((int)((double)s / 2.) + 1);
    for (int k = 0; k < n_angles; ++k) {
        const double a = (angles[k] - alpha0) * (M_PI / 180.0);
        const double ca = cos(a), sa = sin(a);
        rot[4 * k + 0] = ca; rot[4 * k + 1] = sa;
        rot[4 * k + 2] = tc * ca + tc * sa;
        rot[4 * k + 3] = tc * (-sa) + tc * ca;
    }
    return rot;
}

int sid_oracle_pm_batch_gap(const uint8_t *img1, int64_t rows1, int64_t cols1, int64_t stride1,
                            const uint8_t *img2, int64_t rows2, int64_t cols2, int64_t stride2,
                            const double *c1, const double *r1, const double *c2fg, const double *r2fg,
                            const double *border, int64_t n, int img_size, double alpha0,
                            const double *angles, const double *rot_in, int n_angles, unsigned flags,
                            int nthreads, double *out, int32_t *out_ij, float *gap);

int sid_oracle_pm_batch(const uint8_t *img1, int64_t rows1, int64_t cols1, int64_t stride1,
                        const uint8_t *img2, int64_t rows2, int64_t cols2, int64_t stride2,
                        const double *c1, const double *r1, const double *c2fg, const double *r2fg,
                        const double *border, int64_t n, int img_size, double alpha0,
                        const double *angles, const double *rot_in, int n_angles, unsigned flags,
                        int nthreads, double *out, int32_t *out_ij)
{
    return sid_oracle_pm_batch_gap(img1, rows1, cols1, stride1, img2, rows2, cols2, stride2, c1, r1, c2fg, r2fg, border, n,
                                   img_size, alpha0, angles, rot_in, n_angles, flags, nthreads, out, out_ij, NULL);
}

/* same, plus gap[n] = peak value minus the second-largest NCC value of the point (NULL: not computed) */
int sid_oracle_pm_batch_gap(const uint8_t *img1, int64_t rows1, int64_t cols1, int64_t stride1,
                            const uint8_t *img2, int64_t rows2, int64_t cols2, int64_t stride2,
                            const double *c1, const double *r1, const double *c2fg, const double *r2fg,
                            const double *border, int64_t n, int img_size, double alpha0,
                            const double *angles, const double *rot_in, int n_angles, unsigned flags,
                            int nthreads, double *out, int32_t *out_ij, float *gap)
{
    if (n_angles < 1 || img_size < 2 || n < 0 || SID_ROT_ORDER(flags) > 5) return -1;
    double *rot = make_rot(angles, n_angles, alpha0, img_size, rot_in);
    double *coef = NULL;
    if (SID_ROT_ORDER(flags) >= 2 && n > 0) {                       /* the whole image 1 through the spline prefilter, once per batch */
        coef = (double *)malloc(sizeof(double) * (size_t)rows1 * (size_t)cols1);
        if (!coef || sid_oracle_spline_coefficients(img1, rows1, cols1, stride1, (int)SID_ROT_ORDER(flags), coef)) { free(coef); free(rot); return -1; }
    }
#ifdef _OPENMP
    if (nthreads < 1) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        sid_ws w; memset(&w, 0, sizeof(w));
        w.coef = coef;
#pragma omp for schedule(dynamic, 4)
        for (int64_t i = 0; i < n; ++i)
            use_mcc_ws(&w, img1, rows1, cols1, stride1, img2, rows2, cols2, stride2,
                       c1[i], r1[i], c2fg[i], r2fg[i], border[i], img_size, angles, rot, n_angles,
                       flags, out + 5 * i, out_ij ? out_ij + 3 * i : NULL, gap ? gap + i : NULL);
        ws_free(&w);
    }
    free(rot);
    free(coef);
    return 0;
}

int sid_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
