// stage.hip - uint8 staging of a float32 SAR image on gfx950 (C ABI: include/sid_stage.h; replaces the two
// full-image passes of get_uint8_image, lib.py:27-59).  All kernels are HBM-bound streaming passes with 16-byte
// loads per lane:
//   order statistics: three-digit radix select (11 + 11 + 10 bits) on the monotone key of the float; the first pass
//                     also counts the non-NaN pixels (its histogram is shared by every rank), so percentiles cost
//                     3 reads of the image.  Per-workgroup LDS histograms; the first pass - where SAR backscatter
//                     piles into a handful of bins - aggregates equal digits across the wavefront before the atomic;
//   scale:            4 B read + 1 B written per pixel, float32 operation for operation as NumPy.
// A workspace handle (sid_stage_create) owns the device and pinned host buffers: no allocation per call.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <vector>

#include "../../include/sid_stage.h"
#include "../../include/sid_pm.h"

#define SID_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kThreads = 256;
thread_local char g_err[256] = "";
int fail(int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(SID_PM_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

// monotone key: a < b  <=>  key(a) < key(b) for all non-NaN floats (-0.0 sorts before +0.0, which NumPy's
// sort may order either way - they are equal as values, so every order statistic is the same value)
__device__ __forceinline__ uint32_t f2key(float f) { uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
inline float key2f(uint32_t k) { union { uint32_t u; float f; } c; c.u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; return c.f; }

constexpr int kMaxStates = 8;            // order statistics resolved together in one pass over the image
constexpr int kBins = 2048;              // 11-bit digits (the last digit has 10 bits)
struct HistStates { uint32_t prefix[kMaxStates]; int n; };

// One pixel into the workgroup's histograms.  AGG (first pass: one state, every pixel takes part, a few hot bins):
// lanes holding the digit of the first active lane are counted with one atomic, twice over, the rest go one by one.
template <bool AGG>
__device__ __forceinline__ void bin_pixel(float v, const HistStates &S, uint32_t mask, int shift, uint32_t nbins_m1,
                                          uint32_t (*h)[kBins])
{
    bool act = v == v;
    const uint32_t key = f2key(v), digit = (key >> shift) & nbins_m1;
    if (AGG) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const unsigned long long am = __ballot(act);
            if (am == 0ull) break;                                     // wavefront-uniform
            const int first = __ffsll((long long)am) - 1;
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)digit, first);
            const unsigned long long same = __ballot(act && digit == d0);
            if ((int)(threadIdx.x & 63) == first) atomicAdd(&h[0][d0], (uint32_t)__popcll(same));
            act = act && digit != d0;
        }
        if (act) atomicAdd(&h[0][digit], 1u);
    } else if (act) {
        const uint32_t pk = key & mask;
        for (int q = 0; q < S.n; ++q)
            if (pk == S.prefix[q]) atomicAdd(&h[q][digit], 1u);
    }
}

// Digit passes after the first: a pixel takes part only when its key matches one of NS prefixes - one in ~10 at best,
// mostly none - so the four pixels of a 16-byte word are tested without a branch (NS compile-time: the prefixes stay in
// scalar registers) and only a lane that holds a match goes on to the atomics.
template <int NS>
__device__ __forceinline__ void bin_word(const float4 v, const HistStates &S, uint32_t mask, int shift, uint32_t nbins_m1,
                                         uint32_t (*h)[kBins])
{
    const uint32_t k0 = f2key(v.x) & mask, k1 = f2key(v.y) & mask, k2 = f2key(v.z) & mask, k3 = f2key(v.w) & mask;
    bool any = false;
#pragma unroll
    for (int q = 0; q < NS; ++q) any |= (k0 == S.prefix[q]) | (k1 == S.prefix[q]) | (k2 == S.prefix[q]) | (k3 == S.prefix[q]);
    if (any) {
        bin_pixel<false>(v.x, S, mask, shift, nbins_m1, h); bin_pixel<false>(v.y, S, mask, shift, nbins_m1, h);
        bin_pixel<false>(v.z, S, mask, shift, nbins_m1, h); bin_pixel<false>(v.w, S, mask, shift, nbins_m1, h);
    }
}

// For every state q: histogram of digit (key >> shift) & (nbins - 1) over the non-NaN pixels whose key matches
// prefix[q] under `mask`.  hist: [n][kBins] (64-bit, global).  VEC: rows are 16-byte aligned and cols % 4 == 0.
// NS: number of states when known at compile time (contiguous vector path of the later passes), else 0
template <bool AGG, bool VEC, int NS = 0>
__global__ __launch_bounds__(kThreads) void hist_kernel(const float *img, int64_t rows, int64_t cols, int64_t stride,
                                                        HistStates S, uint32_t mask, int shift, int nbins,
                                                        unsigned long long *hist)
{
    extern __shared__ uint32_t h_raw[];
    uint32_t (*h)[kBins] = reinterpret_cast<uint32_t (*)[kBins]>(h_raw);
    for (int i = threadIdx.x; i < S.n * kBins; i += kThreads) h_raw[i] = 0;
    __syncthreads();
    const uint32_t nm1 = (uint32_t)nbins - 1u;
    const float nan = __int_as_float(0x7fc00000);
    if (VEC && stride == cols) {
        // contiguous image: one flat array of 16-byte words, kUnroll of them in flight per lane before any is binned
        // (every workgroup streams whole chunks: no row granularity, no tail of partly idle workgroups)
        constexpr int kUnroll = 4;
        const float4 *p4 = reinterpret_cast<const float4 *>(img);
        const int64_t n4 = (rows * cols) >> 2, step = (int64_t)gridDim.x * kThreads * kUnroll;
        const int64_t n4r = (n4 + step - 1) / step * step;                 // (all lanes stay in the loop: wavefront ballots)
        for (int64_t base = (int64_t)blockIdx.x * kThreads * kUnroll; base < n4r; base += step) {
            float4 v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t x = base + u * kThreads + threadIdx.x;
                v[u] = x < n4 ? p4[x] : make_float4(nan, nan, nan, nan);
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                if (NS > 0) { bin_word<NS>(v[u], S, mask, shift, nm1, h); continue; }
                bin_pixel<AGG>(v[u].x, S, mask, shift, nm1, h); bin_pixel<AGG>(v[u].y, S, mask, shift, nm1, h);
                bin_pixel<AGG>(v[u].z, S, mask, shift, nm1, h); bin_pixel<AGG>(v[u].w, S, mask, shift, nm1, h);
            }
        }
    } else
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float *row = img + r * stride;
        if (VEC) {
            const float4 *row4 = reinterpret_cast<const float4 *>(row);
            const int64_t n4 = cols >> 2;
            for (int64_t x = threadIdx.x; x < ((n4 + kThreads - 1) / kThreads) * kThreads; x += kThreads) {
                // (all lanes stay in the loop: the aggregation uses wavefront ballots)
                const bool in = x < n4;
                const float4 v = in ? row4[x] : make_float4(nan, nan, nan, nan);
                bin_pixel<AGG>(v.x, S, mask, shift, nm1, h); bin_pixel<AGG>(v.y, S, mask, shift, nm1, h);
                bin_pixel<AGG>(v.z, S, mask, shift, nm1, h); bin_pixel<AGG>(v.w, S, mask, shift, nm1, h);
            }
        } else {
            for (int64_t x = threadIdx.x; x < ((cols + kThreads - 1) / kThreads) * kThreads; x += kThreads)
                bin_pixel<AGG>(x < cols ? row[x] : nan, S, mask, shift, nm1, h);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S.n * kBins; i += kThreads)
        if (h_raw[i]) atomicAdd(&hist[i], (unsigned long long)h_raw[i]);
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void scale_kernel(const float *img, int64_t rows, int64_t cols, int64_t stride,
                                                         float vmin, float denom, uint8_t *out, int64_t out_stride)
{
    auto one = [&](float x) -> uint32_t {
        float t = x - vmin;                                  // lib.py:54, one float32 rounding per operation
        t = 254.0f * t;
        t = t / denom;
        t = 1.0f + t;
        t = t < 1.0f ? 1.0f : t;                             // lib.py:55-56 (NaN fails both comparisons)
        t = t > 255.0f ? 255.0f : t;
        const bool finite = fabsf(x) <= 3.402823466e38f;     // false for NaN and +-inf (lib.py:57)
        return (finite && t == t) ? (uint32_t)(uint8_t)t : 0u;
    };
    if (VEC && stride == cols && out_stride == cols) {       // contiguous in and out: flat, four 16-byte loads in flight per lane
        constexpr int kUnroll = 4;
        const float4 *p4 = reinterpret_cast<const float4 *>(img);
        uint32_t *o4 = reinterpret_cast<uint32_t *>(out);
        const int64_t n4 = (rows * cols) >> 2, step = (int64_t)gridDim.x * kThreads * kUnroll;
        for (int64_t base = (int64_t)blockIdx.x * kThreads * kUnroll; base < n4; base += step) {
            float4 v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t x = base + u * kThreads + threadIdx.x;
                v[u] = x < n4 ? p4[x] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t x = base + u * kThreads + threadIdx.x;
                if (x < n4) o4[x] = one(v[u].x) | (one(v[u].y) << 8) | (one(v[u].z) << 16) | (one(v[u].w) << 24);
            }
        }
        return;
    }
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float *row = img + r * stride;
        uint8_t *orow = out + r * out_stride;
        if (VEC) {                                           // 16 B in, 4 B out per lane
            const float4 *row4 = reinterpret_cast<const float4 *>(row);
            uint32_t *o4 = reinterpret_cast<uint32_t *>(orow);
            for (int64_t c = threadIdx.x; c < (cols >> 2); c += kThreads) {
                const float4 v = row4[c];
                o4[c] = one(v.x) | (one(v.y) << 8) | (one(v.z) << 16) | (one(v.w) << 24);
            }
        } else {
            for (int64_t c = threadIdx.x; c < cols; c += kThreads) orow[c] = (uint8_t)one(row[c]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Round 4: order statistics near HINTED fractions in two passes instead of three (sid_stage_begin_hint).
//   sample_kernel  one workgroup sorts 8192 pixels taken at hashed positions of a regular grid and brackets every hinted
//                  fraction f by the sample's order statistics 6 sigma of the sampling error below and above f: a key range
//                  [lo, hi] that holds the wanted ranks (if it ever does not, the counts say so and the three-pass radix
//                  select runs instead - the result is exact either way);
//   range_kernel   COUNT: one pass counts the non-NaN pixels, the pixels below each range and a 2048-bin histogram INSIDE
//                  each range (bin = (key - lo) >> shift, lo aligned to the bin width) - a few per cent of the pixels touch
//                  the LDS atomics, not the third of them that shares the first radix digit of a percentile of SAR backscatter;
//                  !COUNT: the same kernel resolves the chosen bin(s) to single keys (shift 0).
// ---------------------------------------------------------------------------------------------
constexpr int kSample = 8192, kSampleThreads = 1024, kMaxHint = 4;
constexpr int kHintBins = 2048;        // bins inside a hinted range (every workgroup flushes all of them: fewer bins, fewer global atomics)
constexpr int kRangeGrid = 1024;       // workgroups of a range pass: four per CU, each streaming 1 / 1024 of the image
constexpr int kSampleLds = (kSample + 2 * kMaxHint * kBins) * 4;   // keys + one histogram per bracket end
struct RangeStates { uint32_t lo[kMaxStates], span[kMaxStates], shift[kMaxStates]; uint32_t m_valid, n; };   // device memory
struct HintIn { double frac[kMaxHint]; int n; };

__global__ __launch_bounds__(kSampleThreads) void sample_kernel(const float *img, int64_t rows, int64_t cols, int64_t stride,
                                                                HintIn H, RangeStates *out)
{
    // dynamic LDS: [kSample] keys | [2 kMaxHint][kBins] histograms (one per wanted order statistic of the sample)
    extern __shared__ uint32_t s_raw[];
    uint32_t *keys = s_raw;
    uint32_t (*hist)[kBins] = reinterpret_cast<uint32_t (*)[kBins]>(s_raw + kSample);
    __shared__ uint32_t nvalid, t_prefix[2 * kMaxHint], t_rank[2 * kMaxHint], t_open[2 * kMaxHint];
    constexpr int kPer = kSample / kSampleThreads;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t npix = rows * cols;
    const int64_t step = npix / kSample;                               // (0: fewer pixels than samples - every pixel once)
    if (tid == 0) nvalid = 0;
    for (int i = tid; i < 2 * kMaxHint * kBins; i += kSampleThreads) (&hist[0][0])[i] = 0;
    __syncthreads();
    uint32_t mykeys[kPer], mine = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = j * kSampleThreads + tid;
        uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const int64_t p = step > 0 ? (int64_t)i * step + (int64_t)(h % (uint32_t)(step > 0x7fffffff ? 0x7fffffff : step)) : (int64_t)i;
        uint32_t key = 0xffffffffu;                                    // (not the key of any non-NaN value)
        if (p < npix) {
            const int64_t r = p / cols, c = p - r * cols;
            const float v = img[r * stride + c];
            if (v == v) { key = f2key(v); ++mine; }
        }
        mykeys[j] = key; keys[i] = key;
    }
    atomicAdd(&nvalid, mine);
    __syncthreads();
    const int m = (int)nvalid, nt = 2 * H.n;
    // wanted order statistics of the sample: target 2 q = the lower bracket of fraction q, 2 q + 1 = the upper one
    if (tid < 2 * kMaxHint) {
        uint32_t open = 0, rank = 0;
        if (tid < nt && m >= 256) {
            const double f0 = H.frac[tid >> 1], f = f0 < 0.0 ? 0.0 : (f0 > 1.0 ? 1.0 : f0);
            const double idx = f * (double)(m - 1), dev = 6.0 * sqrt(f * (1.0 - f) * (double)m) + 3.0;
            const int a = (tid & 1) ? (int)ceil(idx + dev) : (int)floor(idx - dev);
            open = (a < 0 || a >= m) ? 1u : 0u;                        // the bracket reaches the end of the key space
            rank = open ? 0u : (uint32_t)a;
        }
        t_prefix[tid] = 0; t_rank[tid] = rank; t_open[tid] = open;
    }
    __syncthreads();
    // three-digit radix select of all targets at once on the keys in registers
    uint32_t mask = 0;
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0;
        const uint32_t nm1 = pass == 2 ? 1023u : 2047u;
        if (pass > 0) {
            for (int i = tid; i < nt * kBins; i += kSampleThreads) (&hist[0][0])[i] = 0;
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const uint32_t key = mykeys[j];
            if (key == 0xffffffffu) continue;
            if (pass == 0) atomicAdd(&hist[0][(key >> 21) & nm1], 1u);  // (no digit chosen yet: one histogram serves every target)
            else for (int q = 0; q < nt; ++q) if ((key & mask) == t_prefix[q]) atomicAdd(&hist[q][(key >> shift) & nm1], 1u);
        }
        __syncthreads();
        if (wv < nt) {                                                 // wavefront q: the bin that holds target q's rank
            const uint32_t *hq = hist[pass == 0 ? 0 : wv];
            uint32_t c[32], tot = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) { c[j] = (32 * lane + j) <= (int)nm1 ? hq[32 * lane + j] : 0u; tot += c[j]; }
            uint32_t inc = tot;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d); if (lane >= d) inc += o; }
            const uint32_t exc = inc - tot, r = t_rank[wv];
            if (r >= exc && r < inc) {                                 // exactly one lane (the rank is below the number of valid keys)
                uint32_t rr = r - exc; int b = 0;
#pragma unroll
                for (int j = 0; j < 32; ++j) { if (rr >= c[j] && b == j) { rr -= c[j]; b = j + 1; } }
                t_prefix[wv] |= (uint32_t)(32 * lane + b) << shift; t_rank[wv] = rr;
            }
        }
        mask |= nm1 << shift;
        __syncthreads();
    }
    if (tid < kMaxStates) {
        const int q = tid;
        uint32_t lo = 0xffffffffu, span = 0, shift = 0;                // (matches no valid key)
        if (q < H.n && m >= 256) {
            const uint32_t klo = t_open[2 * q] ? 0u : t_prefix[2 * q], khi = t_open[2 * q + 1] ? 0xfffffffeu : t_prefix[2 * q + 1];
            while (((khi >> shift) - (klo >> shift)) >= (uint32_t)kHintBins) ++shift;
            lo = (klo >> shift) << shift;
            span = khi - lo;
        }
        out->lo[q] = lo; out->span[q] = span; out->shift[q] = shift;
        if (q == 0) { out->m_valid = (uint32_t)m; out->n = (uint32_t)H.n; }
    }
}

// counts: [0] = non-NaN pixels, [1 + q] = non-NaN pixels whose key lies below range q (COUNT only)
template <int NQ, bool COUNT>
__global__ __launch_bounds__(kThreads) void range_kernel(const float *img, int64_t rows, int64_t cols, int64_t stride, bool vec,
                                                         const RangeStates *S, unsigned long long *hist, unsigned long long *counts)
{
    extern __shared__ uint32_t h_raw[];
    uint32_t (*h)[kBins] = reinterpret_cast<uint32_t (*)[kBins]>(h_raw);
    __shared__ uint32_t wg_counts[1 + kMaxStates];
    for (int i = threadIdx.x; i < NQ * kBins; i += kThreads) h_raw[i] = 0;
    if (threadIdx.x <= kMaxStates) wg_counts[threadIdx.x] = 0;
    uint32_t lo[NQ], span[NQ], shift[NQ];                              // uniform: scalar registers
#pragma unroll
    for (int q = 0; q < NQ; ++q) { lo[q] = S->lo[q]; span[q] = S->span[q]; shift[q] = S->shift[q]; }
    __syncthreads();
    uint32_t nval = 0, below[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) below[q] = 0;
    auto one = [&](float v) {
        const bool ok = v == v;
        const uint32_t key = f2key(v);
        if (COUNT) nval += ok ? 1u : 0u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint32_t d = key - lo[q];
            if (COUNT) below[q] += (ok && key < lo[q]) ? 1u : 0u;
            if (ok && d <= span[q]) atomicAdd(&h[q][d >> shift[q]], 1u);
        }
    };
    if (vec && stride == cols) {
        constexpr int kUnroll = 4;
        const float4 *p4 = reinterpret_cast<const float4 *>(img);
        const float nan = __int_as_float(0x7fc00000);
        const int64_t n4 = (rows * cols) >> 2, step = (int64_t)gridDim.x * kThreads * kUnroll;
        for (int64_t base = (int64_t)blockIdx.x * kThreads * kUnroll; base < n4; base += step) {
            float4 v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t x = base + u * kThreads + threadIdx.x;
                v[u] = x < n4 ? p4[x] : make_float4(nan, nan, nan, nan);
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) { one(v[u].x); one(v[u].y); one(v[u].z); one(v[u].w); }
        }
    } else {
        for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
            const float *row = img + r * stride;
            for (int64_t x = threadIdx.x; x < cols; x += kThreads) one(row[x]);
        }
    }
    if (COUNT) {
        atomicAdd(&wg_counts[0], nval);
#pragma unroll
        for (int q = 0; q < NQ; ++q) atomicAdd(&wg_counts[1 + q], below[q]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NQ * kBins; i += kThreads)
        if (h_raw[i]) atomicAdd(&hist[i], (unsigned long long)h_raw[i]);
    if (COUNT && threadIdx.x <= NQ && wg_counts[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)wg_counts[threadIdx.x]);
}

// one workgroup per row up to a few waves of the chip; narrow images (few columns) still fill it through rows
int grid_rows(int64_t rows) { return (int)std::max<int64_t>(1, std::min<int64_t>(rows, 256 * 16)); }

int check_img(const void *p, int64_t rows, int64_t cols, int64_t stride)
{
    if (!p) return fail(SID_PM_ERR_ARG, "null image pointer");
    if (rows < 1 || cols < 1 || stride < cols) return fail(SID_PM_ERR_ARG, "bad image shape/stride");
    return SID_PM_OK;
}

bool vec_ok(const void *p, int64_t cols, int64_t stride)
{
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (cols & 3) == 0 && (stride & 3) == 0;
}

}  // namespace

struct sid_stage_ws {
    int device = 0;
    unsigned long long *d_hist = nullptr;      // [kMaxStates][kBins]
    unsigned long long *h_hist = nullptr;      // pinned host copy
    unsigned long long first[kBins];           // first-digit histogram of the image of the last sid_stage_begin
    bool have_first = false;                   // (sid_stage_begin_hint leaves it to the first rank that needs the radix route)
    // hinted ranges (sid_stage_begin_hint): device states + counters, pinned host copies, the host's view of the last image
    RangeStates *d_states = nullptr, *h_states = nullptr;
    unsigned long long *d_counts = nullptr, *h_counts = nullptr;     // [1 + kMaxStates]
    int n_hint = 0;
    uint32_t hint_lo[kMaxHint], hint_shift[kMaxHint];
    unsigned long long hint_below[kMaxHint], hint_inside[kMaxHint];
    std::vector<unsigned long long> hint_hist;                       // [n_hint][kBins]
    unsigned long long n_valid = 0;
    const float *img = nullptr; int64_t rows = 0, cols = 0, stride = 0;
    hipStream_t stream = nullptr;
    bool have = false;
};

namespace {

// the workspace's buffers (and the images handed to it) live on ws->device: every entry point runs there, whatever the
// calling thread's current device is, and puts the previous device back
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int run_hist(sid_stage_ws *ws, const HistStates &S, uint32_t mask, int shift, int nbins, bool agg)
{
    hipStream_t st = ws->stream;
    const size_t bytes = (size_t)S.n * kBins * sizeof(unsigned long long);
    hipError_t e = hipMemsetAsync(ws->d_hist, 0, bytes, st);
    if (e == hipSuccess) {
        const dim3 grid((unsigned)grid_rows(ws->rows)), block(kThreads);
        const size_t lds = (size_t)S.n * kBins * sizeof(uint32_t);
        const bool vec = vec_ok(ws->img, ws->cols, ws->stride);
        if (agg && vec) hipLaunchKernelGGL((hist_kernel<true, true>), grid, block, lds, st, ws->img, ws->rows, ws->cols, ws->stride, S, mask, shift, nbins, ws->d_hist);
        else if (agg) hipLaunchKernelGGL((hist_kernel<true, false>), grid, block, lds, st, ws->img, ws->rows, ws->cols, ws->stride, S, mask, shift, nbins, ws->d_hist);
        else if (vec && ws->stride == ws->cols && S.n == 1) hipLaunchKernelGGL((hist_kernel<false, true, 1>), grid, block, lds, st, ws->img, ws->rows, ws->cols, ws->stride, S, mask, shift, nbins, ws->d_hist);
        else if (vec && ws->stride == ws->cols && S.n == 2) hipLaunchKernelGGL((hist_kernel<false, true, 2>), grid, block, lds, st, ws->img, ws->rows, ws->cols, ws->stride, S, mask, shift, nbins, ws->d_hist);
        else if (vec && ws->stride == ws->cols && S.n <= 4) hipLaunchKernelGGL((hist_kernel<false, true, 4>), grid, block, lds, st, ws->img, ws->rows, ws->cols, ws->stride, S, mask, shift, nbins, ws->d_hist);
        else if (vec) hipLaunchKernelGGL((hist_kernel<false, true>), grid, block, lds, st, ws->img, ws->rows, ws->cols, ws->stride, S, mask, shift, nbins, ws->d_hist);
        else hipLaunchKernelGGL((hist_kernel<false, false>), grid, block, lds, st, ws->img, ws->rows, ws->cols, ws->stride, S, mask, shift, nbins, ws->d_hist);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ws->h_hist, ws->d_hist, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "histogram pass: %s", hipGetErrorString(e));
    return SID_PM_OK;
}

int radix_select(sid_stage_ws *ws, const int64_t *ranks, int n_ranks, float *values);

// first radix digit of every non-NaN pixel (shared by every rank that takes the three-pass route) and their number
int first_digit_pass(sid_stage_ws *ws)
{
    HistStates S; S.n = 1; S.prefix[0] = 0;
    if (int rc = run_hist(ws, S, 0u, 21, kBins, true)) return rc;
    unsigned long long n = 0;
    for (int b = 0; b < kBins; ++b) { ws->first[b] = ws->h_hist[b]; n += ws->h_hist[b]; }
    ws->n_valid = n; ws->have_first = true;
    return SID_PM_OK;
}

// range_kernel over the workspace's image with the states in ws->d_states (count: the hinted first pass; else a resolving
// pass, states uploaded by the caller); histograms (and counters) come back in the pinned copies; synchronises unless count
int run_range(sid_stage_ws *ws, int n_states, bool count)
{
    hipStream_t st = ws->stream;
    const int nq = n_states <= 1 ? 1 : n_states <= 2 ? 2 : n_states <= 4 ? 4 : 8;
    const size_t bytes = (size_t)nq * kBins * sizeof(unsigned long long);
    HIP_TRY(hipMemsetAsync(ws->d_hist, 0, bytes, st));
    if (count) HIP_TRY(hipMemsetAsync(ws->d_counts, 0, (1 + kMaxStates) * sizeof(unsigned long long), st));
    const dim3 grid((unsigned)std::min(grid_rows(ws->rows), kRangeGrid)), block(kThreads);
    const size_t lds = (size_t)nq * kBins * sizeof(uint32_t);
    const bool vec = vec_ok(ws->img, ws->cols, ws->stride);
#define SID_RANGE(NQ, C) hipLaunchKernelGGL((range_kernel<NQ, C>), grid, block, lds, st, ws->img, ws->rows, ws->cols, ws->stride, vec, ws->d_states, ws->d_hist, ws->d_counts)
    if (count) { if (nq == 1) SID_RANGE(1, true); else if (nq == 2) SID_RANGE(2, true); else SID_RANGE(4, true); }
    else { if (nq == 1) SID_RANGE(1, false); else if (nq == 2) SID_RANGE(2, false); else if (nq == 4) SID_RANGE(4, false); else SID_RANGE(8, false); }
#undef SID_RANGE
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ws->h_hist, ws->d_hist, (size_t)n_states * kBins * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    if (count) HIP_TRY(hipMemcpyAsync(ws->h_counts, ws->d_counts, (1 + kMaxStates) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    else HIP_TRY(hipStreamSynchronize(st));
    return SID_PM_OK;
}

}  // namespace

SID_EXPORT const char *sid_stage_last_error(void) { return g_err; }

SID_EXPORT int sid_stage_create(int device, sid_stage_ws **out)
{
    if (!out) return fail(SID_PM_ERR_ARG, "null workspace pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(SID_PM_ERR_NODEVICE, "no such HIP device");
    int prev = 0; (void)hipGetDevice(&prev); (void)hipSetDevice(device);
    sid_stage_ws *ws = new (std::nothrow) sid_stage_ws();
    hipError_t e = ws ? hipSuccess : hipErrorOutOfMemory;
    if (e == hipSuccess) { ws->device = device; e = hipMalloc(&ws->d_hist, (size_t)kMaxStates * kBins * sizeof(unsigned long long)); }
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&ws->h_hist), (size_t)kMaxStates * kBins * sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ws->d_states), sizeof(RangeStates));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&ws->h_states), sizeof(RangeStates), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ws->d_counts), (1 + kMaxStates) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&ws->h_counts), (1 + kMaxStates) * sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(range_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxStates * kBins * 4);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kSampleLds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(hist_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxStates * kBins * 4);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(hist_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxStates * kBins * 4);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) { sid_stage_destroy(ws); return fail(SID_PM_ERR_HIP, "workspace: %s", hipGetErrorString(e)); }
    *out = ws;
    return SID_PM_OK;
}

SID_EXPORT void sid_stage_destroy(sid_stage_ws *ws)
{
    if (!ws) return;
    DeviceGuard guard(ws->device);
    (void)hipFree(ws->d_hist); (void)hipFree(ws->d_states); (void)hipFree(ws->d_counts);
    if (ws->h_hist) (void)hipHostFree(ws->h_hist);
    if (ws->h_states) (void)hipHostFree(ws->h_states);
    if (ws->h_counts) (void)hipHostFree(ws->h_counts);
    delete ws;
}

SID_EXPORT int sid_stage_begin(sid_stage_ws *ws, const float *d_img, int64_t rows, int64_t cols, int64_t stride, int64_t *n_valid,
                               void *hip_stream)
{
    if (!ws) return fail(SID_PM_ERR_ARG, "null workspace");
    if (int rc = check_img(d_img, rows, cols, stride)) return rc;
    if (!n_valid) return fail(SID_PM_ERR_ARG, "null output");
    DeviceGuard guard(ws->device);
    {   // an image on another device than the workspace's is a caller error, not something to launch on
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, d_img) == hipSuccess && at.type == hipMemoryTypeDevice && at.device != ws->device)
            return fail(SID_PM_ERR_ARG, "image lives on device %d, the workspace on device %d", at.device, ws->device);
    }
    ws->img = d_img; ws->rows = rows; ws->cols = cols; ws->stride = stride; ws->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ws->have = false; ws->n_hint = 0;
    if (int rc = first_digit_pass(ws)) return rc;
    ws->have = true;
    *n_valid = (int64_t)ws->n_valid;
    return SID_PM_OK;
}

SID_EXPORT int sid_stage_begin_hint(sid_stage_ws *ws, const float *d_img, int64_t rows, int64_t cols, int64_t stride,
                                    const double *fractions, int n_fractions, int64_t *n_valid, void *hip_stream)
{
    if (n_fractions < 1 || n_fractions > kMaxHint || !fractions || getenv("SID_STAGE_NO_HINT") != nullptr)
        return sid_stage_begin(ws, d_img, rows, cols, stride, n_valid, hip_stream);
    if (!ws) return fail(SID_PM_ERR_ARG, "null workspace");
    if (int rc = check_img(d_img, rows, cols, stride)) return rc;
    if (!n_valid) return fail(SID_PM_ERR_ARG, "null output");
    DeviceGuard guard(ws->device);
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, d_img) == hipSuccess && at.type == hipMemoryTypeDevice && at.device != ws->device)
            return fail(SID_PM_ERR_ARG, "image lives on device %d, the workspace on device %d", at.device, ws->device);
    }
    ws->img = d_img; ws->rows = rows; ws->cols = cols; ws->stride = stride; ws->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ws->have = false; ws->have_first = false; ws->n_hint = 0;
    hipStream_t st = ws->stream;
    HintIn H; H.n = n_fractions;
    for (int q = 0; q < kMaxHint; ++q) H.frac[q] = q < n_fractions ? fractions[q] : 0.0;
    hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(kSampleThreads), kSampleLds, st, d_img, rows, cols, stride, H, ws->d_states);
    HIP_TRY(hipGetLastError());
    if (int rc = run_range(ws, n_fractions, true)) return rc;
    HIP_TRY(hipMemcpyAsync(ws->h_states, ws->d_states, sizeof(RangeStates), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    ws->n_valid = ws->h_counts[0];
    ws->hint_hist.assign(ws->h_hist, ws->h_hist + (size_t)n_fractions * kBins);
    for (int q = 0; q < n_fractions; ++q) {
        ws->hint_lo[q] = ws->h_states->lo[q]; ws->hint_shift[q] = ws->h_states->shift[q];
        ws->hint_below[q] = ws->h_counts[1 + q];
        unsigned long long in = 0;
        for (int b = 0; b < kBins; ++b) in += ws->hint_hist[(size_t)q * kBins + b];
        ws->hint_inside[q] = in;
    }
    ws->n_hint = n_fractions;
    ws->have = true;
    *n_valid = (int64_t)ws->n_valid;
    return SID_PM_OK;
}

SID_EXPORT int sid_stage_order_stats_ws(sid_stage_ws *ws, const int64_t *ranks, int n_ranks, float *values)
{
    if (!ws || !ws->have) return fail(SID_PM_ERR_STATE, "order_stats_ws needs sid_stage_begin on the image first");
    if (n_ranks < 0 || (n_ranks > 0 && (!ranks || !values))) return fail(SID_PM_ERR_ARG, "bad rank list");
    for (int q = 0; q < n_ranks; ++q) if (ranks[q] < 0) return fail(SID_PM_ERR_ARG, "negative rank");
    DeviceGuard guard(ws->device);
    // ---- ranks inside a hinted range (sid_stage_begin_hint): the bin is known, one more pass resolves it to a key ----
    std::vector<int64_t> slow_ranks; std::vector<int> slow_at;
    {
        struct Want { int k; uint32_t base, width; unsigned long long res; int state; };
        std::vector<Want> wants;
        for (int k = 0; k < n_ranks; ++k) {
            const unsigned long long rk = (unsigned long long)ranks[k];
            if (rk >= ws->n_valid) return fail(SID_PM_ERR_ARG, "rank %lld is not below the number of non-NaN pixels", (long long)ranks[k]);
            bool done = false;
            for (int q = 0; q < ws->n_hint && !done; ++q) {
                if (rk < ws->hint_below[q] || rk >= ws->hint_below[q] + ws->hint_inside[q] || ws->hint_shift[q] > 14u) continue;   // (a bin of up to 8 x 2048 keys)
                unsigned long long r = rk - ws->hint_below[q];
                const unsigned long long *hq = ws->hint_hist.data() + (size_t)q * kBins;
                int b = 0;
                while (r >= hq[b]) { r -= hq[b]; ++b; }
                const uint32_t base = ws->hint_lo[q] + ((uint32_t)b << ws->hint_shift[q]);
                if (ws->hint_shift[q] == 0u) values[k] = key2f(base);       // a bin is one key
                else wants.push_back(Want{k, base, 1u << ws->hint_shift[q], r, -1});
                done = true;
            }
            if (!done) { slow_ranks.push_back(ranks[k]); slow_at.push_back(k); }
        }
        for (size_t w0 = 0; w0 < wants.size();) {                          // up to kMaxStates pieces of 2048 keys per pass
            RangeStates &S = *ws->h_states;
            int ns = 0;
            size_t w1 = w0;
            for (; w1 < wants.size(); ++w1) {                              // a bin wider than 2048 keys takes consecutive states
                const int pieces = (int)std::max<uint32_t>(1u, wants[w1].width >> 11);
                int f = -1;
                for (int t = 0; t < ns; ++t) if (S.lo[t] == wants[w1].base) f = t;
                if (f < 0) {
                    if (ns + pieces > kMaxStates) break;
                    f = ns;
                    for (int pc = 0; pc < pieces; ++pc, ++ns) {
                        S.lo[ns] = wants[w1].base + ((uint32_t)pc << 11);
                        S.span[ns] = std::min<uint32_t>(wants[w1].width, 2048u) - 1u; S.shift[ns] = 0u;
                    }
                }
                wants[w1].state = f;
            }
            for (int t = ns; t < kMaxStates; ++t) { S.lo[t] = 0xffffffffu; S.span[t] = 0u; S.shift[t] = 0u; }
            HIP_TRY(hipMemcpyAsync(ws->d_states, ws->h_states, sizeof(RangeStates), hipMemcpyHostToDevice, ws->stream));
            if (int rc = run_range(ws, ns, false)) return rc;
            for (size_t w = w0; w < w1; ++w) {
                // (the states of one bin are consecutive rows of the histogram: bins 0 .. 2047 of each, in key order)
                const unsigned long long *hq = ws->h_hist + (size_t)wants[w].state * kBins;
                unsigned long long r = wants[w].res;
                uint32_t b = 0;
                while (b < wants[w].width && r >= hq[b]) { r -= hq[b]; ++b; }
                if (b >= wants[w].width) return fail(SID_PM_ERR_STATE, "order statistics: the image changed between the passes");
                values[wants[w].k] = key2f(wants[w].base + b);
            }
            w0 = w1;
        }
        if (slow_ranks.empty()) return SID_PM_OK;
        if (!ws->have_first) { if (int rc = first_digit_pass(ws)) return rc; }
    }
    // ---- three-pass radix select for the rest (everything after a plain sid_stage_begin) ----
    std::vector<float> slow_values(slow_ranks.size());
    if (int rc = radix_select(ws, slow_ranks.data(), (int)slow_ranks.size(), slow_values.data())) return rc;
    for (size_t i = 0; i < slow_at.size(); ++i) values[slow_at[i]] = slow_values[i];
    return SID_PM_OK;
}

namespace {
int radix_select(sid_stage_ws *ws, const int64_t *ranks, int n_ranks, float *values)
{
    // up to kMaxStates ranks per sweep; ranks that still share all chosen digits share a histogram (neighbouring
    // order statistics usually part only in the last digit)
    for (int q0 = 0; q0 < n_ranks; q0 += kMaxStates) {
        const int nr = std::min(kMaxStates, n_ranks - q0);
        uint32_t prefix[kMaxStates] = {0};
        unsigned long long k[kMaxStates];
        auto pick = [&](int q, const unsigned long long *hq, int nbins, int shift) -> int {
            for (int b = 0; b < nbins; ++b) {
                if (k[q] < hq[b]) { prefix[q] |= (uint32_t)b << shift; return SID_PM_OK; }
                k[q] -= hq[b];
            }
            return fail(SID_PM_ERR_ARG, "rank %lld is not below the number of non-NaN pixels", (long long)ranks[q0 + q]);
        };
        for (int q = 0; q < nr; ++q) { k[q] = (unsigned long long)ranks[q0 + q]; if (int rc = pick(q, ws->first, kBins, 21)) return rc; }
        const int shifts[2] = {10, 0}, nb[2] = {kBins, 1024};
        uint32_t mask = 0xffe00000u;
        for (int pass = 0; pass < 2; ++pass) {
            HistStates S; S.n = 0;
            int state_of[kMaxStates];
            for (int q = 0; q < nr; ++q) {
                int f = -1;
                for (int t = 0; t < S.n; ++t) if (S.prefix[t] == prefix[q]) f = t;
                if (f < 0) { f = S.n; S.prefix[S.n++] = prefix[q]; }
                state_of[q] = f;
            }
            for (int t = S.n; t < kMaxStates; ++t) S.prefix[t] = S.prefix[0];     // (the compile-time state counts test up to 4)
            if (int rc = run_hist(ws, S, mask, shifts[pass], nb[pass], false)) return rc;
            for (int q = 0; q < nr; ++q) if (int rc = pick(q, ws->h_hist + (size_t)state_of[q] * kBins, nb[pass], shifts[pass])) return rc;
            mask |= (uint32_t)(nb[pass] - 1) << shifts[pass];
        }
        for (int q = 0; q < nr; ++q) values[q0 + q] = key2f(prefix[q]);
    }
    return SID_PM_OK;
}
}  // namespace

// one-shot forms (a temporary workspace per call)
SID_EXPORT int sid_stage_count_valid(const float *d_img, int64_t rows, int64_t cols, int64_t stride, int64_t *n_valid,
                                     void *hip_stream)
{
    int dev = 0; (void)hipGetDevice(&dev);
    sid_stage_ws *ws = nullptr;
    int rc = sid_stage_create(dev, &ws);
    if (!rc) rc = sid_stage_begin(ws, d_img, rows, cols, stride, n_valid, hip_stream);
    sid_stage_destroy(ws);
    return rc;
}

SID_EXPORT int sid_stage_order_stats(const float *d_img, int64_t rows, int64_t cols, int64_t stride,
                                     const int64_t *ranks, int n_ranks, float *values, void *hip_stream)
{
    int dev = 0; (void)hipGetDevice(&dev);
    sid_stage_ws *ws = nullptr;
    int64_t n = 0;
    int rc = sid_stage_create(dev, &ws);
    if (!rc) rc = sid_stage_begin(ws, d_img, rows, cols, stride, &n, hip_stream);
    if (!rc) rc = sid_stage_order_stats_ws(ws, ranks, n_ranks, values);
    sid_stage_destroy(ws);
    return rc;
}

SID_EXPORT int sid_stage_scale_u8(const float *d_img, int64_t rows, int64_t cols, int64_t stride, float vmin, float denom,
                                  uint8_t *d_out, int64_t out_stride, void *hip_stream)
{
    if (int rc = check_img(d_img, rows, cols, stride)) return rc;
    if (!d_out || out_stride < cols) return fail(SID_PM_ERR_ARG, "bad output buffer");
    int dev = -1;
    {   // launch on the device that holds the image (the null stream belongs to the CURRENT device)
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, d_img) == hipSuccess && at.type == hipMemoryTypeDevice) dev = at.device;
    }
    int cur = 0; (void)hipGetDevice(&cur);
    DeviceGuard guard(dev >= 0 ? dev : cur);
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const bool vec = vec_ok(d_img, cols, stride) && (reinterpret_cast<uintptr_t>(d_out) & 3) == 0 && (out_stride & 3) == 0;
    if (vec) hipLaunchKernelGGL(scale_kernel<true>, dim3(grid_rows(rows)), dim3(kThreads), 0, st, d_img, rows, cols, stride, vmin, denom, d_out, out_stride);
    else hipLaunchKernelGGL(scale_kernel<false>, dim3(grid_rows(rows)), dim3(kThreads), 0, st, d_img, rows, cols, stride, vmin, denom, d_out, out_stride);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "scale_u8: %s", hipGetErrorString(e));
    return SID_PM_OK;
}
