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
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(hist_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxStates * kBins * 4);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(hist_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxStates * kBins * 4);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) { if (ws) { (void)hipFree(ws->d_hist); if (ws->h_hist) (void)hipHostFree(ws->h_hist); delete ws; } return fail(SID_PM_ERR_HIP, "workspace: %s", hipGetErrorString(e)); }
    *out = ws;
    return SID_PM_OK;
}

SID_EXPORT void sid_stage_destroy(sid_stage_ws *ws)
{
    if (!ws) return;
    DeviceGuard guard(ws->device);
    (void)hipFree(ws->d_hist);
    if (ws->h_hist) (void)hipHostFree(ws->h_hist);
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
    ws->have = false;
    HistStates S; S.n = 1; S.prefix[0] = 0;
    if (int rc = run_hist(ws, S, 0u, 21, kBins, true)) return rc;
    unsigned long long n = 0;
    for (int b = 0; b < kBins; ++b) { ws->first[b] = ws->h_hist[b]; n += ws->h_hist[b]; }
    ws->have = true;
    *n_valid = (int64_t)n;
    return SID_PM_OK;
}

SID_EXPORT int sid_stage_order_stats_ws(sid_stage_ws *ws, const int64_t *ranks, int n_ranks, float *values)
{
    if (!ws || !ws->have) return fail(SID_PM_ERR_STATE, "order_stats_ws needs sid_stage_begin on the image first");
    if (n_ranks < 0 || (n_ranks > 0 && (!ranks || !values))) return fail(SID_PM_ERR_ARG, "bad rank list");
    for (int q = 0; q < n_ranks; ++q) if (ranks[q] < 0) return fail(SID_PM_ERR_ARG, "negative rank");
    DeviceGuard guard(ws->device);
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
