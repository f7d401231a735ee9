// stage.hip - uint8 staging of a float32 SAR image on gfx950 (C ABI: include/sid_stage.h; replaces the two
// full-image passes of get_uint8_image, lib.py:27-59).  Both kernels are HBM-bound streaming passes:
//   order statistics: 4 x 8-bit radix select on the monotone key of the float (per-workgroup LDS histograms,
//                     256 global atomics per workgroup), one pass of 4 B/pixel per digit;
//   scale:            4 B read + 1 B written per pixel, float32 operation for operation as NumPy.
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

// All three kernels walk the image row by row (workgroup = row, thread = column, both strided): coalesced
// dword loads, no per-pixel division.
__global__ __launch_bounds__(kThreads) void count_kernel(const float *img, int64_t rows, int64_t cols, int64_t stride,
                                                         unsigned long long *out)
{
    unsigned long long c = 0;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float *row = img + r * stride;
        for (int64_t x = threadIdx.x; x < cols; x += kThreads) { const float v = row[x]; c += (v == v) ? 1ull : 0ull; }
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

constexpr int kMaxStates = 8;            // order statistics resolved together in one pass over the image
struct HistStates { uint32_t prefix[kMaxStates]; int n; };

// For every state q: histogram of digit (key >> shift) & 255 over the pixels whose key matches prefix[q] under
// `mask` (the same mask for all: they are all at the same digit).  hist: [n][256].
__global__ __launch_bounds__(kThreads) void hist_kernel(const float *img, int64_t rows, int64_t cols, int64_t stride,
                                                        HistStates S, uint32_t mask, int shift, unsigned long long *hist)
{
    __shared__ uint32_t h[kMaxStates][256];
    for (int q = 0; q < S.n; ++q) h[q][threadIdx.x] = 0;
    __syncthreads();
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float *row = img + r * stride;
        for (int64_t x = threadIdx.x; x < cols; x += kThreads) {
            const float v = row[x];
            if (v == v) {
                const uint32_t key = f2key(v), pk = key & mask, digit = (key >> shift) & 255u;
                for (int q = 0; q < S.n; ++q)
                    if (pk == S.prefix[q]) atomicAdd(&h[q][digit], 1u);
            }
        }
    }
    __syncthreads();
    for (int q = 0; q < S.n; ++q)
        if (h[q][threadIdx.x]) atomicAdd(&hist[q * 256 + threadIdx.x], (unsigned long long)h[q][threadIdx.x]);
}

__global__ __launch_bounds__(kThreads) void scale_kernel(const float *img, int64_t rows, int64_t cols, int64_t stride,
                                                         float vmin, float denom, uint8_t *out, int64_t out_stride)
{
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float *row = img + r * stride;
        uint8_t *orow = out + r * out_stride;
        for (int64_t c = threadIdx.x; c < cols; c += kThreads) {
            const float x = row[c];
            float t = x - vmin;                              // lib.py:54, one float32 rounding per operation
            t = 254.0f * t;
            t = t / denom;
            t = 1.0f + t;
            t = t < 1.0f ? 1.0f : t;                         // lib.py:55-56 (NaN fails both comparisons)
            t = t > 255.0f ? 255.0f : t;
            const bool finite = fabsf(x) <= 3.402823466e38f; // false for NaN and +-inf (lib.py:57)
            orow[c] = (finite && t == t) ? (uint8_t)t : (uint8_t)0;
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

}  // namespace

SID_EXPORT const char *sid_stage_last_error(void) { return g_err; }

SID_EXPORT int sid_stage_count_valid(const float *d_img, int64_t rows, int64_t cols, int64_t stride, int64_t *n_valid,
                                     void *hip_stream)
{
    if (int rc = check_img(d_img, rows, cols, stride)) return rc;
    if (!n_valid) return fail(SID_PM_ERR_ARG, "null output");
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof *d));
    hipError_t e = hipMemsetAsync(d, 0, sizeof *d, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(count_kernel, dim3(grid_rows(rows)), dim3(kThreads), 0, st, d_img, rows, cols, stride, d);
        e = hipGetLastError();
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "count_valid: %s", hipGetErrorString(e));
    *n_valid = (int64_t)h;
    return SID_PM_OK;
}

SID_EXPORT int sid_stage_order_stats(const float *d_img, int64_t rows, int64_t cols, int64_t stride,
                                     const int64_t *ranks, int n_ranks, float *values, void *hip_stream)
{
    if (int rc = check_img(d_img, rows, cols, stride)) return rc;
    if (n_ranks < 0 || (n_ranks > 0 && (!ranks || !values))) return fail(SID_PM_ERR_ARG, "bad rank list");
    for (int q = 0; q < n_ranks; ++q) if (ranks[q] < 0) return fail(SID_PM_ERR_ARG, "negative rank");
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    unsigned long long *d_hist = nullptr;
    HIP_TRY(hipMalloc(&d_hist, kMaxStates * 256 * sizeof *d_hist));
    const int grid = grid_rows(rows);
    int rc = SID_PM_OK;
    std::vector<unsigned long long> h((size_t)kMaxStates * 256);
    // up to kMaxStates ranks per sweep of four digit passes; ranks that still share all chosen digits share a
    // histogram (neighbouring order statistics usually part only in the last digit)
    for (int q0 = 0; q0 < n_ranks && rc == SID_PM_OK; q0 += kMaxStates) {
        const int nr = std::min(kMaxStates, n_ranks - q0);
        uint32_t prefix[kMaxStates] = {0};
        unsigned long long k[kMaxStates];
        for (int q = 0; q < nr; ++q) k[q] = (unsigned long long)ranks[q0 + q];
        uint32_t mask = 0;
        for (int pass = 0; pass < 4 && rc == SID_PM_OK; ++pass) {
            const int shift = 24 - 8 * pass;
            HistStates S; S.n = 0;
            int state_of[kMaxStates];
            for (int q = 0; q < nr; ++q) {
                int f = -1;
                for (int t = 0; t < S.n; ++t) if (S.prefix[t] == prefix[q]) f = t;
                if (f < 0) { f = S.n; S.prefix[S.n++] = prefix[q]; }
                state_of[q] = f;
            }
            hipError_t e = hipMemsetAsync(d_hist, 0, (size_t)S.n * 256 * sizeof *d_hist, st);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(hist_kernel, dim3(grid), dim3(kThreads), 0, st, d_img, rows, cols, stride, S, mask, shift, d_hist);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_hist, (size_t)S.n * 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { rc = fail(SID_PM_ERR_HIP, "order_stats: %s", hipGetErrorString(e)); break; }
            for (int q = 0; q < nr && rc == SID_PM_OK; ++q) {
                const unsigned long long *hq = h.data() + (size_t)state_of[q] * 256;
                int digit = -1;
                for (int b = 0; b < 256; ++b) {
                    if (k[q] < hq[b]) { digit = b; break; }
                    k[q] -= hq[b];
                }
                if (digit < 0) { rc = fail(SID_PM_ERR_ARG, "rank %lld is not below the number of non-NaN pixels", (long long)ranks[q0 + q]); break; }
                prefix[q] |= (uint32_t)digit << shift;
            }
            mask |= 255u << shift;
        }
        for (int q = 0; q < nr && rc == SID_PM_OK; ++q) values[q0 + q] = key2f(prefix[q]);
    }
    (void)hipFree(d_hist);
    return rc;
}

SID_EXPORT int sid_stage_scale_u8(const float *d_img, int64_t rows, int64_t cols, int64_t stride, float vmin, float denom,
                                  uint8_t *d_out, int64_t out_stride, void *hip_stream)
{
    if (int rc = check_img(d_img, rows, cols, stride)) return rc;
    if (!d_out || out_stride < cols) return fail(SID_PM_ERR_ARG, "bad output buffer");
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    hipLaunchKernelGGL(scale_kernel, dim3(grid_rows(rows)), dim3(kThreads), 0, st, d_img, rows, cols, stride, vmin, denom,
                       d_out, out_stride);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "scale_u8: %s", hipGetErrorString(e));
    return SID_PM_OK;
}
