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

__global__ __launch_bounds__(kThreads) void count_kernel(const float *img, int64_t rows, int64_t cols, int64_t stride,
                                                         unsigned long long *out)
{
    unsigned long long c = 0;
    const int64_t n = rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const int64_t r = i / cols, cc = i - r * cols;
        const float v = img[r * stride + cc];
        c += (v == v) ? 1ull : 0ull;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// histogram of digit (key >> shift) & 255 over the pixels whose key matches `prefix` under `mask`
__global__ __launch_bounds__(kThreads) void hist_kernel(const float *img, int64_t rows, int64_t cols, int64_t stride,
                                                        uint32_t prefix, uint32_t mask, int shift, unsigned long long *hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t n = rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const int64_t r = i / cols, cc = i - r * cols;
        const float v = img[r * stride + cc];
        const uint32_t key = f2key(v);
        if (v == v && (key & mask) == prefix) atomicAdd(&h[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

__global__ __launch_bounds__(kThreads) void scale_kernel(const float *img, int64_t rows, int64_t cols, int64_t stride,
                                                         float vmin, float denom, uint8_t *out, int64_t out_stride)
{
    const int64_t n = rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const int64_t r = i / cols, c = i - r * cols;
        const float x = img[r * stride + c];
        float t = x - vmin;                                  // lib.py:54, one float32 rounding per operation
        t = 254.0f * t;
        t = t / denom;
        t = 1.0f + t;
        t = t < 1.0f ? 1.0f : t;                             // lib.py:55-56 (NaN fails both comparisons)
        t = t > 255.0f ? 255.0f : t;
        const bool finite = fabsf(x) <= 3.402823466e38f;     // false for NaN and +-inf (lib.py:57)
        out[r * out_stride + c] = (finite && t == t) ? (uint8_t)t : (uint8_t)0;
    }
}

int grid_for(int64_t n)
{
    const int64_t b = (n + kThreads - 1) / kThreads;
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, 256 * 16));
}

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
        hipLaunchKernelGGL(count_kernel, dim3(grid_for(rows * cols)), dim3(kThreads), 0, st, d_img, rows, cols, stride, d);
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
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    unsigned long long *d_hist = nullptr;
    HIP_TRY(hipMalloc(&d_hist, 256 * sizeof *d_hist));
    const int grid = grid_for(rows * cols);
    int rc = SID_PM_OK;
    std::vector<unsigned long long> h(256);
    for (int q = 0; q < n_ranks && rc == SID_PM_OK; ++q) {
        if (ranks[q] < 0) { rc = fail(SID_PM_ERR_ARG, "negative rank"); break; }
        uint32_t prefix = 0, mask = 0;
        unsigned long long k = (unsigned long long)ranks[q];
        for (int pass = 0; pass < 4 && rc == SID_PM_OK; ++pass) {
            const int shift = 24 - 8 * pass;
            hipError_t e = hipMemsetAsync(d_hist, 0, 256 * sizeof *d_hist, st);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(hist_kernel, dim3(grid), dim3(kThreads), 0, st, d_img, rows, cols, stride, prefix, mask, shift, d_hist);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_hist, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { rc = fail(SID_PM_ERR_HIP, "order_stats: %s", hipGetErrorString(e)); break; }
            int digit = -1;
            for (int b = 0; b < 256; ++b) {
                if (k < h[(size_t)b]) { digit = b; break; }
                k -= h[(size_t)b];
            }
            if (digit < 0) { rc = fail(SID_PM_ERR_ARG, "rank %lld is not below the number of non-NaN pixels", (long long)ranks[q]); break; }
            prefix |= (uint32_t)digit << shift;
            mask |= 255u << shift;
        }
        if (rc == SID_PM_OK) values[q] = key2f(prefix);
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
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(rows * cols)), dim3(kThreads), 0, st, d_img, rows, cols, stride, vmin, denom,
                       d_out, out_stride);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "scale_u8: %s", hipGetErrorString(e));
    return SID_PM_OK;
}
