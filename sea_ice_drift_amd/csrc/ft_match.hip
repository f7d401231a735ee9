// ft_match.hip - Hamming k=2 nearest-neighbour matcher for 256-bit ORB descriptors on gfx950.
// C ABI: include/sid_ft.h (replaces cv2.BFMatcher(NORM_HAMMING).knnMatch(d1, d2, k=2), ftlib.py:92-99).
//
// One thread owns one query descriptor (8 dwords in VGPRs).  The train descriptors are the same for every
// lane, so they come through the scalar cache (s_load_dwordx8) and cost no LDS and no vector memory; per
// pair: 8 v_xor + 8 v_bcnt (accumulating) + 1 key build + 3 min/max for the running top two.  The train set
// is cut into chunks (second grid dimension) so that a 20k x 20k problem fills 256 CUs; a second tiny kernel
// merges the per-chunk top twos.  Candidates are ordered by key = distance << 22 | train index, i.e. by
// distance, then by the smaller index.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <algorithm>

#include "../../include/sid_ft.h"
#include "../../include/sid_pm.h"

#define SID_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kThreads = 256;
constexpr uint32_t kNone = 0xffffffffu;
constexpr int64_t kMaxTrain = (int64_t)1 << 22;             // index bits of the key

thread_local char g_err[256] = "";
int fail(int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}

__device__ __forceinline__ void push(uint32_t &k1, uint32_t &k2, uint32_t key)
{
    const uint32_t t = k1 > key ? k1 : key;                 // the larger of (best, new)
    k1 = k1 < key ? k1 : key;
    k2 = k2 < t ? k2 : t;
}

__global__ __launch_bounds__(kThreads) void ft_knn2_partial(const uint4 *__restrict__ d1, int n1,
                                                            const uint4 *__restrict__ d2, int n2, int chunk,
                                                            uint32_t *__restrict__ part, int nchunks)
{
    const int q = blockIdx.x * kThreads + threadIdx.x;
    const int qc = q < n1 ? q : n1 - 1;
    const uint4 qa = d1[2 * qc], qb = d1[2 * qc + 1];
    const int j0 = blockIdx.y * chunk, j1 = j0 + chunk < n2 ? j0 + chunk : n2;
    uint32_t k1 = kNone, k2 = kNone;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {                          // j is uniform: scalar loads of the train descriptor
        const uint4 ta = d2[2 * j], tb = d2[2 * j + 1];
        uint32_t d = __builtin_popcount(qa.x ^ ta.x);
        d += __builtin_popcount(qa.y ^ ta.y);
        d += __builtin_popcount(qa.z ^ ta.z);
        d += __builtin_popcount(qa.w ^ ta.w);
        d += __builtin_popcount(qb.x ^ tb.x);
        d += __builtin_popcount(qb.y ^ tb.y);
        d += __builtin_popcount(qb.z ^ tb.z);
        d += __builtin_popcount(qb.w ^ tb.w);
        push(k1, k2, (d << 22) | (uint32_t)j);
    }
    if (q < n1) {
        part[((size_t)q * nchunks + blockIdx.y) * 2 + 0] = k1;
        part[((size_t)q * nchunks + blockIdx.y) * 2 + 1] = k2;
    }
}

__global__ __launch_bounds__(kThreads) void ft_knn2_merge(const uint32_t *__restrict__ part, int n1, int nchunks,
                                                          int32_t *__restrict__ idx, int32_t *__restrict__ dist)
{
    const int q = blockIdx.x * kThreads + threadIdx.x;
    if (q >= n1) return;
    uint32_t k1 = kNone, k2 = kNone;
    for (int c = 0; c < nchunks; ++c) {
        push(k1, k2, part[((size_t)q * nchunks + c) * 2 + 0]);
        push(k1, k2, part[((size_t)q * nchunks + c) * 2 + 1]);
    }
    idx[2 * q + 0] = k1 == kNone ? -1 : (int32_t)(k1 & 0x3fffffu);
    dist[2 * q + 0] = k1 == kNone ? -1 : (int32_t)(k1 >> 22);
    idx[2 * q + 1] = k2 == kNone ? -1 : (int32_t)(k2 & 0x3fffffu);
    dist[2 * q + 1] = k2 == kNone ? -1 : (int32_t)(k2 >> 22);
}

// chunk size: enough blocks to fill the chip (>= ~2048 blocks), at least 256 train descriptors per chunk
void plan(int64_t n1, int64_t n2, int &chunk, int &nchunks)
{
    const int64_t qblocks = (n1 + kThreads - 1) / kThreads;
    int64_t want = std::max<int64_t>(1, (2048 + qblocks - 1) / std::max<int64_t>(qblocks, 1));
    int64_t c = std::max<int64_t>(256, (n2 + want - 1) / want);
    c = std::min<int64_t>(c, std::max<int64_t>(n2, 1));
    chunk = (int)c;
    nchunks = (int)((n2 + c - 1) / c);
    if (nchunks < 1) nchunks = 1;
}

}  // namespace

SID_EXPORT const char *sid_ft_last_error(void) { return g_err; }

SID_EXPORT int64_t sid_ft_workspace_bytes(int64_t n1, int64_t n2)
{
    if (n1 < 0 || n2 < 0) return 0;
    int chunk, nchunks;
    plan(n1, n2, chunk, nchunks);
    return (int64_t)sizeof(uint32_t) * 2 * std::max<int64_t>(n1, 1) * nchunks;
}

SID_EXPORT int sid_ft_knn2_device(const uint8_t *d_desc1, int64_t n1, const uint8_t *d_desc2, int64_t n2,
                                  int32_t *d_idx, int32_t *d_dist, void *d_workspace, void *hip_stream)
{
    if (n1 < 0 || n2 < 0) return fail(SID_PM_ERR_ARG, "negative descriptor count");
    if (n1 == 0) return SID_PM_OK;
    if (!d_desc1 || !d_idx || !d_dist || !d_workspace || (n2 > 0 && !d_desc2)) return fail(SID_PM_ERR_ARG, "null pointer");
    if (n1 > 0x7fffffff / 4 || n2 >= kMaxTrain)
        return fail(SID_PM_ERR_UNSUPPORTED, "at most %lld train descriptors", (long long)kMaxTrain - 1);
    if ((reinterpret_cast<uintptr_t>(d_desc1) | reinterpret_cast<uintptr_t>(d_desc2)) & 15)
        return fail(SID_PM_ERR_ARG, "descriptors must be 16-byte aligned");
    int chunk, nchunks;
    plan(n1, n2, chunk, nchunks);
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const dim3 grid((unsigned)((n1 + kThreads - 1) / kThreads), (unsigned)nchunks);
    hipLaunchKernelGGL(ft_knn2_partial, grid, dim3(kThreads), 0, st, reinterpret_cast<const uint4 *>(d_desc1), (int)n1,
                       reinterpret_cast<const uint4 *>(d_desc2), (int)n2, chunk, reinterpret_cast<uint32_t *>(d_workspace), nchunks);
    hipLaunchKernelGGL(ft_knn2_merge, dim3(grid.x), dim3(kThreads), 0, st, reinterpret_cast<const uint32_t *>(d_workspace),
                       (int)n1, nchunks, d_idx, d_dist);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "matcher launch failed: %s", hipGetErrorString(e));
    return SID_PM_OK;
}

SID_EXPORT int sid_ft_knn2(int device, const uint8_t *desc1, int64_t n1, const uint8_t *desc2, int64_t n2,
                           int32_t *idx, int32_t *dist)
{
    if (n1 < 0 || n2 < 0) return fail(SID_PM_ERR_ARG, "negative descriptor count");
    if (n1 == 0) return SID_PM_OK;
    if (!desc1 || !idx || !dist || (n2 > 0 && !desc2)) return fail(SID_PM_ERR_ARG, "null pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(SID_PM_ERR_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(SID_PM_ERR_ARG, "device %d out of range", device);
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(device);
    uint8_t *b1 = nullptr, *b2 = nullptr, *ws = nullptr;
    int32_t *bi = nullptr, *bd = nullptr;
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    const size_t s1 = (size_t)n1 * SID_FT_DESC_BYTES, s2 = (size_t)std::max<int64_t>(n2, 1) * SID_FT_DESC_BYTES;
    step(hipMalloc(&b1, s1)); step(hipMalloc(&b2, s2)); step(hipMalloc(&ws, (size_t)sid_ft_workspace_bytes(n1, n2)));
    step(hipMalloc(&bi, sizeof(int32_t) * 2 * (size_t)n1)); step(hipMalloc(&bd, sizeof(int32_t) * 2 * (size_t)n1));
    if (e == hipSuccess) {
        step(hipMemcpy(b1, desc1, s1, hipMemcpyHostToDevice));
        if (n2 > 0) step(hipMemcpy(b2, desc2, (size_t)n2 * SID_FT_DESC_BYTES, hipMemcpyHostToDevice));
    }
    int rc = SID_PM_OK;
    if (e == hipSuccess) rc = sid_ft_knn2_device(b1, n1, b2, n2, bi, bd, ws, nullptr);
    if (e == hipSuccess && rc == SID_PM_OK) {
        step(hipDeviceSynchronize());
        step(hipMemcpy(idx, bi, sizeof(int32_t) * 2 * (size_t)n1, hipMemcpyDeviceToHost));
        step(hipMemcpy(dist, bd, sizeof(int32_t) * 2 * (size_t)n1, hipMemcpyDeviceToHost));
    }
    (void)hipFree(b1); (void)hipFree(b2); (void)hipFree(ws); (void)hipFree(bi); (void)hipFree(bd);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "sid_ft_knn2: %s", hipGetErrorString(e));
    return rc;
}
