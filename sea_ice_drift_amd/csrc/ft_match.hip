// ft_match.hip - Hamming k=2 nearest-neighbour matcher for 256-bit ORB descriptors on gfx950.
// C ABI: include/sid_ft.h (replaces cv2.BFMatcher(NORM_HAMMING).knnMatch(d1, d2, k=2), ftlib.py:92-99).
//
// Two kernels for the distances.  Large problems (ft_knn2_mfma): the Hamming distance of bit vectors is
//     |a xor b| = |a| + |b| - 2 a.b,
// and a.b over 256 bits is a dot product of 0/1 bytes - an int8 GEMM of shape n1 x n2 x 256 on v_mfma_i32_16x16x64_i8 (4
// MFMAs per 16 x 16 pairs instead of 16 x 16 x 16 VALU instructions).  Descriptors are expanded once to one byte per bit; a
// workgroup holds 256 queries as MFMA A fragments in registers (4 wavefronts x 4 sets of 16 rows), streams the train
// descriptors through LDS 256 at a time (rows of 256 + 32 bytes: every ds_read_b128 lane group conflict-free) and keeps, per
// lane, the running top two of its 16 (query, column) cells; the ordering key (|b| + 256 - 2 a.b) << 22 | j comes out of ONE
// v_mad_i32_i24 per value (dot x -2^23 + c_j with c_j = (|b_j| + 256) << 22 | j precomputed; |a| is constant per query and
// added at the end), then three min / max.  Small problems (ft_knn2_partial): as in round 1 -
// one thread owns one query descriptor (8 dwords in VGPRs).  The train descriptors are the same for every
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
#include <mutex>
#include <stdlib.h>

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

// ---------------------------------------------------------------------------------------------------------------
// MFMA form
// ---------------------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kQB = 256;            // queries per workgroup
constexpr int kTB = 256;            // train descriptors per LDS batch
constexpr int kRowB = 288;          // LDS bytes per expanded descriptor: 256 + 32 (slot of lane (n, kg) = 2 n + kg mod 16: distinct in every ds_read_b128 lane group)

// one byte per bit: E[d][8 b + k] = bit k of byte b of descriptor d (rows n .. npad-1 zero); 8 output bytes per thread
__global__ __launch_bounds__(kThreads) void ft_expand(const uint8_t *__restrict__ desc, int n, int npad, uint8_t *__restrict__ E)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (int64_t)npad * 32) return;
    const int d = (int)(i >> 5);
    const uint32_t v = d < n ? desc[i] : 0u;
    uint2 o;
    o.x = (v & 1u) | ((v & 2u) << 7) | ((v & 4u) << 14) | ((v & 8u) << 21);
    o.y = ((v >> 4) & 1u) | ((v & 32u) << 3) | ((v & 64u) << 10) | ((v & 128u) << 17);
    reinterpret_cast<uint2 *>(E)[i] = o;
}

// c[j] = (|b_j| + 256) << 22 | j; kNone for the padding rows (whatever the dot product, their key stays kNone: dot = 0)
__global__ __launch_bounds__(kThreads) void ft_train_terms(const uint4 *__restrict__ d2, int n2, int n2pad, uint32_t *__restrict__ c)
{
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= n2pad) return;
    if (j >= n2) { c[j] = kNone; return; }
    const uint4 a = d2[2 * j], b = d2[2 * j + 1];
    const uint32_t pc = __builtin_popcount(a.x) + __builtin_popcount(a.y) + __builtin_popcount(a.z) + __builtin_popcount(a.w) +
                        __builtin_popcount(b.x) + __builtin_popcount(b.y) + __builtin_popcount(b.z) + __builtin_popcount(b.w);
    c[j] = ((pc + 256u) << 22) | (uint32_t)j;
}

// part[q][chunk][2]: the two smallest keys (|b| + 256 - 2 a.b) << 22 | j of query q over the chunk's train descriptors
__global__ __launch_bounds__(kThreads, 2) void ft_knn2_mfma(const uint8_t *__restrict__ E1, const uint8_t *__restrict__ E2,
                                                            const uint32_t *__restrict__ c2, int n1, int n2pad, int chunk,
                                                            uint32_t *__restrict__ part, int nchunks)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];      // kTB rows of kRowB bytes + kTB terms
    uint32_t *lc = reinterpret_cast<uint32_t *>(lds + kTB * kRowB);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 15, kg = lane >> 4;
    const int qbase = blockIdx.x * kQB + wv * 64;
    v4i a[4][4];                                                       // [set of 16 queries][64-bit step]: row n, bytes 16 kg ..
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            a[s][t] = *reinterpret_cast<const v4i *>(E1 + (size_t)(qbase + 16 * s + n) * 256 + 64 * t + 16 * kg);
    uint32_t k1[4][4], k2[4][4];                                       // [set][row 4 kg + j of the set]
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) { k1[s][j] = kNone; k2[s][j] = kNone; }
    const int j0 = blockIdx.y * chunk, j1 = j0 + chunk < n2pad ? j0 + chunk : n2pad;
    for (int jb = j0; jb < j1; jb += kTB) {
        __syncthreads();
        {   // stage kTB expanded descriptors (64 KB, coalesced 16-byte units) and their terms
            const uint4 *src = reinterpret_cast<const uint4 *>(E2 + (size_t)jb * 256);
#pragma unroll 4
            for (int u = 0; u < 16; ++u) {
                const int idx = u * kThreads + threadIdx.x, d = idx >> 4, w = idx & 15;
                *reinterpret_cast<uint4 *>(lds + d * kRowB + w * 16) = src[idx];
            }
            lc[threadIdx.x] = c2[jb + threadIdx.x];
        }
        __syncthreads();
#pragma unroll 2
        for (int tile = 0; tile < kTB / 16; ++tile) {
            const uint8_t *bp = lds + (tile * 16 + n) * kRowB + 16 * kg;
            const v4i b0 = *reinterpret_cast<const v4i *>(bp), b1 = *reinterpret_cast<const v4i *>(bp + 64),
                      b2 = *reinterpret_cast<const v4i *>(bp + 128), b3 = *reinterpret_cast<const v4i *>(bp + 192);
            const int cj = (int)lc[tile * 16 + n];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                v4i acc = {0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s][0], b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s][1], b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s][2], b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s][3], b3, acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j)                            // acc[j]: query row 4 kg + j of the set, train column n
                    push(k1[s][j], k2[s][j], (uint32_t)__mul24(acc[j], -(1 << 23)) + (uint32_t)cj);
            }
        }
    }
    // the 16 lanes of equal kg hold the same rows for different columns: merge their top twos
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t x1 = k1[s][j], x2 = k2[s][j];
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const uint32_t o1 = (uint32_t)__shfl_xor((int)x1, d), o2 = (uint32_t)__shfl_xor((int)x2, d);
                push(x1, x2, o1); push(x1, x2, o2);
            }
            const int q = qbase + 16 * s + 4 * kg + j;
            if (n == 0 && q < n1) {
                part[((size_t)q * nchunks + blockIdx.y) * 2 + 0] = x1;
                part[((size_t)q * nchunks + blockIdx.y) * 2 + 1] = x2;
            }
        }
}

// merge of the MFMA form: keys carry |b| + 256 - 2 a.b; the distance adds |a| - 256
__global__ __launch_bounds__(kThreads) void ft_knn2_merge_mfma(const uint32_t *__restrict__ part, const uint4 *__restrict__ d1, int n1, int nchunks,
                                                               int32_t *__restrict__ idx, int32_t *__restrict__ dist)
{
    const int q = blockIdx.x * kThreads + threadIdx.x;
    if (q >= n1) return;
    uint32_t k1 = kNone, k2 = kNone;
    for (int c = 0; c < nchunks; ++c) {
        push(k1, k2, part[((size_t)q * nchunks + c) * 2 + 0]);
        push(k1, k2, part[((size_t)q * nchunks + c) * 2 + 1]);
    }
    const uint4 a = d1[2 * q], b = d1[2 * q + 1];
    const int pa = (int)(__builtin_popcount(a.x) + __builtin_popcount(a.y) + __builtin_popcount(a.z) + __builtin_popcount(a.w) +
                         __builtin_popcount(b.x) + __builtin_popcount(b.y) + __builtin_popcount(b.z) + __builtin_popcount(b.w));
    idx[2 * q + 0] = k1 == kNone ? -1 : (int32_t)(k1 & 0x3fffffu);
    dist[2 * q + 0] = k1 == kNone ? -1 : (int32_t)(k1 >> 22) - 256 + pa;
    idx[2 * q + 1] = k2 == kNone ? -1 : (int32_t)(k2 & 0x3fffffu);
    dist[2 * q + 1] = k2 == kNone ? -1 : (int32_t)(k2 >> 22) - 256 + pa;
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

// MFMA form: queries and train descriptors padded to 256; chunks of whole LDS batches, >= ~512 workgroups (two per CU)
struct MfmaPlan { int64_t n1pad, n2pad; int chunk, nchunks; size_t off_e1, off_e2, off_c, total; };
constexpr int64_t kMfmaMinPairs = (int64_t)1 << 24;                    // below this the one-thread-per-query kernel is as fast
bool use_mfma(int64_t n1, int64_t n2) { return n1 * n2 >= kMfmaMinPairs && n2 >= 1024 && getenv("SID_FT_NO_MFMA") == nullptr; }
MfmaPlan plan_mfma(int64_t n1, int64_t n2)
{
    MfmaPlan P;
    P.n1pad = (std::max<int64_t>(n1, 1) + kQB - 1) / kQB * kQB; P.n2pad = (std::max<int64_t>(n2, 1) + kTB - 1) / kTB * kTB;   // (empty sets: sized, never run)
    const int64_t qblocks = P.n1pad / kQB, batches = P.n2pad / kTB;
    int64_t want = std::max<int64_t>(1, (512 + qblocks - 1) / qblocks);
    want = std::min<int64_t>(want, batches);
    const int64_t per = (batches + want - 1) / want;
    P.chunk = (int)(per * kTB); P.nchunks = (int)((batches + per - 1) / per);
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    P.off_e1 = up(sizeof(uint32_t) * 2 * (size_t)std::max<int64_t>(n1, 1) * P.nchunks);
    P.off_e2 = P.off_e1 + (size_t)P.n1pad * 256;
    P.off_c = P.off_e2 + (size_t)P.n2pad * 256;
    P.total = P.off_c + up(sizeof(uint32_t) * (size_t)P.n2pad);
    return P;
}

}  // namespace

SID_EXPORT const char *sid_ft_last_error(void) { return g_err; }

SID_EXPORT int64_t sid_ft_workspace_bytes(int64_t n1, int64_t n2)
{
    if (n1 < 0 || n2 < 0) return 0;
    int chunk, nchunks;
    plan(n1, n2, chunk, nchunks);
    const int64_t valu = (int64_t)sizeof(uint32_t) * 2 * std::max<int64_t>(n1, 1) * nchunks;
    return std::max<int64_t>(valu, (int64_t)plan_mfma(n1, n2).total);      // (either kernel may run: SID_FT_NO_MFMA, size)
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
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    if (use_mfma(n1, n2)) {
        if (reinterpret_cast<uintptr_t>(d_workspace) & 15) return fail(SID_PM_ERR_ARG, "workspace must be 16-byte aligned");
        const MfmaPlan P = plan_mfma(n1, n2);
        uint8_t *ws = reinterpret_cast<uint8_t *>(d_workspace);
        uint32_t *part = reinterpret_cast<uint32_t *>(ws);
        uint8_t *e1 = ws + P.off_e1, *e2 = ws + P.off_e2;
        uint32_t *c2 = reinterpret_cast<uint32_t *>(ws + P.off_c);
        const int lds = kTB * kRowB + kTB * (int)sizeof(uint32_t);
        static std::once_flag once[16];
        int dev = 0; (void)hipGetDevice(&dev);
        hipError_t ea = hipSuccess;
        std::call_once(once[dev & 15], [&] { ea = hipFuncSetAttribute(reinterpret_cast<const void *>(ft_knn2_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
        if (ea != hipSuccess) return fail(SID_PM_ERR_HIP, "matcher LDS limit: %s", hipGetErrorString(ea));
        hipLaunchKernelGGL(ft_expand, dim3((unsigned)((P.n1pad * 32 + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, d_desc1, (int)n1, (int)P.n1pad, e1);
        hipLaunchKernelGGL(ft_expand, dim3((unsigned)((P.n2pad * 32 + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, d_desc2, (int)n2, (int)P.n2pad, e2);
        hipLaunchKernelGGL(ft_train_terms, dim3((unsigned)((P.n2pad + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, reinterpret_cast<const uint4 *>(d_desc2), (int)n2, (int)P.n2pad, c2);
        hipLaunchKernelGGL(ft_knn2_mfma, dim3((unsigned)(P.n1pad / kQB), (unsigned)P.nchunks), dim3(kThreads), lds, st, e1, e2, c2, (int)n1, (int)P.n2pad, P.chunk, part, P.nchunks);
        hipLaunchKernelGGL(ft_knn2_merge_mfma, dim3((unsigned)((n1 + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, part, reinterpret_cast<const uint4 *>(d_desc1), (int)n1, P.nchunks, d_idx, d_dist);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "matcher launch failed: %s", hipGetErrorString(e));
        return SID_PM_OK;
    }
    int chunk, nchunks;
    plan(n1, n2, chunk, nchunks);
    const dim3 grid((unsigned)((n1 + kThreads - 1) / kThreads), (unsigned)nchunks);
    hipLaunchKernelGGL(ft_knn2_partial, grid, dim3(kThreads), 0, st, reinterpret_cast<const uint4 *>(d_desc1), (int)n1,
                       reinterpret_cast<const uint4 *>(d_desc2), (int)n2, chunk, reinterpret_cast<uint32_t *>(d_workspace), nchunks);
    hipLaunchKernelGGL(ft_knn2_merge, dim3(grid.x), dim3(kThreads), 0, st, reinterpret_cast<const uint32_t *>(d_workspace),
                       (int)n1, nchunks, d_idx, d_dist);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "matcher launch failed: %s", hipGetErrorString(e));
    return SID_PM_OK;
}

// device scratch of sid_ft_knn2: one grow-only block per device (sid_ft_release frees it)
static std::mutex g_ft_mu[16];
static unsigned char *g_ft_pool[16] = {nullptr};
static size_t g_ft_cap[16] = {0};

SID_EXPORT int sid_ft_release(int device)
{
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int d = 0; d < 16; ++d) {
        if (device >= 0 && d != (device & 15)) continue;
        std::lock_guard<std::mutex> lock(g_ft_mu[d]);
        if (g_ft_pool[d]) { (void)hipSetDevice(d); (void)hipFree(g_ft_pool[d]); g_ft_pool[d] = nullptr; g_ft_cap[d] = 0; }
    }
    (void)hipSetDevice(prev);
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
    // device scratch: one grow-only block per device, kept between calls (five hipMalloc / hipFree cost about as much as
    // the matching of the reference notebook's 24 000 x 23 000 case); calls on one device are serialised by its mutex
    std::mutex *mu = g_ft_mu;
    unsigned char **pool = g_ft_pool;
    size_t *pool_cap = g_ft_cap;
    std::lock_guard<std::mutex> lock(mu[device & 15]);
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t s1 = up((size_t)n1 * SID_FT_DESC_BYTES), s2 = up((size_t)std::max<int64_t>(n2, 1) * SID_FT_DESC_BYTES);
    const size_t sw = up((size_t)sid_ft_workspace_bytes(n1, n2)), so = up(sizeof(int32_t) * 2 * (size_t)n1);
    const size_t need = s1 + s2 + sw + 2 * so;
    if (pool_cap[device & 15] < need) {
        if (pool[device & 15]) (void)hipFree(pool[device & 15]);
        pool[device & 15] = nullptr; pool_cap[device & 15] = 0;
        step(hipMalloc(reinterpret_cast<void **>(&pool[device & 15]), need + need / 4));
        if (e == hipSuccess) pool_cap[device & 15] = need + need / 4;
    }
    uint8_t *b1 = pool[device & 15], *b2 = b1 ? b1 + s1 : nullptr, *ws = b1 ? b2 + s2 : nullptr;
    int32_t *bi = b1 ? reinterpret_cast<int32_t *>(ws + sw) : nullptr, *bd = b1 ? reinterpret_cast<int32_t *>(ws + sw + so) : nullptr;
    if (e == hipSuccess) {
        step(hipMemcpyAsync(b1, desc1, (size_t)n1 * SID_FT_DESC_BYTES, hipMemcpyHostToDevice, nullptr));
        if (n2 > 0) step(hipMemcpyAsync(b2, desc2, (size_t)n2 * SID_FT_DESC_BYTES, hipMemcpyHostToDevice, nullptr));
    }
    int rc = SID_PM_OK;
    if (e == hipSuccess) rc = sid_ft_knn2_device(b1, n1, b2, n2, bi, bd, ws, nullptr);
    if (e == hipSuccess && rc == SID_PM_OK) {
        step(hipMemcpyAsync(idx, bi, sizeof(int32_t) * 2 * (size_t)n1, hipMemcpyDeviceToHost, nullptr));
        step(hipMemcpyAsync(dist, bd, sizeof(int32_t) * 2 * (size_t)n1, hipMemcpyDeviceToHost, nullptr));
        step(hipStreamSynchronize(nullptr));
    } else {
        (void)hipStreamSynchronize(nullptr);                           // (nothing of a failed call stays in flight in the pool)
    }
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "sid_ft_knn2: %s", hipGetErrorString(e));
    return rc;
}
