// orb.hip - key points and 256-bit descriptors on gfx950 (C ABI: include/sid_orb.h; serves the interface of the
// reference's ftlib.find_key_points, ftlib.py:26-61, whose arithmetic is OpenCV's ORB and is not reproduced - see
// the header).  Every step is integer arithmetic with the specification of oracle/orb_oracle.py.
//
// Data flow per pyramid level: resample -> FAST-9 score map -> 3x3 non-maximum suppression + Harris response ->
// candidate list -> the level's share in (response desc, y, x) order (radix selection + ranking by counting, all on the
// device since round 4: detect_on_device; the host-side route of rounds 2-3 remains as SID_ORB_HOST_SELECT=1 and as the
// fallback for a selection that overflows) -> orientation (intensity centroid, 32 directions) -> binomial blur -> 256
// comparisons through the pre-rotated pattern.  One synchronisation per image.
// All kernels are streaming passes over uint8 images (one thread per pixel or per key point); the detector runs once
// per image and is a small part of the feature-tracking + pattern-matching chain (tools/ftpm_bench.py).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include <chrono>
#include <mutex>

#include "../../include/sid_orb.h"
#include "../../include/sid_pm.h"

#define SID_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local char g_err[256] = "";
int fail(int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = fail(SID_PM_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)

struct Cand { int32_t x, y; long long resp; };

// ---- level l of the pyramid from level 0: 16.16 fixed-point source coordinates, 8-bit bilinear weights ----
__global__ void k_resize(const uint8_t *src, int rows0, int cols0, long long stride0, uint8_t *dst, int rows, int cols,
                         unsigned long long step_x, unsigned long long step_y)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    long long fx = (long long)x * (long long)step_x + (long long)(step_x >> 1) - 32768;
    long long fy = (long long)y * (long long)step_y + (long long)(step_y >> 1) - 32768;
    const long long mx = (long long)(cols0 - 1) << 16, my = (long long)(rows0 - 1) << 16;
    fx = fx < 0 ? 0 : (fx > mx ? mx : fx);
    fy = fy < 0 ? 0 : (fy > my ? my : fy);
    const int x0 = (int)(fx >> 16), y0 = (int)(fy >> 16);
    const int wx = (int)((fx >> 8) & 255), wy = (int)((fy >> 8) & 255);
    const int x1 = x0 + 1 < cols0 ? x0 + 1 : cols0 - 1, y1 = y0 + 1 < rows0 ? y0 + 1 : rows0 - 1;
    const int a = src[y0 * stride0 + x0], b = src[y0 * stride0 + x1], c = src[y1 * stride0 + x0], d = src[y1 * stride0 + x1];
    const int top = a * (256 - wx) + b * wx, bot = c * (256 - wx) + d * wx;
    dst[(long long)y * cols + x] = (uint8_t)((top * (256 - wy) + bot * wy + 32768) >> 16);
}

// ---- FAST-9 score (0 = no corner at threshold t) ----
__constant__ int8_t c_ring[16][2] = {{0, -3}, {1, -3}, {2, -2}, {3, -1}, {3, 0}, {3, 1}, {2, 2}, {1, 3},
                                     {0, 3}, {-1, 3}, {-2, 2}, {-3, 1}, {-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}};

__global__ void k_fast(const uint8_t *img, int rows, int cols, int edge, int t, uint8_t *score)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    int sc = 0;
    if (x >= edge && x < cols - edge && y >= edge && y < rows - edge) {
        const uint8_t *p = img + (long long)y * cols + x;
        const int c = p[0];
        int d[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = (int)p[c_ring[i][1] * cols + c_ring[i][0]] - c;
        // quick reject: a 9-arc contains at least two of the four compass points on either side
        int best = -256;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            int mb = 255, md = 255;
#pragma unroll
            for (int k = 0; k < 9; ++k) { const int v = d[(s + k) & 15]; mb = v < mb ? v : mb; md = -v < md ? -v : md; }
            const int m = mb > md ? mb : md;
            best = m > best ? m : best;
        }
        sc = best > t ? best : 0;
    }
    score[(long long)y * cols + x] = (uint8_t)sc;
}

// one atomicAdd per wavefront instead of one per candidate (millions of appends to a single counter serialise)
__device__ __forceinline__ unsigned int wave_append(bool take, unsigned int *count)
{
    const unsigned long long mask = __ballot(take);
    if (mask == 0ull) return 0xffffffffu;
    const int lane = (int)(threadIdx.x & 63u), leader = __ffsll((long long)mask) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned int)__popcll(mask));
    base = (unsigned int)__shfl((int)base, leader);
    return take ? base + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull)) : 0xffffffffu;
}

// ---- 3x3 non-maximum suppression -> candidate list (x, y); the Harris response is filled in by k_harris, one
//      thread per candidate (the maxima are a few per cent of the pixels: computed in place they leave most lanes of
//      every wavefront idle through the 7x7 loop) ----
constexpr int kNmsRows = 16;                   // image rows per block of k_nms (256 columns wide)
__global__ void k_nms(const uint8_t *score, int rows, int cols, int edge, Cand *out, unsigned int *count, unsigned int cap)
{
    // the maxima of a 256 x 16 tile are collected in LDS (at most one per 2 x 2 pixels) and appended with ONE atomic per
    // block: a few million appends to a single counter take longer than everything else in the detector together
    __shared__ uint32_t list[256 * kNmsRows / 4 + 64];
    __shared__ unsigned int nlist, gbase;
    if (threadIdx.x == 0) nlist = 0;
    __syncthreads();
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < kNmsRows; ++k) {
        const int y = blockIdx.y * kNmsRows + k;
        bool peak = false;
        if (x >= edge && x < cols - edge && y >= edge && y < rows - edge) {
            const uint8_t *s = score + (long long)y * cols + x;
            const int v = s[0];
            peak = v != 0 && v > s[-1] && v > s[1] && v > s[-cols] && v > s[cols] && v > s[-cols - 1] && v > s[-cols + 1] &&
                   v > s[cols - 1] && v > s[cols + 1];
        }
        const unsigned int slot = wave_append(peak, &nlist);
        if (peak) list[slot] = ((uint32_t)y << 16) | (uint32_t)x;      // rows, cols <= 65535 (checked by the caller)
    }
    __syncthreads();
    const unsigned int nl = nlist;
    if (threadIdx.x == 0 && nl) gbase = atomicAdd(count, nl);
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < nl; i += blockDim.x) {
        const unsigned int o = gbase + i;
        if (o < cap) { out[o].x = (int32_t)(list[i] & 0xffffu); out[o].y = (int32_t)(list[i] >> 16); out[o].resp = 0; }
    }
}

__global__ void k_harris(const uint8_t *img, int cols, Cand *cand, unsigned int n)
{
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = cand[i].x, y = cand[i].y;
    int a = 0, b = 0, c = 0;                                           // <= 49 * 255^2: int32 is enough
    for (int dy = -3; dy <= 3; ++dy) {
        const uint8_t *p = img + (long long)(y + dy) * cols + x;
#pragma unroll
        for (int dx = -3; dx <= 3; ++dx) {
            const int ix = (int)p[dx + 1] - (int)p[dx - 1], iy = (int)p[dx + cols] - (int)p[dx - cols];
            a += ix * ix; b += iy * iy; c += ix * iy;
        }
    }
    const long long A = a, B = b, C = c;
    cand[i].resp = 25 * (A * B - C * C) - (A + B) * (A + B);
}

// ---- the n best candidates without sorting them all: order-preserving 32-bit key of the response (float32 of the
//      int64, so a < b never turns into key(a) > key(b)), three histogram passes (11 / 11 / 10 bits, most significant
//      first) find the key of rank n from the top, a compaction keeps every candidate at or above it.  Everything left
//      out has a strictly smaller response than everything kept; the host then orders the kept ones exactly. ----
__device__ __forceinline__ uint32_t resp_key(long long r)
{
    const uint32_t b = __float_as_uint((float)r);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

constexpr int kHistPer = 16;
__global__ void k_key_hist(const Cand *c, unsigned int n, uint32_t prefix, uint32_t prefix_mask, int shift, uint32_t digit_mask,
                           unsigned int *hist)
{
    // a block counts kHistPer candidates per thread in LDS and flushes the bins it touched
    __shared__ unsigned int h[2048];
    for (int b = threadIdx.x; b < 2048; b += blockDim.x) h[b] = 0;
    __syncthreads();
    const unsigned int i0 = blockIdx.x * blockDim.x * kHistPer + threadIdx.x;
#pragma unroll 4
    for (int u = 0; u < kHistPer; ++u) {
        const unsigned int i = i0 + u * blockDim.x;
        if (i < n) {
            const uint32_t k = resp_key(c[i].resp);
            if ((k & prefix_mask) == prefix) atomicAdd(&h[(k >> shift) & digit_mask], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 2048; b += blockDim.x) { const unsigned int v = h[b]; if (v) atomicAdd(&hist[b], v); }
}

__global__ void k_compact(const Cand *c, unsigned int n, uint32_t kmin, Cand *out, unsigned int *count, unsigned int cap)
{
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool take = i < n && resp_key(c[i].resp) >= kmin;
    const unsigned int k = wave_append(take, count);
    if (take && k < cap) out[k] = c[i];
}

// ---- orientation: intensity centroid over the disc of radius R, quantised to 32 directions; one wavefront per key
//      point (the lanes share the (2R+1)^2 pixels of the bounding square) ----
__global__ void k_orient(const uint8_t *img, int cols, const int32_t *kp /* [n][4]: x, y, level, dir */, int n, int R,
                         const int32_t *dirs, int32_t *dir_out)
{
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const int x = kp[4 * i], y = kp[4 * i + 1];
    const int w = 2 * R + 1, np = w * w;
    int m10 = 0, m01 = 0;                                              // <= R * 255 * (2R+1)^2 < 2^31 for R <= 32
    for (int q = lane; q < np; q += 64) {
        const int dy = q / w - R, dx = q - (dy + R) * w - R;
        if (dx * dx + dy * dy <= R * R) { const int v = img[(long long)(y + dy) * cols + x + dx]; m10 += dx * v; m01 += dy * v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
    if (lane == 0) {
        int best = 0; long long bv = (long long)m10 * dirs[0] + (long long)m01 * dirs[1];
        for (int b = 1; b < 32; ++b) { const long long v = (long long)m10 * dirs[2 * b] + (long long)m01 * dirs[2 * b + 1]; if (v > bv) { bv = v; best = b; } }
        dir_out[i] = best;
    }
}

// ---- 5x5 binomial blur (1 4 6 4 1 / 16 per axis, replicated borders, round half up once) ----
__global__ void k_blur(const uint8_t *img, int rows, int cols, uint8_t *out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= cols || y >= rows) return;
    const int w[5] = {1, 4, 6, 4, 1};
    int acc = 0;
#pragma unroll
    for (int j = -2; j <= 2; ++j) {
        const int yy = y + j < 0 ? 0 : (y + j >= rows ? rows - 1 : y + j);
        const uint8_t *p = img + (long long)yy * cols;
        int h = 0;
#pragma unroll
        for (int k = -2; k <= 2; ++k) { const int xx = x + k < 0 ? 0 : (x + k >= cols ? cols - 1 : x + k); h += w[k + 2] * p[xx]; }
        acc += w[j + 2] * h;
    }
    out[(long long)y * cols + x] = (uint8_t)((acc + 128) >> 8);
}

// ---- descriptor: 256 comparisons through the pattern of the key point's direction; one wavefront per key point ----
__global__ void k_describe(const uint8_t *blur, int cols, const int32_t *kp, const int32_t *dir, int n, const int8_t *pattern,
                           uint8_t *desc)
{
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const int x = kp[4 * i], y = kp[4 * i + 1];
    const int8_t *pt = pattern + (long long)dir[i] * 1024;
    const uint8_t *c = blur + (long long)y * cols + x;
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                      // lane handles bits 4 lane .. 4 lane + 3
        const int8_t *q = pt + 4 * (4 * lane + k);
        const int a = c[q[1] * cols + q[0]], b = c[q[3] * cols + q[2]];
        bits |= (a < b ? 1u : 0u) << k;
    }
    // two lanes make a byte (bits 8 j .. 8 j + 7, least significant first)
    const uint32_t hi = __shfl_down(bits, 1);
    if (!(lane & 1)) desc[(long long)i * 32 + (lane >> 1)] = (uint8_t)(bits | (hi << 4));
}

// ---------------------------------------------------------------------------------------------
// Round 4: selection and exact ordering ON THE DEVICE - a level no longer visits the host (seven synchronisations and a
// host sort per level before: candidate count, three histograms, compaction count, candidates, results).  Counts live in a
// DState block of device memory; kernels whose extent is a device-side count run a fixed grid and stride over it.
//   k_pick_d   one wavefront turns a 2048-bin histogram into the next digit of the key of rank `keep` from the top
//   k_rank_d   exact order (response descending, then y, then x) by COUNTING: a candidate's position is the number of
//              candidates before it - n^2 comparisons through LDS tiles, n = the ~keep candidates that survived the
//              selection (2.4 x 10^4 at level 0 of a 10^8-pixel image: ~0.1 ms), no sort network, no second buffer
//   k_emit_d   the level's share of key points -> the output arrays at the running total
// A level whose selection overflows its buffer (massive ties) raises DState::flags and the call is repeated with the
// host-side route (SID_ORB_HOST_SELECT=1 forces that route: A/B runs and the equality test).
// ---------------------------------------------------------------------------------------------
struct DState { unsigned int nc, nsel, n_lvl, total, prefix, above, flags, pad; };

__global__ void k_lvl_begin(DState *S) { S->nc = 0; S->nsel = 0; S->n_lvl = 0; S->prefix = 0; S->above = 0; }
__global__ void k_lvl_end(DState *S) { S->total += S->n_lvl; }

__global__ void k_harris_d(const uint8_t *img, int cols, Cand *cand, DState *S, unsigned int cap)
{
    const unsigned int n = S->nc;
    if (n > cap) { if (blockIdx.x == 0 && threadIdx.x == 0) S->flags |= 1u; return; }
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int x = cand[i].x, y = cand[i].y;
        int a = 0, b = 0, c = 0;
        for (int dy = -3; dy <= 3; ++dy) {
            const uint8_t *p = img + (long long)(y + dy) * cols + x;
#pragma unroll
            for (int dx = -3; dx <= 3; ++dx) {
                const int ix = (int)p[dx + 1] - (int)p[dx - 1], iy = (int)p[dx + cols] - (int)p[dx - cols];
                a += ix * ix; b += iy * iy; c += ix * iy;
            }
        }
        const long long A = a, B = b, C = c;
        cand[i].resp = 25 * (A * B - C * C) - (A + B) * (A + B);
    }
}

__global__ void k_key_hist_d(const Cand *c, const DState *S, unsigned int cap, int pass, unsigned int *hist)
{
    __shared__ unsigned int h[2048];
    const unsigned int n = S->nc <= cap ? S->nc : 0u;
    if (blockIdx.x * blockDim.x >= n) return;                          // (uniform)
    const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0;
    const uint32_t dmask = pass == 2 ? 1023u : 2047u, pmask = pass == 0 ? 0u : pass == 1 ? 0xffe00000u : 0xfffffc00u;
    const uint32_t prefix = S->prefix;
    for (int b = threadIdx.x; b < 2048; b += blockDim.x) h[b] = 0;
    __syncthreads();
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t k = resp_key(c[i].resp);
        if ((k & pmask) == prefix) atomicAdd(&h[(k >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 2048; b += blockDim.x) { const unsigned int v = h[b]; if (v) atomicAdd(&hist[b], v); }
}

// one wavefront: digit d (from the top) at which the number of candidates with a larger key reaches keep - the host loop
// `for (d = max; d > 0; --d) { if (above + hist[d] >= keep) break; above += hist[d]; }` of the host-side route; clears the
// histogram for the next pass
__global__ void k_pick_d(unsigned int *hist, DState *S, int pass, int want, long long max_out)
{
    const int lane = threadIdx.x;
    const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0;
    const unsigned int dmask = pass == 2 ? 1023u : 2047u;
    const long long room = max_out - (long long)S->total;
    const unsigned int keep = (unsigned int)(room < (long long)want ? (room > 0 ? room : 0) : (long long)want);
    unsigned int c[32], tot = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const unsigned int idx = 32u * lane + j;
        c[j] = idx <= dmask ? hist[idx] : 0u;
        hist[idx] = 0u;
        if (idx == 0u) c[j] = keep;                                    // (the loop above never tests digit 0: it ends there)
        tot += c[j];
    }
    unsigned int suf = tot;                                            // inclusive suffix sum over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned int o = (unsigned int)__shfl_down((int)suf, d); if (lane + d < 64) suf += o; }
    const unsigned int above0 = S->above;
    const unsigned long long ok = __ballot(above0 + suf >= keep);      // true for lane 0 at least
    const int L = 63 - __builtin_clzll(ok);
    if (lane == L) {
        unsigned int a = above0 + (suf - tot);
        int d = 0;
        bool found = false;
#pragma unroll
        for (int j = 31; j >= 0; --j) {
            if (!found) { if (a + c[j] >= keep) { d = 32 * lane + j; found = true; } else a += c[j]; }
        }
        S->prefix |= (uint32_t)d << shift; S->above = a;
    }
}

// a block owns a contiguous piece of the candidate list: it counts its takers, reserves their slots with ONE atomic on the
// global counter and writes them (a per-wavefront append - k_compact - serialises ~15 000 atomics on that one address: 120 us)
__global__ void k_compact_d(const Cand *c, DState *S, unsigned int cap, Cand *out, unsigned int sel_cap, unsigned int *rank)
{
    __shared__ unsigned int cnt, base, pos;
    const unsigned int n = S->nc <= cap ? S->nc : 0u;
    const uint32_t kmin = S->prefix;
    const unsigned int per = ((n + gridDim.x - 1) / gridDim.x + 255u) & ~255u, i0 = blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    if (i0 >= i1) return;                                              // (uniform)
    if (threadIdx.x == 0) { cnt = 0; pos = 0; }
    __syncthreads();
    unsigned int mine = 0;
    for (unsigned int i = i0 + threadIdx.x; i < i1; i += blockDim.x) mine += resp_key(c[i].resp) >= kmin ? 1u : 0u;
    if (mine) atomicAdd(&cnt, mine);
    __syncthreads();
    if (cnt == 0) return;
    if (threadIdx.x == 0) base = atomicAdd(&S->nsel, cnt);
    __syncthreads();
    for (unsigned int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const Cand v = c[i];
        if (resp_key(v.resp) >= kmin) {
            const unsigned int k = base + atomicAdd(&pos, 1u);
            if (k < sel_cap) { out[k] = v; rank[k] = 0u; }
        }
    }
}

// position of candidate i = number of candidates before it.  The list is read through UNIFORM addresses - scalar loads, the
// comparison operands sit in scalar registers, no LDS and no barrier - and cut into gridDim.y pieces whose counts are added
// up in rank[] (zeroed by k_compact_d).  (First version: one thread walked the whole list through LDS tiles - 2 ms at level 0.)
constexpr int kRankPieces = 32;
__global__ void k_rank_d(const Cand *__restrict__ sel, DState *S, unsigned int sel_cap, unsigned int *__restrict__ rank)
{
    const unsigned int n = S->nsel;
    if (n > sel_cap) { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) S->flags |= 2u; return; }
    if (blockIdx.x * blockDim.x >= n) return;                          // (uniform)
    const unsigned int per = (n + gridDim.y - 1) / gridDim.y, j0 = blockIdx.y * per, j1 = j0 + per < n ? j0 + per : n;
    if (j0 >= j1) return;
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    const Cand me = sel[i < n ? i : n - 1];
    const uint32_t myyx = ((uint32_t)me.y << 16) | (uint32_t)me.x;
    unsigned int r = 0;
    // branch-free, eight records per trip: the scalar loads of a trip are issued together (with `||` / `&&` the compiler emitted a
    // load, a wait and a branch per record)
    const uint4 *s4 = reinterpret_cast<const uint4 *>(sel);            // Cand = {x, y, resp lo, resp hi}
    auto before = [&](const uint4 o) -> unsigned int {
        const long long resp = (long long)(((unsigned long long)o.w << 32) | (unsigned long long)o.z);
        const uint32_t q = (o.y << 16) | o.x;
        return (unsigned int)((resp > me.resp) | ((resp == me.resp) & (q < myyx)));
    };
    unsigned int j = j0;
    for (; j + 8 <= j1; j += 8) {
        uint4 o[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = s4[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) r += before(o[u]);
    }
    for (; j < j1; ++j) r += before(s4[j]);
    if (i < n && r) atomicAdd(&rank[i], r);
}

__global__ void k_scatter_d(const Cand *__restrict__ sel, const DState *S, unsigned int sel_cap, const unsigned int *__restrict__ rank, Cand *__restrict__ sorted)
{
    const unsigned int n = S->nsel;
    if (n > sel_cap) return;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) sorted[rank[i]] = sel[i];
}

__global__ void k_emit_d(const Cand *sorted, DState *S, int want, long long max_out, int level, double scale,
                         int32_t *kp_all, float *xy_all, long long *resp_all)
{
    const long long room = max_out - (long long)S->total;
    long long n = (long long)S->nsel;
    if (S->flags) n = 0;
    if (n > (long long)want) n = want;
    if (n > room) n = room > 0 ? room : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) S->n_lvl = (unsigned int)n;
    const unsigned int base = S->total;
    for (long long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const Cand c = sorted[i];
        const long long o = (long long)base + i;
        kp_all[4 * o] = c.x; kp_all[4 * o + 1] = c.y; kp_all[4 * o + 2] = level; kp_all[4 * o + 3] = 0;
        xy_all[2 * o] = (float)((double)c.x * scale); xy_all[2 * o + 1] = (float)((double)c.y * scale);
        resp_all[o] = c.resp;
    }
}

// orientation / descriptor of the level's key points at kp_all[total ..]; the direction goes into kp_all[..][3]
__global__ void k_orient_d(const uint8_t *img, int cols, int32_t *kp_all, const DState *S, int R, const int32_t *dirs)
{
    const int n = (int)S->n_lvl;
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += gridDim.x * (blockDim.x >> 6)) {
        int32_t *kp = kp_all + 4 * ((long long)S->total + i);
        const int x = kp[0], y = kp[1];
        const int w = 2 * R + 1, np = w * w;
        int m10 = 0, m01 = 0;
        for (int q = lane; q < np; q += 64) {
            const int dy = q / w - R, dx = q - (dy + R) * w - R;
            if (dx * dx + dy * dy <= R * R) { const int v = img[(long long)(y + dy) * cols + x + dx]; m10 += dx * v; m01 += dy * v; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
        if (lane == 0) {
            int best = 0; long long bv = (long long)m10 * dirs[0] + (long long)m01 * dirs[1];
            for (int b = 1; b < 32; ++b) { const long long v = (long long)m10 * dirs[2 * b] + (long long)m01 * dirs[2 * b + 1]; if (v > bv) { bv = v; best = b; } }
            kp[3] = best;
        }
    }
}

__global__ void k_describe_d(const uint8_t *blur, int cols, const int32_t *kp_all, const DState *S, const int8_t *pattern, uint8_t *desc_all)
{
    const int n = (int)S->n_lvl;
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += gridDim.x * (blockDim.x >> 6)) {
        const long long o = (long long)S->total + i;
        const int32_t *kp = kp_all + 4 * o;
        const int x = kp[0], y = kp[1];
        const int8_t *pt = pattern + (long long)kp[3] * 1024;
        const uint8_t *c = blur + (long long)y * cols + x;
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int8_t *q = pt + 4 * (4 * lane + k);
            const int a = c[q[1] * cols + q[0]], b = c[q[3] * cols + q[2]];
            bits |= (a < b ? 1u : 0u) << k;
        }
        const uint32_t hi = __shfl_down(bits, 1);
        if (!(lane & 1)) desc_all[o * 32 + (lane >> 1)] = (uint8_t)(bits | (hi << 4));
    }
}

}  // namespace

namespace {

// Workspaces: device buffers, a private stream and pinned staging, kept between calls (a call used to spend ~3 ms of its 17
// in twelve hipMalloc / hipFree and ran on the null stream).  A call takes a free workspace of its device or makes one, so
// two host threads - the two images of a pair (ftlib.track) - run side by side on streams of their own.
struct OrbWs {
    int device = -1; bool busy = false;
    hipStream_t stream = nullptr;
    unsigned char *blk = nullptr; size_t cap = 0;
    unsigned int *h_small = nullptr;                                   // pinned: a counter / a 2048-bin histogram
};
std::mutex g_ws_mu;
std::vector<OrbWs *> g_ws;

OrbWs *ws_acquire(int device)
{
    std::lock_guard<std::mutex> lock(g_ws_mu);
    for (OrbWs *w : g_ws) if (w->device == device && !w->busy) { w->busy = true; return w; }
    OrbWs *w = new (std::nothrow) OrbWs();
    if (!w) return nullptr;
    w->device = device; w->busy = true;
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&w->h_small), 2048 * sizeof(unsigned int), hipHostMallocDefault) != hipSuccess) {
        if (w->stream) (void)hipStreamDestroy(w->stream);
        delete w;
        return nullptr;
    }
    g_ws.push_back(w);
    return w;
}
void ws_free(OrbWs *w)                                                 // (g_ws_mu held; the workspace is idle)
{
    int prev = 0; (void)hipGetDevice(&prev); (void)hipSetDevice(w->device);
    if (w->stream) { (void)hipStreamSynchronize(w->stream); (void)hipStreamDestroy(w->stream); }
    if (w->blk) (void)hipFree(w->blk);
    if (w->h_small) (void)hipHostFree(w->h_small);
    (void)hipSetDevice(prev);
    delete w;
}
constexpr int kMaxIdleWs = 2;                                          // idle workspaces kept per device (the two images of a pair)
void ws_release(OrbWs *w)
{
    if (!w) return;
    std::lock_guard<std::mutex> lock(g_ws_mu);
    w->busy = false;
    int idle = 0;
    for (OrbWs *o : g_ws) if (o->device == w->device && !o->busy) ++idle;
    if (idle > kMaxIdleWs) {                                           // more threads than that detected at once: do not keep their ~7 B/pixel each
        for (size_t k = 0; k < g_ws.size(); ++k) if (g_ws[k] == w) { g_ws.erase(g_ws.begin() + (long)k); break; }
        ws_free(w);
    }
}
hipError_t ws_reserve(OrbWs *w, size_t bytes)
{
    if (w->cap >= bytes) return hipSuccess;
    if (w->blk) (void)hipFree(w->blk);
    w->blk = nullptr; w->cap = 0;
    const hipError_t e = hipMalloc(reinterpret_cast<void **>(&w->blk), bytes);
    if (e == hipSuccess) w->cap = bytes;
    return e;
}
struct WsCarver {
    unsigned char *base; size_t off = 0;
    template <typename T> T *take(size_t n) { T *r = reinterpret_cast<T *>(base + off); off += (n * sizeof(T) + 255) / 256 * 256; return r; }
};
size_t up256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace

// Every level on the stream without a visit to the host (kernels above); one synchronisation at the end.  *overflow: a level's
// selection did not fit its buffer - nothing was written to the outputs, the caller repeats the call with the host-side route.
struct DevBufs {
    uint8_t *img0, *lvl, *aux, *desc; Cand *cand, *sel; int8_t *pat; int32_t *dirs, *kp; unsigned int *hist;
    float *xy; long long *resp; DState *state; unsigned int *rank;
};
int detect_on_device(OrbWs *ws, const DevBufs &B, int64_t rows, int64_t cols, const sid_orb_params *P, const std::vector<double> &sc,
                     const std::vector<int> &lr, const std::vector<int> &lc, const std::vector<int> &want, unsigned int sel_cap,
                     float *xy, int32_t *meta, int64_t *response, uint8_t *desc, int64_t max_out, int64_t *total_out, bool *overflow)
{
    int rc = SID_PM_OK;
    hipStream_t st = ws->stream;
    const int L = P->n_levels, edge = P->edge_threshold, R = P->patch_size / 2;
    *overflow = false; *total_out = 0;
    HIP_TRY(hipMemsetAsync(B.state, 0, sizeof(DState), st));
    HIP_TRY(hipMemsetAsync(B.hist, 0, 2048 * sizeof(unsigned int), st));
    for (int l = 0; l < L; ++l) {
        const int r = lr[l], c = lc[l];
        if (r <= 2 * edge || c <= 2 * edge || want[l] <= 0) continue;
        const dim3 blk(256), grd((unsigned)((c + 255) / 256), (unsigned)r);
        const uint8_t *lvl = B.img0;
        if (l > 0) {
            const unsigned long long sx = ((unsigned long long)cols << 16) / (unsigned long long)c,
                                     sy = ((unsigned long long)rows << 16) / (unsigned long long)r;
            hipLaunchKernelGGL(k_resize, grd, blk, 0, st, B.img0, (int)rows, (int)cols, (long long)cols, B.lvl, r, c, sx, sy);
            lvl = B.lvl;
        }
        hipLaunchKernelGGL(k_lvl_begin, dim3(1), dim3(1), 0, st, B.state);
        hipLaunchKernelGGL(k_fast, grd, blk, 0, st, lvl, r, c, edge, P->fast_threshold, B.aux);
        const unsigned int cap = (unsigned int)((size_t)r * c / 4 + 16);
        hipLaunchKernelGGL(k_nms, dim3(grd.x, (unsigned)((r + kNmsRows - 1) / kNmsRows)), blk, 0, st, B.aux, r, c, edge, B.cand, &B.state->nc, cap);
        // candidates are a few per cent of the pixels; the grids below stride over the device-side count
        const unsigned int gcand = (unsigned int)std::min<size_t>(((size_t)cap + 255) / 256, 4096);
        hipLaunchKernelGGL(k_harris_d, dim3(gcand), blk, 0, st, lvl, c, B.cand, B.state, cap);
        for (int pass = 0; pass < 3; ++pass) {
            hipLaunchKernelGGL(k_key_hist_d, dim3(std::min(gcand, 1024u)), blk, 0, st, B.cand, B.state, cap, pass, B.hist);
            hipLaunchKernelGGL(k_pick_d, dim3(1), dim3(64), 0, st, B.hist, B.state, pass, want[l], (long long)max_out);
        }
        hipLaunchKernelGGL(k_compact_d, dim3(std::min(gcand, 512u)), blk, 0, st, B.cand, B.state, cap, B.sel, sel_cap, B.rank);
        // (the ordered list goes where the level's candidates were: at most as many as there were candidates)
        const unsigned int nsel_bound = std::min(sel_cap, cap);
        hipLaunchKernelGGL(k_rank_d, dim3((nsel_bound + 255) / 256, kRankPieces), blk, 0, st, B.sel, B.state, nsel_bound, B.rank);
        hipLaunchKernelGGL(k_scatter_d, dim3(std::min((nsel_bound + 255) / 256, 1024u)), blk, 0, st, B.sel, B.state, nsel_bound, B.rank, B.cand);
        const unsigned int gkp = (unsigned int)std::min<int64_t>(((int64_t)want[l] + 3) / 4, 65535);
        hipLaunchKernelGGL(k_emit_d, dim3(std::min<unsigned int>((unsigned int)(want[l] + 255) / 256, 1024u)), blk, 0, st, B.cand, B.state, want[l],
                           (long long)max_out, l, sc[l], B.kp, B.xy, B.resp);
        hipLaunchKernelGGL(k_orient_d, dim3(gkp), blk, 0, st, lvl, c, B.kp, B.state, R, B.dirs);
        hipLaunchKernelGGL(k_blur, grd, blk, 0, st, lvl, r, c, B.aux);                    // the score map is no longer needed
        hipLaunchKernelGGL(k_describe_d, dim3(gkp), blk, 0, st, B.aux, c, B.kp, B.state, B.pat, B.desc);
        hipLaunchKernelGGL(k_lvl_end, dim3(1), dim3(1), 0, st, B.state);
        HIP_TRY(hipGetLastError());
    }
    {
        DState *hs = reinterpret_cast<DState *>(ws->h_small);
        HIP_TRY(hipMemcpyAsync(hs, B.state, sizeof(DState), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (hs->flags) { *overflow = true; return SID_PM_OK; }
        const int64_t n = (int64_t)hs->total;
        if (n > 0) {
            HIP_TRY(hipMemcpyAsync(xy, B.xy, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(desc, B.desc, (size_t)n * 32, hipMemcpyDeviceToHost, st));
            if (meta) HIP_TRY(hipMemcpyAsync(meta, B.kp, (size_t)n * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            if (response) HIP_TRY(hipMemcpyAsync(response, B.resp, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        *total_out = n;
    }
done:
    return rc;
}

SID_EXPORT const char *sid_orb_last_error(void) { return g_err; }

SID_EXPORT int sid_orb_release(int device)
{
    std::lock_guard<std::mutex> lock(g_ws_mu);
    for (size_t k = 0; k < g_ws.size();) {
        OrbWs *w = g_ws[k];
        if (!w->busy && (device < 0 || w->device == device)) { g_ws.erase(g_ws.begin() + (long)k); ws_free(w); }
        else ++k;
    }
    return SID_PM_OK;
}

SID_EXPORT int sid_orb_detect(int device, const uint8_t *img, int64_t rows, int64_t cols, int64_t stride,
                              const sid_orb_params *P, const int8_t *pattern, const int32_t *dirs,
                              float *xy, int32_t *meta, int64_t *response, uint8_t *desc, int64_t max_out, int64_t *n_out)
{
    if (!img || !P || !pattern || !dirs || !xy || !desc || !n_out || max_out < 0) return fail(SID_PM_ERR_ARG, "null argument");
    if (rows < 1 || cols < 1 || stride < cols || rows > 65535 || cols > 65535) return fail(SID_PM_ERR_ARG, "bad image shape/stride");
    if (P->n_levels < 1 || P->n_levels > 16 || P->edge_threshold < 16 || P->patch_size < 2 || P->patch_size > 200 || P->patch_size / 2 + 1 > P->edge_threshold ||
        P->n_features < 0 || P->fast_threshold < 1 || P->fast_threshold > 254 || !(P->scale_factor > 1.0f))
        return fail(SID_PM_ERR_ARG, "bad detector parameters (edge_threshold >= 16 and >= patch_size / 2 + 1, 1..16 levels, scale > 1)");
    *n_out = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(SID_PM_ERR_NODEVICE, "no such HIP device");
    int prev = 0; (void)hipGetDevice(&prev); (void)hipSetDevice(device);
    int rc = SID_PM_OK;
    const int L = P->n_levels, edge = P->edge_threshold, R = P->patch_size / 2;
    // SID_ORB_VERBOSE=1: wall-clock milliseconds per stage on stderr (synchronises after every stage)
    const bool verbose = getenv("SID_ORB_VERBOSE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    OrbWs *ws = nullptr;
    hipStream_t st = nullptr;
    auto tick = [&](const char *what, int level) {
        if (!verbose) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sid_orb] level %d %-22s %8.3f ms\n", level, what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    uint8_t *d_img0 = nullptr, *d_lvl = nullptr, *d_aux = nullptr, *d_desc = nullptr;
    Cand *d_cand = nullptr, *d_sel = nullptr; unsigned int *d_count = nullptr, *d_hist = nullptr;
    float *d_xy = nullptr; long long *d_resp = nullptr; DState *d_state = nullptr; unsigned int *d_rank = nullptr;
    const unsigned int sel_cap = (unsigned int)std::min<int64_t>((int64_t)4 * (int64_t)P->n_features + 65536, (int64_t)1 << 26);
    int8_t *d_pat = nullptr; int32_t *d_dirs = nullptr, *d_kp = nullptr, *d_dir = nullptr;
    std::vector<Cand> cand;
    std::vector<int32_t> kp, dir_h;
    std::vector<uint8_t> desc_h;
    int64_t total = 0;
    {
        // level geometry and the per-level share of n_features (OpenCV's geometric split), all in double
        std::vector<double> sc(L); std::vector<int> lr(L), lc(L), want(L);
        double s = 1.0;
        for (int l = 0; l < L; ++l) { sc[l] = s; lr[l] = (int)floor((double)rows / s + 0.5); lc[l] = (int)floor((double)cols / s + 0.5); s *= (double)P->scale_factor; }
        {
            const double factor = 1.0 / (double)P->scale_factor;
            double nd = (double)P->n_features * (1.0 - factor) / (1.0 - pow(factor, (double)L));
            int sum = 0;
            for (int l = 0; l < L - 1; ++l) { want[l] = (int)floor(nd + 0.5); sum += want[l]; nd *= factor; }
            want[L - 1] = std::max(P->n_features - sum, 0);
        }
        const size_t area0 = (size_t)rows * cols;
        const size_t nkp = (size_t)std::max<int64_t>(std::min<int64_t>((int64_t)P->n_features, max_out), 1);   // most key points of one level
        ws = ws_acquire(device);
        if (!ws) { rc = fail(SID_PM_ERR_NOMEM, "no detector workspace"); goto done; }
        st = ws->stream;
        HIP_TRY(ws_reserve(ws, 3 * up256(area0) + up256((area0 / 4 + 16) * sizeof(Cand)) + up256((size_t)sel_cap * sizeof(Cand)) + up256(32 * 1024) +
                               up256(nkp * 4 * sizeof(int32_t)) + up256(nkp * sizeof(int32_t)) + up256(nkp * 32) + 4 * 256 + up256(2048 * sizeof(unsigned int)) +
                               up256(nkp * 2 * sizeof(float)) + up256(nkp * sizeof(long long)) + 256 + up256((size_t)sel_cap * sizeof(unsigned int))));
        {
            WsCarver cv{ws->blk};
            d_img0 = cv.take<uint8_t>(area0); d_lvl = cv.take<uint8_t>(area0); d_aux = cv.take<uint8_t>(area0);
            d_cand = cv.take<Cand>(area0 / 4 + 16); d_sel = cv.take<Cand>(sel_cap); d_pat = cv.take<int8_t>(32 * 1024);
            d_kp = cv.take<int32_t>(nkp * 4); d_dir = cv.take<int32_t>(nkp); d_desc = cv.take<uint8_t>(nkp * 32);
            d_count = cv.take<unsigned int>(1); d_dirs = cv.take<int32_t>(64); d_hist = cv.take<unsigned int>(2048);
            d_xy = cv.take<float>(nkp * 2); d_resp = cv.take<long long>(nkp); d_state = cv.take<DState>(1); d_rank = cv.take<unsigned int>(sel_cap);
        }
        HIP_TRY(hipMemcpy2DAsync(d_img0, (size_t)cols, img, (size_t)stride, (size_t)cols, (size_t)rows, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_pat, pattern, 32 * 1024, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_dirs, dirs, 64 * sizeof(int32_t), hipMemcpyHostToDevice, st));
        tick("workspace + upload", -1);
        // the levels on the device from end to end (SID_ORB_HOST_SELECT=1, or the per-stage timing of SID_ORB_VERBOSE: the
        // host-side selection and sort below; also the route of a call whose selection overflowed)
        if (!verbose && getenv("SID_ORB_HOST_SELECT") == nullptr) {
            const DevBufs B{d_img0, d_lvl, d_aux, d_desc, d_cand, d_sel, d_pat, d_dirs, d_kp, d_hist, d_xy, d_resp, d_state, d_rank};
            bool overflow = false;
            rc = detect_on_device(ws, B, rows, cols, P, sc, lr, lc, want, sel_cap, xy, meta, response, desc, max_out, &total, &overflow);
            if (rc != SID_PM_OK || !overflow) { *n_out = rc == SID_PM_OK ? total : 0; goto done; }
            total = 0;
        }
        for (int l = 0; l < L && total < max_out; ++l) {
            const int r = lr[l], c = lc[l];
            if (r <= 2 * edge || c <= 2 * edge || want[l] <= 0) continue;
            const dim3 blk(256), grd((unsigned)((c + 255) / 256), (unsigned)r);
            const uint8_t *lvl = d_img0;
            if (l > 0) {
                const unsigned long long sx = ((unsigned long long)cols << 16) / (unsigned long long)c,
                                         sy = ((unsigned long long)rows << 16) / (unsigned long long)r;
                hipLaunchKernelGGL(k_resize, grd, blk, 0, st, d_img0, (int)rows, (int)cols, (long long)cols, d_lvl, r, c, sx, sy);
                lvl = d_lvl;
            }
            hipLaunchKernelGGL(k_fast, grd, blk, 0, st, lvl, r, c, edge, P->fast_threshold, d_aux);
            HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned int), st));
            const unsigned int cap = (unsigned int)((size_t)r * c / 4 + 16);
            hipLaunchKernelGGL(k_nms, dim3(grd.x, (unsigned)((r + kNmsRows - 1) / kNmsRows)), blk, 0, st, d_aux, r, c, edge, d_cand, d_count, cap);
            unsigned int nc = 0;
            HIP_TRY(hipMemcpyAsync(ws->h_small, d_count, sizeof nc, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            nc = ws->h_small[0];
            tick("resize + fast + nms", l);
            if (nc > cap) { rc = fail(SID_PM_ERR_HIP, "candidate list overflow (cannot happen: one maximum per 2x2 block)"); goto done; }
            if (nc == 0) continue;
            hipLaunchKernelGGL(k_harris, dim3((nc + 255) / 256), dim3(256), 0, st, lvl, c, d_cand, nc);
            const Cand *d_src = d_cand;
            const int64_t keep = std::min<int64_t>(want[l], max_out - total);
            if ((int64_t)nc > 2 * keep + 1024) {
                // key of rank `keep` from the top, digit by digit
                uint32_t prefix = 0, pmask = 0;
                unsigned int above = 0;                                // candidates with a larger key than the prefix so far
                const int shifts[3] = {21, 10, 0}; const uint32_t dmask[3] = {2047u, 2047u, 1023u};
                std::vector<unsigned int> hist(2048);
                for (int pass = 0; pass < 3; ++pass) {
                    HIP_TRY(hipMemsetAsync(d_hist, 0, 2048 * sizeof(unsigned int), st));
                    hipLaunchKernelGGL(k_key_hist, dim3((nc + 256 * kHistPer - 1) / (256 * kHistPer)), dim3(256), 0, st, d_cand, nc, prefix, pmask, shifts[pass], dmask[pass], d_hist);
                    HIP_TRY(hipMemcpyAsync(ws->h_small, d_hist, 2048 * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                    memcpy(hist.data(), ws->h_small, 2048 * sizeof(unsigned int));
                    int d = (int)dmask[pass];
                    for (; d > 0; --d) { if ((int64_t)above + hist[(size_t)d] >= keep) break; above += hist[(size_t)d]; }
                    prefix |= (uint32_t)d << shifts[pass]; pmask |= dmask[pass] << shifts[pass];
                }
                HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned int), st));
                hipLaunchKernelGGL(k_compact, dim3((nc + 255) / 256), dim3(256), 0, st, d_cand, nc, prefix, d_sel, d_count, sel_cap);
                unsigned int ns = 0;
                HIP_TRY(hipMemcpyAsync(ws->h_small, d_count, sizeof ns, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                ns = ws->h_small[0];
                if (ns <= sel_cap && (int64_t)ns >= keep) { d_src = d_sel; nc = ns; }   // (else: massive ties - order them all)
            }
            tick("select", l);
            cand.resize(nc);
            HIP_TRY(hipMemcpyAsync(cand.data(), d_src, (size_t)nc * sizeof(Cand), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            std::sort(cand.begin(), cand.end(), [](const Cand &p, const Cand &q) {
                if (p.resp != q.resp) return p.resp > q.resp;
                if (p.y != q.y) return p.y < q.y;
                return p.x < q.x;
            });
            tick("copy + host sort", l);
            int64_t n = std::min<int64_t>(std::min<int64_t>(nc, want[l]), max_out - total);
            kp.resize((size_t)(4 * n));
            for (int64_t i = 0; i < n; ++i) { kp[4 * i] = cand[(size_t)i].x; kp[4 * i + 1] = cand[(size_t)i].y; kp[4 * i + 2] = l; kp[4 * i + 3] = 0; }
            if ((size_t)n > nkp) { rc = fail(SID_PM_ERR_HIP, "level key points exceed the workspace (cannot happen)"); goto done; }
            HIP_TRY(hipMemcpyAsync(d_kp, kp.data(), (size_t)(4 * n) * sizeof(int32_t), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_orient, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, lvl, c, d_kp, (int)n, R, d_dirs, d_dir);
            hipLaunchKernelGGL(k_blur, grd, blk, 0, st, lvl, r, c, d_aux);                // the score map is no longer needed
            hipLaunchKernelGGL(k_describe, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, d_aux, c, d_kp, d_dir, (int)n, d_pat, d_desc);
            HIP_TRY(hipGetLastError());
            dir_h.resize((size_t)n); desc_h.resize((size_t)n * 32);
            HIP_TRY(hipMemcpyAsync(dir_h.data(), d_dir, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(desc_h.data(), d_desc, (size_t)n * 32, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (int64_t i = 0; i < n; ++i) {
                const int64_t o = total + i;
                xy[2 * o] = (float)((double)cand[(size_t)i].x * sc[l]);
                xy[2 * o + 1] = (float)((double)cand[(size_t)i].y * sc[l]);
                if (meta) { meta[4 * o] = cand[(size_t)i].x; meta[4 * o + 1] = cand[(size_t)i].y; meta[4 * o + 2] = l; meta[4 * o + 3] = dir_h[(size_t)i]; }
                if (response) response[o] = cand[(size_t)i].resp;
            }
            memcpy(desc + total * 32, desc_h.data(), (size_t)n * 32);
            total += n;
            tick("orient + describe", l);
        }
        *n_out = total;
    }
done:
    if (st) (void)hipStreamSynchronize(st);                            // (nothing of this call may still be in flight in a workspace that is free)
    ws_release(ws);
    (void)hipSetDevice(prev);
    return rc;
}
