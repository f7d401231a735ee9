// pm_kernel_mfma.hip - the pattern-matching kernel on the gfx950 matrix cores.
//
// Operator: reference pmlib.py:117-212, :36-59; NCC specification in DESIGN.md section 3.  The 34x34 uint8
// correlation runs on v_mfma_i32_16x16x64_i8 as an implicit correlation - no im2col is materialised:
//
//   bytes are re-centred (w' = w ^ 0x80, t' = t ^ 0x80 as int8); numer, dI and dT are covariances
//   and do not change under the shift, so the integer sums stay exact and the spec is untouched.
//
//   sweep ("angle-major"):  D[slot][x] += A[slot][c] * B[c][x]
//       M = 16 template slots: up to 15 trial angles + one all-ones template (gives sum w' = S_I')
//       N = 16 adjacent placements x0..x0+15 of one output row y
//       K = 64 window columns c of window row rho = y + i ; A[slot][c] = T'_slot[i][c] (0 for c >= s)
//     B[c][x] = W'[rho][x0 + x + c] is the same for every (y, i) with y + i = rho, so a wavefront
//     keeps a band of BAND output rows in accumulators and a ring of BAND template-row fragments in
//     registers: one window fragment built from LDS (5 ds_read_b32 + 4 v_alignbyte_b32) feeds BAND
//     MFMAs.  BAND = 4 with three wavefronts per SIMD (168 VGPRs), BAND = 8 in the instantiation for the
//     LDS class that fits two workgroups per CU (256 VGPRs): the sweep is LDS-bandwidth-bound, and the
//     taller band halves the LDS bytes per MFMA.
//
//   winner ("row-major"):   the NCC matrix of the best angle is recomputed with M = 16 output rows
//     (A[m][c] = T'_best[rho - y0 - m][c]) and a second all-ones operand for S_I'.
//
//   S_II' = sum w'^2 comes from two running-sum passes over LDS (columns, then rows).
//   The double-precision normalisation of the spec is evaluated only for arg-max candidates picked
//   by a float32 pre-filter (|r~ - r| <= 4e-7 << margin 1e-5) and for the winner's matrix.
//
//   templates: for an integral template centre (what the reference's driver always passes) the rotated
//   nearest-neighbour sample positions depend only on the angle; the host precomputes them once per
//   set_points as LDS patch offsets (pm_capi.hip make_samp) and the kernel gathers through that table;
//   any other centre, and templates at the image border, are sampled on the fly.
//
// One workgroup per grid point; 256 or 768 threads by LDS-footprint class (see kMaxBlockM).  The phases
// other than the sweep are separate non-inlined functions sharing a geometry block in LDS, so each has its
// own register allocation; throughput is set by the latency of a workgroup's phase chain.
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <mutex>
#include <set>
#include <utility>
#include <math.h>
#include <type_traits>
#include "pm_kernel.h"

namespace sid {

namespace {

typedef uint32_t u32;
typedef int v4i __attribute__((ext_vector_type(4)));

// Threads per grid point are chosen at launch: 256, or 768 for the LDS-footprint class that fits one
// point per CU only (12 wavefronts = 3 per SIMD, the register budget).
#ifdef SID_OCC4_TU
// pm_kernel_rp_occ4.hip compiles this file a second time for the launches whose LDS footprint fits FOUR workgroups per CU
// (row-pair kernel with slot groups, borders 20 and 21): 128 VGPRs, 256 threads.  A translation unit of its own because
// the register budget of a kernel is handed down to the non-inlined phase functions it calls: instantiated side by side,
// the 128-VGPR kernels slowed the 168-VGPR ones by 0.9 %.
constexpr int kMaxBlockM = 256;
constexpr int kOccM = 4;
#else
constexpr int kMaxBlockM = 768;
constexpr int kOccM = 3;             // wavefronts per SIMD the register allocation must allow
#endif
#define kBlockM ((int)blockDim.x)
#define kWavesM ((int)(blockDim.x >> 6))
constexpr int kSlots = 16;           // MFMA M: 15 angles + ones
constexpr int kAnglesPerGroup = kRpGroup;
constexpr float kMargin = 1e-5f;     // pre-filter margin (see header)

struct MiscM {                       // LDS offset 0, kMiscMfmaBytes reserved
    float red_f[16];
    int red_i[16];
    u32 sel_key, sel_cle;
    int zero_flag;
    int best_key; float best_val;
    u32 gmax_key;                    // block-wide running max of the float32 estimates (ordered key)
    u32 qcount;                      // candidate queue fill
    int pad_;
    double gw[5];                    // hes_smth: Gaussian taps (PMArgs::gauss_w)
    int isT[kSlots], isTT[kSlots];   // integer template sums of the current group
    double rot[kSlots][4];           // cos, sin, tcT0, tcT1 of the current group's angles
    double rTd[kMaxAngles];          // 1/sqrt(dT) per angle
    double sTd[kMaxAngles];          // sum t' per angle
    int constT[kMaxAngles];
    // row-pair kernel: float32 pre-filter terms of the current group's slots (pm_kernel_rp.inc), 16-byte aligned:
    // a lane fetches the terms of its four slots with one ds_read_b128 each
    __attribute__((aligned(16))) int nmTi[kSlots];      // -round(S_T' / N)
    __attribute__((aligned(16))) float ncTf[kSlots];    // -(S_T' - N round(S_T' / N)) / N
    __attribute__((aligned(16))) float aTf[kSlots];     // N / sqrt(dT); NaN: constant template; 0: dead slot
    __attribute__((aligned(16))) float biasf[kSlots];   // 0, or -inf for a dead slot
    float maxA;
    int any_const;
};

// Per-point geometry, written once by thread 0 and read by every phase (the phases are separate
// non-inlined functions so that each gets its own register allocation; their shared state is here).
struct Geo {
    int wh, ww, rh, rw, npos, wpitch, arow, s, K;
    int gpitch, arow0, tab_rows, ones_slot;   // operand-table geometry (mfma_lds_layout) and the all-ones slot: 15, or 7 paired
    int win_off, sii_off, u_off, patch_off, ppitch, pdim, pradius, queue_off, trow_bytes;
    int pr0, pc0;                    // patch origin on image 1
    u32 win_magic;                   // floor(2^32 / (wpitch/4)) + 1: idx / (wpitch/4) == umulhi(idx, win_magic) for idx < 2^16
    u32 patch_magic, rw_magic;       // same for ppitch/4 and for rw
    int band;                        // output rows per sweep work item (kernel template parameter)
    int wp_off, wp_pitch, wp_rows, strip_off, wrows, npair, nsingle;   // row-pair kernel (RpLdsLayout)
    int queue_cap;                   // row-pair kernel: entries of the candidate queue (RpLdsLayout::queue_cap)
    int hes_off, ccm_off;            // Hessian magnitudes and NCC matrix of the winning angle (f32 per placement)
    int si_off;                      // row-pair kernel: sum w' per placement kept from the sweep for the winner (0: none)
    int gsi;                         // ... the same in the point's block of global memory, behind sum w'^2: offset in u32 entries (0: none)
    int hist_off;                    // 5 KB behind the NCC matrix of the winning angle for ph_hessian_fast (0: none - the general ph_hessian runs)
    int gsa, gsa_rw;                 // kept accumulators (PMArgs::gs_keep_acc): offset of the table in the point's block (u32 entries; 0: none) and its row pitch in placements
    int blk, blk_xcd;                // recycled block of this point and its home word of the free lists (PMArgs::ring; blk -1: none)
    int lin;                         // rot_order = 1 (SID_PM_ROT_ORDER1, pmlib.py:89): templates sampled bilinearly, every sample through sample_exact
    long long r0, c0;                // window origin on image 2
    double c1, r1, nd;
    unsigned long long big;          // row-pair kernel, big layouts (rp_lds_layout): the point's block of global memory
    unsigned long long pre;          // rot_order 2..5: this point's K templates in global memory [K][s][s], sampled from the spline coefficients
                                     // of image 1 by lw_presample (pm_large.hip); lin is then 2 and sample_exact loads them (0: none)
};
// per-placement table at `off`: LDS, or - big layouts of the row-pair kernel - the point's block of global memory
template <bool BIG, typename T>
__device__ __forceinline__ T *tab_ptr(const Geo &G, unsigned char *smem, int off)
{
    if constexpr (BIG) return reinterpret_cast<T *>(reinterpret_cast<unsigned char *>(G.big) + (unsigned)off);
    else return reinterpret_cast<T *>(smem + off);
}
constexpr int kGeoOff = 2432;
static_assert(sizeof(MiscM) <= kGeoOff, "misc header too large");
static_assert(kGeoOff + sizeof(Geo) <= kMiscMfmaBytes, "geometry block does not fit the LDS header");

#define SID_PHASE_LOCALS                                                                        \
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                         \
    [[maybe_unused]] MiscM *m = reinterpret_cast<MiscM *>(smem);                                 \
    [[maybe_unused]] const Geo &G = *reinterpret_cast<const Geo *>(smem + kGeoOff);              \
    [[maybe_unused]] const int tid = threadIdx.x, lane = tid & 63;                               \
    /* wavefront index as a scalar: loops over work items become uniform (SALU counters, scalar branches) */ \
    [[maybe_unused]] const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);                    \
    [[maybe_unused]] const int n_l = lane & 15, q_l = lane >> 4


// ---- wavefront reductions on the VALU (DPP row shifts + row broadcasts, gfx9 encoding): the
// __shfl_* forms go through the LDS crossbar (~100 cycles per step) and were the slowest part of
// the epilogues.  Result is returned in every lane. ----
template <typename Op>
__device__ __forceinline__ int wave_reduce_i32(int v, int ident, Op op) {
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1,3
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2,3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_sum_dpp(int v) { return wave_reduce_i32(v, 0, [](int a, int b) { return a + b; }); }
__device__ __forceinline__ float wave_max_dpp(float v) {
    // NaN-free inputs expected (callers map NaN away); float max on the bit patterns via fmaxf
    const int r = wave_reduce_i32(__float_as_int(v), __float_as_int(-INFINITY),
                                  [](int a, int b) { return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b))); });
    return __int_as_float(r);
}

__device__ __forceinline__ int wave_min_dpp(int v) { return wave_reduce_i32(v, 0x7fffffff, [](int a, int b) { return a < b ? a : b; }); }
__device__ __forceinline__ u32 wave_umin_dpp(u32 v) {
    return (u32)wave_reduce_i32((int)(v ^ 0x80000000u), 0x7fffffff, [](int a, int b) { return a < b ? a : b; }) ^ 0x80000000u;
}
// inclusive prefix sum across the wavefront (same DPP sequence, result per lane)
__device__ __forceinline__ u32 wave_scan_dpp(u32 x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return (u32)v;
}
template <int CTRL, int RMASK>
__device__ __forceinline__ double dpp_add_step_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, RMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, RMASK, 0xf, false);
    return v + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);       // + 0.0 where no source lane
}
__device__ __forceinline__ double wave_sum_dpp_d(double v) {
    v = dpp_add_step_d<0x111, 0xf>(v); v = dpp_add_step_d<0x112, 0xf>(v); v = dpp_add_step_d<0x114, 0xf>(v);
    v = dpp_add_step_d<0x118, 0xf>(v); v = dpp_add_step_d<0x142, 0xa>(v); v = dpp_add_step_d<0x143, 0xc>(v);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- small block utilities ----

__device__ __forceinline__ u32 block_min(u32 v, MiscM *m) {
    v = wave_umin_dpp(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) m->red_i[threadIdx.x >> 6] = (int)v;
    __syncthreads();
    u32 a = (u32)m->red_i[0];
    for (int w = 1; w < kWavesM; ++w) { const u32 b = (u32)m->red_i[w]; a = a < b ? a : b; }
    return a;
}
__device__ __forceinline__ u32 f2key(float f) { u32 b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float key2f(u32 k) { u32 b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; return __uint_as_float(b); }

// k-th smallest (0-based) of v[0..n) by 4 x 8-bit radix select.  The four histograms are zeroed
// up front (hist4: 4 x 256 counters, 16-byte aligned scratch); per pass: all threads bin their elements (LDS atomics), barrier, wavefront 0 scans the
// 256 bins (4 per lane + a wavefront prefix sum) and publishes the digit, barrier.
__device__ __forceinline__ u32 block_select(const float *v, int n, u32 k, MiscM *m, u32 *hist4, u32 *count_le) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += kBlockM) hist4[i] = 0;
    if (threadIdx.x == 0) { m->sel_key = 0; m->sel_cle = 0; }
    __syncthreads();
    u32 prefix = 0, mask = 0, kk = k, less_total = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        u32 *hist = hist4 + 256 * pass;
        for (int i = threadIdx.x; i < n; i += kBlockM) {
            const u32 key = f2key(v[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (w == 0) {
            const uint4 c4 = *reinterpret_cast<const uint4 *>(hist + 4 * lane);
            const u32 tot = c4.x + c4.y + c4.z + c4.w;
            const u32 inc = wave_scan_dpp(tot);
            const u32 exc = inc - tot;
            if (kk >= exc && kk < inc) {                       // exactly one lane
                u32 e = exc, bin = 4 * lane, cnt = c4.x;
                if (kk >= e + c4.x) { e += c4.x; bin += 1; cnt = c4.y;
                    if (kk >= e + c4.y) { e += c4.y; bin += 1; cnt = c4.z;
                        if (kk >= e + c4.z) { e += c4.z; bin += 1; cnt = c4.w; } } }
                m->sel_key = bin; m->sel_cle = e; m->red_i[0] = (int)cnt;
            }
        }
        __syncthreads();
        const u32 bin = m->sel_key, less = m->sel_cle, cnt = (u32)m->red_i[0];
        prefix |= bin << shift;
        mask |= 255u << shift;
        kk -= less;
        less_total += less;
        if (pass == 3) *count_le = less_total + cnt;
    }
    return prefix;
}

// np.median of a float32 LDS array (mean of the two middle values for even n, float32).
__device__ __forceinline__ float block_median(const float *v, int n, MiscM *m, u32 *hist4) {
    u32 cle;
    if (n & 1) return key2f(block_select(v, n, (u32)(n / 2), m, hist4, &cle));
    const u32 k1 = (u32)(n / 2 - 1);
    const u32 key1 = block_select(v, n, k1, m, hist4, &cle);
    u32 key2 = key1;
    if (cle < k1 + 2) {                                        // next order statistic is a larger value
        u32 mn = 0xffffffffu;
        for (int i = threadIdx.x; i < n; i += kBlockM) { const u32 key = f2key(v[i]); if (key > key1 && key < mn) mn = key; }
        key2 = block_min(mn, m);
    }
    return (key2f(key1) + key2f(key2)) / 2.0f;
}

constexpr int kMedList = 256;
// The part after the key range [kmin, kmax] is known to every thread.  Expects hist[0..1023] = 0 and m->sel_cle = 0, made
// visible by a barrier.
__device__ __forceinline__ float block_median_ranged(const float *v, int n, MiscM *m, u32 *hist, u32 *list, u32 kmin, u32 kmax,
                                                     long long *dbg = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 range = kmax - kmin;
    const int shift = range < 1024u ? 0 : 22 - __clz(range);           // (range >> shift) < 1024
    for (int i = tid; i < n; i += kBlockM) atomicAdd(&hist[(f2key(v[i]) - kmin) >> shift], 1u);
    __syncthreads();
    if (dbg && tid == 0) dbg[21] = (long long)clock64();
    const bool two = !(n & 1);
    const u32 k1 = two ? (u32)(n / 2 - 1) : (u32)(n / 2);
    if (w == 0) {                                                      // 16 buckets per lane + a wavefront prefix sum
        u32 c[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 c4 = *reinterpret_cast<const uint4 *>(hist + 16 * lane + 4 * q);
            c[4 * q] = c4.x; c[4 * q + 1] = c4.y; c[4 * q + 2] = c4.z; c[4 * q + 3] = c4.w;
        }
        u32 tot = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += c[q];
        const u32 inc = wave_scan_dpp(tot), exc = inc - tot;
        if (k1 >= exc && k1 < inc) {                                   // exactly one lane
            u32 e = exc, bin = 16 * lane, cnt = 0;
            bool found = false;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (!found) { if (k1 < e + c[q]) { found = true; bin = 16 * lane + q; cnt = c[q]; } else e += c[q]; }
            }
            m->sel_key = bin; m->red_i[14] = (int)e; m->red_i[15] = (int)cnt;
        }
    }
    __syncthreads();
    if (dbg && tid == 0) dbg[22] = (long long)clock64();
    const u32 bin = m->sel_key, before = (u32)m->red_i[14], cnt = (u32)m->red_i[15];
    if (cnt > (u32)kMedList) return block_median(v, n, m, hist);       // block-uniform
    // the second middle value of an even-sized set lies in the same bucket unless the first is the bucket's last
    // element (block-uniform): only then is the smallest key beyond the bucket needed
    const bool need_above = two && !(k1 + 1 - before < cnt);
    u32 above = 0xffffffffu;
    for (int i = tid; i < n; i += kBlockM) {
        const u32 key = f2key(v[i]), b = (key - kmin) >> shift;
        if (b == bin) list[atomicAdd(&m->sel_cle, 1u)] = key;
        else if (need_above && b > bin && key < above) above = key;
    }
    if (need_above) above = block_min(above, m);                       // (barriers inside: the list is complete)
    else __syncthreads();
    if (dbg && tid == 0) dbg[23] = (long long)clock64();
    if ((u32)tid < cnt) {
        const u32 mine = list[tid];
        u32 rank = 0;
        for (u32 j = 0; j < cnt; ++j) { const u32 o = list[j]; rank += (o < mine || (o == mine && j < (u32)tid)) ? 1u : 0u; }
        if (rank == k1 - before) m->sel_key = mine;
        if (rank == k1 + 1 - before) m->best_key = (int)mine;
    }
    __syncthreads();
    if (dbg && tid == 0) dbg[24] = (long long)clock64();
    const u32 key1 = m->sel_key;
    if (!two) return key2f(key1);
    const u32 key2 = need_above ? above : (u32)m->best_key;
    return (key2f(key1) + key2f(key2)) / 2.0f;
}

// np.median, fast path: one log-spaced histogram over [min, max] of the keys (1024 buckets: the float bit
// patterns are spread evenly, no hot bins), the bucket holding the middle rank is compacted into a list and
// ranked by brute force.  Two passes over the data instead of the radix select's five; falls back to the
// radix select when the bucket holds more than kMedList elements (massive ties).
// hist: 1024 counters, list: kMedList keys (LDS scratch); tmin / tmax: this thread's min / max key of v.
__device__ __forceinline__ float block_median_fast(const float *v, int n, MiscM *m, u32 *hist, u32 *list, u32 tmin, u32 tmax)
{
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32 *red_max = reinterpret_cast<u32 *>(m->red_f);
    tmin = wave_umin_dpp(tmin);
    tmax = ~wave_umin_dpp(~tmax);
    __syncthreads();
    if (lane == 0) { m->red_i[w] = (int)tmin; red_max[w] = tmax; }
    for (int i = tid; i < 1024; i += kBlockM) hist[i] = 0;
    if (tid == 0) m->sel_cle = 0;
    __syncthreads();
    u32 kmin = (u32)m->red_i[0], kmax = red_max[0];
    for (int q = 1; q < kWavesM; ++q) { const u32 a = (u32)m->red_i[q], b = red_max[q]; kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax; }
    return block_median_ranged(v, n, m, hist, list, kmin, kmax);
}

// two block sums with one pair of barriers (scratch: the rotation table of the template phase, dead by now)
__device__ __forceinline__ void block_sum2(double &a, double &b, MiscM *m) {
    a = wave_sum_dpp_d(a); b = wave_sum_dpp_d(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { m->rot[threadIdx.x >> 6][0] = a; m->rot[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < kWavesM; ++w) { ta += m->rot[w][0]; tb += m->rot[w][1]; }
    a = ta; b = tb;
}

// np.std of a float32 LDS array from its double sums: sqrt(E[x^2] - E[x]^2) evaluated in double
// (NumPy works in float32; the difference is ~1e-7 relative, inside the 1e-5 bar on h).
__device__ __forceinline__ float std_from_sums(double sx, double sxx, int n) {
    const double mean = sx / (double)n;
    double var = sxx / (double)n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    return sqrtf((float)var);
}


// The spec's normalisation in IEEE double -> float32 (shared by candidates and the winner's matrix).
__device__ __forceinline__ float exact_ncc(double numer, double dI, double rTd, bool constT, bool lowvar)
{
    if (constT) return 1.0f;
    if (lowvar) return 0.0f;
    const double rI = 1.0 / sqrt(dI);
    double q = numer * rI;
    q = q * rTd;
    const double aq = fabs(q);
    return aq < 1.0 ? (float)q : (aq < 1.125 ? (q > 0.0 ? 1.0f : -1.0f) : 0.0f);
}

__device__ __forceinline__ float exact_from_sums(int p, int swp, u32 siiv, double nd, double sT, double rTd, bool cT)
{
    const double swd = (double)swp, siid = (double)siiv;
    const double dI = nd * siid - swd * swd;                           // exact
    const double s2 = siid + 256.0 * swd + 16384.0 * nd;               // S_II in the uint8 domain
    const bool lowvar = (2.0 * dI <= nd) && (dI * 8388608.0 <= 10.0 * nd * s2);
    const double numer = nd * (double)p - swd * sT;                    // exact
    return exact_ncc(numer, dI, rTd, cT, lowvar);
}

// The same value by a shorter route (the winner's matrix evaluates thousands of these per point).  1 / sqrt(dI) comes
// from v_rsq_f64 and two coupled Newton steps - no IEEE square root, no IEEE division - so the product q' differs from
// the spec's q (whose own 1 / sqrt carries two roundings) by less than 2^-49 |q|.  (float)q' = (float)q unless q' lies
// that close to a rounding boundary of float32 (the middle of the 29 discarded mantissa bits) or to one of the clamp
// thresholds 1 and 1.125: those lanes - about 1e-7 of them, plus the perfect matches - take the spec's route.
__device__ __forceinline__ float exact_from_sums_fast(int p, int swp, u32 siiv, double nd, double sT, double rTd, bool cT)
{
    const double swd = (double)swp, siid = (double)siiv;
    const double dI = nd * siid - swd * swd;                           // exact
    const double numer = nd * (double)p - swd * sT;                    // exact
    bool slow = 2.0 * dI <= nd;                                        // the low-variance rule needs a second look
    const double x = slow ? 1.0 : dI;
    const double y0 = __builtin_amdgcn_rsq(x);
    double g = x * y0, h = 0.5 * y0;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    h = __builtin_fma(h, r, h);                                        // 1 / (2 sqrt(dI))
    const double q = (numer * h) * (2.0 * rTd);
    const double aq = fabs(q);
    const u32 lo29 = (u32)__double2loint(q) & 0x1fffffffu;
    slow |= (lo29 - (0x10000000u - 64u)) <= 128u;
    slow |= fabs(aq - 1.0) < 1e-13 || fabs(aq - 1.125) < 1e-13;
    float out = aq < 1.0 ? (float)q : (aq < 1.125 ? (q > 0.0 ? 1.0f : -1.0f) : 0.0f);
    if (cT) out = 1.0f;
    if (slow && !cT) out = exact_from_sums(p, swp, siiv, nd, sT, rTd, cT);
    return out;
}

// Branch-free part of exact_from_sums_fast: the value by the short route and whether the lane must take the spec's route
// instead.  Callers evaluate a batch of these back to back (independent dependency chains interleave) and handle the rare
// flagged ones afterwards - a branch around every single value serialises the ~20-deep double-precision chains.
__device__ __forceinline__ float ncc_fast_nobranch(int p, int swp, u32 siiv, double nd, double sT, double rTd, bool cT, bool &slow)
{
    const double swd = (double)swp, siid = (double)siiv;
    const double dI = nd * siid - swd * swd;                           // exact
    const double numer = nd * (double)p - swd * sT;                    // exact
    bool sl = 2.0 * dI <= nd;                                          // the low-variance rule needs a second look
    const double x = sl ? 1.0 : dI;
    const double y0 = __builtin_amdgcn_rsq(x);
    double g = x * y0, h = 0.5 * y0;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    h = __builtin_fma(h, r, h);                                        // 1 / (2 sqrt(dI))
    const double q = (numer * h) * (2.0 * rTd);
    const double aq = fabs(q);
    const u32 lo29 = (u32)__double2loint(q) & 0x1fffffffu;
    sl |= (lo29 - (0x10000000u - 64u)) <= 128u;
    sl |= fabs(aq - 1.0) < 1e-13 || fabs(aq - 1.125) < 1e-13;
    const float clamp = aq < 1.125 ? (q > 0.0 ? 1.0f : -1.0f) : 0.0f;
    float out = aq < 1.0 ? (float)q : clamp;
    out = cT ? 1.0f : out;
    slow = sl && !cT;
    return out;
}

struct Score { float lmax, bestv; int bestkey; };

__device__ __forceinline__ void take_better(Score &sc, float rv, int key)
{
    if (rv > sc.bestv || (rv == sc.bestv && key < sc.bestkey)) { sc.bestv = rv; sc.bestkey = key; }
}

// ---------------------------------------------------------------------------------------------
// Phase 0a: search window of image 2 -> LDS, re-centred to int8 (w ^ 0x80), zero beyond the window.
// Loads are issued four dwords deep before any is consumed (HBM/L2 latency overlaps).
// ---------------------------------------------------------------------------------------------
// PATCH: the image-1 patch (ph_patch's fast path; the caller checked that it lies inside image 1 and that its LDS
// region is free already) is fetched in the same round trip: its loads are issued before the window's.
// PRIO >= 0: the wave priority of the phases up to the sweep, set here - inside a non-inlined phase - rather than in the kernel
// body (pm_kernel_rp.inc: SID_SETPRIO_TS)
// the image bytes of a point are read once by its workgroup (neighbouring points re-read the overlap within microseconds):
// -DSID_IMG_NT loads them non-temporally, so that they do not push the recycled blocks' dirty lines out of L2 (round 6 experiment)
#ifdef SID_IMG_NT
#define SID_IMG_LOAD(p) __builtin_nontemporal_load(p)
#else
#define SID_IMG_LOAD(p) (*(p))
#endif
template <bool PATCH, int PRIO = -1, int KWIN = 10, int KPAT = 4>
__device__ __noinline__ void ph_window_t(const uint8_t *img2, long long rows2, long long cols2, long long stride2,
                                         const uint8_t *img1, long long rows1, long long cols1, long long stride1)
{
    if (PRIO >= 0) __builtin_amdgcn_s_setprio(PRIO);
    SID_PHASE_LOCALS;
    constexpr int kPat = KPAT;                                         // patch dwords per thread and trip (ph_patch)
    u32 plo[kPat] = {}, phi[kPat] = {}, psh[kPat] = {};
    const int pdw = G.ppitch >> 2, pndw = G.pdim * pdw;
    if (PATCH) {
        const u32 pmagic = G.patch_magic;
        const int pst = (int)stride1;
        const uint8_t *porg = img1 + (long long)G.pr0 * stride1 + G.pc0;
        const u32 pmis = (u32)(reinterpret_cast<uintptr_t>(porg) & 3);
        const uint8_t *porg4 = porg - pmis;
        const long long plast_ll = (long long)((reinterpret_cast<uintptr_t>(img1 + (rows1 - 1) * stride1 + cols1) - 1) & ~(uintptr_t)3) -
                                   (long long)reinterpret_cast<uintptr_t>(porg4);
        const u32 plast = plast_ll > 0x7ffffff0ll ? 0x7ffffff0u : (u32)plast_ll;
#pragma unroll
        for (int u = 0; u < kPat; ++u) {
            const int idx = u * kBlockM + tid;
            const int idc = idx < pndw ? idx : 0;
            const int row = (int)__umulhi((u32)idc, pmagic);
            const int dq = idc - row * pdw;
            const u32 o = (u32)(row * pst + 4 * dq) + pmis;
            const u32 oa = o & ~3u;
            const u32 ob = oa + 4 <= plast ? oa + 4 : plast;
            psh[u] = o & 3u;
            plo[u] = SID_IMG_LOAD(reinterpret_cast<const u32 *>(porg4 + oa));
            phi[u] = SID_IMG_LOAD(reinterpret_cast<const u32 *>(porg4 + ob));
        }
    }
    uint8_t *win = smem + G.win_off;
    const int wpitch = G.wpitch, ww = G.ww, wh = G.wh;
    const int dw_per_row = wpitch >> 2, ndw = (wh + G.band - 1) * dw_per_row;   // + band-1 zero rows below
    const u32 magic = G.win_magic;
    const int st = (int)stride2;
    // all addresses as 32-bit offsets from the (dword-aligned) window origin: no 64-bit multiplies, no divisions
    const uint8_t *org = img2 + G.r0 * stride2 + G.c0;
    const u32 mis = (u32)(reinterpret_cast<uintptr_t>(org) & 3);
    const uint8_t *org4 = org - mis;
    const long long last_ll = (long long)((reinterpret_cast<uintptr_t>(img2 + (rows2 - 1) * stride2 + cols2) - 1) & ~(uintptr_t)3) -
                              (long long)reinterpret_cast<uintptr_t>(org4);
    const u32 last_off = last_ll > 0x7ffffff0ll ? 0x7ffffff0u : (u32)last_ll;  // offset of the last legal dword
    // kWin dwords per thread and round trip: the common small windows (border <= 23 at 256 threads) load in
    // one trip, i.e. one HBM/L2 latency per point
    constexpr int kWin = KWIN;
    for (int base = 0; base < ndw; base += kWin * kBlockM) {
        u32 lo[kWin], hi[kWin], shv[kWin];
        int rowv[kWin], dqv[kWin];
#pragma unroll
        for (int u = 0; u < kWin; ++u) {
            const int idx = base + u * kBlockM + tid;
            const int idc = idx < ndw ? idx : 0;
            const int row = (int)__umulhi((u32)idc, magic);
            const int dq = idc - row * dw_per_row;
            rowv[u] = row; dqv[u] = dq;
            const int rc = row < wh ? row : wh - 1;                    // clamp so that the loads are always legal
            const int dqc = 4 * dq < ww ? dq : 0;
            const u32 o = (u32)(rc * st + 4 * dqc) + mis;
            const u32 oa = o & ~3u;
            const u32 ob = oa + 4 <= last_off ? oa + 4 : last_off;
            shv[u] = o & 3u;
            lo[u] = SID_IMG_LOAD(reinterpret_cast<const u32 *>(org4 + oa));
            hi[u] = SID_IMG_LOAD(reinterpret_cast<const u32 *>(org4 + ob));
        }
#pragma unroll
        for (int u = 0; u < kWin; ++u) {
            const int idx = base + u * kBlockM + tid;
            if (idx < ndw) {
                u32 v = __builtin_amdgcn_alignbyte(hi[u], lo[u], shv[u]) ^ 0x80808080u;
                const int nvalid = rowv[u] < wh ? ww - 4 * dqv[u] : 0;
                if (nvalid < 4) v = nvalid > 0 ? (v & ((1u << (8 * nvalid)) - 1u)) : 0u;
                reinterpret_cast<u32 *>(win)[idx] = v;                 // idx == row * dw_per_row + dq
            }
        }
    }
    if (PATCH) {
        uint8_t *patch = smem + G.patch_off;
        if (tid == 0) patch[G.pdim * G.ppitch] = 128;                  // (as ph_patch)
#pragma unroll
        for (int u = 0; u < kPat; ++u) {
            const int idx = u * kBlockM + tid;
            if (idx < pndw) reinterpret_cast<u32 *>(patch)[idx] = __builtin_amdgcn_alignbyte(phi[u], plo[u], psh[u]);
        }
        // (larger patches than kPat * blockDim dwords do not occur: pdim <= 58 for s <= 35, 256 threads)
    }
}

__device__ __forceinline__ void ph_window(const uint8_t *img2, long long rows2, long long cols2, long long stride2)
{
    ph_window_t<false>(img2, rows2, cols2, stride2, nullptr, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// Phase 1: S_II' = box sums of w'^2 (running sums down the columns, then along the rows).
// ---------------------------------------------------------------------------------------------
[[maybe_unused]] __device__ __noinline__ void ph_sums()
{
    SID_PHASE_LOCALS;
    const uint8_t *win = smem + G.win_off;
    u32 *sii = reinterpret_cast<u32 *>(smem + G.sii_off);
    u32 *colsum = reinterpret_cast<u32 *>(smem + G.u_off);
    const int s = G.s, ww = G.ww, rh = G.rh, rw = G.rw, wpitch = G.wpitch;
    // Both passes cut their running sums into segments so that the whole workgroup works and the chain of
    // dependent LDS reads is short: a segment starts from a directly summed value (s independent reads)
    // and slides over its few rows / columns.
    {   // down the columns: thread = (column x, segment of rows)
        int nseg = kBlockM / ww; nseg = nseg < 1 ? 1 : (nseg > rh ? rh : nseg);
        const int per = (rh + nseg - 1) / nseg;
        for (int task = tid; task < ww * nseg; task += kBlockM) {
            const int seg = task / ww, x = task - seg * ww;
            const int ya = seg * per, yb = ya + per < rh ? ya + per : rh;
            if (ya >= yb) continue;
            const int8_t *col = reinterpret_cast<const int8_t *>(win) + x + ya * wpitch;
            int c = 0;
#pragma unroll 8
            for (int i = 0; i < s; ++i) { const int v = col[i * wpitch]; c += v * v; }
            colsum[ya * ww + x] = (u32)c;
#pragma unroll 4
            for (int y = 1; y < yb - ya; ++y) {
                const int vo = col[(y - 1) * wpitch], vn = col[(y + s - 1) * wpitch];
                c += vn * vn - vo * vo;
                colsum[(ya + y) * ww + x] = (u32)c;
            }
        }
    }
    __syncthreads();
    {   // along the rows: thread = (row y, segment of columns)
        int nseg = kBlockM / rh; nseg = nseg < 1 ? 1 : (nseg > rw ? rw : nseg);
        const int per = (rw + nseg - 1) / nseg;
        for (int task = tid; task < rh * nseg; task += kBlockM) {
            const int y = task / nseg, seg = task - y * nseg;
            const int xa = seg * per, xb = xa + per < rw ? xa + per : rw;
            if (xa >= xb) continue;
            const u32 *cr = colsum + y * ww + xa;
            u32 acc = 0;
#pragma unroll 8
            for (int j = 0; j < s; ++j) acc += cr[j];
            sii[y * rw + xa] = acc;
#pragma unroll 4
            for (int x = 1; x < xb - xa; ++x) {
                acc += cr[x + s - 1] - cr[x - 1];
                sii[y * rw + xa + x] = acc;
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Phase 0b: neighbourhood of the template centre on image 1 -> LDS patch (8 byte loads in flight).
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ void ph_patch(const uint8_t *img1, long long rows1, long long cols1, long long stride1)
{
    SID_PHASE_LOCALS;
    uint8_t *patch = smem + G.patch_off;
    const int pdim = G.pdim, ppitch = G.ppitch, np = pdim * pdim;
    const long long pr0 = G.pr0, pc0 = G.pc0;
    if (tid == 0) patch[pdim * ppitch] = 128;                          // spare byte: what a flagged table entry gathers
    if (pr0 >= 0 && pc0 >= 0 && pr0 + pdim <= rows1 && pc0 + ppitch <= cols1) {
        // the patch lies inside image 1 (block-uniform): dword loads re-aligned in registers, 32-bit offsets,
        // division by the row pitch through the precomputed reciprocal
        const int pdw = ppitch >> 2, ndw = pdim * pdw;
        const u32 magic = G.patch_magic;
        const int st = (int)stride1;
        const uint8_t *org = img1 + pr0 * stride1 + pc0;
        const u32 mis = (u32)(reinterpret_cast<uintptr_t>(org) & 3);
        const uint8_t *org4 = org - mis;
        const long long last_ll = (long long)((reinterpret_cast<uintptr_t>(img1 + (rows1 - 1) * stride1 + cols1) - 1) & ~(uintptr_t)3) -
                                  (long long)reinterpret_cast<uintptr_t>(org4);
        const u32 last_off = last_ll > 0x7ffffff0ll ? 0x7ffffff0u : (u32)last_ll;
        for (int base = 0; base < ndw; base += 4 * kBlockM) {
            u32 lo[4], hi[4], shv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * kBlockM + tid;
                const int idc = idx < ndw ? idx : 0;
                const int row = (int)__umulhi((u32)idc, magic);
                const int dq = idc - row * pdw;
                const u32 o = (u32)(row * st + 4 * dq) + mis;
                const u32 oa = o & ~3u;
                const u32 ob = oa + 4 <= last_off ? oa + 4 : last_off;
                shv[u] = o & 3u;
                lo[u] = *reinterpret_cast<const u32 *>(org4 + oa);
                hi[u] = *reinterpret_cast<const u32 *>(org4 + ob);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * kBlockM + tid;
                if (idx < ndw) reinterpret_cast<u32 *>(patch)[idx] = __builtin_amdgcn_alignbyte(hi[u], lo[u], shv[u]);
            }
        }
        return;
    }
    // near the image border: byte loads with clamped coordinates
    for (int base = 0; base < np; base += 8 * kBlockM) {
        uint8_t pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * kBlockM + tid;
            const int idc = idx < np ? idx : 0;
            const int pr = idc / pdim, pc = idc - pr * pdim;
            long long gr = pr0 + pr, gc = pc0 + pc;
            gr = gr < 0 ? 0 : (gr > rows1 - 1 ? rows1 - 1 : gr);       // clamped rows/cols are never sampled
            gc = gc < 0 ? 0 : (gc > cols1 - 1 ? cols1 - 1 : gc);
            pv[u] = img1[gr * stride1 + gc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * kBlockM + tid;
            if (idx < np) { const int pr = idx / pdim, pc = idx - pr * pdim; patch[pr * ppitch + pc] = pv[u]; }
        }
    }
}

// Rotated-template sampling (reference pmlib.py:105-113; scipy order-0 arithmetic) from the LDS patch.
// Thread -> one template column j and every ngrp-th row.  Fast pass in float32 on patch-relative
// coordinates (one fma per coordinate and sample; branch-free so that several angles interleave);
// whenever the rounding or the image-bounds decision could depend on the last bits (|doubt| < kGuard,
// float error < 2.5e-5) the sample is flagged and redone afterwards with scipy's double arithmetic.
struct SampleGeom {
    const uint8_t *patch; const uint8_t *pre; int ppitch, pr0, pc0, s, ngrp, ig, j; bool act, inside, lin;
    float lo_r, hi_r, lo_c, hi_c;
    double c1, r1, rmax1, cmax1;
};
constexpr float kGuard = 1e-4f;
constexpr int kRowsPerThread = 5;       // rows of one template column a thread samples per chunk

__device__ __forceinline__ SampleGeom sample_geom(const Geo &G, int s, long long rows1, long long cols1, const uint8_t *patch)
{
    SampleGeom g;
    g.patch = patch; g.ppitch = G.ppitch; g.pr0 = G.pr0; g.pc0 = G.pc0; g.s = s;
    g.ngrp = kBlockM / s;
    g.ig = (int)threadIdx.x / s; g.j = (int)threadIdx.x - g.ig * s;
    g.act = g.ig < g.ngrp;
    // whole patch at least one pixel inside image 1: the image-bounds tests can be skipped (block-uniform)
    g.inside = G.pr0 >= 1 && G.pc0 >= 1 && (long long)G.pr0 + G.pdim <= rows1 - 1 && (long long)G.pc0 + G.pdim <= cols1 - 1;
    g.lo_r = (float)(-G.pr0); g.hi_r = (float)(rows1 - 1 - G.pr0);
    g.lo_c = (float)(-G.pc0); g.hi_c = (float)(cols1 - 1 - G.pc0);
    g.c1 = G.c1; g.r1 = G.r1; g.rmax1 = (double)(rows1 - 1); g.cmax1 = (double)(cols1 - 1);
    g.lin = G.lin != 0;
    g.pre = reinterpret_cast<const uint8_t *>(G.pre);
    return g;
}

// exact coordinates of sample (i, j) -> pixel value (0 outside image 1)
// ka = index of the angle in the run's list (needed for pre-sampled templates only: rot_order 2..5)
__device__ __forceinline__ int sample_exact(const SampleGeom &g, const double *rot4, int i, int j, int ka)
{
    if (g.pre) return g.pre[((size_t)ka * g.s + i) * g.s + j];         // sampled from the spline coefficients by lw_presample
    const double cosa = rot4[0], sina = rot4[1];
    const double off0 = g.r1 - rot4[2], off1 = g.c1 - rot4[3];
    double rr = 0.0 + (double)i * cosa;                                // NI_GeometricTransform order (matrix = transform.T)
    rr = rr + (double)j * sina;
    rr = rr + off0;
    double cc = 0.0 + (double)i * (-sina);
    cc = cc + (double)j * cosa;
    cc = cc + off1;
    int v = 0;
    if (rr >= 0.0 && rr <= g.rmax1 && cc >= 0.0 && cc <= g.cmax1) {
        if (g.lin) {
            // rot_order = 1: scipy's spline order 1 (NI_GeometricTransform, mode 'constant', cval 0, uint8 output) - weights
            // (1 - y, y), the four taps accumulated in float64 in the order (0,0) (0,1) (1,0) (1,1) as ((v w_row) w_col), then
            // t > 0 ? t + 0.5 : 0, clamped to 255, truncated.  A tap behind the last image row / column carries the weight 0
            // (the coordinate is then integral) and reads a finite patch byte.  Oracle: get_template_order1, fixture G1b.
            const double fr = floor(rr), fc = floor(cc);
            const double yr = rr - fr, yc = cc - fc;
            const double w0r = 1.0 - yr, w0c = 1.0 - yc;
            const uint8_t *p = g.patch + ((int)fr - g.pr0) * g.ppitch + ((int)fc - g.pc0);
            double t = 0.0;
            t = t + ((double)p[0] * w0r) * w0c;
            t = t + ((double)p[1] * w0r) * yc;
            t = t + ((double)p[g.ppitch] * yr) * w0c;
            t = t + ((double)p[g.ppitch + 1] * yr) * yc;
            t = t > 0.0 ? t + 0.5 : 0.0;
            t = t > 255.0 ? 255.0 : t;
            return (int)t;
        }
        const int ri = (int)floor(rr + 0.5) - g.pr0, ci = (int)floor(cc + 0.5) - g.pc0;
        v = g.patch[ri * g.ppitch + ci];
    }
    return v;
}

// Fast pass for rows ig + (k0..k0+4)*ngrp of column j of one angle.  Returns the doubt bits; sure
// samples are stored and summed, doubtful ones are left to the caller.  Branch-free: a rejected sample
// is handed to `store` with take = false (it writes a scratch byte), so that all the chains of a chunk
// - and of the angles in flight - sit in one basic block and overlap.  INSIDE: the patch lies wholly
// inside image 1, no bounds tests.
template <bool INSIDE, typename Store>
__device__ __forceinline__ u32 sample_fast5(const SampleGeom &g, const double *rot4, bool angle_ok, int k0,
                                            int &st, int &stt, int &sawzero, Store store)
{
    const double cosa = rot4[0], sina = rot4[1];
    const float cf = (float)cosa, sf = (float)sina;
    const float orf = (float)((g.r1 - rot4[2]) - (double)g.pr0), ocf = (float)((g.c1 - rot4[3]) - (double)g.pc0);
    const float fj = (float)g.j, fig = (float)g.ig;
    const float base_r = fmaf(fig, cf, fmaf(fj, sf, orf)) + 0.5f, step_r = (float)g.ngrp * cf;
    const float base_c = fmaf(fig, -sf, fmaf(fj, cf, ocf)) + 0.5f, step_c = -(float)g.ngrp * sf;
    const bool lane_ok = angle_ok && g.act;
    u32 doubt = 0;
#pragma unroll
    for (int u = 0; u < kRowsPerThread; ++u) {
        const int k = k0 + u;
        const int i = g.ig + k * g.ngrp;
        const bool valid = lane_ok && i < g.s;
        const float fk = (float)k;
        const float tr = fmaf(fk, step_r, base_r), tc_ = fmaf(fk, step_c, base_c);   // coordinate + 0.5
        const float flr = floorf(tr), flc = floorf(tc_);
        // doubtful when the coordinate is within kGuard of a rounding boundary
        bool sure = fabsf((tr - flr) - 0.5f) < 0.5f - kGuard && fabsf((tc_ - flc) - 0.5f) < 0.5f - kGuard;
        sure = sure && !g.lin;                                         // (bilinear sampling: every sample takes the float64 route)
        bool in_f = true;
        if (!INSIDE) {
            const float rrf = tr - 0.5f, ccf = tc_ - 0.5f;
            in_f = rrf >= g.lo_r + kGuard && rrf <= g.hi_r - kGuard && ccf >= g.lo_c + kGuard && ccf <= g.hi_c - kGuard;
            const bool out_f = rrf < g.lo_r - kGuard || rrf > g.hi_r + kGuard || ccf < g.lo_c - kGuard || ccf > g.hi_c + kGuard;
            sure = sure && (in_f || out_f);
        }
        const int ri = in_f ? (int)flr : 0, ci = in_f ? (int)flc : 0; // in-image samples always lie inside the patch
        int v = g.patch[ri * g.ppitch + ci];
        v = in_f ? v : 0;
        const bool take = valid && sure;
        doubt |= ((valid && !sure) ? 1u : 0u) << u;
        sawzero |= (take && v == 0) ? 1 : 0;
        store(k, i, v, take);
        const int sv = take ? v - 128 : 0;
        st += sv; stt += sv * sv;
    }
    return doubt;
}

// Table path (integral template centre, patch inside image 1): four samples of one template row through the
// host-built offset table (pm_capi.hip make_samp) -> packed raw pixels j = 4*jq .. 4*jq+3.  A flagged entry
// points at the spare byte behind the patch (128 = re-centred 0); ph_tpl_fix supplies the real sample.
__device__ __forceinline__ u32 gather_quad(const uint8_t *patch, const uint2 e)
{
    return (u32)patch[e.x & 0x7fffu] | ((u32)patch[(e.x >> 16) & 0x7fffu] << 8) |
           ((u32)patch[e.y & 0x7fffu] << 16) | ((u32)patch[(e.y >> 16) & 0x7fffu] << 24);
}
// block-uniform: integral template centre and the patch at least one pixel inside image 1
__device__ __forceinline__ bool table_usable(const Geo &G, long long rows1, long long cols1)
{
    const bool inside = G.pr0 >= 1 && G.pc0 >= 1 && (long long)G.pr0 + G.pdim <= rows1 - 1 && (long long)G.pc0 + G.pdim <= cols1 - 1;
    return inside && G.c1 == floor(G.c1) && G.r1 == floor(G.r1) && fabs(G.c1) < 1e9 && fabs(G.r1) < 1e9;
}

// ---------------------------------------------------------------------------------------------
// Phase 0c: operands of one group of <= 15 angles.  afrag[i][lane = g*16 + slot][16 bytes], byte jj
// <-> column c = 16 g + jj; slot 15 is the all-ones template; row s is all zero.
// Four small functions (begin / one of the two samplers / end) so that the common sampler - the table
// path - does not inherit the register appetite (and callee-saved spills) of the general one.
// ---------------------------------------------------------------------------------------------
template <int S>
__device__ __noinline__ void ph_tpl_begin(const double *rot, int a0, int Kg, long long *dbg_cycles)
{
    SID_PHASE_LOCALS;
    uint8_t *afrag = smem + G.u_off;
    const int arow = G.arow;
    if (tid < 4 * Kg) (&m->rot[0][0])[tid] = rot[4 * a0 + tid];        // one global round trip for the whole group
    for (int idx = tid; idx < G.tab_rows * arow / 16; idx += kBlockM) reinterpret_cast<uint4 *>(afrag)[idx] = make_uint4(0, 0, 0, 0);
    if (tid < kSlots) { m->isT[tid] = 0; m->isTT[tid] = 0; }
    __syncthreads();                                                   // also: patch complete
    if (dbg_cycles && tid == 0) dbg_cycles[8] = (long long)clock64();
}

// general sampler: any centre, any position relative to the image border
template <int S>
__device__ __noinline__ void ph_tpl_general(int a0, int Kg, long long rows1, long long cols1)
{
    SID_PHASE_LOCALS;
    uint8_t *afrag = smem + G.u_off;
    const uint8_t *patch = smem + G.patch_off;
    const int s = S > 0 ? S : G.s, arow = G.arow;                      // compile-time s: divisions by s become multiplies
    const SampleGeom g = sample_geom(G, s, rows1, cols1, patch);
    const int dump = G.tab_rows * arow;                                // 16 scratch bytes behind the operand table
    // byte offset of (row ig + k*ngrp, column j, slot a) = wbase + k*wstep + 16 a
    const int wbase = G.arow0 + g.ig * arow + (g.j >> 4) * G.gpitch + (g.j & 15), wstep = g.ngrp * arow;
    int sawzero = 0;
    auto run = [&](auto inside_tag) {
        constexpr bool INSIDE = decltype(inside_tag)::value;
        unsigned long long dlo = 0, dhi = 0;
        for (int k0 = 0; k0 * g.ngrp < s; k0 += kRowsPerThread) {
            for (int a = 0; a < Kg; a += 3) {                          // three angles in flight: their chains interleave
                int st[3] = {0, 0, 0}, stt[3] = {0, 0, 0};
                u32 db[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int aa = a + q < Kg ? a + q : Kg - 1;
                    db[q] = sample_fast5<INSIDE>(g, m->rot[aa], a + q < Kg, k0, st[q], stt[q], sawzero,
                        [&](int k, int, int v, bool take) { afrag[take ? wbase + k * wstep + 16 * aa : dump] = (uint8_t)(v ^ 0x80); });
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (a + q < Kg) {                                  // wavefront-uniform
                        const int ws = wave_sum_dpp(st[q]), wss = wave_sum_dpp(stt[q]);
                        if (lane == 0) { atomicAdd(&m->isT[a + q], ws); atomicAdd(&m->isTT[a + q], wss); }
                        // remember the doubtful samples: bit 5*(a+q) + u of a 128-bit mask
                        const int pos = 5 * (a + q);
                        const unsigned long long d = (unsigned long long)db[q];
                        if (pos < 64) { dlo |= d << pos; if (pos > 59) dhi |= d >> (64 - pos); }
                        else dhi |= d << (pos - 64);
                    }
                }
            }
            // rare: redo the doubtful samples of this chunk with scipy's double arithmetic (kept apart
            // from the DPP reductions above: no divergent code between them)
            if (dlo | dhi) {
                for (int aa = 0; aa < Kg; ++aa) {
                    const int pos = 5 * aa;
                    const u32 dbits = (u32)((pos < 64 ? (dlo >> pos) | (pos > 59 ? dhi << (64 - pos) : 0ull) : dhi >> (pos - 64)) & 31ull);
                    for (int u = 0; u < kRowsPerThread; ++u) {
                        if ((dbits >> u) & 1u) {
                            const int i = g.ig + (k0 + u) * g.ngrp;
                            const int v = sample_exact(g, m->rot[aa], i, g.j, a0 + aa);
                            sawzero |= (v == 0) ? 1 : 0;
                            afrag[wbase + (k0 + u) * wstep + 16 * aa] = (uint8_t)(v ^ 0x80);
                            atomicAdd(&m->isT[aa], v - 128); atomicAdd(&m->isTT[aa], (v - 128) * (v - 128));
                        }
                    }
                }
            }
            dlo = 0; dhi = 0;
        }
    };
    if (g.inside) run(std::true_type{}); else run(std::false_type{});
    if (sawzero) m->zero_flag = 1;
}

// table sampler: integral template centre, patch inside image 1 (table_usable)
template <int S>
__device__ __noinline__ void ph_tpl_table(const uint16_t *samp, int a0, int Kg, long long rows1, long long cols1)
{
    SID_PHASE_LOCALS;
    uint8_t *afrag = smem + G.u_off;
    const uint8_t *patch = smem + G.patch_off;
    const int s = S > 0 ? S : G.s, arow = G.arow;
    int sawzero = 0;
    {
        // one wavefront per angle: four samples per lane and step through the offset table, one dword store
        // into the operand table; pixel sums by v_dot4 on the packed raw bytes, converted to the re-centred
        // domain per angle: sum(v-128) = sum v - 128 N, sum(v-128)^2 = sum v^2 - 256 sum v + 16384 N
        const int sp = samp_pitch(s), nq = sp >> 2, upa = s * nq;
        const u32 tailmask = (s & 3) ? (1u << (8 * (s & 3))) - 1u : 0xffffffffu;
        // table entries are fetched a whole chunk (kCh steps of 64 lanes) ahead - those of the next angle while
        // this one is gathered - so that the L2 latency of the table is paid once per wavefront, not per step
        constexpr int kCh = S > 0 ? (S * ((S + 3) / 4) + 63) / 64 : 4;
        auto fetch = [&](int a, int u0, uint2 (&e)[kCh]) {
            const uint2 *ta = reinterpret_cast<const uint2 *>(samp + (size_t)(a0 + a) * s * sp);
#pragma unroll
            for (int c = 0; c < kCh; ++c) {
                const int u = u0 + 64 * c + lane;
                e[c] = ta[u < upa ? u : 0];
            }
        };
        uint2 cur[kCh] = {}, nxt[kCh] = {};
        if (wv < Kg) fetch(wv, 0, cur);
        for (int a = wv; a < Kg; a += kWavesM) {
            u32 sv = 0, svv = 0, zero = 0;
            for (int u0 = 0; u0 < upa; u0 += 64 * kCh) {
                // what comes next: the following chunk of this angle, else the first chunk of the next angle
                const bool more = u0 + 64 * kCh < upa;
                if (more) fetch(a, u0 + 64 * kCh, nxt);
                else if (a + kWavesM < Kg) fetch(a + kWavesM, 0, nxt);
                // all gathers of the chunk first (branch-free: their LDS latencies overlap)
                u32 rawv[kCh];
#pragma unroll
                for (int c = 0; c < kCh; ++c) rawv[c] = gather_quad(patch, cur[c]);
#pragma unroll
                for (int c = 0; c < kCh; ++c) {
                    const int u = u0 + 64 * c + lane;
                    if (u < upa) {
                        const int i = u / nq, jq = u - i * nq;
                        const u32 vm = jq == nq - 1 ? tailmask : 0xffffffffu;
                        const u32 raw = rawv[c] & vm;
                        const u32 z = raw | ~vm;
                        zero |= (z - 0x01010101u) & ~z & 0x80808080u;  // some valid byte == 0
                        sv = __builtin_amdgcn_udot4(raw, 0x01010101u, sv, false);
                        svv = __builtin_amdgcn_udot4(raw, raw, svv, false);
                        *reinterpret_cast<u32 *>(afrag + G.arow0 + i * arow + (jq >> 2) * G.gpitch + a * 16 + (jq & 3) * 4) = (raw ^ 0x80808080u) & vm;
                    }
                }
#pragma unroll
                for (int c = 0; c < kCh; ++c) cur[c] = nxt[c];
            }
            const int ws = wave_sum_dpp((int)sv), wss = wave_sum_dpp((int)svv);
            if (lane == 0) { m->isT[a] = ws - 128 * s * s; m->isTT[a] = wss - 256 * ws + 16384 * s * s; }
            sawzero |= zero != 0u ? 1 : 0;
        }
    }
    if (sawzero) m->zero_flag = 1;
}

// Rare (only when the host flagged table entries, A.samp_nflag > 0): the samples whose coordinate sits
// within kSampGuard of a rounding boundary, recomputed with the reference's float64 operation order.
// winner_ka < 0: operand table of the angle group (+ template sums, zero guard); else: the winner's operands.
template <int S>
__device__ __noinline__ void ph_tpl_fix(const uint16_t *samp, int a0, int Kg, long long rows1, long long cols1, int winner_ka)
{
    SID_PHASE_LOCALS;
    __syncthreads();                                                   // the sampler's stores and sums are complete
    const uint8_t *patch = smem + G.patch_off;
    const int s = S > 0 ? S : G.s, arow = G.arow, sp = samp_pitch(s);
    const SampleGeom g = sample_geom(G, s, rows1, cols1, patch);
    const int a_lo = winner_ka < 0 ? 0 : 0, a_n = winner_ka < 0 ? Kg : 1;
    for (int idx = tid; idx < a_n * s * sp; idx += kBlockM) {
        const int a = idx / (s * sp), rem = idx - a * s * sp;
        const int i = rem / sp, j = rem - i * sp;
        const int ag = winner_ka < 0 ? a0 + a_lo + a : winner_ka;      // angle index in the table
        if (j >= s || !(samp[(size_t)ag * s * sp + rem] & 0x8000u)) continue;
        const double *rot4 = winner_ka < 0 ? m->rot[a] : m->rot[0];
        const int v = sample_exact(g, rot4, i, j, -1);                  // (table path: never with pre-sampled templates)
        if (winner_ka < 0) {
            (smem + G.u_off)[G.arow0 + i * arow + (j >> 4) * G.gpitch + a * 16 + (j & 15)] = (uint8_t)(v ^ 0x80);
            atomicAdd(&m->isT[a], v - 128); atomicAdd(&m->isTT[a], (v - 128) * (v - 128));
            if (v == 0) m->zero_flag = 1;
        } else {
            const int trows = s + kTrowPad;
            (smem + G.u_off)[((j >> 4) * trows + 16 + i) * 16 + (j & 15)] = (uint8_t)(v ^ 0x80);
        }
    }
}

template <int S>
__device__ __noinline__ void ph_tpl_end(int a0, int Kg, uint8_t *dbg_templates, long long *dbg_cycles)
{
    SID_PHASE_LOCALS;
    uint8_t *afrag = smem + G.u_off;
    const int s = S > 0 ? S : G.s, arow = G.arow;
    const double nd = G.nd;
    if (dbg_cycles && tid == 0) dbg_cycles[15] = (long long)clock64();
    for (int idx = tid; idx < s * s; idx += kBlockM) {                  // slot 15: all-ones template
        const int i = idx / s, j = idx - i * s;
        afrag[G.arow0 + i * arow + (j >> 4) * G.gpitch + G.ones_slot * 16 + (j & 15)] = 1;
    }
    __syncthreads();
    if (dbg_cycles && tid == 0) dbg_cycles[9] = (long long)clock64();
    if (dbg_templates) {
        for (int idx = tid; idx < Kg * s * s; idx += kBlockM) {
            const int a = idx / (s * s), rem = idx - a * s * s;
            const int i = rem / s, j = rem - i * s;
            dbg_templates[(a0 + a) * s * s + rem] = afrag[G.arow0 + i * arow + (j >> 4) * G.gpitch + a * 16 + (j & 15)] ^ 0x80;
        }
    }
    if (tid < Kg) {                                                    // per-angle terms (signed domain)
        const double st = (double)m->isT[tid], stt = (double)m->isTT[tid];
        const double dT = nd * stt - st * st;                          // exact
        const double rT = 1.0 / sqrt(dT);
        m->sTd[a0 + tid] = st;
        m->constT[a0 + tid] = dT == 0.0 ? 1 : 0;
        m->rTd[a0 + tid] = rT;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// One sweep work item: kBand output rows x 16 placements x 16 template slots.
// Window row rho = y0 + step feeds output row y0 + t through template row i = step - t, whose
// fragment sits in ring slot (i & (kBand-1)).  S > 0: template side known at compile time - the whole
// schedule is static (no branch, exactly kBand*S MFMAs).  S == 0: runtime side; the ring holds zero
// fragments for i outside [0, s) so the MFMAs of a step are unconditional.
// Lane (n = l&15, g = l>>4): B fragment = bytes W'[row][x0+n+16g .. +15], built from 5 dwords read at
// the lane's dword-aligned address + v_alignbyte_b32; A fragment = 16 bytes of the operand table
// (lanes of k-group 3 read the all-zero row when s <= 48: their per-lane row stride is 0).
// ---------------------------------------------------------------------------------------------
struct Raw5 { u32 r[5]; };
__device__ __forceinline__ Raw5 read_raw(const uint8_t *p)
{
    const u32 *q = reinterpret_cast<const u32 *>(p);
    Raw5 w;
    w.r[0] = q[0]; w.r[1] = q[1]; w.r[2] = q[2]; w.r[3] = q[3]; w.r[4] = q[4];
    return w;
}
__device__ __forceinline__ v4i align_raw(const Raw5 &w, u32 sh)
{
    v4i b;
    b[0] = (int)__builtin_amdgcn_alignbyte(w.r[1], w.r[0], sh);
    b[1] = (int)__builtin_amdgcn_alignbyte(w.r[2], w.r[1], sh);
    b[2] = (int)__builtin_amdgcn_alignbyte(w.r[3], w.r[2], sh);
    b[3] = (int)__builtin_amdgcn_alignbyte(w.r[4], w.r[3], sh);
    return b;
}

template <int S, int kBand>
__device__ __forceinline__ void sweep_item(v4i (&acc)[kBand], const uint8_t *ap, int arow_l, const uint8_t *bp,
                                           int wpitch, int s, u32 sh)
{
#pragma unroll
    for (int t = 0; t < kBand; ++t) acc[t] = v4i{0, 0, 0, 0};
    v4i ring[kBand];
#pragma unroll
    for (int t = 0; t < kBand; ++t) ring[t] = v4i{0, 0, 0, 0};
    if (S > 0) {
        constexpr int NS = kBand + (S > 0 ? S : 1) - 1;
        // Software pipeline of depth 2, interleaved with the MFMAs of the current step (a single
        // wavefront issues in order: anything not placed between two MFMAs adds to the step time):
        //   step k:  MFMAs(k)  ||  finish the window fragment of k+1 (alignbyte x4)  ||  issue LDS reads of k+2
        ring[0] = *reinterpret_cast<const v4i *>(ap);
        v4i b_cur = align_raw(read_raw(bp), sh);
        v4i a_nxt = *reinterpret_cast<const v4i *>(ap + arow_l);
        Raw5 b_raw = read_raw(bp + wpitch);
        ap += 2 * arow_l; bp += 2 * wpitch;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int step = 0; step < NS; ++step) {
            const v4i a_fin = a_nxt;
            const v4i b_nxt = align_raw(b_raw, sh);                  // reads were issued during step-1
            if (step + 2 < NS) {                                     // reads of step+2
                if (step + 2 < S) a_nxt = *reinterpret_cast<const v4i *>(ap);
                b_raw = read_raw(bp);
                ap += arow_l; bp += wpitch;
            }
#pragma unroll
            for (int t = 0; t < kBand; ++t) {
                const int i = step - t;
                if (i >= 0 && i < S)
                    acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[i & (kBand - 1)], b_cur, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < kBand; ++g) {                        // one MFMA, then up to four of the others
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x106, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (step + 1 < S) ring[(step + 1) & (kBand - 1)] = a_fin;
            b_cur = b_nxt;
        }
    } else {
        const int nsteps = kBand + s - 1;
        for (int cbase = 0; cbase < nsteps; cbase += kBand) {
#pragma unroll
            for (int u = 0; u < kBand; ++u) {
                const int step = cbase + u;
                const int ia = step < s ? step : s;                   // row s of the operand table is all zero
                ring[u] = *reinterpret_cast<const v4i *>(ap + ia * arow_l);
                const v4i b = align_raw(read_raw(bp + step * wpitch), sh);
#pragma unroll
                for (int t = 0; t < kBand; ++t)
                    acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[(u - t) & (kBand - 1)], b, acc[t], 0, 0, 0);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Phase 2: angle-major sweep of one group.  Every wavefront takes work items wv, wv+W, ... and scores
// the kBand x 16 x 15 results of each item:
//   pass A (always, branch-free): float32 estimate of this lane's kBand x 4 values and their maximum;
//   pass B (only if the item can hold an arg-max candidate: its maximum is within kMargin of the
//   running maximum, or an estimate is NaN = degenerate window/template): estimates within kMargin
//   of the running maximum are queued for exact evaluation.  |estimate - value| <= 4e-7, so the
//   true arg-max is always queued.
// ---------------------------------------------------------------------------------------------
// PAIRED (at most 7 angles, one group): the 16 MFMA slots hold the angles + the all-ones template twice;
// the lanes of slots 8..15 read the operand table four rows earlier, so their accumulators belong to the
// output rows kBand below those of slots 0..7 - an item covers 2 * kBand rows with the same kBand
// accumulators, i.e. nearly half the MFMAs, window fragments and LDS traffic per output row.
template <int S, int kBand, bool PAIRED>
__device__ __forceinline__ Score ph_sweep(Score sc, int a0, int Kg, long long *dbg_cycles)
{
    static_assert(!PAIRED || kBand == 4, "the operand table of the paired mode is shifted by four rows");
    SID_PHASE_LOCALS;
    const uint8_t *afrag = smem + G.u_off;
    const uint8_t *win = smem + G.win_off;
    const u32 *sii = reinterpret_cast<const u32 *>(smem + G.sii_off);
    uint4 *queue = reinterpret_cast<uint4 *>(smem + G.queue_off);
    const int s = S > 0 ? S : G.s, rh = G.rh, rw = G.rw, wpitch = G.wpitch, arow = G.arow;
    const double nd = G.nd;
    const bool zero_a = (s <= 48) && q_l == 3;
    // lanes of k-group 3 sit on a zero row with stride 0
    const uint8_t *abase = PAIRED ? (zero_a ? afrag : afrag + (n_l >= 8 ? 0 : 4) * arow + q_l * 128 + (n_l & 7) * 16)
                                  : (zero_a ? afrag + s * arow : afrag + lane * 16);
    const int arow_l = zero_a ? 0 : arow;
    constexpr int kRows = PAIRED ? 2 * kBand : kBand;                  // output rows per work item
    const int yq = PAIRED ? kBand * (q_l >> 1) : 0;                    // this lane's accumulators: rows y0 + yq + t
    // accumulator register r of this lane belongs to slot 4 q + r: angle index (the ones slot never is < Kg)
    auto angle_of = [&](int r) { return PAIRED ? ((4 * q_l + r) & 7) : 4 * q_l + r; };

    // estimate of slot r at a placement: float(acc * A4[r] - Sw' * B4[r]) * rsq(dI), with the template's
    // 1/sqrt(dT) folded into both per-slot factors (NaN for a constant template => always a candidate)
    double A4[4], B4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int a = angle_of(r);
        const double rTa = a < Kg ? (m->constT[a0 + a] ? (double)NAN : m->rTd[a0 + a]) : 0.0;
        A4[r] = nd * rTa;
        B4[r] = (a < Kg ? m->sTd[a0 + a] : 0.0) * rTa;
    }
    const u32 livebits = (angle_of(0) < Kg ? 1u : 0u) | (angle_of(1) < Kg ? 2u : 0u) |
                         (angle_of(2) < Kg ? 4u : 0u) | (angle_of(3) < Kg ? 8u : 0u);
#define STAMP2(k) do { if (dbg_cycles && tid == 0) dbg_cycles[k] = (long long)clock64(); } while (0)

    const int nbands = (rh + kRows - 1) / kRows, ntx = (rw + 15) / 16;
    for (int item = wv; item < nbands * ntx; item += kWavesM) {
        const int band = item / ntx, xt = item - band * ntx;
        const int y0 = band * kRows, x0 = xt * 16;
        // k-group 3 multiplies the all-zero template columns (s <= 48): its lanes read the window bytes of
        // k-group 0 (same addresses = LDS broadcast, and the window pitch needs no room for columns 48..63)
        const u32 sbyte = (u32)(x0 + n_l + 16 * (zero_a ? 0 : q_l));
        const u32 sh = sbyte & 3u;
        if (item == wv) STAMP2(10);
        v4i acc[kBand];
        sweep_item<(PAIRED && S > 0) ? S + 4 : S, kBand>(acc, abase, arow_l, win + y0 * wpitch + (sbyte & ~3u), wpitch,
                                                         PAIRED ? s + 4 : s, sh);
        if (item == wv) STAMP2(11);

        {
            const float gm = key2f(m->gmax_key);
            sc.lmax = gm > sc.lmax ? gm : sc.lmax;
        }
        const int x = x0 + n_l;
        const bool xok = x < rw;
        // operands of all rows first (one LDS round trip), then straight-line arithmetic
        int swp[kBand];
        u32 siiv[kBand];
#pragma unroll
        for (int t = 0; t < kBand; ++t) {
            // the ones slot = sum of w': slot 15 (lanes 48..63, register 3), or slots 7 / 15 in paired mode
            swp[t] = __builtin_amdgcn_ds_bpermute((PAIRED ? n_l + 16 + 32 * (q_l >> 1) : n_l + 48) << 2, acc[t][3]);
            const int y = y0 + yq + t;
            siiv[t] = sii[(xok & (y < rh)) ? y * rw + x : 0];
        }
        auto estimate_row = [&](int t, float (&e)[4]) {
            const double swd = (double)swp[t], siid = (double)siiv[t];
            const double dI = nd * siid - swd * swd;                  // exact
            float rIf = __builtin_amdgcn_rsqf((float)dI);
            rIf = (2.0 * dI <= nd) ? NAN : rIf;                       // (nearly) flat window: the exact path decides
            const u32 lv = (xok & (y0 + yq + t < rh)) ? livebits : 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (float)__builtin_fma((double)acc[t][r], A4[r], -(swd * B4[r])) * rIf;
                e[r] = ((lv >> r) & 1u) ? v : -INFINITY;
            }
        };
        float itemmax = -INFINITY;
        u32 nanbits = 0;
#pragma unroll
        for (int t = 0; t < kBand; ++t) {
            float e[4];
            estimate_row(t, e);
#pragma unroll
            for (int r = 0; r < 4; ++r) { itemmax = fmaxf(itemmax, e[r]); nanbits |= (e[r] != e[r]) ? 1u : 0u; }
        }
        const float wavemax = wave_max_dpp(itemmax);
        const bool need_b = (wavemax >= sc.lmax - kMargin) || (__ballot(nanbits != 0) != 0ull);   // wavefront-uniform
        sc.lmax = fmaxf(sc.lmax, wavemax);
        if (need_b) {
            const float thr = sc.lmax - kMargin;
            u32 ovf = 0;                                              // candidates that found the queue full
#pragma unroll
            for (int t = 0; t < kBand; ++t) {
                float e[4];
                estimate_row(t, e);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (e[r] != -INFINITY && !(e[r] < thr)) {         // live, and true for NaN
                        const int a = angle_of(r);
                        const int key = ((a0 + a) * rh + y0 + yq + t) * rw + x;
                        const u32 slot = atomicAdd(&m->qcount, 1u);
                        if (slot < (u32)kQueueCap) queue[slot] = make_uint4((u32)acc[t][r], (u32)swp[t], siiv[t], (u32)key);
                        else ovf |= 1u << (4 * t + r);
                    }
                }
            }
            // Cold path (queue full: massive exact ties, flat windows): evaluate the left-over candidates
            // here, one at a time.  A rolled loop with one copy of the double-precision sequence and
            // operand selection by compare chains - no call, so the sweep stays a leaf function and
            // needs no callee-saved registers.
            if (ovf) {
#pragma unroll 1
                for (int b = 0; b < 4 * kBand; ++b) {
                    if (!((ovf >> b) & 1u)) continue;
                    const int t = b >> 2, r = b & 3;
                    int pv = 0, sw = 0; u32 si = 0;
#pragma unroll
                    for (int tt = 0; tt < kBand; ++tt) {
                        sw = t == tt ? swp[tt] : sw;
                        si = t == tt ? siiv[tt] : si;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) pv = b == 4 * tt + rr ? acc[tt][rr] : pv;
                    }
                    const int a = angle_of(r);
                    const int key = ((a0 + a) * rh + y0 + yq + t) * rw + x;
                    take_better(sc, exact_from_sums(pv, sw, si, nd, m->sTd[a0 + a], m->rTd[a0 + a], m->constT[a0 + a] != 0), key);
                }
            }
        }
        if (lane == 0) atomicMax(&m->gmax_key, f2key(sc.lmax));
        if (item == wv) STAMP2(12);
    }
#undef STAMP2
    __syncthreads();
    // exact (double) evaluation of the queued candidates of this group
    {
        const int nq = (int)(m->qcount < (u32)kQueueCap ? m->qcount : (u32)kQueueCap);
        for (int idx = tid; idx < nq; idx += kBlockM) {
            const uint4 rec = queue[idx];
            const int key = (int)rec.w;
            const int a = key / G.npos;
            take_better(sc, exact_from_sums((int)rec.x, (int)rec.y, rec.z, nd, m->sTd[a], m->rTd[a], m->constT[a] != 0), key);
        }
    }
    __syncthreads();                                                   // operands / queue are rewritten by the next group
    if (tid == 0) m->qcount = 0;
    return sc;
}

// ---------------------------------------------------------------------------------------------
// Phase 4: NCC matrix of the winning angle, row-major MFMA (M = 16 output rows): ph_winner_stage builds the
// operands, ph_winner runs the tiles.
// trow[g][16 + i][16 B] holds the winner's template, trow1 the all-ones one; zero rows around.
// ---------------------------------------------------------------------------------------------
template <int S>
__device__ __noinline__ void ph_winner_stage(const double *rot4, const uint16_t *samp, long long rows1, long long cols1, int ka)
{
    SID_PHASE_LOCALS;
    const uint8_t *patch = smem + G.patch_off;
    uint8_t *trow = smem + G.u_off;
    uint8_t *trow1 = trow + G.trow_bytes;
    const int s = S > 0 ? S : G.s;
    const int trows = s + kTrowPad;
    for (int idx = tid; idx < 2 * G.trow_bytes / 4; idx += kBlockM) reinterpret_cast<u32 *>(trow)[idx] = 0;
    __syncthreads();
    {
        if (tid < 4) m->rot[0][tid] = rot4[tid];
        __syncthreads();
        const SampleGeom g = sample_geom(G, s, rows1, cols1, patch);
        const int pdump = 0;                                           // row 0 of k-group 0 is never read
        auto put = [&](int i, int j, int v, bool take = true) {
            const int off = take ? ((j >> 4) * trows + 16 + i) * 16 + (j & 15) : pdump;
            trow[off] = (uint8_t)(v ^ 0x80);
            trow1[off] = take ? 1 : 0;
        };
        if (samp) {                                                    // caller checked table_usable
            const int sp = samp_pitch(s), nq = sp >> 2, upa = s * nq;
            const u32 tailmask = (s & 3) ? (1u << (8 * (s & 3))) - 1u : 0xffffffffu;
            const uint2 *ta = reinterpret_cast<const uint2 *>(samp + (size_t)ka * s * sp);
            for (int u = tid; u < upa; u += kBlockM) {
                const int i = u / nq, jq = u - i * nq;
                const u32 vm = jq == nq - 1 ? tailmask : 0xffffffffu;
                const u32 raw = gather_quad(patch, ta[u]);
                const int off = ((jq >> 2) * trows + 16 + i) * 16 + (jq & 3) * 4;
                *reinterpret_cast<u32 *>(trow + off) = (raw ^ 0x80808080u) & vm;
                *reinterpret_cast<u32 *>(trow1 + off) = 0x01010101u & vm;
            }
        } else
        for (int k0 = 0; k0 * g.ngrp < s; k0 += kRowsPerThread) {
            int st = 0, stt = 0, sz = 0;
            const u32 db = sample_fast5<false>(g, m->rot[0], true, k0, st, stt, sz,
                                               [&](int, int i, int v, bool take) { put(i, g.j, v, take); });
            for (int u = 0; u < kRowsPerThread; ++u)
                if ((db >> u) & 1u) { const int i = g.ig + (k0 + u) * g.ngrp; put(i, g.j, sample_exact(g, m->rot[0], i, g.j, ka)); }
        }
    }
}

template <int S>
__device__ __noinline__ void ph_winner(int ka, long long *dbg_cycles)
{
    SID_PHASE_LOCALS;
    const uint8_t *win = smem + G.win_off;
    const u32 *sii = reinterpret_cast<const u32 *>(smem + G.sii_off);
    const uint8_t *trow = smem + G.u_off;
    const uint8_t *trow1 = trow + G.trow_bytes;
    float *ccm = reinterpret_cast<float *>(smem + G.u_off + 2 * G.trow_bytes);
    const int s = S > 0 ? S : G.s, rh = G.rh, rw = G.rw, wpitch = G.wpitch;
    const double nd = G.nd;
    const int trows = s + kTrowPad;
    __syncthreads();                                                   // the staged operands (and their fixes) are complete
    if (dbg_cycles && tid == 0) dbg_cycles[13] = (long long)clock64();
    const double sT = m->sTd[ka], rT = m->rTd[ka];
    const bool cT = m->constT[ka] != 0;
    const int nty = (rh + 15) / 16, ntx = (rw + 15) / 16;
    for (int item = wv; item < nty * ntx; item += kWavesM) {
        const int yt = item / ntx, xt = item - yt * ntx;
        const int y0 = yt * 16, x0 = xt * 16;
        const int rows_here = (rh - y0) < 16 ? (rh - y0) : 16;
        const int nsteps = rows_here + s - 1;
        const u32 sbyte = (u32)(x0 + n_l + 16 * ((s <= 48 && q_l == 3) ? 0 : q_l));   // k-group 3: zero template columns
        const u32 sh = sbyte & 3u;
        const uint8_t *bp = win + y0 * wpitch + (sbyte & ~3u);
        // lane (m = n_l, g = q_l): template row i = step - m  ->  trow[g][16 + step - m]
        const uint8_t *ta = trow + (q_l * trows + 16 - n_l) * 16;
        const uint8_t *ta1 = trow1 + (q_l * trows + 16 - n_l) * 16;
        // even and odd steps accumulate separately: four independent MFMA chains instead of two
        v4i accT = {0, 0, 0, 0}, accS = {0, 0, 0, 0}, accT1 = {0, 0, 0, 0}, accS1 = {0, 0, 0, 0};
        // Software pipeline on running pointers: the LDS reads of step k+2 are issued before the MFMAs of
        // step k, three operand sets used in turn.  Five LDS instructions per step keep at most 15 reads in
        // flight - the lgkmcnt counter saturates at 15, beyond that the compiler must wait for everything.
        // S > 0: fixed trip count (a full 16-row tile), fully unrolled, so every wait names exactly the reads
        // of one step; a partial last tile runs the same steps against the zero rows behind the template
        // (kTrowPad).  Reads past the last step are never used.
        struct Ops { Raw5 w; v4i at, a1; };
        auto issue = [&]() {
            Ops o;
            o.w = read_raw(bp);
            o.at = *reinterpret_cast<const v4i *>(ta);
            o.a1 = *reinterpret_cast<const v4i *>(ta1);
            bp += wpitch; ta += 16; ta1 += 16;
            return o;
        };
        auto mma = [&](const Ops &o, v4i &aT, v4i &aS) {
            const v4i bb = align_raw(o.w, sh);
            aT = __builtin_amdgcn_mfma_i32_16x16x64_i8(o.at, bb, aT, 0, 0, 0);
            aS = __builtin_amdgcn_mfma_i32_16x16x64_i8(o.a1, bb, aS, 0, 0, 0);
        };
        Ops o0 = issue(), o1 = issue(), o2;
        constexpr int kStepsFixed = S > 0 ? 16 + S - 1 : 0;
        // six steps per trip (three operand sets x even / odd accumulators), leaving the loop after the last step
#define SID_WINNER_SIX(step, n)                                                                                        \
        o2 = issue(); __builtin_amdgcn_sched_barrier(0); mma(o0, accT, accS); __builtin_amdgcn_sched_barrier(0);      \
        if ((step) + 1 >= (n)) break;                                                                                  \
        o0 = issue(); __builtin_amdgcn_sched_barrier(0); mma(o1, accT1, accS1); __builtin_amdgcn_sched_barrier(0);    \
        if ((step) + 2 >= (n)) break;                                                                                  \
        o1 = issue(); __builtin_amdgcn_sched_barrier(0); mma(o2, accT, accS); __builtin_amdgcn_sched_barrier(0);      \
        if ((step) + 3 >= (n)) break;                                                                                  \
        o2 = issue(); __builtin_amdgcn_sched_barrier(0); mma(o0, accT1, accS1); __builtin_amdgcn_sched_barrier(0);    \
        if ((step) + 4 >= (n)) break;                                                                                  \
        o0 = issue(); __builtin_amdgcn_sched_barrier(0); mma(o1, accT, accS); __builtin_amdgcn_sched_barrier(0);      \
        if ((step) + 5 >= (n)) break;                                                                                  \
        o1 = issue(); __builtin_amdgcn_sched_barrier(0); mma(o2, accT1, accS1); __builtin_amdgcn_sched_barrier(0);
        if (S > 0) {
#pragma unroll
            for (int step = 0; step < kStepsFixed; step += 6) { SID_WINNER_SIX(step, kStepsFixed) }
        } else {
            for (int step = 0; step < nsteps; step += 6) { SID_WINNER_SIX(step, nsteps) }
        }
#undef SID_WINNER_SIX
        accT += accT1; accS += accS1;
        const int x = x0 + n_l;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = y0 + 4 * q_l + r;
            if (y < rh && x < rw)
                ccm[y * rw + x] = exact_from_sums(accT[r], accS[r], sii[y * rw + x], nd, sT, rT, cT);
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Phase 5: Hessian at the peak (pmlib.py:36-59, :167) and the optional MCC normalisation.
// hes aliases sii.  Returns h, r in m->red_f[0..1].
// ---------------------------------------------------------------------------------------------
template <int PRIO = -1, bool BIG = false>
__device__ __noinline__ void ph_hessian(unsigned flags, int iy, int ix, float best_r, float *dbg_ccm, float *dbg_hes,
                                        long long dbg_cap, long long *dbg_cycles)
{
    if (PRIO >= 0) __builtin_amdgcn_s_setprio(PRIO);
    SID_PHASE_LOCALS;
    float *hes = tab_ptr<BIG, float>(G, smem, G.hes_off);
    const float *ccm = tab_ptr<BIG, float>(G, smem, G.ccm_off);
    u32 *hist4 = reinterpret_cast<u32 *>(smem + G.u_off);             // winner operands are dead: 4 KB of histograms
    u32 *medlist = hist4 + 1024;                                       // + 1 KB of keys (2 * trow_bytes >= 5 KB for every s)
    const int rh = G.rh, rw = G.rw, npos = G.npos;
    float rr = best_r;
    if (flags & 4u) {                                                  // mcc_norm (pmlib.py:171-172): on the unsmoothed matrix
        double cx = 0.0, cxx = 0.0;
        float cmin = INFINITY, cmax = -INFINITY;
        for (int idx = tid; idx < npos; idx += kBlockM) {
            const float c = ccm[idx];
            const double v = (double)c;
            cx += v; cxx += v * v;
            cmin = fminf(cmin, c); cmax = fmaxf(cmax, c);
        }
        block_sum2(cx, cxx, m);
        const float sd = std_from_sums(cx, cxx, npos);
        const float med = block_median_fast(ccm, npos, m, hist4, medlist, f2key(cmin), f2key(cmax));
        rr = (best_r - med) / sd;
    }
    if (flags & 2u) {
        // hes_smth (pmlib.py:46-47): scipy.ndimage.gaussian_filter(ccm, 1) in place - per axis a radius-4
        // kernel in double ('reflect' boundary; centre tap first, then the pairs from the outermost inwards,
        // as scipy's correlate1d does for a symmetric kernel), rounded to float32 after each axis
        float *tmp = hes;                                              // the Hessian buffer is free until the gradient pass
        float *cw = tab_ptr<BIG, float>(G, smem, G.ccm_off);
        const double w4 = m->gw[4], w3 = m->gw[3], w2 = m->gw[2], w1 = m->gw[1], w0 = m->gw[0];
        auto refl = [](int q, int n) { while (q < 0 || q >= n) { if (q < 0) q = -q - 1; if (q >= n) q = 2 * n - 1 - q; } return q; };
        __syncthreads();
        for (int pass = 0; pass < 2; ++pass) {
            const float *src = pass == 0 ? ccm : tmp;
            float *dst = pass == 0 ? tmp : cw;
            const int n = pass == 0 ? rh : rw, stride = pass == 0 ? rw : 1;
            for (int idx = tid; idx < npos; idx += kBlockM) {
                const int y = idx / rw, x = idx - y * rw;
                const int p = pass == 0 ? y : x;
                const float *line = src + (pass == 0 ? x : y * rw);
                double acc = (double)line[p * stride] * w4;
                acc += ((double)line[refl(p - 4, n) * stride] + (double)line[refl(p + 4, n) * stride]) * w0;
                acc += ((double)line[refl(p - 3, n) * stride] + (double)line[refl(p + 3, n) * stride]) * w1;
                acc += ((double)line[refl(p - 2, n) * stride] + (double)line[refl(p + 2, n) * stride]) * w2;
                acc += ((double)line[refl(p - 1, n) * stride] + (double)line[refl(p + 1, n) * stride]) * w3;
                dst[idx] = (float)acc;
            }
            __syncthreads();
        }
    }
    double sx = 0.0, sxx = 0.0;
    float hmin = INFINITY, hmax = -INFINITY;
    {
        // np.gradient twice (unit spacing, one-sided edges) without branches:
        //   g(j) = (f[min(j+1,n-1)] - f[max(j-1,0)]) * (0 < j < n-1 ? 0.5 : 1),  d2(k) = same formula on g
        auto d2 = [&](const float *f, int stride, int k, int n) {
            const int kp = k + 1 < n ? k + 1 : n - 1, km = k > 0 ? k - 1 : 0;
            const int kpp = kp + 1 < n ? kp + 1 : n - 1, kpm = kp > 0 ? kp - 1 : 0;
            const int kmp = km + 1 < n ? km + 1 : n - 1, kmm = km > 0 ? km - 1 : 0;
            const float fa = f[kpp * stride], fb = f[kpm * stride], fc = f[kmp * stride], fd = f[kmm * stride];
            const float gp = (fa - fb) * ((kp > 0 && kp < n - 1) ? 0.5f : 1.0f);
            const float gm = (fc - fd) * ((km > 0 && km < n - 1) ? 0.5f : 1.0f);
            return (gp - gm) * ((k > 0 && k < n - 1) ? 0.5f : 1.0f);
        };
        // Interior placements (two or more away from every edge: 80-90 % of the matrix) need no index clamping
        // and only five reads: g(k+1) = (f[k+2] - f[k]) / 2 and g(k-1) = (f[k] - f[k-2]) / 2 are central differences
        // themselves, d2 = (g(k+1) - g(k-1)) / 2 - the same float32 operations np.gradient performs there.
        // The frame of width two around them takes the general form.  Four placements per thread in flight.
        constexpr int kHes = 4;
        auto emit = [&](int idx, float d2x, float d2y) {
            const double hh = (double)d2x * (double)d2x + (double)d2y * (double)d2y;
            const float hv = (float)sqrt(hh);                          // hypotf: double sqrt, narrowed
            hes[idx] = hv;
            sx += (double)hv; sxx += (double)hv * (double)hv;
            hmin = fminf(hmin, hv); hmax = fmaxf(hmax, hv);
        };
        const int iw = rw - 4, ih = rh - 4;
        if (iw > 0 && ih > 0) {
            const u32 imagic = 0xffffffffu / (u32)iw + 1u;
            const int nin = iw * ih;
            for (int base = 0; base < nin; base += kHes * kBlockM) {
                float d2xv[kHes], d2yv[kHes];
#pragma unroll
                for (int u = 0; u < kHes; ++u) {
                    const int q = base + u * kBlockM + tid;
                    const int qc = q < nin ? q : 0;
                    const int yy = (int)__umulhi((u32)qc, imagic), xx = qc - yy * iw;
                    const float *f = ccm + (yy + 2) * rw + (xx + 2);
                    const float c = f[0], xr = f[2], xl = f[-2], yd = f[2 * rw], yu = f[-2 * rw];
                    d2xv[u] = ((xr - c) * 0.5f - (c - xl) * 0.5f) * 0.5f;
                    d2yv[u] = ((yd - c) * 0.5f - (c - yu) * 0.5f) * 0.5f;
                }
#pragma unroll
                for (int u = 0; u < kHes; ++u) {
                    const int q = base + u * kBlockM + tid;
                    if (q < nin) { const int yy = (int)__umulhi((u32)q, imagic), xx = q - yy * iw; emit((yy + 2) * rw + xx + 2, d2xv[u], d2yv[u]); }
                }
            }
        }
        // frame: rows 0, 1, rh-2, rh-1 in full, columns 0, 1, rw-2, rw-1 of the rows between (everything when the
        // matrix is too small to have an interior)
        const bool has_in = iw > 0 && ih > 0;
        const int nfr = has_in ? 4 * rw + 4 * ih : npos;
        for (int q = tid; q < nfr; q += kBlockM) {
            int y, x;
            if (!has_in) { y = q / rw; x = q - y * rw; }
            else if (q < 4 * rw) { const int rr = q / rw; x = q - rr * rw; y = rr < 2 ? rr : rh - 4 + rr; }
            else { const int e = q - 4 * rw; const int yy = e >> 2, cc = e & 3; y = yy + 2; x = cc < 2 ? cc : rw - 4 + cc; }
            emit(y * rw + x, d2(ccm + y * rw, 1, x, rw), d2(ccm + x, rw, y, rh));
        }
    }
    if (dbg_cycles && tid == 0) dbg_cycles[19] = (long long)clock64();
    __syncthreads();
    if (dbg_ccm || dbg_hes) {
        for (int idx = tid; idx < npos && idx < dbg_cap; idx += kBlockM) {
            if (dbg_ccm) dbg_ccm[idx] = ccm[idx];
            if (dbg_hes) dbg_hes[idx] = hes[idx];
        }
    }
    if (dbg_cycles && tid == 0) dbg_cycles[14] = (long long)clock64();
    float h = hes[iy * rw + ix];
    if (flags & 1u) {
        // one reduction round for the two sums and the key range (instead of one for the sums and one for the range)
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
        u32 *red_max = reinterpret_cast<u32 *>(m->red_f);
        const double wsx = wave_sum_dpp_d(sx), wsxx = wave_sum_dpp_d(sxx);
        const u32 wmin = wave_umin_dpp(f2key(hmin)), wmax = ~wave_umin_dpp(~f2key(hmax));
        if (lane == 0) { m->rot[w][0] = wsx; m->rot[w][1] = wsxx; m->red_i[w] = (int)wmin; red_max[w] = wmax; }
        for (int i = tid; i < 1024; i += kBlockM) hist4[i] = 0;
        if (tid == 0) m->sel_cle = 0;
        __syncthreads();
        double tx = 0.0, txx = 0.0;
        u32 kmin = 0xffffffffu, kmax = 0u;
        for (int q = 0; q < kWavesM; ++q) {
            tx += m->rot[q][0]; txx += m->rot[q][1];
            const u32 a = (u32)m->red_i[q], b = red_max[q];
            kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
        }
        const float sd = std_from_sums(tx, txx, npos);
        if (dbg_cycles && tid == 0) dbg_cycles[20] = (long long)clock64();
        const float med = block_median_ranged(hes, npos, m, hist4, medlist, kmin, kmax, dbg_cycles);
        h = (h - med) / sd;
    }
    __syncthreads();
    if (tid == 0) { m->red_f[0] = h; m->red_f[1] = rr; }
    __syncthreads();
}

// hypotf as NumPy's float32 np.hypot evaluates it (glibc: (float)sqrt((double)x * x + (double)y * y)), by a shorter route
// with the same result: sqrt(hh) from v_rsq_f64 and two coupled Newton steps (within ~1 ulp of the correctly rounded
// double); (float) of it equals (float) of the IEEE square root unless a float32 rounding boundary - the middle of the 29
// discarded mantissa bits - lies that close, and those lanes (2.4e-7 of them) take the IEEE route.  Checked on the device
// against the IEEE route by sid_pm_debug_hypot_selftest.
__device__ __forceinline__ float hypot_spec(float x, float y)
{
    const double hh = (double)x * (double)x + (double)y * (double)y;
    return (float)sqrt(hh);
}
// Round 5, measured and NOT the default (-DSID_HYPOT_F32): the same value in FLOAT32 arithmetic.  The idea was that the float64
// form below costs 29 double-rate instructions per magnitude; it does not - plain float64 FMAs issue at the float32 rate on
// gfx950 - and the 38 float32 instructions of this form measured +-0 (15 and 3 angles) to +3 % (7 angles).  Kept because it is
// verified (0 differences against the specification on the device and in a NumPy emulation with +-1 ulp v_sqrt / v_rcp).  x^2 + y^2 = s + t exactly up to 2^-46 s (error-free products by FMA, TwoSum); r0 =
// v_sqrt_f32(s) is within 1 ulp; the residual e = (s - r0^2) + t is known to 2^-24 of itself, so the true root is r0 + z ulp(r0)
// with z = e / (2 r0 ulp) known to ~1e-6 (|z| <= ~3), and the correctly rounded float32 root is r0 + round(z) ulp - unless z lies
// within 1e-4 of a half-integer (that includes every case in which the specification's double rounding - to float64, then to
// float32 - could matter), r0 sits next to a power of two (the ulp changes there), or s is zero / tiny / huge / NaN: those lanes,
// ~2e-4 of them, take the specification's route.  sid_pm_debug_hypot_selftest compares both forms on 2^27 pairs on the device.
__device__ __forceinline__ float hypot_fast(float x, float y)
{
#ifdef SID_HYPOT_F32
    const float p = x * x, q = y * y;
    const float ep = __builtin_fmaf(x, x, -p), eq = __builtin_fmaf(y, y, -q);    // x^2 = p + ep, y^2 = q + eq (exact)
    const float s = p + q;
    const float bb = s - p;
    const float es = (p - (s - bb)) + (q - bb);                                    // TwoSum: p + q = s + es (exact)
    const float t = (es + ep) + eq;
    const float r0 = __builtin_amdgcn_sqrtf(s);
    const float e = __builtin_fmaf(-r0, r0, s) + t;
    const int ex = __builtin_amdgcn_frexp_expf(r0);                                // r0 = m 2^ex, m in [0.5, 1): ulp(r0) = 2^(ex - 24)
    const float z = __builtin_amdgcn_ldexpf(e * __builtin_amdgcn_rcpf(r0), 23 - ex);   // e / (2 r0) in ulps
    const float k = __builtin_rintf(z);
    const u32 rb = __float_as_uint(r0), mb = rb & 0x7fffffu;
    const bool slow = !(__builtin_fabsf(z - k) < 0.5f - 1e-4f) || (mb - 8u) > 0x7fffefu || !(s > 0x1p-60f && s < 0x1p60f);
    float out = __uint_as_float(rb + (u32)(int)k);                                 // k ulps on (same binade: checked)
    if (slow) out = (float)sqrt((double)x * (double)x + (double)y * (double)y);
    return out;
#else
    const double hh = (double)x * (double)x + (double)y * (double)y;
    const double y0 = __builtin_amdgcn_rsq(hh);
    double g = hh * y0, h = 0.5 * y0;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);                                        // sqrt(hh)
    const u32 lo29 = (u32)__double2loint(g) & 0x1fffffffu;
    // not (1e-30 < hh < 1e30): zero, NaN, and results near the float32 denormal range (other rounding boundaries)
    const bool slow = ((lo29 - (0x10000000u - 64u)) <= 128u) || !(hh > 1e-60 && hh < 1e60);
    float out = (float)g;
    if (slow) out = (float)sqrt(hh);
    return out;
#endif
}

// Bucket of a Hessian magnitude (>= 0) in the fixed histogram of ph_hessian_fast: 16 binades 2^-15 .. 2 with 128 buckets
// each (the top 7 mantissa bits, 0.5 % wide); monotone in the value, smaller / larger values clamp to the first / last
// bucket.  2048 16-bit counters packed in pairs (a matrix holds fewer than 65536 values).
constexpr int kHesBuckets = 2048;
__device__ __forceinline__ u32 hes_bucket(u32 bits)
{
    const int b = (int)(bits >> 16) - (112 << 7);
    return (u32)(b < 0 ? 0 : (b > kHesBuckets - 1 ? kHesBuckets - 1 : b));
}

// ---------------------------------------------------------------------------------------------
// Phase 5, short form (hes_norm with neither hes_smth nor mcc_norm - the reference's defaults - and G.hist_off set):
// same values as ph_hessian with two barriers instead of eight.  The magnitudes go into a FIXED log-spaced histogram
// (zeroed by the winner's staging) in the pass that computes them, so that no min / max reduction precedes the
// histogram; every wavefront scans the counters itself (no "wavefront 0 scans, barrier, broadcast"); the elements
// of the bucket holding the middle rank (and of the next non-empty one when the second middle value lies there) are
// compacted and ranked by wavefront 0, which also forms the result.  Buckets too full for the list (massive ties, flat
// matrices, magnitudes below 2^-15) fall back to the radix select.
// ---------------------------------------------------------------------------------------------
template <int PRIO = -1, bool BIG = false>
__device__ __noinline__ void ph_hessian_fast(unsigned flags, int iy, int ix, float best_r, float *dbg_ccm, float *dbg_hes,
                                             long long dbg_cap, long long *dbg_cycles)
{
    if (PRIO >= 0) __builtin_amdgcn_s_setprio(PRIO);
    SID_PHASE_LOCALS;
    float *hes = tab_ptr<BIG, float>(G, smem, G.hes_off);
    const float *ccm = tab_ptr<BIG, float>(G, smem, G.ccm_off);
    u32 *hist = reinterpret_cast<u32 *>(smem + G.hist_off);            // 2048 16-bit counters, zero on entry; m->sel_cle = 0
    u32 *list = hist + 1024;                                           // kMedList keys
    const int rh = G.rh, rw = G.rw, npos = G.npos;
    const bool norm = (flags & 1u) != 0;
    double sx = 0.0, sxx = 0.0;
    {
        auto d2 = [&](const float *f, int stride, int k, int n) {
            const int kp = k + 1 < n ? k + 1 : n - 1, km = k > 0 ? k - 1 : 0;
            const int kpp = kp + 1 < n ? kp + 1 : n - 1, kpm = kp > 0 ? kp - 1 : 0;
            const int kmp = km + 1 < n ? km + 1 : n - 1, kmm = km > 0 ? km - 1 : 0;
            const float fa = f[kpp * stride], fb = f[kpm * stride], fc = f[kmp * stride], fd = f[kmm * stride];
            const float gp = (fa - fb) * ((kp > 0 && kp < n - 1) ? 0.5f : 1.0f);
            const float gm = (fc - fd) * ((km > 0 && km < n - 1) ? 0.5f : 1.0f);
            return (gp - gm) * ((k > 0 && k < n - 1) ? 0.5f : 1.0f);
        };
        auto emit = [&](int idx, float hv) {
            hes[idx] = hv;
            sx += (double)hv; sxx += (double)hv * (double)hv;
            if (norm) { const u32 b = hes_bucket(__float_as_uint(hv)); atomicAdd(&hist[b >> 1], 1u << (16 * (b & 1u))); }
        };
        // Interior placements (two or more away from every edge): g(k+1) = (f[k+2] - f[k]) / 2 and g(k-1) = (f[k] - f[k-2]) / 2
        // are central differences themselves, d2 = (g(k+1) - g(k-1)) / 2 - the float32 operations np.gradient performs there.
        // A thread owns one column and a run of rows and slides down it: the rows y-2 .. y+1 of its column stay in registers
        // (one new vertical read per placement + the two horizontal neighbours) and nothing is divided to find a position.
        // Four placements are in flight (their square-root chains interleave); only the stores are predicated.
        const int iw = rw - 4, ih = rh - 4;
        if (iw > 0 && ih > 0) {
            int nseg = kBlockM / iw; nseg = nseg < 1 ? 1 : (nseg > ih ? ih : nseg);
            const int per = (ih + nseg - 1) / nseg;
            constexpr int kB = 4;
            for (int task = tid; task < iw * nseg; task += kBlockM) {  // (one pass unless the matrix is wider than the workgroup)
                const int seg = task / iw, xx = task - seg * iw;
                const int ya = 2 + seg * per, yb = (ya + per < rh - 2) ? ya + per : rh - 2;
                if (ya >= yb) continue;
                const float *col = ccm + xx + 2;                       // column x = xx + 2
                float ra = col[(ya - 2) * rw], rb = col[(ya - 1) * rw], rc = col[ya * rw], rd = col[(ya + 1) * rw];
                for (int y = ya; y < yb; y += kB) {
                    float e[kB], xl[kB], xr[kB], hvv[kB];
#pragma unroll
                    for (int u = 0; u < kB; ++u) {
                        const int yr = (y + u < rh - 3) ? y + u : rh - 3;   // (rows past the run are read where legal and dropped)
                        const float *f = col + yr * rw;
                        e[u] = f[2 * rw]; xl[u] = f[-2]; xr[u] = f[2];
                    }
                    const float cc[kB] = {rc, rd, e[0], e[1]}, uu[kB] = {ra, rb, rc, rd};
#pragma unroll
                    for (int u = 0; u < kB; ++u) {
                        const float d2x = ((xr[u] - cc[u]) * 0.5f - (cc[u] - xl[u]) * 0.5f) * 0.5f;
                        const float d2y = ((e[u] - cc[u]) * 0.5f - (cc[u] - uu[u]) * 0.5f) * 0.5f;
                        hvv[u] = hypot_fast(d2x, d2y);
                    }
#pragma unroll
                    for (int u = 0; u < kB; ++u)
                        if (y + u < yb) emit((y + u) * rw + xx + 2, hvv[u]);
                    ra = e[0]; rb = e[1]; rc = e[2]; rd = e[3];
                }
            }
        }
        const bool has_in = iw > 0 && ih > 0;
        const int nfr = has_in ? 4 * rw + 4 * ih : npos;
        for (int q0 = 0; q0 < nfr; q0 += 2 * kBlockM) {                // two frame placements per thread in flight
            float hv2[2]; int idx2[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int qq = q0 + u * kBlockM + tid, q = qq < nfr ? qq : 0;
                int y, x;
                if (!has_in) { y = q / rw; x = q - y * rw; }
                else if (q < 4 * rw) { const int rr = q / rw; x = q - rr * rw; y = rr < 2 ? rr : rh - 4 + rr; }
                else { const int e = q - 4 * rw; const int yy = e >> 2, cc = e & 3; y = yy + 2; x = cc < 2 ? cc : rw - 4 + cc; }
                idx2[u] = qq < nfr ? y * rw + x : -1;
                hv2[u] = hypot_fast(d2(ccm + y * rw, 1, x, rw), d2(ccm + x, rw, y, rh));
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) if (idx2[u] >= 0) emit(idx2[u], hv2[u]);
        }
    }
    if (norm) {
        const double wsx = wave_sum_dpp_d(sx), wsxx = wave_sum_dpp_d(sxx);
        if (lane == 0) { m->rot[wv][0] = wsx; m->rot[wv][1] = wsxx; }
    }
    if (dbg_cycles && tid == 0) dbg_cycles[19] = (long long)clock64();
    __syncthreads();                                                   // A: magnitudes, histogram, partial sums
    if (dbg_ccm || dbg_hes) {
        for (int idx = tid; idx < npos && idx < dbg_cap; idx += kBlockM) {
            if (dbg_ccm) dbg_ccm[idx] = ccm[idx];
            if (dbg_hes) dbg_hes[idx] = hes[idx];
        }
    }
    if (dbg_cycles && tid == 0) dbg_cycles[14] = (long long)clock64();
    if (!norm) {
        if (tid == 0) { m->red_f[0] = hes[iy * rw + ix]; m->red_f[1] = best_r; }
        return;
    }
    // ---- every wavefront: the bucket of the middle rank.  Lane l sums the counters of buckets 32 l .. 32 l + 31; the lane
    //      whose range holds rank k1 hands its 32 counters to lanes 0..31 through a private LDS slot, and a second
    //      wavefront prefix sum finds the bucket (two short scans instead of a 32-step search in one lane) ----
    u32 c[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 c4 = *reinterpret_cast<const uint4 *>(hist + 16 * lane + 4 * q);
        c[4 * q] = c4.x; c[4 * q + 1] = c4.y; c[4 * q + 2] = c4.z; c[4 * q + 3] = c4.w;
    }
    u32 tot = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += (c[q] & 0xffffu) + (c[q] >> 16);
    const u32 inc = wave_scan_dpp(tot), exc = inc - tot;
    const bool two = !(npos & 1);
    const u32 k1 = two ? (u32)(npos / 2 - 1) : (u32)(npos / 2);
    const bool mine = k1 >= exc && k1 < inc;                           // exactly one lane
    const int L = (int)__builtin_ctzll(__ballot(mine));                // wavefront-uniform
    u32 *xch = reinterpret_cast<u32 *>(m->rTd) + 16 * wv;             // (1 / sqrt(dT) and sum t' per angle are dead: 1 KB of scratch)
    if (mine) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4 *>(xch + 4 * q) = make_uint4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
    }
    const u32 base_e = (u32)__builtin_amdgcn_readlane((int)exc, L);    // values below bucket 32 L
    const u32 pk = xch[(lane & 31) >> 1];
    const u32 cq = lane < 32 ? ((lane & 1) ? pk >> 16 : pk & 0xffffu) : 0u;
    const u32 inc2 = wave_scan_dpp(cq) + base_e, exc2 = inc2 - cq;
    const int J = (int)__builtin_ctzll(__ballot(k1 < inc2));           // first bucket whose inclusive count passes k1
    const u32 bin1 = 32u * (u32)L + (u32)J;
    const u32 before = (u32)__builtin_amdgcn_readlane((int)exc2, J), cnt1 = (u32)__builtin_amdgcn_readlane((int)cq, J);
    // the second middle value of an even-sized set lies in the same bucket unless the first is the bucket's last element;
    // then the next three buckets join the list (the next non-empty one is among them unless the matrix is degenerate)
    const bool need2 = two && !(k1 + 1 - before < cnt1);
    const u32 bin_hi = bin1 + (need2 ? 3u : 0u);
    u32 cnt2 = 0;
    if (need2) {
#pragma unroll
        for (u32 d = 1; d <= 3; ++d) { const u32 bb = bin1 + d; if (bb < (u32)kHesBuckets) cnt2 += (hist[bb >> 1] >> (16 * (bb & 1u))) & 0xffffu; }
    }
    const u32 nlist = cnt1 + cnt2;
    if (dbg_cycles && tid == 0) dbg_cycles[22] = (long long)clock64();
    if (nlist > (u32)kMedList || (need2 && cnt2 == 0u)) {              // block-uniform: massive ties, gaps -> radix select
        double tx = 0.0, txx = 0.0;
        for (int q = 0; q < kWavesM; ++q) { tx += m->rot[q][0]; txx += m->rot[q][1]; }
        const float h = hes[iy * rw + ix];
        const float med = block_median(hes, npos, m, hist);
        if (tid == 0) { m->red_f[0] = (h - med) / std_from_sums(tx, txx, npos); m->red_f[1] = best_r; }
        return;
    }
    for (int i = tid; i < npos; i += kBlockM) {
        const u32 bits = __float_as_uint(hes[i]), b = hes_bucket(bits);
        if (b >= bin1 && b <= bin_hi) list[atomicAdd(&m->sel_cle, 1u)] = bits;  // (non-negative floats order like their bits)
    }
    __syncthreads();                                                   // B: the list is complete
    if (dbg_cycles && tid == 0) dbg_cycles[23] = (long long)clock64();
    if (wv != 0) return;
    // wavefront 0: rank the list (both buckets together: bucket order = value order), pick the middle value(s), finish
    double tx = 0.0, txx = 0.0;
    for (int q = 0; q < kWavesM; ++q) { tx += m->rot[q][0]; txx += m->rot[q][1]; }
    const float h = hes[iy * rw + ix];
    const u32 r1 = k1 - before, r2 = r1 + 1;
    u32 key1 = 0, key2 = 0;
    for (u32 base = 0; base < nlist; base += 64) {
        const u32 j = base + (u32)lane;
        const u32 mv = j < nlist ? list[j] : 0xffffffffu;
        u32 rank = 0;
        for (u32 t0 = 0; t0 < nlist; t0 += 4) {                        // (the list region holds kMedList entries: reads past nlist are harmless)
            const uint4 o = *reinterpret_cast<const uint4 *>(list + t0);
            rank += (t0 + 0 < nlist && (o.x < mv || (o.x == mv && t0 + 0 < j))) ? 1u : 0u;
            rank += (t0 + 1 < nlist && (o.y < mv || (o.y == mv && t0 + 1 < j))) ? 1u : 0u;
            rank += (t0 + 2 < nlist && (o.z < mv || (o.z == mv && t0 + 2 < j))) ? 1u : 0u;
            rank += (t0 + 3 < nlist && (o.w < mv || (o.w == mv && t0 + 3 < j))) ? 1u : 0u;
        }
        const unsigned long long h1 = __ballot(j < nlist && rank == r1), h2 = __ballot(j < nlist && rank == r2);
        if (h1) key1 = (u32)__builtin_amdgcn_readlane((int)mv, (int)__builtin_ctzll(h1));
        if (h2) key2 = (u32)__builtin_amdgcn_readlane((int)mv, (int)__builtin_ctzll(h2));
    }
    const float med = two ? (__uint_as_float(key1) + __uint_as_float(key2)) / 2.0f : __uint_as_float(key1);
    if (dbg_cycles && tid == 0) dbg_cycles[24] = (long long)clock64();
    if (tid == 0) { m->red_f[0] = (h - med) / std_from_sums(tx, txx, npos); m->red_f[1] = best_r; }
}

// BAND = 4: up to 768 threads, three wavefronts per SIMD (168 VGPRs).  BAND = 8: 256 threads, two per SIMD (the
// class whose LDS footprint admits two workgroups per CU anyway): an 8-row band halves the LDS bytes per MFMA.
template <int S, int BAND, bool PAIRED>
__global__ __launch_bounds__(BAND == 8 ? 512 : kMaxBlockM, BAND == 8 ? 2 : kOccM) void pm_kernel_mfma(const PMArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    MiscM *m = reinterpret_cast<MiscM *>(smem);
    Geo *G = reinterpret_cast<Geo *>(smem + kGeoOff);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pt = A.order[blockIdx.x];
    const int s = S > 0 ? S : A.img_size, K = A.n_angles;

    double *out = A.out + (int64_t)pt * 5;
    int32_t *oij = A.out_ij ? A.out_ij + (int64_t)pt * 3 : nullptr;
#define SID_STAMP(k) do { if (A.dbg_cycles && tid == 0) A.dbg_cycles[k] = (long long)clock64(); } while (0)
    SID_STAMP(0);

    // ---- window geometry (pmlib.py:200-202) ----
    const double c2fg = A.c2fg[pt], r2fg = A.r2fg[pt], border = A.border[pt];
    const int hws = (int)((double)s / 2.0);
    const double r0d = r2fg - hws - border, r1d = r2fg + hws + border + 1;
    const double c0d = c2fg - hws - border, c1d = c2fg + hws + border + 1;
    const bool finite = fabs(r0d) < 1e15 && fabs(r1d) < 1e15 && fabs(c0d) < 1e15 && fabs(c1d) < 1e15;
    const int64_t r0 = finite ? (int64_t)r0d : -1, r1e = finite ? (int64_t)r1d : -1;
    const int64_t c0 = finite ? (int64_t)c0d : -1, c1e = finite ? (int64_t)c1d : -1;
    const bool inside = finite && r0 >= 0 && c0 >= 0 && r1e <= A.rows2 && c1e <= A.cols2 &&
                        r1e - r0 >= s + 1 && c1e - c0 >= s + 1;
    if (!inside) {
        if (tid < 5) out[tid] = NAN;
        if (oij && tid < 3) oij[tid] = -1;
        return;
    }
    const int wh = (int)(r1e - r0), ww = (int)(c1e - c0);
    const int rh = wh - s + 1, rw = ww - s + 1, npos = rh * rw;
    // the launch was sized by the host for the image shape it classified the points with (pm_capi.hip
    // classify_points); a window that needs more LDS than that must never be touched
    if (mfma_lds_layout(wh, ww, s, BAND, PAIRED).total > A.lds_bytes) {
        if (tid == 0 && A.refused) atomicAdd(A.refused, 1);            // an error the host reports (PMArgs::refused), never a silent NaN
        if (tid < 5) out[tid] = NAN;
        if (oij && tid < 3) oij[tid] = -1;
        return;
    }
    if (tid == 0) {
        const MfmaLdsLayout L = mfma_lds_layout(wh, ww, s, BAND, PAIRED);
        const double c1 = A.c1[pt], r1 = A.r1[pt];
        G->band = PAIRED ? 2 * BAND : BAND;
        G->wh = wh; G->ww = ww; G->rh = rh; G->rw = rw; G->npos = npos; G->wpitch = L.wpitch; G->arow = L.arow;
        G->gpitch = L.gpitch; G->arow0 = L.arow0; G->tab_rows = L.tab_rows; G->ones_slot = PAIRED ? 7 : 15;
        G->s = s; G->K = K;
        G->win_off = L.win_off; G->sii_off = L.sii_off; G->u_off = L.u_off; G->patch_off = L.patch_off;
        G->hes_off = L.sii_off; G->ccm_off = L.u_off + 2 * L.trow_bytes;   // (the sums' LDS holds the Hessian magnitudes later)
        G->ppitch = L.ppitch; G->pdim = L.pdim; G->pradius = L.pradius; G->queue_off = L.queue_off;
        G->trow_bytes = L.trow_bytes;
        G->pr0 = (int)floor(r1) - L.pradius; G->pc0 = (int)floor(c1) - L.pradius;
        G->win_magic = 0xffffffffu / (u32)(L.wpitch >> 2) + 1u; G->patch_magic = 0xffffffffu / (u32)(L.ppitch >> 2) + 1u;
        G->rw_magic = 0xffffffffu / (u32)rw + 1u;
        G->r0 = r0; G->c0 = c0; G->c1 = c1; G->r1 = r1; G->nd = (double)(s * s);
        G->hist_off = 0; G->lin = (int)((A.flags >> 3) & 7u); G->pre = A.pre ? (unsigned long long)(A.pre + (size_t)pt * K * s * s) : 0ull;
        m->zero_flag = 0; m->gmax_key = 0x007fffffu /* f2key(-inf) */; m->qcount = 0;
        for (int k = 0; k < 5; ++k) m->gw[k] = A.gauss_w[k];
    }
    __syncthreads();

    ph_window(A.img2, A.rows2, A.cols2, A.stride2);
    __syncthreads();
    SID_STAMP(1);
#ifndef SID_ABLATE_SUMS
    ph_sums();
#endif
    SID_STAMP(2);
    ph_patch(A.img1, A.rows1, A.cols1, A.stride1);

    // offset table for the samplers, or null: on-the-fly sampling (block-uniform)
    const uint16_t *samp = (A.samp && table_usable(*G, A.rows1, A.cols1)) ? A.samp : nullptr;
    Score sc{-INFINITY, -INFINITY, 0x7fffffff};
    for (int a0 = 0; a0 < K; a0 += kAnglesPerGroup) {
        const int Kg = (K - a0) < kAnglesPerGroup ? (K - a0) : kAnglesPerGroup;
        ph_tpl_begin<S>(A.rot, a0, Kg, A.dbg_cycles);
        if (samp) ph_tpl_table<S>(samp, a0, Kg, A.rows1, A.cols1);
        else ph_tpl_general<S>(a0, Kg, A.rows1, A.cols1);
        if (samp && A.samp_nflag > 0) ph_tpl_fix<S>(samp, a0, Kg, A.rows1, A.cols1, -1);
        ph_tpl_end<S>(a0, Kg, A.dbg_templates, A.dbg_cycles);
        if (m->zero_flag) {                                            // pmlib.py:152-154
            if (tid < 5) out[tid] = NAN;
            if (oij && tid < 3) oij[tid] = -1;
            if (A.dbg_shape && tid == 0) { A.dbg_shape[0] = rh; A.dbg_shape[1] = rw; }
            return;
        }
        SID_STAMP(3);
#ifndef SID_ABLATE_SWEEP
        sc = ph_sweep<S, BAND, PAIRED>(sc, a0, Kg, A.dbg_cycles);
#else
        sc.bestv = 0.5f; sc.bestkey = 17;
#endif
        SID_STAMP(4);
    }

    // ---- P3: block arg-max (first angle, then first row-major index on ties) ----
    {
        const float wbest = wave_max_dpp(sc.bestv);                    // exact values: never NaN
        sc.bestkey = wave_min_dpp(sc.bestv == wbest ? sc.bestkey : 0x7fffffff);
        sc.bestv = wbest;
    }
    if (lane == 0) { m->red_f[wv] = sc.bestv; m->red_i[wv] = sc.bestkey; }
    __syncthreads();
    if (tid == 0) {
        float bv = m->red_f[0]; int bk = m->red_i[0];
        for (int w = 1; w < kWavesM; ++w) {
            const float ov = m->red_f[w]; const int ok = m->red_i[w];
            if (ov > bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
        }
        m->best_val = bv; m->best_key = bk;
    }
    __syncthreads();
    const float best_r = m->best_val;
    const int best_key = m->best_key;
    const int ka = best_key / npos;
    const int bidx = best_key - ka * npos;
    const int iy = bidx / rw, ix = bidx - iy * rw;
    SID_STAMP(5);

#ifndef SID_ABLATE_WINNER
    ph_winner_stage<S>(A.rot + 4 * ka, samp, A.rows1, A.cols1, ka);
    if (samp && A.samp_nflag > 0) ph_tpl_fix<S>(samp, 0, 0, A.rows1, A.cols1, ka);
    ph_winner<S>(ka, A.dbg_cycles);
#endif
    SID_STAMP(6);
    if (A.dbg_shape && tid == 0) { A.dbg_shape[0] = rh; A.dbg_shape[1] = rw; }
#ifndef SID_ABLATE_HESSIAN
    ph_hessian<>(A.flags, iy, ix, best_r, A.dbg_ccm, A.dbg_hes, A.dbg_cap, A.dbg_cycles);
#endif
    SID_STAMP(7);
    if (tid == 0) {
        out[0] = c2fg + ((double)ix - (double)(ww - s) / 2.0);
        out[1] = r2fg + ((double)iy - (double)(wh - s) / 2.0);
        out[2] = A.angles[ka];
        out[3] = (double)m->red_f[1];
        out[4] = (double)m->red_f[0];
        if (oij) { oij[0] = iy; oij[1] = ix; oij[2] = ka; }
    }
}

#include "pm_kernel_rp.inc"

#ifndef SID_OCC4_TU
// exact_from_sums_fast against exact_from_sums on pseudo-random sums of plausible windows (diagnostic):
// out[0] = evaluations, out[1] = results that differ, out[2] = evaluations that took the spec's route
__global__ void ncc_selftest_kernel(unsigned long long seed, int per_thread, int s, unsigned long long *out)
{
    unsigned long long st = seed ^ (0x9e3779b97f4a7c15ull * (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x + 1));
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    const double nd = (double)(s * s);
    unsigned long long bad = 0, slow = 0;
    for (int it = 0; it < per_thread; ++it) {
        // template: sum sT (re-centred), denominator dT = N S_TT - S_T^2 > 0
        const int sT = (int)(next() % (unsigned)(2 * 128 * s * s + 1)) - 128 * s * s;
        const double dTmin = 1.0, dT = dTmin + (double)(next() % (1ull << (20 + (int)(next() % 28))));
        const double rTd = 1.0 / sqrt(dT);
        const int swp = (int)(next() % (unsigned)(2 * 128 * s * s + 1)) - 128 * s * s;
        const unsigned long long sw2 = (unsigned long long)((long long)swp * swp);
        const u32 sii_min = (u32)((sw2 + (unsigned long long)(s * s) - 1) / (unsigned long long)(s * s));
        const u32 room = (u32)(s * s) * 16384u - sii_min;                 // S_I'I' <= N 128^2
        const u32 siiv = sii_min + (u32)(next() % (unsigned long long)(room + 1u)) % (1u << (1 + (int)(next() % 27)));
        const double dI = nd * (double)siiv - (double)swp * (double)swp;
        // numerator within (and sometimes just beyond) the Cauchy-Schwarz bound: nd p - swp sT = f sqrt(dI dT)
        const double f = ((double)(long long)(next() % 2400001ull) - 1200000.0) / 1000000.0;
        const double target = f * sqrt(dI > 0 ? dI : 0.0) * sqrt(dT);
        const int p = (int)llrint((target + (double)swp * (double)sT) / nd);
        const float a = exact_from_sums(p, swp, siiv, nd, (double)sT, rTd, false);
        const float b = exact_from_sums_fast(p, swp, siiv, nd, (double)sT, rTd, false);
        bad += __float_as_uint(a) != __float_as_uint(b) ? 1ull : 0ull;
        // (the count of spec-route evaluations is re-derived: same predicate as in exact_from_sums_fast)
        const double y = 1.0 / sqrt(dI > 0 ? dI : 1.0);
        const double q = ((nd * (double)p - (double)swp * (double)sT) * y) * rTd, aq = fabs(q);
        const u32 lo29 = (u32)__double2loint(q) & 0x1fffffffu;
        slow += (2.0 * dI <= nd || (lo29 - (0x10000000u - 64u)) <= 128u || fabs(aq - 1.0) < 1e-13 || fabs(aq - 1.125) < 1e-13) ? 1ull : 0ull;
    }
    atomicAdd(&out[0], (unsigned long long)per_thread);
    if (bad) atomicAdd(&out[1], bad);
    if (slow) atomicAdd(&out[2], slow);
}

// hypot_fast against hypot_spec on pseudo-random float32 pairs of the magnitudes second differences of NCC values take
// (and, every fourth sample, pairs constructed to land next to a float32 rounding boundary): out = evaluations, mismatches
__global__ void hypot_selftest_kernel(unsigned long long seed, int per_thread, unsigned long long *out)
{
    unsigned long long st = seed ^ (0x9e3779b97f4a7c15ull * (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x + 1));
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    unsigned long long bad = 0;
    for (int it = 0; it < per_thread; ++it) {
        const unsigned long long a = next(), b = next();
        // mantissas uniform, exponents 2^-40 .. 2^1 (and occasionally zero / denormal / huge)
        const int ea = 127 - (int)(a % 41) + 1, eb = 127 - (int)((a >> 8) % 41) + 1;
        float x = __uint_as_float(((u32)ea << 23) | (u32)(b & 0x7fffffu)), y = __uint_as_float(((u32)eb << 23) | (u32)((b >> 23) & 0x7fffffu));
        const u32 kind = (u32)(a >> 60);
        if (kind == 0) y = 0.0f;                                       // perfect squares: sqrt lands on a float exactly
        if (kind == 1) { x = 0.0f; y = 0.0f; }
        if (kind == 2) x = __uint_as_float((u32)(b & 0x7fffffu));      // denormal
        if (kind == 3) { x = __uint_as_float(0x7e000000u | (u32)(b & 0x7fffffu)); }
        if (a & (1ull << 40)) x = -x;
        if (a & (1ull << 41)) y = -y;
        bad += __float_as_uint(hypot_fast(x, y)) != __float_as_uint(hypot_spec(x, y)) ? 1ull : 0ull;
    }
    atomicAdd(&out[0], (unsigned long long)per_thread);
    if (bad) atomicAdd(&out[1], bad);
}

__global__ void rsqrt_kernel(const double *x, double *y, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 1.0 / sqrt(x[i]);
}

#endif  // SID_OCC4_TU

}  // namespace

#ifdef SID_OCC4_TU
// the only export of the second compilation: its kernels (slot groups, window pitches 104 and 112)
void (*rp_occ4_kernel(int img_size, int paired, int pitch))(const PMArgs)
{
    if ((pitch != 104 && pitch != 112) || (paired != 1 && paired != 2)) return nullptr;
    if (img_size == 34) return paired == 2 ? (pitch == 104 ? pm_kernel_rp<34, 4, 2, 104> : pm_kernel_rp<34, 4, 2, 112>)
                                           : (pitch == 104 ? pm_kernel_rp<34, 4, 1, 104> : pm_kernel_rp<34, 4, 1, 112>);
    if (img_size == 35) return paired == 2 ? (pitch == 104 ? pm_kernel_rp<35, 4, 2, 104> : pm_kernel_rp<35, 4, 2, 112>)
                                           : (pitch == 104 ? pm_kernel_rp<35, 4, 1, 104> : pm_kernel_rp<35, 4, 1, 112>);
    return nullptr;
}
#else
void (*rp_occ4_kernel(int img_size, int paired, int pitch))(const PMArgs);   // pm_kernel_rp_occ4.hip

int max_lds_bytes() { return 160 * 1024; }

int launch_rsqrt(const double *x, double *y, int64_t n, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n <= 0) return (int)hipSuccess;
    hipLaunchKernelGGL(rsqrt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n);
    return (int)hipGetLastError();
}

int launch_ncc_selftest(unsigned long long seed, int blocks, int per_thread, int s, unsigned long long *out, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ncc_selftest_kernel, dim3((unsigned)blocks), dim3(256), 0, st, seed, per_thread, s, out);
    return (int)hipGetLastError();
}

int launch_hypot_selftest(unsigned long long seed, int blocks, int per_thread, unsigned long long *out, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(hypot_selftest_kernel, dim3((unsigned)blocks), dim3(256), 0, st, seed, per_thread, out);
    return (int)hipGetLastError();
}

bool mfma_img_size_supported(int s) { return s >= 2 && s <= 64; }   // K = 64 template columns per matrix instruction

// The dynamic-LDS limit belongs to the function and the device, not to the stream: it is raised to the maximum once per
// (kernel, device) - always the maximum, so that launches of different footprints from different host threads cannot
// undercut each other - instead of on every launch.
static hipError_t allow_max_lds(void (*kern)(const PMArgs))
{
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const std::pair<const void *, int> key(reinterpret_cast<const void *>(kern), dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(key)) return hipSuccess;
    e = hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds_bytes());
    if (e == hipSuccess) done.insert(key);
    return e;
}

bool mfma_band8_supported(int s) { return s == 34 || s == 35; }

int launch_pm_mfma(const PMArgs &args, int lds_bytes, int nthreads, int band, bool paired, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (args.n_launch <= 0) return (int)hipSuccess;
    void (*kern)(const PMArgs) = nullptr;
    if (paired) {                                                      // at most 7 angles: one group, paired slots
        if (band != 4 || args.n_angles > kPairedMaxAngles) return (int)hipErrorInvalidValue;
        kern = args.img_size == 34 ? pm_kernel_mfma<34, 4, true> : args.img_size == 35 ? pm_kernel_mfma<35, 4, true>
                                                                                        : pm_kernel_mfma<0, 4, true>;
    } else if (band == 8) {
        if (!mfma_band8_supported(args.img_size) || nthreads != 256) return (int)hipErrorInvalidValue;
        kern = args.img_size == 34 ? pm_kernel_mfma<34, 8, false> : pm_kernel_mfma<35, 8, false>;
    } else if (band == 4) {
        kern = args.img_size == 34 ? pm_kernel_mfma<34, 4, false> : args.img_size == 35 ? pm_kernel_mfma<35, 4, false>
                                                                                         : pm_kernel_mfma<0, 4, false>;
    } else return (int)hipErrorInvalidValue;
    const hipError_t e = allow_max_lds(kern);
    if (e != hipSuccess) return (int)e;
    if (lds_bytes > max_lds_bytes() || (nthreads != 256 && nthreads != 768)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3(args.n_launch), dim3(nthreads), lds_bytes, st, args);
    return (int)hipGetLastError();
}

template <int S>
static void (*rp_kernel_for(int band, int paired, int pitch))(const PMArgs)
{
    if (paired == 2) return pitch == 1104 ? pm_kernel_rp<S, 4, 2, 1104> : pitch == 104 ? pm_kernel_rp<S, 4, 2, 104> : pitch == 112 ? pm_kernel_rp<S, 4, 2, 112> : pitch == 136 ? pm_kernel_rp<S, 4, 2, 136>
                          : pitch == 168 ? pm_kernel_rp<S, 4, 2, 168> : pitch == 0 ? pm_kernel_rp<S, 4, 2, 0> : nullptr;
    if (paired == 1) return pitch == 1104 ? pm_kernel_rp<S, 4, 1, 1104> : pitch == 104 ? pm_kernel_rp<S, 4, 1, 104> : pitch == 112 ? pm_kernel_rp<S, 4, 1, 112> : pitch == 136 ? pm_kernel_rp<S, 4, 1, 136>
                          : pitch == 168 ? pm_kernel_rp<S, 4, 1, 168> : pitch == 0 ? pm_kernel_rp<S, 4, 1, 0> : nullptr;
    if (band == 8) return pitch == 136 ? pm_kernel_rp<S, 8, 0, 136> : pitch == 168 ? pm_kernel_rp<S, 8, 0, 168> : pitch == 0 ? pm_kernel_rp<S, 8, 0, 0> : nullptr;
    return pitch == 104 ? pm_kernel_rp<S, 4, 0, 104> : pitch == 112 ? pm_kernel_rp<S, 4, 0, 112> : pitch == 136 ? pm_kernel_rp<S, 4, 0, 136> : pitch == 1104 ? pm_kernel_rp<S, 4, 0, 1104>
         : pitch == 168 ? pm_kernel_rp<S, 4, 0, 168> : pitch == 0 ? pm_kernel_rp<S, 4, 0, 0> : nullptr;
}

// true when launch_pm_rp carries an instantiation with this compile-time window pitch (0 = run-time pitch: always) and
// register budget (occ = 3 wavefronts per SIMD, or 4: pm_kernel_rp_occ4.hip)
bool rp_pitch_instantiated(int band, int paired, int pitch, int occ)
{
    if (occ == 4) return band == 4 && rp_occ4_kernel(34, paired, pitch) != nullptr;
    return rp_kernel_for<34>(band, paired, pitch) != nullptr;
}

int launch_pm_rp(const PMArgs &args, int lds_bytes, int nthreads, int band, int paired, int pitch, int occ, void *stream, bool big)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (args.n_launch <= 0) return (int)hipSuccess;
    if (!rp_size_supported(args.img_size)) return (int)hipErrorInvalidValue;
    if ((band != 4 && band != 8) || (band == 8 && nthreads != 256)) return (int)hipErrorInvalidValue;
    if ((occ != 3 && occ != 4) || (occ == 4 && (nthreads != 256 || band != 4))) return (int)hipErrorInvalidValue;
    if (paired && (band != 4 || args.n_angles > (paired == 2 ? kQuadMaxAngles : kPairedMaxAngles))) return (int)hipErrorInvalidValue;
    if (big && (band != 4 || paired || pitch || occ != 3)) return (int)hipErrorInvalidValue;
    void (*kern)(const PMArgs) = big ? (args.img_size == 34 ? pm_kernel_rp<34, 4, 0, 0, true> : pm_kernel_rp<35, 4, 0, 0, true>)
                               : occ == 4 ? rp_occ4_kernel(args.img_size, paired, pitch)
                               : args.img_size == 34 ? rp_kernel_for<34>(band, paired, pitch) : rp_kernel_for<35>(band, paired, pitch);
    if (!kern) return (int)hipErrorInvalidValue;
    const hipError_t e = allow_max_lds(kern);
    if (e != hipSuccess) return (int)e;
    if (lds_bytes > max_lds_bytes() || nthreads < 192 || nthreads > 768 || (nthreads & 63)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3(args.n_launch), dim3(nthreads), lds_bytes, st, args);
    return (int)hipGetLastError();
}
#endif  // SID_OCC4_TU

}  // namespace sid
