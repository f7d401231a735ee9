// pm_kernel_mfma.hip - the pattern-matching kernel on the gfx950 matrix cores.
//
// Same operator and same numerical specification as pm_kernel.hip (reference pmlib.py:117-212,
// :36-59; NCC specification in DESIGN.md section 3), but the 34x34 uint8 correlation runs on
// v_mfma_i32_16x16x64_i8 as an implicit correlation - no im2col is materialised:
//
//   bytes are re-centred (w' = w ^ 0x80, t' = t ^ 0x80 as int8); numer, dI and dT are covariances
//   and do not change under the shift, so the integer sums stay exact and the spec is untouched.
//
//   sweep ("angle-major"):  D[slot][x] += A[slot][c] * B[c][x]
//       M = 16 template slots: up to 15 trial angles + one all-ones template (gives sum w' = S_I')
//       N = 16 adjacent placements x0..x0+15 of one output row y
//       K = 64 window columns c of window row rho = y + i ; A[slot][c] = T'_slot[i][c] (0 for c >= s)
//     B[c][x] = W'[rho][x0 + x + c] is the same for every (y, i) with y + i = rho, so a wavefront
//     keeps a band of 8 output rows in accumulators and a ring of 8 template-row fragments in
//     registers: one window fragment built from LDS (5 ds_read_b32 + 4 v_alignbyte_b32) feeds 8 MFMAs.
//
//   winner ("row-major"):   the NCC matrix of the best angle is recomputed with M = 16 output rows
//     (A[m][c] = T'_best[rho - y0 - m][c]) and a second all-ones operand for S_I'.
//
//   S_II' = sum w'^2 comes from two running-sum passes over LDS (columns, then rows).
//   The double-precision normalisation of the spec is evaluated only for arg-max candidates picked
//   by a float32 pre-filter (|r~ - r| <= 4e-7 << margin 1e-5) and for the winner's matrix.
//
// One workgroup of 512 threads (8 wavefronts) per grid point.
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <math.h>
#include "pm_kernel.h"

namespace sid {

namespace {

typedef uint32_t u32;
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kBlockM = 256;          // threads per grid point in this kernel: 4 wavefronts
constexpr int kWavesM = kBlockM / 64;
constexpr int kBand = 4;             // output rows per sweep work item
constexpr int kSlots = 16;           // MFMA M: 15 angles + ones
constexpr int kAnglesPerGroup = 15;
constexpr float kMargin = 1e-5f;     // pre-filter margin (see header)

struct MiscM {                       // LDS offset 0, kMiscMfmaBytes reserved
    double red_d[8];
    float red_f[8];
    int red_i[8];
    u32 sel_key, sel_cle;
    int zero_flag;
    int best_key; float best_val;
    u32 gmax_key;                    // block-wide running max of the float32 estimates (ordered key)
    u32 qcount;                      // candidate queue fill
    int pad_;
    int isT[kSlots], isTT[kSlots];   // integer template sums of the current group
    double rTd[kMaxAngles];          // 1/sqrt(dT) per angle
    double sTd[kMaxAngles];          // sum t' per angle
    float rTf[kMaxAngles];
    int constT[kMaxAngles];
};
static_assert(sizeof(MiscM) <= kMiscMfmaBytes, "misc header too large");

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

// ---- small block utilities (same semantics as in pm_kernel.hip) ----
__device__ __forceinline__ double wave_sum_d(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ int wave_sum_i(int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

__device__ __forceinline__ double block_sum(double v, MiscM *m) {
    v = wave_sum_dpp_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) m->red_d[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kWavesM; ++w) t += m->red_d[w];
    return t;
}
__device__ __forceinline__ u32 block_min(u32 v, MiscM *m) {
    v = wave_umin_dpp(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) m->red_i[threadIdx.x >> 6] = (int)v;
    __syncthreads();
    u32 a = (u32)m->red_i[0];
#pragma unroll
    for (int w = 1; w < kWavesM; ++w) { const u32 b = (u32)m->red_i[w]; a = a < b ? a : b; }
    return a;
}
__device__ __forceinline__ u32 f2key(float f) { u32 b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float key2f(u32 k) { u32 b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; return __uint_as_float(b); }

// k-th smallest (0-based) of v[0..n) by 4 x 8-bit radix select.  The four histograms are zeroed
// up front (hist4: 4 x 256 counters, 16-byte aligned scratch); per pass: all threads bin their elements (LDS atomics), barrier, wavefront 0 scans the
// 256 bins (4 per lane + a wavefront prefix sum) and publishes the digit, barrier.
__device__ u32 block_select(const float *v, int n, u32 k, MiscM *m, u32 *hist4, u32 *count_le) {
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
__device__ float block_median(const float *v, int n, MiscM *m, u32 *hist4) {
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

// np.std of a float32 LDS array from its double sums: sqrt(E[x^2] - E[x]^2) evaluated in double
// (NumPy works in float32; the difference is ~1e-7 relative, inside the 1e-5 bar on h).
__device__ __forceinline__ float std_from_sums(double sx, double sxx, int n) {
    const double mean = sx / (double)n;
    double var = sxx / (double)n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    return sqrtf((float)var);
}

__device__ __forceinline__ float grad1(const float *f, int stride, int k, int n) {
    if (k == 0) return f[stride] - f[0];
    if (k == n - 1) return f[(n - 1) * stride] - f[(n - 2) * stride];
    return (f[(k + 1) * stride] - f[(k - 1) * stride]) * 0.5f;
}
__device__ __forceinline__ float grad2(const float *f, int stride, int k, int n) {
    if (k == 0) return grad1(f, stride, 1, n) - grad1(f, stride, 0, n);
    if (k == n - 1) return grad1(f, stride, n - 1, n) - grad1(f, stride, n - 2, n);
    return (grad1(f, stride, k + 1, n) - grad1(f, stride, k - 1, n)) * 0.5f;
}

// One rotated-template sample (reference pmlib.py:105-113; scipy order-0 arithmetic), read from the
// LDS copy of the image-1 neighbourhood.  Returns the uint8 pixel (0 outside the image).
struct Patch { const uint8_t *p; int r0, c0; int pitch; };

__device__ __forceinline__ uint8_t sample_template(const PMArgs &A, const Patch &P, const double *rot4,
                                                   double c1, double r1, int i, int j)
{
    const double cosa = rot4[0], sina = rot4[1];
    const double off0 = r1 - rot4[2], off1 = c1 - rot4[3];
    double rr = 0.0 + (double)i * cosa;
    rr = rr + (double)j * sina;
    rr = rr + off0;
    double cc = 0.0 + (double)i * (-sina);
    cc = cc + (double)j * cosa;
    cc = cc + off1;
    uint8_t v = 0;
    if (rr >= 0.0 && rr <= (double)(A.rows1 - 1) && cc >= 0.0 && cc <= (double)(A.cols1 - 1)) {
        const int ri = (int)floor(rr + 0.5), ci = (int)floor(cc + 0.5);
        v = P.p[(ri - P.r0) * P.pitch + (ci - P.c0)];
    }
    return v;
}

// Window fragment for one MFMA: lane (n = l&15, g = l>>4) gets bytes W'[row][x0+n+16g .. +15].
// `p` points at the lane's dword-aligned start, `sh` = byte misalignment (0..3).
__device__ __forceinline__ v4i load_bfrag(const uint8_t *p, u32 sh)
{
    const u32 *q = reinterpret_cast<const u32 *>(p);
    const u32 r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3], r4 = q[4];
    v4i b;
    b[0] = (int)__builtin_amdgcn_alignbyte(r1, r0, sh);
    b[1] = (int)__builtin_amdgcn_alignbyte(r2, r1, sh);
    b[2] = (int)__builtin_amdgcn_alignbyte(r3, r2, sh);
    b[3] = (int)__builtin_amdgcn_alignbyte(r4, r3, sh);
    return b;
}

// One sweep work item: 8 output rows x 16 placements x 16 template slots.
// Window row rho = y0 + step feeds output row y0 + t through template row i = step - t, whose
// fragment sits in ring slot (i & 7).  S > 0: template side known at compile time - the whole
// schedule is static (no branch, exactly 8*S MFMAs).  S == 0: runtime side; the ring holds zero
// fragments for i outside [0, s) so the 8 MFMAs of a step are unconditional.
template <int S>
__device__ __forceinline__ void sweep_item(v4i (&acc)[kBand], const uint8_t *abase, int arow, bool zero_a,
                                           const uint8_t *bbase, int wpitch, int y0, int wh, int s, u32 sh)
{
    // lanes of k-group 3 (columns 48..63) hold an all-zero template fragment unless s = 49
    auto load_a = [&](int i) {
        v4i a = *reinterpret_cast<const v4i *>(abase + i * arow);
        if (zero_a) a = v4i{0, 0, 0, 0};
        return a;
    };
#pragma unroll
    for (int t = 0; t < kBand; ++t) acc[t] = v4i{0, 0, 0, 0};
    v4i ring[kBand];
#pragma unroll
    for (int t = 0; t < kBand; ++t) ring[t] = v4i{0, 0, 0, 0};
    if (S > 0) {
        constexpr int NS = kBand + (S > 0 ? S : 1) - 1;
        // Software pipeline of depth 2, interleaved with the MFMAs of the current step (a single
        // wavefront issues in order: anything not placed between two MFMAs adds to the step time):
        //   step k:  8 MFMAs(k)  ||  finish operands of k+1 (alignbyte x4, zeroing x4)  ||  issue LDS reads of k+2
        struct Raw { u32 r[5]; };
        auto issue_b = [&](int step) {
            int row = y0 + step;
            row = row < wh - 1 ? row : wh - 1;                    // last band: rows past the window are unused
            const u32 *q = reinterpret_cast<const u32 *>(bbase + row * wpitch);
            Raw w;
            w.r[0] = q[0]; w.r[1] = q[1]; w.r[2] = q[2]; w.r[3] = q[3]; w.r[4] = q[4];
            return w;
        };
        auto finish_b = [&](const Raw &w) {
            v4i b;
            b[0] = (int)__builtin_amdgcn_alignbyte(w.r[1], w.r[0], sh);
            b[1] = (int)__builtin_amdgcn_alignbyte(w.r[2], w.r[1], sh);
            b[2] = (int)__builtin_amdgcn_alignbyte(w.r[3], w.r[2], sh);
            b[3] = (int)__builtin_amdgcn_alignbyte(w.r[4], w.r[3], sh);
            return b;
        };
        auto finish_a = [&](v4i a) { if (zero_a) a = v4i{0, 0, 0, 0}; return a; };
        // prologue: operands of step 0 ready, raw reads of step 1 in flight
        ring[0] = finish_a(*reinterpret_cast<const v4i *>(abase));
        v4i b_cur = finish_b(issue_b(0));
        v4i a_raw = *reinterpret_cast<const v4i *>(abase + (1 < S ? 1 : 0) * arow);
        Raw b_raw = issue_b(1 < NS ? 1 : 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int step = 0; step < NS; ++step) {
            // operands of step+1 (their reads were issued during step-1)
            v4i a_fin = finish_a(a_raw);
            v4i b_nxt = finish_b(b_raw);
            // reads of step+2
            if (step + 2 < NS) {
                if (step + 2 < S) a_raw = *reinterpret_cast<const v4i *>(abase + (step + 2) * arow);
                b_raw = issue_b(step + 2);
            }
#pragma unroll
            for (int t = 0; t < kBand; ++t) {
                const int i = step - t;
                if (i >= 0 && i < S)
                    acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[i & (kBand - 1)], b_cur, acc[t], 0, 0, 0);
            }
            // interleave: one MFMA, then up to three of the other instructions
#pragma unroll
            for (int g = 0; g < kBand; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x106, 3, 0);   // VALU | SALU | DS read
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
                const int ia = step < s ? step : s;               // row s of the fragment table is all zero
                ring[u] = load_a(ia);
                int row = y0 + step;
                row = row < wh - 1 ? row : wh - 1;
                const v4i b = load_bfrag(bbase + row * wpitch, sh);
#pragma unroll
                for (int t = 0; t < kBand; ++t)
                    acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[(u - t) & (kBand - 1)], b, acc[t], 0, 0, 0);
            }
        }
    }
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

struct Score { float lmax, bestv; int bestkey; };

// Cold path: the candidate queue is full (only with massive exact ties).  Out of line so that its
// double-precision sqrt/divide sequence does not inflate the register allocation of the sweep.
__device__ __noinline__ Score overflow_eval(Score sc, int p, int swp, u32 siiv, int key, int s, double sT, double rTd, int cT)
{
    const double nd = (double)(s * s);
    const double swd = (double)swp, siid = (double)siiv;
    const double dI = nd * siid - swd * swd;
    const double s2 = siid + 256.0 * swd + 16384.0 * nd;
    const bool lowvar = (2.0 * dI <= nd) && (dI * 8388608.0 <= 10.0 * nd * s2);
    const double numer = nd * (double)p - swd * sT;
    const float rv = exact_ncc(numer, dI, rTd, cT != 0, lowvar);
    if (rv > sc.bestv || (rv == sc.bestv && key < sc.bestkey)) { sc.bestv = rv; sc.bestkey = key; }
    return sc;
}

// P2 of the kernel for one group of <= 15 angles: every wavefront takes sweep work items (8 output
// rows x 16 placements), runs the MFMA schedule and scores the 8 x 16 x 15 results.
template <int S>
__device__ __forceinline__ Score sweep_group(Score sc, int afrag_off, int win_off, int sii_off, int queue_off,
                                          int arow, int wpitch, int wh, int rh, int rw, int s, int Kg, int a0,
                                          long long *dbg_cycles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    MiscM *m = reinterpret_cast<MiscM *>(smem);
    const uint8_t *afrag = smem + afrag_off;
    const uint8_t *win = smem + win_off;
    const u32 *sii = reinterpret_cast<const u32 *>(smem + sii_off);
    uint4 *queue = reinterpret_cast<uint4 *>(smem + queue_off);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n_l = lane & 15, q_l = lane >> 4;
    const bool zero_a = (s <= 48) && q_l == 3;
    const int alane = zero_a ? lane - 16 : lane;
    const double nd = (double)(s * s);
    float lmax = sc.lmax, bestv = sc.bestv;
    int bestkey = sc.bestkey;
#define STAMP2(k) do { if (dbg_cycles && tid == 0) dbg_cycles[k] = (long long)clock64(); } while (0)
        // this lane's four template slots (rows 4 q + r of the MFMA result)
        double sT4[4];
        float rT4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = 4 * q_l + r;
            sT4[r] = a < Kg ? m->sTd[a0 + a] : 0.0;
            rT4[r] = a < Kg ? m->rTf[a0 + a] : 0.0f;                // unused slots: estimate 0 * x, masked below
        }

        // ---- P2: angle-major sweep; wave wv takes work items wv, wv+4, ... ----
        const int nbands = (rh + kBand - 1) / kBand, ntx = (rw + 15) / 16;
        for (int item = wv; item < nbands * ntx; item += kWavesM) {
            const int band = item / ntx, xt = item - band * ntx;
            const int y0 = band * kBand, x0 = xt * 16;
            const u32 sbyte = (u32)(x0 + n_l + 16 * q_l);
            const u32 sh = sbyte & 3u;
            const uint8_t *bbase = win + (sbyte & ~3u);

            if (item == wv) STAMP2(10);
            v4i acc[kBand];
            sweep_item<S>(acc, afrag + alane * 16, arow, zero_a, bbase, wpitch, y0, wh, s, sh);
            if (item == wv) STAMP2(11);

            // ---- epilogue.  Pass A (always, branch-free): float32 estimate of all 8 x 4 NCC values of
            //      this lane and their maximum.  Pass B (only if the item can hold an arg-max candidate:
            //      its maximum is within kMargin of the running maximum, or an estimate is NaN = degenerate
            //      window/template): estimates within kMargin of the running maximum are queued for exact
            //      evaluation.  |estimate - value| <= 4e-7, so the true arg-max is always queued. ----
            {
                const float gm = key2f(m->gmax_key);
                lmax = gm > lmax ? gm : lmax;
            }
            const int x = x0 + n_l;
            const bool xok = x < rw;
            // operands of all 8 rows first (one LDS round trip), then straight-line arithmetic
            int swp[kBand];
            u32 siiv[kBand];
#pragma unroll
            for (int t = 0; t < kBand; ++t) {
                swp[t] = __builtin_amdgcn_ds_bpermute((n_l + 48) << 2, acc[t][3]);   // slot 15 = sum of w'
                const int y = y0 + t;
                const int pos = (xok & (y < rh)) ? y * rw + x : 0;
                siiv[t] = sii[pos];
            }
            const u32 livebits = (4 * q_l + 0 < Kg ? 1u : 0u) | (4 * q_l + 1 < Kg ? 2u : 0u) |
                                 (4 * q_l + 2 < Kg ? 4u : 0u) | (4 * q_l + 3 < Kg ? 8u : 0u);
            auto estimate_row = [&](int t, float (&e)[4]) {
                const double swd = (double)swp[t], siid = (double)siiv[t];
                const double dI = nd * siid - swd * swd;                  // exact
                float rIf = __builtin_amdgcn_rsqf((float)dI);
                rIf = (2.0 * dI <= nd) ? NAN : rIf;                       // (nearly) flat window: the exact path decides
                const u32 lv = (xok & (y0 + t < rh)) ? livebits : 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double numer = nd * (double)acc[t][r] - swd * sT4[r];   // exact
                    const float v = ((float)numer * rIf) * rT4[r];
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
            const bool anynan = nanbits != 0;
            const float wavemax = wave_max_dpp(itemmax);
            const bool need_b = (wavemax >= lmax - kMargin) || (__ballot(anynan) != 0ull);   // wavefront-uniform
            lmax = fmaxf(lmax, wavemax);
            if (need_b) {
                const float thr = lmax - kMargin;
#pragma unroll
                for (int t = 0; t < kBand; ++t) {
                    float e[4];
                    estimate_row(t, e);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (e[r] != -INFINITY && !(e[r] < thr)) {         // live, and true for NaN
                            const int a = 4 * q_l + r, y = y0 + t;
                            const int key = ((a0 + a) * rh + y) * rw + x;
                            const u32 slot = atomicAdd(&m->qcount, 1u);
                            if (slot < (u32)kQueueCap) {
                                queue[slot] = make_uint4((u32)acc[t][r], (u32)swp[t], siiv[t], (u32)key);
                            } else {                                      // queue full (massive ties): cold path
                                const Score o = overflow_eval(Score{lmax, bestv, bestkey}, acc[t][r], swp[t], siiv[t], key, s,
                                                              sT4[r], m->rTd[a0 + a], m->constT[a0 + a]);
                                bestv = o.bestv; bestkey = o.bestkey;
                            }
                        }
                    }
                }
            }
            if (lane == 0) atomicMax(&m->gmax_key, f2key(lmax));
            if (item == wv) STAMP2(12);
        }
    return Score{lmax, bestv, bestkey};
#undef STAMP2
}

template <int S>
__global__ __launch_bounds__(kBlockM, 3) void pm_kernel_mfma(const PMArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    MiscM *m = reinterpret_cast<MiscM *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
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
    const int rh = wh - s + 1, rw = ww - s + 1;
    const int npos = rh * rw;
    const MfmaLdsLayout L = mfma_lds_layout(wh, ww, s);
    uint8_t *win = smem + L.win_off;
    u32 *sii = reinterpret_cast<u32 *>(smem + L.sii_off);
    float *hes = reinterpret_cast<float *>(smem + L.sii_off);             // aliases sii (dead by then)
    u32 *colsum = reinterpret_cast<u32 *>(smem + L.u_off);                // U region, stage 1 of S_II
    uint8_t *afrag = smem + L.u_off;                                      // U region, sweep operands
    uint8_t *patch = smem + L.patch_off;
    uint4 *queue = reinterpret_cast<uint4 *>(smem + L.queue_off);
    uint8_t *trow = smem + L.u_off;                                       // U region, winner operands
    uint8_t *trow1 = trow + L.trow_bytes;
    float *ccm = reinterpret_cast<float *>(smem + L.u_off + 2 * L.trow_bytes);
    const int wpitch = L.wpitch, arow = L.arow;
    const double nd = (double)(s * s);
    const double c1 = A.c1[pt], r1 = A.r1[pt];
    const int n_l = lane & 15, q_l = lane >> 4;

    if (tid == 0) { m->zero_flag = 0; m->gmax_key = 0x007fffffu /* f2key(-inf) */; m->qcount = 0; }

    // ---- P0a: search window -> LDS, re-centred to int8 (w ^ 0x80), zero beyond the window.
    //      Loads are issued four dwords deep before any is consumed (HBM/L2 latency overlaps). ----
    {
        const int dw_per_row = wpitch / 4, ndw = wh * dw_per_row;
        const uintptr_t last_dw = (reinterpret_cast<uintptr_t>(A.img2 + (A.rows2 - 1) * A.stride2 + A.cols2) - 1) & ~(uintptr_t)3;
        for (int base = 0; base < ndw; base += 4 * kBlockM) {
            u32 lo[4], hi[4], shv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * kBlockM + tid;
                const int idc = idx < ndw ? idx : 0;
                const int row = idc / dw_per_row, dq = idc - row * dw_per_row;
                const int dqc = 4 * dq < ww ? dq : 0;               // clamp so that the loads are always legal
                const uint8_t *gp = A.img2 + (r0 + row) * A.stride2 + c0 + 4 * dqc;
                const uintptr_t ga = reinterpret_cast<uintptr_t>(gp) & ~(uintptr_t)3;
                const uintptr_t gb = ga + 4 <= last_dw ? ga + 4 : last_dw;
                shv[u] = (u32)(reinterpret_cast<uintptr_t>(gp) & 3);
                lo[u] = *reinterpret_cast<const u32 *>(ga);
                hi[u] = *reinterpret_cast<const u32 *>(gb);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * kBlockM + tid;
                if (idx < ndw) {
                    const int row = idx / dw_per_row, dq = idx - row * dw_per_row;
                    u32 v = __builtin_amdgcn_alignbyte(hi[u], lo[u], shv[u]) ^ 0x80808080u;
                    const int nvalid = ww - 4 * dq;
                    if (nvalid < 4) v = nvalid > 0 ? (v & ((1u << (8 * nvalid)) - 1u)) : 0u;
                    reinterpret_cast<u32 *>(win + row * wpitch)[dq] = v;
                }
            }
        }
    }
    __syncthreads();
    SID_STAMP(1);

    // ---- P1: S_II' = box sums of w'^2.  Stage 1: running sums down each window column ----
    for (int x = tid; x < ww; x += kBlockM) {
        const int8_t *col = reinterpret_cast<const int8_t *>(win) + x;
        int c = 0;
#pragma unroll 8
        for (int i = 0; i < s; ++i) { const int v = col[i * wpitch]; c += v * v; }
        colsum[x] = (u32)c;
#pragma unroll 8
        for (int y = 1; y < rh; ++y) {
            const int vo = col[(y - 1) * wpitch], vn = col[(y + s - 1) * wpitch];
            c += vn * vn - vo * vo;
            colsum[y * ww + x] = (u32)c;
        }
    }
    __syncthreads();
    // Stage 2: running sums along each output row
    for (int y = tid; y < rh; y += kBlockM) {
        const u32 *cr = colsum + y * ww;
        u32 acc = 0;
#pragma unroll 8
        for (int j = 0; j < s; ++j) acc += cr[j];
        sii[y * rw] = acc;
#pragma unroll 8
        for (int x = 1; x < rw; ++x) {
            acc += cr[x + s - 1] - cr[x - 1];
            sii[y * rw + x] = acc;
        }
    }
    __syncthreads();
    SID_STAMP(2);

    // ---- P0b: neighbourhood of the template centre on image 1 -> LDS patch ----
    Patch P;
    P.p = patch; P.pitch = L.ppitch;
    P.r0 = (int)floor(r1) - L.pradius;
    P.c0 = (int)floor(c1) - L.pradius;
    {
        const int np = L.pdim * L.pdim;
        for (int base = 0; base < np; base += 8 * kBlockM) {
            uint8_t pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                           // 8 independent byte loads in flight
                const int idx = base + u * kBlockM + tid;
                const int idc = idx < np ? idx : 0;
                const int pr = idc / L.pdim, pc = idc - pr * L.pdim;
                int64_t gr = (int64_t)P.r0 + pr, gc = (int64_t)P.c0 + pc;
                gr = gr < 0 ? 0 : (gr > A.rows1 - 1 ? A.rows1 - 1 : gr);     // clamped rows/cols are never sampled
                gc = gc < 0 ? 0 : (gc > A.cols1 - 1 ? A.cols1 - 1 : gc);
                pv[u] = A.img1[gr * A.stride1 + gc];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * kBlockM + tid;
                if (idx < np) { const int pr = idx / L.pdim, pc = idx - pr * L.pdim; patch[pr * L.ppitch + pc] = pv[u]; }
            }
        }
    }

    // ---- per group of <= 15 angles: template operands, sweep, exact evaluation of candidates ----
    float bestv = -INFINITY;         // exact best of this thread
    int bestkey = 0x7fffffff;
    float lmax = -INFINITY;          // running max of the float32 estimates (wavefront-uniform)
    const bool zero_a = (s <= 48) && q_l == 3;
    const int alane = zero_a ? lane - 16 : lane;                   // any legal address for the zeroed lanes
    const double rmax1 = (double)(A.rows1 - 1), cmax1 = (double)(A.cols1 - 1);

    for (int a0 = 0; a0 < K; a0 += kAnglesPerGroup) {
        const int Kg = (K - a0) < kAnglesPerGroup ? (K - a0) : kAnglesPerGroup;
        // afrag[i][lane = g*16 + slot][16 bytes], byte jj <-> column c = 16 g + jj; row s is all zero
        for (int idx = tid; idx < (s + 1) * arow / 4; idx += kBlockM) reinterpret_cast<u32 *>(afrag)[idx] = 0;
        if (tid < kSlots) { m->isT[tid] = 0; m->isTT[tid] = 0; }
        __syncthreads();                                           // also: patch complete
        SID_STAMP(8);
        {
            int sawzero = 0;
            const int nss = s * s;
            for (int a = 0; a < Kg; ++a) {
                const double *rot4 = A.rot + 4 * (a0 + a);
                const double cosa = rot4[0], sina = rot4[1], msin = -sina;
                const double off0 = r1 - rot4[2], off1 = c1 - rot4[3];
                int st = 0, stt = 0;
                for (int base = 0; base < nss; base += 4 * kBlockM) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {                  // independent chains: latencies overlap
                        const int idx = base + u * kBlockM + tid;
                        const bool valid = idx < nss;
                        const int idc = valid ? idx : 0;
                        const int i = idc / s, j = idc - i * s;
                        // scipy NI_GeometricTransform order of operations (matrix = transform.T)
                        double rr = 0.0 + (double)i * cosa;
                        rr = rr + (double)j * sina;
                        rr = rr + off0;
                        double cc = 0.0 + (double)i * msin;
                        cc = cc + (double)j * cosa;
                        cc = cc + off1;
                        const bool in = rr >= 0.0 && rr <= rmax1 && cc >= 0.0 && cc <= cmax1;
                        int ri = (int)floor(rr + 0.5) - P.r0, ci = (int)floor(cc + 0.5) - P.c0;
                        ri = in ? ri : 0; ci = in ? ci : 0;       // in-image samples always lie inside the patch
                        int v = patch[ri * P.pitch + ci];
                        v = in ? v : 0;
                        if (valid) {
                            if (v == 0) sawzero = 1;
                            afrag[i * arow + ((j >> 4) * 16 + a) * 16 + (j & 15)] = (uint8_t)(v ^ 0x80);
                            const int sv = v - 128;
                            st += sv; stt += sv * sv;
                        }
                    }
                }
                st = wave_sum_dpp(st); stt = wave_sum_dpp(stt);
                if (lane == 0) { atomicAdd(&m->isT[a], st); atomicAdd(&m->isTT[a], stt); }
            }
            for (int idx = tid; idx < nss; idx += kBlockM) {        // slot 15: all-ones template
                const int i = idx / s, j = idx - i * s;
                afrag[i * arow + ((j >> 4) * 16 + 15) * 16 + (j & 15)] = 1;
            }
            if (sawzero) m->zero_flag = 1;
        }
        __syncthreads();
        SID_STAMP(9);
        if (A.dbg_templates) {
            for (int idx = tid; idx < Kg * s * s; idx += kBlockM) {
                const int a = idx / (s * s), rem = idx - a * s * s;
                const int i = rem / s, j = rem - i * s;
                A.dbg_templates[(a0 + a) * s * s + rem] = afrag[i * arow + ((j >> 4) * 16 + a) * 16 + (j & 15)] ^ 0x80;
            }
        }
        if (m->zero_flag) {                                        // pmlib.py:152-154
            if (tid < 5) out[tid] = NAN;
            if (oij && tid < 3) oij[tid] = -1;
            if (A.dbg_shape && tid == 0) { A.dbg_shape[0] = rh; A.dbg_shape[1] = rw; }
            return;
        }
        if (tid < Kg) {                                            // per-angle terms (signed domain)
            const double st = (double)m->isT[tid], stt = (double)m->isTT[tid];
            const double dT = nd * stt - st * st;                  // exact
            const double rT = 1.0 / sqrt(dT);
            m->sTd[a0 + tid] = st;
            m->constT[a0 + tid] = dT == 0.0 ? 1 : 0;
            m->rTd[a0 + tid] = rT;
            m->rTf[a0 + tid] = dT == 0.0 ? NAN : (float)rT;       // NaN estimate => always a candidate
        }
        __syncthreads();
        SID_STAMP(3);

        // ---- P2: angle-major sweep (out of line: keeps the hot loop's register allocation apart) ----
        {
            const Score sc = sweep_group<S>(Score{lmax, bestv, bestkey}, L.u_off, L.win_off, L.sii_off, L.queue_off,
                                            arow, wpitch, wh, rh, rw, s, Kg, a0, A.dbg_cycles);
            lmax = sc.lmax; bestv = sc.bestv; bestkey = sc.bestkey;
        }
        __syncthreads();
        SID_STAMP(4);
        // ---- exact (double) evaluation of the queued candidates of this group ----
        {
            const int nq = (int)(m->qcount < (u32)kQueueCap ? m->qcount : (u32)kQueueCap);
            for (int idx = tid; idx < nq; idx += kBlockM) {
                const uint4 rec = queue[idx];
                const int key = (int)rec.w;
                const int a = key / npos;
                const double swd = (double)(int)rec.y, siid = (double)rec.z;
                const double dI = nd * siid - swd * swd;
                const double s2 = siid + 256.0 * swd + 16384.0 * nd;
                const bool lowvar = (2.0 * dI <= nd) && (dI * 8388608.0 <= 10.0 * nd * s2);
                const double numer = nd * (double)(int)rec.x - swd * m->sTd[a];
                const float rv = exact_ncc(numer, dI, m->rTd[a], m->constT[a] != 0, lowvar);
                if (rv > bestv || (rv == bestv && key < bestkey)) { bestv = rv; bestkey = key; }
            }
        }
        __syncthreads();                                           // afrag / queue are rewritten by the next group
        if (tid == 0) m->qcount = 0;
    }

    // ---- P3: block arg-max (first angle, then first row-major index on ties) ----
    {
        const float wbest = wave_max_dpp(bestv);                   // exact values: no NaN (exact_ncc never returns one)
        bestkey = wave_min_dpp(bestv == wbest ? bestkey : 0x7fffffff);
        bestv = wbest;
    }
    if (lane == 0) { m->red_f[wv] = bestv; m->red_i[wv] = bestkey; }
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

    // ---- P4: winner's NCC matrix, row-major MFMA.  trow[g][16 + i][16 B], zero rows around ----
    const int trows = s + 32;
    for (int idx = tid; idx < 2 * L.trow_bytes / 4; idx += kBlockM) reinterpret_cast<u32 *>(trow)[idx] = 0;
    __syncthreads();
    for (int idx = tid; idx < s * s; idx += kBlockM) {
        const int i = idx / s, j = idx - i * s;
        const uint8_t v = sample_template(A, P, A.rot + 4 * ka, c1, r1, i, j);
        const int off = ((j >> 4) * trows + 16 + i) * 16 + (j & 15);
        trow[off] = v ^ 0x80;
        trow1[off] = 1;
    }
    __syncthreads();
    SID_STAMP(13);
    {
        const double sT = m->sTd[ka], rT = m->rTd[ka];
        const bool cT = m->constT[ka] != 0;
        const int nty = (rh + 15) / 16, ntx = (rw + 15) / 16;
        for (int item = wv; item < nty * ntx; item += kWavesM) {
            const int yt = item / ntx, xt = item - yt * ntx;
            const int y0 = yt * 16, x0 = xt * 16;
            const int rows_here = (rh - y0) < 16 ? (rh - y0) : 16;
            const int nsteps = rows_here + s - 1;
            const u32 sbyte = (u32)(x0 + n_l + 16 * q_l);
            const u32 sh = sbyte & 3u;
            const uint8_t *bbase = win + (sbyte & ~3u);
            // lane (m = n_l, g = q_l): template row i = step - m  ->  trow[g][16 + step - m]
            const uint8_t *ta = trow + (q_l * trows + 16 - n_l) * 16;
            const uint8_t *ta1 = trow1 + (q_l * trows + 16 - n_l) * 16;
            v4i accT = {0, 0, 0, 0}, accS = {0, 0, 0, 0};
            // depth-2 software pipeline: the LDS reads of step+2 are in flight while step's MFMAs issue
            struct Ops { u32 r[5]; v4i at, a1; };
            auto issue = [&](int step) {
                const int st = step < nsteps ? step : nsteps - 1;        // clamped: harmless re-read past the end
                const u32 *q = reinterpret_cast<const u32 *>(bbase + (y0 + st) * wpitch);
                Ops o;
                o.r[0] = q[0]; o.r[1] = q[1]; o.r[2] = q[2]; o.r[3] = q[3]; o.r[4] = q[4];
                o.at = *reinterpret_cast<const v4i *>(ta + st * 16);
                o.a1 = *reinterpret_cast<const v4i *>(ta1 + st * 16);
                return o;
            };
            Ops o0 = issue(0), o1 = issue(1);
            for (int step = 0; step < nsteps; step += 2) {
                const Ops o2 = issue(step + 2), o3 = issue(step + 3);
                {
                    v4i bb;
                    bb[0] = (int)__builtin_amdgcn_alignbyte(o0.r[1], o0.r[0], sh); bb[1] = (int)__builtin_amdgcn_alignbyte(o0.r[2], o0.r[1], sh);
                    bb[2] = (int)__builtin_amdgcn_alignbyte(o0.r[3], o0.r[2], sh); bb[3] = (int)__builtin_amdgcn_alignbyte(o0.r[4], o0.r[3], sh);
                    accT = __builtin_amdgcn_mfma_i32_16x16x64_i8(o0.at, bb, accT, 0, 0, 0);
                    accS = __builtin_amdgcn_mfma_i32_16x16x64_i8(o0.a1, bb, accS, 0, 0, 0);
                }
                if (step + 1 < nsteps) {
                    v4i bb;
                    bb[0] = (int)__builtin_amdgcn_alignbyte(o1.r[1], o1.r[0], sh); bb[1] = (int)__builtin_amdgcn_alignbyte(o1.r[2], o1.r[1], sh);
                    bb[2] = (int)__builtin_amdgcn_alignbyte(o1.r[3], o1.r[2], sh); bb[3] = (int)__builtin_amdgcn_alignbyte(o1.r[4], o1.r[3], sh);
                    accT = __builtin_amdgcn_mfma_i32_16x16x64_i8(o1.at, bb, accT, 0, 0, 0);
                    accS = __builtin_amdgcn_mfma_i32_16x16x64_i8(o1.a1, bb, accS, 0, 0, 0);
                }
                o0 = o2; o1 = o3;
            }
            const int x = x0 + n_l;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = y0 + 4 * q_l + r;
                if (y < rh && x < rw) {
                    const double swd = (double)accS[r];
                    const double siid = (double)sii[y * rw + x];
                    const double dI = nd * siid - swd * swd;
                    const double s2 = siid + 256.0 * swd + 16384.0 * nd;
                    const bool lowvar = (2.0 * dI <= nd) && (dI * 8388608.0 <= 10.0 * nd * s2);
                    const double numer = nd * (double)accT[r] - swd * sT;
                    ccm[y * rw + x] = exact_ncc(numer, dI, rT, cT, lowvar);
                }
            }
        }
    }
    __syncthreads();
    SID_STAMP(6);

    // ---- P5: Hessian at the peak (pmlib.py:36-59, :167); hes aliases sii ----
    double sx = 0.0, sxx = 0.0;
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
#pragma unroll 2
        for (int idx = tid; idx < npos; idx += kBlockM) {
            const int y = idx / rw, x = idx - y * rw;
            const float d2x = d2(ccm + y * rw, 1, x, rw);
            const float d2y = d2(ccm + x, rw, y, rh);
            const double hh = (double)d2x * (double)d2x + (double)d2y * (double)d2y;
            const float hv = (float)sqrt(hh);                      // hypotf: double sqrt, narrowed
            hes[idx] = hv;
            sx += (double)hv; sxx += (double)hv * (double)hv;
        }
    }
    __syncthreads();
    if (A.dbg_ccm || A.dbg_hes) {
        for (int idx = tid; idx < npos && idx < A.dbg_cap; idx += kBlockM) {
            if (A.dbg_ccm) A.dbg_ccm[idx] = ccm[idx];
            if (A.dbg_hes) A.dbg_hes[idx] = hes[idx];
        }
    }
    if (A.dbg_shape && tid == 0) { A.dbg_shape[0] = rh; A.dbg_shape[1] = rw; }
    SID_STAMP(14);
    float h = hes[iy * rw + ix];
    if (A.flags & 1u) {
        sx = block_sum(sx, m);
        sxx = block_sum(sxx, m);
        const float sd = std_from_sums(sx, sxx, npos);
        const float med = block_median(hes, npos, m, reinterpret_cast<u32 *>(trow));   // winner operands are dead: 4 KB of histograms
        h = (h - med) / sd;
    }
    float rr = best_r;
    if (A.flags & 4u) {
        double cx = 0.0, cxx = 0.0;
        for (int idx = tid; idx < npos; idx += kBlockM) { const double v = (double)ccm[idx]; cx += v; cxx += v * v; }
        cx = block_sum(cx, m);
        cxx = block_sum(cxx, m);
        const float sd = std_from_sums(cx, cxx, npos);
        const float med = block_median(ccm, npos, m, reinterpret_cast<u32 *>(trow));
        rr = (best_r - med) / sd;
    }
    SID_STAMP(7);
    if (tid == 0) {
        out[0] = c2fg + ((double)ix - (double)(ww - s) / 2.0);
        out[1] = r2fg + ((double)iy - (double)(wh - s) / 2.0);
        out[2] = A.angles[ka];
        out[3] = (double)rr;
        out[4] = (double)h;
        if (oij) { oij[0] = iy; oij[1] = ix; oij[2] = ka; }
    }
}

}  // namespace

bool mfma_img_size_supported(int s) { return s >= 2 && s + 15 <= 64; }

int launch_pm_mfma(const PMArgs &args, int lds_bytes, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (args.n_launch <= 0) return (int)hipSuccess;
    void (*kern)(const PMArgs) = pm_kernel_mfma<0>;
    if (args.img_size == 34) kern = pm_kernel_mfma<34>;
    else if (args.img_size == 35) kern = pm_kernel_mfma<35>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(args.n_launch), dim3(kBlockM), lds_bytes, st, args);
    return (int)hipGetLastError();
}

}  // namespace sid
