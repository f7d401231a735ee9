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
// One workgroup (256 threads = 4 wavefronts) per grid point, as in pm_kernel.hip.
// Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <math.h>
#include "pm_kernel.h"

namespace sid {

namespace {

typedef uint32_t u32;
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kBand = 8;             // output rows per sweep work item
constexpr int kSlots = 16;           // MFMA M: 15 angles + ones
constexpr int kAnglesPerGroup = 15;
constexpr float kMargin = 1e-5f;     // pre-filter margin (see header)

struct MiscM {                       // LDS offset 0, kMiscMfmaBytes reserved
    u32 hist[256];
    double red_d[8];
    float red_f[8];
    int red_i[8];
    u32 sel_bin, sel_less;
    int zero_flag;
    int best_key; float best_val;
    int pad_[3];
    double rTd[kSlots];              // 1/sqrt(dT) per slot of the current group
    double sTd[kSlots];              // sum t' per slot
    float rTf[kSlots];
    int constT[kSlots];
};
static_assert(sizeof(MiscM) <= kMiscMfmaBytes, "misc header too large");

// ---- small block utilities (same semantics as in pm_kernel.hip) ----
__device__ __forceinline__ double wave_sum_d(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ int wave_sum_i(int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

__device__ __forceinline__ double block_sum(double v, MiscM *m) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) m->red_d[threadIdx.x >> 6] = v;
    __syncthreads();
    return (m->red_d[0] + m->red_d[1]) + (m->red_d[2] + m->red_d[3]);
}
__device__ __forceinline__ u32 block_min(u32 v, MiscM *m) {
    for (int o = 32; o > 0; o >>= 1) { u32 t = __shfl_xor(v, o); v = t < v ? t : v; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) m->red_i[threadIdx.x >> 6] = (int)v;
    __syncthreads();
    u32 a = (u32)m->red_i[0], b = (u32)m->red_i[1], c = (u32)m->red_i[2], d = (u32)m->red_i[3];
    a = a < b ? a : b; c = c < d ? c : d;
    return a < c ? a : c;
}
__device__ __forceinline__ u32 f2key(float f) { u32 b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float key2f(u32 k) { u32 b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; return __uint_as_float(b); }

__device__ u32 block_select(const float *v, int n, u32 k, MiscM *m, u32 *count_le) {
    u32 prefix = 0, mask = 0, less_total = 0, kk = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        __syncthreads();
        m->hist[threadIdx.x] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += kBlock) {
            const u32 key = f2key(v[i]);
            if ((key & mask) == prefix) atomicAdd(&m->hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        const u32 c = m->hist[threadIdx.x];
        u32 inc = c;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int o = 1; o < 64; o <<= 1) { u32 t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) m->red_i[w] = (int)inc;
        __syncthreads();
        u32 base = 0;
        for (int j = 0; j < w; ++j) base += (u32)m->red_i[j];
        inc += base;
        const u32 exc = inc - c;
        if (kk >= exc && kk < inc) { m->sel_bin = threadIdx.x; m->sel_less = exc; }
        __syncthreads();
        const u32 bin = m->sel_bin, less = m->sel_less;
        prefix |= bin << shift;
        mask |= 255u << shift;
        kk -= less;
        less_total += less;
        if (shift == 0) *count_le = less_total + m->hist[bin];
    }
    __syncthreads();
    return prefix;
}

__device__ void block_median_std(const float *v, int n, MiscM *m, float *med, float *sd) {
    u32 cle;
    float md;
    if (n & 1) {
        md = key2f(block_select(v, n, (u32)(n / 2), m, &cle));
    } else {
        const u32 k1 = (u32)(n / 2 - 1);
        const u32 key1 = block_select(v, n, k1, m, &cle);
        u32 key2 = key1;
        if (cle < k1 + 2) {
            u32 mn = 0xffffffffu;
            for (int i = threadIdx.x; i < n; i += kBlock) { const u32 key = f2key(v[i]); if (key > key1 && key < mn) mn = key; }
            key2 = block_min(mn, m);
        }
        md = (key2f(key1) + key2f(key2)) / 2.0f;
    }
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) s += (double)v[i];
    s = block_sum(s, m);
    const float mean = (float)(s / (double)n);
    double q = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) { const float x = v[i] - mean; q += (double)(x * x); }
    q = block_sum(q, m);
    *med = md;
    *sd = sqrtf((float)(q / (double)n));
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

// One rotated-template sample (reference pmlib.py:105-113; scipy order-0 arithmetic). Returns the
// uint8 pixel (0 outside the image).
__device__ __forceinline__ uint8_t sample_template(const PMArgs &A, const double *rot4, double c1, double r1,
                                                   int i, int j)
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
        const int64_t ri = (int64_t)floor(rr + 0.5), ci = (int64_t)floor(cc + 0.5);
        v = A.img1[ri * A.stride1 + ci];
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
__device__ __forceinline__ void sweep_item(v4i (&acc)[kBand], const uint8_t *abase, const uint8_t *bbase,
                                           int wpitch, int y0, int wh, int s, u32 sh)
{
#pragma unroll
    for (int t = 0; t < kBand; ++t) acc[t] = v4i{0, 0, 0, 0};
    v4i ring[kBand];
#pragma unroll
    for (int t = 0; t < kBand; ++t) ring[t] = v4i{0, 0, 0, 0};
    if (S > 0) {
        constexpr int NS = kBand + (S > 0 ? S : 1) - 1;
        // software pipeline of depth 1: operands of step+1 are requested before the MFMAs of step;
        // the scheduling barrier keeps the compiler from hoisting every LDS read to the top.
        v4i a_nxt = *reinterpret_cast<const v4i *>(abase);
        v4i b_nxt = load_bfrag(bbase + (y0 < wh - 1 ? y0 : wh - 1) * wpitch, sh);
#pragma unroll
        for (int step = 0; step < NS; ++step) {
            if (step < S) ring[step & (kBand - 1)] = a_nxt;
            const v4i b = b_nxt;
            if (step + 1 < NS) {
                if (step + 1 < S) a_nxt = *reinterpret_cast<const v4i *>(abase + (step + 1) * 1024);
                int row = y0 + step + 1;
                row = row < wh - 1 ? row : wh - 1;                // last band: rows past the window are unused
                b_nxt = load_bfrag(bbase + row * wpitch, sh);
            }
#pragma unroll
            for (int t = 0; t < kBand; ++t) {
                const int i = step - t;
                if (i >= 0 && i < S)
                    acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ring[i & (kBand - 1)], b, acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const int nsteps = kBand + s - 1;
        for (int cbase = 0; cbase < nsteps; cbase += kBand) {
#pragma unroll
            for (int u = 0; u < kBand; ++u) {
                const int step = cbase + u;
                const int ia = step < s ? step : s;               // row s of the fragment table is all zero
                ring[u] = *reinterpret_cast<const v4i *>(abase + ia * 1024);
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

template <int S>
__global__ __launch_bounds__(kBlock) void pm_kernel_mfma(const PMArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    MiscM *m = reinterpret_cast<MiscM *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pt = A.order[blockIdx.x];
    const int s = S > 0 ? S : A.img_size, K = A.n_angles;

    double *out = A.out + (int64_t)pt * 5;
    int32_t *oij = A.out_ij ? A.out_ij + (int64_t)pt * 3 : nullptr;

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
    uint8_t *trow = smem + L.u_off;                                       // U region, winner operands
    uint8_t *trow1 = trow + L.trow_bytes;
    float *ccm = reinterpret_cast<float *>(smem + L.u_off + 2 * L.trow_bytes);
    const int wpitch = L.wpitch;
    const double nd = (double)(s * s);
    const double c1 = A.c1[pt], r1 = A.r1[pt];

    if (tid == 0) m->zero_flag = 0;

    // ---- P0a: search window -> LDS, re-centred to int8 (w ^ 0x80), zero beyond the window ----
    {
        const int dw_per_row = wpitch / 4;
        const uint8_t *img_end = A.img2 + (A.rows2 - 1) * A.stride2 + A.cols2;
        for (int idx = tid; idx < wh * dw_per_row; idx += kBlock) {
            const int row = idx / dw_per_row, dq = idx - row * dw_per_row;
            u32 v = 0;
            if (4 * dq < ww) {
                const uint8_t *gp = A.img2 + (r0 + row) * A.stride2 + c0 + 4 * dq;
                const uintptr_t ga = reinterpret_cast<uintptr_t>(gp) & ~(uintptr_t)3;
                const u32 sh = (u32)(reinterpret_cast<uintptr_t>(gp) & 3);
                const u32 lo = *reinterpret_cast<const u32 *>(ga);
                u32 hi = 0;
                if (sh != 0 && ga + 4 < reinterpret_cast<uintptr_t>(img_end)) hi = *reinterpret_cast<const u32 *>(ga + 4);
                v = __builtin_amdgcn_alignbyte(hi, lo, sh) ^ 0x80808080u;
                const int nvalid = ww - 4 * dq;
                if (nvalid < 4) v &= (1u << (8 * nvalid)) - 1u;
            }
            reinterpret_cast<u32 *>(win + row * wpitch)[dq] = v;
        }
    }
    __syncthreads();

    // ---- P1: S_II' = box sums of w'^2.  Stage 1: running sums down each window column ----
    for (int x = tid; x < ww; x += kBlock) {
        const int8_t *col = reinterpret_cast<const int8_t *>(win) + x;
        int c = 0;
        for (int i = 0; i < s; ++i) { const int v = col[i * wpitch]; c += v * v; }
        colsum[x] = (u32)c;
#pragma unroll 4
        for (int y = 1; y < rh; ++y) {
            const int vo = col[(y - 1) * wpitch], vn = col[(y + s - 1) * wpitch];
            c += vn * vn - vo * vo;
            colsum[y * ww + x] = (u32)c;
        }
    }
    __syncthreads();
    // Stage 2: running sums along each output row
    for (int y = tid; y < rh; y += kBlock) {
        const u32 *cr = colsum + y * ww;
        u32 acc = 0;
        for (int j = 0; j < s; ++j) acc += cr[j];
        sii[y * rw] = acc;
#pragma unroll 4
        for (int x = 1; x < rw; ++x) {
            acc += cr[x + s - 1] - cr[x - 1];
            sii[y * rw + x] = acc;
        }
    }
    __syncthreads();

    // ---- P0b + P2 per group of <= 15 angles ----
    float bestv = -INFINITY;         // exact best of this lane
    int bestkey = 0x7fffffff;
    float lmax = -INFINITY;          // running max of the float32 estimates seen by this lane
    const int n_l = lane & 15, q_l = lane >> 4;

    for (int a0 = 0; a0 < K; a0 += kAnglesPerGroup) {
        const int Kg = (K - a0) < kAnglesPerGroup ? (K - a0) : kAnglesPerGroup;
        // template fragments: afrag[i][lane = g*16 + slot][16 bytes], byte jj <-> column c = 16 g + jj
        for (int idx = tid; idx < (s + 1) * 256; idx += kBlock) reinterpret_cast<u32 *>(afrag)[idx] = 0;   // + a zero row
        __syncthreads();
        {
            int sawzero = 0;
            for (int idx = tid; idx < Kg * s * s; idx += kBlock) {
                const int a = idx / (s * s), rem = idx - a * s * s;
                const int i = rem / s, j = rem - i * s;
                const uint8_t v = sample_template(A, A.rot + 4 * (a0 + a), c1, r1, i, j);
                if (v == 0) sawzero = 1;
                afrag[i * 1024 + ((j >> 4) * 16 + a) * 16 + (j & 15)] = v ^ 0x80;
            }
            // slot 15: the all-ones template (sum of w' over the box)
            for (int idx = tid; idx < s * s; idx += kBlock) {
                const int i = idx / s, j = idx - i * s;
                afrag[i * 1024 + ((j >> 4) * 16 + 15) * 16 + (j & 15)] = 1;
            }
            if (sawzero) m->zero_flag = 1;
        }
        __syncthreads();
        if (A.dbg_templates) {
            for (int idx = tid; idx < Kg * s * s; idx += kBlock) {
                const int a = idx / (s * s), rem = idx - a * s * s;
                const int i = rem / s, j = rem - i * s;
                A.dbg_templates[(a0 + a) * s * s + rem] = afrag[i * 1024 + ((j >> 4) * 16 + a) * 16 + (j & 15)] ^ 0x80;
            }
        }
        if (m->zero_flag) {                                        // pmlib.py:152-154
            if (tid < 5) out[tid] = NAN;
            if (oij && tid < 3) oij[tid] = -1;
            if (A.dbg_shape && tid == 0) { A.dbg_shape[0] = rh; A.dbg_shape[1] = rw; }
            return;
        }
        // per-slot template sums (signed domain): wave w takes slots w, w+4, ...
        for (int a = wv; a < Kg; a += 4) {
            int st = 0, stt = 0;
            for (int idx = lane; idx < s * s; idx += 64) {
                const int i = idx / s, j = idx - i * s;
                const int v = (int)(int8_t)afrag[i * 1024 + ((j >> 4) * 16 + a) * 16 + (j & 15)];
                st += v; stt += v * v;
            }
            st = wave_sum_i(st); stt = wave_sum_i(stt);
            if (lane == 0) {
                const double dT = nd * (double)stt - (double)st * (double)st;
                m->sTd[a] = (double)st;
                m->constT[a] = dT == 0.0 ? 1 : 0;
                const double rT = 1.0 / sqrt(dT);
                m->rTd[a] = rT;
                m->rTf[a] = (float)rT;
            }
        }
        __syncthreads();

        // ---- P2: angle-major sweep; wave wv takes work items wv, wv+4, ... ----
        const int nbands = (rh + kBand - 1) / kBand, ntx = (rw + 15) / 16;
        for (int item = wv; item < nbands * ntx; item += 4) {
            const int band = item / ntx, xt = item - band * ntx;
            const int y0 = band * kBand, x0 = xt * 16;
            const u32 sbyte = (u32)(x0 + n_l + 16 * q_l);
            const u32 sh = sbyte & 3u;
            const uint8_t *bbase = win + (sbyte & ~3u);
            const uint8_t *abase = afrag + lane * 16;

            v4i acc[kBand];
            sweep_item<S>(acc, abase, bbase, wpitch, y0, wh, s, sh);

            // ---- epilogue: pre-filter, exact evaluation of candidates ----
            const int x = x0 + n_l;
#pragma unroll
            for (int t = 0; t < kBand; ++t) {
                const int y = y0 + t;
                const int swp = __shfl(acc[t][3], n_l + 48);              // slot 15 = sum of w'
                if (y < rh && x < rw) {
                    const double swd = (double)swp;
                    const double siid = (double)sii[y * rw + x];
                    const double dI = nd * siid - swd * swd;              // exact
                    const double s2 = siid + 256.0 * swd + 16384.0 * nd;   // S_II in the uint8 domain
                    const bool lowvar = (2.0 * dI <= nd) && (dI * 8388608.0 <= 10.0 * nd * s2);
                    const float rIf = __builtin_amdgcn_rsqf((float)dI);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int a = 4 * q_l + r;
                        if (a < Kg) {
                            const double numer = nd * (double)acc[t][r] - swd * m->sTd[a];   // exact
                            const bool cT = m->constT[a] != 0;
                            float est = ((float)numer * rIf) * m->rTf[a];
                            if (cT) est = 1.0f;
                            else if (lowvar) est = 0.0f;
                            if (est >= lmax - kMargin) {
                                const float rv = exact_ncc(numer, dI, m->rTd[a], cT, lowvar);
                                const int key = ((a0 + a) * rh + y) * rw + x;
                                if (rv > bestv || (rv == bestv && key < bestkey)) { bestv = rv; bestkey = key; }
                            }
                            lmax = est > lmax ? est : lmax;
                        }
                    }
                }
            }
        }
        __syncthreads();                                           // afrag is rewritten by the next group
    }

    // ---- P3: block arg-max (first angle, then first row-major index on ties) ----
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bestv, o);
        const int ok = __shfl_xor(bestkey, o);
        if (ov > bestv || (ov == bestv && ok < bestkey)) { bestv = ov; bestkey = ok; }
    }
    if (lane == 0) { m->red_f[wv] = bestv; m->red_i[wv] = bestkey; }
    __syncthreads();
    if (tid == 0) {
        float bv = m->red_f[0]; int bk = m->red_i[0];
        for (int w = 1; w < 4; ++w) {
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

    // ---- P4: winner's NCC matrix, row-major MFMA.  trow[g][16 + i][16 B], zero rows around ----
    const int trows = s + 32;
    for (int idx = tid; idx < 2 * L.trow_bytes / 4; idx += kBlock) reinterpret_cast<u32 *>(trow)[idx] = 0;
    __syncthreads();
    {
        int st = 0, stt = 0;
        for (int idx = tid; idx < s * s; idx += kBlock) {
            const int i = idx / s, j = idx - i * s;
            const uint8_t v = sample_template(A, A.rot + 4 * ka, c1, r1, i, j);
            const int off = ((j >> 4) * trows + 16 + i) * 16 + (j & 15);
            trow[off] = v ^ 0x80;
            trow1[off] = 1;
            const int sv = (int)v - 128;
            st += sv; stt += sv * sv;
        }
        const double std_ = block_sum((double)st, m);              // exact: small integers
        const double sttd = block_sum((double)stt, m);
        if (tid == 0) {
            const double dT = nd * sttd - std_ * std_;
            m->sTd[0] = std_;
            m->constT[0] = dT == 0.0 ? 1 : 0;
            m->rTd[0] = 1.0 / sqrt(dT);
        }
    }
    __syncthreads();
    {
        const double sT = m->sTd[0], rT = m->rTd[0];
        const bool cT = m->constT[0] != 0;
        const int nty = (rh + 15) / 16, ntx = (rw + 15) / 16;
        for (int item = wv; item < nty * ntx; item += 4) {
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
#pragma unroll 2
            for (int step = 0; step < nsteps; ++step) {
                const v4i b = load_bfrag(bbase + (y0 + step) * wpitch, sh);
                const v4i at = *reinterpret_cast<const v4i *>(ta + step * 16);
                const v4i a1 = *reinterpret_cast<const v4i *>(ta1 + step * 16);
                accT = __builtin_amdgcn_mfma_i32_16x16x64_i8(at, b, accT, 0, 0, 0);
                accS = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b, accS, 0, 0, 0);
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

    // ---- P5: Hessian at the peak (pmlib.py:36-59, :167); hes aliases sii ----
    for (int idx = tid; idx < npos; idx += kBlock) {
        const int y = idx / rw, x = idx - y * rw;
        const float d2x = grad2(ccm + y * rw, 1, x, rw);
        const float d2y = grad2(ccm + x, rw, y, rh);
        const double hh = (double)d2x * (double)d2x + (double)d2y * (double)d2y;
        hes[idx] = (float)sqrt(hh);
    }
    __syncthreads();
    if (A.dbg_ccm || A.dbg_hes) {
        for (int idx = tid; idx < npos && idx < A.dbg_cap; idx += kBlock) {
            if (A.dbg_ccm) A.dbg_ccm[idx] = ccm[idx];
            if (A.dbg_hes) A.dbg_hes[idx] = hes[idx];
        }
    }
    if (A.dbg_shape && tid == 0) { A.dbg_shape[0] = rh; A.dbg_shape[1] = rw; }
    float h = hes[iy * rw + ix];
    if (A.flags & 1u) {
        float med, sd;
        block_median_std(hes, npos, m, &med, &sd);
        h = (h - med) / sd;
    }
    float rr = best_r;
    if (A.flags & 4u) {
        float med, sd;
        block_median_std(ccm, npos, m, &med, &sd);
        rr = (best_r - med) / sd;
    }
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
    hipLaunchKernelGGL(kern, dim3(args.n_launch), dim3(kBlock), lds_bytes, st, args);
    return (int)hipGetLastError();
}

}  // namespace sid
