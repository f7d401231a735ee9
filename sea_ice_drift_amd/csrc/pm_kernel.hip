// pm_kernel.hip - the fused pattern-matching kernel for gfx950 (MI355X, CDNA4).
//
// One workgroup of 256 threads (4 wavefronts of 64) owns one grid point and performs the
// whole of the reference's per-point operator (pmlib.py:176-212 use_mcc ->
// :117-174 rotate_and_match -> :89-115 get_template, :156 matchTemplate, :36-59 get_hessian):
//
//   phase 0  search window of image 2 -> LDS (aligned dword loads, byte-realigned);
//            K rotated nearest-neighbour templates of image 1 -> LDS (float64 coordinates,
//            one rounding per operation, as scipy's affine_transform order 0); zero-pixel
//            guard (pmlib.py:152-154); per-template sums.
//   phase 1  sweep: each thread owns strips of 8 adjacent NCC placements; window bytes are
//            read from LDS as 64-bit words, byte-shifted with v_alignbyte_b32 and multiplied
//            against broadcast template dwords with v_dot4_u32_u8 (exact integer sums);
//            normalisation in IEEE double per the NCC specification; running arg-max with
//            first-index / first-angle tie-breaking (np.argmax + the strict '>' of :160).
//   phase 2  the winning angle's NCC matrix is recomputed into LDS (float32).
//   phase 3  Hessian magnitude (np.gradient twice, hypot) in float32, exact median by
//            radix-select over an LDS histogram, population std; h at the peak.
//
// MFMA is not used: the inner product is a 34x34 uint8 stencil with 8-bit *unsigned* data and
// K<=15 templates; see DESIGN.md for the roofline discussion.
//
// Compile with -ffp-contract=off: the specification counts roundings.
#include <hip/hip_runtime.h>
#include <math.h>
#include "pm_kernel.h"

namespace sid {

namespace {

typedef uint32_t u32;

struct Misc {                       // lives at LDS offset 0, kMiscBytes reserved
    u32 hist[256];
    double red_d[8];
    float red_f[8];
    int red_i[8];
    u32 sel_prefix, sel_k, sel_bin, sel_less;
    int zero_flag;
    int best_key; float best_val;
    int pad_;
    double rT[kMaxAngles];          // 1/sqrt(dT)
    double sT[kMaxAngles];          // sum T (exact integer as double)
    int constT[kMaxAngles];         // dT == 0
};
static_assert(sizeof(Misc) <= kMiscBytes, "misc header too large");

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ u32 wave_sum(u32 v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum over the block of a double; result valid in every thread
__device__ __forceinline__ double block_sum(double v, Misc *m) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) m->red_d[w] = v;
    __syncthreads();
    return (m->red_d[0] + m->red_d[1]) + (m->red_d[2] + m->red_d[3]);
}
__device__ __forceinline__ u32 block_min(u32 v, Misc *m) {
    for (int o = 32; o > 0; o >>= 1) { u32 t = __shfl_xor(v, o); v = t < v ? t : v; }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) m->red_i[w] = (int)v;
    __syncthreads();
    u32 a = (u32)m->red_i[0], b = (u32)m->red_i[1], c = (u32)m->red_i[2], d = (u32)m->red_i[3];
    a = a < b ? a : b; c = c < d ? c : d;
    return a < c ? a : c;
}

// order-preserving map float32 -> uint32 (total order incl. negatives)
__device__ __forceinline__ u32 f2key(float f) {
    u32 b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(u32 k) {
    u32 b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// k-th smallest (0-based) key of v[0..n) by 4 x 8-bit radix select; also returns the number
// of elements <= that key.  All threads of the block call it; results valid in all threads.
__device__ u32 block_select(const float *v, int n, u32 k, Misc *m, u32 *count_le) {
    u32 prefix = 0, mask = 0, less_total = 0;
    u32 kk = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        __syncthreads();
        m->hist[threadIdx.x] = 0;                       // kBlock == 256 bins
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += kBlock) {
            const u32 key = f2key(v[i]);
            if ((key & mask) == prefix) atomicAdd(&m->hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        // inclusive scan of the 256 bins, one bin per thread
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
        if (shift == 0) {
            // elements equal to the selected key = its final-bin count
            const u32 eq = m->hist[bin];
            *count_le = less_total + eq;
        }
    }
    __syncthreads();
    return prefix;
}

// np.median(v) and np.std(v) of a float32 LDS array (float32 results, NumPy's formulas:
// mean of the two middle values for even n; std = sqrt(mean((v-mean)^2)) ).
__device__ void block_median_std(const float *v, int n, Misc *m, float *med, float *sd) {
    u32 cle;
    float md;
    if (n & 1) {
        md = key2f(block_select(v, n, (u32)(n / 2), m, &cle));
    } else {
        const u32 k1 = (u32)(n / 2 - 1);
        const u32 key1 = block_select(v, n, k1, m, &cle);
        u32 key2 = key1;
        if (cle < k1 + 2) {                              // next order statistic is a larger value
            u32 mn = 0xffffffffu;
            for (int i = threadIdx.x; i < n; i += kBlock) {
                const u32 key = f2key(v[i]);
                if (key > key1 && key < mn) mn = key;
            }
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
    const float var = (float)(q / (double)n);
    *med = md;
    *sd = sqrtf(var);
}

// 1-D np.gradient (unit spacing, edge_order 1) and its second application, float32
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

// ---------------------------------------------------------------------------------------
// The strip sweep.  A task = 8 adjacent placements (y, x0..x0+7).  NCH = dwords per template
// row, GA = angles accumulated together.  MODE 0: track the block arg-max over angles
// [0,K).  MODE 1: write the float32 NCC of angle `ka` to ccm.
// ---------------------------------------------------------------------------------------
template <int NCH>
struct XRow {
    u32 x[4][NCH + 1];              // x[sh][q] = bytes [4q+sh, 4q+sh+4) of the row segment
    __device__ __forceinline__ void load(const uint8_t *p) {
        constexpr int NR = (NCH + 2 + 1) / 2 * 2;
        u32 raw[NR];
        const uint2 *p2 = reinterpret_cast<const uint2 *>(p);
#pragma unroll
        for (int q = 0; q < NR / 2; ++q) { const uint2 t = p2[q]; raw[2 * q] = t.x; raw[2 * q + 1] = t.y; }
#pragma unroll
        for (int q = 0; q <= NCH; ++q) {
            x[0][q] = raw[q];
            x[1][q] = __builtin_amdgcn_alignbyte(raw[q + 1], raw[q], 1);
            x[2][q] = __builtin_amdgcn_alignbyte(raw[q + 1], raw[q], 2);
            x[3][q] = __builtin_amdgcn_alignbyte(raw[q + 1], raw[q], 3);
        }
    }
};

template <int NCH, int GA, int MODE>
__device__ void sweep(const uint8_t *win, int wpitch, const u32 *tmpl, int trow_dw, int s, int K,
                      int rh, int rw, const Misc *m, int ka, float *ccm, float &bestv, int &bestkey)
{
    const int nstrips = (rw + kStrip - 1) / kStrip;
    const int ntasks = rh * nstrips;
    const int nb_last = s - 4 * (NCH - 1);                           // valid bytes of the last chunk
    const u32 lastmask = nb_last >= 4 ? 0xffffffffu : ((1u << (8 * nb_last)) - 1u);
    const double nd = (double)(s * s);

    for (int t = threadIdx.x; t < ntasks; t += kBlock) {
        const int y = t / nstrips, x0 = (t - y * nstrips) * kStrip;
        const uint8_t *wbase = win + y * wpitch + x0;

        // ---- window sums S_I, S_II for the 8 placements (exact) ----
        u32 si[kStrip], sii[kStrip];
#pragma unroll
        for (int d = 0; d < kStrip; ++d) { si[d] = 0; sii[d] = 0; }
        for (int i = 0; i < s; ++i) {
            XRow<NCH> xr;
            xr.load(wbase + i * wpitch);
#pragma unroll
            for (int d = 0; d < kStrip; ++d) {
#pragma unroll
                for (int jc = 0; jc < NCH; ++jc) {
                    u32 w = xr.x[d & 3][(d >> 2) + jc];
                    if (jc == NCH - 1) w &= lastmask;
                    si[d] = __builtin_amdgcn_udot4(w, 0x01010101u, si[d], false);
                    sii[d] = __builtin_amdgcn_udot4(w, w, sii[d], false);
                }
            }
        }
        double rI[kStrip];
        u32 lowvar = 0;
#pragma unroll
        for (int d = 0; d < kStrip; ++d) {
            const double sid_ = (double)si[d], siid = (double)sii[d];
            const double dI = nd * siid - sid_ * sid_;                // exact: integers < 2^53
            // OpenCV: diff2 <= min(0.5, 10*FLT_EPSILON*wndSum2) with diff2 = dI/N
            const bool lv = (2.0 * dI <= nd) && (dI * 8388608.0 <= 10.0 * nd * siid);
            lowvar |= (lv ? 1u : 0u) << d;
            rI[d] = 1.0 / sqrt(dI);
        }

        // ---- correlation, GA angles at a time ----
        const int a_begin = MODE == 1 ? ka : 0;
        const int a_end = MODE == 1 ? ka + 1 : K;
        for (int a0 = a_begin; a0 < a_end; a0 += GA) {
            u32 acc[GA][kStrip];
#pragma unroll
            for (int g = 0; g < GA; ++g)
#pragma unroll
                for (int d = 0; d < kStrip; ++d) acc[g][d] = 0;

            for (int i = 0; i < s; ++i) {
                XRow<NCH> xr;
                xr.load(wbase + i * wpitch);
#pragma unroll
                for (int g = 0; g < GA; ++g) {
                    if (a0 + g < a_end) {                             // wave-uniform
                        const uint4 *tr = reinterpret_cast<const uint4 *>(tmpl + ((a0 + g) * s + i) * trow_dw);
                        u32 tw[(NCH + 3) / 4 * 4];
#pragma unroll
                        for (int q = 0; q < (NCH + 3) / 4; ++q) {
                            const uint4 v = tr[q];
                            tw[4 * q] = v.x; tw[4 * q + 1] = v.y; tw[4 * q + 2] = v.z; tw[4 * q + 3] = v.w;
                        }
#pragma unroll
                        for (int d = 0; d < kStrip; ++d)
#pragma unroll
                            for (int jc = 0; jc < NCH; ++jc)
                                acc[g][d] = __builtin_amdgcn_udot4(xr.x[d & 3][(d >> 2) + jc], tw[jc], acc[g][d], false);
                    }
                }
            }

            // ---- normalise (NCC specification) and consume ----
#pragma unroll
            for (int g = 0; g < GA; ++g) {
                const int a = a0 + g;
                if (a < a_end) {
                    const double sTa = m->sT[a], rTa = m->rT[a];
                    const bool cT = m->constT[a] != 0;
#pragma unroll
                    for (int d = 0; d < kStrip; ++d) {
                        float r;
                        if (cT) r = 1.0f;
                        else if ((lowvar >> d) & 1u) r = 0.0f;
                        else {
                            const double numer = nd * (double)acc[g][d] - (double)si[d] * sTa;   // exact
                            double q = numer * rI[d];
                            q = q * rTa;
                            const double aq = fabs(q);
                            r = aq < 1.0 ? (float)q : (aq < 1.125 ? (q > 0.0 ? 1.0f : -1.0f) : 0.0f);
                        }
                        const int x = x0 + d;
                        if (x < rw) {
                            if (MODE == 1) {
                                ccm[y * rw + x] = r;
                            } else {
                                const int key = (a * rh + y) * rw + x;
                                if (r > bestv || (r == bestv && key < bestkey)) { bestv = r; bestkey = key; }
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int NCH, int GA>
__global__ __launch_bounds__(kBlock) void pm_kernel(const PMArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Misc *m = reinterpret_cast<Misc *>(smem);
    const int tid = threadIdx.x;
    const int pt = A.order[blockIdx.x];
    const int s = A.img_size, K = A.n_angles;

    double *out = A.out + (int64_t)pt * 5;
    int32_t *oij = A.out_ij ? A.out_ij + (int64_t)pt * 3 : nullptr;

    // ---- window geometry (pmlib.py:200-202; Python int() truncates toward zero) ----
    const double c2fg = A.c2fg[pt], r2fg = A.r2fg[pt], border = A.border[pt];
    const int hws = (int)((double)s / 2.0);
    const double r0d = r2fg - hws - border, r1d = r2fg + hws + border + 1;
    const double c0d = c2fg - hws - border, c1d = c2fg + hws + border + 1;
    const bool finite = fabs(r0d) < 1e15 && fabs(r1d) < 1e15 && fabs(c0d) < 1e15 && fabs(c1d) < 1e15;
    const int64_t r0 = finite ? (int64_t)r0d : -1, r1e = finite ? (int64_t)r1d : -1;
    const int64_t c0 = finite ? (int64_t)c0d : -1, c1e = finite ? (int64_t)c1d : -1;
    const bool inside = finite && r0 >= 0 && c0 >= 0 && r1e <= A.rows2 && c1e <= A.cols2 &&
                        r1e - r0 >= s + 1 && c1e - c0 >= s + 1;    // >= 2 placements per axis (np.gradient)
    if (!inside) {
        if (tid < 5) out[tid] = NAN;
        if (oij && tid < 3) oij[tid] = -1;
        return;
    }
    const int wh = (int)(r1e - r0), ww = (int)(c1e - c0);
    const int rh = wh - s + 1, rw = ww - s + 1;
    const LdsLayout L = lds_layout(wh, ww, s, K);
    uint8_t *win = smem + L.win_off;
    float *ccm = reinterpret_cast<float *>(smem + L.ccm_off);
    u32 *tmpl = reinterpret_cast<u32 *>(smem + L.tmpl_off);
    uint8_t *tmplb = smem + L.tmpl_off;
    float *hes = reinterpret_cast<float *>(smem + L.tmpl_off);     // aliases tmpl (dead by then)
    const int trow_dw = tmpl_row_dwords(s);

    if (tid == 0) { m->zero_flag = 0; }

    // ---- phase 0a: search window -> LDS ----
    {
        const int dw_per_row = L.wpitch / 4;
        const uint8_t *img_end = A.img2 + (A.rows2 - 1) * A.stride2 + A.cols2;
        for (int idx = tid; idx < wh * dw_per_row; idx += kBlock) {
            const int row = idx / dw_per_row, dq = idx - row * dw_per_row;
            u32 v = 0;
            if (4 * dq < ww) {
                const uint8_t *g = A.img2 + (r0 + row) * A.stride2 + c0 + 4 * dq;
                const uintptr_t ga = reinterpret_cast<uintptr_t>(g) & ~(uintptr_t)3;
                const u32 sh = (u32)(reinterpret_cast<uintptr_t>(g) & 3);
                const u32 lo = *reinterpret_cast<const u32 *>(ga);
                u32 hi = 0;
                if (sh != 0 && ga + 4 < reinterpret_cast<uintptr_t>(img_end)) hi = *reinterpret_cast<const u32 *>(ga + 4);
                v = __builtin_amdgcn_alignbyte(hi, lo, sh);
            }
            reinterpret_cast<u32 *>(win + row * L.wpitch)[dq] = v;
        }
    }
    // ---- phase 0b: zero the template rows' padding, then sample the K rotated templates ----
    for (int idx = tid; idx < K * s * trow_dw; idx += kBlock) tmpl[idx] = 0;
    __syncthreads();
    {
        const double c1 = A.c1[pt], r1 = A.r1[pt];
        const double rmax = (double)(A.rows1 - 1), cmax = (double)(A.cols1 - 1);
        int sawzero = 0;
        for (int idx = tid; idx < K * s * s; idx += kBlock) {
            const int a = idx / (s * s), rem = idx - a * s * s;
            const int i = rem / s, j = rem - i * s;
            const double cosa = A.rot[4 * a], sina = A.rot[4 * a + 1];
            const double off0 = r1 - A.rot[4 * a + 2], off1 = c1 - A.rot[4 * a + 3];
            // scipy NI_GeometricTransform order of operations (matrix = transform.T)
            double rr = 0.0 + (double)i * cosa;
            rr = rr + (double)j * sina;
            rr = rr + off0;
            double cc = 0.0 + (double)i * (-sina);
            cc = cc + (double)j * cosa;
            cc = cc + off1;
            uint8_t v = 0;
            if (rr >= 0.0 && rr <= rmax && cc >= 0.0 && cc <= cmax) {
                const int64_t ri = (int64_t)floor(rr + 0.5), ci = (int64_t)floor(cc + 0.5);
                v = A.img1[ri * A.stride1 + ci];
            }
            if (v == 0) sawzero = 1;
            tmplb[((a * s + i) * trow_dw) * 4 + j] = v;
        }
        if (sawzero) m->zero_flag = 1;                            // benign race: all writers store 1
    }
    __syncthreads();
    if (A.dbg_templates) {
        for (int idx = tid; idx < K * s * s; idx += kBlock) {
            const int a = idx / (s * s), rem = idx - a * s * s;
            const int i = rem / s, j = rem - i * s;
            A.dbg_templates[idx] = tmplb[((a * s + i) * trow_dw) * 4 + j];
        }
    }
    if (m->zero_flag) {                                            // pmlib.py:152-154 -> NaN x 5
        if (tid < 5) out[tid] = NAN;
        if (oij && tid < 3) oij[tid] = -1;
        if (A.dbg_shape && tid == 0) { A.dbg_shape[0] = rh; A.dbg_shape[1] = rw; }
        return;
    }
    // per-template sums: wave w takes angles w, w+4, ...
    {
        const int lane = tid & 63, w = tid >> 6;
        const double nd = (double)(s * s);
        for (int a = w; a < K; a += 4) {
            u32 st = 0, stt = 0;
            for (int idx = lane; idx < s * s; idx += 64) {
                const int i = idx / s, j = idx - i * s;
                const u32 v = tmplb[((a * s + i) * trow_dw) * 4 + j];
                st += v; stt += v * v;
            }
            st = wave_sum(st); stt = wave_sum(stt);
            if (lane == 0) {
                const double dT = nd * (double)stt - (double)st * (double)st;   // exact
                m->sT[a] = (double)st;
                m->constT[a] = dT == 0.0 ? 1 : 0;
                m->rT[a] = 1.0 / sqrt(dT);
            }
        }
    }
    __syncthreads();

    // ---- phase 1: sweep all angles, block arg-max ----
    float bestv = -INFINITY;
    int bestkey = 0x7fffffff;
    sweep<NCH, GA, 0>(win, L.wpitch, tmpl, trow_dw, s, K, rh, rw, m, 0, nullptr, bestv, bestkey);
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bestv, o);
        const int ok = __shfl_xor(bestkey, o);
        if (ov > bestv || (ov == bestv && ok < bestkey)) { bestv = ov; bestkey = ok; }
    }
    if ((tid & 63) == 0) { m->red_f[tid >> 6] = bestv; m->red_i[tid >> 6] = bestkey; }
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
    const int ka = best_key / (rh * rw);
    const int bidx = best_key - ka * rh * rw;
    const int iy = bidx / rw, ix = bidx - iy * rw;

    // ---- phase 2: NCC matrix of the winning angle -> LDS ----
    {
        float dv = 0.f; int dk = 0;
        sweep<NCH, 1, 1>(win, L.wpitch, tmpl, trow_dw, s, K, rh, rw, m, ka, ccm, dv, dk);
    }
    __syncthreads();

    // ---- phase 3: Hessian at the peak (pmlib.py:36-59, :167) ----
    const int n = rh * rw;
    for (int idx = tid; idx < n; idx += kBlock) {
        const int y = idx / rw, x = idx - y * rw;
        const float d2x = grad2(ccm + y * rw, 1, x, rw);
        const float d2y = grad2(ccm + x, rw, y, rh);
        const double hh = (double)d2x * (double)d2x + (double)d2y * (double)d2y;
        hes[idx] = (float)sqrt(hh);                               // hypotf: double sqrt, narrowed
    }
    __syncthreads();
    if (A.dbg_ccm || A.dbg_hes) {
        for (int idx = tid; idx < n && idx < A.dbg_cap; idx += kBlock) {
            if (A.dbg_ccm) A.dbg_ccm[idx] = ccm[idx];
            if (A.dbg_hes) A.dbg_hes[idx] = hes[idx];
        }
    }
    if (A.dbg_shape && tid == 0) { A.dbg_shape[0] = rh; A.dbg_shape[1] = rw; }
    float h = hes[iy * rw + ix];
    if (A.flags & 1u) {                                            // hes_norm
        float med, sd;
        block_median_std(hes, n, m, &med, &sd);
        h = (h - med) / sd;
    }
    float rr = best_r;
    if (A.flags & 4u) {                                            // mcc_norm (pmlib.py:171-172)
        float med, sd;
        block_median_std(ccm, n, m, &med, &sd);
        rr = (best_r - med) / sd;
    }
    if (tid == 0) {
        const double dr = (double)iy - (double)(wh - s) / 2.0;
        const double dc = (double)ix - (double)(ww - s) / 2.0;
        out[0] = c2fg + dc;
        out[1] = r2fg + dr;
        out[2] = A.angles[ka];
        out[3] = (double)rr;
        out[4] = (double)h;
        if (oij) { oij[0] = iy; oij[1] = ix; oij[2] = ka; }
    }
}

__global__ void rsqrt_kernel(const double *x, double *y, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 1.0 / sqrt(x[i]);
}

}  // namespace

bool img_size_supported(int s) { return s >= 33 && s <= 36; }

int max_lds_bytes() { return 160 * 1024; }

int launch_pm(const PMArgs &args, int lds_bytes, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (args.n_launch <= 0) return (int)hipSuccess;
    constexpr int GA = 8;
    auto kern = pm_kernel<9, GA>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(args.n_launch), dim3(kBlock), lds_bytes, st, args);
    return (int)hipGetLastError();
}

int launch_rsqrt(const double *x, double *y, int64_t n, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n <= 0) return (int)hipSuccess;
    hipLaunchKernelGGL(rsqrt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n);
    return (int)hipGetLastError();
}

}  // namespace sid
