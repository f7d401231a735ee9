// Shared between the HIP kernel (pm_kernel.hip) and the host C-ABI (pm_capi.hip).
#pragma once
#include <stdint.h>

namespace sid {

constexpr int kBlock = 256;          // threads per grid point: 4 wavefronts of 64
constexpr int kStrip = 8;            // NCC outputs per thread-task along a row
constexpr int kMaxAngles = 64;
constexpr int kMiscBytes = 4096;     // fixed LDS header: histogram, reduction scratch, per-angle terms

// Arguments of one launch (one block per grid point of this launch).
struct PMArgs {
    const uint8_t *img1; int64_t rows1, cols1, stride1;
    const uint8_t *img2; int64_t rows2, cols2, stride2;
    const double *c1, *r1, *c2fg, *r2fg, *border;   // [n_total], original point order
    const int32_t *order;                           // [n_launch] -> original point index
    int32_t n_launch;
    int32_t img_size;                               // s
    int32_t n_angles;                               // K
    uint32_t flags;
    const double *angles;                           // [K] degrees (reported as-is)
    const double *rot;                              // [K][4] cos, sin, tcT0, tcT1
    // MFMA kernel: sampling table [K][s][samp_pitch(s)] of LDS patch offsets for integral template centres;
    // bit 15 = coordinate too close to a rounding boundary: the offset then points at the spare byte behind
    // the patch (value 128 = re-centred 0) and the sample is redone in double by ph_tpl_fix.
    // null = always sample on the fly
    const uint16_t *samp;
    int32_t samp_nflag;                             // number of flagged entries in samp (0 almost always)
    double gauss_w[5];                              // hes_smth: normalised sigma-1 Gaussian taps w[0] (|k| = 4) .. w[4] (centre), host-computed
    double *out;                                    // [n_total][5]
    int32_t *out_ij;                                // [n_total][3]
    // diagnostics (debug_point only; null in production launches)
    uint8_t *dbg_templates; float *dbg_ccm; float *dbg_hes; int32_t *dbg_shape; int64_t dbg_cap;
    long long *dbg_cycles;                          // [16] shader-clock stamps at phase boundaries
};

// LDS carve-up for a window of wh x ww pixels, template side s, K angles.
struct LdsLayout {
    int wpitch;      // bytes per window row in LDS
    int win_off, ccm_off, tmpl_off, total;
};

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// dwords per template row in LDS (zero padded, multiple of 4 so rows are 16-byte aligned)
__host__ __device__ inline int tmpl_row_dwords(int s) { return round_up((s + 3) / 4, 4); }

__host__ __device__ inline LdsLayout lds_layout(int wh, int ww, int s, int K)
{
    LdsLayout L;
    const int rh = wh - s + 1, rw = ww - s + 1;
    const int nch = (s + 3) / 4;
    // a strip task reads (nch + 2) dwords rounded up to an even count from its first output
    L.wpitch = round_up(rw, kStrip) - kStrip + round_up(nch + 2, 2) * 4;
    if (L.wpitch < round_up(ww, 8)) L.wpitch = round_up(ww, 8);
    L.win_off = kMiscBytes;
    L.ccm_off = round_up(L.win_off + wh * L.wpitch, 16);
    L.tmpl_off = round_up(L.ccm_off + rh * rw * 4, 16);
    const int tmpl_bytes = K * s * tmpl_row_dwords(s) * 4;
    const int hes_bytes = rh * rw * 4;              // aliases the templates once they are dead
    L.total = round_up(L.tmpl_off + (tmpl_bytes > hes_bytes ? tmpl_bytes : hes_bytes), 16);
    return L;
}

// ---- MFMA kernel (pm_kernel_mfma.hip) ----
constexpr int kMiscMfmaBytes = 2816;
constexpr int kTrowPad = 36;         // zero rows around a winner operand block: 16 above, 20 below (the step loop runs in fours)
constexpr int kQueueCap = 192;       // arg-max candidates waiting for exact evaluation (16 B each)

struct MfmaLdsLayout {
    int wpitch;        // bytes per window row (int8, re-centred)
    int win_off;       // window
    int sii_off;       // sum w'^2 per placement (u32); later the Hessian (f32)
    int u_off;         // union: column sums | sweep operands + patch + queue | winner operands + NCC matrix
    int arow;          // bytes per template row in the sweep operand table (48 or 64 lanes x 16 B; half in paired mode)
    int gpitch;        // bytes between the k-groups of a table row: 16 slots x 16 B, or 8 x 16 B in paired mode
    int arow0;         // byte offset of template row 0 in the table (paired mode: 4 zero rows precede it)
    int tab_rows;      // rows of the table incl. its zero rows; 16 scratch bytes follow at tab_rows * arow
    int patch_off, ppitch, pdim, pradius;   // image-1 patch the templates are sampled from
    int queue_off;
    int trow_bytes;    // one winner operand block: 4 k-groups x (s+kTrowPad) rows x 16 B
    int total;
};

// band = accumulator rows per sweep work item (4, or 8 in the two-workgroups-per-CU kernel).
// paired (at most 7 angles): the 16 MFMA slots carry the <= 7 angles + the all-ones template twice, slots 8..15
// working four output rows below slots 0..7, so an item covers 2 * band rows; the table stores 8 slots per
// k-group and the lanes of slots 8..15 read it four rows earlier (zero rows at both ends).
// Zero rows follow the window so that the sweep steps past it without clamping.
__host__ __device__ inline MfmaLdsLayout mfma_lds_layout(int wh, int ww, int s, int band = 4, bool paired = false)
{
    MfmaLdsLayout L;
    const int rh = wh - s + 1, rw = ww - s + 1;
    const int ntx = (rw + 15) / 16;
    // a window fragment reads 5 dwords from (x0 + 15 + 16 g) & ~3; for s <= 48 the lanes of k-group g = 3 only
    // multiply zero template columns and read along with k-group 0, so the last needed byte is x0 + 63
    L.wpitch = 16 * ntx + (s <= 48 ? 52 : 68);
    if (L.wpitch < round_up(ww, 4)) L.wpitch = round_up(ww, 4);
    L.win_off = kMiscMfmaBytes;
    L.sii_off = round_up(L.win_off + (wh + (paired ? 2 * band : band) - 1) * L.wpitch, 16);
    L.u_off = round_up(L.sii_off + rh * rw * 4, 16);
    L.gpitch = paired ? 128 : 256;
    L.arow = (s <= 48 ? 3 : 4) * L.gpitch;          // columns >= 48 of a template row are zero unless s = 49
    L.arow0 = paired ? 4 * L.arow : 0;
    L.tab_rows = paired ? s + 9 : s + 1;            // unpaired: one all-zero row behind the template
    // rotated template samples stay within hypot(tc, tc) of the centre, tc = int(s/2)+1 (pmlib.py:105)
    const int tc = s / 2 + 1;
    int r = 0;
    while (r * r < 2 * tc * tc) ++r;                // ceil(hypot(tc, tc))
    L.pradius = r + 1;
    L.pdim = 2 * L.pradius + 2;
    L.ppitch = round_up(L.pdim, 4);
    L.patch_off = L.u_off + round_up(L.tab_rows * L.arow + 16, 16);   // + 16 scratch bytes
    L.queue_off = round_up(L.patch_off + L.pdim * L.ppitch + 1, 16);  // + one byte = 128 behind the patch (flagged table entries)
    L.trow_bytes = 4 * (s + kTrowPad) * 16;
    int u = rh * ww * 4;                                          // column sums
    const int sweep = L.queue_off + kQueueCap * 16 - L.u_off;
    if (u < sweep) u = sweep;
    if (u < 2 * L.trow_bytes + rh * rw * 4) u = 2 * L.trow_bytes + rh * rw * 4;
    L.total = round_up(L.u_off + u, 16);
    return L;
}

__host__ __device__ inline int samp_pitch(int s) { return round_up(s, 4); }
constexpr double kSampGuard = 1e-5;   // table entries whose coordinate is this close to k + 1/2 are flagged

int launch_pm_mfma(const PMArgs &args, int lds_bytes, int nthreads, int band, bool paired, void *stream);
constexpr int kPairedMaxAngles = 7;   // angle sets this small run the paired sweep
bool mfma_band8_supported(int s);
bool mfma_img_size_supported(int s);

// host-side launcher implemented in pm_kernel.hip; returns a hipError_t as int
int launch_pm(const PMArgs &args, int lds_bytes, void *stream);
int launch_rsqrt(const double *x, double *y, int64_t n, void *stream);
// true when a kernel instantiation exists for this template side
bool img_size_supported(int s);
int max_lds_bytes();

}  // namespace sid
