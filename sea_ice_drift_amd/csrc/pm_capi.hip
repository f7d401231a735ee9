// pm_capi.hip - host side of the C ABI declared in include/sid_pm.h.
//
// Replaces the reference's process pool (pmlib.py:430-448): the "shared_args" of the
// workers become device-resident buffers owned by a handle, the Pool.map over point
// indices becomes a few kernel launches (one per LDS-footprint class), and the pickled
// 5-tuples coming back become one device array copied to the host.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/sid_pm.h"
#include "pm_kernel.h"
#include "pm_large.h"

#define SID_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? SID_PM_ERR_NOMEM : SID_PM_ERR_HIP,         \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                        \
    } while (0)

struct Image {
    const uint8_t *ptr = nullptr;       // device
    int64_t rows = 0, cols = 0, stride = 0;
};

struct Bucket { int offset, count, lds, band, pitch, occ; uint32_t gs_stride = 0; bool big = false; bool keep = false; bool pooled = false; };   // pooled: the launch takes its blocks of global memory from the free lists (PMArgs::ring)   // keep: the launch keeps the sweep's accumulators (PMArgs::gs_keep_acc) and takes recycled blocks   // gs_stride: largest sum w'^2 block of the launch's points, in u32 entries (gs launches; 0 otherwise)   // occ: wavefronts per SIMD of the kernel build (3; 4: the four-per-CU class of the slot-group layouts); band: output rows per sweep work item of this launch (4 or 8); pitch: compile-time window pitch of the row-pair kernel (0: run-time)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return SID_PM_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(n, 1) * sizeof(T)));
        cap = n;
        return SID_PM_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct sid_pm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // image pairs: two owned slots + one borrowed binding
    DevBuf<uint8_t> own[2][2];
    Image slot_img[2][2];
    Image cur[2];
    bool have_pair = false;
    int cur_slot = -1;                  // -1: borrowed binding (or none)
    // pair streaming: uploads run on their own stream; slot_ready[s] = upload of slot s complete,
    // slot_done[s] = last kernels reading slot s complete (an upload into s waits for it)
    hipStream_t copy_stream = nullptr;
    // short runs (a rank's shard of an N-GPU run: two or three launches of a few rounds of workgroups each) put their launches
    // side by side on these streams, so that one launch's half-empty last round is filled by the next (sid_pm_run)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr, join_ev[3] = {nullptr, nullptr, nullptr};
    hipEvent_t slot_ready[2] = {nullptr, nullptr}, slot_done[2] = {nullptr, nullptr};
    bool ready_rec[2] = {false, false}, done_rec[2] = {false, false};
    // resident points: one device arena (a single upload per set_points) carved into the vectors below
    DevBuf<uint8_t> arena;
    double *d_vec = nullptr;            // 5 * n
    int32_t *d_order = nullptr;
    double *d_angles = nullptr, *d_rot = nullptr;
    uint16_t *d_samp = nullptr;         // sampling table of the kernel (make_samp)
    uint32_t *d_samp2 = nullptr;        // ... sorted per angle for the row-pair kernel (make_samp2; SID_PM_NO_SAMP2=1: not built)
    bool rp = false;                    // the resident points run the row-pair kernel (decided at set_points: use_rp) ...
    int rp_paired = 0;                  // ... with slot groups: 0 none, 1 two groups (<= 7 angles), 2 four groups (<= 3 angles)
    bool gs_keep_si = true;             // the blocks of global memory are sized for sum w' as well (classify_points; SID_PM_NO_GSI=1: not)
    bool gs_keep_acc = false;           // ... or for the sweep's accumulators (PMArgs::gs_keep_acc; classify_points: keep_acc_policy)
    bool have_samp = false;             // SID_PM_NO_SAMP_TABLE=1 keeps the on-the-fly sampling (tests of the general sampler)
    int samp_nflag = 0;
    // host copies of what the classification needs: the launch classes depend on the shape of image 2, so a
    // pair of another shape (select_pair / bind_pair / upload_pair after set_points) is re-classified at run()
    std::vector<double> h_c2fg, h_r2fg, h_border, h_c1, h_r1;
    int64_t cls_rows2 = -1, cls_cols2 = -1;
    DevBuf<double> out;
    DevBuf<int32_t> out_ij;
    DevBuf<int32_t> dbg_err;            // debugging builds only (SID_PM_DEBUG_CHECK=1)
    DevBuf<uint32_t> gsii;              // row-pair kernel: sum w'^2 per placement of every resident point (PMArgs::gsii) ...
    DevBuf<uint32_t> d_goff;            // ... and the offset of every launch position's block in it (units of 64 entries)
    DevBuf<sid::PointRec> d_rec;        // row-pair kernel: one record per launch position (index, block offset, the five inputs)
    DevBuf<uint32_t> ring, pool;        // recycled blocks of the launches that keep accumulators (PMArgs::ring / pool; SID_PM_NO_RECYCLE=1: none)
    uint32_t pool_stride = 0;           // u32 entries per recycled block: the largest block of any launch that takes them
    int32_t *h_refused = nullptr;       // pinned, device-visible: valid points a launch could not hold (PMArgs::refused)
    double *user_out = nullptr;         // caller-owned result arrays (bind_results)
    int32_t *user_ij = nullptr;
    std::vector<Bucket> buckets;
    // points beyond the launch classes of the one-workgroup-per-point kernels (search window too large for the LDS, template side
    // above 64): each runs the large-window pipeline (pm_large.hip) after the launches of the others
    std::vector<int32_t> large_idx;
    DevBuf<int32_t> d_nan_idx;          // ... and, when NO other kernel of the run exists (template side above 64), the points without a valid window
    int n_nan_idx = 0;
    sid::LwWorkspace lw;
    // rot_order 2..5: image 1 through scipy's spline prefilter (coef[1]; coef[0] = scratch), valid for (pair_serial, order); and the
    // templates of the resident points sampled from it (pre), valid for (pair_serial, points_serial)
    DevBuf<double> coef[2];
    uint64_t pair_serial = 1, points_serial = 1, coef_pair = 0, pre_pair = 0, pre_points = 0;
    int coef_order = 0;
    DevBuf<uint8_t> pre;
    DevBuf<double> lw_small;            // rotate_and_match as a call of its own: angles, rotation terms, the five results
    int64_t n = 0;
    int img_size = 0, n_angles = 0;
    uint32_t flags = 0;
    bool have_points = false;
    double info[6] = {0, 0, 0, 0, 0, 0};
};

namespace {

struct Guard {
    int prev = -1;
    explicit Guard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~Guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// rotation terms [K][4] = cos, sin, tcT0, tcT1 of angle - alpha0 (pmlib.py:105-110), as the CALLER's NumPy computed them:
// required (include/sid_pm.h) - libm's cos / sin may differ from NumPy's in the last bit, which can tip a sample at a rounding tie
int make_rot(int n_angles, const double *rot_in, std::vector<double> &rot)
{
    if (!rot_in) return fail(SID_PM_ERR_ARG, "rot is required: [n_angles][4] = cos, sin, tcT0, tcT1 as the caller's NumPy computed them (pmlib.py:105-110; include/sid_pm.h)");
    rot.assign(rot_in, rot_in + 4 * (size_t)n_angles);
    for (const double v : rot) if (!(fabs(v) < 1e6)) return fail(SID_PM_ERR_ARG, "rot holds a non-finite or absurd value");
    return SID_PM_OK;
}

// Sampling table of the MFMA kernel.  The reference rounds the template centres to integers before the
// dispatch (pmlib.py:401), and for an integral centre (R, C) the nearest-neighbour sample positions of
// get_template (pmlib.py:105-113; scipy order 0: floor(x + 0.5)) are those of the centre (0, 0) shifted by
// (R, C) - unless a coordinate lies so close to k + 1/2 that the float64 roundings at the magnitude of R
// could tip it.  Those entries (|frac - boundary| < kSampGuard; expected: none or a handful per table) carry
// a flag and are recomputed per point in the kernel (ph_tpl_fix) with the reference's operation order.
// Entry = (row + pradius) * ppitch + (col + pradius): the byte offset inside the kernel's LDS patch.
int make_samp(const std::vector<double> &rot, int K, int s, std::vector<uint16_t> &tab)
{
    int nflag = 0;
    const sid::MfmaLdsLayout L = sid::mfma_lds_layout(s + 1, s + 1, s);
    const int sp = sid::samp_pitch(s);
    tab.assign((size_t)K * s * sp, 0);
    for (int k = 0; k < K; ++k) {
        const double cosa = rot[4 * k + 0], sina = rot[4 * k + 1];
        const double off0 = 0.0 - rot[4 * k + 2], off1 = 0.0 - rot[4 * k + 3];
        for (int i = 0; i < s; ++i)
            for (int j = 0; j < s; ++j) {
                double rr = 0.0 + (double)i * cosa;                    // NI_GeometricTransform order (matrix = transform.T)
                rr = rr + (double)j * sina;
                rr = rr + off0;
                double cc = 0.0 + (double)i * (-sina);
                cc = cc + (double)j * cosa;
                cc = cc + off1;
                const double fr = floor(rr + 0.5), fc = floor(cc + 0.5);
                const double dr = (rr + 0.5) - fr, dc = (cc + 0.5) - fc;
                bool doubt = !(dr >= sid::kSampGuard && dr <= 1.0 - sid::kSampGuard && dc >= sid::kSampGuard && dc <= 1.0 - sid::kSampGuard);
                int pr = (int)fr + L.pradius, pc = (int)fc + L.pradius;
                if (!(pr >= 0 && pr < L.pdim && pc >= 0 && pc < L.pdim)) doubt = true;
                // flagged: gather the spare byte behind the patch (128), the kernel's fix pass supplies the sample
                tab[((size_t)k * s + i) * sp + j] = doubt ? (uint16_t)((L.pdim * L.ppitch) | 0x8000) : (uint16_t)(pr * L.ppitch + pc);
                nflag += doubt ? 1 : 0;
            }
    }
    return nflag;
}

// The sampling table sorted per angle for the row-pair kernel (PMArgs::samp2): run records first, gather records after.
// A quad (row i, columns 4 jq .. 4 jq + 3, all inside the 32 main columns) is a RUN when none of its entries is flagged and
// its four patch offsets are consecutive.
void make_samp2(const std::vector<uint16_t> &tab, int K, int s, std::vector<uint32_t> &out)
{
    const int sp = sid::samp_pitch(s), nq = sp >> 2;
    out.assign((size_t)K * sid::kSamp2Words, 0u);
    for (int k = 0; k < K; ++k) {
        uint32_t *blk = out.data() + (size_t)k * sid::kSamp2Words;
        uint32_t nrun = 0, ngat = 0;
        uint16_t *ids = reinterpret_cast<uint16_t *>(blk + sid::kSamp2Id);
        for (int i = 0; i < s; ++i)
            for (int jq = 0; jq < nq; ++jq) {
                const uint16_t *e = tab.data() + ((size_t)k * s + i) * sp + 4 * jq;
                const bool full = 4 * jq + 3 < s && jq < 8;
                const bool run = full && !((e[0] | e[1] | e[2] | e[3]) & 0x8000u) && e[1] == e[0] + 1 && e[2] == e[0] + 2 && e[3] == e[0] + 3;
                if (run) blk[sid::kSamp2Run + nrun++] = (uint32_t)(e[0] & ~3u) | ((uint32_t)(e[0] & 3u) << 15) | ((uint32_t)i << 17) | ((uint32_t)jq << 23);
                else {
                    blk[sid::kSamp2Gat + 2 * ngat] = (uint32_t)e[0] | ((uint32_t)e[1] << 16);
                    blk[sid::kSamp2Gat + 2 * ngat + 1] = (uint32_t)e[2] | ((uint32_t)e[3] << 16);
                    ids[ngat++] = (uint16_t)(i * nq + jq);
                }
            }
        blk[0] = nrun; blk[1] = ngat;
    }
}

// window geometry of one point, the same arithmetic as the kernel (pmlib.py:200-202)
// Workgroups of `lds` bytes that fit one CU.  gfx950 hands LDS out in 1280-byte granules (160 KB / 128;
// measured with tools/ubench/lds_granule.hip: 53760 B -> 3 per CU, 53761 B -> 2).
// workgroups of four wavefronts a CU takes at the kernels' register budget (3 wavefronts per SIMD): launch classes beyond
// this differ in nothing.  The row-pair kernel's slot-group layouts (<= 7 angles) have a 128-VGPR build as well, for the
// borders whose LDS footprint fits four times (SID_PM_NO_OCC4=1: off; A/B runs).
constexpr bool kRecycleAllDefault = true;    // (round 6: -1.0 % on the 15-angle step, HBM traffic 12.7x -> 10.1x the algorithmic bytes: profiles/r06_ab_recycle_all.txt)
constexpr int kMaxPerCu = 3;
constexpr int kW3MaxLds = 40960;   // every window of pitch 104 that fits four times (borders 20 .. 23 at 34 / 35 px; measured per border: tools/archive/r4_w3.sh)
// four workgroups per CU need the 128-VGPR build of the row-pair kernel (pm_kernel_rp_occ4.hip): the slot-group layouts only.
// (Round 4 measured it for the full operand table as well - with sum w'^2 in global memory borders 20-26 fit four per CU:
// border 20 +1.6 %, border 26 +5 % against three per CU with the sums in LDS; not shipped.)  SID_PM_NO_OCC4=1: never (A/B runs)
// Four points per CU with three wavefronts per point (192 threads: 4 x 3 = the 12 wavefronts of 168 VGPRs a CU holds; pitch code
// 1104: pm_kernel.h rp_pitch_is_w3) - the smallest windows only (SID_PM_W3_MAX_LDS: footprint up to which the class is used; SID_PM_NO_W3=1: never; A/B runs)
int w3_max_lds() { const char *e = getenv("SID_PM_W3_MAX_LDS"); return getenv("SID_PM_NO_W3") ? 0 : e ? atoi(e) : kW3MaxLds; }
int max_per_cu(bool rp, int rpp) { return (rp && (w3_max_lds() > 0 || (rpp > 0 && getenv("SID_PM_NO_OCC4") == nullptr))) ? 4 : kMaxPerCu; }

int blocks_per_cu(int lds)
{
    constexpr int kLdsGranule = 1280;
    const int granules = (lds + kLdsGranule - 1) / kLdsGranule;
    return granules > 0 ? (sid::max_lds_bytes() / kLdsGranule) / granules : 8;
}

bool window_dims(double c2fg, double r2fg, double border, int s, int64_t rows2, int64_t cols2,
                 int &wh, int &ww)
{
    const int hws = (int)((double)s / 2.0);
    const double r0d = r2fg - hws - border, r1d = r2fg + hws + border + 1;
    const double c0d = c2fg - hws - border, c1d = c2fg + hws + border + 1;
    if (!(fabs(r0d) < 1e15 && fabs(r1d) < 1e15 && fabs(c0d) < 1e15 && fabs(c1d) < 1e15)) return false;
    const int64_t r0 = (int64_t)r0d, r1e = (int64_t)r1d, c0 = (int64_t)c0d, c1e = (int64_t)c1d;
    if (!(r0 >= 0 && c0 >= 0 && r1e <= rows2 && c1e <= cols2 && r1e - r0 >= s + 1 && c1e - c0 >= s + 1))
        return false;
    wh = (int)(r1e - r0); ww = (int)(c1e - c0);
    return true;
}

// Launches of a SHORT run go side by side on the handle's side streams (sid_pm_run): all launches of the run together are at most
// kSideRounds rounds of workgroups (SID_PM_SIDE_BY_SIDE=0 / 1: never / always; A/B runs).  ONE rule for the launcher and for
// the estimate the shard cuts are made with (sid_pm_estimate_run_time).
constexpr double kSideRounds = 16.0;
bool side_by_side_rule(size_t n_launches, double rounds)
{
    if (n_launches < 2) return false;
    const char *e = getenv("SID_PM_SIDE_BY_SIDE");
    return e ? atoi(e) > 0 : rounds <= kSideRounds;
}

int check_images(const Image &a, const Image &b)
{
    if (!a.ptr || !b.ptr) return fail(SID_PM_ERR_ARG, "null image pointer");
    if (a.rows < 1 || a.cols < 1 || a.stride < a.cols || b.rows < 1 || b.cols < 1 || b.stride < b.cols)
        return fail(SID_PM_ERR_ARG, "bad image shape/stride");
    return SID_PM_OK;
}

// paired: the kernel's two-row-phase sweep for angle sets of at most kPairedMaxAngles (pm_kernel.h)
bool use_paired(int K)
{
    static const bool off = getenv("SID_PM_NO_PAIRED") != nullptr;                            // A/B runs
    return K <= sid::kPairedMaxAngles && !off;
}

// row-pair kernel (pm_kernel_rp.inc): template sides 34 / 35, any number of angles.  (Up to 7 angles the classic kernel
// pairs its slots - 8 output rows per item - and still loses to the row-pair kernel with half of its slots idle: 200x200
// benchmark, 7 angles, mixed borders 4.59 vs 4.35 ms, border 20 3.28 vs 3.05, border 50 16.2 vs 14.8; 3 angles 4.52 vs 4.25.)
// SID_PM_NO_RP (A/B runs and the tests of the classic kernel at these sizes) is read at every set_points / debug_point;
// the decision is kept with the handle.
bool use_rp(int s, int K)
{
    (void)K;
    return sid::rp_size_supported(s) && getenv("SID_PM_NO_RP") == nullptr;
}

// row-pair kernel with at most 7 angles: paired slots (8 output rows per item with the 4-row kernel's registers; the LDS
// layout is that of an 8-row band)
int rp_paired(int K)
{
    if (K > sid::kPairedMaxAngles || getenv("SID_PM_NO_PAIRED") != nullptr) return 0;
    return (K <= sid::kQuadMaxAngles && getenv("SID_PM_NO_QUAD") == nullptr) ? 2 : 1;   // (SID_PM_NO_QUAD: two groups also for <= 3 angles; A/B runs)
}

// band argument of rp_lds_layout: output rows per sweep work item
int rp_rows(int rpp, int band) { return rpp == 2 ? 16 : rpp == 1 ? 8 : band; }

// own_hes of rp_lds_layout: the general Hessian (hes_smth / mcc_norm, several groups of angles) keeps the magnitudes in LDS of
// their own (as the kernel decides: pm_kernel_rp prologue)
bool rp_own_hes(int K, uint32_t flags) { return K > sid::kRpGroup || (flags & (SID_PM_HES_SMTH | SID_PM_MCC_NORM)) != 0u; }

// gs: sum w'^2 per placement in global memory (rp_lds_layout; decided per window shape by rp_use_gs, and what the kernel
// instantiation of the launch's window pitch does: sid::rp_pitch_is_gs)
int lds_need(bool rp, int rpp, int wh, int ww, int s, int K, uint32_t flags, int band = 4, int pitch = 0, bool gs = true)
{
    if (rp) return sid::rp_lds_layout(wh, ww, s, K <= sid::kRpGroup, rp_rows(rpp, band), sid::rp_pitch_bytes(pitch), sid::rp_tab_pitch(rpp), rp_own_hes(K, flags), gs).total;
    return sid::mfma_lds_layout(wh, ww, s, band, use_paired(K) && band == 4).total;
}

// big layouts (rp_lds_layout): every per-placement table in the point's block of global memory; always the full-table 4-row kernel
sid::RpLdsLayout big_layout(int wh, int ww, int s, int K, uint32_t flags)
{
    return sid::rp_lds_layout(wh, ww, s, K <= sid::kRpGroup, 4, 0, 512, rp_own_hes(K, flags), true, true);
}

// full-table row-pair launches with the sums in global memory also keep sum w' there (the winner then multiplies no all-ones
// operand); decided when the blocks are sized (classify_points) and remembered for the launches (SID_PM_NO_GSI=1: off - A/B runs)
bool gs_keep_si(const sid_pm_ctx *ctx) { return ctx->rp && ctx->gs_keep_si; }

// Kept accumulators (PMArgs::gs_keep_acc, round 5): the gs launches of a run with ONE group of angles store the sweep's exact
// S_IT' of every slot and placement in the point's block of global memory, and the winner's NCC matrix is a normalisation of
// stored sums.  Default: the slot-group layouts (at most 7 angles - 16 / 32 bytes per placement; the reference's default is 3
// angles); SID_PM_KEEP_ACC=1: also the full table (64 bytes per placement), =0: never (A/B runs and the tests of both routes).
bool keep_acc_policy(bool rp, int rpp, int K)
{
    if (!rp || K > sid::kRpGroup) return false;
    if (rpp == 0 && !sid::kKeepAccFull) return false;                  // (the full-table kernels carry the code in measurement builds only)
    const char *e = getenv("SID_PM_KEEP_ACC");
    return e ? atoi(e) > 0 : rpp > 0;
}
// ... per launch: four slot groups (at most 3 angles, 16 B per placement) everywhere; two groups (at most 7 angles, 32 B) only
// in the four-per-CU class of the smallest windows - at larger borders the stores cost more than the winner's matrix
// instructions they replace (7 angles, mixed borders: +1.5 % with, -2 % at border 20; SID_PM_KEEP_ACC=1 keeps them everywhere)
bool keep_acc_launch(const sid_pm_ctx *ctx, bool gs, bool big, int cls)
{
    if (!ctx->gs_keep_acc || !gs || big) return false;
    if (ctx->rp_paired == 1 && cls != 4) { const char *e = getenv("SID_PM_KEEP_ACC"); return e && atoi(e) > 0; }
    return true;
}
// u32 entries of the block of a point of that shape in a gs launch
uint32_t block_entries(const sid_pm_ctx *ctx, int wh, int ww, int band, bool keep)
{
    const int s = ctx->img_size;
    return sid::rp_block_entries(wh - s + 1, ww - s + 1, rp_rows(ctx->rp_paired, band), 16 >> ctx->rp_paired, gs_keep_si(ctx), keep);
}

int check_sweep(int img_size, const double *angles, int n_angles, uint32_t flags)
{
    if (!angles || n_angles < 1)
        return fail(SID_PM_ERR_ARG, "angles must hold at least one angle (the reference's loop, pmlib.py:150, "
                                    "leaves best_result undefined for an empty list)");
    if (n_angles > sid::kMaxAngles) return fail(SID_PM_ERR_UNSUPPORTED, "more than %d angles", sid::kMaxAngles);
    if (img_size < 2 || img_size > sid::kLargeMaxSide)
        return fail(SID_PM_ERR_UNSUPPORTED, "img_size=%d: the kernels support 2..%d", img_size, sid::kLargeMaxSide);
    if (flags & ~(SID_PM_HES_NORM | SID_PM_HES_SMTH | SID_PM_MCC_NORM | SID_PM_ROT_ORDER(7))) return fail(SID_PM_ERR_ARG, "unknown flag bits");
    if (((flags >> 3) & 7u) > 5u) return fail(SID_PM_ERR_UNSUPPORTED, "rot_order %u: scipy's spline orders are 0..5", (flags >> 3) & 7u);
    return SID_PM_OK;
}

// scipy.ndimage.gaussian_filter(ccm, 1) taps (pmlib.py:46-47): exp(-k^2 / 2), k = -4..4, divided by their sum in
// NumPy's pairwise order for nine terms; host libm, as in the oracle
void gauss_taps(double w5[5])
{
    double w[9];
    for (int k = -4; k <= 4; ++k) w[k + 4] = exp(-0.5 * (double)(k * k));
    const double sum = (((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]))) + w[8];
    for (int k = 0; k < 5; ++k) w5[k] = w[k] / sum;
}

int fill_args(sid_pm_ctx *ctx, sid::PMArgs &A)
{
    memset(&A, 0, sizeof A);
    gauss_taps(A.gauss_w);
    A.img1 = ctx->cur[0].ptr; A.rows1 = ctx->cur[0].rows; A.cols1 = ctx->cur[0].cols; A.stride1 = ctx->cur[0].stride;
    A.img2 = ctx->cur[1].ptr; A.rows2 = ctx->cur[1].rows; A.cols2 = ctx->cur[1].cols; A.stride2 = ctx->cur[1].stride;
    const int64_t n = ctx->n;
    A.c1 = ctx->d_vec; A.r1 = ctx->d_vec + n; A.c2fg = ctx->d_vec + 2 * n; A.r2fg = ctx->d_vec + 3 * n;
    A.border = ctx->d_vec + 4 * n;
    A.img_size = ctx->img_size; A.n_angles = ctx->n_angles; A.flags = ctx->flags;
    A.angles = ctx->d_angles; A.rot = ctx->d_rot; A.samp = ctx->have_samp ? ctx->d_samp : nullptr; A.samp_nflag = ctx->samp_nflag; A.samp2 = ctx->have_samp ? ctx->d_samp2 : nullptr;
    A.out = ctx->user_out ? ctx->user_out : ctx->out.p;
    A.out_ij = ctx->user_out ? ctx->user_ij : ctx->out_ij.p;
    A.refused = ctx->h_refused;
    A.gsii = ctx->gsii.p; A.gsii_off = ctx->d_goff.p; A.rec = ctx->d_rec.p;
    A.gs_keep_si = gs_keep_si(ctx) ? 1u : 0u;
    A.gs_keep_acc = 0u; A.ring = nullptr; A.pool = ctx->pool.p; A.pool_stride = ctx->pool_stride;   // (per launch: sid_pm_run)
    if (getenv("SID_PM_DEBUG_CHECK")) {
        if (!ctx->dbg_err.p && ctx->dbg_err.reserve(320) == SID_PM_OK) (void)hipMemset(ctx->dbg_err.p, 0, 320 * sizeof(int32_t));
        A.dbg_err = ctx->dbg_err.p;
    }
    return SID_PM_OK;
}

// Class of one window shape: band height of its sweep items, workgroups per CU, LDS bytes (natural window pitch) and - row-pair
// kernel - whether sum w'^2 per placement lives in global memory (gs).  gs costs ~5 % where it changes nothing and buys
// -10 .. -17 % where its smaller footprint lifts the shape into the next residency class, so it is chosen exactly there;
// and wherever only a gs instantiation exists (window pitch above 112, the 8-row-band kernel, the run-time pitch).
struct ShapeClass { bool gs; int band, cls, lds, nat_pitch; bool big = false; int w3_pitch = 0; };   // w3_pitch: pitch code of the three-wavefront class (0: not in it)   // big: cls 0, a launch of its own (one workgroup per CU)
ShapeClass shape_class(bool rp, int rpp, int wh, int ww, int s, int K, uint32_t flags, bool band8_ok, bool force_gs)
{
    auto eval = [&](bool gs) {
        ShapeClass c{gs, 4, 0, lds_need(rp, rpp, wh, ww, s, K, flags, 4, 0, gs), 0};
        // the two-per-CU class runs the 8-row-band kernel (two wavefronts per SIMD leave it 256 VGPRs); its
        // window carries a few more zero rows, and a point that then no longer fits twice joins the one-per-CU
        // class (a launch of their own for the few points in between costs more than it saves)
        if (band8_ok && blocks_per_cu(c.lds) == 2) {
            const int need8 = lds_need(rp, rpp, wh, ww, s, K, flags, 8, 0, gs);
            if (blocks_per_cu(need8) >= 2) { c.lds = need8; c.band = 8; c.cls = 2; }
            else c.cls = 1;
        } else c.cls = std::max(1, std::min(max_per_cu(rp, rpp), blocks_per_cu(c.lds)));
        if (rp) c.nat_pitch = sid::rp_lds_layout(wh, ww, s, K <= sid::kRpGroup, rp_rows(rpp, c.band), 0, sid::rp_tab_pitch(rpp), rp_own_hes(K, flags), gs).wpitch;
        return c;
    };
    if (!rp) return eval(false);
    ShapeClass g = eval(true);
    // (no four-per-CU build with gs - except the three-wavefront class of the full table, for the smallest windows)
    // The three-wavefront class: window pitch 104 (borders 20 .. 23), sums in global memory, a footprint that fits four times.
    // (Pitch 112 - borders 24 .. 28 - measured: full table +-0 / +0.5 / +4 %, three angles +1.5 / +-0 %, seven angles +1 %: not
    // instantiated.  Sums in LDS - pitch code 2104, slot groups at borders 20 / 21 - won until the blocks of global memory held
    // sum w' as well; since then the gs form is 1.5 % ahead there too and the code 2104 is no longer instantiated.)
    if (K <= sid::kRpGroup && g.band == 4 && g.nat_pitch <= 104 && !force_gs) {
        const int code = 1000 + sid::rp_class_pitch(g.nat_pitch), need = lds_need(rp, rpp, wh, ww, s, K, flags, 4, code, true);
        if (need <= std::min(w3_max_lds(), 32 * 1280) && sid::rp_pitch_instantiated(4, rpp, code)) {
            ShapeClass c = g; c.gs = true; c.cls = 4; c.lds = need; c.w3_pitch = code; return c;
        }
    }
    g.cls = std::min(g.cls, 3);
    if (g.lds > sid::max_lds_bytes() && getenv("SID_PM_NO_BIG") == nullptr) {   // beyond the LDS even so: the tables go to global memory
        const sid::RpLdsLayout B = big_layout(wh, ww, s, K, flags);
        if (B.total <= sid::max_lds_bytes()) return ShapeClass{true, 4, 0, B.total, B.wpitch, true};
    }
    if (force_gs || getenv("SID_PM_ALWAYS_GS") != nullptr) return g;
    // slot groups: since their gs launches keep sum w' for the winner as well, the gs form is ahead at every border (3 / 7 angles,
    // mixed: -0.9 %; borders 25 / 26: -0.8 %; tools/archive/r4_env_ab.sh) - unless the round-3 classes were asked for (SID_PM_NO_W3)
    if (rpp > 0 && w3_max_lds() > 0 && getenv("SID_PM_NO_GS") == nullptr && getenv("SID_PM_NO_GSI") == nullptr) return g;
    const ShapeClass l = eval(false);
    if (l.band == 8 || l.nat_pitch > 112 || l.lds > sid::max_lds_bytes()) return g;   // (no instantiation without gs)
    // a class the gs footprint reaches only pays where a kernel build exists for it: the four-per-CU build (128 VGPRs) is
    // instantiated for the pitches without gs alone, so gs never buys the step from three to four
    if (getenv("SID_PM_NO_GS") == nullptr && g.cls > l.cls) return g;
    return l;
}

// Launch classes of the resident points for the pair that is current now: LDS footprint -> residency class
// (workgroups per CU) -> one launch per (class, band), XCD-aware order inside.  The footprint depends on the shape
// of image 2 (a window outside the image is a NaN point and gets the minimal footprint), so this runs in
// set_points and again in run() whenever the current pair's image 2 has another shape.  Uploads `order`.
int classify_points(sid_pm_ctx *ctx)
{
    const int64_t n = ctx->n;
    const int s = ctx->img_size, K = ctx->n_angles;
    const int64_t rows2 = ctx->cur[1].rows, cols2 = ctx->cur[1].cols;
    const double *c2fg = ctx->h_c2fg.data(), *r2fg = ctx->h_r2fg.data(), *border = ctx->h_border.data();
    static const bool no_band8 = getenv("SID_PM_NO_BAND8") != nullptr;                      // A/B runs
    const bool rp = ctx->rp;
    const int rpp = ctx->rp_paired;
    const bool band8_ok = sid::mfma_band8_supported(s) && !no_band8 && (rp ? !ctx->rp_paired : !use_paired(K));   // (classic and row-pair kernels alike)
    static const bool no_fixed_pitch = getenv("SID_PM_NO_FIXED_PITCH") != nullptr;         // (run-time pitch everywhere: gs instantiations; A/B runs)
    // Everything the launch needs to know about a point follows from the SHAPE of its search window, and a run has a few
    // dozen shapes (one per border): the LDS layouts are evaluated per shape, the points are only binned.
    struct Shape { int wh, ww, lds, band, cls, nat_pitch, pitch; double work; std::vector<int32_t> idx; bool gs = false, big = false; int w3p = 0; bool keep = false; bool large = false; bool pooled = false; };
    std::vector<Shape> shapes;
    std::vector<int32_t> slot_of((size_t)1 << 16, -1);                // (wh, ww) -> shape, direct-mapped on a hash of the pair
    auto find_shape = [&](int wh, int ww) -> int {
        uint32_t h = ((uint32_t)wh * 2654435761u ^ (uint32_t)ww * 40503u) & 0xffffu;
        for (;; h = (h + 1) & 0xffffu) {
            const int k = slot_of[h];
            if (k < 0) { slot_of[h] = (int32_t)shapes.size(); return -1 - (int)h; }
            if (shapes[(size_t)k].wh == wh && shapes[(size_t)k].ww == ww) return k;
        }
    };
    ctx->gs_keep_si = getenv("SID_PM_NO_GSI") == nullptr;
    ctx->gs_keep_acc = keep_acc_policy(rp, rpp, K);
    const uint32_t flags = ctx->flags;
    // template sides the one-point kernels do not take: every point with a valid window runs the large-window pipeline
    const bool small_ok = sid::mfma_img_size_supported(s);
    ctx->large_idx.clear(); ctx->n_nan_idx = 0;
    const int lds_min = small_ok ? lds_need(rp, rpp, s + 1, s + 1, s, K, flags, 4, 0, false) : 0;
    {   // shape 0: points whose window does not lie inside image 2 (they write NaN at once; minimal footprint)
        Shape z{0, 0, lds_min, 4, std::min(kMaxPerCu, blocks_per_cu(lds_min)), 0, 0, 0.0, {}};   // (NaN writers: any class)
        shapes.push_back(z);
    }
    double macs = 0, bytes = 0, valid = 0;
    int lds_max = lds_min;
    std::vector<int32_t> shape_of((size_t)n, 0);                      // shape of every point (0: its window does not lie inside image 2)
    for (int64_t i = 0; i < n; ++i) {
        int wh = 0, ww = 0;
        if (!window_dims(c2fg[i], r2fg[i], border[i], s, rows2, cols2, wh, ww)) { shapes[0].idx.push_back((int32_t)i); continue; }
        if (wh > 65535 || ww > 4000000) return fail(SID_PM_ERR_UNSUPPORTED, "point %lld: search window %dx%d too large (65535 rows: a launch dimension of the large-window pipeline)", (long long)i, wh, ww);
        int k = find_shape(wh, ww);
        if (k < 0) {
            Shape sh{wh, ww, 0, 4, 0, 0, 0, 0.0, {}};
            ShapeClass sc{false, 4, 1, 0, 0};
            if (small_ok) sc = shape_class(rp, rpp, wh, ww, s, K, flags, band8_ok, no_fixed_pitch);
            // a window whose tables do not fit the LDS of one workgroup: the placement space is tiled over the device instead
            if (!small_ok || sc.lds > sid::max_lds_bytes() || getenv("SID_PM_ALL_LARGE") != nullptr) { sh.large = true; sc = ShapeClass{false, 4, 1, 0, 0}; }
            sh.lds = sc.lds; sh.band = sc.band; sh.cls = sc.cls; sh.gs = sc.gs; sh.nat_pitch = sc.nat_pitch; sh.big = sc.big; sh.w3p = sc.w3_pitch;
            sh.work = (double)(wh - s + 1) * (double)(ww - s + 1);
            k = (int)shapes.size();
            shapes.push_back(sh);
        }
        Shape &sh = shapes[(size_t)k];
        if (sh.large) ctx->large_idx.push_back((int32_t)i);
        else sh.idx.push_back((int32_t)i);
        shape_of[(size_t)i] = (int32_t)k;
        macs += (double)K * sh.work * s * s;
        // window + bounding box of the rotated template + 5 inputs + outputs
        bytes += (double)wh * ww + 51.0 * 51.0 + 40.0 + 52.0;
        valid += 1;
    }
    if (!small_ok) {                                                  // no one-point kernel of this template side: the NaN rows are written by lw_write_nan
        ctx->n_nan_idx = (int)shapes[0].idx.size();
        if (int rc = ctx->d_nan_idx.reserve(std::max<size_t>(shapes[0].idx.size(), 1))) return rc;
        if (ctx->n_nan_idx) HIP_TRY(hipMemcpy(ctx->d_nan_idx.p, shapes[0].idx.data(), sizeof(int32_t) * shapes[0].idx.size(), hipMemcpyHostToDevice));
        shapes[0].idx.clear();
    }
    // launch order of the shapes: biggest footprints (fewest workgroups per CU) first, one launch per (class, band), longest first
    std::vector<int> ord;
    for (size_t k = 0; k < shapes.size(); ++k) if (!shapes[k].idx.empty()) ord.push_back((int)k);
    std::sort(ord.begin(), ord.end(), [&](int a, int b) {
        const Shape &x = shapes[(size_t)a], &y = shapes[(size_t)b];
        if (x.cls != y.cls) return x.cls < y.cls;
        if (x.band != y.band) return x.band < y.band;
        if (x.gs != y.gs) return x.gs;                                // (gs first: the larger windows of a class)
        if (x.w3p != y.w3p) return x.w3p > y.w3p;                     // (three-wavefront class: one launch per window pitch)
        if (x.work != y.work) return x.work > y.work;
        return x.idx[0] < y.idx[0];
    });
    // Row-pair kernel: one window pitch per launch, a compile-time constant of the kernel instantiation (every LDS offset of
    // the sweep's and the winner's fragment loops is then an immediate).  The pitch of a launch = the smallest instantiated
    // pitch that holds the natural pitch of all of its points; if a footprint with that pitch no longer fits its residency
    // class, the whole launch keeps the run-time pitch (SID_PM_NO_FIXED_PITCH=1: always; A/B runs).
    // gs groups take a pitch of 136 or more (or the run-time pitch): those instantiations keep sum w'^2 in global memory.
    if (rp) {
        for (size_t a = 0; a < ord.size();) {
            size_t b = a + 1;
            const Shape &first = shapes[(size_t)ord[a]];
            while (b < ord.size() && shapes[(size_t)ord[b]].cls == first.cls && shapes[(size_t)ord[b]].band == first.band &&
                   shapes[(size_t)ord[b]].gs == first.gs && shapes[(size_t)ord[b]].w3p == first.w3p) ++b;
            const bool gs = first.gs;
            int nat = gs ? 136 : 0;
            for (size_t i = a; i < b; ++i) nat = std::max(nat, shapes[(size_t)ord[i]].nat_pitch);
            int pitch = (nat > 0 && !no_fixed_pitch && !first.big) ? sid::rp_class_pitch(nat) : 0;
            if (first.w3p) pitch = first.w3p;                                  // (the three-wavefront class: its own instantiations)
            if (pitch && !sid::rp_pitch_instantiated(first.band, rpp, pitch)) pitch = 0;
            for (size_t i = a; i < b && pitch; ++i) {
                const Shape &sh = shapes[(size_t)ord[i]];
                if (sh.wh > 0 && std::min(max_per_cu(rp, rpp), blocks_per_cu(lds_need(rp, rpp, sh.wh, sh.ww, s, K, flags, sh.band, pitch, gs))) < first.cls) pitch = 0;
            }
            // (a group without gs that lost its compile-time pitch - it does not happen with the instantiated pitches - runs the
            // run-time-pitch kernel, which is a gs one)
            const bool gs_now = gs || pitch == 0;
            for (size_t i = a; i < b; ++i) {
                Shape &sh = shapes[(size_t)ord[i]];
                sh.pitch = pitch; sh.gs = gs_now;
                if (sh.wh > 0 && !sh.big) sh.lds = lds_need(rp, rpp, sh.wh, sh.ww, s, K, flags, sh.band, pitch, gs_now);
            }
            a = b;
        }
    }
    for (Shape &sh : shapes) sh.keep = rp && sh.wh > 0 && keep_acc_launch(ctx, sh.gs, sh.big, sh.cls);   // (the launch class is final here)
    // Recycled blocks (PMArgs::ring): the launches that keep accumulators always (their blocks are 3 - 5x larger); the others that
    // keep per-placement tables in global memory with SID_PM_RECYCLE_ALL != 0 (SID_PM_NO_RECYCLE=1: a block per launch position everywhere)
    {
        const bool recycle = getenv("SID_PM_NO_RECYCLE") == nullptr;
        const char *ra = getenv("SID_PM_RECYCLE_ALL");
        const bool recycle_all = ra ? atoi(ra) != 0 : kRecycleAllDefault;
        for (Shape &sh : shapes) sh.pooled = rp && sh.wh > 0 && sh.gs && !sh.big && recycle && (sh.keep || recycle_all);
    }
    // XCD-aware launch order.  Workgroup j of a launch runs on XCD j mod 8 and every XCD has its own L2, so
    // within a run of points of equal class and work (= equal border: the order there is the caller's, i.e.
    // spatial for a grid) the run is cut into 8 contiguous chunks and chunk c goes to the XCD of slot
    // (start + c) mod 8: neighbouring points - whose search windows overlap - meet in the same L2.
    // Runs keep their place in the launch (long first), so the load balance across XCDs is unchanged.
    static const bool no_xcd = getenv("SID_PM_NO_XCD_ORDER") != nullptr;                         // A/B runs
    std::vector<int32_t> order;
    order.reserve((size_t)n);
    ctx->buckets.clear();
    std::vector<int32_t> run;
    for (size_t a = 0; a < ord.size();) {
        // a run = the shapes of equal (class, band, work): their points in the caller's order
        size_t b = a + 1;
        const Shape &first = shapes[(size_t)ord[a]];
        while (b < ord.size() && shapes[(size_t)ord[b]].cls == first.cls && shapes[(size_t)ord[b]].band == first.band &&
               shapes[(size_t)ord[b]].pitch == first.pitch && shapes[(size_t)ord[b]].work == first.work) ++b;
        const std::vector<int32_t> *src = &first.idx;
        if (b - a > 1) {
            run.clear();
            for (size_t i = a; i < b; ++i) run.insert(run.end(), shapes[(size_t)ord[i]].idx.begin(), shapes[(size_t)ord[i]].idx.end());
            std::sort(run.begin(), run.end());
            src = &run;
        }
        int lds_run = 0;
        for (size_t i = a; i < b; ++i) lds_run = std::max(lds_run, shapes[(size_t)ord[i]].lds);
        lds_max = std::max(lds_max, lds_run);
        if (ctx->buckets.empty() || ctx->buckets.back().band != first.band || ctx->buckets.back().pitch != first.pitch ||
            ctx->info[5] != (double)first.cls)
            ctx->buckets.push_back(Bucket{(int)order.size(), 0, 0, first.band, first.pitch,
                                          (rp && first.cls >= 4 && sid::rp_pitch_instantiated(first.band, rpp, first.pitch, 4)) ? 4 : 3});
        ctx->info[5] = (double)first.cls;                             // (class of the bucket being filled)
        Bucket &bk = ctx->buckets.back();
        bk.big = first.big; bk.keep = first.keep; bk.pooled = first.pooled;
        bk.count += (int)src->size();
        bk.lds = std::max(bk.lds, lds_run);
        if (rp && first.gs)                                           // (launches that keep sum w'^2 in global memory: the largest block)
            for (size_t i = a; i < b; ++i) {
                const Shape &sh = shapes[(size_t)ord[i]];
                if (sh.wh > 0) bk.gs_stride = std::max<uint32_t>(bk.gs_stride, sh.big ? (uint32_t)(big_layout(sh.wh, sh.ww, s, K, flags).big_bytes / 4)
                                                                                      : block_entries(ctx, sh.wh, sh.ww, sh.band, sh.keep));
            }
        constexpr int64_t kXcd = 8;
        const int64_t L = (int64_t)src->size(), m = (L + kXcd - 1) / kXcd;
        if (!no_xcd && L >= 4 * kXcd) {
            for (int64_t p = 0; p < m; ++p)                           // slot t = p * 8 + c takes element c * m + p
                for (int64_t c = 0; c < kXcd; ++c) {
                    const int64_t e = c * m + p;
                    if (e < L) order.push_back((*src)[(size_t)e]);
                }
        } else order.insert(order.end(), src->begin(), src->end());
        a = b;
    }
    // row-pair kernel: every launch position of a launch that keeps per-placement tables in global memory (gs and big launches)
    // gets a block of its own there, in 256-byte granules: sum w'^2, then sum w' or the sweep's accumulators (block_entries);
    // big layouts: every table of the point.  Points whose launch keeps its sums in LDS get none.  Scratch per run =
    // the sum over those points: 14 KB per point at border 20 with sum w', 48 KB with the accumulators of four slot groups,
    // 0.6 MB at border 111 (sid_pm.h).
    std::vector<uint32_t> goff;
    uint64_t gsii_granules = 0;
    if (rp) {
        // Launches that keep accumulators take their blocks from the per-XCD free lists (PMArgs::ring): kPoolBlocks blocks (1024 per
        // XCD) of the largest such block, whatever the number of points (SID_PM_NO_RECYCLE=1: a block per launch position, as the others)
        uint32_t stride = 0;
        std::vector<uint32_t> gran_shape(shapes.size(), 0u);
        for (size_t k = 1; k < shapes.size(); ++k) {
            const Shape &sh = shapes[k];
            if (sh.big) gran_shape[k] = (uint32_t)(big_layout(sh.wh, sh.ww, s, K, flags).big_bytes / 256);
            else if (sh.pooled) stride = std::max(stride, block_entries(ctx, sh.wh, sh.ww, sh.band, sh.keep));
            else if (sh.gs) gran_shape[k] = block_entries(ctx, sh.wh, sh.ww, sh.band, sh.keep) / 64u;
        }
        ctx->pool_stride = stride;
        if (stride) {
            if (int rc = ctx->pool.reserve((size_t)sid::kPoolBlocks * stride)) return rc;
            if (int rc = ctx->ring.reserve((size_t)sid::kRingU64 * 2)) return rc;
            // every block free (all bits set).  Rewritten at every classification - no launch of this handle is in flight here -
            // so that an aborted run cannot leave a list short of blocks
            std::vector<uint32_t> init((size_t)sid::kRingU64 * 2, 0u);
            for (int w = 0; w < 8 * sid::kRingWordsPerXcd; ++w) { init[(size_t)w * sid::kRingStride * 2] = 0xffffffffu; init[(size_t)w * sid::kRingStride * 2 + 1] = 0xffffffffu; }
            HIP_TRY(hipMemcpyAsync(ctx->ring.p, init.data(), init.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        goff.resize(order.size());
        for (int64_t p = 0; p < (int64_t)order.size(); ++p) {
            goff[(size_t)p] = (uint32_t)gsii_granules;
            gsii_granules += gran_shape[(size_t)shape_of[(size_t)order[(size_t)p]]];
        }
        if (gsii_granules >= 0xffffffffull) return fail(SID_PM_ERR_UNSUPPORTED, "per-placement tables in global memory beyond 1 TB");
        {   // fail with the numbers before hipMalloc does
            size_t free_b = 0, total_b = 0;
            if (gsii_granules * 256ull > ctx->gsii.cap * sizeof(uint32_t) && hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
                gsii_granules * 256ull > (uint64_t)free_b + ctx->gsii.cap * sizeof(uint32_t))
                return fail(SID_PM_ERR_NOMEM, "per-placement tables of %lld points need %.1f GB of device memory, %.1f GB are free (smaller batches, or borders below 69 px)",
                            (long long)n, (double)gsii_granules * 256e-9, (double)free_b * 1e-9);
        }
        if (int rc = ctx->gsii.reserve((size_t)std::max<uint64_t>(gsii_granules, 1) * 64)) return rc;
        if (int rc = ctx->d_goff.reserve((size_t)std::max<int64_t>(n, 1))) return rc;
        if (int rc = ctx->d_rec.reserve((size_t)std::max<int64_t>(n, 1))) return rc;
    }
    std::vector<sid::PointRec> recs;
    if (rp) {
        const double *c1v = ctx->h_c1.data(), *r1v = ctx->h_r1.data();
        recs.resize(order.size());
        for (int64_t p = 0; p < (int64_t)order.size(); ++p) {
            const int32_t i = order[(size_t)p];
            recs[(size_t)p] = sid::PointRec{i, goff[(size_t)p], c1v[i], r1v[i], c2fg[i], r2fg[i], border[i]};
        }
    }
    if (!order.empty()) {                                             // (the points of the large-window pipeline are in no launch)
        HIP_TRY(hipMemcpyAsync(ctx->d_order, order.data(), sizeof(int32_t) * order.size(), hipMemcpyHostToDevice, ctx->stream));
        if (rp) HIP_TRY(hipMemcpyAsync(ctx->d_goff.p, goff.data(), sizeof(uint32_t) * order.size(), hipMemcpyHostToDevice, ctx->stream));
        if (rp) HIP_TRY(hipMemcpyAsync(ctx->d_rec.p, recs.data(), sizeof(sid::PointRec) * order.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));               // `order` and `goff` are locals
    }
    if (getenv("SID_PM_VERBOSE") != nullptr)                          // the launches of a step, one line each
        for (const Bucket &b : ctx->buckets)
            fprintf(stderr, "sid_pm: launch of %d points, %d B of LDS (%d per CU), band %d, window pitch %d, %d wavefronts per SIMD\n",
                    b.count, b.lds, std::min(b.occ, blocks_per_cu(b.lds)), b.band, b.pitch, b.occ);
    if (getenv("SID_PM_VERBOSE") != nullptr && !ctx->large_idx.empty())
        fprintf(stderr, "sid_pm: %zu points through the large-window pipeline (one point at a time, tiled over the device)\n", ctx->large_idx.size());
    ctx->cls_rows2 = rows2; ctx->cls_cols2 = cols2;
    const double img_bytes = (double)ctx->cur[0].rows * ctx->cur[0].cols + (double)rows2 * cols2;
    ctx->info[0] = (double)ctx->buckets.size();
    ctx->info[1] = valid;
    ctx->info[2] = macs;
    ctx->info[3] = std::min(bytes, img_bytes + 92.0 * valid);
    ctx->info[4] = (double)lds_max;
    ctx->info[5] = 0;
    return SID_PM_OK;
}

// rot_order 2..5 (pmlib.py:112-113): image 1 of the current pair through scipy's whole-image spline prefilter, once per pair and
// order (8 B per pixel x 2 buffers, kept with the handle); enqueued on the handle's stream behind the upload of the pair
int ensure_coef(sid_pm_ctx *ctx, int order)
{
    if (order < 2) return SID_PM_OK;
    if (ctx->coef_pair == ctx->pair_serial && ctx->coef_order == order && ctx->coef[1].p) return SID_PM_OK;
    const size_t px = (size_t)ctx->cur[0].rows * (size_t)ctx->cur[0].cols;
    if (ctx->cur[0].rows < 2 && ctx->cur[0].cols < 2) return fail(SID_PM_ERR_UNSUPPORTED, "rot_order %d on a 1 x 1 image", order);
    {
        size_t free_b = 0, total_b = 0;
        const size_t need = (ctx->coef[0].cap < px ? px * 8 : 0) + (ctx->coef[1].cap < px ? px * 8 : 0);
        if (need && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b)
            return fail(SID_PM_ERR_NOMEM, "rot_order %d: the spline coefficients of image 1 need %.1f GB of device memory, %.1f GB are free", order, (double)need * 1e-9, (double)free_b * 1e-9);
    }
    if (int rc = ctx->coef[0].reserve(px)) return rc;
    if (int rc = ctx->coef[1].reserve(px)) return rc;
    if (ctx->cur_slot >= 0 && ctx->ready_rec[ctx->cur_slot]) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->slot_ready[ctx->cur_slot], 0));
    const int e = sid::lw_spline_prefilter(ctx->cur[0].ptr, ctx->cur[0].rows, ctx->cur[0].cols, ctx->cur[0].stride, order, ctx->coef[0].p, ctx->coef[1].p, ctx->stream);
    if (e) return fail(SID_PM_ERR_HIP, "spline prefilter: %s", hipGetErrorString((hipError_t)e));
    ctx->coef_pair = ctx->pair_serial; ctx->coef_order = order;
    ctx->pre_pair = 0;                                                // (templates sampled from the old coefficients are stale)
    return SID_PM_OK;
}

// ... and the K templates of every resident point sampled from them (the one-workgroup-per-point kernels load these: PMArgs::pre)
int ensure_presampled(sid_pm_ctx *ctx, int order)
{
    if (order < 2 || ctx->n == 0) return SID_PM_OK;
    if (ctx->pre_pair == ctx->pair_serial && ctx->pre_points == ctx->points_serial && ctx->pre.p) return SID_PM_OK;
    const size_t bytes = (size_t)ctx->n * (size_t)ctx->n_angles * (size_t)ctx->img_size * (size_t)ctx->img_size;
    {
        size_t free_b = 0, total_b = 0;
        if (bytes > ctx->pre.cap && hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes > free_b)
            return fail(SID_PM_ERR_NOMEM, "rot_order %d: the pre-sampled templates of %lld points need %.1f GB of device memory, %.1f GB are free (smaller batches)",
                        order, (long long)ctx->n, (double)bytes * 1e-9, (double)free_b * 1e-9);
    }
    if (int rc = ctx->pre.reserve(bytes)) return rc;
    const int e = sid::lw_presample(ctx->coef[1].p, ctx->cur[0].rows, ctx->cur[0].cols, ctx->d_vec, ctx->d_vec + ctx->n, ctx->n, ctx->d_rot, ctx->n_angles,
                                    ctx->img_size, order, ctx->pre.p, ctx->stream);
    if (e) return fail(SID_PM_ERR_HIP, "template pre-sampling: %s", hipGetErrorString((hipError_t)e));
    ctx->pre_pair = ctx->pair_serial; ctx->pre_points = ctx->points_serial;
    return SID_PM_OK;
}

// The points of the resident set that are beyond the one-point kernels, one after the other through the large-window pipeline
// (pm_large.hip): a stream of launches, nothing read back; results straight into the rows of the run's result arrays.
int run_large_points(sid_pm_ctx *ctx, const sid::PMArgs &A)
{
    if (ctx->n_nan_idx) {
        const int e = sid::lw_write_nan(ctx->d_nan_idx.p, ctx->n_nan_idx, A.out, A.out_ij, ctx->stream);
        if (e) return fail(SID_PM_ERR_HIP, "large-window pipeline: %s", hipGetErrorString((hipError_t)e));
    }
    std::vector<sid::LargeCall> calls;
    calls.reserve(ctx->large_idx.size());
    size_t worst = 0;
    for (const int32_t i : ctx->large_idx) {
        int wh = 0, ww = 0;
        const double c2 = ctx->h_c2fg[(size_t)i], r2 = ctx->h_r2fg[(size_t)i], b = ctx->h_border[(size_t)i];
        if (!window_dims(c2, r2, b, ctx->img_size, ctx->cur[1].rows, ctx->cur[1].cols, wh, ww)) continue;   // (cannot happen: classified with this pair)
        const int hws = (int)((double)ctx->img_size / 2.0);
        sid::LargeCall c;
        c.img1 = A.img1; c.rows1 = A.rows1; c.cols1 = A.cols1; c.stride1 = A.stride1;
        c.img2 = A.img2; c.stride2 = A.stride2;
        c.win_r0 = (int64_t)(r2 - hws - b); c.win_c0 = (int64_t)(c2 - hws - b); c.wh = wh; c.ww = ww;
        c.c1 = ctx->h_c1[(size_t)i]; c.r1 = ctx->h_r1[(size_t)i];
        c.s = ctx->img_size; c.K = ctx->n_angles; c.flags = ctx->flags;
        c.d_rot = ctx->d_rot; c.d_angles = ctx->d_angles;
        c.d_coef = ((ctx->flags >> 3) & 7u) >= 2u ? ctx->coef[1].p : nullptr;
        c.add_c = c2; c.add_r = r2;
        memcpy(c.gauss_w, A.gauss_w, sizeof c.gauss_w);
        c.out5 = A.out + 5 * (size_t)i; c.ij3 = A.out_ij ? A.out_ij + 3 * (size_t)i : nullptr;
        worst = std::max(worst, sid::lw_scratch_bytes(wh, ww, c.s, c.K, c.flags));
        calls.push_back(c);
    }
    if (!calls.empty()) {
        const int e = sid::lw_run_batch(calls.data(), (int)calls.size(), ctx->lw, ctx->stream);
        if (e == -1) return fail(SID_PM_ERR_NOMEM, "the large-window pipeline could not allocate its device scratch (%.2f GB for the largest of %zu points, batches of up to 3 GB)",
                                 (double)worst * 1e-9, calls.size());
        if (e) return fail(SID_PM_ERR_HIP, "large-window pipeline: %s", hipGetErrorString((hipError_t)e));
    }
    return SID_PM_OK;
}

}  // namespace

// ------------------------------------------------------------------------------- API
SID_EXPORT int sid_pm_abi_version(void) { return SID_PM_ABI_VERSION; }

SID_EXPORT const char *sid_pm_strerror(int code)
{
    switch (code) {
    case SID_PM_OK: return "ok";
    case SID_PM_ERR_ARG: return "bad argument";
    case SID_PM_ERR_HIP: return "HIP runtime error";
    case SID_PM_ERR_NOMEM: return "out of memory";
    case SID_PM_ERR_UNSUPPORTED: return "unsupported option or size";
    case SID_PM_ERR_NODEVICE: return "no gfx950 device";
    case SID_PM_ERR_STATE: return "call order violated";
    default: return "unknown error";
    }
}

SID_EXPORT const char *sid_pm_last_error(void) { return g_err.c_str(); }

SID_EXPORT int sid_pm_device_count(int *count)
{
    if (!count) return fail(SID_PM_ERR_ARG, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(SID_PM_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_create(int device, sid_pm_ctx **out)
{
    if (!out) return fail(SID_PM_ERR_ARG, "null ctx pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(SID_PM_ERR_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(SID_PM_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SID_PM_ERR_NODEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    sid_pm_ctx *ctx = new (std::nothrow) sid_pm_ctx();
    if (!ctx) return fail(SID_PM_ERR_NOMEM, "host allocation failed");
    ctx->device = device;
    {
        Guard g(device);
        hipError_t e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
        for (int k = 0; k < 2 && e == hipSuccess; ++k) {
            e = hipEventCreateWithFlags(&ctx->slot_ready[k], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->slot_done[k], hipEventDisableTiming);
        }
        for (int k = 0; k < 3 && e == hipSuccess; ++k) {
            e = hipStreamCreateWithFlags(&ctx->side[k], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->join_ev[k], hipEventDisableTiming);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&ctx->h_refused), 4 * sizeof(int32_t), hipHostMallocMapped);
        if (e != hipSuccess) { sid_pm_destroy(ctx); return fail(SID_PM_ERR_HIP, "stream/event creation failed: %s", hipGetErrorString(e)); }
        ctx->h_refused[0] = ctx->h_refused[1] = ctx->h_refused[2] = ctx->h_refused[3] = 0;
    }
    *out = ctx;
    return SID_PM_OK;
}

SID_EXPORT void sid_pm_destroy(sid_pm_ctx *ctx)
{
    if (!ctx) return;
    Guard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    for (int k = 0; k < 2; ++k) {
        if (ctx->slot_ready[k]) (void)hipEventDestroy(ctx->slot_ready[k]);
        if (ctx->slot_done[k]) (void)hipEventDestroy(ctx->slot_done[k]);
    }
    for (int k = 0; k < 3; ++k) {
        if (ctx->side[k]) { (void)hipStreamSynchronize(ctx->side[k]); (void)hipStreamDestroy(ctx->side[k]); }
        if (ctx->join_ev[k]) (void)hipEventDestroy(ctx->join_ev[k]);
    }
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    for (auto &pair : ctx->own) for (auto &b : pair) b.release();
    ctx->arena.release();
    sid::lw_workspace_release(ctx->lw); ctx->d_nan_idx.release(); ctx->lw_small.release(); ctx->coef[0].release(); ctx->coef[1].release(); ctx->pre.release();
    ctx->out.release(); ctx->out_ij.release(); ctx->dbg_err.release(); ctx->gsii.release(); ctx->d_goff.release(); ctx->d_rec.release(); ctx->ring.release(); ctx->pool.release();
    if (ctx->h_refused) (void)hipHostFree(ctx->h_refused);
    delete ctx;
}

SID_EXPORT int sid_pm_set_stream(sid_pm_ctx *ctx, void *hip_stream)
{
    if (!ctx) return fail(SID_PM_ERR_ARG, "null ctx");
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_upload_pair(sid_pm_ctx *ctx, int slot,
                                  const uint8_t *img1, int64_t rows1, int64_t cols1, int64_t stride1,
                                  const uint8_t *img2, int64_t rows2, int64_t cols2, int64_t stride2)
{
    if (!ctx) return fail(SID_PM_ERR_ARG, "null ctx");
    if (slot < 0 || slot > 1) return fail(SID_PM_ERR_ARG, "slot must be 0 or 1");
    Image h[2] = {{img1, rows1, cols1, stride1}, {img2, rows2, cols2, stride2}};
    if (int rc = check_images(h[0], h[1])) return rc;
    Guard g(ctx->device);
    // The copy runs on the context's copy stream, so that the upload of the next pair overlaps the kernels
    // of the current one (asynchronous when the host buffers are pinned, e.g. hipHostRegister /
    // torch pin_memory; staged synchronously otherwise).  It waits for the kernels that still read this slot.
    if (ctx->done_rec[slot]) HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ctx->slot_done[slot], 0));
    for (int k = 0; k < 2; ++k) {
        // device copy is packed to a 256-byte multiple pitch (coalesced, dword-aligned rows)
        const int64_t pitch = (h[k].cols + 255) / 256 * 256;
        const size_t need = (size_t)(pitch * h[k].rows);
        if (ctx->own[slot][k].cap < need) {                // growing frees the old buffer: nothing may still read it
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
        }
        if (int rc = ctx->own[slot][k].reserve(need)) return rc;
        HIP_TRY(hipMemcpy2DAsync(ctx->own[slot][k].p, (size_t)pitch, h[k].ptr, (size_t)h[k].stride,
                                 (size_t)h[k].cols, (size_t)h[k].rows, hipMemcpyHostToDevice, ctx->copy_stream));
        ctx->slot_img[slot][k] = Image{ctx->own[slot][k].p, h[k].rows, h[k].cols, pitch};
    }
    HIP_TRY(hipEventRecord(ctx->slot_ready[slot], ctx->copy_stream));
    ctx->ready_rec[slot] = true;
    ++ctx->pair_serial;
    // first pair, refresh of the selected slot, or a borrowed binding is current (an upload replaces it: the
    // caller who streams through both slots switches with select_pair)
    if (!ctx->have_pair || ctx->cur_slot == slot || ctx->cur_slot < 0) {
        ctx->cur[0] = ctx->slot_img[slot][0]; ctx->cur[1] = ctx->slot_img[slot][1];
        ctx->have_pair = true; ctx->cur_slot = slot;
    }
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_select_pair(sid_pm_ctx *ctx, int slot)
{
    if (!ctx) return fail(SID_PM_ERR_ARG, "null ctx");
    if (slot < 0 || slot > 1 || !ctx->slot_img[slot][0].ptr) return fail(SID_PM_ERR_STATE, "slot %d holds no pair", slot);
    ctx->cur[0] = ctx->slot_img[slot][0]; ctx->cur[1] = ctx->slot_img[slot][1];
    ctx->have_pair = true; ctx->cur_slot = slot;
    ++ctx->pair_serial;
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_bind_pair(sid_pm_ctx *ctx,
                                const uint8_t *d_img1, int64_t rows1, int64_t cols1, int64_t stride1,
                                const uint8_t *d_img2, int64_t rows2, int64_t cols2, int64_t stride2)
{
    if (!ctx) return fail(SID_PM_ERR_ARG, "null ctx");
    Image d[2] = {{d_img1, rows1, cols1, stride1}, {d_img2, rows2, cols2, stride2}};
    if (int rc = check_images(d[0], d[1])) return rc;
    ctx->cur[0] = d[0]; ctx->cur[1] = d[1];
    ctx->have_pair = true; ctx->cur_slot = -1;
    ++ctx->pair_serial;
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_set_points(sid_pm_ctx *ctx, const double *c1, const double *r1, const double *c2fg,
                                 const double *r2fg, const double *border, int64_t n, int img_size,
                                 double alpha0, const double *angles, const double *rot, int n_angles,
                                 uint32_t flags)
{
    if (!ctx) return fail(SID_PM_ERR_ARG, "null ctx");
    if (n < 0 || n > 0x7fffffff / 8) return fail(SID_PM_ERR_ARG, "bad point count");
    if (n > 0 && (!c1 || !r1 || !c2fg || !r2fg || !border)) return fail(SID_PM_ERR_ARG, "null point vector");
    if (int rc = check_sweep(img_size, angles, n_angles, flags)) return rc;
    if (!ctx->have_pair) return fail(SID_PM_ERR_STATE, "set_points needs an image pair (upload_pair/bind_pair first)");
    Guard g(ctx->device);
    const int s = img_size, K = n_angles;

    (void)alpha0;
    std::vector<double> rotv;
    if (int rc = make_rot(K, rot, rotv)) return rc;
    std::vector<uint16_t> sampv;
    int nflag = 0;
    // (rot_order = 1: no offset table - every template sample is interpolated in float64 by the general sampler)
    if (!getenv("SID_PM_NO_SAMP_TABLE") && ((flags >> 3) & 7u) == 0u && sid::mfma_img_size_supported(s)) nflag = make_samp(rotv, K, s, sampv);
    std::vector<uint32_t> samp2v;
    // (measured +2 % on the 15-angle step - fifteen table loads per angle instead of five, a uniform branch per chunk - although
    // it executes a third fewer VALU instructions in the template phase: built on request only, SID_PM_SAMP2=1)
    if (!sampv.empty() && use_rp(s, K) && getenv("SID_PM_SAMP2") != nullptr) make_samp2(sampv, K, s, samp2v);

    // one arena, one upload: [5n doubles | K angles | 4K rotation terms | order (int32 n) | sampling table]
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_vec = 0, o_ang = up(o_vec + sizeof(double) * 5 * (size_t)n), o_rot = up(o_ang + sizeof(double) * (size_t)K),
                 o_ord = up(o_rot + sizeof(double) * 4 * (size_t)K), o_smp = up(o_ord + sizeof(int32_t) * (size_t)n),
                 o_smp2 = up(o_smp + sizeof(uint16_t) * (sampv.size() + 4)), total = up(o_smp2 + sizeof(uint32_t) * samp2v.size());
    HIP_TRY(hipStreamSynchronize(ctx->stream));               // nothing may still read the old arena
    if (int rc = ctx->arena.reserve(total)) return rc;
    if (int rc = ctx->out.reserve((size_t)(5 * n))) return rc;
    if (int rc = ctx->out_ij.reserve((size_t)(3 * n))) return rc;
    std::vector<uint8_t> host(total, 0);
    const double *src[5] = {c1, r1, c2fg, r2fg, border};
    for (int k = 0; k < 5 && n > 0; ++k) memcpy(host.data() + o_vec + sizeof(double) * (size_t)(k * n), src[k], sizeof(double) * (size_t)n);
    memcpy(host.data() + o_ang, angles, sizeof(double) * (size_t)K);
    memcpy(host.data() + o_rot, rotv.data(), sizeof(double) * rotv.size());
    if (!sampv.empty()) memcpy(host.data() + o_smp, sampv.data(), sizeof(uint16_t) * sampv.size());
    if (!samp2v.empty()) memcpy(host.data() + o_smp2, samp2v.data(), sizeof(uint32_t) * samp2v.size());
    // synchronous: the host vectors are caller-owned and not retained
    HIP_TRY(hipMemcpy(ctx->arena.p, host.data(), total, hipMemcpyHostToDevice));
    ctx->d_vec = reinterpret_cast<double *>(ctx->arena.p + o_vec);
    ctx->d_angles = reinterpret_cast<double *>(ctx->arena.p + o_ang);
    ctx->d_rot = reinterpret_cast<double *>(ctx->arena.p + o_rot);
    ctx->d_order = reinterpret_cast<int32_t *>(ctx->arena.p + o_ord);
    ctx->d_samp = reinterpret_cast<uint16_t *>(ctx->arena.p + o_smp);
    ctx->d_samp2 = samp2v.empty() ? nullptr : reinterpret_cast<uint32_t *>(ctx->arena.p + o_smp2);
    ctx->have_samp = !sampv.empty(); ctx->samp_nflag = nflag;
    ctx->h_c2fg.assign(c2fg, c2fg + n); ctx->h_r2fg.assign(r2fg, r2fg + n); ctx->h_border.assign(border, border + n);
    ctx->h_c1.assign(c1, c1 + n); ctx->h_r1.assign(r1, r1 + n);

    ctx->user_out = nullptr; ctx->user_ij = nullptr;
    ctx->n = n; ctx->img_size = s; ctx->n_angles = K; ctx->flags = flags; ctx->rp = use_rp(s, K); ctx->rp_paired = ctx->rp ? rp_paired(K) : 0;
    ctx->have_points = false;
    ++ctx->points_serial;
    if (int rc = classify_points(ctx)) return rc;
    ctx->have_points = true;
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_bind_results(sid_pm_ctx *ctx, double *d_out, int32_t *d_out_ij)
{
    if (!ctx) return fail(SID_PM_ERR_ARG, "null ctx");
    if (!ctx->have_points) return fail(SID_PM_ERR_STATE, "bind_results before set_points");
    if (!d_out && d_out_ij) return fail(SID_PM_ERR_ARG, "d_out_ij without d_out");
    ctx->user_out = d_out; ctx->user_ij = d_out_ij;
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_run(sid_pm_ctx *ctx)
{
    if (!ctx) return fail(SID_PM_ERR_ARG, "null ctx");
    if (!ctx->have_points || !ctx->have_pair) return fail(SID_PM_ERR_STATE, "run needs set_points and an image pair");
    Guard g(ctx->device);
    // the launch classes were sized for the image-2 shape current at set_points; a pair of another shape
    // (select_pair / bind_pair / upload_pair since then) gets its own classification before anything is launched
    if (ctx->cur[1].rows != ctx->cls_rows2 || ctx->cur[1].cols != ctx->cls_cols2) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (int rc = classify_points(ctx)) return rc;
    }
    sid::PMArgs A;
    fill_args(ctx, A);
    if (ctx->cur_slot >= 0 && ctx->ready_rec[ctx->cur_slot])
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->slot_ready[ctx->cur_slot], 0));
    {   // rot_order 2..5: spline coefficients of image 1 (once per pair), templates of the points sampled from them (once per pair and point set)
        const int order = (int)((ctx->flags >> 3) & 7u);
        if (int rc = ensure_coef(ctx, order)) return rc;
        if (!ctx->buckets.empty()) { if (int rc = ensure_presampled(ctx, order)) return rc; }
        A.pre = (order >= 2 && !ctx->buckets.empty()) ? ctx->pre.p : nullptr;
    }
    // Launches side by side for SHORT runs.  A full grid is dozens of rounds of workgroups per launch and its launches run one
    // after the other (side by side they measured 1-6 % slower: a CU that took a workgroup of a large-window class is lost to
    // several small ones, rounds 2-4).  A rank's shard of an 8-GPU run is two or three launches of one to seven rounds each,
    // and every launch ends with a half-empty round - a quarter of such a run; side by side the next launch's workgroups fill
    // the CUs the previous one drains.  Rule: all launches of a run together are at most kSideRounds rounds of workgroups
    // (SID_PM_SIDE_BY_SIDE=0 / 1: never / always; A/B runs).  The launches keep their order (largest footprint first).
    bool side_by_side = false;
    if (ctx->buckets.size() > 1) {
        double rounds = 0;
        for (const Bucket &b : ctx->buckets) rounds += (double)b.count / (256.0 * std::max(1, std::min(b.band == 8 ? 2 : 4, blocks_per_cu(b.lds))));
        side_by_side = side_by_side_rule(ctx->buckets.size(), rounds);
    }
    // (SID_PM_SIDE_FIRST=n, experiments: only the first n launches of a long run side by side - the large-window classes, a
    // round or two of workgroups each - then the rest in sequence)
    int n_side = side_by_side ? (int)ctx->buckets.size() : 0;
    if (!side_by_side) if (const char *e = getenv("SID_PM_SIDE_FIRST")) n_side = std::min((int)ctx->buckets.size(), std::max(0, atoi(e)));
    if (n_side > 1) HIP_TRY(hipEventRecord(ctx->fork_ev, ctx->stream));
    int nb = 0;
    unsigned used = 0;
    auto join = [&]() -> int {
        for (int k = 0; k < 3; ++k)                                  // the handle's stream continues when every side launch is done
            if (used & (1u << k)) {
                HIP_TRY(hipEventRecord(ctx->join_ev[k], ctx->side[k]));
                HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->join_ev[k], 0));
            }
        used = 0;
        return SID_PM_OK;
    };
    for (const Bucket &b : ctx->buckets) {
        // launch k of a side-by-side run: the handle's stream, then the side streams in turn
        hipStream_t st = ctx->stream;
        if (nb == n_side && n_side > 1) { if (int rc = join()) return rc; }
        if (nb > 0 && nb < n_side) {
            const int k = (nb - 1) % 3;
            st = ctx->side[k];
            if (!(used & (1u << k))) HIP_TRY(hipStreamWaitEvent(st, ctx->fork_ev, 0));
            used |= 1u << k;
        }
        ++nb;
        A.order = ctx->d_order + b.offset;
        A.gsii_off = ctx->d_goff.p ? ctx->d_goff.p + b.offset : nullptr;
        A.rec = ctx->d_rec.p ? ctx->d_rec.p + b.offset : nullptr;
        A.n_launch = b.count;
        A.gs_keep_acc = b.keep ? 1u : 0u;
        A.ring = (b.pooled && ctx->pool_stride) ? ctx->ring.p : nullptr;
        const int lds_launch = std::min(b.lds, sid::max_lds_bytes());
        A.lds_bytes = lds_launch;
        // 256 threads per point; 768 when the LDS footprint leaves room for one point per CU only, so that
        // the CU still carries 12 wavefronts (3 per SIMD = the register budget).  (Measured for the
        // two-per-CU class, border 28: 256 threads 5.8 ms, 384: 9.1, 512: 7.8, 768: 7.8 - six wavefronts per group
        // land 2/2/1/1 on the SIMDs, a second group of 168-VGPR wavefronts then no longer fits.)
        const int per_cu = std::max(1, std::min(b.band == 8 ? 2 : 8, blocks_per_cu(b.lds)));
        const int nthreads = b.band == 8 ? 256 : (per_cu == 1 ? 768 : (ctx->rp && sid::rp_pitch_is_w3(b.pitch)) ? 192 : 256);
        const int e = ctx->rp
                          ? sid::launch_pm_rp(A, lds_launch, nthreads, b.band, b.big ? 0 : ctx->rp_paired, b.pitch, b.occ, st, b.big)
                          : sid::launch_pm_mfma(A, lds_launch, nthreads, b.band, use_paired(ctx->n_angles), st);
        if (e != 0) {
            (void)join();                                            // (what the side streams already hold still orders before the handle's stream)
            return fail(SID_PM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
        }
    }
    if (int rc = join()) return rc;
    if (int rc = run_large_points(ctx, A)) return rc;
    if (ctx->cur_slot >= 0) {
        HIP_TRY(hipEventRecord(ctx->slot_done[ctx->cur_slot], ctx->stream));
        ctx->done_rec[ctx->cur_slot] = true;
    }
    return SID_PM_OK;
}

// Valid points the kernels refused since the last check (the launch their classification put them in could not hold
// their LDS layout: host and device disagree - a bug, reported as an error instead of the NaN row the point received).
// Reads a pinned host word: call it once the launch stream is synchronised, by whatever means.
SID_EXPORT int sid_pm_check(sid_pm_ctx *ctx)
{
    if (!ctx) return fail(SID_PM_ERR_ARG, "null ctx");
    if (!ctx->h_refused) return SID_PM_OK;
    const int32_t n = __atomic_exchange_n(ctx->h_refused, 0, __ATOMIC_ACQ_REL);
    const int32_t polls = __atomic_exchange_n(ctx->h_refused + 1, 0, __ATOMIC_ACQ_REL);
    if (polls > 0 && getenv("SID_PM_VERBOSE") != nullptr) fprintf(stderr, "sid_pm: %d workgroup(s) found the home words of their XCD's free list empty and took a block of its reserve words\n", (int)polls);
    if (n != 0)
        return fail(SID_PM_ERR_STATE, "%d grid point(s) with a valid search window were refused by their launch (LDS layout of the "
                                      "kernel and classification of the host disagree, or all %d recycled blocks of global memory of an XCD "
                                      "were taken - %d pops went to the reserve words); their rows hold NaN", (int)n, sid::kRingWordsPerXcd * 64, (int)polls);
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_sync(sid_pm_ctx *ctx)
{
    if (!ctx) return fail(SID_PM_ERR_ARG, "null ctx");
    Guard g(ctx->device);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return sid_pm_check(ctx);
}

SID_EXPORT int sid_pm_fetch(sid_pm_ctx *ctx, double *out, int32_t *out_ij)
{
    if (!ctx) return fail(SID_PM_ERR_ARG, "null ctx");
    if (!ctx->have_points) return fail(SID_PM_ERR_STATE, "fetch before set_points");
    if (ctx->n > 0 && !out) return fail(SID_PM_ERR_ARG, "null out");
    Guard g(ctx->device);
    if (ctx->n > 0) {
        const double *src = ctx->user_out ? ctx->user_out : ctx->out.p;
        const int32_t *src_ij = ctx->user_out ? ctx->user_ij : ctx->out_ij.p;
        HIP_TRY(hipMemcpyAsync(out, src, sizeof(double) * 5 * (size_t)ctx->n, hipMemcpyDeviceToHost, ctx->stream));
        if (out_ij && src_ij)
            HIP_TRY(hipMemcpyAsync(out_ij, src_ij, sizeof(int32_t) * 3 * (size_t)ctx->n, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->dbg_err.p) {
        int32_t e[320];
        if (hipMemcpy(e, ctx->dbg_err.p, sizeof e, hipMemcpyDeviceToHost) == hipSuccess && (e[0] || e[1] || e[2] || e[3])) {
            if (e[4] > 0) {
                fprintf(stderr, "first event, (i,j got want):");
                for (int k = 0; k < e[4] && k < 200; ++k) fprintf(stderr, " (%d,%d %d %d)", (e[64 + k] >> 24) & 255, (e[64 + k] >> 16) & 255, (e[64 + k] >> 8) & 255, e[64 + k] & 255);
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "SID_PM_DEBUG_CHECK: sum mismatches after sampling %d, after sweep %d; table bytes != exact resampling %d (pt %d slot %d n %d); patch bytes != image %d;", e[0], e[1], e[2], e[40], e[41], e[42], e[3]);
            for (int t = 0; t < 2; ++t) for (int k = 0; k < 4 && k < e[t]; ++k)
                fprintf(stderr, " [tag %d pt %d slot %d dST %d dSTT %d]", t, e[8 + 16 * t + 4 * k], e[9 + 16 * t + 4 * k], e[10 + 16 * t + 4 * k], e[11 + 16 * t + 4 * k]);
            fprintf(stderr, "\n");
            (void)hipMemset(ctx->dbg_err.p, 0, sizeof e);
        }
    }
    return sid_pm_check(ctx);
}

// ---- exchange step of the N-GPU path: the gathered per-rank blocks -> original point order, in ONE pass ----
// stack: `world` blocks of m rows [m x 5 float64 | m x 3 int32] (what every rank's kernels wrote, gathered by RCCL);
// perm[i] = row of point i in the stacked blocks.  One thread per (point, field): eight threads move the 52 bytes of a
// point, so the writes of a wavefront are 8 x 40 + 8 x 12 contiguous bytes.  `out` / `out_ij` may be pinned host memory:
// the results then reach the host without a copy after the kernel (posted PCIe writes).
namespace {
__global__ void unpermute_rows_kernel(const uint8_t *stack, int64_t m, const int32_t *perm, int64_t n, double *out, int32_t *out_ij)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t >> 3;
    const int f = (int)(t & 7);
    if (i >= n) return;
    const int64_t row = perm[i], blk = row / m, r = row - blk * m;
    const uint8_t *base = stack + blk * m * 52;
    if (f < 5) out[i * 5 + f] = reinterpret_cast<const double *>(base)[r * 5 + f];
    else if (out_ij) out_ij[i * 3 + (f - 5)] = reinterpret_cast<const int32_t *>(base + m * 40)[r * 3 + (f - 5)];
}
}  // namespace

SID_EXPORT int sid_pm_unpermute(const void *d_stack, int64_t world, int64_t m, const int32_t *d_perm, int64_t n,
                                double *out, int32_t *out_ij, void *hip_stream)
{
    if (n < 0 || world < 1 || m < 1 || (m & 1) || (n > 0 && (!d_stack || !d_perm || !out)))
        return fail(SID_PM_ERR_ARG, "unpermute: bad argument (rows per block must be even: the blocks then keep their doubles aligned)");
    if (n == 0) return SID_PM_OK;
    const int64_t threads = n * 8;
    hipLaunchKernelGGL(unpermute_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream),
                       static_cast<const uint8_t *>(d_stack), m, d_perm, n, out, out_ij);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "unpermute: %s", hipGetErrorString(e));
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_device_results(sid_pm_ctx *ctx, double **d_out, int32_t **d_out_ij)
{
    if (!ctx) return fail(SID_PM_ERR_ARG, "null ctx");
    if (!ctx->have_points) return fail(SID_PM_ERR_STATE, "device_results before set_points");
    if (d_out) *d_out = ctx->user_out ? ctx->user_out : ctx->out.p;
    if (d_out_ij) *d_out_ij = ctx->user_out ? ctx->user_ij : ctx->out_ij.p;
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_work_info(sid_pm_ctx *ctx, double info[6])
{
    if (!ctx || !info) return fail(SID_PM_ERR_ARG, "null argument");
    if (!ctx->have_points) return fail(SID_PM_ERR_STATE, "work_info before set_points");
    memcpy(info, ctx->info, sizeof ctx->info);
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_batch(const uint8_t *img1, int64_t rows1, int64_t cols1, int64_t stride1,
                            const uint8_t *img2, int64_t rows2, int64_t cols2, int64_t stride2,
                            const double *c1, const double *r1, const double *c2fg, const double *r2fg,
                            const double *border, int64_t n, int img_size, double alpha0,
                            const double *angles, const double *rot, int n_angles, uint32_t flags,
                            double *out, int32_t *out_ij)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(SID_PM_ERR_NODEVICE, "no current HIP device");
    sid_pm_ctx *ctx = nullptr;
    int rc = sid_pm_create(dev, &ctx);
    if (rc) return rc;
    rc = sid_pm_upload_pair(ctx, 0, img1, rows1, cols1, stride1, img2, rows2, cols2, stride2);
    if (!rc) rc = sid_pm_set_points(ctx, c1, r1, c2fg, r2fg, border, n, img_size, alpha0, angles, rot, n_angles, flags);
    if (!rc) rc = sid_pm_run(ctx);
    if (!rc) rc = sid_pm_fetch(ctx, out, out_ij);
    const std::string keep = g_err;
    sid_pm_destroy(ctx);
    if (rc) g_err = keep;
    return rc;
}

// ---- rotate_and_match / get_template / get_hessian as calls of their own (reference pmlib.py:117-174, :89-115, :36-59) ----
SID_EXPORT int sid_pm_rotate_and_match(sid_pm_ctx *ctx, double c1, double r1, int img_size,
                                       int64_t win_row0, int64_t win_col0, int64_t win_rows, int64_t win_cols,
                                       double alpha0, const double *angles, const double *rot, int n_angles, uint32_t flags,
                                       double out5[5], int32_t ij3[3], float *ccm, int64_t ccm_cap, uint8_t *best_template)
{
    (void)alpha0;
    if (!ctx || !out5) return fail(SID_PM_ERR_ARG, "null argument");
    if (!ctx->have_pair) return fail(SID_PM_ERR_STATE, "rotate_and_match needs an image pair (upload_pair / bind_pair first)");
    if (int rc = check_sweep(img_size, angles, n_angles, flags)) return rc;
    if (!rot) return fail(SID_PM_ERR_ARG, "rot is required: the rotation terms as the caller's NumPy computed them (pmlib.py:105-110)");
    const int s = img_size, K = n_angles;
    if (win_row0 < 0 || win_col0 < 0 || win_rows < 1 || win_cols < 1 || win_row0 + win_rows > ctx->cur[1].rows || win_col0 + win_cols > ctx->cur[1].cols)
        return fail(SID_PM_ERR_ARG, "the window does not lie inside image 2");
    // cv2.matchTemplate needs a window at least as large as the template, np.gradient two values along each axis (pmlib.py:156, :51)
    if (win_rows - s + 1 < 2 || win_cols - s + 1 < 2)
        return fail(SID_PM_ERR_ARG, "window %lldx%lld: fewer than two placements of a %d px template along an axis", (long long)win_rows, (long long)win_cols, s);
    if (win_rows > 65535 || win_cols > 4000000)
        return fail(SID_PM_ERR_UNSUPPORTED, "window %lldx%lld too large (at most 65535 rows: a launch dimension of the large-window pipeline)", (long long)win_rows, (long long)win_cols);
    const int64_t np = (win_rows - s + 1) * (win_cols - s + 1);
    if (np >= 0xffffffffll) return fail(SID_PM_ERR_UNSUPPORTED, "more than 2^32 placements");
    if (ccm && ccm_cap < np) return fail(SID_PM_ERR_ARG, "ccm capacity %lld < %lld placements", (long long)ccm_cap, (long long)np);
    Guard g(ctx->device);
    // small device block: [K angles | 4K rotation terms | 5 results | 3 int32]
    if (int rc = ctx->lw_small.reserve((size_t)K * 5 + 8)) return rc;
    std::vector<double> host((size_t)K * 5);
    memcpy(host.data(), angles, sizeof(double) * (size_t)K);
    memcpy(host.data() + K, rot, sizeof(double) * 4 * (size_t)K);
    HIP_TRY(hipMemcpyAsync(ctx->lw_small.p, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));                       // (`host` is a local)
    if (ctx->cur_slot >= 0 && ctx->ready_rec[ctx->cur_slot]) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->slot_ready[ctx->cur_slot], 0));
    sid::LargeCall c;
    c.img1 = ctx->cur[0].ptr; c.rows1 = ctx->cur[0].rows; c.cols1 = ctx->cur[0].cols; c.stride1 = ctx->cur[0].stride;
    c.img2 = ctx->cur[1].ptr; c.stride2 = ctx->cur[1].stride;
    c.win_r0 = win_row0; c.win_c0 = win_col0; c.wh = (int)win_rows; c.ww = (int)win_cols;
    c.c1 = c1; c.r1 = r1; c.s = s; c.K = K; c.flags = flags;
    if (int rc = ensure_coef(ctx, (int)((flags >> 3) & 7u))) return rc;
    c.d_coef = ((flags >> 3) & 7u) >= 2u ? ctx->coef[1].p : nullptr;
    c.d_angles = ctx->lw_small.p; c.d_rot = ctx->lw_small.p + K;
    c.add_c = 0.0; c.add_r = 0.0;
    gauss_taps(c.gauss_w);
    c.out5 = ctx->lw_small.p + 5 * (size_t)K;
    c.ij3 = reinterpret_cast<int32_t *>(ctx->lw_small.p + 5 * (size_t)K + 5);
    const int e = sid::lw_run(c, ctx->lw, ctx->stream);
    if (e == -1) return fail(SID_PM_ERR_NOMEM, "rotate_and_match needs %.2f GB of device scratch for a %dx%d window and %d angles",
                             (double)sid::lw_scratch_bytes(c.wh, c.ww, s, K, flags) * 1e-9, c.wh, c.ww, K);
    if (e) return fail(SID_PM_ERR_HIP, "large-window pipeline: %s", hipGetErrorString((hipError_t)e));
    if (ctx->cur_slot >= 0) { HIP_TRY(hipEventRecord(ctx->slot_done[ctx->cur_slot], ctx->stream)); ctx->done_rec[ctx->cur_slot] = true; }
    int32_t ij[3] = {-1, -1, -1};
    HIP_TRY(hipMemcpyAsync(out5, c.out5, sizeof(double) * 5, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ij, c.ij3, sizeof ij, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ij3) memcpy(ij3, ij, sizeof ij);
    if (ij[2] >= 0) {                                                 // (a NaN point has no matrix and no template: pmlib.py:152-154)
        if (ccm) HIP_TRY(hipMemcpy(ccm, sid::lw_ncc_matrix(ctx->lw, c.wh, c.ww, s, ij[2]), sizeof(float) * (size_t)np, hipMemcpyDeviceToHost));
        if (best_template) HIP_TRY(hipMemcpy(best_template, sid::lw_template(ctx->lw, s, ij[2]), (size_t)s * s, hipMemcpyDeviceToHost));
    }
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_get_template(int device, const uint8_t *img, int64_t rows, int64_t cols, int64_t stride, double c, double r,
                                   const double rot4[4], int img_size, int rot_order, uint8_t *out)
{
    if (!img || !rot4 || !out || rows < 1 || cols < 1 || stride < cols) return fail(SID_PM_ERR_ARG, "bad argument");
    if (img_size < 1 || img_size > 4096) return fail(SID_PM_ERR_UNSUPPORTED, "img_size=%d", img_size);
    if (rot_order < 0 || rot_order > 5) return fail(SID_PM_ERR_UNSUPPORTED, "rot_order=%d: scipy's spline orders are 0..5", rot_order);
    if (!(fabs(c) < 1e15 && fabs(r) < 1e15)) return fail(SID_PM_ERR_ARG, "non-finite centre");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(SID_PM_ERR_NODEVICE, "no such device");
    Guard g(device);
    const int s = img_size;
    // only the part of the image the samples can touch travels to the device: the samples lie within hypot(s, s) + |tcT| of (r, c)
    const double reach = 1.5 * (double)s + fabs(rot4[2]) + fabs(rot4[3]) + 4.0;
    // (orders 2..5: scipy prefilters the WHOLE image - pmlib.py:112-113 - so the whole image travels)
    const bool whole = rot_order >= 2;
    const int64_t row0 = whole ? 0 : std::max<int64_t>(0, (int64_t)floor(r - reach)), row1 = whole ? rows : std::min<int64_t>(rows, (int64_t)ceil(r + reach) + 1);
    const int64_t col0 = whole ? 0 : std::max<int64_t>(0, (int64_t)floor(c - reach)), col1 = whole ? cols : std::min<int64_t>(cols, (int64_t)ceil(c + reach) + 1);
    DevBuf<uint8_t> dimg, dout;
    DevBuf<double> drot, dcoef0, dcoef1;
    int rc = SID_PM_OK;
    const int64_t nr = std::max<int64_t>(row1 - row0, 0), nc = std::max<int64_t>(col1 - col0, 0);
    if ((rc = dimg.reserve((size_t)std::max<int64_t>(nr * nc, 1))) || (rc = dout.reserve((size_t)s * s)) || (rc = drot.reserve(4)) ||
        (whole && ((rc = dcoef0.reserve((size_t)(rows * cols))) || (rc = dcoef1.reserve((size_t)(rows * cols)))))) {
        dimg.release(); dout.release(); drot.release(); dcoef0.release(); dcoef1.release(); return rc;
    }
    hipError_t e = hipSuccess;
    if (nr > 0 && nc > 0) e = hipMemcpy2D(dimg.p, (size_t)nc, img + row0 * stride + col0, (size_t)stride, (size_t)nc, (size_t)nr, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(drot.p, rot4, sizeof(double) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && whole) e = (hipError_t)sid::lw_spline_prefilter(dimg.p, rows, cols, nc, rot_order, dcoef0.p, dcoef1.p, nullptr);
    // (a template wholly outside the image samples nothing: every coordinate fails the bounds test and yields 0)
    if (e == hipSuccess) e = (hipError_t)sid::lw_get_template(dimg.p, nc > 0 ? nc : 1, row0, col0, nr, nc, rows, cols, c, r, drot.p, s, rot_order, dout.p, nullptr, whole ? dcoef1.p : nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, dout.p, (size_t)s * s, hipMemcpyDeviceToHost);
    dimg.release(); dout.release(); drot.release(); dcoef0.release(); dcoef1.release();
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "get_template: %s", hipGetErrorString(e));
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_get_hessian(int device, const float *ccm, int64_t rows, int64_t cols, uint32_t flags, float *hes)
{
    if (!ccm || !hes) return fail(SID_PM_ERR_ARG, "null argument");
    if (flags & ~(SID_PM_HES_NORM | SID_PM_HES_SMTH)) return fail(SID_PM_ERR_ARG, "get_hessian takes SID_PM_HES_NORM and SID_PM_HES_SMTH");
    // np.gradient needs two values along each axis (pmlib.py:51)
    if (rows < 2 || cols < 2 || rows * cols >= 0xffffffffll) return fail(SID_PM_ERR_ARG, "matrix %lldx%lld: at least 2x2, fewer than 2^32 values", (long long)rows, (long long)cols);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(SID_PM_ERR_NODEVICE, "no such device");
    Guard g(device);
    const size_t n = (size_t)(rows * cols);
    DevBuf<float> din, dout;
    sid::LwWorkspace W;
    int rc = SID_PM_OK;
    if ((rc = din.reserve(n)) || (rc = dout.reserve(n))) { din.release(); dout.release(); return rc; }
    double gw[5];
    gauss_taps(gw);
    hipError_t e = hipMemcpy(din.p, ccm, sizeof(float) * n, hipMemcpyHostToDevice);
    int le = 0;
    if (e == hipSuccess) le = sid::lw_get_hessian(din.p, (int)rows, (int)cols, flags, gw, dout.p, W, nullptr);
    if (le == 0 && e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (le == 0 && e == hipSuccess) e = hipMemcpy(hes, dout.p, sizeof(float) * n, hipMemcpyDeviceToHost);
    din.release(); dout.release(); sid::lw_workspace_release(W);
    if (le == -1) return fail(SID_PM_ERR_NOMEM, "get_hessian: device scratch");
    if (le) return fail(SID_PM_ERR_HIP, "get_hessian: %s", hipGetErrorString((hipError_t)le));
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "get_hessian: %s", hipGetErrorString(e));
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_debug_point(sid_pm_ctx *ctx, double c1, double r1, double c2fg, double r2fg,
                                  double border, int img_size, double alpha0, const double *angles,
                                  const double *rot, int n_angles, uint32_t flags,
                                  uint8_t *templates, float *ccm, float *hes, int64_t cap, int32_t rh_rw[2],
                                  double out5[5], int32_t ij3[3], int64_t phase_cycles[32])
{
    if (!ctx) return fail(SID_PM_ERR_ARG, "null ctx");
    if (!ctx->have_pair) return fail(SID_PM_ERR_STATE, "debug_point needs an image pair");
    if (int rc = check_sweep(img_size, angles, n_angles, flags)) return rc;
    Guard g(ctx->device);
    const int s = img_size, K = n_angles;
    const bool rp = use_rp(s, K);
    const int rpp = rp ? rp_paired(K) : 0;
    int wh = 0, ww = 0, lds = lds_need(rp, rpp, s + 1, s + 1, s, K, flags);
    if (window_dims(c2fg, r2fg, border, s, ctx->cur[1].rows, ctx->cur[1].cols, wh, ww))
        lds = lds_need(rp, rpp, wh, ww, s, K, flags);
    if (lds > sid::max_lds_bytes()) return fail(SID_PM_ERR_UNSUPPORTED, "search window too large for LDS");
    (void)alpha0;
    if (!sid::mfma_img_size_supported(s)) return fail(SID_PM_ERR_UNSUPPORTED, "debug_point: template sides 2..64 (sid_pm_rotate_and_match returns the matrix and the template of any size)");
    std::vector<double> rotv;
    if (int rc = make_rot(K, rot, rotv)) return rc;

    DevBuf<double> dv, dang, drot, dout;
    DevBuf<int32_t> dord, dij, dshape;
    DevBuf<uint8_t> dt;
    DevBuf<uint16_t> dsamp;
    std::vector<uint16_t> sampv;
    int nflag = 0;
    // (rot_order = 1: no offset table - every template sample is interpolated in float64 by the general sampler)
    if (!getenv("SID_PM_NO_SAMP_TABLE") && ((flags >> 3) & 7u) == 0u) nflag = make_samp(rotv, K, s, sampv);
    DevBuf<float> dccm, dhes;
    DevBuf<uint8_t> dpre;                                              // rot_order 2..5: this point's templates, sampled from the spline coefficients
    DevBuf<long long> dcyc;
    DevBuf<uint32_t> dgs;                                              // row-pair kernel: this point's sum w'^2 block + its offset (0)
    DevBuf<sid::PointRec> drec;                                        // ... and its record
    int rc = SID_PM_OK;
    auto cleanup = [&]() { dv.release(); dang.release(); drot.release(); dout.release(); dord.release(); dgs.release(); drec.release();
                           dij.release(); dshape.release(); dt.release(); dccm.release(); dhes.release(); dcyc.release(); dsamp.release(); dpre.release(); };
    const size_t tcount = (size_t)K * s * s;
    if ((rc = dv.reserve(5)) || (rc = dang.reserve((size_t)K)) || (rc = drot.reserve(4 * (size_t)K)) ||
        (rc = dout.reserve(5)) || (rc = dord.reserve(1)) || (rc = dij.reserve(3)) || (rc = dshape.reserve(2)) ||
        (rc = dt.reserve(tcount)) || (rc = dccm.reserve((size_t)std::max<int64_t>(cap, 1))) ||
        (rc = dhes.reserve((size_t)std::max<int64_t>(cap, 1))) || (rc = dcyc.reserve(32)) ||
        (rc = dsamp.reserve(sampv.size() + 4)) || (rc = dgs.reserve(64 + (size_t)(wh > s ? sid::rp_block_entries(wh - s + 1, ww - s + 1, rp_rows(rpp, 4), 16 >> rpp, false, keep_acc_policy(rp, rpp, K)) : 64))) || (rc = drec.reserve(1))) { cleanup(); return rc; }
    const double v5[5] = {c1, r1, c2fg, r2fg, border};
    const int32_t zero = 0, shape0[2] = {0, 0};
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    step(hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream) step(hipStreamSynchronize(ctx->copy_stream));   // a pending upload of the selected pair
    step(hipMemcpy(dv.p, v5, sizeof v5, hipMemcpyHostToDevice));
    step(hipMemcpy(dang.p, angles, sizeof(double) * K, hipMemcpyHostToDevice));
    step(hipMemcpy(drot.p, rotv.data(), sizeof(double) * rotv.size(), hipMemcpyHostToDevice));
    step(hipMemcpy(dord.p, &zero, sizeof zero, hipMemcpyHostToDevice));
    step(hipMemset(dgs.p, 0, 64 * sizeof(uint32_t)));                  // entry 0 = the offset (0 granules); the block starts at entry 64
    {
        const sid::PointRec r1rec{0, 0u, c1, r1, c2fg, r2fg, border};
        step(hipMemcpy(drec.p, &r1rec, sizeof r1rec, hipMemcpyHostToDevice));
    }
    if (!sampv.empty()) step(hipMemcpy(dsamp.p, sampv.data(), sizeof(uint16_t) * sampv.size(), hipMemcpyHostToDevice));
    step(hipMemcpy(dshape.p, shape0, sizeof shape0, hipMemcpyHostToDevice));
    step(hipMemset(dt.p, 0, tcount));
    step(hipMemset(dcyc.p, 0, sizeof(long long) * 32));
    if (cap > 0) { step(hipMemset(dccm.p, 0, sizeof(float) * cap)); step(hipMemset(dhes.p, 0, sizeof(float) * cap)); }
    if (e == hipSuccess) {
        sid::PMArgs A;
        memset(&A, 0, sizeof A);
        A.img1 = ctx->cur[0].ptr; A.rows1 = ctx->cur[0].rows; A.cols1 = ctx->cur[0].cols; A.stride1 = ctx->cur[0].stride;
        A.img2 = ctx->cur[1].ptr; A.rows2 = ctx->cur[1].rows; A.cols2 = ctx->cur[1].cols; A.stride2 = ctx->cur[1].stride;
        A.c1 = dv.p; A.r1 = dv.p + 1; A.c2fg = dv.p + 2; A.r2fg = dv.p + 3; A.border = dv.p + 4;
        A.order = dord.p; A.n_launch = 1; A.img_size = s; A.n_angles = K; A.flags = flags;
        A.angles = dang.p; A.rot = drot.p; A.out = dout.p; A.out_ij = dij.p;
        A.dbg_templates = dt.p; A.dbg_ccm = dccm.p; A.dbg_hes = dhes.p; A.dbg_shape = dshape.p; A.dbg_cap = cap;
        A.dbg_cycles = dcyc.p;
        gauss_taps(A.gauss_w);
        A.samp = sampv.empty() ? nullptr : dsamp.p; A.samp_nflag = nflag;
        A.lds_bytes = lds;
        const int order = (int)((flags >> 3) & 7u);
        if (order >= 2) {
            if (int rc2 = ensure_coef(ctx, order)) { cleanup(); return rc2; }
            if (int rc2 = dpre.reserve(tcount)) { cleanup(); return rc2; }
            step((hipError_t)sid::lw_presample(ctx->coef[1].p, ctx->cur[0].rows, ctx->cur[0].cols, dv.p, dv.p + 1, 1, drot.p, K, s, order, dpre.p, ctx->stream));
            A.pre = dpre.p;
        }
        A.gsii = dgs.p + 64; A.gsii_off = dgs.p; A.rec = drec.p;
        A.gs_keep_acc = keep_acc_policy(rp, rpp, K) ? 1u : 0u;
        step((hipError_t)(rp ? sid::launch_pm_rp(A, lds, 256, 4, rpp, 0, 3, ctx->stream)
                                       : sid::launch_pm_mfma(A, lds, 256, 4, use_paired(K), ctx->stream)));
        step(hipStreamSynchronize(ctx->stream));
        if (templates) step(hipMemcpy(templates, dt.p, tcount, hipMemcpyDeviceToHost));
        if (ccm && cap > 0) step(hipMemcpy(ccm, dccm.p, sizeof(float) * cap, hipMemcpyDeviceToHost));
        if (hes && cap > 0) step(hipMemcpy(hes, dhes.p, sizeof(float) * cap, hipMemcpyDeviceToHost));
        if (rh_rw) step(hipMemcpy(rh_rw, dshape.p, sizeof(int32_t) * 2, hipMemcpyDeviceToHost));
        if (out5) step(hipMemcpy(out5, dout.p, sizeof(double) * 5, hipMemcpyDeviceToHost));
        if (ij3) step(hipMemcpy(ij3, dij.p, sizeof(int32_t) * 3, hipMemcpyDeviceToHost));
        if (phase_cycles) step(hipMemcpy(phase_cycles, dcyc.p, sizeof(long long) * 32, hipMemcpyDeviceToHost));
    }
    cleanup();
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "debug_point: %s", hipGetErrorString(e));
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_debug_rsqrt(sid_pm_ctx *ctx, const double *x, double *y, int64_t n)
{
    if (!ctx || !x || !y || n < 0) return fail(SID_PM_ERR_ARG, "bad argument");
    Guard g(ctx->device);
    DevBuf<double> dx, dy;
    int rc;
    if ((rc = dx.reserve((size_t)n)) || (rc = dy.reserve((size_t)n))) { dx.release(); dy.release(); return rc; }
    hipError_t e = hipMemcpy(dx.p, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = (hipError_t)sid::launch_rsqrt(dx.p, dy.p, n, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(y, dy.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost);
    dx.release(); dy.release();
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "debug_rsqrt: %s", hipGetErrorString(e));
    return SID_PM_OK;
}


SID_EXPORT int sid_pm_debug_ncc_selftest(sid_pm_ctx *ctx, uint64_t seed, int64_t evaluations, int img_size, uint64_t counts[3])
{
    if (!ctx || !counts || evaluations <= 0 || img_size < 2 || img_size > 64) return fail(SID_PM_ERR_ARG, "bad argument");
    Guard g(ctx->device);
    DevBuf<unsigned long long> d;
    int rc;
    if ((rc = d.reserve(3))) return rc;
    const int per_thread = 256;
    const int blocks = (int)std::min<int64_t>((evaluations + 256 * per_thread - 1) / (256 * per_thread), 1 << 20);
    hipError_t e = hipMemsetAsync(d.p, 0, 3 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) e = (hipError_t)sid::launch_ncc_selftest(seed, blocks, per_thread, img_size, d.p, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d.p, sizeof(h), hipMemcpyDeviceToHost);
    d.release();
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "debug_ncc_selftest: %s", hipGetErrorString(e));
    for (int k = 0; k < 3; ++k) counts[k] = h[k];
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_debug_hypot_selftest(sid_pm_ctx *ctx, uint64_t seed, int64_t evaluations, uint64_t counts[2])
{
    if (!ctx || !counts || evaluations <= 0) return fail(SID_PM_ERR_ARG, "bad argument");
    Guard g(ctx->device);
    DevBuf<unsigned long long> d;
    int rc;
    if ((rc = d.reserve(2))) return rc;
    const int per_thread = 256;
    const int blocks = (int)std::min<int64_t>((evaluations + 256 * per_thread - 1) / (256 * per_thread), 1 << 20);
    hipError_t e = hipMemsetAsync(d.p, 0, 2 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) e = (hipError_t)sid::launch_hypot_selftest(seed, blocks, per_thread, d.p, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    unsigned long long h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d.p, sizeof(h), hipMemcpyDeviceToHost);
    d.release();
    if (e != hipSuccess) return fail(SID_PM_ERR_HIP, "debug_hypot_selftest: %s", hipGetErrorString(e));
    counts[0] = h[0]; counts[1] = h[1];
    return SID_PM_OK;
}

// Estimated cost of a grid point in nanoseconds of one MI355X, for cutting a set of points into shards of equal cost
// (sea_ice_drift_amd/dist.py).  Not a table per border: the cost follows what the kernel executes for the point - the
// matrix instructions of the sweep (bands x placement tiles x template row pairs, per group of angles), those of the
// winner's NCC matrix, the placements themselves - times a factor for the residency class of its LDS footprint (fewer
// co-resident workgroups hide less latency).  The six constants were fitted to tools/border_cost.py (template side 34, 15
// angles, borders 20..50: within 4 % everywhere); what matters to the sharding are the ratios between points.
static int estimate_points(const double *border, int64_t n, int img_size, int n_angles, uint32_t flags, double *cost_ns, int32_t *per_cu_out)
{
    if (n < 0 || (n > 0 && (!border || (!cost_ns && !per_cu_out)))) return fail(SID_PM_ERR_ARG, "bad argument");
    const int s = img_size, K = n_angles;
    if (s < 2 || s > sid::kLargeMaxSide || K < 1) return fail(SID_PM_ERR_UNSUPPORTED, "img_size / angle count not supported");
    const bool small_ok = sid::mfma_img_size_supported(s);
    // a point of the large-window pipeline (pm_large.hip: batches of up to 64 points, each tiled over the device): the matrix
    // instructions of lw_corr and the per-placement passes, fitted to profiles/r06_large_window_bench_batched.jsonl (11 / 19 / 46 us
    // at borders 112 / 160 / 250 with 15 angles, 9 us at 100 px, 6 us at 65 px: within 40 %); the ~25 launches of a batch
    // (0.15 ms) are priced per batch by sid_pm_estimate_run_time
    auto large_cost = [&](int wn) {
        const double r = (double)(wn - s + 1);
        const double tiles = ceil(r / 64.0) * ceil(r / 16.0) * (double)((K + 15) / 16);
        return 3000.0 + 4.5 * tiles * (double)(s + 3) * (double)((s + 63) / 64);
    };
    const bool rp = use_rp(s, K);
    const int rpp = rp ? rp_paired(K) : 0;
    const int hws = (int)((double)s / 2.0);
    const int groups = (K + sid::kRpGroup - 1) / sid::kRpGroup;
    // round 4 (transposed strip columns, sum w'^2 in global memory where it lifts the residency class): refitted to
    // tools/border_cost.py on the shipped library, one set per kernel family - full table / two / four slot groups - within
    // 2.3 / 4.3 / 7 % of the measured staircase (profiles/r04_border_cost.json)
    struct Fit { double sweep, winner, pos, fixed, two, one, gs, four; };
    static const Fit kFit[3] = {{3.4646e-3, 2.2848e-2, 7.6082e-3, 29.955, 1.1228, 1.2800, 1.06500, 0.88700},
                                {3.9853e-3, 1.7996e-2, 7.6297e-3, 25.686, 1.2348, 1.3287, 0.99236, 0.91000},
                                {2.5923e-3, 1.8513e-2, 6.4734e-3, 24.173, 1.2518, 1.3574, 1.00890, 0.86000}};   // (last column: the three-wavefront class, refitted to tools/archive/r4_w3.sh / r4_w3p.sh)
    const Fit &F = kFit[rpp];
    constexpr double kBigFactor = 1.25;   // every per-placement table through L2 / HBM (measured at borders 70 .. 100: tools/border_cost.py)
    const double kSweep = F.sweep, kWinner = F.winner, kPos = F.pos, kFixed = F.fixed, kTwoPerCu = F.two, kOnePerCu = F.one, kFourPerCu = F.four;
    for (int64_t i = 0; i < n; ++i) {
        const double b = border[i];
        if (per_cu_out) per_cu_out[i] = kMaxPerCu;
        if (!(b >= 0.0 && b < 4096.0)) { if (cost_ns) cost_ns[i] = kFixed; continue; }      // NaN / absurd: a point that writes NaN at once
        const int wn = 2 * hws + 2 * (int)b + 1, r = wn - s + 1;
        if (r < 2) { if (cost_ns) cost_ns[i] = kFixed; continue; }
        double sweep, winner, cls_factor;
        int cls = kMaxPerCu;
        if (!small_ok) {
            if (cost_ns) cost_ns[i] = large_cost(wn);
            if (per_cu_out) per_cu_out[i] = 1 + SID_PM_CLASS_LARGE;
            continue;
        }
        if (rp) {
            static const bool no_band8 = getenv("SID_PM_NO_BAND8") != nullptr;
            const ShapeClass sc = shape_class(rp, rpp, wn, wn, s, K, flags, sid::mfma_band8_supported(s) && !no_band8 && !rpp, false);   // (the flags decide the Hessian's LDS: rp_own_hes)
            if (sc.lds > sid::max_lds_bytes()) {                               // beyond the LDS of one workgroup: the large-window pipeline
                if (cost_ns) cost_ns[i] = large_cost(wn);
                if (per_cu_out) per_cu_out[i] = 1 + SID_PM_CLASS_LARGE;
                continue;
            }
            const sid::RpLdsLayout L4 = sid::rp_lds_layout(wn, wn, s, K <= sid::kRpGroup, rp_rows(rpp, 4), 0, sid::rp_tab_pitch(rpp), rp_own_hes(K, flags), sc.gs);
            const int per_cu = std::max(1, sc.cls), band = sc.band;            // (big layouts - class 0 - run one workgroup per CU, full table)
            const int rows = sc.big ? 4 : rp_rows(rpp, band), nb = (r + rows - 1) / rows, tiles = 2 * L4.npair + L4.nsingle;
            const double per_row_tile = (double)((s + 1) / 2 + s / 2 + 1) / 2.0 + (double)(((s - 32 + 1) / 2) * 2);
            // work items are dealt to the wavefronts of the workgroup: the busiest wavefront sets the pace
            const int nwaves = per_cu == 1 ? 12 : 4;
            const int units = ((nb * tiles + nwaves - 1) / nwaves) * nwaves, wunits = ((((r + 15) / 16) * tiles + nwaves - 1) / nwaves) * nwaves;
            sweep = groups * (sc.big ? 1.0 : rpp == 2 ? 0.33 : rpp == 1 ? 0.55 : 1.0) * units * (rows * per_row_tile + 2.0);
            winner = wunits * 76.0;
            cls_factor = ((per_cu >= 4 && max_per_cu(rp, rpp) == 4) ? kFourPerCu : per_cu >= 3 ? 1.0 : per_cu == 2 ? kTwoPerCu : kOnePerCu) * (sc.gs ? F.gs : 1.0) * (sc.big ? kBigFactor : 1.0);
            cls = std::max(1, std::min(per_cu, max_per_cu(rp, rpp))) + (sc.gs ? 16 : 0) + (sc.big ? 32 : 0) + (sc.w3_pitch > 1104 ? 64 : 0);   // (+ 64: the second launch of the three-wavefront class)   // (+ 16: the launches that keep sum w'^2 in global memory are launches of their own)
        } else {
            const sid::MfmaLdsLayout L = sid::mfma_lds_layout(wn, wn, s, 4, use_paired(K));
            if (L.total > sid::max_lds_bytes()) {
                if (cost_ns) cost_ns[i] = large_cost(wn);
                if (per_cu_out) per_cu_out[i] = 1 + SID_PM_CLASS_LARGE;
                continue;
            }
            const int per_cu = blocks_per_cu(L.total), ntx = (r + 15) / 16;
            sweep = groups * (double)r * ntx * s * (use_paired(K) ? 0.55 : 1.0) * 1.3;    // one template row per MFMA
            winner = (double)((r + 15) / 16) * ntx * (16 + s - 1) * 2.0;
            cls_factor = per_cu >= 3 ? 1.0 : per_cu == 2 ? kTwoPerCu : kOnePerCu;
            cls = std::max(1, std::min(per_cu, kMaxPerCu));
        }
        if (cost_ns) cost_ns[i] = (kSweep * sweep + kWinner * winner + kPos * (double)r * r + kFixed) * cls_factor;
        if (per_cu_out) per_cu_out[i] = cls;
    }
    return SID_PM_OK;
}

SID_EXPORT int sid_pm_estimate_cost(const double *border, int64_t n, int img_size, int n_angles, uint32_t flags, double *cost_ns)
{
    if (n > 0 && !cost_ns) return fail(SID_PM_ERR_ARG, "bad argument");
    return estimate_points(border, n, img_size, n_angles, flags, cost_ns, nullptr);
}

// Launch class of a point of that border (include/sid_pm.h): workgroups per CU in the low four bits - a launch runs 256 x that
// many points at a time, which is what the tail of a SHORT launch costs (dist.py) - plus the SID_PM_CLASS_* bits that tell the
// launches of equal residency apart (points of equal value share a launch).
SID_EXPORT int sid_pm_estimate_residency(const double *border, int64_t n, int img_size, int n_angles, uint32_t flags, int32_t *per_cu)
{
    if (n > 0 && !per_cu) return fail(SID_PM_ERR_ARG, "bad argument");
    return estimate_points(border, n, img_size, n_angles, flags, nullptr, per_cu);
}

// Estimated kernel time of ONE run over these points (nanoseconds): per launch class the sum of the point costs plus the tail of
// the launch - with 256 x (workgroups per CU) points in flight the last round is half empty on average; the classes with few
// slots run the large borders, whose run times differ by up to 1.5x inside one launch, and end less evenly (fitted to the
// shards tools/shard_sim.py measures: 1.0 / 0.7 / 0.5 / 0.5 rounds for 1 / 2 / 3 / 4 workgroups per CU) - a launch shorter than
// one round still takes a full one; a SHORT run (side_by_side_rule, the launcher's own) ends with ONE tail, the longest, and
// SID_DIST_TAIL_BLEND (0.7) of the others; the points of the large-window pipeline run one after the other behind the launches.
SID_EXPORT int sid_pm_estimate_run_time(const double *border, int64_t n, int img_size, int n_angles, uint32_t flags, double *time_ns)
{
    if (!time_ns || n < 0 || (n > 0 && !border)) return fail(SID_PM_ERR_ARG, "bad argument");
    *time_ns = 0.0;
    if (n == 0) return SID_PM_OK;
    std::vector<double> cost((size_t)n);
    std::vector<int32_t> cls((size_t)n);
    if (int rc = estimate_points(border, n, img_size, n_angles, flags, cost.data(), cls.data())) return rc;
    struct Part { double sum = 0, count = 0; };
    Part parts[256];
    for (int64_t i = 0; i < n; ++i) { Part &p = parts[cls[(size_t)i] & 255]; p.sum += cost[(size_t)i]; p.count += 1; }
    static const double kTail[5] = {0.5, 1.0, 0.7, 0.5, 0.5};
    const char *be = getenv("SID_DIST_TAIL_BLEND");
    const double blend = be ? atof(be) : 0.7;
    double large = 0, sum_all = 0, rounds = 0, tail_max = 0, tail_sum = 0, lat_max = 0, seq = 0;
    size_t nparts = 0;
    for (int c = 0; c < 256; ++c) {
        const Part &p = parts[c];
        if (p.count == 0) continue;
        if (c & SID_PM_CLASS_LARGE) { large += p.sum + 150000.0 * ceil(p.count / 64.0); continue; }   // (+ the launches of every batch of 64)
        const int per_cu = std::max(1, c & SID_PM_CLASS_PER_CU);
        const double latency = 256.0 * per_cu * (p.sum / p.count), tail = kTail[std::min(per_cu, 4)] * latency;
        ++nparts; sum_all += p.sum; rounds += p.count / (256.0 * per_cu);
        tail_max = std::max(tail_max, tail); tail_sum += tail; lat_max = std::max(lat_max, latency);
        seq += std::max(p.sum + tail, latency);
    }
    double t = seq;
    if (side_by_side_rule(nparts, rounds)) t = std::max(sum_all + tail_max + blend * (tail_sum - tail_max), lat_max);
    *time_ns = t + large;
    return SID_PM_OK;
}
