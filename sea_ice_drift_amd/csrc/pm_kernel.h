// Shared between the HIP kernel (pm_kernel_mfma.hip) and the host C-ABI (pm_capi.hip).
#pragma once
#include <stdint.h>

namespace sid {

constexpr int kMaxAngles = 64;

// Row-pair kernel: everything a workgroup needs to know about its point in ONE record per launch position - the point's index,
// the offset of its sum w'^2 block (PMArgs::gsii, units of 64 entries) and its five input values - so that the prologue
// waits for one memory round trip (launch position -> record) instead of two (position -> index -> five vectors).
struct PointRec { int32_t pt; uint32_t gsii_off; double c1, r1, c2fg, r2fg, border; };

// Arguments of one launch (one block per grid point of this launch).
struct PMArgs {
    const uint8_t *img1; int64_t rows1, cols1, stride1;
    const uint8_t *img2; int64_t rows2, cols2, stride2;
    const double *c1, *r1, *c2fg, *r2fg, *border;   // [n_total], original point order
    const int32_t *order;                           // [n_launch] -> original point index
    int32_t n_launch;
    int32_t lds_bytes;                              // dynamic LDS of this launch: a point whose layout needs more writes NaN
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
    // round 5, row-pair kernel: the same table SORTED per angle (make_samp2 in pm_capi.hip; null: not built).  A quad of four
    // template samples whose patch bytes are CONSECUTIVE - three quarters of them at +-7 degrees: a nearest-neighbour rotation by a
    // few degrees copies runs of a patch row - is one record {aligned patch offset, byte shift, row, quad} and costs two aligned
    // dwords + one v_alignbyte instead of four byte gathers; the rest (a run ends inside the quad, the tail quad of a row, flagged
    // entries) keep their four offsets.  Per angle kSamp2Words u32: [n_run, n_gather, -, -][kSamp2Cap run records]
    // [kSamp2Cap x uint2 gather offsets][kSamp2Cap x u16 unit numbers i * nq + jq].
    const uint32_t *samp2;
    const uint8_t *pre;                             // rot_order 2..5: the templates of every point [n_total][K][s][s], sampled from the spline coefficients of image 1 (lw_presample); null otherwise
    double gauss_w[5];                              // hes_smth: normalised sigma-1 Gaussian taps w[0] (|k| = 4) .. w[4] (centre), host-computed
    double *out;                                    // [n_total][5]
    int32_t *out_ij;                                // [n_total][3]
    // points with a valid search window that the kernel refused because the launch did not fit them (LDS size or window
    // pitch chosen by the host's classification): counted here - pinned host memory, read by sid_pm_sync / sid_pm_fetch /
    // sid_pm_check - so that a disagreement between host and device layout is an error, never a silent NaN
    int32_t *refused;                               // [0] refused points; [1] pops that went to the spare blocks (diagnostic)
    // row-pair kernel: sum w'^2 per placement of every point in global memory instead of LDS (7 .. 42 KB per point: the
    // largest LDS item of a point with a large search border, written once and read twice).  gsii_off[j] = offset of launch
    // position j's block in units of 64 entries (256 bytes); the block is written and read by the point's own workgroup
    // only and stays in L2 in between.
    uint32_t *gsii;
    const uint32_t *gsii_off;
    const PointRec *rec;                            // [n_launch] row-pair kernel: one record per launch position (above)
    uint32_t gs_keep_si;                            // blocks of global memory hold 2 x round_up(placements, 64) entries: sum w'^2, then sum w' kept by the sweep for the winner
    // round 5: the launch keeps the sweep's ACCUMULATORS - the exact integer S_IT' of every slot at every placement, sum w' in the
    // ones slot - in the point's block of global memory, behind sum w'^2: [round_up(placements, 64) u32 | rows_pad x cols_pad
    // placements x (slots of one group) i32] (rp_acc_dims).  The NCC matrix of the winning angle is then a normalisation of
    // stored sums (rp_winner_kept): no second pass over the window, no operand staging, no matrix instructions.  16 B per
    // placement with four slot groups (the reference's default: 3 angles), 32 B with two (<= 7 angles), 64 B with the full table.
    uint32_t gs_keep_acc;
    // Recycled blocks: the blocks of global memory of a launch that keeps accumulators come from FREE LISTS per XCD instead of
    // one block per launch position - 1024 - 2048 blocks cycle through L2 / the Infinity Cache instead of 40 000 write-once blocks
    // travelling to HBM.  A block belongs to whoever popped it until that workgroup pushes it back: ownership is a property of the
    // list, not of where the workgroup happens to run (tools/ubench/slot_life.hip: under a queue eviction workgroups are saved and
    // restored on OTHER CUs - round 4's pool picked by hardware slot broke there).
    // Round 6: the lists are BITMAPS - kRingWordsPerXcd 64-bit words per XCD, a set bit = a free block - and no operation ever
    // waits for another workgroup: pop = one atomic AND that clears the lowest set bit of the popper's home word (found in a
    // copy read first thing in the kernel; the AND's return value tells whether the bit was still there, and is the fresh copy
    // for the next try if not), then the XCD's other words, then - all of them empty: eight times what an XCD holds resident
    // would have to be in flight - the point is refused (PMArgs::refused: an error, never a hang); push = one atomic OR without
    // a return value.  Round 5's ticket ring committed a popper to ONE future push: a workgroup that was context-saved (queue
    // eviction) between taking its ticket and reading its entry found the entry overwritten by the pushes of a whole lap when it
    // came back and waited for ever - the eviction soak's "timeouts" (reproduced in round 6 with a bounded wait:
    // profiles/r06_ring_deadlock.txt).  Lock-free stacks (64-bit compare-and-swap, a `next` word per block) fixed that and cost
    // +12-15 % on the reference's defaults - five serial round trips per point at agent scope (profiles/r06_ab_free_lists.txt).
    // `ring` = [8 XCDs][kRingWordsPerXcd] words, 64 B apart; block number = word index * 64 + bit; null: the exclusive block of
    // the launch position (gsii_off).
    uint32_t *ring;
    uint32_t *pool;                                 // [kPoolBlocks][pool_stride u32]
    uint32_t pool_stride;
    // diagnostics (debug_point only; null in production launches)
    uint8_t *dbg_templates; float *dbg_ccm; float *dbg_hes; int32_t *dbg_shape; int64_t dbg_cap;
    long long *dbg_cycles;                          // [32] shader-clock stamps at phase boundaries
    int32_t *dbg_err;                               // [64] consistency-check counters of debugging builds (SID_DBG_CHECK); null otherwise
};

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- MFMA kernel (pm_kernel_mfma.hip) ----
constexpr int kMiscMfmaBytes = 2688;
constexpr int kTrowPad = 36;         // zero rows around a winner operand block: 16 above, 20 below (the step loop runs in fours)
constexpr int kQueueCap = 192;       // arg-max candidates waiting for exact evaluation (16 B each)
constexpr int kRingWordsPerXcd = 16, kRingHomeWords = 4;    // bitmap words per XCD (PMArgs::ring); a workgroup's home word is one of the first kRingHomeWords, by launch position
constexpr int kPoolBlocks = 8 * kRingWordsPerXcd * 64;     // 1024 blocks per XCD: eight times what an XCD holds resident (workgroups that a queue eviction
                                                           // saved hold theirs while others start: four queues - a run's launches side by side - can strand 4 x 128)
constexpr int kRingStride = 8;                             // u64 units between words (a cache line each)
constexpr int kRingU64 = 8 * kRingWordsPerXcd * kRingStride;
constexpr int kRpQueueWithSi = 128;  // ... and the sums kept for the winner (RpLdsLayout::si_off) leave it at least this
constexpr int kRpQueueMin = 64;      // row-pair kernel: the queue shrinks to this where it decides the residency class (overflow is evaluated in place)

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

// ---- row-pair kernel (pm_kernel_rp.inc): template sides 34 and 35, more than kPairedMaxAngles angles ----
// Sweep operands: the template columns 0..31 live in a table [row -1 .. s+1][16 slots][32 B] (zero rows around),
// from which one MFMA operand = two consecutive template rows x 32 columns (K = 64 all useful); the columns
// 32.. ("strip") are multiplied through a TRANSPOSED copy of the window columns 32.., WT[v][rho] = W[rho][32 + v],
// so that K = 2 window columns x 32 rows and a lane's operand is 16 contiguous bytes of one WT row (round 4: half the
// LDS of the column-pair copy of rounds 2-3, which moves borders 24-26 to three and 37-41 to two workgroups per CU).
constexpr int kRpGroup = 15;    // angles per group of MFMA slots (the 16th slot is the all-ones template)
struct RpLdsLayout {
    int wpitch, wrows;          // window pitch (multiple of 8) and rows written (window + zero rows)
    int win_off, sii_off;       // sii_off: sum w'^2 per placement (u32) in LDS; 0 when it lives in global memory (gs layouts, PMArgs::gsii)
    int si_off;                 // sum w' per placement (i32) kept from the sweep for the winner's matrix, where the residency class has room for it (else 0)
    int hes_off;                // Hessian magnitudes (f32 per placement): over the dead window + winner operands, or behind the histogram
    int ccm_off;                // NCC matrix of the winning angle (f32 per placement), behind the winner's operands
    int wp_off, wp_pitch, wp_rows, wp_len;   // transposed window columns 32..: wp_rows rows (window column 32 + row) of wp_len bytes (one per window row)
    int u_off;                  // union: column sums | table + strip + patch + queue | winner operands + NCC matrix
    int tab_rows, tab_pitch;    // s + 3 rows of tab_pitch bytes (32 per stored slot)
    int strip_off, ncp, nrg;    // strip operand fragments: ncp column pairs x nrg row groups, 1 KB each
    int patch_off, ppitch, pdim, pradius, queue_off, trow_bytes;
    int queue_cap;              // candidate-queue entries (16 B each): kRpQueueMin .. kQueueCap, as many as the residency class leaves room for
    int npair, nsingle;         // x tiling of a band: npair items of 32 placements, then nsingle (0/1) of 16
    int big_bytes;              // big layouts (below): bytes of the point's block of global memory; 0 otherwise
    int total;
};

__host__ __device__ inline bool rp_size_supported(int s) { return s == 34 || s == 35; }

// one_group: all angles fit one group of MFMA slots (K <= 15).  The winner then takes its template from the sweep's
// operand table, the image-1 patch is dead once the table is built and the candidate queue lies over it.
// first output row of band `b` of `nbands` (multiple of 4: the column-pair reads are 8-byte aligned)
__host__ __device__ inline int rp_band_y0(int b, int nbands, int rh, int band)
{
    int y0 = b * band;
    if (band >= 8 && b == nbands - 1 && nbands > 1) {
        const int up = round_up(rh - band, 4);
        if (up < y0) y0 = up;
    }
    return y0;
}

// band: output rows per sweep work item (4, or 8 for the two-workgroups-per-CU class).
// force_pitch: window pitch fixed by the launch class (a compile-time constant of the kernel instantiation: rp_class_pitch);
// 0 = the natural pitch.  A forced pitch below the natural one is ignored (the kernel then refuses the point).
// largest LDS footprint with the same number of workgroups per CU as `lds_bytes` (1280-byte granules of the 160 KB: 32
// granules each for four - slot-group layouts, whose 128-VGPR build allows them -, 42 for three, 64 for two)
__host__ __device__ constexpr int rp_class_limit(int lds_bytes)
{
    return lds_bytes <= 32 * 1280 ? 32 * 1280 : lds_bytes <= 42 * 1280 ? 42 * 1280 : lds_bytes <= 64 * 1280 ? 64 * 1280 : 128 * 1280;
}
// tab_pitch: bytes per operand-table row = 32 x (slots that hold distinct operands): 512, or rp_tab_pitch(paired) with slot
// groups - the lanes of the later groups read the first group's operands at a lagging address, so only those are stored
__host__ __device__ constexpr int rp_tab_pitch(int paired) { return paired == 2 ? 128 : paired == 1 ? 256 : 512; }
__host__ __device__ constexpr int rp_tab_shift(int paired) { return paired == 2 ? 7 : paired == 1 ? 8 : 9; }
// gs: sum w'^2 per placement lives in GLOBAL memory (PMArgs::gsii) instead of LDS - 7 .. 42 KB less per point, which lifts
// most borders above 27 into the next residency class (-10 .. -17 % there), at the price of ~5 % where it does not (measured
// at border 20: the values are written once and read twice through L2).  The host chooses per launch (pm_capi.hip
// classify_points); the kernel instantiations with a window pitch of 136 or more, or the run-time pitch, are the gs ones
// (rp_pitch_is_gs).  Without gs the Hessian magnitudes take the LDS of the sums, as in rounds 2-3.
// own_hes (gs only): the Hessian magnitudes get LDS of their own behind the winner's NCC matrix (the general ph_hessian -
// hes_smth / mcc_norm, several groups of angles - keeps its histograms in the winner's operand block); otherwise they lie
// over the window and the winner's operands, both dead by then (the NCC matrix moves up where those are too short).
// Pitch CODES from 1000 up name the instantiations of the three-wavefront class (round 4: 192 threads per point, four points
// per CU with the 168-VGPR build): window pitch = code % 1000; 1000 + pitch keeps the sums in global memory, 2000 + pitch in LDS.
__host__ __device__ constexpr int rp_pitch_bytes(int pitch) { return pitch % 1000; }
__host__ __device__ constexpr bool rp_pitch_is_w3(int pitch) { return pitch >= 1000; }
__host__ __device__ constexpr bool rp_pitch_is_gs(int pitch) { return pitch == 0 || rp_pitch_bytes(pitch) >= 136 || (pitch >= 1000 && pitch < 2000); }
// big (gs only; search borders 69 .. ~100 at s = 34, whose per-placement tables no longer fit the 160 KB): the row sums of
// rp_sums, the NCC matrix of the winning angle and the Hessian magnitudes live in the point's block of GLOBAL memory as well
// - [sum w'^2 | row sums, later the NCC matrix | Hessian magnitudes], 256-byte aligned; ccm_off / hes_off are byte offsets
// into that block - and LDS holds the window, the sweep's operands and the winner's operands + histogram only.
__host__ __device__ inline RpLdsLayout rp_lds_layout(int wh, int ww, int s, bool one_group, int band = 4, int force_pitch = 0,
                                                     int tab_pitch = 512, bool own_hes = false, bool gs = true, bool big = false)
{
    RpLdsLayout L;
    L.big_bytes = 0;
    const int rh = wh - s + 1, rw = ww - s + 1;
    const int nE = (s + 1) / 2, nO = s / 2 + 1, nst = (nE > nO ? nE : nO) + band / 2 - 1;
    const int rem = rw % 32;
    L.npair = rw / 32 + (rem > 16 ? 1 : 0);
    L.nsingle = (rem > 0 && rem <= 16) ? 1 : 0;
    // bytes a window row must hold: a pair item at x0 reads 24 B from x0 + 8 (n >> 2) + 16 (g & 1) (<= x0 + 64);
    // a single item reads 5 dwords from (x0 + n + 16 (g & 1)) & ~3 (<= x0 + 51)
    int need = ww;
    // (+ 16 / + 16: the winner's strip operands read the same runs 32 columns further right)
    if (L.npair && 32 * (L.npair - 1) + 80 > need) need = 32 * (L.npair - 1) + 80;
    if (L.nsingle && 32 * L.npair + 68 > need) need = 32 * L.npair + 68;
    L.wpitch = round_up(need, 8);
    if (force_pitch > L.wpitch) L.wpitch = force_pitch;
    // first row of the last band: an 8-row band whose last item would hang over the matrix by more than 4 rows is pulled
    // up instead (it overlaps the one before; rp_band_y0) - fewer zero rows below the window, shorter column-pair rows
    const int y0max = rp_band_y0((rh + band - 1) / band - 1, (rh + band - 1) / band, rh, band);
    L.wrows = y0max + 2 * nst;                       // last step reads rows y0 + 2 (nst - 1) and + 1
    if (L.wrows < wh + 3) L.wrows = wh + 3;
    L.win_off = kMiscMfmaBytes;
    L.sii_off = gs ? 0 : round_up(L.win_off + L.wrows * L.wpitch, 16);
    L.u_off = gs ? round_up(L.win_off + L.wrows * L.wpitch, 16) : round_up(L.sii_off + rh * rw * 4, 16);
    L.ncp = (s - 32 + 1) / 2; L.nrg = 2;
    // transposed copy: one row per window column 32 .. 32 + rw - 1 + 2 ncp - 1 (the builder writes four rows at a time);
    // a lane reads 20 bytes (4-row band: shifts 0..3 + 16) or 24 (8-row band) from rho = y0 + 16 (g >> 1) (template rows
    // 0..31) and from y0 + 32 (rows 32..47), so the last byte read is y0max + 32 + 23
    L.wp_rows = rw + 2 * L.ncp - 1;
    L.wp_len = y0max + 56;
    L.wp_pitch = round_up(L.wp_len, 4);
    if (!((L.wp_pitch / 4) & 1)) L.wp_pitch += 4;    // 4 x odd: the 16 rows read by the lanes of one k-group fall into distinct banks
    L.tab_rows = s + 3;
    L.tab_pitch = tab_pitch;
    L.strip_off = L.u_off + L.tab_rows * tab_pitch;
    const int tc = s / 2 + 1;
    int r = 0;
    while (r * r < 2 * tc * tc) ++r;
    L.pradius = r + 1;
    L.pdim = 2 * L.pradius + 2;
    L.ppitch = round_up(L.pdim, 4);
    L.trow_bytes = 4096;                            // half of the 8 KB the row-pair winner's operands take (rp_winner_stage)
    // Everything behind the strip operands, for a given patch offset.  The image-1 patch (and the column-pair copy) can be
    // loaded together with the window only where they do not lie on the column sums (pm_kernel_rp: patch_early); with a
    // short operand table (slot groups) the natural place does, so the patch is moved up behind the column sums - unless
    // the larger footprint would cost a workgroup per CU.
    auto place = [&](int patch_off, int cap) {
        L.patch_off = patch_off;
        L.queue_cap = cap;
        const int patch_end = round_up(L.patch_off + L.pdim * L.ppitch + 1, 16);
        L.queue_off = one_group ? round_up(L.patch_off, 16) : patch_end;
        // the transposed columns only live during the sweep: behind the queue, inside the union (built when the templates
        // are - with one group of angles the patch is dead by then and they lie over it, like the queue -, overwritten by
        // the winner's NCC matrix)
        L.wp_off = round_up(L.queue_off + cap * 16, 16);
        int u = big ? 0 : wh * rw * 4;                           // row sums of w'^2 (rp_sums)
        const int sweep = L.wp_off + L.wp_rows * L.wp_pitch - L.u_off;
        if (u < sweep) u = sweep;
        // winner: operands + NCC matrix, and (one group of angles: the fused Hessian, pm_kernel_rp prologue) 5 KB of
        // histogram and key list behind them; then the Hessian magnitudes unless they fit over the window + operands
        int winner;
        if (big) {
            L.ccm_off = round_up(rh * rw * 4, 256);
            L.hes_off = L.ccm_off + round_up(wh * rw * 4, 256);
            L.big_bytes = L.hes_off + round_up(rh * rw * 4, 256);
            winner = 2 * L.trow_bytes + (one_group ? 5120 : 0);
        } else if (gs) {
            // (the magnitudes start at the window; where window + operands are too short for them the NCC matrix moves up)
            const int avail = L.u_off + 2 * L.trow_bytes - L.win_off;
            const int extra = (own_hes || rh * rw * 4 <= avail) ? 0 : round_up(rh * rw * 4 - avail, 16);
            L.ccm_off = L.u_off + 2 * L.trow_bytes + extra;
            winner = L.ccm_off - L.u_off + round_up(rh * rw * 4, 16) + (one_group ? 5120 : 0);
            L.hes_off = own_hes ? L.u_off + winner : L.win_off;
            if (own_hes) winner += round_up(rh * rw * 4, 16);
        } else {
            L.ccm_off = L.u_off + 2 * L.trow_bytes;
            L.hes_off = L.sii_off;
            winner = 2 * L.trow_bytes + round_up(rh * rw * 4, 16) + (one_group ? 5120 : 0);
        }
        if (u < winner) u = winner;
        L.total = round_up(L.u_off + u, 16);
    };
    const int natural = L.strip_off + L.ncp * L.nrg * 1024 + 16, clear = round_up(L.u_off + wh * rw * 4, 16);   // (+ 16 scratch bytes)
    // Sum w' per placement - the output of the all-ones slot, which the sweep computes for every placement anyway - is kept
    // for the winner's matrix where the residency class has 4 B per placement to spare (full table, sums in LDS: border 20,
    // 58 % of the benchmark's points): the winner then multiplies no all-ones operand, half of its matrix instructions.
    // It goes behind sum w'^2 and everything from the union on moves up (applied at the end).
    const int si_pad = (!gs && one_group && tab_pitch == 512) ? round_up(rh * rw * 4, 16) : 0;
    bool si_fits = false;
    // the candidate queue takes what its residency class leaves: the smallest queue decides the class, then the kept sums
    // take their share, then the queue grows
    auto place_q = [&](int patch_off) {
        place(patch_off, kRpQueueMin);
        const int limit = rp_class_limit(L.total);
        si_fits = si_pad > 0 && L.total + si_pad + (kRpQueueWithSi - kRpQueueMin) * 16 <= limit;   // (not at the price of a short queue)
        const int room = (limit - (si_fits ? si_pad : 0) - L.wp_rows * L.wp_pitch - L.queue_off) / 16;   // entries that keep the class
        const int cap = room > kQueueCap ? kQueueCap : room;
        if (cap > kRpQueueMin) place(patch_off, cap);
    };
    place_q(natural);
    if (tab_pitch < 512 && clear > natural) {                    // (full table: measured no gain at the borders it would change)
        const int limit = rp_class_limit(L.total);
        place_q(clear);
        if (L.total > limit) place_q(natural);
    }
    L.si_off = 0;
    if (si_fits) {
        L.si_off = L.u_off;
        L.u_off += si_pad; L.strip_off += si_pad; L.patch_off += si_pad; L.queue_off += si_pad; L.wp_off += si_pad; L.ccm_off += si_pad; L.total += si_pad;
    }
    return L;
}

// kept accumulators (PMArgs::gs_keep_acc): the table is padded to whole work items, so that the sweep stores without predicates -
// cols_pad = the placement columns its tiles cover, rows_pad = bands x output rows per item.  (Measured, 3 angles: predicated
// stores +6 % at border 30, stores of the lanes beyond the matrix redirected to one spare entry +2 %; the padding costs 30 %
// more bytes, which the recycled blocks keep in cache.)
__host__ __device__ inline void rp_acc_dims(int rh, int rw, int rows_per_item, int &cols_pad, int &rows_pad)
{
    const int rem = rw % 32;
    cols_pad = 32 * (rw / 32 + (rem > 16 ? 1 : 0)) + ((rem > 0 && rem <= 16) ? 16 : 0);
    rows_pad = (rh + rows_per_item - 1) / rows_per_item * rows_per_item;
}
// u32 entries of a point's block of global memory: sum w'^2 (+ sum w' when kept on its own), + the accumulator table
__host__ __device__ inline uint32_t rp_block_entries(int rh, int rw, int rows_per_item, int slots_per_group, bool keep_si, bool keep_acc)
{
    const uint32_t npos64 = (uint32_t)round_up(rh * rw, 64);
    if (!keep_acc) return npos64 * (keep_si ? 2u : 1u);
    int cp, rp;
    rp_acc_dims(rh, rw, rows_per_item, cp, rp);
    return npos64 + (uint32_t)round_up(cp * rp * slots_per_group, 64);
}

// window pitches the row-pair kernel is instantiated for (besides the run-time pitch): the smallest one that holds
// `natural`, or 0 (run-time pitch) beyond the largest
__host__ __device__ inline int rp_class_pitch(int natural)
{
    return natural <= 104 ? 104 : natural <= 112 ? 112 : natural <= 136 ? 136 : natural <= 168 ? 168 : 0;
}
// paired: 0 = one group of 16 slots, 1 = two groups (at most 7 angles), 2 = four groups (at most 3 angles)
// occ: wavefronts per SIMD the build allows - 3, or 4 (128 VGPRs; slot groups with pitch 104 only: four workgroups per CU)
// kept accumulators are compiled into the slot-group kernels (paired != 0); into all with -DSID_KEEP_ACC_FULL (pm_kernel_rp.inc)
#ifdef SID_KEEP_ACC_FULL
constexpr bool kKeepAccFull = true;
#else
constexpr bool kKeepAccFull = false;
#endif
int launch_pm_rp(const PMArgs &args, int lds_bytes, int nthreads, int band, int paired, int pitch, int occ, void *stream, bool big = false);
bool rp_pitch_instantiated(int band, int paired, int pitch, int occ = 3);

__host__ __device__ inline int samp_pitch(int s) { return round_up(s, 4); }
constexpr int kSamp2Cap = 320;                                              // units per angle: 35 rows x 9 quads = 315
constexpr int kSamp2Run = 4, kSamp2Gat = kSamp2Run + kSamp2Cap, kSamp2Id = kSamp2Gat + 2 * kSamp2Cap, kSamp2Words = kSamp2Id + kSamp2Cap / 2;
constexpr double kSampGuard = 1e-5;   // table entries whose coordinate is this close to k + 1/2 are flagged

int launch_pm_mfma(const PMArgs &args, int lds_bytes, int nthreads, int band, bool paired, void *stream);
constexpr int kPairedMaxAngles = 7;   // angle sets this small run the paired sweep
constexpr int kQuadMaxAngles = 3;     // ... and these four slot groups (row-pair kernel)
bool mfma_band8_supported(int s);
bool mfma_img_size_supported(int s);

int launch_rsqrt(const double *x, double *y, int64_t n, void *stream);
int launch_ncc_selftest(unsigned long long seed, int blocks, int per_thread, int s, unsigned long long *out, void *stream);
int launch_hypot_selftest(unsigned long long seed, int blocks, int per_thread, unsigned long long *out, void *stream);
int max_lds_bytes();

}  // namespace sid
