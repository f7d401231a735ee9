// Host-side invariants of the row-pair kernel's LDS layout (sea_ice_drift_amd/csrc/pm_kernel.h rp_lds_layout): compiled and
// run by tests/test_lds_layout.py.  Prints one line per violated invariant; exit code = number of violations (capped).
#include <cstdio>
#include <cstdint>
#include <initializer_list>
#define __host__
#define __device__
#include "../../sea_ice_drift_amd/csrc/pm_kernel.h"

static int bad = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++bad; if (bad < 40) { printf(__VA_ARGS__); printf("  [%s]\n", #cond); } } } while (0)

int main()
{
    using namespace sid;
    for (int s = 34; s <= 35; ++s)
        for (int paired = 0; paired <= 2; ++paired)
            for (int band = 4; band <= (paired ? 4 : 8); band += 4)
                for (int b = 2; b <= 60; ++b)
                    for (int dw = 0; dw <= 6; dw += 3) {                       // square and clipped (non-square) windows
                        const int hws = s / 2, wh = 2 * hws + 2 * b + 1, ww = wh - dw;
                        if (ww < s + 1) continue;
                        const int rows = paired == 2 ? 16 : paired == 1 ? 8 : band, tp = rp_tab_pitch(paired);
                        // several angle groups | one group | one group + own Hessian buffer (all with sum w'^2 in global memory: gs) |
                        // one group, sums in LDS | several groups, sums in LDS
                        for (int variant = 0; variant <= 4; ++variant) {
                        const int one_group = variant == 1 || variant == 2 || variant == 3;
                        const bool gs = variant <= 2;
                        const bool own_hes = gs && variant != 1;
                        const RpLdsLayout N = rp_lds_layout(wh, ww, s, one_group != 0, rows, 0, tp, own_hes, gs);
                        const int cp = rp_class_pitch(N.wpitch);
                        for (int pitch : {0, cp}) {
                            if (pitch && pitch < N.wpitch) continue;
                            const RpLdsLayout L = rp_lds_layout(wh, ww, s, one_group != 0, rows, pitch, tp, own_hes, gs);
                            const int rh = wh - s + 1, rw = ww - s + 1;
                            char tag[96]; snprintf(tag, sizeof tag, "s=%d paired=%d band=%d b=%d dw=%d pitch=%d variant=%d", s, paired, band, b, dw, pitch, variant);
                            CHECK(L.wpitch >= ww && L.wpitch % 8 == 0 && (!pitch || L.wpitch == pitch), "%s: window pitch %d", tag, L.wpitch);
                            // (gs: sum w'^2 per placement lives in global memory and the union follows the window)
                            if (gs) CHECK(L.win_off >= kMiscMfmaBytes && L.u_off >= L.win_off + L.wrows * L.wpitch && L.sii_off == 0, "%s: window overlaps the union", tag);
                            else {
                                CHECK(L.win_off >= kMiscMfmaBytes && L.sii_off >= L.win_off + L.wrows * L.wpitch, "%s: window overlaps the sums", tag);
                                CHECK(L.u_off >= L.sii_off + rh * rw * 4, "%s: sums overlap the union", tag);
                                // sum w' kept for the winner: behind sum w'^2, in front of the union, never at the price of a residency class
                                if (L.si_off) CHECK(L.si_off >= L.sii_off + rh * rw * 4 && L.u_off >= L.si_off + rh * rw * 4 && one_group && tp == 512 &&
                                                    rp_class_limit(L.total) == rp_class_limit(L.total - ((rh * rw * 4 + 15) / 16) * 16), "%s: kept sums", tag);
                            }
                            if (gs) CHECK(L.si_off == 0, "%s: kept sums with the sums in global memory", tag);
                            CHECK(L.tab_pitch == tp && L.strip_off == L.u_off + L.tab_rows * tp, "%s: table", tag);
                            CHECK(L.patch_off >= L.strip_off + L.ncp * L.nrg * 1024 + 16, "%s: patch overlaps the strip operands", tag);
                            CHECK(L.queue_cap >= kRpQueueMin && L.queue_cap <= kQueueCap, "%s: queue of %d entries", tag, L.queue_cap);
                            CHECK(L.queue_off >= L.patch_off && L.wp_off >= L.queue_off + L.queue_cap * 16, "%s: queue / transposed columns", tag);
                            // several groups of angles: the patch stays live through the sweeps; one group: queue and columns lie over it
                            if (!one_group) CHECK(L.queue_off >= L.patch_off + L.pdim * L.ppitch + 1, "%s: queue overlaps the live patch", tag);
                            CHECK(L.patch_off + L.pdim * L.ppitch + 1 <= L.total, "%s: patch beyond the end", tag);
                            CHECK(L.wp_rows >= rw + 2 * L.ncp - 1 && L.wp_pitch % 4 == 0 && ((L.wp_pitch / 4) & 1), "%s: transposed columns %d x %d", tag, L.wp_rows, L.wp_pitch);
                            // last strip read: 24 bytes from y0max + 32 (rp_item_strip); y0max <= rh - 1 rounded up to the band
                            CHECK(L.wp_pitch >= rp_band_y0((rh + rows - 1) / rows - 1, (rh + rows - 1) / rows, rh, rows) + 56, "%s: transposed row too short", tag);
                            CHECK(L.wp_off + L.wp_rows * L.wp_pitch <= L.total, "%s: transposed columns beyond the end", tag);
                            CHECK(L.u_off + wh * rw * 4 <= L.total, "%s: row sums beyond the end", tag);
                            CHECK(L.ccm_off >= L.u_off + 2 * L.trow_bytes && L.ccm_off + rh * rw * 4 + (one_group ? 5120 : 0) <= L.total, "%s: winner + histogram beyond the end", tag);
                            // the Hessian magnitudes: over the window + winner operands (never into the NCC matrix), or LDS of their own
                            if (!gs) CHECK(L.hes_off == L.sii_off && L.ccm_off == L.u_off + 2 * L.trow_bytes, "%s: Hessian magnitudes take the LDS of the sums", tag);
                            else if (own_hes) CHECK(L.hes_off >= L.ccm_off + rh * rw * 4 + (one_group ? 5120 : 0) && L.hes_off + rh * rw * 4 <= L.total, "%s: own Hessian buffer", tag);
                            else CHECK(L.hes_off == L.win_off && L.hes_off + rh * rw * 4 <= L.ccm_off, "%s: Hessian magnitudes reach the NCC matrix", tag);
                            CHECK(L.total % 16 == 0, "%s: total %d", tag, L.total);
                            if (b <= 50 && dw == 0) CHECK(L.total <= 160 * 1024, "%s: %d bytes do not fit the LDS", tag, L.total);
                        }
                        }
                    }
    // big layouts (search borders beyond the LDS: 69 .. 111 at s = 34 / 35): the per-placement tables - sum w'^2, the row sums of
    // rp_sums / the NCC matrix of the winning angle, the Hessian magnitudes - are 256-byte aligned pieces of a block of global
    // memory; LDS holds the window, the sweep's operands and the winner's operands + histogram
    for (int s = 34; s <= 35; ++s)
        for (int b = 40; b <= 111; ++b)
            for (int variant = 0; variant < 3; ++variant) {
                const int w = 2 * (s / 2) + 2 * b + 1, wh = w, ww = w - (variant == 2 ? 5 : 0), rh = wh - s + 1, rw = ww - s + 1;
                const bool one_group = variant != 1;
                const RpLdsLayout L = rp_lds_layout(wh, ww, s, one_group, 4, 0, 512, !one_group, true, true);
                char tag[96]; snprintf(tag, sizeof tag, "big s=%d b=%d variant=%d", s, b, variant);
                CHECK(L.total <= 160 * 1024 && L.total % 16 == 0, "%s: %d bytes of LDS", tag, L.total);
                CHECK(L.sii_off == 0 && L.si_off == 0 && L.u_off >= L.win_off + L.wrows * L.wpitch, "%s: window / union", tag);
                CHECK(L.ccm_off % 256 == 0 && L.hes_off % 256 == 0 && L.big_bytes % 256 == 0, "%s: alignment of the global pieces", tag);
                CHECK(L.ccm_off >= rh * rw * 4 && L.hes_off >= L.ccm_off + wh * rw * 4 && L.big_bytes >= L.hes_off + rh * rw * 4, "%s: global pieces overlap", tag);
                CHECK(L.wp_off + L.wp_rows * L.wp_pitch <= L.total && L.wp_off >= L.queue_off + L.queue_cap * 16, "%s: transposed columns", tag);
                CHECK(L.u_off + 2 * L.trow_bytes + (one_group ? 5120 : 0) <= L.total, "%s: winner operands + histogram", tag);
                CHECK(L.patch_off + L.pdim * L.ppitch + 1 <= L.total && L.patch_off >= L.strip_off + L.ncp * L.nrg * 1024 + 16, "%s: patch", tag);
                if (!one_group) CHECK(L.queue_off >= L.patch_off + L.pdim * L.ppitch + 1, "%s: queue overlaps the live patch", tag);
                CHECK(rh * rw < 65536, "%s: %d placements (16-bit histogram counters, reciprocal divisions)", tag, rh * rw);
            }
    CHECK(rp_lds_layout(259, 259, 34, true, 4, 0, 512, false, true, true).total > 160 * 1024, "border 112 fits the LDS?");
    CHECK(rp_lds_layout(171, 171, 34, true, 4, 0, 512, false, true, false).total <= 160 * 1024 &&
          rp_lds_layout(173, 173, 34, true, 4, 0, 512, false, true, false).total > 160 * 1024, "tables in LDS: up to border 68");
    // residency classes of the benchmark's borders (the numbers DESIGN.md quotes)
    auto total = [](int b, int paired, bool gs) {
        const int w = 35 + 2 * b, rows = paired == 2 ? 16 : paired == 1 ? 8 : 4, tp = rp_tab_pitch(paired);
        int cp = rp_class_pitch(rp_lds_layout(w, w, 34, true, rows, 0, tp, false, gs).wpitch);
        if (gs && cp < 136) cp = 136;                                  // (the gs instantiations start at pitch 136)
        return rp_lds_layout(w, w, 34, true, rows, cp, tp, false, gs).total;
    };
    // sums in LDS: borders 20..27 three per CU; with the sums in global memory 28..36 as well, 37..47 two per CU
    CHECK(total(20, 0, false) <= 42 * 1280 && total(27, 0, false) <= 42 * 1280 && total(28, 0, false) > 42 * 1280, "15 angles, LDS sums: borders 20..27 three per CU");
    CHECK(total(28, 0, true) <= 42 * 1280 && total(36, 0, true) <= 42 * 1280 && total(37, 0, true) > 42 * 1280, "15 angles, gs: borders 28..36 three per CU");
    CHECK(total(37, 0, true) <= 64 * 1280 && total(47, 0, true) <= 64 * 1280 && total(48, 0, true) > 64 * 1280, "15 angles, gs: borders 37..47 two per CU");
    CHECK(total(20, 1, false) <= 32 * 1280 && total(21, 1, false) <= 32 * 1280 && total(22, 1, false) > 32 * 1280, "7 angles, LDS sums: borders 20, 21 four per CU");
    CHECK(total(20, 2, false) <= 32 * 1280 && total(21, 2, false) <= 32 * 1280, "3 angles, LDS sums: borders 20, 21 four per CU");
    {   // border 20 (58 % of the benchmark's points) keeps sum w' for the winner and stays at three per CU; border 21 has no room
        const RpLdsLayout a = rp_lds_layout(75, 75, 34, true, 4, 112, 512, false, false), b = rp_lds_layout(77, 77, 34, true, 4, 112, 512, false, false);
        CHECK(a.si_off != 0 && a.total <= 42 * 1280 && b.si_off == 0, "kept sums at border 20 only (%d, %d)", a.total, b.total);
    }
    // the three-wavefront class of the full table (pitch code 1104): borders 20 .. 23 with the sums in global memory fit FOUR times
    for (int b = 20; b <= 24; ++b) {
        const int w = 35 + 2 * b;
        const RpLdsLayout L = rp_lds_layout(w, w, 34, true, 4, rp_pitch_bytes(1104), 512, false, true);
        if (b <= 23) CHECK(L.total <= 32 * 1280 && L.wpitch == 104 && L.queue_cap >= kRpQueueMin, "border %d: %d bytes, four per CU expected", b, L.total);
        else CHECK(L.wpitch > 104, "border 24 keeps a window pitch of 104?");
    }
    static_assert(rp_pitch_is_gs(1104) && rp_pitch_is_w3(1104) && !rp_pitch_is_w3(168) && rp_pitch_bytes(1104) == 104 && rp_pitch_bytes(136) == 136, "pitch codes");
    printf("%d violations\n", bad);
    return bad > 100 ? 100 : bad;
}
