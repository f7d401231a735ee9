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
                        const RpLdsLayout N = rp_lds_layout(wh, ww, s, true, rows, 0, tp);
                        const int cp = rp_class_pitch(N.wpitch);
                        for (int pitch : {0, cp}) {
                            if (pitch && pitch < N.wpitch) continue;
                            const RpLdsLayout L = rp_lds_layout(wh, ww, s, true, rows, pitch, tp);
                            const int rh = wh - s + 1, rw = ww - s + 1;
                            char tag[96]; snprintf(tag, sizeof tag, "s=%d paired=%d band=%d b=%d dw=%d pitch=%d", s, paired, band, b, dw, pitch);
                            CHECK(L.wpitch >= ww && L.wpitch % 8 == 0 && (!pitch || L.wpitch == pitch), "%s: window pitch %d", tag, L.wpitch);
                            CHECK(L.win_off >= kMiscMfmaBytes && L.sii_off >= L.win_off + L.wrows * L.wpitch, "%s: window overlaps the sums", tag);
                            CHECK(L.u_off >= L.sii_off + rh * rw * 4, "%s: sums overlap the union", tag);
                            CHECK(L.tab_pitch == tp && L.strip_off == L.u_off + L.tab_rows * tp, "%s: table", tag);
                            CHECK(L.patch_off >= L.strip_off + L.ncp * L.nrg * 1024 + 16, "%s: patch overlaps the strip operands", tag);
                            CHECK(L.queue_off >= L.patch_off && L.wp_off >= L.queue_off + kQueueCap * 16, "%s: queue / column-pair copy", tag);
                            CHECK(L.wp_off >= L.patch_off + L.pdim * L.ppitch, "%s: column-pair copy overlaps the patch", tag);
                            CHECK(L.wp_off + L.wp_rows * L.wp_pitch <= L.total, "%s: column-pair copy beyond the end", tag);
                            CHECK(L.u_off + rh * ww * 4 <= L.total, "%s: column sums beyond the end", tag);
                            CHECK(L.u_off + 2 * L.trow_bytes + rh * rw * 4 + 5120 <= L.total, "%s: winner + histogram beyond the end", tag);
                            CHECK(L.total % 16 == 0, "%s: total %d", tag, L.total);
                            if (b <= 50 && dw == 0) CHECK(L.total <= 160 * 1024, "%s: %d bytes do not fit the LDS", tag, L.total);
                        }
                    }
    // residency classes of the benchmark's borders (the numbers DESIGN.md quotes)
    auto total = [](int b, int paired) {
        const int w = 34 + 2 * b, rows = paired == 2 ? 16 : paired == 1 ? 8 : 4, tp = rp_tab_pitch(paired);
        const int cp = rp_class_pitch(rp_lds_layout(w, w, 34, true, rows, 0, tp).wpitch);
        return rp_lds_layout(w, w, 34, true, rows, cp, tp).total;
    };
    CHECK(total(20, 0) <= 42 * 1280 && total(23, 0) <= 42 * 1280 && total(24, 0) > 42 * 1280, "15 angles: borders 20..23 three per CU");
    CHECK(total(20, 1) <= 32 * 1280 && total(21, 1) <= 32 * 1280 && total(22, 1) > 32 * 1280, "7 angles: borders 20, 21 four per CU");
    CHECK(total(20, 2) <= 32 * 1280 && total(22, 2) <= 32 * 1280, "3 angles: borders 20..22 four per CU");
    printf("%d violations\n", bad);
    return bad > 100 ? 100 : bad;
}
