#!/bin/bash
# A longer determinism campaign than tests/test_gpu_soak.py (through gpurun): every line must report 0 differing results.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/campaign.txt; : > $OUT
run() { timeout 900 python3 $R/tools/determinism_check.py "$@" 2>/dev/null | tail -1 >> $OUT; }
run 4000 1000 --angles 7
run 4000 500 --angles 7 --no-table
run 4000 500 --angles 3
run 4000 300 --angles 3 --no-table
run 4000 300 --angles 1
run 4000 300 --angles 7 --img-size 35
run 4000 300 --angles 3 --img-size 35
run 4000 200 --angles 7 --border 28
run 4000 200 --angles 7 --border 36
run 4000 100 --angles 7 --border 50
run 4000 200 --angles 3 --border 30
run 10000 100 --angles 7
run 10000 60 --angles 3
SID_PM_NO_RP=1 run 4000 300 --angles 7
SID_PM_NO_RP=1 run 4000 300 --angles 3
run 4000 200 --angles 10 --img-size 40
# round 3: the four-per-CU class of the slot-group layouts (128-VGPR build) and the short operand table
run 4000 400 --angles 3 --border 20
run 4000 400 --angles 1 --border 21
run 4000 300 --angles 1 --no-table
run 4000 300 --angles 1 --img-size 35
run 4000 200 --angles 3 --border 26
run 4000 200 --angles 1 --border 38
run 10000 100 --angles 1
run 10000 100 --angles 3
run 10000 40 --angles 1 --img-size 35
# round 4: the launches that keep sum w'^2 in global memory (borders 28 .. 47), the transposed strip columns, the new class boundaries
run 4000 300 --angles 7 --border 27
run 4000 300 --angles 7 --border 30
run 4000 200 --angles 7 --border 34
run 4000 150 --angles 7 --border 42
run 4000 100 --angles 7 --border 47
run 4000 300 --angles 3 --border 30
run 4000 200 --angles 1 --border 42
SID_PM_ALWAYS_GS=1 run 4000 300 --angles 7 --border 20
SID_PM_ALWAYS_GS=1 run 4000 300 --angles 3
SID_PM_NO_GS=1 run 4000 300 --angles 7
# round 4 (late): three wavefronts per point and four points per CU (full table, borders 20 .. 23), the per-XCD pool of
# global-memory blocks, search borders beyond the LDS (tables in global memory)
run 4000 500 --angles 7 --border 20
run 4000 400 --angles 7 --border 23
run 4000 300 --angles 7 --border 21 --img-size 35
run 10000 100 --angles 7 --border 20
run 3000 40 --angles 7 --border 80
SID_PM_NO_W3=1 run 4000 200 --angles 7 --border 20
# (round 4's pool of blocks picked by hardware slot is gone from the library: DESIGN.md section 5.5)
run 4000 20000 --angles 7 --img-size 35 --border 44
run 10000 2000 --angles 1
# round 5: blocks recycled through the per-XCD free lists (launches of at most 3 angles everywhere, of at most 7 in the
# four-per-CU class), accumulators kept for the winner, the sorted sampling table, exclusive blocks for comparison
run 4000 20000 --angles 1 --img-size 35
run 4000 5000 --angles 1 --border 20
run 4000 5000 --angles 3 --border 20
run 4000 2000 --angles 3
run 10000 1000 --angles 1 --img-size 35
SID_PM_NO_RECYCLE=1 run 4000 1000 --angles 1 --img-size 35
SID_PM_KEEP_ACC=0 run 4000 1000 --angles 1 --img-size 35
SID_PM_SAMP2=1 run 4000 300 --angles 7
# round 6: bitmap free lists, for EVERY launch that keeps tables in global memory (the 15- and 7-angle kernels too); round 5's
# split for comparison; the large-window pipeline (every point through it)
run 4000 3000 --angles 7
run 4000 3000 --angles 7 --border 20
run 4000 2000 --angles 3
run 10000 300 --angles 7
run 4000 10000 --angles 1 --img-size 35
SID_PM_RECYCLE_ALL=0 run 4000 1000 --angles 7
SID_PM_ALL_LARGE=1 run 1000 30 --angles 1
SID_PM_ALL_LARGE=1 run 1000 20 --angles 7 --img-size 35
echo "library md5 $(md5sum $R/sea_ice_drift_amd/libsid_pm.so | cut -d' ' -f1)" >> $OUT
cat $OUT
