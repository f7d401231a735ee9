#!/bin/bash
# kernel times of the first-guess prelude (nearest key point through the buckets vs brute force) + the determinism campaign's round-5 lines
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r05_extra; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in grid brute; do
  if [ $mode = brute ]; then export SID_FG_NO_GRID=1; else unset SID_FG_NO_GRID; fi
  rm -rf /tmp/kt_fg_$mode
  rocprofv3 --kernel-trace --stats -d /tmp/kt_fg_$mode -o kt -- python3 $R/tools/prelude_profile.py > $OUT/prelude_profile_$mode.txt 2>&1
  python3 $R/tools/rocpd_summary.py $(find /tmp/kt_fg_$mode -name "*.db" | head -1) k_nearest | sed -n '1,/DISPATCHES/p' | grep -E "k_nearest|k_seed_bin|k_grid_scan|k_locate|name" > $OUT/first_guess_kernels_$mode.txt
done
unset SID_FG_NO_GRID
cd $R
bash tools/determinism_campaign.sh > $OUT/campaign.log 2>&1; cp gpurun_out/campaign.txt $OUT/determinism_campaign.txt
cat $OUT/first_guess_kernels_grid.txt $OUT/first_guess_kernels_brute.txt; tail -12 $OUT/determinism_campaign.txt
