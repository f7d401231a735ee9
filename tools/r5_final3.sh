#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r05_final; mkdir -p $OUT; cd $R
md5sum sea_ice_drift_amd/libsid_pm.so > $OUT/lib_md5_suite.txt
timeout 1500 python3 -m pytest tests -x -q -m gpu -rs > $OUT/pytest_gpu.txt 2>&1; echo "rc $?" >> $OUT/pytest_gpu.txt; tail -6 $OUT/pytest_gpu.txt
for k in 1 2; do timeout 200 python3 tools/eviction_soak.py 1 35 300; timeout 200 python3 tools/eviction_soak.py 3 34 300; done > $OUT/eviction_soak.txt 2>/dev/null; cat $OUT/eviction_soak.txt
timeout 400 python3 tools/shard_sim.py 8 > $OUT/shard_sim_8_new_cuts.json 2>> $OUT/err.txt; grep -A12 "contiguous" $OUT/shard_sim_8_new_cuts.json | tr -d '\n ' | cut -c1-600
