#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r05_final; mkdir -p $OUT; cd $R
md5sum sea_ice_drift_amd/libsid_pm.so > $OUT/lib_md5_suite.txt
timeout 1500 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "rc $?" >> $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
bash tools/archive/r5_knearest.sh
