#!/bin/bash
# quick correctness + speed look after a kernel change (through gpurun); output under gpurun_out/<tag>/
# usage: bash tools/r4_check.sh tag [baseline_lib.so]   (the baseline library, if given, is timed on the same box)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r4_check}; mkdir -p $OUT
BASE=$2
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -8 $OUT/pytest.txt
SID_PM_VERBOSE=1 timeout 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --check 4000 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench.json; grep "sid_pm: launch" $OUT/bench.err | sort | uniq -c; tail -2 $OUT/bench.err
for round in 1 2; do
  for lib in $BASE sea_ice_drift_amd/libsid_pm.so; do
    SID_PM_LIB=$R/$lib timeout 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --check 64 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib round $round: %.4f ms  kernel %.4f ms  ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d.get('parity_check', {}).get('ok')))" | tee -a $OUT/ab.txt
  done
done
for cfg in "--border 20" "--border 26" "--border 38" "--angles 3" "--angles 3 --img-size 35" "--angles 1"; do timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 512 $cfg 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['parity_check']['ok'])" | tee -a $OUT/cfgs.txt; done
