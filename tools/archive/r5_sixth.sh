#!/bin/bash
# round 5, sixth GPU call: sorted sampling table, leading launches side by side, the eviction soak
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_sixth; mkdir -p $OUT; cd $R
timeout 1200 python3 -m pytest tests/test_gpu_round5.py -x -q > $OUT/pytest.txt 2>&1; echo "rc $?" >> $OUT/pytest.txt; tail -5 $OUT/pytest.txt
timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_soak.py -x -q -s -k "evictions" > $OUT/pytest_evict.txt 2>&1; echo "rc $?" >> $OUT/pytest_evict.txt; tail -12 $OUT/pytest_evict.txt
one() {  # lib env check cfg...
  lib=$1; envs=$2; chk=$3; shift 3
  env $envs SID_PM_LIB=$R/build/ab/lib_$lib.so timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check $chk "$@" 2>>$OUT/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib $envs [$*]: %.4f ms  kernel %.4f ms  ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d.get('parity_check', {}).get('ok')))" | tee -a $OUT/ab.txt
}
for round in 1 2 3; do
  for cfg in "" "--border 20" "--angles 1 --img-size 35" "--angles 1 --img-size 35 --border 20" "--angles 3"; do
    one hyp64 X=1 4000 $cfg
    one new X=1 4000 $cfg
    one new SID_PM_NO_SAMP2=1 4000 $cfg
  done
  one new SID_PM_SIDE_FIRST=2 4000
  one new SID_PM_SIDE_FIRST=3 4000
  one new SID_PM_SIDE_FIRST=2 4000 --angles 1 --img-size 35
  one new SID_PM_SIDE_FIRST=3 4000 --angles 1 --img-size 35
done
