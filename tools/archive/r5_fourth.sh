#!/bin/bash
# round 5, fourth GPU call: recycled blocks + kept accumulators (slot groups), okbits / b32 single-tile reads (all kernels)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_fourth; mkdir -p $OUT; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
one() {  # lib env check cfg...
  lib=$1; envs=$2; chk=$3; shift 3
  env $envs SID_PM_LIB=$R/build/ab/lib_$lib.so timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check $chk "$@" 2>>$OUT/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib $envs [$*]: %.4f ms  kernel %.4f ms  ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d.get('parity_check', {}).get('ok')))" | tee -a $OUT/ab.txt
}
for round in 1 2 3; do
  one base X=1 2000
  one new X=1 2000
  one read2 X=1 2000
  one base X=1 2000 --border 20
  one new X=1 2000 --border 20
  one read2 X=1 2000 --border 20
  for cfg in "--angles 1 --img-size 35" "--angles 1 --img-size 35 --border 20" "--angles 1 --img-size 35 --border 30" "--angles 3" "--angles 3 --border 20"; do
    one base X=1 2000 $cfg
    one new X=1 2000 $cfg
    one new SID_PM_NO_RECYCLE=1 2000 $cfg
    one new SID_PM_KEEP_ACC=0 2000 $cfg
  done
done
