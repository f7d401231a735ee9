#!/bin/bash
# round 5: the x-shift form of the winner's matrix against the output-row form
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_seventh; mkdir -p $OUT; cd $R
timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_round5.py -x -q > $OUT/pytest.txt 2>&1; echo "rc $?" >> $OUT/pytest.txt; tail -15 $OUT/pytest.txt
one() {  # lib env check cfg...
  lib=$1; envs=$2; chk=$3; shift 3
  env $envs SID_PM_LIB=$R/build/ab/lib_$lib.so timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check $chk "$@" 2>>$OUT/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib $envs [$*]: %.4f ms  kernel %.4f ms  ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d.get('parity_check', {}).get('ok')))" | tee -a $OUT/ab.txt
}
for round in 1 2 3; do
  for cfg in "" "--border 20" "--border 26" "--border 30" "--border 40" "--border 50" "--img-size 35" "--angles 3 --border 30"; do
    one rows X=1 4000 $cfg
    one new X=1 4000 $cfg
  done
done
for a in 7; do SID_PM_LIB=$R/build/ab/lib_new.so SID_PHASE_ANGLES=$a SID_PHASE_BORDERS=20,30 timeout 300 python3 tools/phase_cycles.py >> $OUT/phase_cycles.txt 2>&1; done
