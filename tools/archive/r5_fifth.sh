#!/bin/bash
# round 5, fifth GPU call: float32 hypot, padded accumulators + recycled blocks, per-launch keep policy for 7 angles
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_fifth; mkdir -p $OUT; cd $R
timeout 1800 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
timeout 900 python3 -m pytest tests/test_gpu_soak.py -x -q -k "evictions" > $OUT/pytest_evict.txt 2>&1; tail -5 $OUT/pytest_evict.txt
one() {  # lib env check cfg...
  lib=$1; envs=$2; chk=$3; shift 3
  env $envs SID_PM_LIB=$R/build/ab/lib_$lib.so timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check $chk "$@" 2>>$OUT/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib $envs [$*]: %.4f ms  kernel %.4f ms  ok %s maxdh %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d.get('parity_check', {}).get('ok'), d.get('parity_check', {}).get('max_abs_h_difference')))" | tee -a $OUT/ab.txt
}
for round in 1 2 3; do
  for cfg in "" "--border 20" "--angles 1 --img-size 35" "--angles 1 --img-size 35 --border 20" "--angles 1 --img-size 35 --border 30" "--angles 3" "--angles 3 --border 20"; do
    one base X=1 4000 $cfg
    one new X=1 4000 $cfg
    one hyp64 X=1 4000 $cfg
  done
  one new SID_PM_NO_RECYCLE=1 4000 --angles 1 --img-size 35
  one new SID_PM_KEEP_ACC=1 4000 --angles 3
done
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench_err.txt; tail -c 1200 $OUT/bench.json
