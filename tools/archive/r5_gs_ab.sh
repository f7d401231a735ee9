#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_gs_ab; mkdir -p $OUT; cd $R
one() {  envs=$1; shift
  env $envs timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check 2000 "$@" 2>>$OUT/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$envs [$*]: %.4f ms  kernel %.4f ms  launches %d ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['launches_per_step'], d.get('parity_check', {}).get('ok')))" | tee -a $OUT/ab.txt
}
for round in 1 2 3; do
  one X=1
  one SID_PM_ALWAYS_GS=1
  one X=1 --angles 1 --img-size 35
  one SID_PM_ALWAYS_GS=1 --angles 1 --img-size 35
done
