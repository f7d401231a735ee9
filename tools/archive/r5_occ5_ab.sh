#!/bin/bash
# A/B: five workgroups per CU for the slot-group launches at borders 20 / 21 (SID_PM_NO_OCC5=1 = four, as before)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_occ5; mkdir -p $OUT; cd $R
one() {  envs=$1; shift
  env $envs timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check 4000 "$@" 2>>$OUT/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$envs [$*]: %.4f ms  kernel %.4f ms  launches %d ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['launches_per_step'], d.get('parity_check', {}).get('ok')))" | tee -a $OUT/ab.txt
}
for round in 1 2 3; do
  for cfg in "--angles 1 --img-size 35" "--angles 1 --img-size 35 --border 20" "--angles 1" "--angles 3" "--angles 3 --border 20"; do
    one SID_PM_NO_OCC5=1 $cfg
    one X=1 $cfg
  done
done
