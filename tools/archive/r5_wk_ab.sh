#!/bin/bash
# A/B: rp_winner_kept with the loads of all batches issued up front (wk) against one round trip per batch (wkserial)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_wk; mkdir -p $OUT; cd $R
one() {  lib=$1; shift
  SID_PM_LIB=$R/build/ab/lib_$lib.so timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check 4000 "$@" 2>>$OUT/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib [$*]: %.4f ms  kernel %.4f ms  ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d.get('parity_check', {}).get('ok')))" | tee -a $OUT/ab.txt
}
for round in 1 2 3; do
  for cfg in "--angles 1 --img-size 35" "--angles 1 --img-size 35 --border 20" "--angles 1 --border 26" "--angles 3" "--angles 3 --border 20"; do
    one wkserial $cfg
    one wk $cfg
  done
done
