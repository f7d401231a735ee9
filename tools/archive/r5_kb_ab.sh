#!/bin/bash
# A/B: values per thread and batch in rp_winner_kept (shipped: 4)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_kb; mkdir -p $OUT; cd $R
one() {  lib=$1; shift
  L=$R/build/ab/lib_$lib.so; [ $lib = shipped ] && L=$R/sea_ice_drift_amd/libsid_pm.so
  SID_PM_LIB=$L timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check 4000 "$@" 2>>$OUT/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib [$*]: %.4f ms  kernel %.4f ms  ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d.get('parity_check', {}).get('ok')))" | tee -a $OUT/ab.txt
}
for round in 1 2 3; do
  for cfg in "--angles 1 --img-size 35" "--angles 1 --img-size 35 --border 20"; do
    for lib in shipped kb2 kb3 kb6; do one $lib $cfg; done
  done
done
