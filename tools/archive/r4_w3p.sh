#!/bin/bash
# three-wavefront four-per-CU class for the slot-group kernels (3 / 7 angles) against the shipped classes (through gpurun)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r4_w3p}; mkdir -p $OUT; cd $R
IFS=';' read -ra CFGS <<< "${W3_CFGS:---angles 1 --border 20;--angles 1 --border 22;--angles 1 --border 25;--angles 1 --border 28;--angles 1;--angles 1 --img-size 35;--angles 3 --border 20;--angles 3 --border 23;--angles 3 --border 26;--angles 3}"
for cfg in "${CFGS[@]}"; do
for mode in off on off on; do
  if [ $mode = on ]; then export SID_PM_W3_PAIRED=1; else unset SID_PM_W3_PAIRED; fi
  SID_PM_VERBOSE=1 timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check 2000 $cfg 2>/tmp/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('w3p $mode [$cfg]', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_per_step'],4), d['parity_check']['ok'])" | tee -a $OUT/w3p.txt
done
grep "sid_pm: launch" /tmp/err.txt | sort | uniq -c | head -6
done
