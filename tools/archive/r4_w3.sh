#!/bin/bash
# the three-wavefront four-per-CU class of the full operand table, border by border, against three per CU (through gpurun)
# usage: bash tools/r4_w3.sh tag   (W3_CFGS="--border 20;--border 24" selects the configurations)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r4_w3}; mkdir -p $OUT; cd $R
IFS=';' read -ra CFGS <<< "${W3_CFGS:---border 20;--border 23;--border 24;--border 25;--border 26;--border 24 --img-size 35;}"
[ -z "${W3_CFGS:-}" ] && CFGS+=("")
for cfg in "${CFGS[@]}"; do
for mode in off on off on; do
  if [ $mode = on ]; then unset SID_PM_NO_W3; else export SID_PM_NO_W3=1; fi
  timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check 2000 $cfg 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('w3 $mode [$cfg]', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_per_step'],4), d['parity_check']['ok'])" | tee -a $OUT/w3.txt
done; done
