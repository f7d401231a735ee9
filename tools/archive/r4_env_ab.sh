#!/bin/bash
# same-box A/B of one environment switch (through gpurun): bash tools/r4_env_ab.sh tag VAR ["cfg;cfg;..."]   ("off" = VAR=1 set)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r4_env}; VAR=$2; mkdir -p $OUT; cd $R
IFS=';' read -ra CFGS <<< "${3:---border 20;--border 23;--border 30;--border 40;}"
[ -z "${3:-}" ] && CFGS+=("")
for cfg in "${CFGS[@]}"; do
for mode in off on off on; do
  if [ $mode = on ]; then unset $VAR; else export $VAR=1; fi
  timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check 2000 $cfg 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$VAR unset=$mode [$cfg]', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_per_step'],4), d['parity_check']['ok'])" | tee -a $OUT/ab.txt
done; done
