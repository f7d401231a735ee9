#!/bin/bash
# same-box A/B of library builds (through gpurun): bash tools/r3_ab.sh tag libA.so libB.so ... ; alternates the builds, 3 rounds
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
ARGS=${AB_ARGS:---check 64}
for round in 1 2 3; do
  for lib in "$@"; do
    SID_PM_LIB=$R/$lib timeout 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib round $round: %.4f ms  kernel %.4f ms  ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d.get('parity_check', {}).get('ok')))" | tee -a $OUT/ab.txt
  done
done
