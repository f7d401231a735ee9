#!/bin/bash
# same-box A/B of library builds (through gpurun): bash tools/r6_ab.sh tag libA.so libB.so ... ; alternates the builds, AB_ROUNDS (3) rounds;
# AB_ARGS = extra bench.py arguments (a quoted string).  Prints the step, the kernel time and the reference-defaults kernel time.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
ARGS=${AB_ARGS:---check 64}
for round in $(seq 1 ${AB_ROUNDS:-3}); do
  for lib in "$@"; do
    SID_PM_LIB=$R/$lib timeout 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline $ARGS 2>$OUT/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); rd = d.get('reference_defaults') or {}
print('$lib round $round [$ARGS]: %.4f ms  kernel %.4f ms  ok %s  defaults kernel %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d.get('parity_check', {}).get('ok'), rd.get('kernel_ms_per_step')))" | tee -a $OUT/ab.txt
  done
done
