#!/bin/bash
# Nested phase ablations of the PM kernel on the GPU box: time + instruction counters per build.
# Builds: tools/ab/lib_A0.so (full) .. lib_A4.so (see the build lines in DESIGN.md / tools/ablation_build.sh).
set -u
TAG=${1:-abl}
BORDER=${2:-20}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-A0 A1 A2 A3 A4 A5 A6}; do
  [ -f $R/tools/ab/lib_$v.so ] || continue
  export SID_PM_LIB=$R/tools/ab/lib_$v.so
  python3 $R/bench.py --border $BORDER --steps 10 --warmup 2 --no-cpu-baseline --check 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms_per_step'])" >> $OUT/times_b$BORDER.txt
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d /tmp/p1_$TAG$v -o p1 -- python3 $R/bench.py --border $BORDER --steps 1 --warmup 1 --no-cpu-baseline --check 0 > /dev/null 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/p2_$TAG$v -o p2 -- python3 $R/bench.py --border $BORDER --steps 1 --warmup 1 --no-cpu-baseline --check 0 > /dev/null 2>&1
  echo "== $v" >> $OUT/pmc_b$BORDER.txt
  for p in p1 p2; do python3 $R/tools/rocpd_summary.py $(find /tmp/${p}_$TAG$v -name "*.db" | head -1) | sed -n '/PMC per dispatch/,$p' >> $OUT/pmc_b$BORDER.txt; done
done
unset SID_PM_LIB
cat $OUT/times_b$BORDER.txt
