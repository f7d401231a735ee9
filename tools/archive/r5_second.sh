#!/bin/bash
# round 5, second GPU call: new tests, A/B of the kept-accumulator winner, the slot probe with XCC ids, per-phase instruction counters
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_second; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_round5.py -x -q > $OUT/pytest_round5.txt 2>&1; tail -5 $OUT/pytest_round5.txt
( cd tools/ubench; timeout 200 ./slot_life 40000 2048 40960 192 400 --evict ) > $OUT/slot_life_evict.txt 2>&1
one() {  # lib env cfg...
  lib=$1; envs=$2; shift 2
  env $envs SID_PM_LIB=$R/build/ab/lib_$lib.so timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check 2000 "$@" 2>>$OUT/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib $envs [$*]: %.4f ms  kernel %.4f ms  ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['parity_check']['ok']))" | tee -a $OUT/ab.txt
}
for round in 1 2; do
  for cfg in "--angles 1 --img-size 35" "--angles 1 --img-size 35 --border 20" "--angles 3" "--angles 3 --border 20" "--angles 1 --img-size 35 --border 30"; do
    one A0 "X=1" $cfg
    one new "X=1" $cfg
    one new "SID_PM_KEEP_ACC=0" $cfg
  done
  one A0 "X=1"
  one new "X=1"
  one keepfull "SID_PM_KEEP_ACC=1"
  one keepfull "SID_PM_KEEP_ACC=0"
  one A0 "X=1" --border 20
  one new "X=1" --border 20
  one keepfull "SID_PM_KEEP_ACC=1" --border 20
  one keepfull "SID_PM_KEEP_ACC=0" --border 20
done
# per-phase instruction counters of the round-4 library (nested ablations), 15 angles, border 20
cd /tmp && export TMPDIR=/tmp
for v in A0 A1 A2 A6 A3 A4 A5; do
  export SID_PM_LIB=$R/build/ab/lib_$v.so
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/p1_$v -o p1 -- python3 $R/bench.py --border 20 --steps 1 --warmup 1 --no-cpu-baseline --no-also-defaults --check 0 > /dev/null 2>&1
  echo "== $v" >> $OUT/pmc_b20.txt
  python3 $R/tools/rocpd_summary.py $(find /tmp/p1_$v -name "*.db" | head -1) | sed -n '/PMC per dispatch/,$p' >> $OUT/pmc_b20.txt
done
unset SID_PM_LIB
