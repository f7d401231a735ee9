#!/bin/bash
# round 5, first GPU call: the slot-identity probe, phase ablations of the round-4 library (15 angles and the reference's defaults),
# lone-workgroup phase clocks, and the driver-style bench line with the defaults block timed before the CPU baselines.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_first; mkdir -p $OUT; cd $R
( cd tools/ubench
  timeout 300 ./slot_life 100000 2048 40960 192 400
  timeout 300 ./slot_life 100000 1536 53760 256 400
  timeout 300 ./slot_life 30000 2048 40960 192 400 --evict ) > $OUT/slot_life.txt 2>&1
one() {  # lib cfg...
  lib=$1; shift
  SID_PM_LIB=$R/build/ab/lib_$lib.so timeout 300 python3 bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-also-defaults --check 0 "$@" 2>>$OUT/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib [$*]: %.4f ms  kernel %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step']))" | tee -a $OUT/ablation.txt
}
for v in A0 A1 A2 A6 A3 A4 A5 A0; do
  [ -f build/ab/lib_$v.so ] || continue
  one $v
  one $v --border 20
  one $v --angles 1 --img-size 35
  one $v --angles 1 --img-size 35 --border 20
done
for a in 7 1; do SID_PHASE_ANGLES=$a SID_PHASE_BORDERS=20,30 timeout 300 python3 tools/phase_cycles.py >> $OUT/phase_cycles.txt 2>&1; done
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench_err.txt
tail -c 1500 $OUT/bench.json
