#!/bin/bash
# phase clocks of single points + a short bench line (through gpurun); output under gpurun_out/r3_phase/
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r3_phase}; mkdir -p $OUT
cd $R
SID_PHASE_BORDERS=20,28,36,45 timeout 300 python3 tools/phase_cycles.py > $OUT/phases.txt 2>&1; cat $OUT/phases.txt
timeout 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err
