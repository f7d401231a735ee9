#!/bin/bash
# second batch of the round's evidence (through gpurun): widening rooflines, shard simulation, border cost staircase
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r04_ev2; mkdir -p $OUT; cd $R
bash tools/r4_widening_profiles.sh r04_widening > $OUT/widening.log 2>&1
for w in 8 4 2; do timeout 600 python3 tools/shard_sim.py $w > $OUT/shard_sim_$w.json 2>> $OUT/err.txt; done
for k in 15 7 3; do timeout 900 python3 tools/border_cost.py $k > $OUT/border_cost_$k.json 2>> $OUT/err.txt; done
timeout 300 python3 tools/stage_probe.py > $OUT/stage_probe.json 2>> $OUT/err.txt
timeout 300 python3 tools/orb_ab.py > $OUT/orb_ab.json 2>> $OUT/err.txt
ls -la $OUT $R/gpurun_out/r04_widening
