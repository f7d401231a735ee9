#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r05_final; mkdir -p $OUT; cd $R
for b in 0.3 0.5 0.7; do SID_DIST_TAIL_BLEND=$b timeout 400 python3 tools/shard_sim.py 8 > $OUT/shard_sim_8_blend_$b.json 2>> $OUT/err.txt; python3 -c "
import json; d=json.load(open('$OUT/shard_sim_8_blend_$b.json')); v=list(d['schemes'].values())[1]; print('$b', d['full_step_kernels_only_ms'], v['points_per_rank'], v['kernels_only_ms_per_rank'], v['kernels_only_slowest_ms'])"; done
