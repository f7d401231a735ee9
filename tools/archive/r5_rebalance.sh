#!/bin/bash
# measured feedback on the shard cuts: one-GPU simulation of the 8- and 4-rank splits + the 2-rank dry run of bench.py
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_rebalance; mkdir -p $OUT; cd $R
timeout 400 python3 tools/shard_sim.py 8 6 > $OUT/shard_sim_8_feedback.json 2> $OUT/err.txt
timeout 300 python3 tools/shard_sim.py 4 4 > $OUT/shard_sim_4_feedback.json 2>> $OUT/err.txt
timeout 300 python3 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/dryrun_2ranks_1gpu.json 2>> $OUT/err.txt
timeout 300 python3 -m pytest tests/test_gpu_golden.py tests/test_gpu_configs.py -m gpu -x -q -k "config3 or config5 or shard or rccl" 2>&1 | tail -4 > $OUT/pytest.txt
tail -3 $OUT/err.txt; cat $OUT/pytest.txt
python3 - <<'PY'
import json
for w in (8, 4):
    d = json.load(open('gpurun_out/r5_rebalance/shard_sim_%d_feedback.json' % w))
    print(w, 'full', d['full_step_kernels_only_ms'])
    for r in d['measured_feedback']['rounds']: print(r['points_per_rank'], r['kernels_only_ms_per_rank'], r['slowest_ms'])
    print('kept', d['measured_feedback']['kept'])
PY
