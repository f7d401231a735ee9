#!/bin/bash
# quick correctness + speed look after a kernel change (through gpurun); output under gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r3_check}; mkdir -p $OUT
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -8 $OUT/pytest.txt
SID_PHASE_BORDERS=20,28,36,45 timeout 300 python3 tools/phase_cycles.py > $OUT/phases.txt 2>&1; cat $OUT/phases.txt
timeout 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err
for cfg in "--border 20" "--angles 3"; do timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 256 $cfg 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['parity_check']['ok'])"; done
