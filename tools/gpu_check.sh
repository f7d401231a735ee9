#!/bin/bash
# Quick look on the GPU box after a kernel change (through gpurun): parity / golden / config tests, the default bench
# line, and the phase clock of single points.  Output under gpurun_out/check/.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/check; mkdir -p $OUT
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
timeout 600 python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 300 python3 tools/phase_cycles.py > $OUT/phases.txt 2>&1; tail -30 $OUT/phases.txt
