#!/bin/bash
# full GPU test suite + the default bench line (through gpurun); output under gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r4_full}; mkdir -p $OUT; cd $R
timeout 3000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt
SID_PM_VERBOSE=1 timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-600 $OUT/bench.json; grep "sid_pm: launch" $OUT/bench.err | sort | uniq -c; tail -3 $OUT/bench.err
