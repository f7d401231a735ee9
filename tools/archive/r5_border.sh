#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r05_final; mkdir -p $OUT; cd $R
timeout 800 python3 tools/border_cost.py 15 > $OUT/border_cost_15.json 2>> $OUT/err.txt
timeout 800 python3 tools/border_cost.py 3 > $OUT/border_cost_3.json 2>> $OUT/err.txt
cat $OUT/border_cost_15.json $OUT/border_cost_3.json | cut -c1-700
