#!/bin/bash
# round 5 final evidence (through gpurun): the whole GPU suite on the shipped library, the eviction soaks, bench + profiles
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r05_final; mkdir -p $OUT; cd $R
md5sum sea_ice_drift_amd/libsid_pm.so > $OUT/lib_md5.txt
timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_soak.py -x -q -k "evictions" > $OUT/pytest_evict.txt 2>&1; echo "rc $?" >> $OUT/pytest_evict.txt; tail -4 $OUT/pytest_evict.txt
timeout 3000 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "rc $?" >> $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
( cd tools/ubench; timeout 200 ./slot_life 100000 2048 40960 192 400; timeout 200 ./slot_life 40000 2048 40960 192 400 --evict ) 2>&1 | grep -v "after launch" > $OUT/slot_life.txt
bash tools/collect_profiles.sh r05_final > $OUT/collect.log 2>&1
for w in 8 4 2; do timeout 600 python3 tools/shard_sim.py $w > $OUT/shard_sim_$w.json 2>> $OUT/err.txt; done
timeout 900 python3 tools/border_cost.py 15 > $OUT/border_cost_15.json 2>> $OUT/err.txt
timeout 900 python3 tools/border_cost.py 3 > $OUT/border_cost_3.json 2>> $OUT/err.txt
for a in 7 1; do SID_PHASE_ANGLES=$a SID_PHASE_BORDERS=20,30 timeout 300 python3 tools/phase_cycles.py >> $OUT/phase_cycles.txt 2>&1; done
ls -la $OUT $R/gpurun_out/r05_final | head -60
