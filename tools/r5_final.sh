#!/bin/bash
# round 5 final evidence (through gpurun): the whole GPU suite on the shipped library, bench + profiles, shard simulation
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r05_final; mkdir -p $OUT; cd $R
md5sum sea_ice_drift_amd/libsid_pm.so > $OUT/lib_md5.txt
timeout 1700 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "rc $?" >> $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
bash tools/collect_profiles.sh r05_final > $OUT/collect.log 2>&1
for w in 8 4; do timeout 400 python3 tools/shard_sim.py $w > $OUT/shard_sim_$w.json 2>> $OUT/err.txt; done
for a in 7 1; do SID_PHASE_ANGLES=$a SID_PHASE_BORDERS=20,30 timeout 200 python3 tools/phase_cycles.py >> $OUT/phase_cycles.txt 2>&1; done
ls -la $OUT | head -60
