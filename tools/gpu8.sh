R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2h; mkdir -p $OUT
cd $R
echo "== chk build" >> $OUT/race.txt
SID_PM_DEBUG_CHECK=1 SID_PM_LIB=$R/tools/ab/lib_chk.so timeout 300 python3 tools/soak_debug.py 20 2>&1 | grep -v amdgpu.ids | cut -c1-700 >> $OUT/race.txt
echo "== regular build" >> $OUT/race.txt
timeout 300 python3 tools/soak_debug.py 20 2>&1 | tail -1 >> $OUT/race.txt
cat $OUT/race.txt
