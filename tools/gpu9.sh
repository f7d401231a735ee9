R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2i; mkdir -p $OUT
cd $R
SID_PM_DEBUG_CHECK=1 SID_PM_LIB=$R/tools/ab/lib_chk.so timeout 300 python3 tools/soak_debug.py 20 2>&1 | grep -v amdgpu.ids | grep "first event" | head -3 | cut -c1-3000 > $OUT/race.txt
cat $OUT/race.txt
