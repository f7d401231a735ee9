R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2f; mkdir -p $OUT
cd $R
echo "== chk build (consistency check of the operand table after sampling / after the sweep)" >> $OUT/race.txt
SID_PM_DEBUG_CHECK=1 SID_PM_LIB=$R/tools/ab/lib_chk.so timeout 300 python3 tools/soak_debug.py 20 >> $OUT/race.txt 2>&1
for pad in 256 512 1024; do
  echo "== v0 pad $pad (3 per CU kept)" >> $OUT/race.txt
  SID_PM_LDS_PAD=$pad SID_PM_LIB=$R/tools/ab/lib_v0.so timeout 300 python3 tools/soak_debug.py 20 2>&1 | tail -1 >> $OUT/race.txt
done
cat $OUT/race.txt | cut -c1-600
