R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2l; mkdir -p $OUT
cd $R
for v in noslp; do
echo "== $v" >> $OUT/race.txt
SID_PM_LIB=$R/tools/ab/lib_$v.so timeout 400 python3 tools/soak_debug.py 20 2>&1 | grep "total bad\|first run" | cut -c1-200 >> $OUT/race.txt
done
cat $OUT/race.txt
