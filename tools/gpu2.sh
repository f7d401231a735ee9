R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2b; mkdir -p $OUT
cd $R
# --- race reconstruction: commit 7758c91 (monolithic template sampler) with the sweep inlined ---
for v in race_orig race0 race1; do
  echo "== $v table sampler" >> $OUT/race.txt
  SID_PM_LIB=$R/tools/ab/lib_$v.so timeout 300 python3 tools/determinism_check.py 4000 150 >> $OUT/race.txt 2>&1
  echo "== $v on-the-fly sampler" >> $OUT/race.txt
  SID_PM_LIB=$R/tools/ab/lib_$v.so timeout 300 python3 tools/determinism_check.py 4000 150 --no-table >> $OUT/race.txt 2>&1
done
for pad in 1280 2560 5120; do
  echo "== race0 LDS pad $pad" >> $OUT/race.txt
  SID_PM_LDS_PAD=$pad SID_PM_LIB=$R/tools/ab/lib_race0.so timeout 300 python3 tools/determinism_check.py 4000 150 >> $OUT/race.txt 2>&1
done
# --- the new build ---
timeout 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_configs.py::test_config5_stream_16_pairs_full_size > $OUT/pytest.txt 2>&1
tail -30 $OUT/pytest.txt
timeout 600 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
timeout 600 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --border 20 > $OUT/bench_b20.json 2>> $OUT/bench.err
SID_PM_NO_RP=1 timeout 600 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --border 20 > $OUT/bench_b20_classic.json 2>> $OUT/bench.err
python3 tools/phase_cycles.py > $OUT/phase_cycles.txt 2>&1
cat $OUT/race.txt; cat $OUT/bench.json $OUT/bench_b20.json | cut -c1-400; tail -5 $OUT/bench.err; cat $OUT/phase_cycles.txt
