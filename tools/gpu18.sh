R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2q; mkdir -p $OUT
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
timeout 600 python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
SID_PM_PHASES=1 timeout 300 python3 tools/phase_cycles.py > $OUT/phases.txt 2>&1; tail -25 $OUT/phases.txt
