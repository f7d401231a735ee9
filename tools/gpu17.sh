R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2p; mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_first_guess.py tests/test_gpu_golden.py -m gpu -q > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
timeout 600 python3 tools/e2e_bench.py > $OUT/e2e.json 2> $OUT/e2e.err; cat $OUT/e2e.json; tail -3 $OUT/e2e.err
