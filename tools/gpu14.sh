R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2n; mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_orb.py tests/test_gpu_ft.py tests/test_ft_pipeline.py tests/test_gpu_stage.py -m gpu -q > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt
timeout 600 python3 tools/stage_bench.py > $OUT/stage_bench.json 2>$OUT/stage.err; cat $OUT/stage_bench.json; tail -3 $OUT/stage.err
timeout 900 python3 tools/ftpm_bench.py > $OUT/ftpm.json 2> $OUT/ftpm.err
cat $OUT/ftpm.json; tail -5 $OUT/ftpm.err
for nt in 256 384 512; do
SID_PM_THREADS2=$nt timeout 600 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --border 28 2>> $OUT/bench.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'threads2': $nt, 'border': 28, 'ms_per_step': d['ms_per_step'], 'parity_check': d['parity_check']['ok']}))" >> $OUT/threads2.jsonl
done
cat $OUT/threads2.jsonl
