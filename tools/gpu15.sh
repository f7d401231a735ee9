R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2o; mkdir -p $OUT
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_soak.py -m gpu -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
for cfg in "" "--border 20" "--border 50" "--img-size 35"; do
timeout 600 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline $cfg 2>> $OUT/bench.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'args': '$cfg', 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms_per_step'], 'value': d['value'], 'mfma_frac': d['roofline']['frac'], 'parity_check': d['parity_check']['ok']}))" >> $OUT/configs.jsonl
done
cat $OUT/configs.jsonl
python3 tools/phase_cycles.py 2>&1 | head -11
