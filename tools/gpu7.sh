R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2g; mkdir -p $OUT
cd $R
echo "== no-spill build" >> $OUT/race.txt
timeout 600 python3 tools/soak_debug.py 20 mixed >> $OUT/race.txt 2>&1
timeout 300 tools/ubench/scratch_stress >> $OUT/race.txt 2>&1
cat $OUT/race.txt | cut -c1-400
timeout 1800 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_configs.py::test_config5_stream_16_pairs_full_size > $OUT/pytest.txt 2>&1
tail -8 $OUT/pytest.txt
for cfg in "" "--border 20" "--border 50" "--img-size 35"; do
timeout 600 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline $cfg 2>> $OUT/bench.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'args': '$cfg', 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms_per_step'], 'value': d['value'], 'mfma_frac': d['roofline']['frac'], 'parity_check': d['parity_check']['ok']}))" >> $OUT/configs.jsonl
done
cat $OUT/configs.jsonl
python3 tools/phase_cycles.py 2>&1 | head -12
