R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2j; mkdir -p $OUT
cd $R
echo "== regular build, simplified general sampler" >> $OUT/race.txt
timeout 600 python3 tools/soak_debug.py 20 mixed 2>&1 | grep "total bad\|first run" | cut -c1-300 >> $OUT/race.txt
echo "== chk build" >> $OUT/race.txt
SID_PM_DEBUG_CHECK=1 SID_PM_LIB=$R/tools/ab/lib_chk.so timeout 300 python3 tools/soak_debug.py 20 2>&1 | grep -v amdgpu.ids | grep "total bad\|DEBUG_CHECK" | head -5 | cut -c1-400 >> $OUT/race.txt
cat $OUT/race.txt
timeout 1800 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_configs.py::test_config5_stream_16_pairs_full_size > $OUT/pytest.txt 2>&1
tail -8 $OUT/pytest.txt
for cfg in "" "--border 20"; do
timeout 600 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline $cfg 2>> $OUT/bench.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'args': '$cfg', 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms_per_step'], 'value': d['value'], 'mfma_frac': d['roofline']['frac'], 'parity_check': d['parity_check']['ok']}))" >> $OUT/configs.jsonl
done
cat $OUT/configs.jsonl
