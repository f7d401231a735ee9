#!/bin/bash
# Collects the round's measurement evidence on the GPU box into gpurun_out/<tag>/ (copy what is to be judged
# into profiles/).  Usage (through gpurun): bash tools/collect_profiles.sh r02_rp
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# HBM traffic first: bench.py quotes profiles/traffic.json and checks that it was collected on the library it runs
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$TAG -o pf -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$TAG -o pw -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $(find /tmp/pf_$TAG -name "*.db" | head -1) $(find /tmp/pw_$TAG -name "*.db" | head -1) pm_kernel > $OUT/pmc_traffic.json
python3 - "$OUT/pmc_traffic.json" "$R" > $OUT/traffic.json <<'PY'
import hashlib, json, sys
raw = json.load(open(sys.argv[1])); root = sys.argv[2]
fetch_kb = sum(v.get('fetch_kb', 0.0) for v in raw.values()); write_kb = sum(v.get('write_kb', 0.0) for v in raw.values())
disp = sum(v.get('dispatches', 0) for v in raw.values())
steps = 6                                   # --steps 4 --warmup 1 + the poisoned step the parity check reads
launches = disp / steps
fetch = fetch_kb * 1024.0 * 2.0 / steps     # FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md; calibration in profiles/r01_hbm_counter_calibration.json)
write = write_kb * 1024.0 / steps
print(json.dumps({
 'workload': {'size': 10000, 'grid': 200, 'angles': 7, 'border': 'mixed', 'img_size': 34},
 'so_md5': hashlib.md5(open(root + '/sea_ice_drift_amd/libsid_pm.so', 'rb').read()).hexdigest(),
 'launches_per_step': launches, 'fetch_bytes_per_step_raw': fetch / 2.0, 'fetch_correction': 2.0, 'fetch_bytes_per_step': fetch,
 'write_bytes_per_step': write, 'hbm_bytes_per_step': fetch + write, 'hbm_bytes_per_launch': (fetch + write) / max(launches, 1),
 'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also-defaults --check 0` (%d dispatches = 6 steps: warm-up, 4 timed, 1 for the parity check), summed with tools/pmc_traffic.py (tools/collect_profiles.sh)' % disp,
 'note': 'fetch: search windows, image-1 patches, the sampling table, and the per-placement sums (sum w^2, sum w) that the ~35 000 points of the gs launches - the three-wavefront class and borders 28 .. 50 - keep in a block of global memory and read back once or twice (from L2 when it still holds them); write: those sums, 8 B per placement, written once per step and point + register spills.  Since round 6 the blocks of EVERY such launch are recycled through per-XCD free lists (8192 blocks cycle instead of 40 000 write-once ones): the read-backs mostly hit L2 / the Infinity Cache (fetch 1.65 -> 1.14 GB per step), the writes still leave L2 (WRITE_SIZE counts them on their way to the memory side, 0.93 -> 0.91 GB); the results go to pinned host memory; algorithmic bytes = both images once (0.2 GB per step)'}, indent=1))
PY
cp $OUT/traffic.json $R/profiles/traffic.json
python3 $R/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
for cfg in "--border 20" "--border 30" "--border 50" "--angles 3" "--border 20 --angles 3" "--angles 1" "--border 20 --angles 1" "--img-size 35" "--img-size 35 --angles 1"; do
  python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline $cfg 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'args': '$cfg', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'workload': d['config']['workload'], 'mfma_frac': d['roofline']['frac'], 'parity_check': d['parity_check']}))" >> $OUT/other_configs.jsonl
done
python3 $R/bench.py --mode stream --pairs 16 --steps 2 --warmup 1 --check 32 > $OUT/stream_bench.json 2>> $OUT/bench.err
python3 $R/bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/dryrun_2ranks_1gpu.json 2>> $OUT/bench.err
python3 $R/bench.py --gpus 1 --force-collective --steps 20 --warmup 3 --no-cpu-baseline > $OUT/force_collective_rccl_1gpu.json 2>> $OUT/bench.err
timeout 300 python3 $R/bench.py --mode ftpm --check 400 > $OUT/ftpm_bench.json 2>> $OUT/bench.err
# kernel trace over 25 steps, stats over the last 20 (5 warm-up steps: the clocks ramp over the first dispatches), so that the
# sum of the average launch durations can be held against bench.py's kernel_ms_per_step of the SAME run (printed into the file)
rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > $OUT/kernel_trace_bench.json 2>/dev/null
python3 $R/tools/rocpd_summary.py $(find /tmp/kt_$TAG -name "*.db" | head -1) pm_kernel --window 5 20 > $OUT/kernel_trace_stats.txt
python3 -c "
import json; d = json.load(open('$OUT/kernel_trace_bench.json'))
print('# bench.py of the SAME traced run: ms_per_step %.4f, kernel_ms_per_step (HIP events over the 20 timed steps) %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step']))" >> $OUT/kernel_trace_stats.txt
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -d /tmp/p1_$TAG -o p1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d /tmp/p2_$TAG -o p2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_CYCLES -d /tmp/p3_$TAG -o p3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/p4_$TAG -o p4 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > /dev/null 2>&1
for p in p1 p2 p3 p4; do python3 $R/tools/rocpd_summary.py $(find /tmp/${p}_$TAG -name "*.db" | head -1) | sed -n '/PMC per dispatch/,$p' >> $OUT/pmc_counters.txt; done
timeout 200 python3 $R/tools/e2e_bench.py > $OUT/e2e_pattern_matching.json 2>> $OUT/bench.err
ls -la $OUT
