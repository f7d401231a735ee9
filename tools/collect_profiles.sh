#!/bin/bash
# Collects the round's measurement evidence on the GPU box into gpurun_out/<tag>/ (copy what is to be judged
# into profiles/).  Usage (through gpurun): bash tools/collect_profiles.sh r01_final
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
for cfg in "--border 20" "--border 50" "--angles 3" "--border 20 --angles 3" "--img-size 35"; do
  python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline $cfg 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'args': '$cfg', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'workload': d['config']['workload'], 'mfma_frac': d['roofline']['frac'], 'parity_check': d['parity_check']}))" >> $OUT/other_configs.jsonl
done
rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o kt -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/rocpd_summary.py $(find /tmp/kt_$TAG -name "*.db" | head -1) > $OUT/kernel_trace_stats.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$TAG -o pf -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$TAG -o pw -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $(find /tmp/pf_$TAG -name "*.db" | head -1) $(find /tmp/pw_$TAG -name "*.db" | head -1) pm_kernel > $OUT/pmc_traffic.json
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -d /tmp/p1_$TAG -o p1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d /tmp/p2_$TAG -o p2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for p in p1 p2; do python3 $R/tools/rocpd_summary.py $(find /tmp/${p}_$TAG -name "*.db" | head -1) | sed -n '/PMC per dispatch/,$p' >> $OUT/pmc_counters.txt; done
ls -la $OUT
