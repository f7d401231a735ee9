#!/bin/bash
# HBM traffic of the default bench step by PMC (FETCH_SIZE x2 + WRITE_SIZE; separate passes), with whatever environment the caller
# sets (through gpurun): gpurun_out/<tag>/traffic.txt
TAG=${1:-r4_traffic}; R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$TAG -o pf -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also-defaults --check 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$TAG -o pw -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also-defaults --check 0 > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $(find /tmp/pf_$TAG -name "*.db" | head -1) $(find /tmp/pw_$TAG -name "*.db" | head -1) pm_kernel > $OUT/pmc_traffic.json
python3 - $OUT/pmc_traffic.json <<'PY' | tee $OUT/traffic.txt
import json, sys
raw = json.load(open(sys.argv[1]))
for k, v in raw.items():
    print('%-72s dispatches %3d  fetch %8.1f MB/launch  write %8.1f MB/launch' % (k[:72], v['dispatches'], v.get('fetch_kb', 0) * 2 / 1024 / v['dispatches'], v.get('write_kb', 0) / 1024 / v['dispatches']))
steps = 6
print('per step: fetch %.1f MB, write %.1f MB' % (sum(v.get('fetch_kb', 0) for v in raw.values()) * 2 / 1024 / steps, sum(v.get('write_kb', 0) for v in raw.values()) / 1024 / steps))
PY
