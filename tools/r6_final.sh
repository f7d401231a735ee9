#!/bin/bash
# Round 6, final pass on the shipped library: the whole GPU test suite (log kept), the PMC traffic of the headline step (profiles/traffic.json
# carries the library's md5), the default bench line.  Usage (through gpurun): bash tools/r6_final.sh
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_last; mkdir -p $OUT; cd $R
timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d /tmp/lf -o pf -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/lw -o pw -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $(find /tmp/lf -name "*.db" | head -1) $(find /tmp/lw -name "*.db" | head -1) pm_kernel > $OUT/pmc_traffic.json
python3 - "$OUT/pmc_traffic.json" "$R" > $OUT/traffic.json <<'PY'
import hashlib, json, sys
raw = json.load(open(sys.argv[1])); root = sys.argv[2]
old = json.load(open(root + '/profiles/traffic.json'))
fetch_kb = sum(v.get('fetch_kb', 0.0) for v in raw.values()); write_kb = sum(v.get('write_kb', 0.0) for v in raw.values())
disp = sum(v.get('dispatches', 0) for v in raw.values()); steps = 6
fetch = fetch_kb * 1024.0 * 2.0 / steps; write = write_kb * 1024.0 / steps; launches = disp / steps
old.update({'so_md5': hashlib.md5(open(root + '/sea_ice_drift_amd/libsid_pm.so', 'rb').read()).hexdigest(), 'launches_per_step': launches,
            'fetch_bytes_per_step_raw': fetch / 2.0, 'fetch_bytes_per_step': fetch, 'write_bytes_per_step': write,
            'hbm_bytes_per_step': fetch + write, 'hbm_bytes_per_launch': (fetch + write) / max(launches, 1)})
print(json.dumps(old, indent=1))
PY
cd $R && cp $OUT/traffic.json profiles/traffic.json
python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/pytest_gpu.txt; python3 -c "
import json; d = json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_note'][:120])"
