#!/bin/bash
# Round 6 experiment: recycled blocks (bitmap free lists) for EVERY launch that keeps per-placement tables in global memory
# (SID_PM_RECYCLE_ALL=1) against exclusive blocks (=0): step time alternating, then the HBM traffic of both by PMC.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_recycle_all; mkdir -p $OUT; cd $R
for round in 1 2 3; do for v in 0 1; do
  SID_PM_RECYCLE_ALL=$v timeout 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-seam --no-also-defaults --check 256 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('SID_PM_RECYCLE_ALL=$v round $round: %.4f ms  kernel %.4f ms  ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['parity_check']['ok']))" | tee -a $OUT/ab.txt
done; done
for cfg in "--border 20" "--angles 3"; do for v in 0 1; do
  SID_PM_RECYCLE_ALL=$v timeout 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-seam --no-also-defaults --check 256 $cfg 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('SID_PM_RECYCLE_ALL=$v [$cfg]: %.4f ms  kernel %.4f ms  ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['parity_check']['ok']))" | tee -a $OUT/ab.txt
done; done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export SID_PM_RECYCLE_ALL=$v
  rocprofv3 --pmc FETCH_SIZE -d /tmp/rf_$v -o pf -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d /tmp/rw_$v -o pw -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > /dev/null 2>&1
  python3 $R/tools/pmc_traffic.py $(find /tmp/rf_$v -name "*.db" | head -1) $(find /tmp/rw_$v -name "*.db" | head -1) pm_kernel > $OUT/pmc_traffic_$v.json
  python3 -c "
import json
raw = json.load(open('$OUT/pmc_traffic_$v.json'))
f = sum(x.get('fetch_kb', 0.0) for x in raw.values()) * 1024 * 2 / 6; w = sum(x.get('write_kb', 0.0) for x in raw.values()) * 1024 / 6
print('SID_PM_RECYCLE_ALL=$v: fetch %.1f MB (x2 corrected) + write %.1f MB = %.1f MB per step = %.2fx the algorithmic 203.7 MB' % (f / 1e6, w / 1e6, (f + w) / 1e6, (f + w) / 203.68e6))" | tee -a $OUT/ab.txt
done
