#!/bin/bash
# Round 6 experiment: image loads of the window phase non-temporal (-DSID_IMG_NT, build/ab/lib_imgnt.so) against the shipped library:
# step time alternating, then the PMC traffic of both.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_imgnt; mkdir -p $OUT; cd $R
AB_ROUNDS=2 bash tools/r6_ab.sh r06_imgnt sea_ice_drift_amd/libsid_pm.so build/ab/lib_imgnt.so 2>&1 | grep round
cd /tmp && export TMPDIR=/tmp
for v in base imgnt; do
  if [ $v = imgnt ]; then export SID_PM_LIB=$R/build/ab/lib_imgnt.so; else unset SID_PM_LIB; fi
  rocprofv3 --pmc FETCH_SIZE -d /tmp/nf_$v -o pf -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d /tmp/nw_$v -o pw -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-also-defaults --no-seam --check 0 > /dev/null 2>&1
  python3 $R/tools/pmc_traffic.py $(find /tmp/nf_$v -name "*.db" | head -1) $(find /tmp/nw_$v -name "*.db" | head -1) pm_kernel > $OUT/pmc_traffic_$v.json
  python3 -c "
import json
raw = json.load(open('$OUT/pmc_traffic_$v.json'))
f = sum(x.get('fetch_kb', 0.0) for x in raw.values()) * 1024 * 2 / 6; w = sum(x.get('write_kb', 0.0) for x in raw.values()) * 1024 / 6
print('$v: fetch %.1f MB (x2 corrected) + write %.1f MB = %.1f MB per step = %.2fx the algorithmic 203.7 MB' % (f / 1e6, w / 1e6, (f + w) / 1e6, (f + w) / 203.68e6))" | tee -a $OUT/ab.txt
done
