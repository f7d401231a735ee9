#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the widening-row kernels (detector, matcher, staging, first guess) with a
# byte / op roofline per kernel (through gpurun): gpurun_out/<tag>/.  Usage: bash tools/r3_widening_profiles.sh r03_widening
set -u
TAG=${1:-r03_widening}
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, program...
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d /tmp/kt_${TAG}_$name -o kt -- "$@" > $OUT/${name}_bench.json 2> $OUT/${name}.err
  python3 $R/tools/rocpd_summary.py $(find /tmp/kt_${TAG}_$name -name "*.db" | head -1) > $OUT/${name}_kernel_trace_stats.txt
}
run stage python3 $R/tools/stage_bench.py
run ft_match python3 $R/tools/ft_bench.py
run orb python3 $R/tools/ft_profile.py
run first_guess python3 $R/tools/prelude_profile.py
python3 - $OUT <<'PY'
import json, re, sys, os
out = sys.argv[1]
def kernels(name):
    rows = []
    for ln in open(os.path.join(out, name + '_kernel_trace_stats.txt')):
        m = re.match(r'^(.{70,}?)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s*$', ln.rstrip())
        if m and not ln.startswith('name'):
            rows.append((m.group(1).strip(), int(m.group(2)), int(m.group(3)), int(m.group(4))))
    return rows
HBM, VALU32 = 8000.0, 2.4e9 * 256 * 64 * 4 / 4      # GB/s; 32-bit lane-ops/s (256 CUs x 4 SIMDs x 16 lanes/clk)
res = {}
px = 1e8
# staging: 10000x10000 float32: a histogram pass reads 4 B/px, the map reads 4 and writes 1
for k, calls, total, avg in kernels('stage'):
    if 'hist_kernel' in k: res.setdefault('stage.hip hist_kernel', []).append({'avg_ns': avg, 'calls': calls, 'bytes_per_launch': 4 * px, 'GBps': 4 * px / avg, 'frac_of_8TBps': 4 * px / avg / HBM})
    if 'scale_kernel' in k: res.setdefault('stage.hip scale_kernel', []).append({'avg_ns': avg, 'calls': calls, 'bytes_per_launch': 5 * px, 'GBps': 5 * px / avg, 'frac_of_8TBps': 5 * px / avg / HBM})
# matcher: 24183 x 22694 pairs.  MFMA form: 256 int8 MACs per pair on the matrix pipe (dense int8 peak 5 POP/s = 2.5e15 MAC/s)
# + 4 VALU lane-ops per pair for the top two (key by one v_mad_i32_i24, three min / max); one-thread-per-query form: 21 lane-ops
for k, calls, total, avg in kernels('ft_match'):
    pairs = 24183.0 * 22694.0
    if 'ft_knn2_mfma' in k:
        res.setdefault('ft_match.hip ft_knn2_mfma', []).append({'avg_ns': avg, 'calls': calls, 'int8_macs_per_launch': pairs * 256, 'TOPs': 2 * pairs * 256 / avg / 1e3,
            'frac_of_5POPs_int8': 2 * pairs * 256 / (avg * 1e-9) / 5e15, 'valu_lane_ops_per_launch': pairs * 4, 'frac_of_valu_issue_rate': pairs * 4 / (avg * 1e-9) / VALU32,
            'note': 'the two pipes of a SIMD barely overlap (DESIGN.md section 6.2): the sum of the two fractions is the figure of merit'})
    elif 'ft_knn2_partial' in k:
        ops = pairs * 21
        res.setdefault('ft_match.hip ' + k[:40], []).append({'avg_ns': avg, 'calls': calls, 'lane_ops_per_launch': ops, 'Glaneops_per_s': ops / avg, 'frac_of_valu_issue_rate': ops / (avg * 1e-9) / VALU32})
# detector / first guess: per-kernel times (bytes models in DESIGN.md)
for nm in ('orb', 'first_guess'):
    for k, calls, total, avg in kernels(nm):
        if 'sid::' in k or k.startswith('k_') or 'k_' in k[:20]:
            res.setdefault(nm + ' ' + k[:60], []).append({'avg_ns': avg, 'calls': calls, 'total_ns': total})
json.dump(res, open(os.path.join(out, 'rooflines.json'), 'w'), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
ls $OUT
