#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the widening-row kernels (detector, matcher, staging, first guess) with a
# byte AND an instruction roofline per kernel (through gpurun): gpurun_out/<tag>/.  Usage: bash tools/r4_widening_profiles.sh r04_widening
# The per-thread VALU instruction counts come from the ISA of the shipped sources (counted here with hipcc -S; the
# kernels named below are straight-line per thread, loops unrolled).
set -u
TAG=${1:-r04_widening}
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only"
for f in orb first_guess; do /opt/rocm/bin/hipcc $FLAGS -o /tmp/isa_$f.s $R/sea_ice_drift_amd/csrc/$f.hip 2>/dev/null; done
run() {   # name, program...
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d /tmp/kt_${TAG}_$name -o kt -- "$@" > $OUT/${name}_bench.json 2> $OUT/${name}.err
  python3 $R/tools/rocpd_summary.py $(find /tmp/kt_${TAG}_$name -name "*.db" | head -1) k_ > $OUT/${name}_kernel_trace_stats.txt
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
        if ln.startswith('#') or ln.startswith('name'):
            continue
        m = re.match(r'^(.{60,}?)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s*$', ln.rstrip())
        if m:
            rows.append((m.group(1).strip(), int(m.group(2)), int(m.group(3)), int(m.group(4))))
    return rows
def dispatches(name, kern):
    """(grid threads, duration ns) of every dispatch of `kern` (rocpd_summary's dispatch table, filter 'k_')."""
    rows, on = [], False
    for ln in open(os.path.join(out, name + '_kernel_trace_stats.txt')):
        if ln.startswith('# DISPATCHES'):
            on = True; continue
        if ln.startswith('# PMC'):
            on = False
        p = ln.split()
        if on and len(p) == 8 and p[0].isdigit():
            rows.append((int(p[1]), int(p[7])))
    return rows
def isa_valu(path, kern):
    """VALU / vector-memory / LDS instructions of one thread of `kern` (static count of its ISA)."""
    on, v, m, l = False, 0, 0, 0
    for ln in open(path):
        if re.match(r'^_Z.*' + kern + r'.*:', ln): on = True
        if on:
            t = ln.strip()
            if t.startswith('v_'): v += 1
            elif t.startswith(('global_', 'buffer_', 'flat_')): m += 1
            elif t.startswith('ds_'): l += 1
            if t.startswith('s_endpgm'): break
    return v, m, l
HBM = 8000.0                                 # GB/s
ISSUE = 2.4e9 * 256 * 4 / 4                  # VALU wave-instructions per second of the whole chip (one per 4 clk and SIMD)
res = {}
px = 1e8
for k, calls, total, avg in kernels('stage'):
    if 'hist_kernel' in k: res.setdefault('stage.hip hist_kernel', []).append({'avg_ns': avg, 'calls': calls, 'bytes_per_launch': 4 * px, 'GBps': 4 * px / avg, 'frac_of_8TBps': 4 * px / avg / HBM, 'bound': 'hbm'})
    if 'range_kernel' in k: res.setdefault('stage.hip range_kernel', []).append({'name': k[:60], 'avg_ns': avg, 'calls': calls, 'bytes_per_launch': 4 * px, 'GBps': 4 * px / avg, 'frac_of_8TBps': 4 * px / avg / HBM, 'bound': 'hbm'})
    if 'sample_kernel' in k: res.setdefault('stage.hip sample_kernel', []).append({'avg_ns': avg, 'calls': calls, 'bound': 'latency (one workgroup: 8192 scattered pixels, three-digit radix select in LDS)'})
    if 'scale_kernel' in k: res.setdefault('stage.hip scale_kernel', []).append({'avg_ns': avg, 'calls': calls, 'bytes_per_launch': 5 * px, 'GBps': 5 * px / avg, 'frac_of_8TBps': 5 * px / avg / HBM, 'bound': 'hbm'})
for k, calls, total, avg in kernels('ft_match'):
    pairs = 24183.0 * 22694.0
    if 'ft_knn2_mfma' in k:
        res.setdefault('ft_match.hip ft_knn2_mfma', []).append({'avg_ns': avg, 'calls': calls, 'int8_macs_per_launch': pairs * 256, 'TOPs': 2 * pairs * 256 / avg / 1e3,
            'frac_of_5POPs_int8': 2 * pairs * 256 / (avg * 1e-9) / 5e15, 'valu_lane_ops_per_launch': pairs * 4, 'frac_of_valu_issue_rate': pairs * 4 / 64 / (avg * 1e-9) / ISSUE,
            'bound': 'mfma + valu (one issue budget per SIMD, DESIGN.md section 6.2)'})
# detector: one thread per pixel of a pyramid level; the dispatch table has every level's grid.  Algorithmic bytes per pixel:
# k_fast reads the level and writes the score map (2 B), k_nms reads the score map (1 B; candidates are a few per cent),
# k_blur reads and writes the level (2 B), k_resize reads ~1.44 source pixels per output pixel (bilinear, cached) + writes 1.
BYTES = {'k_fast': 2.0, 'k_nms': 1.0, 'k_blur': 2.0, 'k_resize': 2.44}
for kern in ('k_fast', 'k_nms', 'k_blur', 'k_resize'):
    v, m, l = isa_valu('/tmp/isa_orb.s', kern)
    rows = [(g, d) for (g, d) in dispatches('orb', kern)]
    # rocpd_summary filters dispatches by name: one pass per kernel
    os.system('python3 %s/tools/rocpd_summary.py $(find /tmp/kt_%s_orb -name "*.db" | head -1) %s > /tmp/disp_%s.txt' % (os.environ.get('GRAFT_REPO_ROOT', os.getcwd()), os.path.basename(out), kern, kern))
    rows = []
    on = False
    for ln in open('/tmp/disp_%s.txt' % kern):
        if ln.startswith('# DISPATCHES'): on = True; continue
        p = ln.split()
        if on and len(p) == 8 and p[0].isdigit(): rows.append((int(p[1]), int(p[7])))
    if not rows: continue
    thr = sum(g for g, d in rows); ns = sum(d for g, d in rows)
    # k_nms handles 16 rows per thread: pixels = threads x 16
    pix = thr * (16 if kern == 'k_nms' else 1)
    per_thread_valu = v
    res['orb.hip ' + kern] = {'dispatches': len(rows), 'total_ns': ns, 'pixels': pix, 'algorithmic_bytes': BYTES[kern] * pix,
        'GBps': BYTES[kern] * pix / ns, 'frac_of_8TBps': BYTES[kern] * pix / ns / HBM,
        'isa_per_thread': {'valu': v, 'vmem': m, 'lds': l}, 'valu_wave_instructions': thr / 64.0 * per_thread_valu,
        'frac_of_valu_issue_rate': thr / 64.0 * per_thread_valu / (ns * 1e-9) / ISSUE,
        'bound': 'valu' if thr / 64.0 * per_thread_valu / ISSUE > BYTES[kern] * pix / (HBM * 1e9) else 'hbm',
        'note': 'static ISA count x threads (branches taken once; k_nms: the 16-row loop body is in the count once per row as unrolled by the compiler)'}
# first guess: k_locate_grid - one thread per query, ~a dozen 64-byte simplex records from L2 per query
os.system('python3 %s/tools/rocpd_summary.py $(find /tmp/kt_%s_first_guess -name "*.db" | head -1) k_locate > /tmp/disp_loc.txt' % (os.environ.get('GRAFT_REPO_ROOT', os.getcwd()), os.path.basename(out)))
rows, on = [], False
for ln in open('/tmp/disp_loc.txt'):
    if ln.startswith('# DISPATCHES'): on = True; continue
    p = ln.split()
    if on and len(p) == 8 and p[0].isdigit(): rows.append((int(p[1]), int(p[7])))
if rows:
    thr = sum(g for g, d in rows); ns = sum(d for g, d in rows)
    v, m, l = isa_valu('/tmp/isa_first_guess.s', 'k_locate_grid')
    res['first_guess.hip k_locate_grid'] = {'dispatches': len(rows), 'total_ns': ns, 'avg_ns': ns / len(rows), 'queries': thr,
        'algorithmic_bytes': thr * (16 + 8 + 12 * 64.0), 'GBps': thr * (16 + 8 + 12 * 64.0) / ns, 'frac_of_8TBps': thr * (16 + 8 + 12 * 64.0) / ns / HBM,
        'isa_per_thread_static': {'valu': v, 'vmem': m}, 'bound': 'latency (a dozen dependent 64-byte L2 reads per query; 1.7 ms as a brute-force pass in round 3)',
        'note': 'bytes: query (16) + result (8) + ~12 simplex records of 64 B per query (the bucket of its cell)'}
for nm in ('orb', 'first_guess'):
    for k, calls, total, avg in kernels(nm):
        res.setdefault('times ' + nm, {})[k[:60]] = {'avg_ns': avg, 'calls': calls, 'total_ns': total}
json.dump(res, open(os.path.join(out, 'rooflines.json'), 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.startswith('times')}, indent=1)[:4000])
PY
ls $OUT
