#!/bin/bash
# PMC counter groups over one bench step sequence (through gpurun): gpurun_out/<tag>/pmc_*.txt
set -u
TAG=${1:-r3_pmc}; shift
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64" \
           "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d /tmp/pmc_${TAG}_$i -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --check 0 "$@" > /dev/null 2> $OUT/err_$i.txt
  python3 $R/tools/rocpd_summary.py $(find /tmp/pmc_${TAG}_$i -name "*.db" | head -1) | sed -n '/PMC per dispatch/,$p' > $OUT/pmc_$i.txt
done
python3 - $OUT <<'PY'
import sys, glob, collections
out = sys.argv[1]
tab = collections.OrderedDict()
for f in sorted(glob.glob(out + '/pmc_*.txt')):
    for ln in open(f):
        p = ln.split()
        if len(p) == 6 and p[1].isdigit():
            # last step's three dispatches: keep the last occurrence per (counter, lds)
            tab[(p[0], p[3])] = (float(p[4]), float(p[5]))
lds = sorted({k[1] for k in tab}, key=int)
print('%-30s' % 'counter' + ''.join('%18s' % ('lds ' + l) for l in lds))
for c in collections.OrderedDict((k[0], 1) for k in tab):
    print('%-30s' % c + ''.join('%18.4g' % tab.get((c, l), (float('nan'),))[0] for l in lds))
print('%-30s' % 'duration_ns' + ''.join('%18.4g' % max(v[1] for k, v in tab.items() if k[1] == l) for l in lds))
PY
