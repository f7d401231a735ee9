#!/bin/bash
# PMC counters of the reference's default configuration (35 px, angles [-3, 0, 3]) at border 20 and mixed: what bounds it?
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_pmc_defaults; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "b20:--border 20" "mixed:"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set -d /tmp/pd_${tag}_$i -o pd -- python3 $R/bench.py --angles 1 --img-size 35 $args --steps 2 --warmup 1 --no-cpu-baseline --no-also-defaults --check 0 > /dev/null 2>&1
    python3 $R/tools/rocpd_summary.py $(find /tmp/pd_${tag}_$i -name "*.db" | head -1) | sed -n '/PMC per dispatch/,$p' >> $OUT/pmc_defaults_$tag.txt
  done
done
ls -la $OUT
