#!/bin/bash
# round 5, third GPU call: tests of the new code, kept accumulators A/B, what recycled blocks would buy (timing experiment),
# per-phase instruction counters, the shard simulation with side-by-side launches
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5_third; mkdir -p $OUT; cd $R
timeout 1200 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_first_guess.py tests/test_gpu_golden.py -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
one() {  # lib env check cfg...
  lib=$1; envs=$2; chk=$3; shift 3
  env $envs SID_PM_LIB=$R/build/ab/lib_$lib.so timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also-defaults --check $chk "$@" 2>>$OUT/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib $envs [$*]: %.4f ms  kernel %.4f ms  ok %s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d.get('parity_check', {}).get('ok')))" | tee -a $OUT/ab.txt
}
for round in 1 2; do
  for cfg in "--angles 1 --img-size 35" "--angles 1 --img-size 35 --border 20" "--angles 1 --img-size 35 --border 30" "--angles 1 --img-size 35 --border 40" "--angles 3" "--angles 3 --border 20"; do
    one new "X=1" 2000 $cfg
    one new "SID_PM_KEEP_ACC=0" 2000 $cfg
    one new "SID_PM_EXPERIMENT_WRAP_BLOCKS=2048" 0 $cfg
  done
  one new "X=1" 2000
  one new "SID_PM_EXPERIMENT_WRAP_BLOCKS=2048" 0
  one new "SID_PM_EXPERIMENT_WRAP_BLOCKS=4096" 0
  one new "X=1" 2000 --border 20
  one new "SID_PM_EXPERIMENT_WRAP_BLOCKS=2048" 0 --border 20
done
for w in 8 4; do
  SID_PM_SIDE_BY_SIDE=0 timeout 600 python3 tools/shard_sim.py $w > $OUT/shard_sim_${w}_sequential.json 2>>$OUT/err.txt
  timeout 600 python3 tools/shard_sim.py $w > $OUT/shard_sim_${w}_side_by_side.json 2>>$OUT/err.txt
done
grep -h "slowest_ms\|full_step" $OUT/shard_sim_*.json
# per-phase instruction counters (nested ablations): 15 angles and the reference's defaults, border 20
cd /tmp && export TMPDIR=/tmp
for cfgname in k15 k3; do
  if [ $cfgname = k15 ]; then CFG="--border 20"; else CFG="--border 20 --angles 1 --img-size 35"; fi
  for v in new A1 A2 A6 A3 A4 A5; do
    export SID_PM_LIB=$R/build/ab/lib_$v.so
    rm -rf /tmp/p1_$v
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/p1_$v -o p1 -- python3 $R/bench.py $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-also-defaults --check 0 > /dev/null 2>&1
    echo "== $cfgname $v" >> $OUT/pmc_b20.txt
    python3 $R/tools/rocpd_summary.py $(find /tmp/p1_$v -name "*.db" | head -1) 2>&1 | sed -n '/PMC per dispatch/,$p' >> $OUT/pmc_b20.txt
  done
done
unset SID_PM_LIB
