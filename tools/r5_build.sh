#!/bin/bash
# A/B library builds (round 5): bash tools/r5_build.sh name "-DFLAG ..." [name2 "-D..."] ...  ->  build/ab/lib_<name>.so
# Only the two PM translation units are recompiled per variant; the other objects come from build/csrc (make first).
set -e
cd "$(dirname "$0")/../sea_ice_drift_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -fvisibility=hidden"
O=../../build/csrc; AB=../../build/ab; mkdir -p $AB $O/ab
make -s -j8 >/dev/null
while [ $# -gt 0 ]; do
  name=$1; defs=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c -o $O/ab/${name}_mfma.o pm_kernel_mfma.hip &
    /opt/rocm/bin/hipcc $FLAGS $defs -c -o $O/ab/${name}_occ4.o pm_kernel_rp_occ4.hip &
    /opt/rocm/bin/hipcc $FLAGS $defs -c -o $O/ab/${name}_capi.o pm_capi.hip &
    wait
    /opt/rocm/bin/hipcc $FLAGS -shared -o $AB/lib_${name}.so $O/ab/${name}_mfma.o $O/ab/${name}_occ4.o $O/ab/${name}_capi.o $O/pm_large.o $O/ft_match.o $O/stage.o $O/orb.o $O/first_guess.o
    echo "built $AB/lib_${name}.so" ) &
  # two variants at a time (8 cores, ~3 compilers each)
  if [ $(jobs -r | wc -l) -ge 2 ]; then wait -n; fi
done
wait
