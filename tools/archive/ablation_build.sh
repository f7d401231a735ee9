#!/bin/bash
# Nested phase ablations of the PM kernels -> tools/ab/lib_A0.so (full) .. lib_A4.so (measurement only; see tools/ablation_run.sh)
cd "$(dirname "$0")/../sea_ice_drift_amd/csrc" && mkdir -p ../../tools/ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -fvisibility=hidden"
SRC="pm_kernel_mfma.hip pm_capi.hip ft_match.hip stage.hip orb.hip first_guess.hip"
D1="-DSID_ABLATE_HESSIAN"; D2="$D1 -DSID_ABLATE_WINNER"; D3="$D2 -DSID_ABLATE_SWEEP"; D4="$D3 -DSID_ABLATE_SUMS"
/opt/rocm/bin/hipcc $FLAGS -shared -o ../../tools/ab/lib_A0.so $SRC &
/opt/rocm/bin/hipcc $FLAGS $D1 -shared -o ../../tools/ab/lib_A1.so $SRC &
/opt/rocm/bin/hipcc $FLAGS $D2 -shared -o ../../tools/ab/lib_A2.so $SRC &
/opt/rocm/bin/hipcc $FLAGS $D3 -shared -o ../../tools/ab/lib_A3.so $SRC &
/opt/rocm/bin/hipcc $FLAGS $D4 -shared -o ../../tools/ab/lib_A4.so $SRC &
/opt/rocm/bin/hipcc $FLAGS $D4 -DSID_ABLATE_TPL -shared -o ../../tools/ab/lib_A5.so $SRC &
/opt/rocm/bin/hipcc $FLAGS $D2 -DSID_ABLATE_SCORE -shared -o ../../tools/ab/lib_A6.so $SRC &
wait
