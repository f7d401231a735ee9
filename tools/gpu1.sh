R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2a; mkdir -p $OUT
$R/tools/ubench/blgp_probe > $OUT/blgp.txt 2>&1
$R/tools/ubench/lds_pair_b64 > $OUT/lds_pair.txt 2>&1
python3 $R/tools/phase_cycles.py > $OUT/phase_cycles.txt 2>&1
bash $R/tools/ablation_run.sh r2a 20
bash $R/tools/ablation_run.sh r2a mixed
cat $OUT/blgp.txt $OUT/lds_pair.txt $OUT/phase_cycles.txt
