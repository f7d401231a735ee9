cd $GRAFT_REPO_ROOT
VARIANTS="A6" bash tools/ablation_run.sh r2abl3 20 2>&1 | tail -8
grep -E "INSTS_MFMA|INSTS_VALU |INSTS_LDS|INSTS_SALU" gpurun_out/r2abl3/pmc_b20.txt | head
