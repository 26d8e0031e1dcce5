cd $GRAFT_REPO_ROOT
STEPS=20 bash tools/ab_env.sh PCC_FIRST_WGS 0 1024 2048 512 2>&1 | grep -E "PCC_FIRST|1000000, 4" | tee gpurun_out/r04p_ab_first.txt
