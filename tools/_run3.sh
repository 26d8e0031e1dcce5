cd $GRAFT_REPO_ROOT
STEPS=30 bash tools/ab_env.sh PCC_XYZ16 0 1 2>&1 | grep -v "^    " | tee gpurun_out/r04k_ab_xyz16.txt
