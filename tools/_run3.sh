cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_octree3.py tests/test_gpu_codec.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r04n_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r04n_tests.log
[ $rc -eq 0 ] || exit $rc
STEPS=30 bash tools/ab_env.sh PCC_OCTREE_V3 0 1 2>&1 | grep -v "^    " | tee gpurun_out/r04n_ab_v3.txt
grep -E "stages ms" /tmp/ab_env.log | tail -2
