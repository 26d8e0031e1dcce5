cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_codec.py tests/test_gpu_cabi.py tests/test_gpu_fullsize.py tests/test_tiled.py -m gpu -x -q > gpurun_out/r04l_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r04l_tests.log
[ $rc -eq 0 ] || exit $rc
STEPS=30 bash tools/ab_env.sh PCC_SIDE_BOOK 0 1 2>&1 | grep -v "^    " | tee gpurun_out/r04l_ab_book.txt
