cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/soak.py 60 0 2>&1 | tail -2 || exit 1
timeout -k 10 300 python tools/soak.py 60 1 2>&1 | tail -2 || exit 1
timeout -k 10 300 python tools/soak_small.py 6 0 2>&1 | tail -2 || exit 1
timeout -k 10 300 python tools/soak_small.py 6 1 2>&1 | tail -2 || exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04o_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r04o_tests.log
exit $rc
