cd $GRAFT_REPO_ROOT
for g in 100000 500000 2000000; do echo "PCC_SMALL_ROWS=$g"; PCC_SMALL_ROWS=$g python3 tools/bench_conv.py --cases stride2,stride1 2>&1 | grep -E "conv " ; done
python3 tools/bench_first.py 2>&1 | grep "first layer"
