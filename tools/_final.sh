set -o pipefail
mkdir -p gpurun_out
python tools/bench_small_conv.py > gpurun_out/r16f_new.log 2>&1; grep -E "stride" gpurun_out/r16f_new.log
python -m pytest tests -m gpu -x -q > gpurun_out/r04v_gputests.log 2>&1; rc=$?; tail -3 gpurun_out/r04v_gputests.log; [ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04v_smoke.log 2>&1 && echo smoke ok || exit 1
bash tools/profile_round.sh r04v > gpurun_out/r04v_profile_round.log 2>&1; echo "profile_round rc $?"; tail -5 gpurun_out/r04v_profile_round.log | cut -c1-300
cd $GRAFT_REPO_ROOT && python bench.py --steps 20 --warmup 5 > gpurun_out/r04v_bench_default_line.json 2> gpurun_out/r04v_bench_default.err; echo "bench rc $?"; python tools/line.py gpurun_out/r04v_bench_default_line.json 2>/dev/null | head -20
