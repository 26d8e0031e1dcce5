set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04f_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r04f_tests.log
[ $rc -eq 0 ] || exit $rc
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/r04f_cl
PCC_CONTAINER_VERSION=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04f_cl -- python3 $R/tools/count_launches.py > $R/gpurun_out/r04f_cl.log 2>&1 || exit 1
python3 $R/tools/count_launches.py --report $R/gpurun_out/r04f_cl | tee $R/gpurun_out/r04f_launches.txt
cd $R
python3 bench.py --no-cpu-baseline --no-psnr --no-configs --inflight 0 --steps 20 --warmup 5 > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err || exit 1
python3 -c "
import json
d=json.loads(open('gpurun_out/r04f_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','value_hbm_resident','value_gpu_rans','value_seek_points')}, d['roofline']['frac'])
"
