set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 tools/step_detail.py 1 > gpurun_out/r04h_detail_v1.txt 2>&1 || exit 1
python3 tools/step_detail.py 0 > gpurun_out/r04h_detail_v0.txt 2>&1 || exit 1
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/r04h_cl
PCC_CONTAINER_VERSION=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04h_cl -- python3 $R/tools/count_launches.py > $R/gpurun_out/r04h_cl.log 2>&1 || exit 1
python3 $R/tools/count_launches.py --report $R/gpurun_out/r04h_cl > $R/gpurun_out/r04h_launches.txt
head -3 $R/gpurun_out/r04h_launches.txt
cat $R/gpurun_out/r04h_detail_v1.txt
