#!/bin/bash
# Every rocprofv3 pass the numbers in DESIGN.md / profiles/ come from, on the GPU box from the repo root:
#   tools/profile_round.sh <prefix>      -> gpurun_out/<prefix>_*   (copy what is to be kept into profiles/)
# Passes are separate runs (kernel trace + stats; FETCH_SIZE; WRITE_SIZE; three SQ passes), the program directly behind --.
set -u
P=${1:-rXX}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-psnr --inflight 0 --no-configs"
echo "== kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${P}_stats -- $B --steps 10 --warmup 3 > $O/${P}_stats.log 2>&1 || exit 1
cp $(ls $O/${P}_stats/*/*_kernel_stats.csv | head -1) $O/${P}_bench_kernel_stats.csv
grep '^{' $O/${P}_stats.log | tail -1 > $O/${P}_bench_line.json
python3 $R/tools/launches_by_grid.py $O/${P}_stats $O/${P}_launches_by_grid.json "kernel trace of the ${P} stats run (bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --inflight 0 --no-configs) grouped by kernel and grid size; durations in microseconds" > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== $c"
  rm -rf $O/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- $B --steps 2 --warmup 1 > $O/${P}_pmc_$c.log 2>&1 || exit 1
done
python3 $R/tools/pmc_summary.py $O $O/${P}_pmc_summary.json
n=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"; do
  n=$((n + 1))
  echo "== SQ pass $n"
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${P}_sq/p$n -- $B --steps 2 --warmup 1 > $O/${P}_sq_p$n.log 2>&1 || exit 1
done
python3 $R/tools/pmc_sq.py $O/${P}_sq "k_gconv_up<true>" > $O/${P}_pmc_sq_conv.json
cat $O/${P}_pmc_sq_conv.json | head -40
cat $O/${P}_bench_line.json | cut -c1-400
# the geometry-only configuration (BASELINE.json configs[2]): kernels of blob version 2 (octree2.hip)
echo "== C3 kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${P}_c3 -- python3 $R/tools/bench_c3.py > $O/${P}_c3.log 2>&1 || exit 1
cp $(ls $O/${P}_c3/*/*_kernel_stats.csv | head -1) $O/${P}_c3_kernel_stats.csv
python3 $R/tools/launches_by_grid.py $O/${P}_c3 $O/${P}_c3_launches_by_grid.json "kernel trace of tools/bench_c3.py (sweep + 1M-point room, blob versions 1 and 2)" > /dev/null
