#!/bin/bash
# What the S3 kernels wait on: two rocprofv3 --pmc passes (SQ issue / wait cycles; LDS instruction and conflict counts) over
# a short S3 run, summarised per kernel by tools/limiter_summary.py -> profiles/<tag>_limiters_S3.csv
#   tools/limiter_pass.sh r03c [S3]
set -e
tag=${1:-r03}
wl=${2:-S3}
root=$(pwd)
export TMPDIR=/tmp
rm -rf gpurun_out/lim_${wl}_a gpurun_out/lim_${wl}_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES \
    --kernel-trace -d $root/gpurun_out/lim_${wl}_a --output-format csv -- python3 $root/bench.py --workload $wl --steps 3 --warmup 2 \
    --no-cpu --no-scale --no-extras > gpurun_out/lim_${wl}_a.log 2>&1
echo "pass a done"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU \
    --kernel-trace -d $root/gpurun_out/lim_${wl}_b --output-format csv -- python3 $root/bench.py --workload $wl --steps 3 --warmup 2 \
    --no-cpu --no-scale --no-extras > gpurun_out/lim_${wl}_b.log 2>&1
echo "pass b done"
python3 tools/limiter_summary.py gpurun_out $tag $wl
