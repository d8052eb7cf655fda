#!/bin/bash
# The profiling passes behind profiles/<tag>_* (run on the GPU box from the repo root):
#   tools/profile_round.sh r03a
# 1. rocprofv3 --kernel-trace --stats of the default bench command      -> gpurun_out/<tag>_stats
# 2. separate --pmc passes (never combined with other counters or trace domains): FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU for
#    S3 and S1                                                            -> gpurun_out/pmc_<S>_<COUNTER>
# then: python tools/pmc_summary.py gpurun_out <tag>   (profiles/pmc_traffic.json, profiles/<tag>_pmc_per_kernel.csv)
set -e
tag=${1:-r03}
root=$(pwd)
export TMPDIR=/tmp
rm -rf gpurun_out/${tag}_stats gpurun_out/pmc_S?_FETCH_SIZE gpurun_out/pmc_S?_WRITE_SIZE gpurun_out/pmc_S?_SQ_INSTS_VALU
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/${tag}_stats --output-format csv -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu --no-extras \
    > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_stats.log
echo "stats pass done"
# 3. the instruction mix of the pair kernel (transcendental and f64 instructions issue at a quarter / half of the f32 rate):
#    one pass with the SQ_INSTS_VALU_* class counters                       -> gpurun_out/pmc_<S>_MIX
for wl in S3 S1; do
  rm -rf gpurun_out/pmc_${wl}_MIX
  timeout -k 5 150 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT \
      --kernel-trace -d $root/gpurun_out/pmc_${wl}_MIX --output-format csv -- python3 $root/bench.py --workload $wl --steps 3 --warmup 2 \
      --no-cpu --no-scale --no-extras > gpurun_out/pmc_${wl}_MIX.log 2>&1
  echo "pmc $wl MIX done"
done
for wl in S3 S1; do
  for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    rocprofv3 --pmc $c --kernel-trace -d $root/gpurun_out/pmc_${wl}_${c} --output-format csv -- python3 $root/bench.py --workload $wl --steps 3 --warmup 2 \
        --no-cpu --no-scale --no-extras > gpurun_out/pmc_${wl}_${c}.log 2>&1
    echo "pmc $wl $c done"
  done
done
