#!/bin/bash
# End-of-round evidence, as two gpurun calls (each well inside 20 minutes):
#   gpurun --timeout 1200 -- 'bash profiles/tools/end_of_round.sh counters r06 <git head>'
#   (copy gpurun_out/<tag>_*{traffic,hbm_kernels,mfma_util}.json into profiles/ and commit: bench.py attaches them by digest)
#   gpurun --timeout 1200 -- 'bash profiles/tools/end_of_round.sh benches r05'
# counters: rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ counters, separate runs) for f32 and bf16 256x256,
#           128x128 and 448x448 bs 14;   benches: kernel traces, the default bench line, the other configurations, the
#           secondary workloads and the data-parallel self test.  Everything lands under gpurun_out/.
set -e
MODE=${1:?counters|benches}
TAG=${2:-r05}
HEAD=${3:-unknown}
O=gpurun_out/${TAG}_end; mkdir -p $O
if [ "$MODE" = counters ]; then
  bash profiles/tools/collect_traffic.sh $TAG $HEAD > $O/traffic.log 2>&1; tail -1 $O/traffic.log
  bash profiles/tools/collect_mfma.sh $TAG $HEAD > $O/mfma.log 2>&1; tail -1 $O/mfma.log
  BENCH_ARGS="--dtype bf16" bash profiles/tools/collect_traffic.sh ${TAG}_bf16 $HEAD > $O/traffic_bf16.log 2>&1; tail -1 $O/traffic_bf16.log
  BENCH_ARGS="--dtype bf16" MOPS=BF16 bash profiles/tools/collect_mfma.sh ${TAG}_bf16 $HEAD > $O/mfma_bf16.log 2>&1; tail -1 $O/mfma_bf16.log
  BENCH_ARGS="--size 128" bash profiles/tools/collect_traffic.sh ${TAG}_128 $HEAD > $O/traffic128.log 2>&1; tail -1 $O/traffic128.log
  BENCH_ARGS="--size 128" bash profiles/tools/collect_mfma.sh ${TAG}_128 $HEAD > $O/mfma128.log 2>&1; tail -1 $O/mfma128.log
  BENCH_ARGS="--size 448 --batch 14" bash profiles/tools/collect_traffic.sh ${TAG}_448 $HEAD > $O/traffic448.log 2>&1; tail -1 $O/traffic448.log
  BENCH_ARGS="--size 448 --batch 14" bash profiles/tools/collect_mfma.sh ${TAG}_448 $HEAD > $O/mfma448.log 2>&1; tail -1 $O/mfma448.log
  rm -rf gpurun_out/pmc_${TAG}* gpurun_out/pmc_mfma_${TAG}*
else
  bash profiles/tools/kernel_stats.sh ${TAG}_end > $O/ks.log 2>&1; tail -1 $O/ks.log
  bash profiles/tools/kernel_stats.sh ${TAG}_end_bf16 --dtype bf16 > $O/ks_bf16.log 2>&1; tail -1 $O/ks_bf16.log
  bash profiles/tools/kernel_stats.sh ${TAG}_end_128 --size 128 > $O/ks_128.log 2>&1; tail -1 $O/ks_128.log
  python bench.py > $O/bench.json 2> $O/bench.err
  python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2>/dev/null
  python bench.py --size 128 --no-cpu-baseline --no-alt > $O/bench_128.json 2>/dev/null
  python bench.py --size 128 --dtype bf16 --no-cpu-baseline > $O/bench_128_bf16.json 2>/dev/null
  python bench.py --size 448 --batch 14 --no-cpu-baseline --no-alt > $O/bench_448.json 2>/dev/null
  python bench.py --workload deepfake > $O/workload_deepfake.json 2>/dev/null
  python bench.py --workload deepfake --pair-fused off > $O/workload_deepfake_sequential.json 2>/dev/null
  python bench.py --workload deepfake --dtype bf16 > $O/workload_deepfake_bf16.json 2>/dev/null
  python bench.py --workload deepfake --mode swap > $O/workload_deepfake_swap.json 2>/dev/null
  python bench.py --workload deepfake --mode swap --dtype bf16 > $O/workload_deepfake_swap_bf16.json 2>/dev/null
  python bench.py --workload sample50 --steps 3 --warmup 1 > $O/workload_sample50.json 2>/dev/null
  python bench.py --workload sample50 --dtype bf16 --steps 3 --warmup 1 > $O/workload_sample50_bf16.json 2>/dev/null
  python bench.py --workload predict > $O/workload_predict.json 2>/dev/null
  python bench.py --dp-selftest --steps 30 > $O/dp_selftest.json 2>/dev/null
  grep -o '"ms_per_step": [0-9.]*, "higher' $O/bench*.json
fi
