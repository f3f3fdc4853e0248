#!/bin/bash
# HBM traffic per kernel class and per launch of the roofline kernel (bench.py roofline.traffic): two rocprofv3 PMC
# passes (FETCH_SIZE and WRITE_SIZE cannot share a pass, MI355X_MICROARCH.md), --kernel-trace only, run from /tmp.
#   bash profiles/tools/collect_traffic.sh [TAG=r02] [git head]   (on the GPU box; writes gpurun_out/TAG_traffic.json,
#   gpurun_out/TAG_hbm_kernels.json -- copy them into profiles/)
#   BENCH_ARGS="--dtype bf16" (or "--size 128") selects another configuration than the f32 256x256 bs 16 headline; it is
#   recorded in the json files ("bench_args")
set -e
TAG=${1:-r03}
HEAD=${2:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d "$OUT/$c" -o pmc --output-format csv -- \
    python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-alt $BENCH_ARGS > "$OUT/$c.log" 2>&1
  echo "pass $c done"
done
cd "$ROOT" && python3 profiles/tools/traffic_from_pmc.py "$OUT" "gpurun_out/${TAG}_traffic.json" \
  "gpurun_out/${TAG}_hbm_kernels.json" "$HEAD" "$(date -u +%Y-%m-%d)" "$BENCH_ARGS"
find "$OUT" -name '*.csv' -size +2M -delete
