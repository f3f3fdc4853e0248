#!/bin/bash
# MFMA-pipe utilisation per kernel class over a training step: one rocprofv3 PMC pass (SQ counters), run from /tmp.
#   bash profiles/tools/collect_mfma.sh [TAG=r02] [git head]   (on the GPU box; prints the table, writes
#   gpurun_out/TAG_mfma_util.json -- copy it into profiles/)
#   BENCH_ARGS="--dtype bf16" selects another configuration (recorded in the json); MOPS=BF16 counts the bf16 MFMA ops
set -e
TAG=${1:-r03}
HEAD=${2:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_mfma_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_${MOPS:-F32} SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace -d "$OUT" -o pmc --output-format csv -- \
  python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-alt $BENCH_ARGS > "$OUT/run.log" 2>&1
cd "$ROOT" && python3 profiles/tools/mfma_from_pmc.py "$OUT" "gpurun_out/${TAG}_mfma_util.json" "$HEAD" "$(date -u +%Y-%m-%d)" "$BENCH_ARGS"
find "$OUT" -name '*.csv' -size +2M -delete
