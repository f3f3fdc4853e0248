#!/bin/bash
# rocprofv3 --kernel-trace --stats summary of the default bench command (10 timed steps), run from /tmp.
#   bash profiles/tools/kernel_stats.sh TAG [extra bench args]   (on the GPU box)
# writes gpurun_out/TAG_kernel_stats.csv (profiles/summarize_trace.py per (kernel, grid) table) and
# gpurun_out/TAG_rocprof_stats_head.csv (head of rocprofv3's own kernel_stats file); copy both into profiles/.
set -e
TAG=${1:?tag}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STEPS=10
WARM=3
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT" -o trace --output-format csv -- \
  python3 "$ROOT/bench.py" --steps $STEPS --warmup $WARM --no-cpu-baseline --no-kernel-events --no-alt "$@" > "$OUT/run.log" 2>&1
cd "$ROOT"
TRACE=$(find "$OUT" -name 'trace_kernel_trace.csv' | head -1)
STATS=$(find "$OUT" -name 'trace_kernel_stats.csv' | head -1)
# keep only the timed steps: drop the launches of the warm-up by time (bench logs nothing into the trace, so take the
# last STEPS/(STEPS+WARM) share of launches of the dominant kernel); simpler and robust: summarise everything and
# divide by STEPS + WARM
python3 profiles/summarize_trace.py "$TRACE" $((STEPS + WARM)) > "gpurun_out/${TAG}_kernel_stats.csv"
head -40 "$STATS" > "gpurun_out/${TAG}_rocprof_stats_head.csv"
python3 profiles/tools/step_timeline.py "$TRACE" > "gpurun_out/${TAG}_step_timeline.txt" 2>&1 || true
python3 profiles/tools/step_launches.py "$TRACE" > "gpurun_out/${TAG}_step_launches.txt" 2>&1 || true
tail -3 "$OUT/run.log"
rm -f "$TRACE"  # tens of MB; the summaries are what is kept
