#!/bin/bash
# HBM traffic of the forward conv_igemm launches (bench.py roofline.traffic): two rocprofv3 PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass, MI355X_MICROARCH.md), --kernel-trace only, run from /tmp.
#   bash profiles/tools/collect_traffic.sh        (on the GPU box; writes profiles/r01_traffic.json)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace -d "$OUT/$c" -o pmc --output-format csv -- \
    python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-alt > "$OUT/$c.log" 2>&1
  echo "pass $c done"
done
cd "$ROOT" && python3 profiles/tools/traffic_from_pmc.py "$OUT" profiles/r01_traffic.json
