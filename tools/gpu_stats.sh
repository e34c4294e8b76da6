#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + stats of bench.py into gpurun_out/<tag>/stats.
set -u
TAG=${1:-st}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$REPO"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $* > "$OUT/bench_stats.json" 2> "$OUT/bench_stats.err" || { echo "stats pass failed"; tail -5 "$OUT/bench_stats.err"; exit 1; }
echo done
