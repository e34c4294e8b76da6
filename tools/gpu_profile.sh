#!/bin/bash
# Runs on the GPU box (through gpurun): kernel stats + PMC passes of bench.py.
#   tools/gpu_profile.sh <tag> [bench args...]
# Outputs land in gpurun_out/<tag>/{stats,fetch,write,sq}; tools/summarize_profile.py turns
# them into the files committed under profiles/.
set -u
TAG=${1:-prof}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
ARGS="--steps 3 --warmup 1 --steady-steps 0 --no-cpu-baseline $*"
echo "== stats"; 
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/bench_stats.json" 2> "$OUT/bench_stats.err" || { echo "stats pass failed"; tail -5 "$OUT/bench_stats.err"; exit 1; }
echo "== pmc FETCH_SIZE"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/bench_fetch.err" || { echo "fetch pass failed"; tail -5 "$OUT/bench_fetch.err"; exit 1; }
echo "== pmc WRITE_SIZE"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py $ARGS > "$OUT/bench_write.json" 2> "$OUT/bench_write.err" || { echo "write pass failed"; tail -5 "$OUT/bench_write.err"; exit 1; }
echo "== pmc SQ"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sq" -- python3 bench.py $ARGS > "$OUT/bench_sq.json" 2> "$OUT/bench_sq.err" || { echo "sq pass failed"; tail -5 "$OUT/bench_sq.err"; exit 1; }
echo "== pmc TCC hit/miss"
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/tcc" -- python3 bench.py $ARGS > "$OUT/bench_tcc.json" 2> "$OUT/bench_tcc.err" || { echo "tcc pass failed"; tail -5 "$OUT/bench_tcc.err"; exit 1; }
find "$OUT" -name '*.csv' | head -40
echo done
