#!/bin/bash
# Runs on the GPU box: SQ counter passes only (instruction mix and waits) of bench.py.
#   tools/gpu_pmc_sq.sh <tag> [bench args...]
set -u
TAG=${1:-sq}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
ARGS="--steps 2 --warmup 1 --steady-steps 0 --no-cpu-baseline $*"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/sq" -- python3 bench.py $ARGS > "$OUT/bench_sq.json" 2> "$OUT/bench_sq.err" || { echo "sq pass failed"; tail -5 "$OUT/bench_sq.err"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/sq2" -- python3 bench.py $ARGS > "$OUT/bench_sq2.json" 2> "$OUT/bench_sq2.err" || { echo "sq2 pass failed"; tail -5 "$OUT/bench_sq2.err"; exit 1; }
echo done
