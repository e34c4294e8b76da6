#!/bin/bash
# Steady-state rate of the drop-in outer loop (examples/_build/sppm_main): two runs of different length, the difference
# of their wall times over the difference of their iteration counts (process start and first-use costs cancel).
# usage: tools/time_loop_steady.sh [VAR=value ...]
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/loop"
mkdir -p "$OUT" && cd "$OUT"
export SPPM_DATA_DIR="$ROOT/raytrace3_amd/data" SPPM_W=800 SPPM_H=600 SPPM_SAVE_EVERY=1000000
for kv in "$@"; do export "$kv"; done
run() { local t0=$(date +%s.%N); SPPM_ITERS=$1 timeout -k 10 200 "$ROOT/examples/_build/sppm_main" > /dev/null 2> err.txt; local t1=$(date +%s.%N); echo "$t0 $t1"; }
run 2 > /dev/null
A=$(run 300); B=$(run 1500)
python3 - $A $B "$*" <<'PY'
import sys
a0, a1, b0, b1 = map(float, sys.argv[1:5])
per = ((b1 - b0) - (a1 - a0)) / (1500 - 300)
print(f"loop {sys.argv[5] or '(defaults)'}: {per * 1e3:.2f} ms per iteration of 4 replicas in the steady state, {4e6 / per / 1e6:.1f} Mphotons/s "
      f"(start-up {(a1 - a0) - 300 * per:.2f} s)")
PY
