#!/bin/bash
# Times the drop-in outer SPPM loop (examples/_build/sppm_main) at 800x600, 1 M photons per replica, on the GPU box.
# usage: tools/time_loop.sh <iterations> [VAR=value ...]
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
ITERS="${1:-40}"; shift || true
OUT="$ROOT/gpurun_out/loop"
mkdir -p "$OUT" && cd "$OUT"
export SPPM_DATA_DIR="$ROOT/raytrace3_amd/data" SPPM_W=800 SPPM_H=600
for kv in "$@"; do export "$kv"; done
SPPM_ITERS=2 timeout -k 10 100 "$ROOT/examples/_build/sppm_main" > /dev/null 2>&1     # page the binary and the GPU runtime in
t0=$(date +%s.%N)
SPPM_ITERS=$ITERS timeout -k 10 200 "$ROOT/examples/_build/sppm_main" > out.txt 2> err.txt
t1=$(date +%s.%N)
python3 - "$t0" "$t1" "$ITERS" "$*" <<'PY'
import sys
t0, t1, n = float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
dt = t1 - t0
print(f"loop {sys.argv[4] or '(defaults)'}: {n} iterations in {dt:.3f} s (process start included): "
      f"{dt / n * 1e3:.2f} ms per iteration of 4 replicas, {4 * n * 1e6 / dt / 1e6:.1f} Mphotons/s")
PY
