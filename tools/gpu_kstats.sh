#!/bin/bash
# tools/gpu_kstats.sh [bench args] : rocprofv3 kernel stats of a short bench run (one iteration in flight), top kernels printed
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$REPO"; export TMPDIR=/tmp
OUT=$REPO/gpurun_out/kstats; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --steady-steps 0 --inflight 1 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
F=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r["Name"][:70], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "%", r["Percentage"])
PY
