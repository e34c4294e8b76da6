#!/bin/bash
# Vector-instruction and wait counters of the trace / gather kernels for several library builds:
#   tools/ab_pmc.sh lib1.so lib2.so ...      (one iteration in flight: full grids, kernels undisturbed)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; export TMPDIR=/tmp
for L in "$@"; do
  TAG=$(basename "$L" .so); OUT=$REPO/gpurun_out/abpmc_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
  export SPPM_LIB_PATH=$REPO/$L
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d "$OUT" -- python3 bench.py --steps 2 --warmup 1 --steady-steps 0 --no-cpu-baseline --inflight 1 > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "pmc pass failed for $L"; tail -5 "$OUT/bench.err"; exit 1; }
  python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_trace' in k or 'k_gather<' in k:
            acc[k.split('(')[0][-24:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(tag, k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in sorted(v.items())})
PY
done
