#!/bin/bash
# Vector-instruction and wait counters of the trace / gather kernels under several environments (one iteration in flight):
#   tools/ab_pmc_env.sh "ENV=.." "ENV=.. ENV=.." ...
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; export TMPDIR=/tmp
i=0
for E in "$@"; do
  i=$((i+1)); OUT=$REPO/gpurun_out/abpmcenv_$i; rm -rf "$OUT"; mkdir -p "$OUT"
  ( export $E; timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d "$OUT" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > "$OUT/bench.json" 2> "$OUT/bench.err" ) || { echo "pmc pass failed for $E"; tail -5 "$OUT/bench.err"; exit 1; }
  python3 - "$OUT" "$E" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_trace' in k:
            acc[k.split('(')[0][-24:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(tag, k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in sorted(v.items())})
PY
done
