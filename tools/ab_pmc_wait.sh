#!/bin/bash
# tools/ab_pmc_wait.sh lib1.so ... : wave-cycle split (active / waiting) of the trace kernels on the sharing grids, per library build
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$REPO"; export TMPDIR=/tmp
for L in "$@"; do
  TAG=$(basename "$L" .so); OUT=$REPO/gpurun_out/abwait_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
  SPPM_LIB_PATH=$REPO/$L timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d "$OUT" -- python3 bench.py --steps 2 --warmup 1 --steady-steps 0 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "pmc pass failed for $L"; tail -5 "$OUT/bench.err"; exit 1; }
  python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections, json
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_trace' in k:
            acc[k[-20:]][r['Counter_Name']].append(float(r['Counter_Value']))
b = json.loads(open(out + '/bench.json').read().strip().splitlines()[-1])
print(tag, {k[:-3]: round(v, 3) for k, v in b['kernel_ms_per_step_one_in_flight'].items() if 'trace' in k}, 'segments', round(b['counters_per_step']['segments']))
for k, v in sorted(acc.items()):
    print('   ', k, {c: round(sum(x) / len(x) / 1e6, 1) for c, x in sorted(v.items())})
PY
done
