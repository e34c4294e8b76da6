#!/bin/bash
# Issue / wait / instruction-cache counters of the trace kernels, three passes:
#   tools/pmc_sets.sh <tag> [bench args...]       (environment, e.g. SPPM_TRACE_BPC=1, is passed through)
set -u
TAG=${1:-sets}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --inflight 1 $*"
run() {
  local n=$1; shift
  local OUT=$REPO/gpurun_out/${TAG}_$n; rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- python3 bench.py $ARGS > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "pass $n failed"; tail -5 "$OUT/bench.err"; exit 1; }
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_trace' in k or 'k_gather<' in k:
            acc[k.split('(')[0][-24:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in sorted(v.items())})
PY
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM
run b SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS
run c SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64
