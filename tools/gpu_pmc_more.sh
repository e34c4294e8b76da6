#!/bin/bash
# tools/gpu_pmc_more.sh [bench args] : instruction-mix and wait counters of the big kernels, three counter passes of a short run
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$REPO"; export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_more; rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--steps 2 --warmup 1 --steady-steps 0 --no-cpu-baseline $*"
pass() { timeout -s KILL 200 rocprofv3 --pmc $2 --output-format csv -d "$OUT/$1" -- python3 bench.py $ARGS > "$OUT/$1.json" 2> "$OUT/$1.err" || { echo "$1 failed"; tail -3 "$OUT/$1.err"; exit 1; }; }
pass a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
pass b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT"
pass c "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_trace' in k or 'k_gather<' in k:
            acc[k[-22:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in sorted(v.items())})
PY
