#!/bin/bash
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE, KiB per launch) and L2 hits of the gather and trace kernels for several library
# builds at the default command:  tools/ab_traffic.sh lib1.so lib2.so ...
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; export TMPDIR=/tmp
for L in "$@"; do
  TAG=$(basename "$L" .so)
  export SPPM_LIB_PATH=$REPO/$L
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    OUT=$REPO/gpurun_out/abtr_${TAG}_$(echo $C | cut -c1-5); rm -rf "$OUT"; mkdir -p "$OUT"
    timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$OUT" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "pass failed for $L $C"; tail -5 "$OUT/bench.err"; exit 1; }
  done
  python3 - "$REPO/gpurun_out" "$TAG" <<'PY'
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/abtr_' + tag + '_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_gather<' in k or 'k_trace_photon' in k or 'k_trace<1' in k or 'k_gather_split' in k:
            acc[k.split('(')[0][-28:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(tag, k, 'HBM MB/launch', round((2 * m.get('FETCH_SIZE', 0) + m.get('WRITE_SIZE', 0)) * 1024 / 1e6, 1), 'L2 hit', round(m.get('TCC_HIT_sum', 0) / max(1, m.get('TCC_HIT_sum', 0) + m.get('TCC_MISS_sum', 0)), 3))
PY
done
