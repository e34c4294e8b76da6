#!/bin/bash
# tools/ab_try.sh "LABEL|ENV=.. ENV=.." ... : the default bench command (100 timed steps and the 200-step steady region) under
# each environment, twice over the list; prints value, value_steady and the kernels' own durations (one iteration in flight).
cd "${GRAFT_REPO_ROOT:-.}"
for ROUND in 1 2; do
for SPEC in "$@"; do
  LABEL="${SPEC%%|*}"; E="${SPEC#*|}"
  OUT=$(env $E timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | tail -1) || { echo "$LABEL FAILED"; exit 1; }
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(sys.argv[2], 'value', round(d['value'],1), 'steady', round(d['value_steady'],1), {k[:-3]:round(v,3) for k,v in d['kernel_ms_per_step_one_in_flight'].items()})" "$OUT" "$LABEL"
done
done
