#!/bin/bash
# tools/ab_cfg_env.sh CONFIG "LABEL|ENV=.. ENV=.." ... : one benchmark configuration (20 timed steps) under several environments
cd "${GRAFT_REPO_ROOT:-.}"
CFG=$1; shift
for SPEC in "$@"; do
  LABEL="${SPEC%%|*}"; E="${SPEC#*|}"
  OUT=$(env $E timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 3 --steady-steps 0 --no-cpu-baseline 2>/dev/null | tail -1) || { echo "$CFG $LABEL FAILED"; exit 1; }
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(sys.argv[2], sys.argv[3], 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), {k[:-3]:round(v,3) for k,v in d['kernel_ms_per_step_one_in_flight'].items()})" "$OUT" "$CFG" "$LABEL"
done
