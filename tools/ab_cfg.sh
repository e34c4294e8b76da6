#!/bin/bash
# tools/ab_cfg.sh <config> "ENV=.." ... : bench one config under different environments
CFG=$1; shift
for E in "$@"; do
  echo "== $CFG $E"
  env $E timeout -k 10 200 python bench.py --steps 2 --no-cpu-baseline --config $CFG 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), round(d['ms_per_step'],2), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" || exit 1
done
