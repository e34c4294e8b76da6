#!/bin/bash
# tools/ab_env.sh "ENV1=.. ENV2=.." "ENV..." : bench under different environments
for E in "$@"; do
  echo "== $E"
  env $E timeout -k 10 200 python bench.py --steps 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" || exit 1
done
