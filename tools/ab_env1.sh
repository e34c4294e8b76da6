#!/bin/bash
# tools/ab_env1.sh "ENV1=.. ENV2=.." "ENV..." : the default command under different environments; whole-iteration rate and
# the kernels' own durations (one iteration in flight on the grids the contexts share)
for E in "$@"; do
  echo "== $E"
  env $E timeout -k 10 200 python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), {k:round(v,3) for k,v in d['kernel_ms_per_step_one_in_flight'].items()})" || exit 1
done
