#!/bin/bash
# A/B of library builds: tools/ab_lib.sh lib1.so lib2.so ...   (whole-iteration rate at the default command, and the
# kernels' own durations with one iteration in flight).  Two rounds over the list, 100 timed steps each: the run-to-run
# spread of the default command is about 1 %.
for ROUND in 1 2; do
for L in "$@"; do
  echo "== $L (round $ROUND)"
  SPPM_LIB_PATH=$PWD/$L timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), {k:round(v,3) for k,v in d['kernel_ms_per_step_one_in_flight'].items()})" || exit 1
done
done
