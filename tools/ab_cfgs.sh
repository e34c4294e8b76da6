#!/bin/bash
# tools/ab_cfgs.sh lib1.so lib2.so ... : the other benchmark configurations (C3, C4, C5; 20 timed steps) under several library builds.
cd "${GRAFT_REPO_ROOT:-.}"
for CFG in c3_teapot_800x600_10M c4_teapot4_1080p c5_caustic_1024; do
for L in "$@"; do
  OUT=$(SPPM_LIB_PATH=$PWD/$L timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 3 --steady-steps 0 --no-cpu-baseline 2>/dev/null | tail -1) || { echo "$CFG $L FAILED"; exit 1; }
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(sys.argv[2], sys.argv[3], 'value', round(d['value'],1), {k[:-3]:round(v,3) for k,v in d['kernel_ms_per_step_one_in_flight'].items()})" "$OUT" "$CFG" "$(basename $L)"
done
done
