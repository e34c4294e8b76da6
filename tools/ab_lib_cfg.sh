#!/bin/bash
# A/B of library builds over the other configurations: tools/ab_lib_cfg.sh lib1.so lib2.so ...
for CFG in c1_spheres_plane_256 c3_teapot_800x600_10M c4_teapot4_1080p c5_caustic_1024; do
for L in "$@"; do
  echo -n "$CFG $L : "
  SPPM_LIB_PATH=$PWD/$L timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --config $CFG 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step_one_in_flight'].items()})" || exit 1
done
done
