#!/bin/bash
# GPU columns of the table of measurements: every configuration at the default command and with one iteration in flight.
#   tools/gpu_table.sh > gpurun_out/gpu_table.txt
for CFG in c1_spheres_plane_256 c2_teapot_800x600 c3_teapot_800x600_10M c4_teapot4_1080p c5_caustic_1024; do
  case $CFG in c3*|c5*) STEPS=20;; *) STEPS=100;; esac
  for MODE in "" "--inflight 1"; do
    echo -n "$CFG [${MODE:-default}] : "
    timeout -k 10 300 python bench.py --steps $STEPS --no-cpu-baseline --config $CFG $MODE 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), 'Mph/s', round(d['ms_per_step'],3), 'ms; steady', round(d.get('value_steady') or 0,1), '; photon pass alone', round(d.get('photon_pass_mphotons_per_s',0),1), {k:round(v,3) for k,v in d['kernel_ms_per_step_one_in_flight'].items()})" || exit 1
  done
done
