#!/bin/bash
# Deposit-layout margins of the buckets out of the HitPoints' reach: slots laid out, overflow per pass, rate.
# tools/scan_far_margins.sh > gpurun_out/far_margins.txt
for S in "-1 0 0" "1 1 1" "1 1 0" "0 0 0" "0 1 1" "2 2 1"; do
  set -- $S
  echo "== far sigma $1 add $2 empty $3"
  SPPM_PLACE_FAR_SIGMA=$1 SPPM_PLACE_FAR_ADD=$2 SPPM_PLACE_FAR_EMPTY=$3 SPPM_VERBOSE=1 timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline 2> /tmp/far_err.txt \
    | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), 'GB', round(d['device_bytes_all_contexts']/1e9,2), {k:round(v,3) for k,v in d['kernel_ms_per_step_one_in_flight'].items() if k in ('photon_trace_ms','gather_ms')})" || exit 1
  grep "overflow list" /tmp/far_err.txt | tail -3
done
