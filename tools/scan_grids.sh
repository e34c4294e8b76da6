#!/bin/bash
# Persistent-grid sizes (workgroups per CU) of contexts that share a GPU: whole-iteration rate at C2.
#   tools/scan_grids.sh > gpurun_out/scan_grids.txt
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1))" || exit 1; }
run SPPM_NOOP=1
run SPPM_TRACE_BPC=2
run SPPM_TRACE_BPC=2 SPPM_GATHER_BPC=2
run SPPM_EYE_BPC=2
run SPPM_GATHER_BPC=4
run SPPM_GATHER_BPC=2
run SPPM_INFLIGHT=6
run SPPM_INFLIGHT=4
run SPPM_INFLIGHT=8 GPU_MAX_HW_QUEUES=8
run SPPM_NOOP=1
