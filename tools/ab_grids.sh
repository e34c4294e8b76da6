#!/bin/bash
# Runs bench.py under several persistent-grid sizes (SPPM_TRACE_BPC / SPPM_GATHER_BPC) with 2..4 iterations in flight.
cd "${GRAFT_REPO_ROOT:-.}"
run() {
  local label="$1"; shift
  local out
  out=$(env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 24 2>/dev/null | tail -1) || { echo "$label FAILED"; exit 1; }
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(sys.argv[2], round(d['value'],1), round(d['ms_per_step'],3))" "$out" "$label"
}
run "base inflight2" SPPM_INFLIGHT=2
run "T1 G2 inflight2" SPPM_TRACE_BPC=1 SPPM_GATHER_BPC=2 SPPM_INFLIGHT=2
run "T1 G2 inflight3" SPPM_TRACE_BPC=1 SPPM_GATHER_BPC=2 SPPM_INFLIGHT=3
run "T1 G2 inflight4" SPPM_TRACE_BPC=1 SPPM_GATHER_BPC=2 SPPM_INFLIGHT=4
run "T1 G3 inflight3" SPPM_TRACE_BPC=1 SPPM_GATHER_BPC=3 SPPM_INFLIGHT=3
run "T2 G2 inflight3" SPPM_TRACE_BPC=2 SPPM_GATHER_BPC=2 SPPM_INFLIGHT=3
run "T1 G1 inflight4" SPPM_TRACE_BPC=1 SPPM_GATHER_BPC=1 SPPM_INFLIGHT=4
run "T1 G2 E1 inflight4" SPPM_TRACE_BPC=1 SPPM_GATHER_BPC=2 SPPM_EYE_BPC=1 SPPM_INFLIGHT=4
