for F in 1 -1 1 -1; do
  echo "== far sigma $F"
  SPPM_PLACE_FAR_SIGMA=$F SPPM_VERBOSE=1 timeout -k 10 200 python bench.py --config c5_caustic_1024 --steps 20 --inflight 1 --no-cpu-baseline 2> /tmp/c5_err.txt \
    | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), 'GB', round(d['device_bytes_all_contexts']/1e9,2), {k:round(v,3) for k,v in d['kernel_ms_per_step_one_in_flight'].items()})" || exit 1
  grep "overflow list" /tmp/c5_err.txt | tail -2
done
