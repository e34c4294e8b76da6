"""bench.py prints the one JSON line the round-end driver parses: the keys it names, the workload BASELINE.json
quotes the metric on, a roofline object and (at one GPU) the CPU baseline beside it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_keeps_the_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-rounds", "30"],
                         cwd=ROOT, check=True, capture_output=True, text=True, timeout=300).stdout
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out                       # exactly one line on stdout
    d = json.loads(lines[0])
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    # BASELINE.json: "Mphotons/s + ms/SPPM-iter, 800x600 teapot scene, ...": value carries the first, ms_per_step the second
    assert base["metric"].startswith(d["metric"]) and d["metric"] == d["unit"] == "Mphotons/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "800x600" in d["config"]["workload"] and "1000000 photons" in d["config"]["workload"] and "model" not in d["config"]
    assert d["value"] > 10 and d["ms_per_step"] > 0
    # value is photons of the timed steps over their wall time
    assert d["value"] == pytest.approx(1e6 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3) / 1e6, rel=1e-6)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and r["achieved"] > 0
    assert r["traffic"] is None or r["traffic"] > 0
    # algorithmic bytes per launch over the launch's HIP-event duration
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] / 1e3) / 1e9, rel=1e-6)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mphotons/s" and c["cores"] == 4 and c["value"] > 0 and "photons" in c["sample"]
    k = d["kernel_ms_per_step_one_in_flight"]
    assert all(k[x] >= 0 for x in ("eye_trace_ms", "grid_build_ms", "photon_trace_ms", "deposit_sort_ms", "gather_ms", "resolve_ms"))
    assert d["counters_per_step"]["photons"] == 1_000_000
