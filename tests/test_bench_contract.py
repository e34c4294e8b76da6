"""bench.py prints the one JSON line the round-end driver parses: the keys it names, the workload BASELINE.json
quotes the metric on, roofline objects that bound (no fraction above 1) and, at one GPU, the CPU baseline beside it.
Started with --gpus 2 and no launcher around it, it starts its two ranks itself."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    e.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, check=True, capture_output=True, text=True,
                         timeout=420, env=e).stdout
    lines = [ln for ln in out.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out                       # exactly one JSON line on stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_keeps_the_contract():
    d = _bench("--steps", "3", "--warmup", "1", "--cpu-rounds", "30")
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
    # the dominant kernel, bound by vector-instruction issue; a fraction of a bound never exceeds 1
    r = d["roofline"]
    assert r["bound"] == "valu" and "k_trace" in r["kernel"] and r["peak"] == pytest.approx(256 * 4 * 64 * 2.4e9 / 1e12)
    assert r["avg_launch_ms"] > 0
    if r["achieved"] is not None:                     # counters come from the committed profile of this command
        assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0 < r["frac"] <= 1
        assert 0 < r["valu_busy_frac"] <= 1 and 0 < r["lanes_active"] <= 64
        assert r["traffic"] is None or r["traffic"] > 0
        a = r["timed_region_all_kernels"]                 # the whole GPU while several iterations share it
        assert 0 < a["valu_busy_frac"] <= 1 and 0 < a["lane_frac"] <= a["valu_busy_frac"]
    g = d["roofline_gather"]
    assert g["bound"] == "hbm" and g["unit"] == "GB/s" and g["peak"] == 8000.0 and g["avg_launch_ms"] > 0
    assert g["compulsory_bytes"] > 0 and g["reference_equivalent_bytes"] > g["compulsory_bytes"]
    assert 0 < g["pairs_in_range_frac"] <= 1
    if g["achieved"] is not None:
        assert g["frac"] == pytest.approx(g["achieved"] / g["peak"]) and 0 < g["frac"] <= 1
        assert g["achieved"] == pytest.approx(g["traffic"] / (g["avg_launch_ms"] / 1e3) / 1e9, rel=1e-6)
        assert g["traffic_over_compulsory"] == pytest.approx(g["traffic"] / g["compulsory_bytes"])
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mphotons/s" and c["cores"] == 4 and c["value"] > 0 and "photons" in c["sample"]
    assert c["host_nproc"] >= 1
    if "all_cores" in c:
        assert c["all_cores"]["cores"] > 4 and c["all_cores"]["value"] > 0
    t = c["all_cores_grid_gather_no_update_sweep"]     # the restatement without the reference's dead sweep: the fairer CPU figure
    assert t["value"] > 0 and t["cores"] >= 1 and "no per-round update() sweep" in t["sample"]
    # the timed region's length, a second longer region beside it, the hardware queues the run had, and whether the committed
    # counters were collected from the kernel sources that ran
    assert d["timed_region_ms"] == pytest.approx(d["ms_per_step"] * d["steps"])
    assert d["value_steady"] > 10 and "200" in d["value_steady_is"]
    assert d["hw_queues"] == "8"
    assert len(d["kernel_source_sha256"]) == 64 and r["counters_stale"] in (True, False, None)
    if r["achieved"] is None:
        assert r["counters_stale"] is None
    k = d["kernel_ms_per_step_one_in_flight"]
    assert all(k[x] >= 0 for x in ("eye_trace_ms", "grid_build_ms", "photon_trace_ms", "deposit_sort_ms", "gather_ms", "resolve_ms"))
    assert d["counters_per_step"]["photons"] == 1_000_000
    assert d["device_bytes_all_contexts"] > 0


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` with no launcher around it: two ranks are started (here both on the one GPU of the box, which
    RCCL refuses, so the images are reduced over gloo through host copies) and the line reports the group's size."""
    d = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--inflight", "2", "--no-cpu-baseline", env={"SPPM_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["counters_per_step"]["photons"] == 1_000_000          # per rank: the pass of a step is 2 M photons, sharded
    assert "gloo" in d["config"]["collective"]
    assert d["value"] == pytest.approx(2e6 / (d["ms_per_step"] / 1e3) / 1e6, rel=1e-6)


@pytest.mark.gpu
def test_bench_strong_scaling_shards_the_configs_own_pass():
    """--scaling strong: the pass is the config's (here 3 000 rounds x 100 photons of config 2's scene), sharded over the ranks,
    where weak scaling gives every rank that many."""
    d = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--inflight", "2", "--no-cpu-baseline", "--scaling", "strong", "--rounds", "3000",
               "--steady-steps", "0", env={"SPPM_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value_steady"] is None
    assert d["counters_per_step"]["photons"] == 150_000            # per rank: half of the 300 000-photon pass
    assert "300000 photons per iteration" in d["config"]["workload"]
    assert d["value"] == pytest.approx(3e5 / (d["ms_per_step"] / 1e3) / 1e6, rel=1e-6)


def test_bench_refuses_a_launcher_of_another_size():
    """Under a launcher (WORLD_SIZE set) the world size must equal --gpus: checked before anything touches a GPU."""
    e = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=ROOT, capture_output=True, text=True,
                       timeout=300, env=e)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr
