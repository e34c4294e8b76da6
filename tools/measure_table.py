#!/usr/bin/env python3
"""Runs on the GPU box: measures every configuration on the CPU oracle and on the GPU (the table of measurements in the documentation).
CPU rows: the oracle's literal restatement of the reference (fp64, global MWC stream, kd-tree radius
search, per-round update() sweep), one replica per thread like the reference's OpenMP loop, on a
bounded photon sample; GPU rows: bench.py per configuration.  Writes gpurun_out/table.json."""
import json, os, subprocess, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytrace3_amd import scenes
from tests import oracle_lib as ol

def cpu(config, threads, rounds):
    factory, (w, h), _, per_round = scenes.CONFIGS[config]
    scene = factory()
    lib = ol.load()
    ctxs = [ol.OracleContext(lib, rng_literal=True) for _ in range(threads)]
    for c in ctxs: c.set_scene(scene)
    cam = scenes.default_camera(w, h)
    def run(fn):
        ts = [threading.Thread(target=fn, args=(i,)) for i in range(threads)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        return time.perf_counter() - t0
    te = run(lambda i: ctxs[i].eye_pass(cam, 1, i))
    tp = run(lambda i: ctxs[i].photon_pass(rounds, per_round, 1, i, mode=ol.GATHER_KDTREE))
    ph = threads * rounds * per_round
    for c in ctxs: c.close()
    return {"threads": threads, "photons": ph, "eye_s": te, "photon_s": tp, "mphotons_per_s": ph / tp / 1e6,
            "ms_per_iter_extrapolated": None}

GPU_ONLY = "--gpu-only" in sys.argv     # the CPU rows do not change with the kernels
out = {"cpu": {}, "gpu": {}, "gpu_single_stream": {}}
for config, rounds in (() if GPU_ONLY else (("c1_spheres_plane_256", 1000), ("c2_teapot_800x600", 500), ("c4_teapot4_1080p", 100), ("c5_caustic_1024", 2000))):
    full_rounds = scenes.CONFIGS[config][2]
    for threads in (4, 16):
        r = cpu(config, threads, rounds)
        # one replica-iteration at the configuration's full photon count, from the sampled rates
        r["ms_per_iter_extrapolated"] = (r["eye_s"] + r["photon_s"] * full_rounds / rounds) * 1e3 / 1.0
        r["sample_rounds"], r["full_rounds"] = rounds, full_rounds
        out["cpu"][f"{config}/{threads}"] = r
        print(config, threads, r, flush=True)
for config in ("c1_spheres_plane_256", "c2_teapot_800x600", "c3_teapot_800x600_10M", "c4_teapot4_1080p", "c5_caustic_1024"):
    for key, extra in (("gpu", []), ("gpu_single_stream", ["--inflight", "1"])):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "20" if config.startswith(("c3", "c5")) else "100", "--warmup", "10", "--no-cpu-baseline"] + extra,
                           capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        out[key][config] = json.loads(line[-1]) if line else {"error": p.stderr[-500:]}
        print(config, key, out[key][config].get("value"), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "table.json"), "w"), indent=1)
