#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/gpu_profile.sh (gpurun_out/<tag>/) into the summaries
committed under profiles/:

    python tools/summarize_profile.py <tag> <name>

writes profiles/<name>_kernel_stats.csv (the --kernel-trace --stats table), profiles/<name>_pmc.json
(per-kernel averages per launch of every counter collected) and profiles/<name>_bench.json.

HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: both counters are in KiB, and on
gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B, so the read side is
doubled; WRITE_SIZE is exact for 16-B-per-lane stores and dword atomics.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^>]*>)?)", name)
    return m.group(1) if m else name


def pmc_table(path):
    """kernel -> counter -> (sum, launches)"""
    per = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            c = per[short(r["Kernel_Name"])][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
    return per


def main():
    tag, name = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    # a tag collected twice leaves the files of both runs behind (the process id is part of the name): the newest counts
    stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(dst, f"{name}_kernel_stats.csv"))
    bj = os.path.join(src, "bench_stats.json")
    if os.path.exists(bj):
        shutil.copy(bj, os.path.join(dst, f"{name}_bench.json"))
    out = defaultdict(dict)
    for sub in ("fetch", "write", "sq", "tcc", "sq2"):
        for p in sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
            for k, cs in pmc_table(p).items():
                for cname, (s, n) in cs.items():
                    out[k][cname] = s / n
                    out[k].setdefault("launches_per_pass", {})[cname] = n
    for k, d in out.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_traffic_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        if "TCC_HIT_sum" in d and d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0) > 0:
            d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
        if "SQ_BUSY_CYCLES" in d and d.get("SQ_ACTIVE_INST_VALU"):
            d["valu_active_per_wave_cycle"] = d["SQ_ACTIVE_INST_VALU"] / max(1.0, d.get("SQ_WAVE_CYCLES", 0.0))
    out = {k: v for k, v in out.items() if k.startswith("k_")}
    # the kernel sources these counters were collected from (bench.py compares: `counters_stale`)
    # The stamp is the one the profiled run printed itself; a run from before bench.py printed it is stamped with the sources
    # as they are now.
    stamp = None
    try:
        with open(bj) as f:
            stamp = json.loads(f.read().strip().splitlines()[-1]).get("kernel_source_sha256")
    except (OSError, ValueError, IndexError):
        pass
    if not stamp:
        sys.path.insert(0, ROOT)
        from bench import kernel_source_sha256
        stamp = kernel_source_sha256()
    out["_meta"] = {"kernel_source_sha256": stamp}
    with open(os.path.join(dst, f"{name}_pmc.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, d in sorted(out.items()):
        if k == "_meta":
            continue
        print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in d.items() if a != "launches_per_pass"})


if __name__ == "__main__":
    main()
