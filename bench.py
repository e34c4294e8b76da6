#!/usr/bin/env python3
"""bench.py — SPPM hot path on MI355X: Mphotons/s and ms per SPPM iteration.

Five replica iterations are in flight per GPU by default (`--inflight 5`: five library contexts =
five HIP streams, used in turn), the way the reference keeps four replicas in flight with its
OpenMP loop (Raytracer.h:442): contexts that share a GPU launch persistent grids small enough for the
trace of one iteration, the gather of another and the eye pass / grid build / resolve of others to be
resident on the CUs together.  `--inflight 1` runs the iterations strictly one after the other, on full grids (per-kernel
HIP-event durations are then undisturbed by the neighbour streams).

One step = one SPPM iteration of one replica on the teapot Cornell scene at 800x600
(eye pass + HitPoint grid + photon pass + resolve, plus the all-reduce of the flux image
when N > 1).  Every rank traces 1 M photons per step (rounds = 10000 x 100 per GPU,
Raytracer.h:218,384); with N ranks the pass is N x 1 M photons of the SAME replica,
sharded by photon index, so per-GPU work is fixed ("weak").  `value` is photons of all
ranks / wall time of the timed steps; the photon-pass-only rate (the metric's own
definition) is reported beside it.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# One hardware queue per context: with HIP's default of 4 queues per process the streams of five contexts (and torch's
# own) are folded onto four queues, and kernels of contexts that share a queue run one after the other.  Read by the HIP
# runtime when it starts, hence set before anything touches the GPU.
# (The GPU boxes of this pool export the default, 4, explicitly, so the value is set here, not merely defaulted;
# SPPM_HW_QUEUES=0 leaves the environment as it is.)
if os.environ.get("SPPM_HW_QUEUES", "8") != "0":
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("SPPM_HW_QUEUES", "8")
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak (spec)
SEED = 20260101


def cpu_baseline(scene, w, h, rounds, threads, literal=True):
    """The reference's CPU path as restated by the oracle (fp64, global MWC stream, kd-tree
    radius search + the per-round update() sweep), `threads` replicas in parallel like the
    reference's `omp parallel for num_threads(4)`.  Bounded sample: full-resolution eye pass,
    `rounds` x 100 photons per replica.
    literal=False: the same restatement with a hashed-grid gather in place of the kd-tree and without the
    per-round update() sweep over all HitPoints (a guarded no-op in the reference that is most of its CPU
    time): what the host cores do when they do not pay for that sweep."""
    from raytrace3_amd import scenes
    from tests import oracle_lib as ol
    lib = ol.load()
    ctxs = [ol.OracleContext(lib, rng_literal=literal) for _ in range(threads)]
    rng = np.random.default_rng(1)
    cams = []
    for _ in range(threads):
        j = rng.normal(size=3)
        cams.append(scenes.default_camera(w, h, jitter=0.00015 * j / np.linalg.norm(j)))
    for c in ctxs:
        c.set_scene(scene)

    def run(fn):
        ts = [threading.Thread(target=fn, args=(i,)) for i in range(threads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return time.perf_counter() - t0

    t_eye = run(lambda i: ctxs[i].eye_pass(cams[i], SEED, i))
    t_ph = run(lambda i: ctxs[i].photon_pass(rounds, 100, SEED, i, mode=ol.GATHER_KDTREE if literal else ol.GATHER_GRID))
    photons = threads * rounds * 100
    for c in ctxs:
        c.close()
    return {
        "value": photons / t_ph / 1e6, "unit": "Mphotons/s", "cores": threads, "kind": "port", "host_nproc": os.cpu_count(),
        "sample": f"{threads} replicas x {rounds * 100} photons ({rounds} rounds x 100) on the {w}x{h} teapot scene, "
                  f"photon pass {t_ph:.1f} s (eye pass {t_eye:.1f} s not counted), "
                  + ("oracle literal mode: fp64, kd-tree radius search, per-round update() sweep" if literal else
                     "oracle with a hashed-grid gather and no per-round update() sweep: fp64, counter-based random streams"),
        "eye_pass_s": t_eye, "photon_pass_s": t_ph,
    }


N_CU, N_SIMD, CLOCK_HZ = 256, 4, 2.4e9
SIMD_CYCLES_PER_S = N_CU * N_SIMD * CLOCK_HZ                 # what the vector units of the chip offer, in SIMD-cycles
PEAK_TLANE_CYCLES = SIMD_CYCLES_PER_S * 64 / 1e12            # ... and in lane-cycles: 64 lanes per SIMD-cycle of a busy vector unit


# the photon and eye trace kernels of a scene with Bezier objects, as profiles name them (template instances up to the
# r2_07 sets, kernels of their own since)
TRACE_PHOTON = ("k_trace_photon_bez", "k_trace<1")
TRACE_EYE = ("k_trace_eye_bez", "k_trace<0")


def _kernel(d, name):
    """The summary's entry for a kernel, template arguments or not (k_gather, k_gather<true>); name may be a tuple of
    alternatives, the first that is present wins."""
    for n in (name if isinstance(name, tuple) else (name,)):
        for k, v in d.items():
            if k == n or k.startswith(n + "<") or (n[-2:] in ("<0", "<1") and k.startswith(n)):
                return v
    return {}


KERNEL_SOURCES = ("raytrace3_amd/csrc/sppm_hip.hip", "raytrace3_amd/csrc/sppm_device.h")


def kernel_source_sha256():
    """sha256 over the kernel sources: tools/summarize_profile.py stamps a profile with it, and a bench line that takes its
    counters from a profile made from other sources says so (`counters_stale`)."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def committed_profile(shared):
    """The newest committed rocprofv3 summary (profiles/*_pmc.json + *_kernel_stats.csv) of the launch mode asked for:
    several iterations in flight (the default command; its persistent grids are sized to share the GPU) or one.
    Returns (per-kernel counters per launch, {kernel: share of GPU time}, file name) or None."""
    import csv
    import glob
    want = "default" if shared else "single"
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), key=lambda f: (os.path.basename(f).split("_")[0:2]))
             if want in os.path.basename(f)]
    for f in reversed(files):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        if not _kernel(d, TRACE_PHOTON):
            continue
        share = {}
        try:
            with open(f.replace("_pmc.json", "_kernel_stats.csv")) as fh:
                for r in csv.DictReader(fh):
                    share[r["Name"].replace("void ", "").split("(")[0]] = float(r["Percentage"]) / 100.0
        except (OSError, ValueError, KeyError):
            pass
        return d, share, os.path.basename(f)
    return None


def valu_fields(c, seconds):
    """Vector-unit figures of one launch from its SQ counters (per launch) and its duration.  SQ_ACTIVE_INST_VALU counts the
    quad-cycles a wave has a vector instruction executing (an fp64 or transcendental instruction counts its longer issue),
    SQ_THREAD_CYCLES_VALU the same weighted by the lanes enabled."""
    act, thr, ins = c.get("SQ_ACTIVE_INST_VALU"), c.get("SQ_THREAD_CYCLES_VALU"), c.get("SQ_INSTS_VALU")
    if not act or seconds <= 0:
        return None
    out = {"valu_busy_frac": act * 4.0 / (seconds * SIMD_CYCLES_PER_S), "valu_wave_instructions": ins}
    if thr:
        out["lanes_active"] = thr / act
        out["lane_cycles"] = thr * 4.0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2_teapot_800x600")
    ap.add_argument("--rounds", type=int, default=0, help="rounds per GPU per step (default: the config's)")
    ap.add_argument("--cpu-rounds", type=int, default=4000, help="photon rounds (x100 photons) per CPU replica: ~15 s of CPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("SPPM_INFLIGHT", "5")),
                    help="replica iterations in flight per GPU (one context = one HIP stream each)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every GPU traces the config's photons per step (the pass grows with N); strong: the config's photons "
                         "are sharded over the N GPUs (config 3 as named: --config c3_teapot_800x600_10M --scaling strong --gpus 8)")
    ap.add_argument("--steady-steps", type=int, default=200,
                    help="iterations of a second timed region behind the first (value_steady): long enough for filling and draining "
                         "the pipeline of iterations in flight not to weigh; 0 skips it")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")

    # `bench.py --gpus N` started by hand (no launcher around it, WORLD_SIZE unset): start the N ranks here, as CHILD
    # processes of torch.distributed.run, before this process has imported torch or touched a GPU, and leave with their
    # exit code.  Under a launcher (the driver's torch.distributed.run line) WORLD_SIZE is set and must equal --gpus.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    from raytrace3_amd import SppmContext, scenes
    from raytrace3_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # one rank per GPU.  SPPM_BENCH_SHARE_GPU=1 lets the ranks of a rehearsal share the GPUs there are (two ranks on a
    # one-GPU box: the collective path runs for real, the numbers mean nothing)
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and os.environ.get("SPPM_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU of its own ({n_dev} visible); SPPM_BENCH_SHARE_GPU=1 shares them")
    device_index = local_rank % n_dev
    torch.cuda.set_device(device_index)
    # RCCL refuses a communicator with two ranks on one GPU: a rehearsal that shares GPUs reduces the images with gloo
    # over host copies instead (launch path, sharding and the world-size checks run for real; the collective does not)
    via_host = world > 1 and world > n_dev
    if world > 1:
        if via_host:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the RCCL process group has {dist.get_world_size()} ranks")
        world = dist.get_world_size()

    factory, (w, h), cfg_rounds, per_round = scenes.CONFIGS[args.config]
    rounds = args.rounds or cfg_rounds
    scene = factory()
    # Every step renders another replica: its own random streams and, as in the reference's loop (Raytracer.h:434), a camera
    # moved by 0.00015 x a random unit vector, so HitPoints, bucket count and deposit reach change from step to step.
    jit = np.random.default_rng(SEED).normal(size=(64, 3))
    cams = [scenes.default_camera(w, h, jitter=0.00015 * j / np.linalg.norm(j)) for j in jit]
    ctxs = [SppmContext(device_index) for _ in range(max(1, args.inflight))]
    # torch's view of each context's own HIP stream: the all-reduce of a context's image is issued under it, so it is
    # ordered with the context's passes by stream order and the host never waits inside the timed loop
    streams = [torch.cuda.ExternalStream(c.stream_handle(), device=torch.device("cuda", device_index)) for c in ctxs]
    for c in ctxs:
        c.set_scene(scene)
    n_lights = len(scene.light_list)
    # weak: every rank brings the config's photons, the pass is world x as many; strong: the config's pass as it is,
    # sharded (ranks' shares differ by one photon at most: distributed.photon_shard)
    total_rounds = rounds * world if args.scaling == "weak" else rounds
    photons_per_step = total_rounds * per_round * n_lights
    per_rank = photons_per_step / world
    pics = [torch.zeros((h, w, 3), dtype=torch.float32, device="cuda") for _ in ctxs]

    def iteration(k, i):
        D.sharded_iteration(ctxs[k], cams[i % len(cams)], SEED, i, total_rounds, per_round, n_lights, rank, world, pic_tensor=pics[k],
                            stream=streams[k], via_host=via_host)

    def step(i):
        iteration(i % len(ctxs), i)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up: every context allocates its buffers in its first iteration; keep that out of warm-up and timing
    for k in range(len(ctxs)):
        iteration(k, 0)
    fence()
    for i in range(args.warmup):
        step(i)
    fence()
    for c in ctxs:
        c.stats_mode(True)      # counters and HIP-event spans accumulate over the timed steps: no per-step readback
    def timed(first, n):
        """n steps from step index `first` on, between two fences; the slowest rank's time."""
        fence()
        t0 = time.perf_counter()
        for i in range(n):
            step(first + i)
        for c in ctxs:
            c.synchronize()         # waits for the library's stream and raises if a queued pass overflowed
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if via_host else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    elapsed = timed(args.warmup, args.steps)
    # HIP-event durations measured on the library's stream(s) and tallies, summed over the timed steps
    tms, cnts = [c.timings() for c in ctxs], [c.counters() for c in ctxs]
    tm = {k: sum(t[k] for t in tms) for k in tms[0]}
    counters = {k: (sum(c[k] for c in cnts) if k != "hitpoints" else cnts[0][k]) for k in cnts[0]}
    footprint = sum(c.device_stats()["device_bytes"] for c in ctxs)
    PHASES = ("eye_trace_ms", "grid_build_ms", "photon_trace_ms", "deposit_sort_ms", "gather_ms", "resolve_ms")
    acc = {k: tm[k] for k in PHASES}
    # The kernels' own durations: with several iterations in flight a HIP-event span also covers whatever the other
    # streams ran in between, so the roofline figures come from a few more iterations with ONE in flight (same work,
    # same stream and events, outside the timed region); rocprofv3 serialises kernels and reports these durations.
    ALONE_STEPS = 4
    if len(ctxs) > 1:
        ctxs[0].stats_mode(True)
        for i in range(ALONE_STEPS):
            iteration(0, args.warmup + args.steps + i)
        ctxs[0].synchronize()
        fence()
        tm1, cn1, ds1, alone_steps = ctxs[0].timings(), ctxs[0].counters(), ctxs[0].device_stats(), ALONE_STEPS
    else:
        tm1, cn1, ds1, alone_steps = tm, counters, ctxs[0].device_stats(), args.steps
    # A second, longer region: with five iterations in flight a region of `steps` iterations spends a good part of its
    # time filling and draining that pipeline (at --steps 20 the whole region is 17 ms), which the driver's `value` contains
    # by contract; the steady-state rate is reported beside it.
    steady_s = timed(args.warmup + args.steps + ALONE_STEPS, args.steady_steps) if args.steady_steps > 0 else None
    alone = {k: tm1[k] / alone_steps for k in PHASES}
    gl, tl = max(1, tm1["gather_launches"]), max(1, tm1["trace_launches"] // 2)      # per step: one eye and one photon trace launch

    if rank == 0:
        photons_total = photons_per_step * args.steps
        pass_s = (alone["photon_trace_ms"] + alone["deposit_sort_ms"] + alone["gather_ms"]) / 1e3    # one photon pass, undisturbed
        how = ("HIP events on the library's stream" + (f", {ALONE_STEPS} iterations with one iteration in flight after the timed region (spans of "
               "the timed region include the other streams' kernels), launched as in the timed region: persistent grids sized to share "
               "the GPU with the other contexts' kernels" if len(ctxs) > 1 else ", over the timed region"))
        # the committed rocprofv3 counters are of the default workload: they say nothing about another configuration
        prof = committed_profile(len(ctxs) > 1) if (args.config == "c2_teapot_800x600" and rounds == cfg_rounds) else None
        pk, share, pname = prof if prof else ({}, {}, None)
        cur_sha = kernel_source_sha256()
        prof_sha = pk.get("_meta", {}).get("kernel_source_sha256") if pname else None
        stale = None if not pname else (None if prof_sha is None else prof_sha != cur_sha)
        src = (f"profiles/{pname}: rocprofv3 --pmc passes of this command, per-launch averages; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; "
               "kernels run one at a time under the profiler") if pname else None

        # ---- dominant kernel: the photon trace (persistent wavefronts, fp64 ray state): bound by vector-instruction issue
        trace_s = alone["photon_trace_ms"] * alone_steps / tl / 1e3
        tc = _kernel(pk, TRACE_PHOTON)
        tv = valu_fields(tc, trace_s) if tc else None
        lane_cycles = tv.get("lane_cycles") if tv else None
        roofline = {
            "kernel": "k_trace_photon_bez (k_trace_photon_plain for a scene without Bezier objects): photon emission + PhotonTrace on persistent "
                      "wavefronts; fp64 ray state, in-wave Bezier solves",
            "share_of_gpu_time": next((v for k, v in share.items() if k.startswith(TRACE_PHOTON)), None),
            "bound": "valu", "unit": "Tlane-cycle/s", "peak": PEAK_TLANE_CYCLES,
            "achieved": lane_cycles / trace_s / 1e12 if lane_cycles else None,
            "frac": lane_cycles / trace_s / 1e12 / PEAK_TLANE_CYCLES if lane_cycles else None,
            "frac_is": "lane-cycles with a vector instruction executing (SQ_THREAD_CYCLES_VALU x 4: fp64 and transcendental instructions at "
                       "their real issue cost, disabled lanes not counted) / (256 CU x 4 SIMD x 64 lanes x 2.4 GHz x launch duration)",
            "valu_busy_frac": tv["valu_busy_frac"] if tv else None, "lanes_active": tv.get("lanes_active") if tv else None,
            "valu_wave_instructions_per_launch": tv["valu_wave_instructions"] if tv else None,
            "traffic": tc.get("hbm_traffic_bytes_per_launch") if tc else None,
            "hbm_frac": (tc["hbm_traffic_bytes_per_launch"] / trace_s / 1e9 / HBM_PEAK_GBS) if tc and tc.get("hbm_traffic_bytes_per_launch") else None,
            "avg_launch_ms": trace_s * 1e3, "segments_per_launch": cn1["segments"] / tl,
            "counters_source": src, "measured": how,
            "counters_stale": stale,
            "counters_stale_is": "true: the kernel sources (sha256 over " + " + ".join(KERNEL_SOURCES) + ") are not the ones the committed counters were "
                                 "collected from, only the durations are live; null: no committed counters in use, or a profile from before profiles were stamped",
        }
        # ---- the gather: what HBM sees of it against what it must move, and its vector-unit figures
        gather_s = alone["gather_ms"] * alone_steps / gl / 1e3
        gc, gs = _kernel(pk, "k_gather"), _kernel(pk, "k_gather_split")
        g_traffic = (gc.get("hbm_traffic_bytes_per_launch", 0.0) + gs.get("hbm_traffic_bytes_per_launch", 0.0)) if gc.get("hbm_traffic_bytes_per_launch") else None
        both = {k: gc.get(k, 0.0) + gs.get(k, 0.0) for k in ("SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU") if gc.get(k)}
        gv = valu_fields(both, gather_s) if both else None
        stored, n_hp = ds1["stored_deposits"] / gl, cn1["hitpoints"]
        compulsory = 64.0 * stored + 96.0 * n_hp                      # every kept deposit read once; HitPoint 32 + 16 + 16 + 16 B read, 16 B written
        ref_bytes = (40.0 * cn1["candidates"] + 32.0 * cn1["accepted"] + 64.0 * cn1["segments"]) / gl
        roofline_gather = {
            "kernel": "k_gather + k_gather_split (the radius gather of one photon pass)",
            "share_of_gpu_time": sum(v for k, v in share.items() if k.startswith("k_gather")) or None,
            "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "achieved": g_traffic / gather_s / 1e9 if g_traffic else None,
            "frac": g_traffic / gather_s / 1e9 / HBM_PEAK_GBS if g_traffic else None,
            "traffic": g_traffic, "compulsory_bytes": compulsory,
            "compulsory_is": "64 B per stored deposit read once + 96 B per HitPoint (position and key 32, normal 16, weight 16, tao read and written 16 + 16)",
            "traffic_over_compulsory": g_traffic / compulsory if g_traffic and compulsory else None,
            "reference_equivalent_bytes": ref_bytes,
            "reference_equivalent_is": "40 B per pair in range + 32 B per accepted update + 64 B per photon segment: what the reference's radius search and "
                                       "update loop touch per pass; divided by the launch duration it is a re-use figure, not a bandwidth",
            "reuse_factor": ref_bytes / g_traffic if g_traffic else None,
            "pair_tests_per_launch": ds1["pair_tests"] / gl,
            "pairs_in_range_frac": cn1["candidates"] / ds1["pair_tests"] if ds1["pair_tests"] else None,
            "valu_busy_frac": gv["valu_busy_frac"] if gv else None, "lanes_active": gv.get("lanes_active") if gv else None,
            "valu_wave_instructions_per_launch": gv["valu_wave_instructions"] if gv else None,
            "avg_launch_ms": gather_s * 1e3, "counters_source": src, "measured": how,
        }
        # ---- the GPU as a whole over the timed region: with several iterations in flight the kernels of different iterations
        # share the vector units, and no single kernel's figure says how busy those are
        iter_s = elapsed / args.steps
        big = [_kernel(pk, TRACE_PHOTON), _kernel(pk, TRACE_EYE), gc, gs]      # one launch each per iteration
        if all(k.get("SQ_ACTIVE_INST_VALU") for k in big) and iter_s > 0:
            act = sum(k["SQ_ACTIVE_INST_VALU"] for k in big)
            thr = sum(k.get("SQ_THREAD_CYCLES_VALU", 0.0) for k in big)
            roofline["timed_region_all_kernels"] = {
                "valu_busy_frac": act * 4.0 / (iter_s * SIMD_CYCLES_PER_S),
                "lane_frac": thr * 4.0 / (iter_s * SIMD_CYCLES_PER_S * 64.0) if thr else None,
                "is": "vector-unit cycles (and lane-cycles) of the four large kernels of an iteration (photon trace, eye trace, gather, gather split; "
                      "committed counters, one launch each) / (wall time of an iteration in the timed region x 256 CU x 4 SIMD [x 64 lanes]): how busy "
                      "the vector units of the whole GPU are while the kernels of several iterations share them",
            }
        out = {
            "metric": "Mphotons/s", "value": photons_total / elapsed / 1e6, "unit": "Mphotons/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "timed_region_ms": elapsed * 1e3,
            "value_steady": photons_per_step * args.steady_steps / steady_s / 1e6 if steady_s else None,
            "value_steady_is": (f"the same steps, {args.steady_steps} of them in a second region behind the timed one ({steady_s * 1e3:.0f} ms): filling and "
                                f"draining the {len(ctxs)} iterations in flight no longer weighs") if steady_s else None,
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "kernel_source_sha256": cur_sha,
            "vs_baseline": None, "dtype": "f64 rays/gather + f32 Newton/flux", "data": "synthetic",
            "config": {"inflight": len(ctxs), "workload": f"{args.config}: {w}x{h}, {photons_per_step} photons per iteration "
                                   f"({total_rounds} rounds x {per_round} x {n_lights} light(s)), {per_rank:.0f} per GPU, one replica per step (own random "
                                   f"streams, camera jittered by 0.00015 as Raytracer.h:434), photon-sharded over {world} GPU(s), {args.scaling} scaling",
                       "hitpoints": counters["hitpoints"],
                       "collective": None if world == 1 else ("gloo over host copies (rehearsal: ranks share GPUs)" if via_host else "rccl all-reduce of the flux image")},
            "photon_pass_mphotons_per_s": per_rank / pass_s / 1e6 if pass_s > 0 else None,
            "photon_pass_is": "the metric by its own definition (trace + gather of one pass, one iteration in flight); `value` is whole iterations, several in flight",
            "kernel_ms_per_step": {k: v / args.steps for k, v in acc.items()},
            "kernel_ms_per_step_one_in_flight": alone,
            "counters_per_step": {k: (v / args.steps if k != "hitpoints" else v) for k, v in counters.items()},
            "device_bytes_all_contexts": footprint,
            "roofline": roofline, "roofline_gather": roofline_gather,
        }
        if not args.no_cpu_baseline and world == 1:
            base = cpu_baseline(scene, w, h, args.cpu_rounds, 4)
            # the reference's loop is four threads (Raytracer.h:442); beside it the same replicas on every core this job may use
            cores = max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16)))
            if cores > 4:
                allc = cpu_baseline(scene, w, h, max(1, args.cpu_rounds * 4 // cores), cores)
                base["all_cores"] = {k: allc[k] for k in ("value", "unit", "cores", "sample")}
            # ... and without the sweep that the reference pays for nothing: the baseline a CPU implementation free of it sets
            # (a larger sample: its fixed cost, the grid over the HitPoints, would weigh on a short one)
            tuned = cpu_baseline(scene, w, h, args.cpu_rounds * 2, cores, literal=False)
            base["all_cores_grid_gather_no_update_sweep"] = {k: tuned[k] for k in ("value", "unit", "cores", "sample")}
            out["cpu_baseline"] = base
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
