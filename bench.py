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


def cpu_baseline(scene, w, h, rounds, threads):
    """The reference's CPU path as restated by the oracle (fp64, global MWC stream, kd-tree
    radius search + the per-round update() sweep), `threads` replicas in parallel like the
    reference's `omp parallel for num_threads(4)`.  Bounded sample: full-resolution eye pass,
    `rounds` x 100 photons per replica."""
    from raytrace3_amd import scenes
    from tests import oracle_lib as ol
    lib = ol.load()
    ctxs = [ol.OracleContext(lib, rng_literal=True) for _ in range(threads)]
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
    t_ph = run(lambda i: ctxs[i].photon_pass(rounds, 100, SEED, i, mode=ol.GATHER_KDTREE))
    photons = threads * rounds * 100
    for c in ctxs:
        c.close()
    return {
        "value": photons / t_ph / 1e6, "unit": "Mphotons/s", "cores": threads, "kind": "port",
        "sample": f"{threads} replicas x {rounds * 100} photons ({rounds} rounds x 100) on the {w}x{h} teapot scene, "
                  f"photon pass {t_ph:.1f} s (eye pass {t_eye:.1f} s not counted), oracle literal mode: fp64, kd-tree "
                  f"radius search, per-round update() sweep",
        "eye_pass_s": t_eye, "photon_pass_s": t_ph,
    }


VALU_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9      # CUs x SIMDs x lanes per cycle x 2.4 GHz: vector instructions per lane per second


def _kernel(d, name):
    """The summary's entry for a kernel, template arguments or not (k_gather, k_gather<true>)."""
    for k, v in d.items():
        if k == name or k.startswith(name + "<"):
            return v
    return {}


def _profiles(shared):
    """The committed PMC summaries, oldest first, of the launch mode asked for: several iterations in flight (the
    default command; its persistent grids are sized to share the GPU) or one."""
    import glob
    want = "default" if shared else "single"
    return [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))) if want in os.path.basename(f)]


def measured_valu(shared):
    """Vector instructions per launch of the two dominant kernels from the committed PMC profile
    (SQ_INSTS_VALU, wave instructions): both kernels live on VALU issue, not on HBM."""
    best = None
    for f in _profiles(shared):
        try:
            with open(f) as fh:
                d = json.load(fh)
            tk = [v for k, v in d.items() if k.startswith("k_trace<1, true")]      # the photon pass with in-wave solves
            g, t = _kernel(d, "k_gather").get("SQ_INSTS_VALU"), (tk[0].get("SQ_INSTS_VALU") if tk else None)
            if g and t:
                g += _kernel(d, "k_gather_split").get("SQ_INSTS_VALU") or 0.0
                best = {"k_gather_wave_insts": float(g), "k_trace_photon_wave_insts": float(t), "source": "profiles/" + os.path.basename(f)}
        except (OSError, ValueError):
            pass
    return best


def measured_traffic(shared):
    """HBM bytes per gather (k_gather and k_gather_split) from the rocprofv3 PMC passes committed under profiles/
    ((2 * FETCH_SIZE + WRITE_SIZE) * 1024, collected by tools/gpu_profile.sh on this same command);
    None when no such profile is in the tree."""
    best = None
    for f in _profiles(shared):
        try:
            with open(f) as fh:
                d = json.load(fh)
            t = _kernel(d, "k_gather").get("hbm_traffic_bytes_per_launch")
            if t:      # the heavy-cell pieces are part of the same gather
                t2 = _kernel(d, "k_gather_split").get("hbm_traffic_bytes_per_launch") or 0.0
                best = (float(t) + float(t2), os.path.basename(f))
        except (OSError, ValueError):
            pass
    return best


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
    cam = scenes.default_camera(w, h)
    ctxs = [SppmContext(device_index) for _ in range(max(1, args.inflight))]
    # torch's view of each context's own HIP stream: the all-reduce of a context's image is issued under it, so it is
    # ordered with the context's passes by stream order and the host never waits inside the timed loop
    streams = [torch.cuda.ExternalStream(c.stream_handle(), device=torch.device("cuda", device_index)) for c in ctxs]
    for c in ctxs:
        c.set_scene(scene)
    ctx = ctxs[0]
    n_lights = len(scene.light_list)
    per_rank = rounds * per_round * n_lights
    total_rounds = rounds * world
    pics = [torch.zeros((h, w, 3), dtype=torch.float32, device="cuda") for _ in ctxs]

    def step(i):
        k = i % len(ctxs)
        D.sharded_iteration(ctxs[k], cam, SEED, i, total_rounds, per_round, n_lights, rank, world, pic_tensor=pics[k], stream=streams[k],
                            via_host=via_host)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up: every context allocates its buffers in its first iteration; keep that out of warm-up and timing
    for k in range(len(ctxs)):
        D.sharded_iteration(ctxs[k], cam, SEED, 0, total_rounds, per_round, n_lights, rank, world, pic_tensor=pics[k], stream=streams[k],
                            via_host=via_host)
    fence()
    for i in range(args.warmup):
        step(i)
    fence()
    for c in ctxs:
        c.stats_mode(True)      # counters and HIP-event spans accumulate over the timed steps: no per-step readback
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    for c in ctxs:
        c.synchronize()         # waits for the library's stream and raises if a queued pass overflowed
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if via_host else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # HIP-event durations measured on the library's stream(s) and tallies, summed over the timed steps
    tms, cnts = [c.timings() for c in ctxs], [c.counters() for c in ctxs]
    tm = {k: sum(t[k] for t in tms) for k in tms[0]}
    counters = {k: (sum(c[k] for c in cnts) if k != "hitpoints" else cnts[0][k]) for k in cnts[0]}
    PHASES = ("eye_trace_ms", "grid_build_ms", "photon_trace_ms", "deposit_sort_ms", "gather_ms", "resolve_ms")
    acc = {k: tm[k] for k in PHASES}
    # The kernels' own durations: with several iterations in flight a HIP-event span also covers whatever the other
    # stream ran in between, so the roofline figures come from a few more iterations with ONE in flight (same work,
    # same stream and events, outside the timed region); rocprofv3 serialises kernels and reports these durations.
    ALONE_STEPS = 4
    if len(ctxs) > 1:
        ctxs[0].stats_mode(True)
        for i in range(ALONE_STEPS):
            D.sharded_iteration(ctxs[0], cam, SEED, args.warmup + args.steps + i, total_rounds, per_round, n_lights, rank, world,
                                pic_tensor=pics[0], stream=streams[0], via_host=via_host)
        ctxs[0].synchronize()
        fence()
        tm1, cn1 = ctxs[0].timings(), ctxs[0].counters()
        alone = {k: tm1[k] / ALONE_STEPS for k in PHASES}
        alone_launches = tm1["gather_launches"]
        alone_bytes = 40.0 * cn1["candidates"] + 32.0 * cn1["accepted"] + 64.0 * cn1["segments"]
    else:
        alone = {k: acc[k] / args.steps for k in PHASES}
        alone_launches = tm["gather_launches"]
        alone_bytes = 40.0 * counters["candidates"] + 32.0 * counters["accepted"] + 64.0 * counters["segments"]
    alone_steps = ALONE_STEPS if len(ctxs) > 1 else args.steps
    gather_launches = alone_launches
    bytes_model = alone_bytes

    if rank == 0:
        photons_total = per_rank * world * args.steps
        # the committed rocprofv3 counters are of the default workload: they say nothing about another configuration
        profiled = args.config == "c2_teapot_800x600" and rounds == cfg_rounds
        traffic = measured_traffic(len(ctxs) > 1) if profiled else None
        gather_s = alone["gather_ms"] * alone_steps / 1e3
        achieved = bytes_model / gather_s / 1e9 if gather_s > 0 else 0.0
        pass_s = (alone["photon_trace_ms"] + alone["deposit_sort_ms"] + alone["gather_ms"]) / 1e3    # one photon pass, undisturbed
        out = {
            "metric": "Mphotons/s", "value": photons_total / elapsed / 1e6, "unit": "Mphotons/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 rays/gather + f32 Newton/flux", "data": "synthetic",
            "config": {"inflight": len(ctxs), "workload": f"{args.config}: teapot.bpt Cornell scene {w}x{h}, {per_rank} photons/iter/GPU "
                                   f"({rounds} rounds x {per_round}), one replica per step, photon-sharded over {world} GPU(s)",
                       "hitpoints": counters["hitpoints"]},
            "photon_pass_mphotons_per_s": per_rank / pass_s / 1e6 if pass_s > 0 else None,
            "kernel_ms_per_step": {k: v / args.steps for k, v in acc.items()},
            "kernel_ms_per_step_one_in_flight": alone,
            "counters_per_step": {k: (v / args.steps if k != "hitpoints" else v) for k, v in counters.items()},
            "roofline": {"kernel": "k_gather + k_gather_split (the gather of one photon pass)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0] if traffic else None,
                         "traffic_source": ("profiles/" + traffic[1] + ": (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, rocprofv3 --pmc")
                         if traffic else None,
                         "bytes_model": "40*candidates + 32*accepted + 64*segments per launch (algorithmic bytes of the "
                                        "reference's radius search + update loop; the cell-shared sweep re-uses them on chip)",
                         "algorithmic_bytes_per_launch": bytes_model / max(1, gather_launches),
                         "avg_launch_ms": alone["gather_ms"] * alone_steps / max(1, gather_launches),
                         "measured": "HIP events on the library's stream" + (f", {ALONE_STEPS} iterations with one iteration in flight after the "
                                     "timed region (spans of the timed region include the other streams' kernels), launched as in the "
                                     "timed region: persistent grids sized to share the GPU with the other contexts' kernels" if len(ctxs) > 1 else
                                     ", over the timed region")},
        }
        valu = measured_valu(len(ctxs) > 1) if profiled else None
        if valu:
            gl = max(1, gather_launches)
            valu["k_gather_frac_of_valu_peak"] = valu["k_gather_wave_insts"] * 64 / (alone["gather_ms"] * alone_steps / gl / 1e3) / VALU_PEAK_LANE_OPS
            valu["k_trace_photon_frac_of_valu_peak"] = (valu["k_trace_photon_wave_insts"] * 64 /
                                                        (alone["photon_trace_ms"] * alone_steps / gl / 1e3) / VALU_PEAK_LANE_OPS)
            valu["note"] = "wave instructions x 64 lanes / HIP-event duration with one iteration in flight / (256 CU x 4 SIMD x 16 lanes x 2.4 GHz)"
            out["valu"] = valu
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, w, h, args.cpu_rounds, 4)
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
