"""What the Bezier object costs the trace kernels: the bench scene with and without its teapot, one context alone on the
GPU (full grids), eye and photon trace time per iteration from the library's HIP events."""
import sys
sys.path.insert(0, ".")
from raytrace3_amd import SppmContext, scenes

for n in (1, 0):
    ctx = SppmContext(0)
    ctx.set_scene(scenes.scene_teapot(n))
    cam = scenes.default_camera(800, 600)
    for i in range(3):                      # set-up iterations: buffers, deposit layout
        ctx.eye_pass(cam, 1, i); ctx.photon_pass(10000, 100, 1, i)
    ctx.synchronize()
    ctx.stats_mode(True)
    t0 = ctx.timings()
    K = 10
    for i in range(K):
        ctx.eye_pass(cam, 1, 10 + i); ctx.photon_pass(10000, 100, 1, 10 + i)
    ctx.synchronize()
    t1 = ctx.timings()
    print("teapots", n, {k: round((t1[k] - t0[k]) / K, 4) for k in t1 if k.endswith("_ms")}, {k: v for k, v in ctx.counters().items() if k in ("segments", "newton_calls", "eye_segments")})
    ctx.close()
