"""Diagnostic (GPU box): the rate of whole iterations when the host asks for the flux image after every one
(sppm_resolve: device -> pinned host copy + widening to doubles), against bench.py, which leaves it in HBM."""
import sys, time
sys.path.insert(0, '.')
from raytrace3_amd import SppmContext, scenes
factory, (w, h), rounds, per = scenes.CONFIGS["c2_teapot_800x600"]
ctxs = [SppmContext(0), SppmContext(0)]
cam = scenes.default_camera(w, h)
for g in ctxs:
    g.set_scene(factory())
    g.eye_pass(cam, 1, 0); g.photon_pass(rounds, per, 1, 0); g.resolve()
for n_ctx in (1, 2):
    t0 = time.perf_counter(); n = 20
    for i in range(n):
        g = ctxs[i % n_ctx]
        g.eye_pass(cam, 1, i); g.photon_pass(rounds, per, 1, i); pic = g.resolve()
    dt = time.perf_counter() - t0
    print(f"host image every iteration, {n_ctx} context(s): {dt / n * 1e3:.3f} ms per iteration, {rounds * per * n / dt / 1e6:.1f} Mphotons/s", flush=True)
