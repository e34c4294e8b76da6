#!/usr/bin/env python3
"""Renders a configuration for a number of SPPM iterations (jittered replicas as SPPMRayTracer::render draws them,
Raytracer.h:429-447) into a film on the GPU and writes the tone-mapped picture:

    python tools/render_picture.py c2_teapot_800x600 400 out.png

The picture is what the reference hands to cv::imwrite (rows flipped, BGR bytes: Raytracer.h:460-470), written here
in RGB order so that a viewer shows the colours the reference's JPEG shows."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from raytrace3_amd import Film, SppmContext, scenes  # noqa: E402


def main():
    config, iters, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    factory, (w, h), rounds, per_round = scenes.CONFIGS[config]
    scene = factory()
    ctxs = [SppmContext(0) for _ in range(int(os.environ.get("SPPM_INFLIGHT", "3")))]
    for c in ctxs:
        c.set_scene(scene)
    film = Film(w, h)
    rng = np.random.default_rng(1)
    t0 = time.perf_counter()
    for i in range(iters):
        c = ctxs[i % len(ctxs)]
        j = rng.normal(size=3)
        c.eye_pass(scenes.default_camera(w, h, jitter=0.00015 * j / np.linalg.norm(j)), 20260101, i)
        c.photon_pass(rounds, per_round, 20260101, i)
        film.add(c)
    for c in ctxs:
        c.synchronize()
    rgb = film.tonemap(iters, rgb=True)
    dt = time.perf_counter() - t0
    print(f"{iters} iterations of {rounds * per_round} photons at {w}x{h}: {dt:.2f} s")
    from PIL import Image
    Image.fromarray(np.asarray(rgb, dtype=np.uint8).reshape(h, w, 3), "RGB").save(out)
    film.close()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
