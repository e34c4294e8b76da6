#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle (parity RNG mode,
fp64).  The reference renderer ships no fixtures of its own and cannot be built in this
environment, so these vectors pin the oracle's behaviour over time (a regression pin) and
give the GPU tests seeded expected outputs that do not need the oracle at run time.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from raytrace3_amd import scenes  # noqa: E402
from tests import oracle_lib as ol  # noqa: E402

SEED = 20260101


def hitpoints_and_image():
    scene = scenes.scene_spheres_plane(True)
    cam = scenes.default_camera(40, 32)
    o = ol.OracleContext()
    o.set_scene(scene)
    o.eye_pass(cam, SEED)
    o.photon_pass(100, 100, SEED)
    hp = o.hitpoints()
    np.savez_compressed(os.path.join(HERE, "c1b_40x32_hitpoints.npz"), w=40, h=32, seed=SEED, rounds=100, per_round=100,
                        x=hp["x"], y=hp["y"], node=hp["node"], obj=hp["obj"], pos=hp["pos"], n=hp["n"],
                        wgt=hp["wgt"], newN=hp["newN"], tao=hp["tao"], pic=o.resolve(),
                        counters=np.array(list(o.counters().values()), dtype=np.int64),
                        counter_names=np.array(list(o.counters().keys())))


def bernstein(t):
    return np.array([(1 - t) ** 3, 3 * t * (1 - t) ** 2, 3 * t * t * (1 - t), t ** 3])


def newton_targets(P, patch, rng):
    """Aim points: interior surface points S(u,v), (u,v) in [0.1,0.9]^2, three quarters of them
    exactly on the surface and a quarter pushed 0.3 off it (near misses)."""
    tgt = []
    for k, p in enumerate(patch):
        u, v = rng.random(2) * 0.8 + 0.1
        cp = P[p].reshape(4, 4, 3)          # [i (v index)][k (u index)]
        s = np.einsum("i,k,ikc->c", bernstein(v), bernstein(u), cp)
        if k % 4 == 3:
            s = s + rng.normal(size=3) * 0.3
        tgt.append(s)
    return np.array(tgt)


def newton_cases():
    scene = scenes.scene_teapot(1, textured=False)
    P = scene.patches
    rng = np.random.default_rng(3)
    n = 256
    patch = rng.integers(0, 32, size=n).astype(np.int32)
    org = np.tile([50., 35., 230.], (n, 1)) + rng.normal(size=(n, 3)) * 20
    tgt = newton_targets(P, patch, rng)
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o64 = ol.OracleContext()
    o64.set_scene(scene)
    o32 = ol.OracleContext(newton_fp32=True)
    o32.set_scene(scene)
    np.savez_compressed(os.path.join(HERE, "teapot_newton_256.npz"), seed=SEED, org=org, dir=d, patch=patch,
                        tuv_fp64=o64.newton(org, d, patch, SEED), tuv_fp32=o32.newton(org, d, patch, SEED))


def photon_deposits():
    scene = scenes.scene_teapot(1, textured=True)
    o = ol.OracleContext()
    o.set_scene(scene)
    dep = o.deposits(2, 100, SEED, photon_begin=0, photon_count=200)
    np.savez_compressed(os.path.join(HERE, "teapot_deposits_200photons.npz"), seed=SEED, rounds=2, per_round=100,
                        deposits=dep)


if __name__ == "__main__":
    hitpoints_and_image()
    newton_cases()
    photon_deposits()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
