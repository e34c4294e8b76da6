"""Diagnostic (GPU box): the fraction of a pass's deposits that have a HitPoint within one grid cell, i.e. that any
gather can reach, per configuration.  python tools/deposit_reach.py"""
import numpy as np, sys
sys.path.insert(0, '.')
from raytrace3_amd import SppmContext, scenes
for name, scene, (w, h) in (("C2", scenes.scene_teapot(1), (800, 600)), ("C4", scenes.scene_teapot(4), (1920, 1080)), ("C1", scenes.scene_spheres_plane(False), (256, 256))):
    g = SppmContext(0); g.set_scene(scene)
    g.eye_pass(scenes.default_camera(w, h), 20260101, 0)
    hp = g.hitpoints()
    dep = g.deposits(10000, 100, 20260101, 0, photon_begin=0, photon_count=100000, cap=1 << 21)
    cs = np.sqrt(2.0) * 1.0001
    def pack(c): return ((c[:, 0] + (1 << 20)) << 42) | ((c[:, 1] + (1 << 20)) << 21) | (c[:, 2] + (1 << 20))
    hc = np.floor(hp["pos"] / cs).astype(np.int64)
    occ = []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                occ.append(pack(hc + np.array([dx, dy, dz])))
    occ = np.unique(np.concatenate(occ))
    dc = np.floor(dep[:, 2:5] / cs).astype(np.int64)
    useful = np.isin(pack(dc), occ)
    print(name, "hitpoints", len(hp), "deposits", len(dep), "useful fraction", useful.mean(), flush=True)
    g.close()
