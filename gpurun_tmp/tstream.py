import sys, os
sys.path.insert(0, '.')
import numpy as np
from raytrace3_amd import SppmContext, scenes
which = sys.argv[1]
g = SppmContext(0)
g.set_scene(scenes.scene_spheres_plane(False))
if which in ("eye", "oracle"):
    g.eye_pass(scenes.default_camera(16, 16), 1)
if which == "oracle":
    from tests import oracle_lib as ol
    o = ol.OracleContext(); o.set_scene(scenes.scene_spheres_plane(False)); o.eye_pass(scenes.default_camera(16, 16), 1)
if which == "debug":
    org = np.array([[50., 35., 230.]]); d = np.array([[0., 0., -1.]])
    print(g.near_collide(org, d, 1)["dist"])
import torch
print(which, "torch sees", torch.cuda.device_count(), flush=True)
try:
    st = torch.cuda.Stream()
    print(which, "stream ok", flush=True)
except Exception as e:
    print(which, "stream FAILED", type(e).__name__, str(e)[:80], flush=True)
