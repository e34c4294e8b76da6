"""One-off size check (GPU box): 3840x2160, four teapots, 20 M photons through eye pass, photon pass, film and tone map."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from raytrace3_amd import SppmContext, Film, scenes
g = SppmContext(0); g.set_scene(scenes.scene_teapot(4))
w, h = 3840, 2160
print("eye 4K", flush=True)
t = time.time(); g.eye_pass(scenes.default_camera(w, h), 1, 0); print("eye done", time.time() - t, flush=True)
print("photons 20M", flush=True)
t = time.time(); g.photon_pass(200000, 100, 1, 0); g.synchronize(); print("photon done", time.time() - t, flush=True)
c = g.counters(); print(c, flush=True)
assert c["photons"] == 20_000_000
hp = g.hitpoints(); assert int(hp["newN"].sum()) == c["accepted"]
f = Film(w, h); f.add(g); img = f.tonemap(1.0); print("tonemap mean", img.mean(), flush=True)
print(g.timings(), flush=True)
