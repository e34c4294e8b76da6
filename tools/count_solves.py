"""Newton solves per iteration at the bench scene: prints the device's tally after one eye pass and one photon pass."""
import sys
sys.path.insert(0, ".")
from raytrace3_amd import SppmContext, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = SppmContext(0)
ctx.set_scene(scenes.scene_teapot(n))
cam = scenes.default_camera(800, 600)
ctx.eye_pass(cam, 1)
c0 = dict(ctx.counters())
ctx.photon_pass(1_000_000, 100, 2)
c1 = dict(ctx.counters())
print("eye solves", c0["newton_calls"], "eye segments", c0["eye_segments"])
print("all solves", c1["newton_calls"], "photon segments", c1["segments"])
