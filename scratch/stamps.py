import sys, os; sys.path.insert(0, '.')
from raytrace3_amd import SppmContext, scenes
g = SppmContext(0)
factory,(w,h),r,p = scenes.CONFIGS["c2_teapot_800x600"]
g.set_scene(factory()); cam = scenes.default_camera(w,h)
for i in range(2):
    g.eye_pass(cam, 1, i)
    print("eye:", file=sys.stderr); g.counters()
    g.eye_pass(cam, 1, i)   # restart tallies so the photon stamps stand alone? (eye stamps included below)
    g.photon_pass(r, p, 1, i)
    print("eye+photon:", file=sys.stderr); g.counters()
