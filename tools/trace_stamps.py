#!/usr/bin/env python3
"""Where a trace-loop iteration spends its cycles (diagnostic): build the library with -DTRACE_STAMPS
(hipcc ... -DTRACE_STAMPS -shared -o /tmp/libsppm_stamps.so raytrace3_amd/csrc/sppm_hip.hip), then on the GPU box
    SPPM_VERBOSE=1 SPPM_LIB_PATH=/tmp/libsppm_stamps.so python tools/trace_stamps.py
prints the cycle sums per loop section (refill / analytic / bezier / merge / finish / bounce) after an eye
pass and after eye + photon pass.  Shares only: a stamped build is slower than the real one."""
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
