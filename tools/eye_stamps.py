#!/usr/bin/env python3
"""Cycle sums per section of the trace loop for the eye pass alone (a -DTRACE_STAMPS build of the library:
SPPM_LIB_PATH=ab/stamps.so SPPM_VERBOSE=1 python tools/eye_stamps.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytrace3_amd import SppmContext, scenes

for name in ("c2_teapot_800x600", "c4_teapot4_1080p"):
    factory, (w, h), rounds, per = scenes.CONFIGS[name]
    ctx = SppmContext(0)
    ctx.set_scene(factory())
    cam = scenes.default_camera(w, h)
    for it in range(3):
        ctx.eye_pass(cam, 1, it)
    print(name, "eye pass alone:", {k: v for k, v in ctx.counters().items() if k in ("hitpoints", "eye_segments", "newton_calls")}, ctx.timings(), flush=True)
    sys.stderr.flush()
    ctx.close()
