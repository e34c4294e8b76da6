import sys, os
sys.path.insert(0, '.')
which = sys.argv[1]
def attempt(label, fn):
    try:
        fn(); print(which, label, "ok", flush=True)
    except Exception as e:
        print(which, label, "FAILED", type(e).__name__, str(e)[:60].replace("\n", " "), flush=True)
if which == "torch_first":
    import torch
    attempt("zeros", lambda: torch.zeros(1, device="cuda"))
from raytrace3_amd import SppmContext, scenes
g = SppmContext(0)
if which == "ctx_only_then_stream":
    import torch
    attempt("stream", lambda: torch.cuda.Stream())
    attempt("zeros", lambda: torch.zeros(1, device="cuda"))
elif which == "ctx_then_zeros":
    import torch
    attempt("zeros", lambda: torch.zeros(1, device="cuda"))
    attempt("stream", lambda: torch.cuda.Stream())
elif which == "ctx_scene_then_zeros":
    g.set_scene(scenes.scene_spheres_plane(False))
    import torch
    attempt("zeros", lambda: torch.zeros(1, device="cuda"))
elif which == "torch_first":
    attempt("stream", lambda: torch.cuda.Stream())
elif which == "import_torch_before_ctx_use_after":
    pass
import subprocess
print(which, [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l][:6], flush=True)
