import sys; sys.path.insert(0, '.')
import numpy as np
from raytrace3_amd import SppmContext, scenes
from tests import oracle_lib as ol
SEED=20260101
def key(hp): return (hp["x"].astype(np.int64) << 40) | (hp["y"].astype(np.int64) << 20) | hp["node"].astype(np.int64)
g=SppmContext(0)
for name in ("c1_spheres_plane_256","c2_teapot_800x600"):
    factory,(w,h),r,p=scenes.CONFIGS[name]
    sc=factory(); cam=scenes.default_camera(w,h)
    g.set_scene(sc); g.eye_pass(cam,SEED)
    o=ol.OracleContext(); o.set_scene(sc); o.eye_pass(cam,SEED)
    a,b=g.hitpoints(),o.hitpoints()
    _,ia,ib=np.intersect1d(key(a),key(b),return_indices=True)
    a,b=a[ia],b[ib]
    same=(a["obj"]==b["obj"])&(b["obj"]!=8)
    err=np.abs(a["pos"]-b["pos"]).max(axis=1)
    rel=err/np.maximum(1,np.abs(b["pos"]).max(axis=1))
    idx=np.argsort(-(err*same))[:8]
    print(name)
    for i in idx: print(" err %.3g rel %.3g pos %s obj %d node %d px (%d,%d)"%(err[i],rel[i],b["pos"][i],b["obj"][i],b["node"][i],b["x"][i],b["y"][i]))
    print(" node0 max err", (err*same*(b["node"]==0)).max(), " nodes>0 max err", (err*same*(b["node"]>0)).max())
