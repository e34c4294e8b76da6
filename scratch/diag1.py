import sys; sys.path.insert(0,'.')
import numpy as np
from raytrace3_amd import SppmContext, scenes
g = np.load('tests/golden/teapot_newton_256.npz')
gpu = SppmContext(0); gpu.set_scene(scenes.scene_teapot(1, textured=False))
r = gpu.newton(g['org'], g['dir'], g['patch'], int(g['seed']))
for name in ('tuv_fp64','tuv_fp32'):
    ref = g[name]
    fr, fc = r[:,0]<1e19, ref[:,0]<1e19
    print(name, 'found gpu', fr.sum(), 'found ref', fc.sum(), 'mismatch', (fr!=fc).sum(), 'gpu only', (fr&~fc).sum(), 'ref only', (~fr&fc).sum())
    both = fr&fc
    err = np.abs(r[both]-ref[both]).max(axis=1)
    print('  err median', np.median(err), 'frac<1e-3', (err<1e-3).mean(), 'max', err.max())
    idx = np.where(fr!=fc)[0][:6]
    for i in idx: print('  ', i, r[i], ref[i])
f64, f32 = g['tuv_fp64'], g['tuv_fp32']
a, b = f64[:,0]<1e19, f32[:,0]<1e19
print('oracle fp64 vs fp32 mismatch', (a!=b).sum())
