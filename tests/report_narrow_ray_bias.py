"""What keeping ray directions and hit normals in float (positions, distances and the 1e-4 guards in double) would do to
the results: the CPU oracle with its measuring switch oracle_set_narrow_rays against the oracle as it is, on the whole of a
benchmark configuration.  CPU only; prints a table.

    python tests/report_narrow_ray_bias.py [config] [threads]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytrace3_amd import scenes            # noqa: E402
from tests import oracle_lib as ol          # noqa: E402

SEED = 20260101


def run(scene, cam, rounds, per_round, threads, narrow):
    o = ol.ParallelOracle(threads, newton_fp32=False)
    o.set_narrow_rays(narrow)
    o.set_scene(scene)
    o.eye_pass(cam, SEED)
    o.photon_pass(rounds, per_round, SEED)
    out = o.hitpoints(), o.counters(), o.resolve()
    o.close()
    return out


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2_teapot_800x600"
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(16, os.cpu_count() or 1)
    factory, (w, h), rounds, per_round = scenes.CONFIGS[cfg]
    scene, cam = factory(), scenes.default_camera(w, h)
    (h0, c0, p0), (h1, c1, p1) = run(scene, cam, rounds, per_round, threads, False), run(scene, cam, rounds, per_round, threads, True)
    key = lambda a: (a["x"].astype(np.int64) << 40) | (a["y"].astype(np.int64) << 20) | a["node"].astype(np.int64)
    common, i0, i1 = np.intersect1d(key(h0), key(h1), return_indices=True)
    a, b = h0[i0], h1[i1]
    same = a["obj"] == b["obj"]
    err = np.abs(a["pos"] - b["pos"]).max(axis=1)[same]
    direct = (a["node"] == 0)[same]
    print(f"{cfg}: {len(h0)} / {len(h1)} HitPoints, {len(common)} common, same object on {same.mean():.6f}")
    print(f"  position shift, seen directly: median {np.median(err[direct]):.2e}, max {err[direct].max():.2e}; "
          f"behind a mirror / glass: median {np.median(err[~direct]):.2e}, 99.9 % {np.quantile(err[~direct], 0.999):.2e}, max {err[~direct].max():.2e}, "
          f"beyond 1e-3: {(err > 1e-3).sum()}")
    for k in ("segments", "deposits", "candidates", "accepted"):
        print(f"  {k}: {c0[k]} -> {c1[k]} ({(c1[k] - c0[k]) / max(1, c0[k]):+.2e})")
    d = np.abs(p1 - p0).max(axis=2) / p0.mean()
    print(f"  image: within 1e-3 of the mean on {(d <= 1e-3).mean():.5f} of the pixels ({int((d > 1e-3).sum())} outliers), MAE / mean {d.mean():.2e}, max {d.max():.3g}")
    dn = (a["newN"] != b["newN"])[same]
    print(f"  per-HitPoint photon counts differ on {dn.mean():.4f} of the HitPoints")


if __name__ == "__main__":
    main()
