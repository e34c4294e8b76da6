"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the CPU
oracle).  CPU: the oracle still reproduces them.  GPU: the HIP path matches them without the
oracle in the loop."""
import os

import numpy as np
import pytest

from raytrace3_amd import scenes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def _key(x, y, node):
    return (x.astype(np.int64) << 40) | (y.astype(np.int64) << 20) | node.astype(np.int64)


# ------------------------------------------------------------------ CPU: oracle regression pins
def test_oracle_reproduces_hitpoints_and_image():
    from tests import oracle_lib as ol
    g = _load("c1b_40x32_hitpoints.npz")
    o = ol.OracleContext()
    o.set_scene(scenes.scene_spheres_plane(True))
    o.eye_pass(scenes.default_camera(int(g["w"]), int(g["h"])), int(g["seed"]))
    o.photon_pass(int(g["rounds"]), int(g["per_round"]), int(g["seed"]))
    hp = o.hitpoints()
    assert np.array_equal(_key(hp["x"], hp["y"], hp["node"]), _key(g["x"], g["y"], g["node"]))
    assert np.allclose(hp["pos"], g["pos"], rtol=0, atol=1e-12)
    assert np.array_equal(hp["newN"], g["newN"])
    assert np.allclose(o.resolve(), g["pic"], rtol=1e-12, atol=0)
    assert list(o.counters().values()) == list(g["counters"])


def test_oracle_reproduces_newton_and_deposits():
    from tests import oracle_lib as ol
    g = _load("teapot_newton_256.npz")
    scene = scenes.scene_teapot(1, textured=False)
    o = ol.OracleContext()
    o.set_scene(scene)
    assert np.allclose(o.newton(g["org"], g["dir"], g["patch"], int(g["seed"])), g["tuv_fp64"], rtol=0, atol=1e-12)
    d = _load("teapot_deposits_200photons.npz")
    o2 = ol.OracleContext()
    o2.set_scene(scenes.scene_teapot(1, textured=True))
    dep = o2.deposits(int(d["rounds"]), int(d["per_round"]), int(d["seed"]), photon_begin=0, photon_count=200)
    assert dep.shape == d["deposits"].shape
    assert np.allclose(dep, d["deposits"], rtol=0, atol=1e-12)


# ------------------------------------------------------------------ GPU vs golden
@pytest.fixture(scope="module")
def gpu():
    from raytrace3_amd import SppmContext
    ctx = SppmContext(0)
    yield ctx
    ctx.close()


@pytest.mark.gpu
def test_gpu_matches_golden_hitpoints_and_image(gpu):
    g = _load("c1b_40x32_hitpoints.npz")
    gpu.set_scene(scenes.scene_spheres_plane(True))
    gpu.eye_pass(scenes.default_camera(int(g["w"]), int(g["h"])), int(g["seed"]))
    gpu.photon_pass(int(g["rounds"]), int(g["per_round"]), int(g["seed"]))
    hp = gpu.hitpoints()
    kg, kc = _key(hp["x"], hp["y"], hp["node"]), _key(g["x"], g["y"], g["node"])
    og, oc = np.argsort(kg), np.argsort(kc)
    assert np.array_equal(kg[og], kc[oc])
    assert np.array_equal(hp["obj"][og], g["obj"][oc])
    assert np.allclose(hp["pos"][og], g["pos"][oc], rtol=0, atol=1e-9)       # analytic scene: fp64 on both sides
    assert np.allclose(hp["wgt"][og], g["wgt"][oc], rtol=2e-6, atol=0)       # fp32 colours
    assert (hp["newN"][og] != g["newN"][oc]).mean() < 0.005                  # counts are integers: exact but for radius-edge flips
    pic = gpu.resolve()
    diff = np.abs(pic - g["pic"]).max(axis=2) / g["pic"].mean()
    assert (diff <= 1e-3).mean() >= 0.995 and diff.mean() <= 1e-4


@pytest.mark.gpu
def test_gpu_matches_golden_newton(gpu):
    g = _load("teapot_newton_256.npz")
    gpu.set_scene(scenes.scene_teapot(1, textured=False))
    r = gpu.newton(g["org"], g["dir"], g["patch"], int(g["seed"]))
    ref = g["tuv_fp64"]
    found_r, found_c = r[:, 0] < 1e19, ref[:, 0] < 1e19
    assert (found_r != found_c).sum() <= 1                                   # measured: none
    both = found_r & found_c
    err = np.abs(r[both] - ref[both]).max(axis=1)
    print(f"golden newton: {int((found_r != found_c).sum())} decision flips, median {np.median(err):.2e}, within 1e-3 {(err < 1e-3).mean():.4f}, max {err.max():.2e}")
    # measured: median 4.0e-6, 5 of 240 beyond 1e-3 (another iterate of the same restart passed the residual test first), worst
    # 0.0124; the residual rule of Bezier.h:135 lets candidates lie 0.01 from the surface, so nothing can be 0.02 apart
    assert np.median(err) < 1e-5 and (err < 1e-3).mean() >= 0.958 and err.max() < 0.02


@pytest.mark.gpu
def test_gpu_matches_golden_photon_deposits(gpu):
    """Every photon path, deposit by deposit: position, normal and flux of each diffuse hit."""
    d = _load("teapot_deposits_200photons.npz")
    gold = d["deposits"]
    gpu.set_scene(scenes.scene_teapot(1, textured=True))
    gpu.eye_pass(scenes.default_camera(8, 8), int(d["seed"]))               # the hook needs a grid; its content is irrelevant
    dep = gpu.deposits(int(d["rounds"]), int(d["per_round"]), int(d["seed"]), photon_begin=0, photon_count=200)
    order = lambda a: a[np.lexsort((a[:, 1], a[:, 0]))]
    dep, gold = order(dep), order(gold)
    kd = dep[:, 0].astype(np.int64) * 16 + dep[:, 1].astype(np.int64)
    kg = gold[:, 0].astype(np.int64) * 16 + gold[:, 1].astype(np.int64)
    common, i, j = np.intersect1d(kd, kg, return_indices=True)
    # a path that touches the teapot can differ at the Newton residual scale and then diverge
    assert len(common) >= 0.99 * len(kg), (len(common), len(kg))              # measured: all 715
    perr = np.abs(dep[i, 2:5] - gold[j, 2:5]).max(axis=1)
    print(f"golden deposits: {len(common)} of {len(kg)} common ({len(common) / len(kg):.4f}), within 1e-6 {(perr < 1e-6).mean():.4f}")
    assert (perr < 1e-6).mean() >= 0.986, (perr < 1e-6).mean()                # measured: 710 of 715
    ok = perr < 1e-6
    assert np.allclose(dep[i][ok, 5:8], gold[j][ok, 5:8], rtol=0, atol=1e-5)     # normals (stored fp32)
    assert np.allclose(dep[i][ok, 8:11], gold[j][ok, 8:11], rtol=2e-5, atol=0)   # flux (fp32 products)
