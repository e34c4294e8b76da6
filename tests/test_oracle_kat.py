"""Known-answer tests that pin the CPU oracle (oracle/sppm_oracle.cpp).

The reference (wondergo2017/raytrace3) ships no tests or golden vectors, so the
oracle is pinned by closed-form answers derived from the reference source; each
test names the reference lines whose behaviour it asserts.
"""
import ctypes as C
import math

import numpy as np
import pytest

from raytrace3_amd import abi, scenes
from tests import oracle_lib as ol
from tests.oracle_lib import dp, vec


@pytest.fixture(scope="module")
def lib():
    return ol.load()


def _teapot_patches():
    return scenes.load_bpt(scenes.TEAPOT_BPT, (20, 0, 120), 4, scenes.teapot_transform())


def test_mwc_matches_cv_rng_recurrence(lib):
    # cv::RNG behind Vec3.h:5: state = (u32)state * 4164903690 + (state >> 32), seed 0xffffffff
    n = 8
    u32 = (C.c_uint32 * n)()
    uni = np.zeros(n)
    lib.oracle_mwc_kat(n, u32, dp(uni))
    state = 0xffffffff
    exp = []
    for _ in range(n):
        state = ((state & 0xffffffff) * 4164903690 + (state >> 32)) & 0xffffffffffffffff
        exp.append(state & 0xffffffff)
    assert list(u32) == exp
    assert list(u32)[:4] == [0x07c09cf6, 0xb302a6a5, 0xe6abd4e7, 0x507b8a5d]
    # double = ((u64)next << 32 | next) * 2^-64, consuming two outputs per draw
    assert uni[0] == pytest.approx(((exp[0] << 32) | exp[1]) * 2.0 ** -64, rel=1e-15)
    assert uni[0] == pytest.approx(0.03028279326373227, rel=1e-12)
    assert uni[1] == pytest.approx(0.901059442971457, rel=1e-12)


def _murmur(seed, stream, path, sub, idx):
    M = 0xffffffff

    def rotl(x, r):
        return ((x << r) | (x >> (32 - r))) & M

    def mm(h, k):
        k = (k * 0xcc9e2d51) & M
        k = rotl(k, 15)
        k = (k * 0x1b873593) & M
        h ^= k
        h = rotl(h, 13)
        return (h * 5 + 0xe6546b64) & M

    h = 0x9747b28c
    for k in (seed & M, seed >> 32, stream, path, sub, idx):
        h = mm(h, k)
    h ^= 24
    h ^= h >> 16
    h = (h * 0x85ebca6b) & M
    h ^= h >> 13
    h = (h * 0xc2b2ae35) & M
    h ^= h >> 16
    return h


def test_counter_rng_is_murmur3_of_its_key(lib):
    rng = np.random.default_rng(1)
    for _ in range(50):
        seed = int(rng.integers(0, 2 ** 63))
        k = [int(x) for x in rng.integers(0, 2 ** 32, size=4)]
        assert lib.oracle_rng_u32(seed, *k) == _murmur(seed, *k)
    # uniformity smoke check of the 24-bit uniforms
    u = np.array([lib.oracle_rng_u32(7, 1, p, 3, 0) >> 8 for p in range(20000)]) / 2.0 ** 24
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005


def test_patch_corners_and_derivatives(lib):
    # Bezier.h:69-111: S(0,0)=P[0], S(1,0)=P[3], S(0,1)=P[12], S(1,1)=P[15] (u runs along the fast index)
    P = _teapot_patches()
    out = np.zeros(9)
    for p in (0, 5, 17, 31):
        flat = np.ascontiguousarray(P[p].reshape(-1))
        for (u, v, k) in ((0, 0, 0), (1, 0, 3), (0, 1, 12), (1, 1, 15)):
            lib.oracle_bezier_eval(dp(flat), float(u), float(v), dp(out))
            assert np.allclose(out[:3], P[p, k], atol=1e-12)
        # derivatives against central finite differences
        u, v, h = 0.37, 0.61, 1e-6
        lib.oracle_bezier_eval(dp(flat), u, v, dp(out))
        s_u, s_v = out[3:6].copy(), out[6:9].copy()
        a, b = np.zeros(9), np.zeros(9)
        lib.oracle_bezier_eval(dp(flat), u + h, v, dp(a))
        lib.oracle_bezier_eval(dp(flat), u - h, v, dp(b))
        assert np.allclose(s_u, (a[:3] - b[:3]) / (2 * h), atol=1e-5)
        lib.oracle_bezier_eval(dp(flat), u, v + h, dp(a))
        lib.oracle_bezier_eval(dp(flat), u, v - h, dp(b))
        assert np.allclose(s_v, (a[:3] - b[:3]) / (2 * h), atol=1e-5)


def test_teapot_world_bbox():
    # world transform Trans2*Trans*(4p)+(20,0,120) ~ (20+4p_y, 4p_z, 120-4p_x)
    P = _teapot_patches().reshape(-1, 3)
    lo, hi = P.min(0), P.max(0)
    assert np.allclose(lo, [12, 0, 105.9], atol=0.2) and np.allclose(hi, [28, 12.6, 132], atol=0.2)
    assert P.shape[0] == 32 * 16


def test_aabb_entry_faces_only(lib):
    # Bezier.h:20-51: only the three entry faces with t > 0 are tested; an origin inside the box misses
    pmin, pmax = vec(0, 0, 0), vec(1, 1, 1)
    hit = lambda o, d: lib.oracle_aabb_intersect(dp(pmin), dp(pmax), dp(vec(*o)), dp(vec(*d)))
    assert hit((-1, .5, .5), (1, 0, 0)) == 1
    assert hit((.5, .5, .5), (1, 0, 0)) == 0          # origin inside
    assert hit((.5, .5, .5), (.3, .4, .5)) == 0
    assert hit((2, .5, .5), (1, 0, 0)) == 0           # pointing away
    assert hit((-1, .5, .5), (1, 5e-5, 0)) == 1       # |d_y| < 1e-4 is skipped, x face decides
    assert hit((-1, 2, .5), (1, 0, 0)) == 0
    assert hit((-1, -1, .5), (1, 1, 0)) == 1          # enters through the corner edge (inclusive bounds)


def test_plane_and_sphere_thresholds(lib):
    # Obj.h:65-85: miss if |d.n| < 1e-4 or dist <= 1e-4; two-sided
    pl = abi.Object()
    pl.p0[:] = (0, 0, 0)
    pl.n[:] = (0, 1, 0)
    dist = C.c_double(0)
    f = lambda o, d: lib.oracle_plane_intersect(C.byref(pl), dp(vec(*o)), dp(vec(*d)), C.byref(dist))
    assert f((0, 1, 0), (0, -1, 0)) == 1 and dist.value == pytest.approx(1.0)
    assert f((0, -1, 0), (0, 1, 0)) == 1 and dist.value == pytest.approx(1.0)   # from behind: still a hit
    assert f((0, 1, 0), (0, 1, 0)) == 0
    assert f((0, 1, 0), (1, -5e-5, 0)) == 0                                      # |project| < 1e-4
    assert f((0, 5e-5, 0), (0, -1, 0)) == 0                                      # dist <= 1e-4
    # Obj.h:111-139
    sp = abi.Object()
    sp.p0[:] = (0, 0, 0)
    sp.radius = 2.0
    inside = C.c_int(0)
    g = lambda o, d: lib.oracle_sphere_intersect(C.byref(sp), dp(vec(*o)), dp(vec(*d)), C.byref(dist), C.byref(inside))
    assert g((0, 0, -5), (0, 0, 1)) == 1 and dist.value == pytest.approx(3.0) and inside.value == 0
    assert g((0, 0, 0), (0, 0, 1)) == 1 and dist.value == pytest.approx(2.0) and inside.value == 1
    assert g((0, 0, 5), (0, 0, 1)) == 0                                          # behind
    assert g((0, 2.0, -5), (0, 0, 1)) == 0                                       # tangent: det2 = 0 < 1e-4
    assert g((0, 0, -2), (0, 0, 1)) == 1 and dist.value == pytest.approx(4.0) and inside.value == 1  # on the surface


def test_refraction_reflection_rotation(lib):
    out = np.zeros(3)
    # Vec3.h:124-134: Snell with the TIR fallback to GetRefl when cosR2 <= 1e-4
    I = vec(math.sin(0.5), -math.cos(0.5), 0)
    N = vec(0, 1, 0)
    lib.oracle_refracted(dp(I), dp(N), 1 / 1.5, dp(out))
    sin_t = math.sin(0.5) / 1.5
    assert np.allclose(out, [sin_t, -math.sqrt(1 - sin_t ** 2), 0], atol=1e-12)
    I = vec(math.sin(1.2), -math.cos(1.2), 0)      # inside glass, beyond the critical angle (41.8 deg)
    lib.oracle_refracted(dp(I), dp(N), 1.5, dp(out))
    assert np.allclose(out, [math.sin(1.2), math.cos(1.2), 0], atol=1e-12)
    lib.oracle_reflected_unit(dp(vec(1, -1, 0)), dp(N), dp(out))
    assert np.allclose(out, np.array([1, 1, 0]) / math.sqrt(2), atol=1e-12)
    # Vec3.h:99-115: identity below 1e-4, Rodrigues otherwise
    v = vec(1, 2, 3)
    lib.oracle_rotated(dp(v), dp(vec(0, 0, 1)), 5e-5, dp(out))
    assert np.array_equal(out, v)
    lib.oracle_rotated(dp(vec(1, 0, 0)), dp(vec(0, 0, 1)), math.pi / 2, dp(out))
    assert np.allclose(out, [0, 1, 0], atol=1e-12)


def test_cosine_hemisphere_sample(lib):
    # Vec3.h:90-98: theta = acos(sqrt(u1)), phi = 2*PI*u2 around N
    out = np.zeros(3)
    rng = np.random.default_rng(3)
    for _ in range(200):
        N = rng.normal(size=3)
        N /= np.linalg.norm(N)
        u1, u2 = rng.random(), rng.random()
        lib.oracle_rand_refl(dp(N), u1, u2, dp(out))
        assert np.linalg.norm(out) == pytest.approx(1.0, abs=1e-9)
        assert float(out @ N) == pytest.approx(math.sqrt(u1), abs=1e-6)
    lib.oracle_rand_refl(dp(vec(0, 0, 1)), 1.0, 0.3, dp(out))     # theta = 0 -> rotated() returns N itself
    assert np.allclose(out, [0, 0, 1])


def test_tone_map(lib):
    # Raytracer.h:24-26
    assert lib.oracle_to_int(0.0) == 0
    assert lib.oracle_to_int(1.0) == 207
    assert lib.oracle_to_int(1e9) == 255
    assert list(scenes.to_int([0.0, 1.0, 1e9])) == [0, 207, 255]


def test_written_image_layout(lib):
    # Raytracer.h:460-466: row h - 1 - i1 of the byte image holds toInt(pic[i1][j][k] / n), channel k at byte k
    rng = np.random.default_rng(4)
    pic = rng.gamma(0.5, 0.6, size=(5, 7, 3))
    got = ol.tonemap(pic, 8.0, lib=lib)
    for i1 in range(5):
        for j in range(7):
            for k in range(3):
                assert got[4 - i1, j, k] == lib.oracle_to_int(pic[i1, j, k] / 8.0)
    assert np.array_equal(got, scenes.tonemap_bgr(pic, 8.0))
    assert np.array_equal(ol.tonemap(pic, 8.0, rgb=True, lib=lib), got[..., ::-1])   # the same picture as a PPM


def test_hitpoint_update_is_dead_with_n_zero(lib):
    # Raytracer.h:64-79: N starts at 0 and is only changed behind `if (N <= 0 || newN <= 0) return;`
    R2, N, newN = C.c_double(2.0), C.c_int32(0), C.c_int32(57)
    tao = vec(1, 2, 3)
    lib.oracle_hitpoint_update(C.byref(R2), C.byref(N), dp(tao), C.byref(newN))
    assert (R2.value, N.value, newN.value) == (2.0, 0, 57) and np.array_equal(tao, [1, 2, 3])
    # the arithmetic when it does run: k = (N + a*newN)/(N + newN), N += a*newN truncated to int
    R2, N, newN = C.c_double(2.0), C.c_int32(10), C.c_int32(5)
    lib.oracle_hitpoint_update(C.byref(R2), C.byref(N), dp(tao), C.byref(newN))
    k = (10 + 0.7 * 5) / 15
    assert R2.value == pytest.approx(2.0 * k) and N.value == 13 and newN.value == 0
    assert np.allclose(tao, np.array([1, 2, 3]) * k)


def test_camera_basis_of_main_cpp(lib):
    # main.cpp:24-27 + Camera.h:32-45
    cam = abi.Camera()
    pos = vec(50, 35, 230)
    look = pos + vec(0, 0.042612, -1)
    lib.oracle_camera_lookat(dp(pos), dp(look), 1024, 1024, C.byref(cam))
    assert np.allclose(list(cam.du), [1, 0, 0], atol=1e-12)
    assert np.allclose(list(cam.dv), [0, 0.99909334, 0.04257337], atol=1e-7)
    assert np.linalg.norm(list(cam.dir)) == pytest.approx(1.07225348, abs=1e-7)
    py = scenes.default_camera(1024, 1024)
    for f in ("pos", "dir", "du", "dv"):
        assert np.allclose(list(getattr(py, f)), list(getattr(cam, f)), atol=1e-15)
    # Camera.h:18-22: the centre of the image looks along dir
    o, d = np.zeros(3), np.zeros(3)
    cam.w = cam.h = 2
    lib.oracle_camera_emit(C.byref(cam), 0, 0, dp(o), dp(d))
    exp = -0.25 * np.array(list(cam.du)) - 0.25 * np.array(list(cam.dv)) + np.array(list(cam.dir))
    assert np.allclose(d, exp / np.linalg.norm(exp), atol=1e-12) and np.allclose(o, pos)


def test_texture_bilinear_wrap(lib):
    # Element.h:61-72
    img = np.zeros((2, 2, 3), dtype=np.uint8)
    img[0, 0] = (255, 0, 0)
    img[0, 1] = (0, 255, 0)
    img[1, 0] = (0, 0, 255)
    img[1, 1] = (255, 255, 255)
    t = abi.Texture()
    t.rows, t.cols = 2, 2
    t.texels = img.ctypes.data_as(C.POINTER(C.c_uint8))
    out = np.zeros(3)
    lib.oracle_texture_color_uv(C.byref(t), 0.0, 0.0, dp(out))
    assert np.allclose(out, [1, 0, 0])
    lib.oracle_texture_color_uv(C.byref(t), 0.25, 0.25, dp(out))   # row = col = 0.5: equal mix of all four
    assert np.allclose(out, (np.array([1, 0, 0]) + [0, 1, 0] + [0, 0, 1] + [1, 1, 1]) / 4)
    lib.oracle_texture_color_uv(C.byref(t), 0.75, 0.0, dp(out))    # row 1.5 wraps r2 to row 0
    assert np.allclose(out, 0.5 * np.array([0, 0, 1]) + 0.5 * np.array([1, 0, 0]))
    lib.oracle_texture_color_uv(C.byref(t), -0.25, 1.0, dp(out))   # u - floor(u) = 0.75, v wraps to 0
    assert np.allclose(out, 0.5 * np.array([0, 0, 1]) + 0.5 * np.array([1, 0, 0]))


def test_newton_finds_the_patch_root(lib):
    # Bezier.h:112-159: shoot a ray at a known surface point; the returned (t,u,v) must satisfy
    # |org + t*dir - S(u,v)|^2 < 1e-4 and t must be the smallest converged root
    s = scenes.scene_teapot(1, textured=False)
    o = ol.OracleContext(lib)
    o.set_scene(s)
    P = s.patches
    out9 = np.zeros(9)
    rng = np.random.default_rng(5)
    n_found = 0
    for p in (0, 3, 9, 20):
        flat = np.ascontiguousarray(P[p].reshape(-1))
        u, v = rng.random(2) * 0.8 + 0.1
        lib.oracle_bezier_eval(dp(flat), u, v, dp(out9))
        target = out9[:3].copy()
        org = np.array([50., 35., 230.])
        d = target - org
        tt = np.linalg.norm(d)
        d /= tt
        tuv = o.newton(org[None], d[None], [p], seed=11)[0]
        if tuv[0] >= 1e19:
            continue
        n_found += 1
        lib.oracle_bezier_eval(dp(flat), tuv[1], tuv[2], dp(out9))
        assert np.sum((org + d * tuv[0] - out9[:3]) ** 2) < 1e-4
        assert tuv[0] <= tt + 0.02
    assert n_found >= 3


def test_gather_modes_agree_and_counters(lib):
    # The predicate, not the search structure, defines the result (Raytracer.h:146-158)
    s = scenes.scene_spheres_plane(True)
    cam = scenes.default_camera(24, 24)
    res = []
    for mode in (ol.GATHER_GRID, ol.GATHER_BRUTE, ol.GATHER_KDTREE):
        o = ol.OracleContext(lib)
        o.set_scene(s)
        o.eye_pass(cam, 3)
        o.photon_pass(30, 100, 3, mode=mode)
        res.append((o.resolve(), o.counters(), o.hitpoints()))
    for r in res[1:]:
        assert np.array_equal(r[0], res[0][0])
        assert r[1] == res[0][1]
    hp = res[0][2]
    assert (hp["N"] == 0).all() and (hp["R2"] == 2.0).all()        # update() never runs
    assert hp["newN"].sum() == res[0][1]["accepted"]
    assert res[0][1]["photons"] == 3000


def test_photon_range_sharding_is_additive(lib):
    # photon g is a pure function of (seed, replica, g): two half ranges sum to the whole pass
    s = scenes.scene_spheres_plane(True)
    cam = scenes.default_camera(16, 16)
    full = ol.OracleContext(lib)
    full.set_scene(s)
    full.eye_pass(cam, 9)
    full.photon_pass(20, 100, 9)
    a = ol.OracleContext(lib)
    a.set_scene(s)
    a.eye_pass(cam, 9)
    a.photon_pass(20, 100, 9, photon_begin=0, photon_count=700)
    b = ol.OracleContext(lib)
    b.set_scene(s)
    b.eye_pass(cam, 9)
    b.photon_pass(20, 100, 9, photon_begin=700, photon_count=1300)
    assert np.allclose(a.resolve() + b.resolve(), full.resolve(), rtol=1e-12, atol=1e-15)
    assert a.counters()["accepted"] + b.counters()["accepted"] == full.counters()["accepted"]


def test_literal_and_parity_rng_agree_statistically(lib):
    # same estimator, different random streams: image means agree within Monte-Carlo noise
    s = scenes.scene_spheres_plane(True)
    cam = scenes.default_camera(16, 16)
    means = []
    for literal in (False, True):
        o = ol.OracleContext(lib, rng_literal=literal)
        o.set_scene(s)
        o.eye_pass(cam, 5)
        o.photon_pass(200, 100, 5)
        means.append(o.resolve().mean())
    assert means[0] == pytest.approx(means[1], rel=0.1)


def test_eye_tree_node_ids_and_weights(lib):
    # Raytracer.h:312-336: diffuse -> HitPoint with wgt = color*wgt*diff; mirror / glass recurse
    s = scenes.scene_spheres_plane(True)
    cam = scenes.default_camera(32, 32)
    o = ol.OracleContext(lib)
    o.set_scene(s)
    o.eye_pass(cam, 1)
    hp = o.hitpoints()
    assert len(hp) > 0 and hp["node"].max() > 0                      # some paths went through mirror / glass
    direct = hp[hp["node"] == 0]
    floor = direct[direct["obj"] == 3]
    assert np.allclose(floor["wgt"], 0.75 * 0.75)                    # colour .75 * diff .75
    assert len(np.unique(np.stack([hp["x"], hp["y"], hp["node"]], 1), axis=0)) == len(hp)
    assert (hp["node"] < 2 ** 15).all()


def test_oracle_over_host_threads_equals_the_serial_oracle():
    """ol.ParallelOracle (strips of rows, ranges of photons, summed) against one serial context: every HitPoint field
    and tally equal, tao and the image to the order of their fp64 sums."""
    from raytrace3_amd import scenes
    scene, cam = scenes.scene_teapot(1), scenes.default_camera(96, 72)
    for count in (0, 3333):
        o = ol.OracleContext(newton_fp32=True)
        o.set_scene(scene)
        o.eye_pass(cam, 11)
        o.photon_pass(100, 100, 11, photon_begin=17, photon_count=count)
        p = ol.ParallelOracle(5, newton_fp32=True)
        p.set_scene(scene)
        p.eye_pass(cam, 11)
        p.photon_pass(100, 100, 11, photon_begin=17, photon_count=count)
        a, b = o.hitpoints(), p.hitpoints()
        assert len(a) == len(b) > 0
        for f in a.dtype.names:
            if f == "tao":
                assert np.allclose(a[f], b[f], rtol=1e-12, atol=0)
            else:
                assert np.array_equal(a[f], b[f]), f
        assert o.counters() == p.counters()
        assert o.counters()["photons"] == (10000 - 17 if count == 0 else count)
        assert np.allclose(o.resolve(), p.resolve(), rtol=1e-12, atol=0)
        o.close()
        p.close()
