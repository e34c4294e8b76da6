"""Host-side scene description: the reference's Scene / Camera objects flattened
for the C-ABI (include/sppm_abi.h).

Mirrors, in Python, what the drop-in C++ headers (include/raytrace3/) do for a
C++ host: build the object list of Scene::init (raytracer/Scene.h:98-164), load a
.bpt Bezier file the way Bezier3Obj's constructor does (raytracer/Bezier.h:199-239),
evaluate Camera::lookAt (raytracer/Camera.h:32-54) and hand flat arrays across.
JPEG decoding is not on the hot path: textures are procedural (seeded) images.
"""
import ctypes as C
import math
import os

import numpy as np

from . import abi

PI = 3.1415926  # raytracer/Vec3.h:8 (truncated on purpose)
DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
TEAPOT_BPT = os.path.join(DATA_DIR, "teapot.bpt")


def material(diff=(0, 0, 0), refl=(0, 0, 0), refr=(0, 0, 0), refrn=1.5, refln=1.0):
    return dict(diff=tuple(map(float, diff)), refl=tuple(map(float, refl)), refr=tuple(map(float, refr)),
                refrn=float(refrn), refln=float(refln))


# raytracer/Scene.h:100-113
WHITE_DIFF = material((.75, .75, .75), refrn=0)
MIRROR = material(refl=(.999, .999, .999), refrn=0)
MIRROR2 = material((.25, .25, .25), (.999, .999, .999), refrn=0)
REFR0 = material(refr=(.999, .999, .999), refrn=1.5)
REFR4 = material((.2, .2, .2), refr=(0, .5, .5), refrn=1.5)


def _elevate(q, axis):
    """Degree elevation of Bezier control points along one axis: the same curve with one more control point."""
    q = np.moveaxis(q, axis, 0)
    d = q.shape[0] - 1
    r = np.empty((d + 2,) + q.shape[1:], dtype=np.float64)
    r[0], r[d + 1] = q[0], q[d]
    for i in range(1, d + 1):
        a = i / (d + 1)
        r[i] = q[i - 1] * a + q[i] * (1 - a)
    return np.moveaxis(r, 0, axis)


def load_bpt(path, C_=(0, 0, 0), scale=1.0, trans=None):
    """Bezier3Obj constructor, raytracer/Bezier.h:199-235: per patch `m n` then
    (m+1)(n+1) points; point = Trans * (scale * p) + C.  Returns [n_patches, 16, 3].
    The reference copies the points into Bezier3::P[16] as they come, which is a bicubic patch only for
    m = n = 3; a patch of lower degree in u or v is raised to degree 3 here (degree elevation: the same
    surface, exactly), a higher degree has no bicubic form and is refused."""
    trans = np.eye(3) if trans is None else np.asarray(trans, dtype=np.float64)
    with open(path) as f:
        tok = f.read().split()
    it = iter(tok)
    size = int(next(it))
    patches = np.zeros((size, 16, 3), dtype=np.float64)
    for i in range(size):
        m, n = int(next(it)), int(next(it))
        if not (0 <= m <= 3 and 0 <= n <= 3):
            raise ValueError(f"{path}: patch {i} has degree {m} x {n}; Bezier3::P[16] holds at most a bicubic patch")
        g = np.zeros((m + 1, n + 1, 3))
        for j in range(m + 1):
            for k in range(n + 1):
                p = np.array([float(next(it)) * scale for _ in range(3)])
                g[j, k] = trans @ p
        while g.shape[1] < 4:
            g = _elevate(g, 1)
        while g.shape[0] < 4:
            g = _elevate(g, 0)
        patches[i] = g.reshape(16, 3) + np.asarray(C_, dtype=np.float64)
    return patches


def teapot_transform():
    """raytracer/Scene.h:142-152: Trans2 * Trans with theta = (90/180)*PI (PI truncated)."""
    trans = np.zeros((3, 3))
    trans[0, 0] = 1
    trans[1, 2] = 1
    trans[2, 1] = 1
    trans2 = np.zeros((3, 3))
    theta = (90. / 180.) * PI
    trans2[0, 0] = math.cos(theta)
    trans2[0, 2] = math.sin(theta)
    trans2[1, 1] = 1
    trans2[2, 0] = -math.sin(theta)
    trans2[2, 2] = math.cos(theta)
    return trans2 @ trans


def procedural_texture(name, rows=256, cols=256, seed=20260101):
    """Seeded stand-in for the reference's JPEG textures (rows x cols x 3, BGR bytes)."""
    h = seed
    for ch in name.encode():
        h = (h * 1000003 + ch) & 0xffffffff
    rng = np.random.default_rng(h)
    r = np.arange(rows)[:, None]
    c = np.arange(cols)[None, :]
    base = rng.integers(60, 200, size=3)
    checker = (((r // 32) + (c // 32)) & 1).astype(np.float64)
    wave = 0.5 + 0.5 * np.sin(2 * math.pi * (r * 3 + c * 5) / 256.0 + rng.uniform(0, 6.28))
    img = np.zeros((rows, cols, 3), dtype=np.float64)
    for k in range(3):
        img[:, :, k] = base[k] + 40 * checker + 30 * wave + rng.integers(0, 8, size=(rows, cols))
    return np.clip(img, 0, 255).astype(np.uint8)


def load_ppm(path):
    """A binary PPM (P6, maxval 255) as the reference's texture map: rows x cols x 3 bytes in the BGR order of
    cv::imread (raytracer/Element.h:47-59)."""
    with open(path, "rb") as f:
        raw = f.read()
    tokens, pos = [], 0
    while len(tokens) < 4:                      # magic, width, height, maxval; '#' starts a comment
        while raw[pos:pos + 1].isspace():
            pos += 1
        if raw[pos:pos + 1] == b"#":
            pos = raw.index(b"\n", pos) + 1
            continue
        end = pos
        while not raw[end:end + 1].isspace():
            end += 1
        tokens.append(raw[pos:end])
        pos = end
    if tokens[0] != b"P6" or int(tokens[3]) != 255:
        raise ValueError(f"{path}: not a binary PPM with maxval 255")
    w, h = int(tokens[1]), int(tokens[2])
    rgb = np.frombuffer(raw, dtype=np.uint8, count=w * h * 3, offset=pos + 1).reshape(h, w, 3)
    return np.ascontiguousarray(rgb[:, :, ::-1])


def save_ppm(path, bgr):
    """Writes a rows x cols x 3 BGR byte image as a binary PPM."""
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (bgr.shape[1], bgr.shape[0]))
        f.write(np.ascontiguousarray(bgr[:, :, ::-1]).tobytes())


class FlatScene:
    """Builder + owner of the arrays behind one sppm_scene."""

    def __init__(self):
        self.objects = []
        self.patch_list = []
        self.texture_list = []
        self.light_list = []
        self._c = None

    # -- Scene::objvec builders (same argument order as the reference constructors)
    def add_texture(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        assert img.ndim == 3 and img.shape[2] == 3
        self.texture_list.append(img)
        self._c = None
        return len(self.texture_list) - 1

    def _obj(self, type_, color, mat, texture=-1, p0=(0, 0, 0), n=(0, 0, 0), radius=0.0, first_patch=0, n_patches=0):
        self.objects.append(dict(type=type_, color=tuple(map(float, color)), material=mat, texture=texture,
                                 p0=tuple(map(float, p0)), n=tuple(map(float, n)), radius=float(radius),
                                 first_patch=first_patch, n_patches=n_patches))
        self._c = None
        return len(self.objects) - 1

    def add_plane(self, p0, n, color, mat, texture=-1):
        """PlaneObj(P0, n, color, material); the constructor normalises n (raytracer/Obj.h:60-62)."""
        n = np.asarray(n, dtype=np.float64)
        mo = math.sqrt(float(n @ n))
        if not abs(mo) < 1e-4:
            n = n * (1 / mo)
        return self._obj(abi.SPPM_OBJ_PLANE, color, mat, texture, p0=p0, n=n)

    def add_sphere(self, radius, center, color, mat, texture=-1):
        """SphereObj(radius, org, color, material), raytracer/Obj.h:110."""
        return self._obj(abi.SPPM_OBJ_SPHERE, color, mat, texture, p0=center, radius=radius)

    def add_bezier(self, C_, scale, path, color, mat, trans=None, texture=-1):
        """Bezier3Obj(C, scale, file, color, material, Trans), raytracer/Bezier.h:199."""
        p = load_bpt(path, C_, scale, trans)
        first = sum(x.shape[0] for x in self.patch_list)
        self.patch_list.append(p)
        return self._obj(abi.SPPM_OBJ_BEZIER, color, mat, texture, first_patch=first, n_patches=p.shape[0])

    def add_light(self, pos, color):
        """SpotLight(pos, color), raytracer/Light.h:19-22."""
        self.light_list.append((tuple(map(float, pos)), tuple(map(float, color))))
        self._c = None

    def set_texture(self, obj, tex):
        self.objects[obj]["texture"] = tex
        self._c = None

    @property
    def patches(self):
        if not self.patch_list:
            return np.zeros((0, 16, 3))
        return np.concatenate(self.patch_list, axis=0)

    @property
    def c_scene(self):
        if self._c is not None:
            return self._c
        objs = (abi.Object * max(1, len(self.objects)))()
        for o, d in zip(objs, self.objects):
            o.type, o.texture = d["type"], d["texture"]
            o.color[:] = d["color"]
            m = d["material"]
            o.material.diff[:] = m["diff"]
            o.material.refl[:] = m["refl"]
            o.material.refr[:] = m["refr"]
            o.material.refrn, o.material.refln = m["refrn"], m["refln"]
            o.p0[:] = d["p0"]
            o.n[:] = d["n"]
            o.radius = d["radius"]
            o.first_patch, o.n_patches = d["first_patch"], d["n_patches"]
        self._patches = np.ascontiguousarray(self.patches.reshape(-1), dtype=np.float64)
        texs = (abi.Texture * max(1, len(self.texture_list)))()
        for t, img in zip(texs, self.texture_list):
            t.rows, t.cols = img.shape[0], img.shape[1]
            t.texels = img.ctypes.data_as(C.POINTER(C.c_uint8))
        lights = (abi.Light * max(1, len(self.light_list)))()
        for l, (p, c) in zip(lights, self.light_list):
            l.pos[:] = p
            l.color[:] = c
        s = abi.Scene()
        s.n_objects, s.objects = len(self.objects), objs
        s.n_patches = self.patches.shape[0]
        s.patches = self._patches.ctypes.data_as(C.POINTER(C.c_double))
        s.n_textures, s.textures = len(self.texture_list), texs
        s.n_lights, s.lights = len(self.light_list), lights
        self._keep = (objs, texs, lights)
        self._c = s
        return s


def look_at(pos, look, w, h):
    """Camera::init + MoveTo + lookAt(look, 1.), raytracer/Camera.h:23-45."""
    pos = np.asarray(pos, dtype=np.float64)
    look = np.asarray(look, dtype=np.float64)

    def unit(v):
        mo = math.sqrt(float(v @ v))
        return v if abs(mo) < 1e-4 else v * (1.0 / mo)

    up = np.array([0., 0., 1.])
    d = unit(look - pos)
    du = unit(np.cross(d, up))
    dv = unit(-np.cross(d, du))
    fov = 50. * PI / 180.
    d = d * (.5 / math.tan(fov / 2.))
    cam = abi.Camera()
    cam.pos[:] = pos
    cam.dir[:] = d
    cam.du[:] = du
    cam.dv[:] = dv
    cam.w, cam.h = int(w), int(h)
    return cam


def default_camera(w, h, jitter=None):
    """raytracer/main.cpp:24-27; jitter = the 0.00015 * unit vector of Raytracer.h:434."""
    org = np.array([50., 35., 230.])
    look = org + np.array([0, 0.042612, -1.])
    pos = org if jitter is None else org + np.asarray(jitter)
    return look_at(pos, look, w, h)


def _spheres(s, planet_tex=-1):
    s.add_sphere(16.5, (27, 16.5, 47), (.999, .999, .999), MIRROR)          # Scene.h:123
    s.add_sphere(16.5, (73, 16.5, 88), (.999, .999, .999), REFR0)           # Scene.h:124
    s.add_sphere(8.5, (50, 8.5, 60), (.999, .999, .999), WHITE_DIFF, planet_tex)  # Scene.h:125


def scene_spheres_plane(box=False):
    """Config 1: floor plane + the three spheres of Scene.h:123-125 (+ the other four
    planes of Scene.h:117-122 when box=True), no Bezier object, no textures."""
    s = FlatScene()
    if box:
        s.add_plane((1, 40.8, 81.6), (1, 0, 0), (.75, .25, .25), WHITE_DIFF)
        s.add_plane((99, 40.8, 81.6), (1, 0, 0), (.25, .25, .75), WHITE_DIFF)
        s.add_plane((50, 40.8, 0), (0, 0, 1), (.75, .75, .75), MIRROR)
    s.add_plane((50, 0, 81.6), (0, -1, 0), (.75, .75, .75), WHITE_DIFF)
    if box:
        s.add_plane((50, 81.6, 81.6), (0, 1, 0), (.75, .75, .75), WHITE_DIFF)
    _spheres(s)
    s.add_light((50, 60, 85), (5000, 5000, 5000))
    return s


def scene_teapot(n_teapots=1, textured=True, textures=None):
    """Config 2 (n_teapots=1): the full scene of Scene.h:116-157 — 5 planes, 3 spheres,
    teapot.bpt scaled 4 at (20,0,120), textures on walls / floor / small sphere / teapot.
    Config 4 (n_teapots=4): three more teapots inside the tunnel (128 patches).
    textures: optional {"timg" | "wall" | "planet" | "blue": rows x cols x 3 BGR bytes} in place of the
    procedural stand-ins (what Texture("timg.jpg") etc. of Scene.h:131-156 load)."""
    s = FlatScene()
    tex = {}
    if textured:
        for name in ("timg", "wall", "planet", "blue"):
            tex[name] = s.add_texture(textures[name] if textures and name in textures else procedural_texture(name))
    t = lambda k: tex.get(k, -1)
    s.add_plane((1, 40.8, 81.6), (1, 0, 0), (.75, .25, .25), WHITE_DIFF, t("wall"))     # Left
    s.add_plane((99, 40.8, 81.6), (1, 0, 0), (.25, .25, .75), WHITE_DIFF, t("wall"))    # Right
    s.add_plane((50, 40.8, 0), (0, 0, 1), (.75, .75, .75), MIRROR)                      # Back
    s.add_plane((50, 0, 81.6), (0, -1, 0), (.75, .75, .75), WHITE_DIFF, t("timg"))      # Bottom
    s.add_plane((50, 81.6, 81.6), (0, 1, 0), (.75, .75, .75), WHITE_DIFF, t("wall"))    # Top
    _spheres(s, t("planet"))
    centers = [(20, 0, 120), (60, 0, 120), (20, 0, 60), (60, 0, 60)][:n_teapots]
    for c in centers:
        s.add_bezier(c, 4, TEAPOT_BPT, (0, .999, .999), WHITE_DIFF, teapot_transform(), t("blue"))
    s.add_light((50, 60, 85), (5000, 5000, 5000))
    return s


def scene_caustic():
    """Config 5: floor plane + one glass sphere + a light above it."""
    s = FlatScene()
    s.add_plane((50, 0, 81.6), (0, -1, 0), (.75, .75, .75), WHITE_DIFF)
    s.add_sphere(16.5, (50, 16.5, 88), (.999, .999, .999), REFR0)
    s.add_light((50, 60, 88), (5000, 5000, 5000))
    return s


# ------------------------------------------------------------------ scene description files
# The reference hard-codes its scene in Scene::init (raytracer/Scene.h:98-164).  A scene file says the same
# things line by line, `keyword key=value ...`, vectors as comma lists, '#' starts a comment:
#   material NAME [diff=r,g,b] [refl=r,g,b] [refr=r,g,b] [refrn=x] [refln=x]      Material(diff, refl, refr, refrn, refln)
#   texture  NAME file=wall.ppm                                                    Texture("wall.ppm"); procedural stand-in if unreadable
#   plane    p0=x,y,z n=x,y,z color=r,g,b material=NAME [texture=NAME]             PlaneObj(P0, n, color, material)
#   sphere   radius=r center=x,y,z color=r,g,b material=NAME [texture=NAME]        SphereObj(radius, org, color, material)
#   bezier   file=teapot.bpt center=x,y,z scale=s color=r,g,b material=NAME [texture=NAME] [trans=m00,...,m22]
#                                                                                  Bezier3Obj(C, scale, file, color, material, Trans)
#   light    pos=x,y,z color=r,g,b                                                 SpotLight(pos, color)
# Files named in a scene file are looked up next to it, then in the data directory.  The C++ host reads the same
# files with Scene::load (include/raytrace3/Scene.h).
def _scene_tokens(path):
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            line = line.split("#", 1)[0].split()
            if not line:
                continue
            kw, args, names = line[0], {}, []
            for tok in line[1:]:
                if "=" in tok:
                    k, v = tok.split("=", 1)
                    args[k] = v
                else:
                    names.append(tok)
            yield ln, kw, names, args


def load_scene(path, data_dir=None):
    """Reads a scene description file into a FlatScene (see the format above)."""
    here = os.path.dirname(os.path.abspath(path))
    data_dir = data_dir or DATA_DIR

    def find(name):
        for d in (here, data_dir):
            if os.path.exists(os.path.join(d, name)):
                return os.path.join(d, name)
        return None

    def vec(text, n, what, ln):
        v = [float(x) for x in text.split(",")]
        if len(v) != n:
            raise ValueError(f"{path}:{ln}: {what} needs {n} numbers")
        return v

    s, mats, texs = FlatScene(), {}, {}
    for ln, kw, names, a in _scene_tokens(path):
        def need(key):
            if key not in a:
                raise ValueError(f"{path}:{ln}: {kw} needs {key}=")
            return a[key]

        def tex_of():
            if "texture" not in a:
                return -1
            if a["texture"] not in texs:
                raise ValueError(f"{path}:{ln}: unknown texture {a['texture']}")
            return texs[a["texture"]]

        def mat_of():
            if need("material") not in mats:
                raise ValueError(f"{path}:{ln}: unknown material {a['material']}")
            return mats[a["material"]]

        if kw == "material":
            if len(names) != 1:
                raise ValueError(f"{path}:{ln}: material needs a name")
            mats[names[0]] = material(vec(a.get("diff", "0,0,0"), 3, "diff", ln), vec(a.get("refl", "0,0,0"), 3, "refl", ln),
                                      vec(a.get("refr", "0,0,0"), 3, "refr", ln), float(a.get("refrn", 1.5)), float(a.get("refln", 1.0)))
        elif kw == "texture":
            if len(names) != 1:
                raise ValueError(f"{path}:{ln}: texture needs a name")
            f = find(need("file"))
            stem = os.path.splitext(os.path.basename(a["file"]))[0]
            texs[names[0]] = s.add_texture(load_ppm(f) if f else procedural_texture(stem))
        elif kw == "plane":
            s.add_plane(vec(need("p0"), 3, "p0", ln), vec(need("n"), 3, "n", ln), vec(need("color"), 3, "color", ln), mat_of(), tex_of())
        elif kw == "sphere":
            s.add_sphere(float(need("radius")), vec(need("center"), 3, "center", ln), vec(need("color"), 3, "color", ln), mat_of(), tex_of())
        elif kw == "bezier":
            f = find(need("file"))
            if not f:
                raise ValueError(f"{path}:{ln}: cannot find {a['file']}")
            trans = np.array(vec(a["trans"], 9, "trans", ln)).reshape(3, 3) if "trans" in a else None
            s.add_bezier(vec(need("center"), 3, "center", ln), float(need("scale")), f, vec(need("color"), 3, "color", ln), mat_of(),
                         trans, tex_of())
        elif kw == "light":
            s.add_light(vec(need("pos"), 3, "pos", ln), vec(need("color"), 3, "color", ln))
        else:
            raise ValueError(f"{path}:{ln}: unknown keyword {kw}")
    return s


# BASELINE.json configs: name -> (scene factory, (w, h), rounds, per_round)
CONFIGS = {
    "c1_spheres_plane_256": (lambda: scene_spheres_plane(False), (256, 256), 1000, 100),
    "c1b_cornell_256": (lambda: scene_spheres_plane(True), (256, 256), 1000, 100),
    "c2_teapot_800x600": (lambda: scene_teapot(1), (800, 600), 10000, 100),
    "c3_teapot_800x600_10M": (lambda: scene_teapot(1), (800, 600), 100000, 100),
    "c4_teapot4_1080p": (lambda: scene_teapot(4), (1920, 1080), 10000, 100),
    "c5_caustic_1024": (scene_caustic, (1024, 1024), 500000, 100),
}


def to_int(x):
    """toInt, raytracer/Raytracer.h:24-26 (vectorised)."""
    x = np.asarray(x, dtype=np.float64)
    return (np.power(1 - np.exp(-x), 1 / 2.2) * 255 + .5).astype(np.int64)


def tonemap_bgr(pic_sum, n_images):
    """raytracer/Raytracer.h:460-470: rows flipped, channel k written verbatim (BGR), mean over n_images."""
    return to_int(pic_sum[::-1] / float(n_images)).astype(np.uint8)
