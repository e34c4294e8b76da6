"""Scene description files (raytrace3_amd/data/*.scene) against the built-in scene builders: the flattened
sppm_scene handed to the library is the same, field for field."""
import ctypes as C
import os

import numpy as np
import pytest

from raytrace3_amd import abi, scenes


def _flat(scene):
    c = scene.c_scene
    objs = bytes(C.string_at(c.objects, C.sizeof(abi.Object) * c.n_objects))
    patches = np.ctypeslib.as_array(c.patches, shape=(c.n_patches * 48,)).copy() if c.n_patches else np.zeros(0)
    lights = bytes(C.string_at(c.lights, C.sizeof(abi.Light) * c.n_lights))
    texs = [np.array(t) for t in scene.texture_list]
    return c.n_objects, objs, patches, c.n_lights, lights, texs


@pytest.mark.parametrize("name,builder", [("cornell_teapot", lambda: scenes.scene_teapot(1)),
                                          ("four_teapots", lambda: scenes.scene_teapot(4)),
                                          ("caustic", scenes.scene_caustic)])
def test_scene_file_equals_builder(name, builder):
    a, b = _flat(scenes.load_scene(os.path.join(scenes.DATA_DIR, name + ".scene"))), _flat(builder())
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(a[2], b[2])
    assert a[3] == b[3] and a[4] == b[4]
    assert len(a[5]) == len(b[5]) and all(np.array_equal(x, y) for x, y in zip(a[5], b[5]))


def test_scene_file_errors(tmp_path):
    def load(text):
        p = tmp_path / "s.scene"
        p.write_text(text)
        return scenes.load_scene(str(p))
    with pytest.raises(ValueError, match="unknown keyword"):
        load("cube size=1\n")
    with pytest.raises(ValueError, match="unknown material"):
        load("sphere radius=1 center=0,0,0 color=1,1,1 material=GOLD\n")
    with pytest.raises(ValueError, match="needs 3 numbers"):
        load("material M diff=1,1\n")
    with pytest.raises(ValueError, match="needs radius="):
        load("material M diff=1,1,1\nsphere center=0,0,0 color=1,1,1 material=M\n")
    with pytest.raises(ValueError, match="cannot find"):
        load("material M diff=1,1,1\nbezier file=nothing.bpt center=0,0,0 scale=1 color=1,1,1 material=M\n")
    # comments, blank lines, a texture read from a PPM next to the scene file
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    scenes.save_ppm(tmp_path / "t.ppm", img)
    s = load("# a comment\n\nmaterial M diff=.5,.5,.5 refrn=0   # trailing comment\ntexture T file=t.ppm\n"
             "plane p0=0,0,0 n=0,2,0 color=1,1,1 material=M texture=T\nlight pos=0,5,0 color=1,1,1\n")
    assert np.array_equal(s.texture_list[0], img)
    assert s.objects[0]["n"] == (0.0, 1.0, 0.0) and s.objects[0]["texture"] == 0     # PlaneObj normalises n


def _bernstein(d, t):
    from math import comb
    return np.array([comb(d, i) * t ** i * (1 - t) ** (d - i) for i in range(d + 1)])


def test_bpt_patches_of_lower_degree_are_raised_to_bicubic(tmp_path):
    """A .bpt patch of degree m x n below 3 x 3 is the same surface after the loaders raise it to the bicubic form
    Bezier3::P[16] stands for (raytracer/Bezier.h:69-111: P[4 j + k], k along u); the Python and the C++ loader give the
    same control points; a degree above 3 is refused."""
    import shutil
    import subprocess
    rng = np.random.default_rng(5)
    degrees = [(2, 2), (1, 3), (3, 1), (0, 2), (3, 3)]
    grids = [rng.normal(size=(m + 1, n + 1, 3)) * 5 for m, n in degrees]
    path = tmp_path / "mixed.bpt"
    with open(path, "w") as f:
        f.write(f"{len(degrees)}\n")
        for (m, n), g in zip(degrees, grids):
            f.write(f"{m} {n}\n")
            for j in range(m + 1):
                for k in range(n + 1):
                    f.write(" ".join(repr(float(x)) for x in g[j, k]) + "\n")
    trans = scenes.teapot_transform()
    patches = scenes.load_bpt(str(path), (1.0, 2.0, 3.0), 2.0, trans)
    assert patches.shape == (len(degrees), 16, 3)
    for (m, n), g, P in zip(degrees, grids, patches):
        world = np.einsum("ab,jkb->jka", trans, 2.0 * g) + np.array([1.0, 2.0, 3.0])
        for u, v in rng.random((20, 2)):
            low = np.einsum("j,k,jkc->c", _bernstein(m, v), _bernstein(n, u), world)
            cubic = np.einsum("j,k,jkc->c", _bernstein(3, v), _bernstein(3, u), P.reshape(4, 4, 3))
            assert np.allclose(low, cubic, rtol=0, atol=1e-12), (m, n)
    # a bicubic patch passes through as it is
    assert np.allclose(patches[4], (np.einsum("ab,jkb->jka", trans, 2.0 * grids[4]) + [1.0, 2.0, 3.0]).reshape(16, 3), rtol=0, atol=1e-13)
    # the C++ host's loader (include/raytrace3/Bezier.h) reads the same file into the same control points
    if shutil.which("g++"):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        src = tmp_path / "dump.cpp"
        src.write_text('#include "Bezier.h"\n#include <cstdio>\n#include <cmath>\n'
                       'int main(int, char** a) {\n'
                       '  cv::Matx33d T, T2; T(0,0)=1; T(1,2)=1; T(2,1)=1; double th=(90./180.)*PI;\n'
                       '  T2(0,0)=cos(th); T2(0,2)=sin(th); T2(1,1)=1; T2(2,0)=-sin(th); T2(2,2)=cos(th);\n'
                       '  Bezier3Obj o(Vec3(1,2,3), 2.0, a[1], Color(1,1,1), Material(Color(.5,.5,.5), Color(0,0,0), Color(0,0,0)), T2*T);\n'
                       '  for (size_t i = 0; i < o.patch_count(); i++) for (int k = 0; k < 16; k++)\n'
                       '    printf("%.17g %.17g %.17g\\n", o.patch(i).P[k].val[0], o.patch(i).P[k].val[1], o.patch(i).P[k].val[2]);\n}\n')
        exe = tmp_path / "dump"
        # Bezier.h alone is enough for a host (its members that solve call the library, so the program links against it;
        # this one only loads a file and makes no GPU call)
        libdir = os.path.join(root, "raytrace3_amd", "csrc")
        subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include", "raytrace3"), "-I", os.path.join(root, "include"),
                        str(src), "-o", str(exe), "-L" + libdir, "-lsppm_hip", "-Wl,-rpath," + libdir, "-pthread"], check=True)
        out = subprocess.run([str(exe), str(path)], check=True, capture_output=True, text=True).stdout
        got = np.array([[float(x) for x in line.split()] for line in out.strip().splitlines()]).reshape(-1, 16, 3)
        assert got.shape == patches.shape and np.allclose(got, patches, rtol=0, atol=1e-12)
    with open(path, "w") as f:
        f.write("1\n4 3\n" + "0 0 0\n" * 20)
    with pytest.raises(ValueError, match="degree 4 x 3"):
        scenes.load_bpt(str(path))
