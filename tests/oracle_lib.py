"""ctypes wrapper of the CPU oracle (oracle/_build/liboracle.so) for the tests.

The oracle is the checker: tests, smoke() and bench.py's cpu_baseline leg are the
only things that may load it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytrace3_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

GATHER_GRID, GATHER_BRUTE, GATHER_KDTREE = 0, 1, 2

_D = C.POINTER(C.c_double)
SYMBOLS = {
    "oracle_create": (C.c_void_p, [C.c_int, C.c_int]),
    "oracle_destroy": (None, [C.c_void_p]),
    "oracle_set_radius_update": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint32]),
    "oracle_set_narrow_rays": (C.c_int, [C.c_void_p, C.c_int]),
    "oracle_set_scene": (C.c_int, [C.c_void_p, C.POINTER(abi.Scene)]),
    "oracle_eye_pass": (C.c_int, [C.c_void_p, C.POINTER(abi.Camera), C.c_uint64, C.c_uint32]),
    "oracle_eye_pass_rows": (C.c_int, [C.c_void_p, C.POINTER(abi.Camera), C.c_uint64, C.c_uint32, C.c_int, C.c_int]),
    "oracle_set_hitpoints": (C.c_int, [C.c_void_p, C.POINTER(abi.Camera), C.c_void_p, C.c_uint64]),
    "oracle_photon_pass": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int]),
    "oracle_resolve": (C.c_int, [C.c_void_p, _D, C.c_int]),
    "oracle_get_hitpoints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "oracle_get_counters": (C.c_int, [C.c_void_p, C.POINTER(abi.Counters)]),
    "oracle_near_collide": (C.c_int, [C.c_void_p, C.c_uint64, _D, _D, C.c_uint64, C.c_uint32, C.c_void_p]),
    "oracle_newton": (C.c_int, [C.c_void_p, C.c_uint64, _D, _D, C.POINTER(C.c_int32), C.c_uint64, C.c_uint32, _D]),
    "oracle_photon_deposits": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, _D, C.c_uint64, C.POINTER(C.c_uint64)]),
    "oracle_bezier_eval": (None, [_D, C.c_double, C.c_double, _D]),
    "oracle_aabb_intersect": (C.c_int, [_D, _D, _D, _D]),
    "oracle_refracted": (None, [_D, _D, C.c_double, _D]),
    "oracle_reflected_unit": (None, [_D, _D, _D]),
    "oracle_rotated": (None, [_D, _D, C.c_double, _D]),
    "oracle_rand_refl": (None, [_D, C.c_double, C.c_double, _D]),
    "oracle_to_int": (C.c_int, [C.c_double]),
    "oracle_tonemap": (None, [_D, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]),
    "oracle_hitpoint_update": (None, [_D, C.POINTER(C.c_int32), _D, C.POINTER(C.c_int32)]),
    "oracle_mwc_kat": (None, [C.c_int, C.POINTER(C.c_uint32), _D]),
    "oracle_rng_u32": (C.c_uint32, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "oracle_texture_color_uv": (None, [C.POINTER(abi.Texture), C.c_double, C.c_double, _D]),
    "oracle_camera_lookat": (None, [_D, _D, C.c_int, C.c_int, C.POINTER(abi.Camera)]),
    "oracle_camera_emit": (None, [C.POINTER(abi.Camera), C.c_int, C.c_int, _D, _D]),
    "oracle_plane_intersect": (C.c_int, [C.POINTER(abi.Object), _D, _D, _D]),
    "oracle_sphere_intersect": (C.c_int, [C.POINTER(abi.Object), _D, _D, _D, C.POINTER(C.c_int)]),
}


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load():
    if not os.path.exists(ORACLE_SO):
        build()
    return abi.bind(C.CDLL(ORACLE_SO), SYMBOLS)


def dp(a):
    return a.ctypes.data_as(_D)


def vec(*x):
    return np.array(x, dtype=np.float64)


class OracleContext:
    """Same method names as raytrace3_amd.SppmContext so parity tests read symmetrically."""

    def __init__(self, lib=None, newton_fp32=False, rng_literal=False):
        self.lib = lib or load()
        self._ctx = C.c_void_p(self.lib.oracle_create(int(newton_fp32), int(rng_literal)))
        self.cam = None

    def close(self):
        if self._ctx:
            self.lib.oracle_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_radius_update(self, initial_n, rounds_per_update=1):
        assert self.lib.oracle_set_radius_update(self._ctx, int(initial_n), int(rounds_per_update)) == 0

    def set_narrow_rays(self, on):
        assert self.lib.oracle_set_narrow_rays(self._ctx, int(bool(on))) == 0

    def set_scene(self, scene):
        self._keep = scene
        assert self.lib.oracle_set_scene(self._ctx, C.byref(scene.c_scene)) == 0

    def eye_pass(self, cam, seed, replica=0):
        self.cam = cam
        assert self.lib.oracle_eye_pass(self._ctx, C.byref(cam), seed, replica) == 0

    def eye_pass_rows(self, cam, seed, row_begin, row_end, replica=0):
        self.cam = cam
        assert self.lib.oracle_eye_pass_rows(self._ctx, C.byref(cam), seed, replica, row_begin, row_end) == 0

    def set_hitpoints(self, cam, hp):
        self.cam = cam
        hp = np.ascontiguousarray(hp, dtype=abi.HITPOINT_DTYPE)
        assert self.lib.oracle_set_hitpoints(self._ctx, C.byref(cam), hp.ctypes.data_as(C.c_void_p), len(hp)) == 0

    def photon_pass(self, rounds, per_round, seed, replica=0, photon_begin=0, photon_count=0, mode=GATHER_GRID):
        assert self.lib.oracle_photon_pass(self._ctx, rounds, per_round, photon_begin, photon_count, seed, replica, mode) == 0

    def resolve(self, accumulate_into=None):
        h, w = self.cam.h, self.cam.w
        pic = accumulate_into if accumulate_into is not None else np.zeros((h, w, 3), dtype=np.float64)
        assert self.lib.oracle_resolve(self._ctx, dp(pic), 1 if accumulate_into is not None else 0) == 0
        return pic

    def hitpoints(self):
        n = C.c_uint64(0)
        self.lib.oracle_get_hitpoints(self._ctx, None, 0, C.byref(n))
        out = np.zeros(n.value, dtype=abi.HITPOINT_DTYPE)
        if n.value:
            self.lib.oracle_get_hitpoints(self._ctx, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
        return out

    def counters(self):
        c = abi.Counters()
        self.lib.oracle_get_counters(self._ctx, C.byref(c))
        return c.as_dict()

    def near_collide(self, org, dir, seed, replica=0):
        org = np.ascontiguousarray(org, dtype=np.float64)
        dir = np.ascontiguousarray(dir, dtype=np.float64)
        out = np.zeros(org.shape[0], dtype=abi.COLLISION_DTYPE)
        assert self.lib.oracle_near_collide(self._ctx, org.shape[0], dp(org), dp(dir), seed, replica,
                                            out.ctypes.data_as(C.c_void_p)) == 0
        return out

    def newton(self, org, dir, patch, seed, replica=0):
        org = np.ascontiguousarray(org, dtype=np.float64)
        dir = np.ascontiguousarray(dir, dtype=np.float64)
        patch = np.ascontiguousarray(patch, dtype=np.int32)
        out = np.zeros((org.shape[0], 3), dtype=np.float64)
        assert self.lib.oracle_newton(self._ctx, org.shape[0], dp(org), dp(dir),
                                      patch.ctypes.data_as(C.POINTER(C.c_int32)), seed, replica, dp(out)) == 0
        return out

    def deposits(self, rounds, per_round, seed, replica=0, photon_begin=0, photon_count=0, cap=1 << 20):
        out = np.zeros((cap, 11), dtype=np.float64)
        n = C.c_uint64(0)
        rc = self.lib.oracle_photon_deposits(self._ctx, rounds, per_round, photon_begin, photon_count, seed, replica,
                                             dp(out), cap, C.byref(n))
        assert rc == 0, rc
        return out[:n.value]


class ParallelOracle:
    """The serial oracle driven over host threads for the sizes one thread does not finish in seconds (1920 x 1080 with
    128 patches, 50 M photons).  Nothing is restated: every piece of work is a call of the serial oracle.  The eye pass
    is cut into strips of rows (counter-based streams: a pixel's HitPoints do not depend on the pixels before it) whose
    HitPoints are concatenated in row order, i.e. in the order of OracleContext.eye_pass; every worker context then holds
    ALL HitPoints and runs ranges of the photon indices, and since the pass is linear in tao and newN the workers'
    images, per-HitPoint counts and tallies add up to the serial pass (tests/test_oracle_kat.py holds the two against
    each other; the fp64 sums differ in their order only).  ctypes releases the interpreter lock during a call."""

    def __init__(self, threads, newton_fp32=False):
        self.lib = load()
        self.parts = [OracleContext(self.lib, newton_fp32=newton_fp32) for _ in range(max(1, int(threads)))]
        self.cam = None
        self._eye, self._hp = {}, None

    def close(self):
        for p in self.parts:
            p.close()

    def set_scene(self, scene):
        for p in self.parts:
            p.set_scene(scene)

    def set_narrow_rays(self, on):
        for p in self.parts:
            p.set_narrow_rays(on)

    def _run(self, jobs, fn):
        """jobs are dealt to the worker contexts from one list, so an uneven job (rows that cross a teapot) does not
        leave the other threads idle."""
        import threading
        lock, nxt, out, errs = threading.Lock(), [0], [None] * len(jobs), []

        def work(part):
            while True:
                with lock:
                    k = nxt[0]
                    nxt[0] += 1
                if k >= len(jobs) or errs:
                    return
                try:
                    out[k] = fn(part, jobs[k])
                except BaseException as e:       # noqa: BLE001 - handed to the caller's thread below
                    errs.append(e)
                    return

        ts = [threading.Thread(target=work, args=(p,)) for p in self.parts]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]
        return out

    def _each(self, fn):
        import threading
        ts = [threading.Thread(target=fn, args=(p,)) for p in self.parts]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    def eye_pass(self, cam, seed, replica=0):
        self.cam = cam
        n = min(cam.h, 8 * len(self.parts))
        edges = [cam.h * k // n for k in range(n + 1)]

        def strip(part, k):
            part.eye_pass_rows(cam, seed, edges[k], edges[k + 1], replica)
            return part.hitpoints(), part.counters()

        res = self._run(list(range(n)), strip)
        self._hp = np.concatenate([r[0] for r in res])
        self._eye = dict.fromkeys(res[0][1], 0)
        for _, c in res:
            for k, v in c.items():
                self._eye[k] += v
        self._each(lambda part: part.set_hitpoints(cam, self._hp))       # resets the contexts' tallies

    def photon_pass(self, rounds, per_round, seed, replica=0, photon_begin=0, photon_count=0, n_lights=1):
        total = rounds * per_round * n_lights
        end = total if photon_count == 0 else min(total, photon_begin + photon_count)
        n = max(1, min(end - photon_begin, 8 * len(self.parts)))
        edges = [photon_begin + (end - photon_begin) * k // n for k in range(n + 1)]
        jobs = [(edges[k], edges[k + 1] - edges[k]) for k in range(n) if edges[k + 1] > edges[k]]   # a count of 0 means "all"
        self._run(jobs, lambda part, j: part.photon_pass(rounds, per_round, seed, replica, j[0], j[1]))

    def _busy(self):
        return [p for p in self.parts if p.counters()["photons"]]

    def resolve(self):
        pic = np.zeros((self.cam.h, self.cam.w, 3), dtype=np.float64)
        for p in self._busy():
            p.resolve(accumulate_into=pic)
        return pic

    def hitpoints(self):
        out = self._hp.copy()
        for p in self._busy():
            hp = p.hitpoints()
            out["tao"] += hp["tao"]
            out["newN"] += hp["newN"]
        return out

    def counters(self):
        tot = dict(self._eye)
        for p in self.parts:
            for k, v in p.counters().items():
                if k != "hitpoints":
                    tot[k] += v
        tot["hitpoints"] = len(self._hp)
        return tot


def tonemap(pic_hw3, n_images, rgb=False, lib=None):
    """The byte image SPPMRayTracer::render writes (raytracer/Raytracer.h:460-466), by the C oracle."""
    lib = lib or load()
    pic = np.ascontiguousarray(pic_hw3, dtype=np.float64)
    h, w, _ = pic.shape
    out = np.zeros((h, w, 3), dtype=np.uint8)
    lib.oracle_tonemap(pic.ctypes.data_as(_D), w, h, float(n_images), int(bool(rgb)), out.ctypes.data_as(C.c_void_p))
    return out
