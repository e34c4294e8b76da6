"""ctypes mirror of include/sppm_abi.h and a thin wrapper over libsppm_hip.so.

The structures here are the C-ABI structs field for field.  `SppmLib` loads the
HIP library and fails loudly when it is missing or no GPU is usable: there is no
CPU fallback in the product path.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsppm_hip.so")

SPPM_OBJ_PLANE, SPPM_OBJ_SPHERE, SPPM_OBJ_BEZIER = 0, 1, 2
NOALLOC_MAX_RAYS = 16384      # SPPM_NOALLOC_MAX_RAYS of include/sppm_abi.h


class Material(C.Structure):
    _fields_ = [("diff", C.c_double * 3), ("refl", C.c_double * 3), ("refr", C.c_double * 3),
                ("refrn", C.c_double), ("refln", C.c_double)]


class Object(C.Structure):
    _fields_ = [("type", C.c_int32), ("texture", C.c_int32), ("color", C.c_double * 3),
                ("material", Material), ("p0", C.c_double * 3), ("n", C.c_double * 3),
                ("radius", C.c_double), ("first_patch", C.c_int32), ("n_patches", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("texels", C.POINTER(C.c_uint8))]


class Light(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("color", C.c_double * 3)]


class Scene(C.Structure):
    _fields_ = [("n_objects", C.c_int32), ("objects", C.POINTER(Object)),
                ("n_patches", C.c_int32), ("patches", C.POINTER(C.c_double)),
                ("n_textures", C.c_int32), ("textures", C.POINTER(Texture)),
                ("n_lights", C.c_int32), ("lights", C.POINTER(Light))]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("dir", C.c_double * 3), ("du", C.c_double * 3),
                ("dv", C.c_double * 3), ("w", C.c_int32), ("h", C.c_int32)]


class HitPoint(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("n", C.c_double * 3), ("v", C.c_double * 3),
                ("wgt", C.c_double * 3), ("tao", C.c_double * 3), ("R2", C.c_double),
                ("obj", C.c_int32), ("x", C.c_int32), ("y", C.c_int32), ("node", C.c_int32),
                ("N", C.c_int32), ("newN", C.c_int32)]


HITPOINT_DTYPE = np.dtype([("pos", "f8", 3), ("n", "f8", 3), ("v", "f8", 3), ("wgt", "f8", 3),
                           ("tao", "f8", 3), ("R2", "f8"), ("obj", "i4"), ("x", "i4"), ("y", "i4"),
                           ("node", "i4"), ("N", "i4"), ("newN", "i4")])
assert HITPOINT_DTYPE.itemsize == C.sizeof(HitPoint)


class Counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("photons", "segments", "deposits", "candidates", "accepted",
                                           "hitpoints", "newton_calls", "eye_segments")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Timings(C.Structure):
    _fields_ = [("eye_trace_ms", C.c_double), ("grid_build_ms", C.c_double), ("photon_trace_ms", C.c_double),
                ("deposit_sort_ms", C.c_double), ("gather_ms", C.c_double), ("resolve_ms", C.c_double),
                ("gather_launches", C.c_int32), ("trace_launches", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class DeviceStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("stored_deposits", "pair_tests", "device_bytes", "deposit_room")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Collision(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("n", C.c_double * 3), ("color", C.c_double * 3),
                ("dist", C.c_double), ("obj", C.c_int32), ("inside", C.c_int32)]


COLLISION_DTYPE = np.dtype([("pos", "f8", 3), ("n", "f8", 3), ("color", "f8", 3), ("dist", "f8"),
                            ("obj", "i4"), ("inside", "i4")])
assert COLLISION_DTYPE.itemsize == C.sizeof(Collision)

# Every symbol include/sppm_abi.h declares, with its ctypes signature.
ABI_SYMBOLS = {
    "sppm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "sppm_destroy": (None, [C.c_void_p]),
    "sppm_last_error": (C.c_char_p, [C.c_void_p]),
    "sppm_set_scene": (C.c_int, [C.c_void_p, C.POINTER(Scene)]),
    "sppm_eye_pass": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.c_uint64, C.c_uint32]),
    "sppm_photon_pass": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]),
    "sppm_resolve": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    "sppm_resolve_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sppm_film_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "sppm_film_destroy": (None, [C.c_void_p]),
    "sppm_film_last_error": (C.c_char_p, [C.c_void_p]),
    "sppm_film_clear": (C.c_int, [C.c_void_p]),
    "sppm_film_add": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sppm_film_tonemap": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_void_p]),
    "sppm_film_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "sppm_film_allreduce": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "sppm_film_tonemap_total": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_void_p]),
    "sppm_film_read_total": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "sppm_film_write": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "sppm_set_radius_update": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint32]),
    "sppm_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sppm_get_stream": (C.c_void_p, [C.c_void_p]),
    "sppm_synchronize": (C.c_int, [C.c_void_p]),
    "sppm_stats_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "sppm_get_hitpoints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "sppm_get_counters": (C.c_int, [C.c_void_p, C.POINTER(Counters)]),
    "sppm_get_timings": (C.c_int, [C.c_void_p, C.POINTER(Timings)]),
    "sppm_get_device_stats": (C.c_int, [C.c_void_p, C.POINTER(DeviceStats)]),
    "sppm_debug_near_collide": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.c_uint64, C.c_uint32, C.c_void_p]),
    "sppm_near_collide_noalloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                            C.c_uint64, C.c_uint32, C.c_void_p]),
    "sppm_debug_photon_deposits": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32,
                                             C.POINTER(C.c_double), C.c_uint64, C.POINTER(C.c_uint64)]),
    "sppm_view_trace": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_int32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                                  C.POINTER(C.c_uint64)]),
    "sppm_photon_trace": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_uint64, C.c_uint32]),
    "sppm_debug_newton": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int32), C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
}


def bind(lib, symbols):
    for name, (res, args) in symbols.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def load_library(path=None):
    # SPPM_LIB_PATH selects another build of the same library (kernel A/B experiments)
    path = path or os.environ.get("SPPM_LIB_PATH") or LIB_PATH
    # PyTorch wheels carry their own copy of the ROCm runtime.  Two copies in one process do not both get the GPU:
    # the one loaded second reports "no HIP GPUs".  Loaded after torch, this library binds to torch's copy and
    # all is well, so a process that has torch (bench.py and the multi-GPU host do use it, for streams and RCCL)
    # lets torch load first.  SPPM_NO_TORCH_PRELOAD=1 skips this for a host that never touches torch.
    # Several contexts on one GPU (iterations in flight) want a hardware queue each: HIP's default of 4 queues per process
    # folds more streams than that onto shared queues, whose kernels then run one after the other.  The runtime reads
    # the variable when it starts; a host that has already used the GPU keeps what it had.
    # SPPM_HW_QUEUES=0 leaves the environment alone.
    if os.environ.get("SPPM_HW_QUEUES", "8") != "0":
        os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("SPPM_HW_QUEUES", "8")
    if "torch" not in sys.modules and not os.environ.get("SPPM_NO_TORCH_PRELOAD"):
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the SPPM hot path)")
    return bind(C.CDLL(path), ABI_SYMBOLS)


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class SppmError(RuntimeError):
    pass


class SppmContext:
    """One GPU context of libsppm_hip.so (mirrors the sppm_* call sequence)."""

    prefix = "sppm"

    def __init__(self, device_id=0, lib=None):
        self.lib = lib if lib is not None else load_library()
        self._ctx = C.c_void_p()
        rc = self.lib.sppm_create(C.byref(self._ctx), int(device_id))
        if rc != 0:
            raise SppmError(f"sppm_create(device {device_id}) failed with {rc}: no usable HIP device?")
        self.cam = None

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.sppm_last_error(self._ctx)
            raise SppmError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if self._ctx:
            self.lib.sppm_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, scene):
        """scene: raytrace3_amd.scenes.FlatScene"""
        self._scene_keepalive = scene
        self._check(self.lib.sppm_set_scene(self._ctx, C.byref(scene.c_scene)), "sppm_set_scene")

    def eye_pass(self, cam, seed, replica=0):
        self.cam = cam
        self._check(self.lib.sppm_eye_pass(self._ctx, C.byref(cam), seed, replica), "sppm_eye_pass")

    def photon_pass(self, rounds, per_round, seed, replica=0, photon_begin=0, photon_count=0):
        self._check(self.lib.sppm_photon_pass(self._ctx, rounds, per_round, photon_begin, photon_count, seed, replica),
                    "sppm_photon_pass")

    def resolve(self, accumulate_into=None):
        h, w = self.cam.h, self.cam.w
        pic = accumulate_into if accumulate_into is not None else np.zeros((h, w, 3), dtype=np.float64)
        assert pic.shape == (h, w, 3) and pic.dtype == np.float64 and pic.flags.c_contiguous
        self._check(self.lib.sppm_resolve(self._ctx, _dptr(pic), 1 if accumulate_into is not None else 0), "sppm_resolve")
        return pic

    def synchronize(self):
        """Wait for the queued passes; raises if one of them overflowed a queue."""
        self._check(self.lib.sppm_synchronize(self._ctx), "sppm_synchronize")

    def set_radius_update(self, initial_n, rounds_per_update=1):
        """initial_n > 0: HitPoints start with that N and HitPoint::update() runs after every rounds_per_update
        rounds of a photon pass (a deviation from the reference, where N = 0 keeps it from running); 0: off."""
        self._check(self.lib.sppm_set_radius_update(self._ctx, int(initial_n), int(rounds_per_update)), "sppm_set_radius_update")

    def set_stream(self, hip_stream):
        """Queue this context's passes on the caller's HIP stream (an integer handle such as
        torch.cuda.Stream().cuda_stream; None / 0: back to the context's own stream)."""
        self._check(self.lib.sppm_set_stream(self._ctx, C.c_void_p(int(hip_stream or 0))), "sppm_set_stream")

    def stream_handle(self):
        """The HIP stream this context queues on, as an integer (torch.cuda.ExternalStream(handle) wraps it)."""
        return int(self.lib.sppm_get_stream(self._ctx) or 0)

    def stats_mode(self, accumulate):
        """accumulate=True: counters / timings run on over passes from now; False: restart at each eye pass."""
        self._check(self.lib.sppm_stats_mode(self._ctx, 1 if accumulate else 0), "sppm_stats_mode")

    def resolve_device(self, dev_ptr):
        self._check(self.lib.sppm_resolve_device(self._ctx, C.c_void_p(dev_ptr)), "sppm_resolve_device")

    def hitpoints(self):
        n = C.c_uint64(0)
        self._check(self.lib.sppm_get_hitpoints(self._ctx, None, 0, C.byref(n)), "sppm_get_hitpoints")
        out = np.zeros(n.value, dtype=HITPOINT_DTYPE)
        if n.value:
            self._check(self.lib.sppm_get_hitpoints(self._ctx, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)),
                        "sppm_get_hitpoints")
        return out

    def counters(self):
        c = Counters()
        self._check(self.lib.sppm_get_counters(self._ctx, C.byref(c)), "sppm_get_counters")
        return c.as_dict()

    def device_stats(self):
        """The library's own tallies: stored deposits, pair tests of the gather, device bytes held, layout room."""
        d = DeviceStats()
        self._check(self.lib.sppm_get_device_stats(self._ctx, C.byref(d)), "sppm_get_device_stats")
        return d.as_dict()

    def timings(self):
        t = Timings()
        self._check(self.lib.sppm_get_timings(self._ctx, C.byref(t)), "sppm_get_timings")
        return t.as_dict()

    def near_collide(self, org, dir, seed, replica=0, noalloc=False):
        """Scene::NearCollide for a batch of rays.  noalloc=True takes the entry point that stages through the library's
        static device array and allocates nothing (at most NOALLOC_MAX_RAYS rays per call)."""
        org = np.ascontiguousarray(org, dtype=np.float64)
        dir = np.ascontiguousarray(dir, dtype=np.float64)
        n = org.shape[0]
        out = np.zeros(n, dtype=COLLISION_DTYPE)
        fn, name = ((self.lib.sppm_near_collide_noalloc, "sppm_near_collide_noalloc") if noalloc else
                    (self.lib.sppm_debug_near_collide, "sppm_debug_near_collide"))
        self._check(fn(self._ctx, n, _dptr(org), _dptr(dir), seed, replica, out.ctypes.data_as(C.c_void_p)), name)
        return out

    def newton(self, org, dir, patch, seed, replica=0):
        org = np.ascontiguousarray(org, dtype=np.float64)
        dir = np.ascontiguousarray(dir, dtype=np.float64)
        patch = np.ascontiguousarray(patch, dtype=np.int32)
        n = org.shape[0]
        out = np.zeros((n, 3), dtype=np.float64)
        self._check(self.lib.sppm_debug_newton(self._ctx, n, _dptr(org), _dptr(dir),
                                               patch.ctypes.data_as(C.POINTER(C.c_int32)), seed, replica, _dptr(out)),
                    "sppm_debug_newton")
        return out

    def view_trace(self, org, dir, wgt, pixel, w, seed, replica=0, dep=None):
        """RayTracer::ViewTrace for eye rays of the caller's making; returns the HitPoints they leave."""
        org = np.ascontiguousarray(org, dtype=np.float64); dir = np.ascontiguousarray(dir, dtype=np.float64)
        wgt = np.ascontiguousarray(wgt, dtype=np.float64); pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        n = org.shape[0]
        depp = None if dep is None else np.ascontiguousarray(dep, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32))
        cap = 15 * n
        out = np.zeros(cap, dtype=HITPOINT_DTYPE)
        got = C.c_uint64(0)
        self._check(self.lib.sppm_view_trace(self._ctx, n, _dptr(org), _dptr(dir), depp, _dptr(wgt), pixel.ctypes.data_as(C.POINTER(C.c_uint32)),
                                             int(w), seed, replica, out.ctypes.data_as(C.c_void_p), cap, C.byref(got)), "sppm_view_trace")
        return out[:got.value]

    def photon_trace(self, org, dir, flux, ids, seed, replica=0, dep=None):
        """RayTracer::PhotonTrace for photons of the caller's making, gathered into the HitPoints of the last eye pass."""
        org = np.ascontiguousarray(org, dtype=np.float64); dir = np.ascontiguousarray(dir, dtype=np.float64)
        flux = np.ascontiguousarray(flux, dtype=np.float64); ids = np.ascontiguousarray(ids, dtype=np.uint32)
        depp = None if dep is None else np.ascontiguousarray(dep, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32))
        self._check(self.lib.sppm_photon_trace(self._ctx, org.shape[0], _dptr(org), _dptr(dir), depp, _dptr(flux),
                                               ids.ctypes.data_as(C.POINTER(C.c_uint32)), seed, replica), "sppm_photon_trace")

    def deposits(self, rounds, per_round, seed, replica=0, photon_begin=0, photon_count=1, cap=1 << 20):
        out = np.zeros((cap, 11), dtype=np.float64)
        n = C.c_uint64(0)
        self._check(self.lib.sppm_debug_photon_deposits(self._ctx, rounds, per_round, photon_begin, photon_count, seed, replica,
                                                        _dptr(out), cap, C.byref(n)), "sppm_debug_photon_deposits")
        return out[:n.value]


class Film:
    """The device-resident image sum of SPPMRayTracer::render (raytracer/Raytracer.h:449-470):
    contexts add their resolved images to it, tonemap() gives the bytes the reference writes."""

    def __init__(self, w, h, device=0, lib=None):
        self.lib = lib or load_library()
        self.w, self.h = int(w), int(h)
        self._film = C.c_void_p()
        rc = self.lib.sppm_film_create(C.byref(self._film), device, self.w, self.h)
        if rc != 0:
            self._film = C.c_void_p()
            raise SppmError(f"sppm_film_create failed (rc={rc})")

    def _check(self, rc, what):
        if rc != 0:
            raise SppmError(f"{what} failed (rc={rc}): {self.lib.sppm_film_last_error(self._film).decode()}")

    def add(self, ctx):
        """film += the image of ctx's last photon pass (queued on ctx's stream)."""
        ctx._check(self.lib.sppm_film_add(ctx._ctx, self._film), "sppm_film_add")

    def clear(self):
        self._check(self.lib.sppm_film_clear(self._film), "sppm_film_clear")

    def read(self):
        out = np.zeros((self.h, self.w, 3), dtype=np.float64)
        self._check(self.lib.sppm_film_read(self._film, _dptr(out)), "sppm_film_read")
        return out

    def write(self, pic):
        pic = np.ascontiguousarray(pic, dtype=np.float64)
        if pic.shape != (self.h, self.w, 3):
            raise ValueError("film image must be [h][w][3]")
        self._check(self.lib.sppm_film_write(self._film, _dptr(pic)), "sppm_film_write")

    def tonemap(self, n_images, rgb=False, total=False):
        """total=True: of the sum over GPUs the last allreduce() left in this film."""
        out = np.zeros((self.h, self.w, 3), dtype=np.uint8)
        fn, name = ((self.lib.sppm_film_tonemap_total, "sppm_film_tonemap_total") if total else (self.lib.sppm_film_tonemap, "sppm_film_tonemap"))
        self._check(fn(self._film, float(n_images), int(bool(rgb)), out.ctypes.data_as(C.c_void_p)), name)
        return out

    @staticmethod
    def allreduce(films):
        """One RCCL all-reduce over films on distinct GPUs of this process: each receives the sum of all (read_total())."""
        arr = (C.c_void_p * len(films))(*[f._film for f in films])
        films[0]._check(films[0].lib.sppm_film_allreduce(arr, len(films)), "sppm_film_allreduce")

    def read_total(self):
        out = np.zeros((self.h, self.w, 3), dtype=np.float64)
        self._check(self.lib.sppm_film_read_total(self._film, _dptr(out)), "sppm_film_read_total")
        return out

    def close(self):
        if self._film:
            self.lib.sppm_film_destroy(self._film)
            self._film = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
