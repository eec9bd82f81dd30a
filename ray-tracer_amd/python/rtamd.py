"""ctypes binding of librt_mi355x.so (the C ABI of include/rt_mi355x.h).

Plumbing for the Python tests and bench.py only: it passes plain pointers and
sizes, never torch types.  Importing it loads the HIP library and fails loudly
when the library has not been built; there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTAMD_LIBRARY", os.path.join(os.path.dirname(_HERE), "librt_mi355x.so"))

SPHERE_DTYPE = np.dtype(
    [("center", "<f4", 3), ("color", "<f4", 3), ("radius", "<f4"), ("material", "<i4"), ("fuzz", "<f4")]
)
assert SPHERE_DTYPE.itemsize == 36

MAT_GLASS, MAT_DIFFUSE, MAT_METAL, MAT_FUZZYMETAL = 0, 1, 2, 3
OPT_FILTER, OPT_TILE_SPHERES, OPT_WORKGROUPS_PER_CU, OPT_BVH_LEAF_SPHERES, OPT_BVH_MIN_SPHERES, OPT_BVH_LDS_NODES = 0, 1, 2, 3, 4, 5
OPT_BVH_SUSPEND_LANES = 6
OPT_CHUNKS_PER_WAVE = 7
OPT_SUBTILES = 8
OPT_COUNT_WORK = 9
OPT_DIRECT_ADD_SAMPLES = 10
OPT_MAX_CHUNKS_PER_LAUNCH = 11
OPT_PROGRESSIVE_PASSES_PER_LAUNCH = 12
FILTER_AUTO, FILTER_EXACT, FILTER_CONSERVATIVE, FILTER_MATRIX, FILTER_BVH = -1, 0, 1, 2, 3

# every symbol include/rt_mi355x.h declares
EXPORTS = [
    "rt_abi_version", "rt_status_string", "rt_last_error", "rt_create", "rt_destroy", "rt_set_scene",
    "rt_camera_init", "rt_render", "rt_render_progressive", "rt_render_device", "rt_render_device_static", "rt_accumulate_device", "rt_accumulate_batch_device", "rt_resolve_device",
    "rt_multi_create", "rt_multi_destroy", "rt_multi_last_error", "rt_multi_device_count", "rt_multi_uses_rccl",
    "rt_multi_context", "rt_multi_set_scene", "rt_multi_set_option", "rt_multi_render", "rt_multi_render_device",
    "rt_multi_render_frames_device", "rt_multi_synchronize", "rt_multi_get_stats", "rt_multi_reset_stats",
    "rt_tile_block_bytes", "rt_pack_tiles_device", "rt_unpack_tiles_device", "rt_primary_hits", "rt_primary_hits_device", "rt_intersect", "rt_set_option", "rt_get_option", "rt_accel_selfcheck", "rt_get_stats", "rt_reset_stats", "rt_get_work_profile",
]


class Camera(C.Structure):
    _fields_ = [
        ("origin", C.c_float * 3),
        ("bottom_left", C.c_float * 3),
        ("horizontal", C.c_float * 3),
        ("vertical", C.c_float * 3),
    ]


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("spp", C.c_uint32),
        ("max_depth", C.c_int32),
        ("seed", C.c_uint64),
        ("sample_begin", C.c_uint32),
        ("sample_count", C.c_uint32),
        ("tile_first", C.c_uint32),
        ("tile_stride", C.c_uint32),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class Job(C.Structure):
    """rt_job: one rt_accumulate_device call as an entry of rt_accumulate_batch_device."""
    _fields_ = [
        ("cam", Camera),
        ("params", Params),
        ("xstart", C.c_int32),
        ("ystart", C.c_int32),
        ("xsize", C.c_int32),
        ("ysize", C.c_int32),
        ("d_sums", C.c_void_p),
    ]


MAX_JOBS = 8


class Stats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("wave_passes", C.c_uint64),
        ("last_kernel_ms", C.c_double),
        ("last_wall_ms", C.c_double),
        ("node_visits", C.c_uint64),
        ("exact_tests", C.c_uint64),
        ("root_evaluations", C.c_uint64),
    ]


class RtError(RuntimeError):
    pass


_lib = None


def load():
    """Loads the shared library (raises if it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    # A process must hold ONE HIP runtime.  PyTorch wheels bundle their own
    # libamdhip64; if this library pulled in /opt/rocm's copy first, torch would
    # later find no device.  So when torch is installed, let it load its runtime
    # first; librt_mi355x.so then binds to that same copy (equal SONAME).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} not built: run `make -C ray-tracer_amd/csrc` (or __graft_entry__.build())")
    L = C.CDLL(LIB_PATH)
    vp, i, u32 = C.c_void_p, C.c_int, C.c_uint32
    f3 = C.POINTER(C.c_float)
    L.rt_abi_version.restype = u32
    L.rt_status_string.argtypes = [i]
    L.rt_status_string.restype = C.c_char_p
    L.rt_last_error.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_create.argtypes = [i, C.POINTER(vp)]
    L.rt_destroy.argtypes = [vp]
    L.rt_set_scene.argtypes = [vp, vp, u32]
    L.rt_camera_init.argtypes = [f3, f3, f3, C.c_float, C.POINTER(Camera)]
    L.rt_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), i, i, i, i, vp]
    L.rt_render_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), i, i, i, i, vp, vp]
    L.rt_render_device_static.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), i, i, i, i, vp, vp]
    L.rt_accumulate_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), i, i, i, i, vp, vp]
    L.rt_render_progressive.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), i, i, i, i, vp, u32, PROGRESS_FN, vp]
    L.rt_accel_selfcheck.argtypes = [vp, u32, i, C.POINTER(u32 * 4)]
    L.rt_resolve_device.argtypes = [vp, vp, u32, u32, u32, u32, i, i, i, i, vp, vp]
    L.rt_tile_block_bytes.argtypes = [u32, u32, u32, u32]
    L.rt_tile_block_bytes.restype = C.c_uint64
    L.rt_pack_tiles_device.argtypes = [vp, vp, u32, u32, u32, u32, u32, u32, vp, vp]
    L.rt_unpack_tiles_device.argtypes = [vp, vp, u32, C.c_uint64, u32, u32, vp, vp]
    L.rt_multi_create.argtypes = [C.POINTER(C.c_int), u32, i, C.POINTER(vp)]
    L.rt_multi_destroy.argtypes = [vp]
    L.rt_multi_last_error.argtypes = [vp]
    L.rt_multi_last_error.restype = C.c_char_p
    L.rt_multi_device_count.argtypes = [vp]
    L.rt_multi_device_count.restype = u32
    L.rt_multi_uses_rccl.argtypes = [vp]
    L.rt_multi_context.argtypes = [vp, u32]
    L.rt_multi_context.restype = vp
    L.rt_multi_set_scene.argtypes = [vp, vp, u32]
    L.rt_multi_set_option.argtypes = [vp, i, C.c_int64]
    L.rt_multi_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), i, vp]
    L.rt_multi_render_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), i, vp]
    if hasattr(L, "rt_multi_render_frames_device"):
        L.rt_multi_render_frames_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), u32, i, C.POINTER(vp)]
    L.rt_multi_synchronize.argtypes = [vp]
    L.rt_multi_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rt_multi_reset_stats.argtypes = [vp]
    L.rt_primary_hits.argtypes = [vp, C.POINTER(Camera), u32, u32, vp]
    L.rt_primary_hits_device.argtypes = [vp, C.POINTER(Camera), u32, u32, vp, vp]
    L.rt_intersect.argtypes = [vp, vp, u32, vp]
    if hasattr(L, "rt_accumulate_batch_device"):  # absent from older builds kept for A/B timing (RTAMD_LIBRARY)
        L.rt_accumulate_batch_device.argtypes = [vp, C.POINTER(Job), u32, vp]
    L.rt_set_option.argtypes = [vp, i, C.c_int64]
    if hasattr(L, "rt_get_option"):
        L.rt_get_option.argtypes = [vp, i, C.POINTER(C.c_int64)]
    L.rt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rt_reset_stats.argtypes = [vp]
    if hasattr(L, "rt_get_work_profile"):  # absent from older builds kept for A/B timing (RTAMD_LIBRARY)
        L.rt_get_work_profile.argtypes = [vp, C.POINTER(C.c_uint64), u32]
    _lib = L
    return L


def make_camera(lookfrom, lookat, up, fov):
    cam = Camera()
    f = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    st = load().rt_camera_init(f(lookfrom), f(lookat), f(up), float(fov), C.byref(cam))
    if st != 0:
        raise RtError("rt_camera_init failed")
    return cam


def default_camera():
    """g_SceneCamera, Source/main.cpp:605-608."""
    return make_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0)


def default_scene():
    """std::vector<Sphere> Spheres, Source/main.cpp:520-526."""
    s = np.zeros(4, dtype=SPHERE_DTYPE)
    s[0] = ((-1.0, 0.0, -1.0), (0.8, 0.6, 0.2), 0.5, MAT_METAL, 0.65)
    s[1] = ((0.0, 0.0, -1.0), (255, 0, 0), 0.5, MAT_DIFFUSE, 0.0)
    s[2] = ((1.0, 0.0, -1.0), (0.8, 0.8, 0.8), 0.5, MAT_METAL, 0.0)
    s[3] = ((0.0, -100.5, -1.0), (255, 215, 10), 100.0, MAT_DIFFUSE, 0.0)
    return s


FLAG_DIELECTRIC, FLAG_GAMMA2, FLAG_FLOAT_SHADING = 1, 2, 4
PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32)


def source_fingerprint():
    """sha256 over the library's sources (ray-tracer_amd/csrc, include/rt_mi355x.h), in sorted order: what
    ties a committed hardware-counter profile to the build it was taken from."""
    import glob
    import hashlib

    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    files = sorted(glob.glob(os.path.join(root, "ray-tracer_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(root, "ray-tracer_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "ray-tracer_amd", "csrc", "*.cpp")) +
                   [os.path.join(root, "ray-tracer_amd", "csrc", "Makefile"), os.path.join(root, "include", "rt_mi355x.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def params(width, height, spp, max_depth, seed=0, sample_begin=0, sample_count=0, tile_first=0, tile_stride=0,
           flags=0):
    return Params(width, height, spp, max_depth, seed, sample_begin, sample_count, tile_first, tile_stride, flags, 0)


def accel_selfcheck(spheres, leaf_spheres=4):
    """Host-only build + structural check of the hierarchy; returns (status, nodes, levels, large, records)."""
    spheres = np.ascontiguousarray(spheres)
    info = (C.c_uint32 * 4)()
    st = load().rt_accel_selfcheck(spheres.ctypes.data if len(spheres) else None, len(spheres), leaf_spheres,
                                   C.byref(info))
    return (st,) + tuple(info)


class Context:
    """One rt_context on one device."""

    def __init__(self, device=0):
        self._L = load()
        self._h = C.c_void_p()
        st = self._L.rt_create(device, C.byref(self._h))
        if st != 0:
            raise RtError(f"rt_create({device}): {self._L.rt_status_string(st).decode()}")

    def close(self):
        if self._h:
            self._L.rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise RtError(f"{self._L.rt_status_string(st).decode()}: {self._L.rt_last_error(self._h).decode()}")

    def call(self, name, *args):
        """Raw status of an entry point (for error-path tests)."""
        return getattr(self._L, name)(self._h, *args)

    def set_scene(self, spheres):
        spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        self._scene = spheres
        self._check(self._L.rt_set_scene(self._h, spheres.ctypes.data if len(spheres) else None, len(spheres)))

    def render(self, cam, p, tile=None, rgb=None, fill=255):
        if tile is None:
            tile = (0, 0, p.width, p.height)
        if rgb is None:
            rgb = np.full((p.height, p.width, 3), fill, dtype=np.uint8)
        self._check(self._L.rt_render(self._h, C.byref(cam), C.byref(p), *tile, rgb.ctypes.data))
        return rgb

    def render_device(self, cam, p, d_rgb, tile=None, stream=None):
        if tile is None:
            tile = (0, 0, p.width, p.height)
        self._check(self._L.rt_render_device(self._h, C.byref(cam), C.byref(p), *tile, d_rgb, stream))

    def render_device_static(self, cam, p, d_rgb, tile=None, stream=None):
        if tile is None:
            tile = (0, 0, p.width, p.height)
        self._check(self._L.rt_render_device_static(self._h, C.byref(cam), C.byref(p), *tile, d_rgb, stream))

    def accumulate_device(self, cam, p, d_sums, tile=None, stream=None):
        if tile is None:
            tile = (0, 0, p.width, p.height)
        self._check(self._L.rt_accumulate_device(self._h, C.byref(cam), C.byref(p), *tile, d_sums, stream))

    def accumulate_batch_device(self, jobs, stream=None):
        """jobs: sequence of (cam, params, d_sums) or (cam, params, d_sums, tile) -- one launch for all of them."""
        arr = (Job * len(jobs))()
        for k, job in enumerate(jobs):
            cam, p, d_sums = job[:3]
            tile = job[3] if len(job) > 3 and job[3] is not None else (0, 0, p.width, p.height)
            arr[k].cam = cam
            arr[k].params = p
            arr[k].xstart, arr[k].ystart, arr[k].xsize, arr[k].ysize = tile
            arr[k].d_sums = d_sums
        self._check(self._L.rt_accumulate_batch_device(self._h, arr, len(jobs), stream))

    def resolve_device(self, d_sums, width, height, spp_div, d_rgb, tile=None, stream=None, flags=0):
        if tile is None:
            tile = (0, 0, width, height)
        self._check(self._L.rt_resolve_device(self._h, d_sums, width, height, spp_div, flags, *tile, d_rgb, stream))

    def pack_tiles_device(self, d_sums, width, height, spp_div, tile_first, tile_stride, d_tiles, stream=None, flags=0):
        self._check(self._L.rt_pack_tiles_device(self._h, d_sums, width, height, spp_div, flags, tile_first,
                                                 tile_stride, d_tiles, stream))

    def unpack_tiles_device(self, d_blocks, n_blocks, block_bytes, width, height, d_rgb, stream=None):
        self._check(self._L.rt_unpack_tiles_device(self._h, d_blocks, n_blocks, block_bytes, width, height, d_rgb, stream))

    def render_progressive(self, cam, p, samples_per_pass, on_pass=None, tile=None, rgb=None, fill=255):
        """on_pass(samples_done, samples_total, rgb) -> truthy to stop."""
        if tile is None:
            tile = (0, 0, p.width, p.height)
        if rgb is None:
            rgb = np.full((p.height, p.width, 3), fill, dtype=np.uint8)
        cb = PROGRESS_FN(lambda user, done, total: 1 if (on_pass and on_pass(done, total, rgb)) else 0)
        self._check(self._L.rt_render_progressive(self._h, C.byref(cam), C.byref(p), *tile, rgb.ctypes.data,
                                                  samples_per_pass, cb, None))
        return rgb

    def primary_hits(self, cam, width, height):
        out = np.empty((height, width), dtype=np.int32)
        self._check(self._L.rt_primary_hits(self._h, C.byref(cam), width, height, out.ctypes.data))
        return out

    def primary_hits_device(self, cam, width, height, d_hits, stream=None):
        self._check(self._L.rt_primary_hits_device(self._h, C.byref(cam), width, height, d_hits, stream))

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        out = np.zeros(len(rays), dtype=np.dtype([("index", "<i4"), ("t", "<f4"), ("point", "<f4", 3), ("normal", "<f4", 3)]))
        self._check(self._L.rt_intersect(self._h, rays.ctypes.data, len(rays), out.ctypes.data))
        return out

    def set_option(self, option, value):
        self._check(self._L.rt_set_option(self._h, option, value))

    def get_option(self, option):
        v = C.c_int64(0)
        self._check(self._L.rt_get_option(self._h, option, C.byref(v)))
        return v.value

    def stats(self):
        s = Stats()
        self._check(self._L.rt_get_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        self._check(self._L.rt_reset_stats(self._h))

    SECTIONS = ["pass", "refill", "walk_begin", "node", "leaf_group", "root", "point_first", "point_help", "scatter",
                "replay_level", "walk_setup", "shade"]

    def work_profile(self):
        """{section: (wave executions, lane executions)} of the counting build since the last reset."""
        out = (C.c_uint64 * 56)()
        self._check(self._L.rt_get_work_profile(self._h, out, 28))
        prof = {name: (int(out[2 * i]), int(out[2 * i + 1])) for i, name in enumerate(self.SECTIONS)}
        prof["cycles"] = dict(zip(["life", "fetch", "refill", "scan", "points", "shade", "flush"], [int(v) for v in out[24:31]]))
        prof["cycles"]["waves"] = int(out[32])
        # per wave: start, queue found empty, end (cycles of the chip's clock after the first wave's start)
        low, waves = (1 << 40) - 1, max(int(out[32]), 1)
        first = low - int(out[39])
        prof["timeline"] = {name: {"mean": int(out[33 + i]) / waves - first, "first": low - int(out[39 + i]) - first,
                                   "last": int(out[36 + i]) - first}
                            for i, name in enumerate(["start", "queue_empty", "end"])}
        return prof


def tile_block_bytes(width, height, tile_first, tile_stride):
    """Bytes of the block rt_pack_tiles_device writes for the tiles first, first + stride, ..."""
    return int(load().rt_tile_block_bytes(width, height, tile_first, tile_stride))


PARTITION_SAMPLES, PARTITION_TILES = 0, 1
EXCHANGE_AUTO, EXCHANGE_RCCL, EXCHANGE_PEER_COPY = 0, 1, 2


class Multi:
    """One rt_multi: several devices (or several ranks on one device) driven from this process."""

    def __init__(self, device_ids, exchange=EXCHANGE_AUTO):
        self._L = load()
        self._h = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*device_ids)
        st = self._L.rt_multi_create(ids, len(device_ids), exchange, C.byref(self._h))
        if st != 0:
            raise RtError(f"rt_multi_create({list(device_ids)}): {self._L.rt_status_string(st).decode()}")

    def close(self):
        if self._h:
            self._L.rt_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise RtError(f"{self._L.rt_status_string(st).decode()}: {self._L.rt_multi_last_error(self._h).decode()}")

    @property
    def uses_rccl(self):
        return bool(self._L.rt_multi_uses_rccl(self._h))

    def set_scene(self, spheres):
        spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        self._check(self._L.rt_multi_set_scene(self._h, spheres.ctypes.data if len(spheres) else None, len(spheres)))

    def set_option(self, option, value):
        self._check(self._L.rt_multi_set_option(self._h, option, value))

    def render(self, cam, p, partition):
        rgb = np.zeros((p.height, p.width, 3), dtype=np.uint8)
        self._check(self._L.rt_multi_render(self._h, C.byref(cam), C.byref(p), partition, rgb.ctypes.data))
        return rgb

    def render_device(self, cam, p, partition, d_rgb):
        self._check(self._L.rt_multi_render_device(self._h, C.byref(cam), C.byref(p), partition, d_rgb))

    def render_frames_device(self, frames, partition):
        """frames: [(cam, params, d_rgb address), ...], at most MAX_JOBS -- one trace launch per device for all of them."""
        n = len(frames)
        cams = (Camera * n)(*[f[0] for f in frames])
        ps = (Params * n)(*[f[1] for f in frames])
        out = (C.c_void_p * n)(*[f[2] for f in frames])
        self._check(self._L.rt_multi_render_frames_device(self._h, cams, ps, n, partition, out))

    def synchronize(self):
        self._check(self._L.rt_multi_synchronize(self._h))

    def stats(self):
        s = Stats()
        self._check(self._L.rt_multi_get_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        self._check(self._L.rt_multi_reset_stats(self._h))
