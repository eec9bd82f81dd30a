"""ctypes binding of oracle/liboracle.so (the CPU checker; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
REF_HARNESS = os.path.join(ORACLE_DIR, "_ref", "ref_harness")

SPHERE_DTYPE = np.dtype(
    [("center", "<f4", 3), ("color", "<f4", 3), ("radius", "<f4"), ("material", "<i4"), ("fuzz", "<f4")]
)
assert SPHERE_DTYPE.itemsize == 36

MAT_GLASS, MAT_DIFFUSE, MAT_METAL, MAT_FUZZYMETAL = 0, 1, 2, 3
RNG_MT19937, RNG_COUNTER = 0, 1


class Camera(C.Structure):
    _fields_ = [
        ("origin", C.c_float * 3),
        ("bottom_left", C.c_float * 3),
        ("horizontal", C.c_float * 3),
        ("vertical", C.c_float * 3),
    ]


class MT(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


def _build():
    if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
        os.path.join(ORACLE_DIR, "rt_oracle.c")
    ):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _build()
        L = C.CDLL(LIB_PATH)
        f32p = C.POINTER(C.c_float)
        L.orc_mt_seed.argtypes = [C.POINTER(MT), C.c_uint32]
        L.orc_mt_next.argtypes = [C.POINTER(MT)]
        L.orc_mt_next.restype = C.c_uint32
        L.orc_u32_to_unit_float.argtypes = [C.c_uint32]
        L.orc_u32_to_unit_float.restype = C.c_float
        L.orc_random_floats.argtypes = [C.c_uint32, C.c_int, f32p]
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_camera_init.argtypes = [f32p, f32p, f32p, C.c_float, C.POINTER(Camera)]
        L.orc_get_ray.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, f32p, f32p]
        L.orc_intersect.argtypes = [C.c_void_p, C.c_uint32, f32p, f32p, C.c_float, C.c_float, f32p, f32p, f32p]
        L.orc_intersect.restype = C.c_int32
        L.orc_sky.argtypes = [f32p, C.POINTER(C.c_uint8)]
        L.orc_to_byte.argtypes = [C.c_float]
        L.orc_to_byte.restype = C.c_uint8
        L.orc_trace_tile.argtypes = [
            C.c_void_p, C.c_uint32, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
            C.c_uint32, C.c_int32, C.c_int, C.c_uint64, C.POINTER(MT), C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
        ]
        L.orc_trace_tile_ext.argtypes = [
            C.c_void_p, C.c_uint32, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
            C.c_uint32, C.c_int32, C.c_int, C.c_uint64, C.POINTER(MT), C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
        ]
        L.orc_gamma2.argtypes = [C.c_int32]
        L.orc_gamma2.restype = C.c_uint8
        L.orc_sample_color.argtypes = [
            C.c_void_p, C.c_uint32, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_int32, C.c_uint64,
            C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64),
        ]
        L.orc_primary_hits.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_trace_scene_threads.argtypes = [
            C.c_void_p, C.c_uint32, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32,
            C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint64),
        ]
        L.orc_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_fnv1a64.restype = C.c_uint64
        _lib = L
    return _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def camera(lookfrom, lookat, up, fov):
    cam = Camera()
    lib().orc_camera_init(_f3(lookfrom), _f3(lookat), _f3(up), float(fov), C.byref(cam))
    return cam


def default_camera():
    """g_SceneCamera, Source/main.cpp:605-608."""
    return camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0)


def default_scene():
    """std::vector<Sphere> Spheres, Source/main.cpp:520-526."""
    s = np.zeros(4, dtype=SPHERE_DTYPE)
    s[0] = ((-1.0, 0.0, -1.0), (0.8, 0.6, 0.2), 0.5, MAT_METAL, 0.65)
    s[1] = ((0.0, 0.0, -1.0), (255, 0, 0), 0.5, MAT_DIFFUSE, 0.0)
    s[2] = ((1.0, 0.0, -1.0), (0.8, 0.8, 0.8), 0.5, MAT_METAL, 0.0)
    s[3] = ((0.0, -100.5, -1.0), (255, 215, 10), 100.0, MAT_DIFFUSE, 0.0)
    return s


def fnv1a64(arr):
    a = np.ascontiguousarray(arr)
    return int(lib().orc_fnv1a64(a.ctypes.data, a.nbytes))


def new_mt(seed=5489):
    g = MT()
    lib().orc_mt_seed(C.byref(g), seed)
    return g


FLAG_DIELECTRIC, FLAG_GAMMA2, FLAG_FLOAT_SHADING = 1, 2, 4


def trace_tile(spheres, cam, width, height, spp, depth, rng_mode, seed=0, mt=None, tile=None,
               s0=0, s1=None, rgb=None, want_sums=False, fill=255, flags=0):
    """TraceThreadFunction on `tile` = (xstart, ystart, xsize, ysize); returns (rgb, sums, rays)."""
    spheres = np.ascontiguousarray(spheres)
    if tile is None:
        tile = (0, 0, width, height)
    if s1 is None:
        s1 = spp
    if rgb is None:
        rgb = np.full((height, width, 3), fill, dtype=np.uint8)
    sums = np.zeros((height, width, 3), dtype=np.uint32) if want_sums else None
    rays = C.c_uint64(0)
    if rng_mode == RNG_MT19937 and mt is None:
        mt = new_mt()
    lib().orc_trace_tile_ext(
        spheres.ctypes.data, len(spheres), C.byref(cam), width, height, s0, s1, spp, depth, rng_mode, seed,
        C.byref(mt) if mt is not None else None, tile[0], tile[1], tile[2], tile[3], flags,
        rgb.ctypes.data, sums.ctypes.data if sums is not None else None, C.byref(rays),
    )
    return rgb, sums, rays.value


def primary_hits(spheres, cam, width, height):
    spheres = np.ascontiguousarray(spheres)
    out = np.empty((height, width), dtype=np.int32)
    lib().orc_primary_hits(spheres.ctypes.data, len(spheres), C.byref(cam), width, height, out.ctypes.data)
    return out


def sample_color(spheres, cam, width, height, depth, seed, x, y, sample):
    spheres = np.ascontiguousarray(spheres)
    rgb = (C.c_uint8 * 3)()
    rays = C.c_uint64(0)
    lib().orc_sample_color(spheres.ctypes.data, len(spheres), C.byref(cam), width, height, depth, seed, x, y,
                           sample, rgb, C.byref(rays))
    return tuple(rgb), rays.value


def have_ref():
    return os.path.exists(REF_HARNESS)


def run_ref(*args):
    """Run the reference harness (fresh process = fresh RNG); returns stdout."""
    return subprocess.check_output([REF_HARNESS] + [str(a) for a in args], text=True)
