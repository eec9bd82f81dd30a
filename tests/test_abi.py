"""The C-ABI library loads and exports every symbol include/rt_mi355x.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np

import rtamd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert declared_symbols() == sorted(rtamd.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = rtamd.load()
    for name in declared_symbols():
        assert getattr(lib, name) is not None, name
    assert lib.rt_abi_version() == 2


def test_struct_layouts_match_header():
    assert rtamd.SPHERE_DTYPE.itemsize == 36  # class Sphere, Source/main.cpp:362-391
    assert C.sizeof(rtamd.Camera) == 48
    assert C.sizeof(rtamd.Params) == 48
    assert C.sizeof(rtamd.Stats) == 64


def test_status_strings():
    lib = rtamd.load()
    assert lib.rt_status_string(0) == b"ok"
    assert lib.rt_status_string(-4) == b"no scene set"


def test_camera_init_matches_oracle_bit_for_bit():
    """Host-only entry point: Camera::Camera, Source/main.cpp:420-438."""
    import oracle_lib as O

    rs = np.random.RandomState(1)
    cases = [((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0), ((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0),
             ((-2, 2, 1), (0, 0, -1), (0, 1, 0), 90.0)]
    for _ in range(50):
        cases.append((rs.uniform(-5, 5, 3), rs.uniform(-5, 5, 3), rs.normal(size=3), float(rs.uniform(5, 150))))
    for c in cases:
        a, b = rtamd.make_camera(*c), O.camera(*c)
        assert bytes(a) == bytes(b)


def test_null_arguments_are_rejected_without_a_device():
    lib = rtamd.load()
    assert lib.rt_camera_init(None, None, None, 1.0, None) == -1
    assert lib.rt_create(0, None) == -1
    assert lib.rt_destroy(None) == -1


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 and as C++11, and a program in plain
    C must link against the library (no C++ or torch types anywhere in the signatures)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "rt_mi355x.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c++", hdr])
    exe = str(tmp_path / "render_c_abi")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "render_c_abi.c"), "-L", os.path.join(root, "ray-tracer_amd"),
                           "-lrt_mi355x", "-o", exe])
    assert os.path.exists(exe)
