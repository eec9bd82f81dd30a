"""The viewer-free hosts on the GPU: rt_render_cli (the reference's main() without the window,
Source/main.cpp:668-679, with a PPM writer) and the plain-C example.  PutPixel's layout has row 0
at the BOTTOM of the picture (Source/main.cpp:273-281 with the quad's UVs, :220-225), a PPM file
its first row at the top: the file must be the oracle's frame flipped."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_lib as O
import rtamd
import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "ray-tracer_amd", "rt_render_cli")
G = os.path.join(ROOT, "tests", "golden")


def read_ppm(path):
    data = open(path, "rb").read()
    # P6\n<width> <height>\n255\n<bytes>
    head = data.split(b"\n", 3)
    assert head[0] == b"P6" and head[2] == b"255"
    w, h = map(int, head[1].split())
    px = np.frombuffer(head[3], dtype=np.uint8)
    assert px.size == w * h * 3
    return px.reshape(h, w, 3)


def run_cli(*args):
    return subprocess.run([CLI] + [str(a) for a in args], check=True, capture_output=True, text=True, timeout=120).stdout


def test_cli_writes_the_oracles_frame_top_row_first(tmp_path):
    out = str(tmp_path / "x.ppm")
    text = run_cli("--width", 96, "--height", 54, "--spp", 4, "--seed", 1, "--out", out)
    want, _, rays = O.trace_tile(O.default_scene(), O.default_camera(), 96, 54, 4, 10, O.RNG_COUNTER, seed=1)
    got = read_ppm(out)
    assert np.array_equal(np.flipud(got), want)
    assert not np.array_equal(got, want)  # the flip is not a no-op on this picture
    assert f"{rays} rays" in text


def test_cli_scene_file_camera_and_progressive_passes(tmp_path):
    out = str(tmp_path / "p.ppm")
    cam = scenes.RANDOM_SPHERES_CAMERA
    args = ["--width", 80, "--height", 45, "--spp", 7, "--depth", 8, "--seed", 5, "--scene",
            os.path.join(G, "scene_c3_500.bin"), "--camera", *cam[0], *cam[1], *cam[2], cam[3], "--out", out]
    sc = np.fromfile(os.path.join(G, "scene_c3_500.bin"), dtype=rtamd.SPHERE_DTYPE)
    want, _, _ = O.trace_tile(sc, O.camera(*cam), 80, 45, 7, 8, O.RNG_COUNTER, seed=5)
    run_cli(*args)
    assert np.array_equal(np.flipud(read_ppm(out)), want)
    os.remove(out)
    text = run_cli(*args, "--progressive", 3)  # passes of 3, 3 and 1 samples; the file is rewritten after each
    assert "3 / 7 samples" in text and "6 / 7 samples" in text and "7 / 7 samples" in text
    assert np.array_equal(np.flipud(read_ppm(out)), want)


def test_plain_c_example_renders_the_shipped_frame(tmp_path):
    """examples/render_c_abi.c: the reference's scene, camera and constants through the ABI from C."""
    exe = os.path.join(ROOT, "examples", "render_c_abi")
    assert os.path.exists(exe), "run __graft_entry__.build() (make -C examples)"
    out = str(tmp_path / "c.ppm")
    subprocess.run([exe, out], check=True, capture_output=True, timeout=120)
    ctx = rtamd.Context(0)
    try:
        ctx.set_scene(rtamd.default_scene())
        want = ctx.render(rtamd.default_camera(), rtamd.params(1024, 576, 100, 10, seed=0))
    finally:
        ctx.close()
    assert np.array_equal(np.flipud(read_ppm(out)), want)
