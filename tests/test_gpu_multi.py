"""rt_multi_*: one frame over several devices from one host process (the reference's TraceScene,
Source/main.cpp:647-659, with devices in place of its four threads).  A one-GPU machine runs it with
one device over RCCL -- the RCCL branch is the same code for one rank as for several, so ncclReduce
(in place), ncclSend / ncclRecv (the rank sending its tile block to itself) and the group calls all
execute, with a single rank -- and with several logical ranks sharing the device (peer-copy exchange):
every split, buffer rotation and exchange step.  What a one-GPU machine cannot run is RCCL (or peer
copies) between DISTINCT devices: test_multi_over_distinct_devices does, and skips where fewer than two GPUs
exist.
The assembled frame must be the single-context frame, bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_lib as O
import rtamd
import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
C3_CAM = scenes.RANDOM_SPHERES_CAMERA


def c3_scene():
    return np.fromfile(os.path.join(G, "scene_c3_500.bin"), dtype=rtamd.SPHERE_DTYPE)


@pytest.mark.parametrize("ids,exchange", [([0], rtamd.EXCHANGE_RCCL), ([0], rtamd.EXCHANGE_PEER_COPY),
                                          ([0, 0], rtamd.EXCHANGE_AUTO), ([0, 0, 0], rtamd.EXCHANGE_PEER_COPY),
                                          ([0] * 8, rtamd.EXCHANGE_AUTO)])
@pytest.mark.parametrize("partition", [rtamd.PARTITION_SAMPLES, rtamd.PARTITION_TILES])
def test_multi_frame_equals_oracle_frame(ids, exchange, partition):
    sc = c3_scene()
    W, H, SPP, D = 100, 61, 6, 8  # ragged tiles; 6 samples over 8 ranks leaves two ranks without any
    want, _, rays = O.trace_tile(sc, O.camera(*C3_CAM), W, H, SPP, D, O.RNG_COUNTER, seed=4)
    m = rtamd.Multi(ids, exchange)
    try:
        assert m.uses_rccl == (exchange == rtamd.EXCHANGE_RCCL)  # duplicate ids and PEER_COPY never use RCCL
        m.set_scene(sc)
        cam, p = rtamd.make_camera(*C3_CAM), rtamd.params(W, H, SPP, D, seed=4)
        for _ in range(3):  # both buffer slots, and the first one again
            m.reset_stats()
            got = m.render(cam, p, partition)
            assert np.array_equal(got, want)
            assert m.stats().rays == rays
    finally:
        m.close()


def test_multi_over_distinct_devices():
    """Two or more GPUs: RCCL and peer copies between distinct devices, both partitions, against the
    oracle.  Skipped on a one-GPU machine (no multi-GPU machine has run it yet)."""
    import torch

    n = min(torch.cuda.device_count(), 8)
    if n < 2:
        pytest.skip("needs at least two GPUs")
    sc = c3_scene()
    W, H, SPP, D = 200, 120, 2 * n + 1, 8
    want, _, rays = O.trace_tile(sc, O.camera(*C3_CAM), W, H, SPP, D, O.RNG_COUNTER, seed=6)
    cam, p = rtamd.make_camera(*C3_CAM), rtamd.params(W, H, SPP, D, seed=6)
    for exchange in (rtamd.EXCHANGE_RCCL, rtamd.EXCHANGE_PEER_COPY, rtamd.EXCHANGE_AUTO):
        m = rtamd.Multi(list(range(n)), exchange)
        try:
            m.set_scene(sc)
            for partition in (rtamd.PARTITION_SAMPLES, rtamd.PARTITION_TILES):
                for _ in range(3):
                    m.reset_stats()
                    assert np.array_equal(m.render(cam, p, partition), want), (exchange, partition)
                    assert m.stats().rays == rays
        finally:
            m.close()


def test_multi_shares_one_hierarchy_build_and_recovers_from_a_failed_frame():
    """rt_multi_set_scene builds the hierarchy once and hands it to the other ranks' contexts (same
    frames as with a build per rank); a frame that fails half way (a rank's call is refused) leaves
    the object usable: the next frame is correct."""
    sc = scenes.random_spheres(3000, grid_half=27)
    cam = rtamd.make_camera(*C3_CAM)
    p = rtamd.params(160, 90, 4, 8, seed=2)
    ctx = rtamd.Context(0)
    ctx.set_scene(sc)
    want = ctx.render(cam, p)
    ctx.close()
    m = rtamd.Multi([0, 0, 0], rtamd.EXCHANGE_PEER_COPY)
    try:
        m.set_scene(sc)
        assert np.array_equal(m.render(cam, p, rtamd.PARTITION_TILES), want)
        L = rtamd.load()
        buf = np.zeros((90, 160, 3), dtype=np.uint8)
        bad = rtamd.params(160, 90, 4, 5000, seed=2)  # max_depth above the limit: every rank's trace call is refused
        assert L.rt_multi_render(m._h, C.byref(cam), C.byref(bad), rtamd.PARTITION_SAMPLES, buf.ctypes.data) != 0
        assert b"max_depth" in L.rt_multi_last_error(m._h)
        for partition in (rtamd.PARTITION_SAMPLES, rtamd.PARTITION_TILES):
            assert np.array_equal(m.render(cam, p, partition), want)
    finally:
        m.close()


def test_multi_pipelined_frames_on_device(tmp_path):
    """Frames enqueued back to back without waiting: slots rotate, each frame lands in its buffer."""
    import torch

    sc = c3_scene()
    W, H, D = 256, 144, 8
    cam = rtamd.make_camera(*C3_CAM)
    ctx = rtamd.Context(0)
    ctx.set_scene(sc)
    frames = [rtamd.params(W, H, 8, D, seed=s) for s in (1, 2, 3, 4, 5)]
    want = [ctx.render(cam, p) for p in frames]
    ctx.close()
    for partition in (rtamd.PARTITION_SAMPLES, rtamd.PARTITION_TILES):
        m = rtamd.Multi([0, 0, 0, 0])
        try:
            m.set_scene(sc)
            outs = [torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda") for _ in frames]
            torch.cuda.synchronize()
            for p, o in zip(frames, outs):
                m.render_device(cam, p, partition, o.data_ptr())
            m.synchronize()
            for o, w in zip(outs, want):
                assert np.array_equal(o.cpu().numpy().reshape(H, W, 3), w)
            # a larger frame afterwards: the buffers grow
            big = rtamd.params(640, 360, 4, D, seed=9)
            got = m.render(cam, big, partition)
        finally:
            m.close()
        ctx = rtamd.Context(0)
        ctx.set_scene(sc)
        assert np.array_equal(got, ctx.render(cam, big))
        ctx.close()


@pytest.mark.parametrize("ids,exchange", [([0], rtamd.EXCHANGE_RCCL), ([0, 0, 0], rtamd.EXCHANGE_PEER_COPY),
                                          ([0] * 8, rtamd.EXCHANGE_AUTO)])
def test_multi_several_frames_per_call(ids, exchange):
    """rt_multi_render_frames_device: every rank traces its shares of a call's frames in one launch; the
    pictures are those of single-frame calls, for frames of different cameras, sizes and sample counts, calls
    of different lengths one after the other (the buffer sets rotate and grow), and single frames in between."""
    import torch

    sc = c3_scene()
    ctx = rtamd.Context(0)
    ctx.set_scene(sc)
    cams = [rtamd.make_camera((13 - k, 2 + 0.25 * k, 3 + 0.5 * k), (0, 0, 0), (0, 1, 0), 20 + k) for k in range(8)]
    shapes = [(160, 90, 5), (96, 54, 3), (200, 120, 2), (160, 90, 9), (64, 40, 1), (160, 90, 5), (100, 61, 6), (128, 72, 4)]
    frames = [(cams[k], rtamd.params(w, h, spp, 8, seed=12)) for k, (w, h, spp) in enumerate(shapes)]
    want = [ctx.render(c, p) for c, p in frames]
    rays = []
    for c, p in frames:
        ctx.reset_stats()
        ctx.render(c, p)
        rays.append(ctx.stats().rays)
    ctx.close()
    m = rtamd.Multi(ids, exchange)
    try:
        m.set_scene(sc)
        for partition in (rtamd.PARTITION_SAMPLES, rtamd.PARTITION_TILES):
            for cut in ([8], [3, 1, 4], [1, 2, 2, 1, 2], [5, 3]):
                outs = [torch.zeros(p.height * p.width * 3, dtype=torch.uint8, device="cuda") for _, p in frames]
                torch.cuda.synchronize()
                m.reset_stats()
                at = 0
                for n in cut:
                    group = [(frames[k][0], frames[k][1], outs[k].data_ptr()) for k in range(at, at + n)]
                    if n == 1:
                        m.render_device(group[0][0], group[0][1], partition, group[0][2])
                    else:
                        m.render_frames_device(group, partition)
                    at += n
                m.synchronize()
                for k, (o, w) in enumerate(zip(outs, want)):
                    p = frames[k][1]
                    assert np.array_equal(o.cpu().numpy().reshape(p.height, p.width, 3), w), (partition, cut, k)
                assert m.stats().rays == sum(rays)
        # the frames of a call share seed and flags; a refused call leaves the object usable
        bad = [(cams[0], rtamd.params(64, 40, 2, 8, seed=1), outs[0].data_ptr()),
               (cams[1], rtamd.params(64, 40, 2, 8, seed=2), outs[1].data_ptr())]
        with pytest.raises(rtamd.RtError, match="share seed and flags"):
            m.render_frames_device(bad, rtamd.PARTITION_TILES)
        with pytest.raises(rtamd.RtError, match="RT_MAX_JOBS"):
            m.render_frames_device([(cams[0], frames[0][1], outs[0].data_ptr())] * 9, rtamd.PARTITION_TILES)
        assert np.array_equal(m.render(frames[0][0], frames[0][1], rtamd.PARTITION_SAMPLES), want[0])
    finally:
        m.close()


def test_multi_full_size_frame_and_bad_arguments():
    sc = c3_scene()
    cam = rtamd.make_camera(*C3_CAM)
    p = rtamd.params(1920, 1080, 16, 8, seed=1)
    ctx = rtamd.Context(0)
    ctx.set_scene(sc)
    want = ctx.render(cam, p)
    ctx.close()
    m = rtamd.Multi([0, 0])
    try:
        m.set_scene(sc)
        assert np.array_equal(m.render(cam, p, rtamd.PARTITION_TILES), want)
        assert np.array_equal(m.render(cam, p, rtamd.PARTITION_SAMPLES), want)
        L = rtamd.load()
        buf = np.zeros((1080, 1920, 3), dtype=np.uint8)
        sliced = rtamd.params(1920, 1080, 16, 8, seed=1, sample_begin=0, sample_count=4)
        assert L.rt_multi_render(m._h, C.byref(cam), C.byref(sliced), 0, buf.ctypes.data) == -1
        assert b"splits the frame itself" in L.rt_multi_last_error(m._h)
        assert L.rt_multi_render(m._h, C.byref(cam), C.byref(p), 7, buf.ctypes.data) == -1
        assert L.rt_multi_render(m._h, None, C.byref(p), 0, buf.ctypes.data) == -1
    finally:
        m.close()
    h = C.c_void_p()
    ids = (C.c_int * 2)(0, 0)
    assert rtamd.load().rt_multi_create(ids, 2, rtamd.EXCHANGE_RCCL, C.byref(h)) == -1  # RCCL: one rank per device
    assert rtamd.load().rt_multi_create(ids, 0, 0, C.byref(h)) == -1
    assert rtamd.load().rt_multi_create((C.c_int * 1)(99), 1, 0, C.byref(h)) != 0


def test_cpp_hosts_drive_several_ranks(tmp_path):
    """examples/multi_gpu_rccl (plain C++ over rt_multi_*) and rt_render_cli --devices: same picture
    for one rank over RCCL, and for three ranks sharing the device, either partition."""
    exe = os.path.join(ROOT, "examples", "multi_gpu_rccl")
    assert os.path.exists(exe), "run __graft_entry__.build() (make -C examples)"
    lines = []
    for args in (("1", "samples", "rccl"), ("1", "tiles", "rccl"), ("3", "samples", "peer"), ("3", "tiles", "auto")):
        out = subprocess.run([exe, *args], check=True, capture_output=True, text=True, timeout=120).stdout
        lines.append(out)
    fnv = {line.split("frame fnv ")[1].strip() for line in lines}
    assert len(fnv) == 1, lines
    assert "RCCL" in lines[0] and "peer copies" in lines[2]
    cli = os.path.join(ROOT, "ray-tracer_amd", "rt_render_cli")
    one, many = str(tmp_path / "one.ppm"), str(tmp_path / "many.ppm")
    base = ["--width", "160", "--height", "90", "--spp", "5", "--seed", "3"]
    subprocess.run([cli, *base, "--out", one], check=True, capture_output=True, timeout=120)
    text = subprocess.run([cli, *base, "--devices", "0,0,0", "--partition", "samples", "--out", many], check=True,
                          capture_output=True, text=True, timeout=120).stdout
    assert "3 rank(s)" in text and open(one, "rb").read() == open(many, "rb").read()
