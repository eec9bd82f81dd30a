"""rt_render_device_static -- the entry point that allocates nothing -- against rt_render_device,
through the C ABI: same bytes on two calls in a row, the second on a smaller frame (the static
array keeps the first call's sums), no change in the device's free memory, sizes beyond the
static scratch refused."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import rtamd
import scenes

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_static_entry_point_matches_render_device_twice_and_allocates_nothing():
    import torch

    ctx = rtamd.Context(0)
    try:
        ctx.set_scene(np.fromfile(os.path.join(G, "scene_c3_500.bin"), dtype=rtamd.SPHERE_DTYPE))
        cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
        calls = [
            (rtamd.params(640, 360, 8, 8, seed=31), None),
            (rtamd.params(200, 120, 3, 8, seed=32), None),            # smaller: stale sums around and under it
            (rtamd.params(200, 120, 5, 6, seed=33), (40, 16, 96, 70)),  # a tile of the same frame
        ]
        want = []
        for p, tile in calls:
            d = torch.full((p.height * p.width * 3,), 255, dtype=torch.uint8, device="cuda")
            ctx.render_device(cam, p, d.data_ptr(), tile=tile)
            torch.cuda.synchronize()
            want.append(d.cpu().numpy())
        outs = [torch.full((p.height * p.width * 3,), 255, dtype=torch.uint8, device="cuda") for p, _ in calls]
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for rounds in range(2):
            for (p, tile), d in zip(calls, outs):
                d.fill_(255)
                ctx.render_device_static(cam, p, d.data_ptr(), tile=tile)
            torch.cuda.synchronize()
            for d, w in zip(outs, want):
                assert np.array_equal(d.cpu().numpy(), w)
        assert torch.cuda.mem_get_info()[0] == free_before  # nothing was allocated or freed on the device
        # on another stream, next to rt_render_device on the context's shared ordering
        s = torch.cuda.Stream()
        p, tile = calls[0]
        outs[0].fill_(255)
        ctx.render_device_static(cam, p, outs[0].data_ptr(), stream=s.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(outs[0].cpu().numpy(), want[0])
        # the list scans use the same static sums and queue
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_MATRIX)
        outs[1].fill_(255)
        ctx.render_device_static(cam, calls[1][0], outs[1].data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(outs[1].cpu().numpy(), want[1])
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)
        # beyond the static scratch: refused, with a message that names the limit
        big = rtamd.params(2048, 1025, 1, 1)
        d = torch.zeros(16, dtype=torch.uint8, device="cuda")
        st = ctx.call("rt_render_device_static", C.byref(cam), C.byref(big), 0, 0, 8, 8, C.c_void_p(d.data_ptr()), None)
        assert st == -1 and b"RT_STATIC_MAX_PIXELS" in rtamd.load().rt_last_error(ctx._h)
        # the largest frame it takes: 2048 x 1024 pixels, one tile traced
        edge = rtamd.params(2048, 1024, 2, 4, seed=3)
        a = torch.zeros(2048 * 1024 * 3, dtype=torch.uint8, device="cuda")
        b = torch.zeros(2048 * 1024 * 3, dtype=torch.uint8, device="cuda")
        ctx.render_device(cam, edge, a.data_ptr(), tile=(1984, 960, 64, 64))
        ctx.render_device_static(cam, edge, b.data_ptr(), tile=(1984, 960, 64, 64))
        torch.cuda.synchronize()
        assert torch.equal(a, b) and int(a.count_nonzero()) > 0
    finally:
        ctx.close()


def test_static_entry_point_under_graph_capture_and_across_contexts():
    """The callers the entry point exists for.  (1) A stream under HIP graph capture: the call is captured
    (it neither waits for nor records the context's ordering events there), the graph replays twice to the
    bytes of rt_render_device, and a plain call on the same context afterwards still works.  (2) Two
    contexts of one GPU calling on different streams without synchronising: the library orders calls on
    the GPU's one scratch array, so both frames are right."""
    import torch

    ctx = rtamd.Context(0)
    other = rtamd.Context(0)
    try:
        sc = np.fromfile(os.path.join(G, "scene_c3_500.bin"), dtype=rtamd.SPHERE_DTYPE)
        ctx.set_scene(sc)
        other.set_scene(sc)
        cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
        p, q = rtamd.params(320, 180, 6, 8, seed=41), rtamd.params(256, 144, 9, 8, seed=42)
        want_p = torch.zeros(p.height * p.width * 3, dtype=torch.uint8, device="cuda")
        want_q = torch.zeros(q.height * q.width * 3, dtype=torch.uint8, device="cuda")
        ctx.render_device(cam, p, want_p.data_ptr())
        ctx.render_device(cam, q, want_q.data_ptr())
        torch.cuda.synchronize()
        # (1) capture, replay twice
        out = torch.zeros_like(want_p)
        s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            ctx.render_device_static(cam, p, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for _ in range(2):
            out.zero_()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, want_p)
        out.zero_()
        ctx.render_device_static(cam, p, out.data_ptr())  # a plain call after the capture
        torch.cuda.synchronize()
        assert torch.equal(out, want_p)
        # (2) two contexts, two streams, no synchronisation in between
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        out_q = torch.zeros_like(want_q)
        for _ in range(3):
            out.zero_()
            out_q.zero_()
            torch.cuda.synchronize()
            ctx.render_device_static(cam, p, out.data_ptr(), stream=s1.cuda_stream)
            other.render_device_static(cam, q, out_q.data_ptr(), stream=s2.cuda_stream)
            ctx.render_device_static(cam, p, out.data_ptr(), stream=s2.cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(out, want_p) and torch.equal(out_q, want_q)
    finally:
        other.close()
        ctx.close()
