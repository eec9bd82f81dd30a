"""GPU parity for what the library adds beside the reference's path: progressive delivery of the
frame (the reference shows partial frames while its threads run, Source/main.cpp:295-321), and the
opt-in dielectric / gamma flags.  All comparisons are bit-exact, through the C ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_lib as O
import rtamd
import scenes

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
C3_CAM = scenes.RANDOM_SPHERES_CAMERA


@pytest.fixture(scope="module")
def ctx():
    c = rtamd.Context(0)
    yield c
    c.close()


def c3_scene():
    return np.fromfile(os.path.join(G, "scene_c3_500.bin"), dtype=rtamd.SPHERE_DTYPE)


def glass_default_scene():
    sc = rtamd.default_scene()
    sc[1]["material"], sc[1]["color"] = rtamd.MAT_GLASS, (1.0, 1.0, 1.0)
    return sc


@pytest.mark.parametrize("flags", [rtamd.FLAG_DIELECTRIC, rtamd.FLAG_GAMMA2, rtamd.FLAG_DIELECTRIC | rtamd.FLAG_GAMMA2,
                                   rtamd.FLAG_FLOAT_SHADING, rtamd.FLAG_FLOAT_SHADING | rtamd.FLAG_DIELECTRIC | rtamd.FLAG_GAMMA2])
@pytest.mark.parametrize("scene", ["glass_default", "c3"])
def test_flagged_frames_bit_exact_against_oracle(ctx, scene, flags):
    sc, cam, W, H, spp, depth = (
        (glass_default_scene(), scenes.DEFAULT_CAMERA, 96, 54, 6, 10)
        if scene == "glass_default"
        else (c3_scene(), C3_CAM, 120, 68, 4, 8)
    )
    ctx.set_scene(sc)
    ctx.reset_stats()
    got = ctx.render(rtamd.make_camera(*cam), rtamd.params(W, H, spp, depth, seed=11, flags=flags))
    want, _, rays = O.trace_tile(sc, O.camera(*cam), W, H, spp, depth, O.RNG_COUNTER, seed=11, flags=flags)
    assert np.array_equal(got, want)
    assert ctx.stats().rays == rays


def test_dielectric_same_bits_under_every_scan(ctx):
    sc = c3_scene()
    ctx.set_scene(sc)
    cam, p = rtamd.make_camera(*C3_CAM), rtamd.params(160, 90, 4, 12, seed=5, flags=rtamd.FLAG_DIELECTRIC)
    frames = []
    for filt in (rtamd.FILTER_EXACT, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX):
        ctx.set_option(rtamd.OPT_FILTER, filt)
        frames.append(ctx.render(cam, p))
    ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)
    assert np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[2])
    want, _, _ = O.trace_tile(sc, O.camera(*C3_CAM), 160, 90, 4, 12, O.RNG_COUNTER, seed=5, flags=O.FLAG_DIELECTRIC)
    assert np.array_equal(frames[0], want)
    plain = ctx.render(cam, rtamd.params(160, 90, 4, 12, seed=5))
    assert not np.array_equal(plain, frames[0])


def test_unknown_flags_are_rejected(ctx):
    ctx.set_scene(rtamd.default_scene())
    with pytest.raises(rtamd.RtError):
        ctx.render(rtamd.default_camera(), rtamd.params(16, 16, 1, 2, flags=8))
    p = rtamd.params(16, 16, 1, 2)
    p.reserved = 1
    with pytest.raises(rtamd.RtError):
        ctx.render(rtamd.default_camera(), p)


def test_progressive_passes_and_final_frame(ctx):
    sc = c3_scene()
    ctx.set_scene(sc)
    cam, W, H, spp, depth = rtamd.make_camera(*C3_CAM), 96, 54, 10, 8
    p = rtamd.params(W, H, spp, depth, seed=3)
    whole = ctx.render(cam, p)
    seen = []

    def on_pass(done, total, rgb):
        seen.append((done, total, rgb.copy()))
        return False

    final = ctx.render_progressive(cam, p, 4, on_pass)
    assert [(d, t) for d, t, _ in seen] == [(4, 10), (8, 10), (10, 10)]
    assert np.array_equal(final, whole) and np.array_equal(seen[-1][2], whole)
    # every intermediate frame is the average of the samples drawn so far
    for done, _, img in seen:
        want, _, _ = O.trace_tile(sc, O.camera(*C3_CAM), W, H, done, depth, O.RNG_COUNTER, seed=3, s0=0, s1=done)
        assert np.array_equal(img, want)


@pytest.mark.parametrize("per_launch", [0, 1, 3, 8])
def test_progressive_small_passes_share_launches(ctx, per_launch):
    """Passes of one or two samples are traced several to a launch (after the first, which goes alone): every
    pass's frame is still delivered, in order, and is the average of the samples so far; a stop in the middle
    of a launch's frames leaves the frame of the pass that asked for it."""
    sc = c3_scene()
    ctx.set_scene(sc)
    cam, W, H, spp, depth = rtamd.make_camera(*C3_CAM), 72, 40, 13, 8
    p = rtamd.params(W, H, spp, depth, seed=6)
    ctx.set_option(rtamd.OPT_PROGRESSIVE_PASSES_PER_LAUNCH, per_launch)
    try:
        seen = []
        final = ctx.render_progressive(cam, p, 1, lambda d, t, rgb: seen.append((d, rgb.copy())) and False)
        assert [d for d, _ in seen] == list(range(1, spp + 1))
        assert np.array_equal(final, ctx.render(cam, p))
        for done, img in seen:
            want, _, _ = O.trace_tile(sc, O.camera(*C3_CAM), W, H, done, depth, O.RNG_COUNTER, seed=6, s0=0, s1=done)
            assert np.array_equal(img, want), done
        # passes of two samples (the last one of one), a tile narrower than the frame, a stop at the sixth sample
        tile = (9, 5, 40, 30)
        calls = []
        img = ctx.render_progressive(cam, p, 2, lambda d, t, rgb: calls.append(d) or d >= 6, tile=tile, fill=3)
        assert calls == [2, 4, 6]
        want = np.full((H, W, 3), 3, dtype=np.uint8)
        O.trace_tile(sc, O.camera(*C3_CAM), W, H, 6, depth, O.RNG_COUNTER, seed=6, s0=0, s1=6, tile=tile, rgb=want)
        assert np.array_equal(img, want)
        # and the context's other calls are not disturbed by what the stopped call had enqueued
        assert np.array_equal(ctx.render(cam, p), final)
    finally:
        ctx.set_option(rtamd.OPT_PROGRESSIVE_PASSES_PER_LAUNCH, 0)


def test_progressive_early_stop_tile_and_errors(ctx):
    sc = rtamd.default_scene()
    ctx.set_scene(sc)
    cam, p = rtamd.default_camera(), rtamd.params(64, 40, 9, 6, seed=8, flags=rtamd.FLAG_GAMMA2)
    calls = []
    tile = (10, 6, 30, 20)
    img = ctx.render_progressive(cam, p, 2, lambda d, t, rgb: calls.append(d) or d >= 4, tile=tile, fill=7)
    assert calls == [2, 4]
    want = np.full((40, 64, 3), 7, dtype=np.uint8)
    O.trace_tile(sc, O.default_camera(), 64, 40, 4, 6, O.RNG_COUNTER, seed=8, s0=0, s1=4, tile=tile, rgb=want,
                 flags=O.FLAG_GAMMA2)
    assert np.array_equal(img, want)  # pixels outside the tile untouched
    # one pass larger than spp == plain render; no callback is fine
    assert np.array_equal(ctx.render_progressive(cam, p, 100), ctx.render(cam, p))
    with pytest.raises(rtamd.RtError):
        ctx.render_progressive(cam, p, 0)
    with pytest.raises(rtamd.RtError):
        ctx.render_progressive(cam, rtamd.params(64, 40, 9, 6, sample_begin=1, sample_count=2), 2)
