"""Parity of the HIP path (through the C ABI) with the CPU oracle and the
reference's golden outputs.  Bit-exact for indices, bytes and integer sums;
PSNR >= 40 dB and mean absolute error <= 1.0 (8-bit) against converged reference
images, where the random streams differ by construction."""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_lib as O
import rtamd
import scenes

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KAT = json.load(open(os.path.join(G, "kat.json")))
C3_CAM = scenes.RANDOM_SPHERES_CAMERA


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available()
    return torch


ALL_OPTIONS = [rtamd.OPT_FILTER, rtamd.OPT_TILE_SPHERES, rtamd.OPT_WORKGROUPS_PER_CU, rtamd.OPT_BVH_LEAF_SPHERES,
               rtamd.OPT_BVH_MIN_SPHERES, rtamd.OPT_BVH_LDS_NODES, rtamd.OPT_BVH_SUSPEND_LANES, rtamd.OPT_CHUNKS_PER_WAVE,
               rtamd.OPT_SUBTILES, rtamd.OPT_COUNT_WORK, rtamd.OPT_DIRECT_ADD_SAMPLES, rtamd.OPT_MAX_CHUNKS_PER_LAUNCH,
               rtamd.OPT_PROGRESSIVE_PASSES_PER_LAUNCH]


@pytest.fixture(scope="module")
def ctx():
    c = rtamd.Context(0)
    c.shipped = {o: c.get_option(o) for o in ALL_OPTIONS}  # what a fresh context runs with
    yield c
    c.close()


def restore_shipped_options(c):
    for o, v in c.shipped.items():
        c.set_option(o, v)


def c3_scene():
    return np.fromfile(os.path.join(G, "scene_c3_500.bin"), dtype=rtamd.SPHERE_DTYPE)


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10 * np.log10(255.0 ** 2 / mse) if mse > 0 else float("inf")


# ---------------------------------------------------------------- deterministic primary rays


def test_primary_hits_default_scene_matches_reference_map(ctx):
    ctx.set_scene(rtamd.default_scene())
    got = ctx.primary_hits(rtamd.default_camera(), 1920, 1080)
    want = np.load(os.path.join(G, "c2_default_hits.npz"))["hits"]
    assert np.array_equal(got, want)
    assert "%016x" % O.fnv1a64(got) == KAT["c2_default_fnv"]


@pytest.mark.parametrize("filt", [rtamd.FILTER_EXACT, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_BVH])
def test_primary_hits_random_spheres_matches_reference_map(ctx, filt):
    ctx.set_option(rtamd.OPT_FILTER, filt)
    ctx.set_scene(c3_scene())
    got = ctx.primary_hits(rtamd.make_camera(*C3_CAM), 1920, 1080)
    ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)
    want = np.load(os.path.join(G, "c2_c3scene_hits.npz"))["hits"]
    assert np.array_equal(got, want)


def test_generated_scene_equals_committed_fixture():
    assert scenes.random_spheres(500).tobytes() == c3_scene().tobytes()


@pytest.mark.parametrize("filt", [rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_BVH])
@pytest.mark.parametrize("name", ["default", "c3scene"])
def test_intersect_matches_reference_vectors(ctx, name, filt):
    z = np.load(os.path.join(G, f"isect_{name}.npz"))
    ctx.set_scene(rtamd.default_scene() if name == "default" else c3_scene())
    ctx.set_option(rtamd.OPT_FILTER, filt)
    try:
        got = ctx.intersect(z["rays"])
    finally:
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)
    assert got.tobytes() == z["out"].tobytes()


@pytest.mark.parametrize("filt", [rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_BVH])
def test_intersect_ties_inside_tangent_and_degenerate(ctx, filt):
    """Later sphere wins equal T; origins inside spheres; zero-length and NaN directions."""
    rs = np.random.RandomState(11)
    n = 70  # not a multiple of 32: ragged last mask word
    sc = np.zeros(n, dtype=rtamd.SPHERE_DTYPE)
    sc["center"] = rs.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    sc["radius"] = rs.uniform(0.3, 1.5, size=n).astype(np.float32)
    sc["material"] = 1
    sc[40] = sc[7]   # exact duplicates: ties
    sc[69] = sc[7]
    ctx.set_scene(sc)
    rays = np.zeros((4096, 6), dtype=np.float32)
    rays[:, :3] = rs.uniform(-5, 5, size=(4096, 3))
    rays[:, 3:] = rs.uniform(-1, 1, size=(4096, 3))
    rays[:64, :3] = sc["center"][:64] + 0.1  # inside
    rays[64:128, 3:] = sc["center"][7] - rays[64:128, :3]  # aimed at the duplicated sphere
    rays[128, 3:] = 0.0  # A == 0: NaN root, accepted by the reference's comparisons
    rays[129, 3:] = (np.nan, 1.0, 0.0)
    rays[130, :3] = (np.inf, 0.0, 0.0)
    rays[131, 3:] = (1e30, 1e30, 1e30)
    rays[132, 3:] = (1e-30, 0.0, 0.0)
    rays[133:160, 3] = 0.0  # axis-parallel directions: zero components
    rays[160:180, 3:5] = 0.0
    rays[180:190, 4] = 1e-35
    ctx.set_option(rtamd.OPT_FILTER, filt)
    try:
        got = ctx.intersect(rays)
    finally:
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)
    t, p, nrm = C.c_float(), (C.c_float * 3)(), (C.c_float * 3)()
    ties = 0
    for i, r in enumerate(rays):
        idx = O.lib().orc_intersect(sc.ctypes.data, n, (C.c_float * 3)(*r[:3]), (C.c_float * 3)(*r[3:]), 0.001,
                                    float("inf"), C.byref(t), p, nrm)
        assert got["index"][i] == idx, (i, r)
        ties += idx == 69
        if idx >= 0 and np.isfinite(r).all() and i not in (128, 131, 132):
            want = np.array([t.value, *p, *nrm], dtype=np.float32)
            have = np.concatenate([[got["t"][i]], got["point"][i], got["normal"][i]]).astype(np.float32)
            assert want.tobytes() == have.tobytes(), i
    assert ties >= 5 and not np.isin(got["index"], [7, 40]).any()  # only the LAST copy of the duplicated sphere ever wins


def test_primary_hits_untamed_scene_falls_back_to_exact_scan(ctx):
    """A zero-radius and a huge sphere violate the filter's preconditions."""
    sc = rtamd.default_scene()
    sc = np.concatenate([sc, sc[:2]])
    sc[4]["radius"] = 0.0
    sc[5]["center"] = (1e12, 0, 0)
    ctx.set_scene(sc)
    got = ctx.primary_hits(rtamd.default_camera(), 320, 180)
    assert np.array_equal(got, O.primary_hits(sc, O.default_camera(), 320, 180))


# ---------------------------------------------------------------- stochastic frames, bit-exact vs oracle


@pytest.mark.parametrize(
    "scene,cam,W,H,spp,depth,seed",
    [
        ("default", scenes.DEFAULT_CAMERA, 128, 72, 8, 10, 7),
        ("default", scenes.DEFAULT_CAMERA, 37, 23, 5, 3, 1),  # ragged: partial 8x8 tiles
        ("default", ((-2.0, 2.0, 1.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 90.0), 64, 36, 4, 10, 2),  # commented camera, :600-603
        ("c3", C3_CAM, 96, 54, 4, 8, 1),
        ("c3", C3_CAM, 40, 30, 2, 1, 99),  # depth 1: primary rays only
    ],
)
def test_frame_bit_exact_against_oracle(ctx, scene, cam, W, H, spp, depth, seed):
    sc = rtamd.default_scene() if scene == "default" else c3_scene()
    ctx.set_scene(sc)
    ctx.reset_stats()
    got = ctx.render(rtamd.make_camera(*cam), rtamd.params(W, H, spp, depth, seed=seed))
    want, _, rays = O.trace_tile(sc, O.camera(*cam), W, H, spp, depth, O.RNG_COUNTER, seed=seed)
    assert np.array_equal(got, want)
    assert ctx.stats().rays == rays


@pytest.mark.parametrize("filt", [rtamd.FILTER_EXACT, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX, rtamd.FILTER_BVH])
def test_multi_tile_sphere_list_bit_exact(ctx, filt):
    """Sphere list longer than one LDS tile: two tiles of 128 plus a ragged one of 44."""
    sc = scenes.random_spheres(300, grid_half=9, seed=3)
    cam = C3_CAM
    ctx.set_option(rtamd.OPT_TILE_SPHERES, 128)
    ctx.set_option(rtamd.OPT_FILTER, filt)
    try:
        ctx.set_scene(sc)
        ctx.reset_stats()
        got = ctx.render(rtamd.make_camera(*cam), rtamd.params(64, 36, 3, 6, seed=4))
        rays_gpu = ctx.stats().rays
        hits = ctx.primary_hits(rtamd.make_camera(*cam), 160, 90)
    finally:
        ctx.set_option(rtamd.OPT_TILE_SPHERES, 512)
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)
    want, _, rays = O.trace_tile(sc, O.camera(*cam), 64, 36, 3, 6, O.RNG_COUNTER, seed=4)
    assert np.array_equal(got, want) and rays_gpu == rays
    assert np.array_equal(hits, O.primary_hits(sc, O.camera(*cam), 160, 90))


def test_all_scans_agree(ctx):
    """Unfused reference discriminant, fused vector filter, matrix-core filter and box hierarchy:
    identical frames."""
    ctx.set_scene(c3_scene())
    cam, p = rtamd.make_camera(*C3_CAM), rtamd.params(480, 270, 8, 8, seed=5)
    frames = []
    for filt in (rtamd.FILTER_EXACT, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX, rtamd.FILTER_BVH):
        ctx.set_option(rtamd.OPT_FILTER, filt)
        frames.append(ctx.render(cam, p))
    ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)
    for f in frames[1:]:
        assert np.array_equal(frames[0], f)


def test_matrix_filter_on_a_four_sphere_scene(ctx):
    """Forced matrix scan on the reference's own scene: 4 spheres padded to one group of 128."""
    sc = rtamd.default_scene()
    ctx.set_scene(sc)
    ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_MATRIX)
    try:
        got = ctx.render(rtamd.default_camera(), rtamd.params(128, 72, 8, 10, seed=7))
    finally:
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)
    want, _, _ = O.trace_tile(sc, O.default_camera(), 128, 72, 8, 10, O.RNG_COUNTER, seed=7)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [1, 4, 5, 11, 12, 13])
def test_lists_around_the_short_list_kernels_limit(ctx, n):
    """Up to 12 spheres the vector scan is the reference's loop over scalar-loaded records in a kernel of
    its own, from 13 on the block filter over an LDS tile: both sides of the switch against the oracle,
    with a duplicated sphere (ties go to the later one, Source/main.cpp:534-546) and the extension kernel."""
    rs = np.random.RandomState(100 + n)
    sc = np.zeros(n, dtype=rtamd.SPHERE_DTYPE)
    sc["center"] = rs.uniform(-2.5, 2.5, size=(n, 3)).astype(np.float32) + np.array([0, 0, -4], dtype=np.float32)
    sc["radius"] = rs.uniform(0.3, 1.2, size=n).astype(np.float32)
    sc["material"] = rs.randint(0, 4, size=n)
    sc["color"] = np.where((sc["material"] == 1)[:, None], rs.uniform(0, 255, (n, 3)), rs.uniform(0, 1, (n, 3)))
    sc["fuzz"] = rs.uniform(0, 1, size=n)
    if n >= 4:
        sc[n - 1] = sc[1]
    ctx.set_scene(sc)
    for flags in (0, rtamd.FLAG_DIELECTRIC | rtamd.FLAG_GAMMA2):
        want, _, rays = O.trace_tile(sc, O.default_camera(), 80, 45, 5, 9, O.RNG_COUNTER, seed=n, flags=flags)
        for filt in (rtamd.FILTER_AUTO, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_EXACT):
            ctx.set_option(rtamd.OPT_FILTER, filt)
            ctx.reset_stats()
            got = ctx.render(rtamd.default_camera(), rtamd.params(80, 45, 5, 9, seed=n, flags=flags))
            assert np.array_equal(got, want) and ctx.stats().rays == rays, (n, flags, filt)
    ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)


def test_matrix_filter_far_from_origin_and_degenerate_paths(ctx):
    """The matrix filter expands about the scene's median centre; a scene 5000 units away from the
    origin, with a tiny and a huge sphere mixed in, must still match the oracle bit for bit."""
    rs = np.random.RandomState(21)
    n = 150
    sc = np.zeros(n, dtype=rtamd.SPHERE_DTYPE)
    sc["center"] = (rs.uniform(-6, 6, size=(n, 3)) + np.array([5000.0, -3000.0, 4000.0])).astype(np.float32)
    sc["radius"] = rs.uniform(0.2, 1.0, size=n).astype(np.float32)
    sc["radius"][5] = 1e-3
    sc["radius"][6] = 40.0
    sc["material"] = rs.randint(0, 4, size=n)
    sc["color"] = np.where((sc["material"] == 1)[:, None], rs.uniform(0, 255, (n, 3)), rs.uniform(0, 1, (n, 3)))
    sc["fuzz"] = rs.uniform(0, 1, size=n)
    cam = ((5000.0, -3000.0, 4020.0), (5000.0, -3000.0, 4000.0), (0.0, 1.0, 0.0), 40.0)
    ctx.set_scene(sc)
    want, _, rays = O.trace_tile(sc, O.camera(*cam), 96, 54, 6, 8, O.RNG_COUNTER, seed=12)
    try:
        for filt in (rtamd.FILTER_MATRIX, rtamd.FILTER_BVH):
            ctx.set_option(rtamd.OPT_FILTER, filt)
            ctx.reset_stats()
            got = ctx.render(rtamd.make_camera(*cam), rtamd.params(96, 54, 6, 8, seed=12))
            assert np.array_equal(got, want) and ctx.stats().rays == rays
    finally:
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)
    assert (got != got[0, 0]).any()  # something was hit


def test_sub_tile_leaves_rest_of_frame_untouched(ctx):
    """TraceThreadFunction(xstart, ystart, xsize, ysize) on a strip, like TraceScene's four."""
    sc = rtamd.default_scene()
    ctx.set_scene(sc)
    W, H = 64, 36
    p = rtamd.params(W, H, 4, 10, seed=3)
    cam = rtamd.default_camera()
    full = ctx.render(cam, p)
    img = np.full((H, W, 3), 255, dtype=np.uint8)  # g_PixelData's memset, Source/main.cpp:175
    ctx.render(cam, p, tile=(16, 0, 16, H), rgb=img)
    assert np.array_equal(img[:, 16:32], full[:, 16:32])
    assert (img[:, :16] == 255).all() and (img[:, 32:] == 255).all()
    # four strips assemble the frame; out-of-range parts are dropped like PutPixel does
    for t in range(4):
        ctx.render(cam, p, tile=(16 * t, -5, 16, H + 20), rgb=img)
    assert np.array_equal(img, full)


def test_four_threads_call_the_seam_at_once_like_tracescene(ctx):
    """TraceScene (Source/main.cpp:647-659) starts four threads that each call
    TraceThreadFunction on a quarter of the columns, at the same time, into the same buffer."""
    import threading

    sc = c3_scene()
    ctx.set_scene(sc)
    W, H = 128, 72
    cam, p = rtamd.make_camera(*C3_CAM), rtamd.params(W, H, 4, 8, seed=13)
    want = ctx.render(cam, p)
    img = np.full((H, W, 3), 255, dtype=np.uint8)
    errors = []

    def strip(i):
        try:
            ctx.render(cam, p, tile=(i * (W // 4), 0, W // 4, H), rgb=img)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    for _ in range(3):
        img[:] = 255
        threads = [threading.Thread(target=strip, args=(i,)) for i in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors
        assert np.array_equal(img, want)


def test_launches_on_two_streams_of_one_context_do_not_disturb_each_other(ctx, torch):
    """Trace launches share the context's work queue; enqueued back to back on different streams,
    without any synchronisation by the caller, each must still trace every chunk exactly once."""
    sc = c3_scene()
    ctx.set_scene(sc)
    cam = rtamd.make_camera(*C3_CAM)
    W, H = 640, 360
    pa, pb = rtamd.params(W, H, 16, 8, seed=5), rtamd.params(W, H, 3, 8, seed=6)
    want = []
    for p in (pa, pb):
        d = torch.zeros(H * W * 3, dtype=torch.int32, device="cuda")
        ctx.accumulate_device(cam, p, d.data_ptr())
        torch.cuda.synchronize()
        want.append(d.cpu().numpy())
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        da = torch.zeros(H * W * 3, dtype=torch.int32, device="cuda")
        db = torch.zeros(H * W * 3, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.reset_stats()
        ctx.accumulate_device(cam, pa, da.data_ptr(), stream=sa.cuda_stream)
        ctx.accumulate_device(cam, pb, db.data_ptr(), stream=sb.cuda_stream)
        ctx.accumulate_device(cam, pb, da.data_ptr(), stream=sb.cuda_stream)  # behind the first launch: same buffer
        torch.cuda.synchronize()
        assert np.array_equal(db.cpu().numpy(), want[1])
        assert np.array_equal(da.cpu().numpy(), want[0] + want[1])


def test_render_device_strips_on_four_streams_share_the_contexts_scratch(ctx, torch):
    """TraceScene's four column strips (Source/main.cpp:647-659) through rt_render_device on four
    streams, nothing synchronised by the caller: the strips share every row of the context's scratch
    sums, so each call's zeroing, tracing and dividing must stay behind the previous call's."""
    sc = c3_scene()
    ctx.set_scene(sc)
    cam = rtamd.make_camera(*C3_CAM)
    W, H = 512, 288
    p = rtamd.params(W, H, 12, 8, seed=17)
    want = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
    ctx.render_device(cam, p, want.data_ptr())
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    for _ in range(4):
        got = torch.full((H * W * 3,), 255, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for i, s in enumerate(streams):
            ctx.render_device(cam, p, got.data_ptr(), tile=(i * (W // 4), 0, W // 4, H), stream=s.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(got, want)


def test_scene_with_nan_and_infinite_spheres_takes_the_exact_scan(ctx):
    """NaN or infinite centres and radii put a scene outside the filters' and the hierarchy's
    preconditions; it must be scanned the reference's way, whatever scan is asked for."""
    sc = scenes.random_spheres(120, grid_half=5, seed=9)
    sc["center"][11, 0] = np.nan
    sc["center"][12, 2] = np.inf
    sc["radius"][13] = np.nan
    sc["radius"][14] = np.inf
    want, _, rays = O.trace_tile(sc, O.camera(*C3_CAM), 64, 36, 3, 6, O.RNG_COUNTER, seed=2)
    try:
        for filt in (rtamd.FILTER_AUTO, rtamd.FILTER_MATRIX, rtamd.FILTER_BVH):
            ctx.set_option(rtamd.OPT_FILTER, filt)
            ctx.set_scene(sc)
            ctx.reset_stats()
            got = ctx.render(rtamd.make_camera(*C3_CAM), rtamd.params(64, 36, 3, 6, seed=2))
            assert np.array_equal(got, want), filt
            assert ctx.stats().rays == rays
    finally:
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)


def test_sample_slices_and_interleaved_tiles_are_partition_invariant(ctx, torch):
    sc = c3_scene()
    ctx.set_scene(sc)
    cam = rtamd.make_camera(*C3_CAM)
    W, H, SPP, D = 100, 60, 6, 8
    want, sums_o, rays_o = O.trace_tile(sc, O.camera(*C3_CAM), W, H, SPP, D, O.RNG_COUNTER, seed=8, want_sums=True)
    # (a) one call
    d_sums = torch.zeros(H * W * 3, dtype=torch.int32, device="cuda")
    ctx.reset_stats()
    ctx.accumulate_device(cam, rtamd.params(W, H, SPP, D, seed=8), d_sums.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_sums.cpu().numpy().view(np.uint32).reshape(H, W, 3), sums_o)
    assert ctx.stats().rays == rays_o
    # (b) 2 sample slices x 3 interleaved tile sets, as 6 "ranks"
    d_sums.zero_()
    ctx.reset_stats()
    for s0, sc_ in ((0, 4), (4, 2)):
        for first in range(3):
            p = rtamd.params(W, H, SPP, D, seed=8, sample_begin=s0, sample_count=sc_, tile_first=first, tile_stride=3)
            ctx.accumulate_device(cam, p, d_sums.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_sums.cpu().numpy().view(np.uint32).reshape(H, W, 3), sums_o)
    assert ctx.stats().rays == rays_o
    d_rgb = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
    ctx.resolve_device(d_sums.data_ptr(), W, H, SPP, d_rgb.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_rgb.cpu().numpy().reshape(H, W, 3), want)


@pytest.mark.parametrize("filt", [rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX, rtamd.FILTER_BVH])
def test_split_sample_ranges_give_the_same_sums(ctx, torch, filt):
    """A small frame's tiles are cut into strips of rows and into sample segments (several
    waves then add into the same pixels); the sums and the ray count are those of the uncut run and
    of the oracle, for ragged tiles and segments, sample slices and interleaved tile sets alike."""
    sc = c3_scene()
    ctx.set_scene(sc)
    ctx.set_option(rtamd.OPT_FILTER, filt)
    cam = rtamd.make_camera(*C3_CAM)
    W, H, SPP, D = 61, 43, 37, 6  # ragged tiles at both edges, 37 = 4*9 + 1 samples
    _, sums_o, rays_o = O.trace_tile(sc, O.camera(*C3_CAM), W, H, SPP, D, O.RNG_COUNTER, seed=21, want_sums=True)
    d_sums = torch.zeros(H * W * 3, dtype=torch.int32, device="cuda")
    try:
        for per_wave, subtiles in ((0, 4), (1, 1), (16, 1), (32, 1), (256, 1), (1, 2), (32, 2), (256, 2), (1, 4), (16, 4),
                                   (32, 4), (256, 4), (1, 8), (32, 8), (256, 8)):
            ctx.set_option(rtamd.OPT_CHUNKS_PER_WAVE, per_wave)
            ctx.set_option(rtamd.OPT_SUBTILES, subtiles)
            d_sums.zero_()
            ctx.reset_stats()
            ctx.accumulate_device(cam, rtamd.params(W, H, SPP, D, seed=21), d_sums.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(d_sums.cpu().numpy().view(np.uint32).reshape(H, W, 3), sums_o), (per_wave, subtiles)
            assert ctx.stats().rays == rays_o
        # split segments within sample slices and interleaved tile sets
        ctx.set_option(rtamd.OPT_CHUNKS_PER_WAVE, 256)
        d_sums.zero_()
        ctx.reset_stats()
        for s0, sc_ in ((0, 9), (9, 20), (29, 8)):
            for first in range(2):
                p = rtamd.params(W, H, SPP, D, seed=21, sample_begin=s0, sample_count=sc_, tile_first=first, tile_stride=2)
                ctx.accumulate_device(cam, p, d_sums.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_sums.cpu().numpy().view(np.uint32).reshape(H, W, 3), sums_o)
        assert ctx.stats().rays == rays_o
    finally:
        restore_shipped_options(ctx)


@pytest.mark.parametrize("filt", [rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_BVH])
def test_direct_adds_of_low_sample_calls_give_the_same_sums(ctx, torch, filt):
    """Calls of very few samples per pixel let every path add its colour to the frame itself
    (RT_OPT_DIRECT_ADD_SAMPLES) instead of through the per-chunk sums: same sums, same rays, also
    into a buffer that already holds values, for whole frames, tiles, slices and cut chunks."""
    sc = c3_scene()
    ctx.set_scene(sc)
    ctx.set_option(rtamd.OPT_FILTER, filt)
    cam = rtamd.make_camera(*C3_CAM)
    W, H, D = 61, 43, 6
    try:
        for spp, kw in ((1, {}), (2, {}), (3, {}), (5, dict(sample_begin=2, sample_count=1)), (24, dict(tile_first=1, tile_stride=3))):
            p = rtamd.params(W, H, spp, D, seed=31, **kw)
            out = []
            for direct in (0, 1, 64):
                ctx.set_option(rtamd.OPT_DIRECT_ADD_SAMPLES, direct)
                d = torch.full((H * W * 3,), 7, dtype=torch.int32, device="cuda")
                ctx.reset_stats()
                ctx.accumulate_device(cam, p, d.data_ptr())
                torch.cuda.synchronize()
                out.append((d.cpu().numpy(), ctx.stats().rays))
            assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][0], out[2][0]), (spp, kw)
            assert out[0][1] == out[1][1] == out[2][1]
            if not kw:
                _, sums_o, rays_o = O.trace_tile(sc, O.camera(*C3_CAM), W, H, spp, D, O.RNG_COUNTER, seed=31, want_sums=True)
                assert np.array_equal(out[2][0].view(np.uint32).reshape(H, W, 3) - 7, sums_o) and out[2][1] == rays_o
    finally:
        ctx.set_option(rtamd.OPT_DIRECT_ADD_SAMPLES, 1)
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)


def test_random_calls_do_not_depend_on_how_the_queue_cuts_them(ctx, torch):
    """Random frame sizes, regions (some hanging over the frame), sample slices and interleaved tile
    sets, each traced with whole-tile chunks and with random cuts: same sums, same ray count."""
    rs = np.random.RandomState(2024)
    cam = rtamd.make_camera(*C3_CAM)
    try:
        for case in range(48):
            if case % 16 == 0:
                ctx.set_scene(scenes.random_spheres(int(rs.choice([3, 40, 200, 700])), grid_half=int(rs.randint(3, 9)), seed=case))
            W, H = int(rs.randint(1, 150)), int(rs.randint(1, 100))
            spp = int(rs.choice([1, 3, 7, 8, 9, 16, 33, 64, 130]))
            s0 = int(rs.randint(0, spp))
            cnt = int(rs.randint(1, spp - s0 + 1))
            stride = int(rs.choice([0, 0, 2, 5]))
            first = int(rs.randint(0, stride)) if stride else 0
            tile = None
            if rs.rand() < 0.5:
                tile = (int(rs.randint(-10, W)), int(rs.randint(-10, H)), int(rs.randint(1, W + 20)), int(rs.randint(1, H + 20)))
            p = rtamd.params(W, H, spp, int(rs.randint(1, 9)), seed=case, sample_begin=s0, sample_count=cnt,
                             tile_first=first, tile_stride=stride)
            ctx.set_option(rtamd.OPT_FILTER, int(rs.choice([rtamd.FILTER_AUTO, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_BVH])))
            out = []
            for per_wave, subtiles in ((0, 4), (int(rs.choice([1, 16, 32, 256])), int(rs.choice([1, 2, 4, 8])))):
                ctx.set_option(rtamd.OPT_CHUNKS_PER_WAVE, per_wave)
                ctx.set_option(rtamd.OPT_SUBTILES, subtiles)
                d_sums = torch.full((H * W * 3,), 5, dtype=torch.int32, device="cuda")
                ctx.reset_stats()
                ctx.accumulate_device(cam, p, d_sums.data_ptr(), tile=tile)
                torch.cuda.synchronize()
                out.append((d_sums.cpu().numpy(), ctx.stats().rays))
            assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1], (case, W, H, spp, s0, cnt, stride, first, tile)
    finally:
        restore_shipped_options(ctx)


@pytest.mark.parametrize("filt", [rtamd.FILTER_AUTO, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX])
def test_batched_jobs_add_what_separate_calls_add(ctx, torch, filt):
    """rt_accumulate_batch_device: several jobs in one launch (one work queue, job after job) leave in
    every job's sums exactly what one rt_accumulate_device call per job leaves -- frames of different
    sizes and cameras into buffers of their own, sample slices and interleaved tile sets of one frame
    into one buffer, ragged regions, different depths -- and trace the same number of rays; the first
    job's frame also equals the oracle's."""
    sc = c3_scene()
    ctx.set_scene(sc)
    ctx.set_option(rtamd.OPT_FILTER, filt)
    try:
        cam_a = rtamd.make_camera(*C3_CAM)
        cam_b = rtamd.make_camera((5.0, 1.5, -9.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 35.0)
        shapes = [(61, 43, 9, 6), (128, 72, 4, 8), (33, 90, 16, 3)]
        bufs = [torch.zeros(h * w * 3, dtype=torch.int32, device="cuda") for w, h, _, _ in shapes]
        shared = torch.zeros(80 * 48 * 3, dtype=torch.int32, device="cuda")
        jobs = []
        for (w, h, spp, d), buf, cam in zip(shapes, bufs, (cam_a, cam_b, cam_a)):
            jobs.append((cam, rtamd.params(w, h, spp, d, seed=77), buf.data_ptr()))
        jobs[2] = jobs[2] + ((-4, 10, 30, 200),)  # a region hanging over the frame
        # one 80x48 frame of 24 spp as two sample slices over two tile sets each: four jobs, one buffer
        for s0, cnt in ((0, 10), (10, 14)):
            for first in range(2):
                p = rtamd.params(80, 48, 24, 8, seed=77, sample_begin=s0, sample_count=cnt, tile_first=first, tile_stride=2)
                jobs.append((cam_b, p, shared.data_ptr()))
        assert len(jobs) == 7
        ctx.reset_stats()
        ctx.accumulate_batch_device(jobs)
        torch.cuda.synchronize()
        rays_batch = ctx.stats().rays
        got = [b.cpu().numpy().copy() for b in bufs] + [shared.cpu().numpy().copy()]
        for b in bufs + [shared]:
            b.zero_()
        ctx.reset_stats()
        for job in jobs:
            ctx.accumulate_device(job[0], job[1], job[2], tile=job[3] if len(job) > 3 else None)
        torch.cuda.synchronize()
        assert ctx.stats().rays == rays_batch
        for g, b in zip(got, bufs + [shared]):
            assert np.array_equal(g, b.cpu().numpy())
        w, h, spp, d = shapes[0]
        _, sums_o, _ = O.trace_tile(sc, O.camera(*C3_CAM), w, h, spp, d, O.RNG_COUNTER, seed=77, want_sums=True)
        assert np.array_equal(got[0].view(np.uint32).reshape(h, w, 3), sums_o)
        _, sums_o, _ = O.trace_tile(sc, O.camera((5.0, 1.5, -9.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 35.0), 80, 48, 24, 8,
                                    O.RNG_COUNTER, seed=77, want_sums=True)
        assert np.array_equal(got[3].view(np.uint32).reshape(48, 80, 3), sums_o)
        # every cut of the work queue
        for per_wave, subtiles in ((0, 4), (1, 1), (32, 2), (256, 4), (256, 8)):
            ctx.set_option(rtamd.OPT_CHUNKS_PER_WAVE, per_wave)
            ctx.set_option(rtamd.OPT_SUBTILES, subtiles)
            for b in bufs + [shared]:
                b.zero_()
            ctx.accumulate_batch_device(jobs)
            torch.cuda.synchronize()
            for g, b in zip(got, bufs + [shared]):
                assert np.array_equal(g, b.cpu().numpy()), (per_wave, subtiles)
        # jobs of one launch share the generator's key and the flags; at most RT_MAX_JOBS of them
        other_seed = (cam_a, rtamd.params(61, 43, 9, 6, seed=78), bufs[0].data_ptr())
        with pytest.raises(rtamd.RtError):
            ctx.accumulate_batch_device([jobs[0], other_seed])
        with pytest.raises(rtamd.RtError):
            ctx.accumulate_batch_device([jobs[0]] * (rtamd.MAX_JOBS + 1))
        # jobs with nothing to trace are skipped
        ctx.accumulate_batch_device([(cam_a, rtamd.params(61, 43, 9, 0, seed=77), bufs[0].data_ptr())])
        torch.cuda.synchronize()
    finally:
        restore_shipped_options(ctx)


def test_a_call_too_large_for_one_work_queue_goes_in_several_launches(ctx, torch):
    """A call whose tiles x 256-sample runs outnumber what a launch's queue can count has its sample
    range traced by several launches (instead of being refused).  With the limit lowered to 40 chunks,
    a 64 x 40 frame (40 tiles) of 700 samples takes three launches: same sums and ray count as the oracle's."""
    sc = rtamd.default_scene()
    ctx.set_scene(sc)
    cam = rtamd.default_camera()
    W, H, SPP, D = 64, 40, 700, 4
    _, sums_o, rays_o = O.trace_tile(sc, O.default_camera(), W, H, SPP, D, O.RNG_COUNTER, seed=5, want_sums=True)
    d_sums = torch.zeros(H * W * 3, dtype=torch.int32, device="cuda")
    try:
        for limit in (40, 100, 1 << 29):
            ctx.set_option(rtamd.OPT_MAX_CHUNKS_PER_LAUNCH, limit)
            d_sums.zero_()
            ctx.reset_stats()
            ctx.accumulate_device(cam, rtamd.params(W, H, SPP, D, seed=5), d_sums.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(d_sums.cpu().numpy().view(np.uint32).reshape(H, W, 3), sums_o), limit
            assert ctx.stats().rays == rays_o
        # the same through the entry points that render a frame
        ctx.set_option(rtamd.OPT_MAX_CHUNKS_PER_LAUNCH, 40)
        want, _, _ = O.trace_tile(sc, O.default_camera(), W, H, SPP, D, O.RNG_COUNTER, seed=5)
        assert np.array_equal(ctx.render(cam, rtamd.params(W, H, SPP, D, seed=5)), want)
    finally:
        restore_shipped_options(ctx)


def test_executed_work_counters(ctx):
    """RT_OPT_COUNT_WORK: the counting build gives the same frame and ray count, and tallies what
    the scan executed -- every test for the reference's own loop, a few per cent of them for the
    hierarchy; without the option the tallies stay 0."""
    sc = c3_scene()
    ctx.set_scene(sc)
    cam, p = rtamd.make_camera(*C3_CAM), rtamd.params(160, 90, 4, 8, seed=9)
    want, _, rays = O.trace_tile(sc, O.camera(*C3_CAM), 160, 90, 4, 8, O.RNG_COUNTER, seed=9)
    seen = {}
    try:
        for filt in (rtamd.FILTER_EXACT, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX, rtamd.FILTER_BVH):
            ctx.set_option(rtamd.OPT_FILTER, filt)
            ctx.set_option(rtamd.OPT_COUNT_WORK, 1)
            ctx.reset_stats()
            got = ctx.render(cam, p)
            st = ctx.stats()
            assert np.array_equal(got, want) and st.rays == rays
            assert st.sphere_tests == rays * len(sc)
            seen[filt] = (st.node_visits, st.exact_tests, st.root_evaluations)
            ctx.set_option(rtamd.OPT_COUNT_WORK, 0)
            ctx.reset_stats()
            ctx.render(cam, p)
            st = ctx.stats()
            assert (st.node_visits, st.exact_tests, st.root_evaluations) == (0, 0, 0) and st.rays == rays
    finally:
        ctx.set_option(rtamd.OPT_COUNT_WORK, 0)
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)
    nodes, exact, roots = seen[rtamd.FILTER_EXACT]
    assert nodes == 0 and exact == rays * len(sc)       # the reference's loop: every sphere, every ray
    assert 0 < roots < exact // 50                       # few discriminants are non-negative
    for filt in (rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX):
        n2, e2, r2 = seen[filt]
        # the conservative filters pass a superset of the non-negative discriminants, slightly larger
        assert n2 == 0 and roots <= r2 <= e2 < roots * 1.05 + 64
    nb, eb, rb = seen[rtamd.FILTER_BVH]
    assert nb > rays and eb < exact // 20 and rb <= eb


def test_edge_cases(ctx):
    cam = rtamd.default_camera()
    # empty scene: every ray is sky
    ctx.set_scene(np.zeros(0, dtype=rtamd.SPHERE_DTYPE))
    got = ctx.render(cam, rtamd.params(33, 17, 2, 5, seed=1))
    want, _, _ = O.trace_tile(np.zeros(0, dtype=O.SPHERE_DTYPE), O.default_camera(), 33, 17, 2, 5, O.RNG_COUNTER, seed=1)
    assert np.array_equal(got, want)
    assert (ctx.primary_hits(cam, 33, 17) == -1).all()
    # depth 0: GetRayColor returns black without tracing
    ctx.set_scene(rtamd.default_scene())
    ctx.reset_stats()
    assert (ctx.render(cam, rtamd.params(16, 9, 3, 0)) == 0).all()
    assert ctx.stats().rays == 0
    # 1x1 frame, 1 sample
    got = ctx.render(cam, rtamd.params(1, 1, 1, 10, seed=2))
    want, _, _ = O.trace_tile(O.default_scene(), O.default_camera(), 1, 1, 1, 10, O.RNG_COUNTER, seed=2)
    assert np.array_equal(got, want)
    # all four materials incl. Glass / FuzzyMetal (flat white, Source/main.cpp:594)
    sc = rtamd.default_scene()
    sc[0]["material"] = rtamd.MAT_GLASS
    sc[2]["material"] = rtamd.MAT_FUZZYMETAL
    ctx.set_scene(sc)
    got = ctx.render(cam, rtamd.params(64, 36, 4, 10, seed=6))
    want, _, _ = O.trace_tile(sc, O.default_camera(), 64, 36, 4, 10, O.RNG_COUNTER, seed=6)
    assert np.array_equal(got, want)
    assert (got[18, 10] == 255).all()  # the glass sphere is flat white


def test_colour_conversion_extremes(ctx):
    """ToRGB(vec3) on values outside int range (Source/main.cpp:117-133): the reference's float ->
    int conversion yields INT_MIN for NaN and for anything from 2^31 up, which clamps to 0, not 255."""
    sc = rtamd.default_scene()
    sc["material"] = rtamd.MAT_METAL
    sc["fuzz"] = 0.0
    sc[0]["color"] = (1e38, np.inf, np.nan)
    sc[1]["color"] = (-1.0, 8421504.0, 8421505.0)        # 255 * c straddles 2^31
    sc[2]["color"] = (16843009.0, 0.99999994, 1.0000001)  # 255 * c = 2^32 + ...; just below / above 255
    sc[3]["color"] = (3.5, 0.0, -0.0)
    ctx.set_scene(sc)
    got = ctx.render(rtamd.default_camera(), rtamd.params(96, 54, 4, 6, seed=4))
    want, _, _ = O.trace_tile(sc, O.default_camera(), 96, 54, 4, 6, O.RNG_COUNTER, seed=4)
    assert np.array_equal(got, want)
    assert (got[27, 20] == 0).all()  # left sphere: 1e38 * c, inf * c, nan * c all end as 0


def test_error_paths(ctx):
    fresh = rtamd.Context(0)
    cam, p = rtamd.default_camera(), rtamd.params(8, 8, 1, 1)
    buf = np.zeros((8, 8, 3), dtype=np.uint8)
    assert fresh.call("rt_render", C.byref(cam), C.byref(p), 0, 0, 8, 8, buf.ctypes.data) == -4  # no scene
    fresh.set_scene(rtamd.default_scene())
    assert fresh.call("rt_render", None, C.byref(p), 0, 0, 8, 8, buf.ctypes.data) == -1
    assert fresh.call("rt_render", C.byref(cam), C.byref(p), 0, 0, 8, 8, None) == -1
    bad = rtamd.params(0, 8, 1, 1)
    assert fresh.call("rt_render", C.byref(cam), C.byref(bad), 0, 0, 8, 8, buf.ctypes.data) == -1
    bad = rtamd.params(8, 8, 4, 1, sample_begin=3, sample_count=2)
    assert fresh.call("rt_render", C.byref(cam), C.byref(bad), 0, 0, 8, 8, buf.ctypes.data) == -1
    assert b"sample range" in rtamd.load().rt_last_error(fresh._h)
    bad = rtamd.params(8, 8, 16843010, 1)  # 255 * spp would not fit the 32-bit sums
    assert fresh.call("rt_render", C.byref(cam), C.byref(bad), 0, 0, 8, 8, buf.ctypes.data) == -1
    assert fresh.call("rt_set_option", 99, 0) == -1
    fresh.close()


@pytest.mark.parametrize("scale,offset", [(1e-3, 0.0), (1.0, 0.0), (37.0, 1e4), (1e3, -3e5), (1e5, 2e7)])
def test_all_scans_agree_across_scene_scales(ctx, scale, offset):
    """The conservative filters' margins are relative: shrink, blow up and move the scene; every scan
    must keep giving the oracle's frame (far offsets make the float coordinates coarse, which is
    exactly where a too-tight margin would drop a hit)."""
    rs = np.random.RandomState(int(abs(offset)) % 1000 + 3)
    n = 200
    sc = np.zeros(n, dtype=rtamd.SPHERE_DTYPE)
    sc["center"] = (rs.uniform(-8, 8, size=(n, 3)) * scale + offset).astype(np.float32)
    sc["radius"] = (rs.uniform(0.3, 1.2, size=n) * scale).astype(np.float32)
    sc["material"] = rs.randint(1, 3, size=n)
    sc["color"] = np.where((sc["material"] == 1)[:, None], rs.uniform(0, 255, (n, 3)), rs.uniform(0.3, 1, (n, 3)))
    sc["fuzz"] = rs.uniform(0, 0.5, size=n)
    eye = np.array([0.0, 2.0, 30.0]) * scale + offset
    cam = (tuple(eye), (offset, offset, offset), (0.0, 1.0, 0.0), 35.0)
    want, _, rays = O.trace_tile(sc, O.camera(*cam), 64, 36, 4, 6, O.RNG_COUNTER, seed=5)
    ctx.set_scene(sc)
    try:
        for filt in (rtamd.FILTER_EXACT, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX, rtamd.FILTER_BVH):
            ctx.set_option(rtamd.OPT_FILTER, filt)
            ctx.reset_stats()
            got = ctx.render(rtamd.make_camera(*cam), rtamd.params(64, 36, 4, 6, seed=5))
            assert np.array_equal(got, want), filt
            assert ctx.stats().rays == rays
    finally:
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)
    assert rays > 64 * 36 * 4  # some rays bounced: spheres were hit


# ---------------------------------------------------------------- statistical parity with the reference


def test_converged_default_scene_vs_reference_image(ctx):
    """Reference: 256x144, 4096 spp, depth 10, its own mt19937 stream."""
    want = np.load(os.path.join(G, "c1_converged_256x144_spp4096_d10.npy"))
    ctx.set_scene(rtamd.default_scene())
    got = ctx.render(rtamd.default_camera(), rtamd.params(256, 144, 4096, 10, seed=2024))
    mae = np.abs(got.astype(int) - want.astype(int)).mean()
    assert psnr(got, want) >= 40.0, psnr(got, want)
    assert mae <= 1.0, mae


def test_converged_random_spheres_vs_reference_image(ctx):
    """Reference: 128x72, 1024 spp, depth 8 (noisier than the GPU's 16384 spp estimate)."""
    want = np.load(os.path.join(G, "c3_converged_128x72_spp1024_d8.npy"))
    ctx.set_scene(c3_scene())
    got = ctx.render(rtamd.make_camera(*C3_CAM), rtamd.params(128, 72, 16384, 8, seed=77))
    mae = np.abs(got.astype(int) - want.astype(int)).mean()
    assert psnr(got, want) >= 40.0, psnr(got, want)
    assert mae <= 1.0, mae


# ---------------------------------------------------------------- full-size properties (BASELINE configs)


def test_full_size_frame_properties(ctx, torch):
    """1920x1080, 64 spp, depth 8, 500 spheres: determinism, partition invariance, sampled oracle pixels.
    This is the frame bench.py times: it must run under the options a fresh context ships with."""
    assert {o: ctx.get_option(o) for o in ALL_OPTIONS} == ctx.shipped
    assert ctx.shipped[rtamd.OPT_CHUNKS_PER_WAVE] == 32 and ctx.shipped[rtamd.OPT_FILTER] == rtamd.FILTER_AUTO
    assert ctx.shipped[rtamd.OPT_SUBTILES] == 8
    sc = c3_scene()
    ctx.set_scene(sc)
    cam = rtamd.make_camera(*C3_CAM)
    W, H, SPP, D = 1920, 1080, 64, 8
    n = W * H * 3
    a = torch.zeros(n, dtype=torch.int32, device="cuda")
    b = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.reset_stats()
    ctx.accumulate_device(cam, rtamd.params(W, H, SPP, D, seed=1), a.data_ptr())
    torch.cuda.synchronize()
    rays_full = ctx.stats().rays
    # 4 sample slices, each over 2 interleaved tile sets
    ctx.reset_stats()
    for k in range(4):
        for first in range(2):
            p = rtamd.params(W, H, SPP, D, seed=1, sample_begin=16 * k, sample_count=16, tile_first=first, tile_stride=2)
            ctx.accumulate_device(cam, p, b.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert ctx.stats().rays == rays_full
    assert int(a.max()) <= 255 * SPP and int(a.min()) >= 0
    # checksum of sums is reproducible run to run
    b.zero_()
    ctx.accumulate_device(cam, rtamd.params(W, H, SPP, D, seed=1), b.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # 48 scattered pixels against the oracle, all 64 samples each
    sums = a.cpu().numpy().view(np.uint32).reshape(H, W, 3)
    rs = np.random.RandomState(0)
    ocam = O.camera(*C3_CAM)
    for x, y in zip(rs.randint(0, W, 48), rs.randint(0, H, 48)):
        acc = np.zeros(3, dtype=np.int64)
        for s in range(SPP):
            rgb, _ = O.sample_color(sc, ocam, W, H, D, 1, int(x), int(y), s)
            acc += rgb
        assert sums[y, x].tolist() == acc.tolist(), (x, y)
