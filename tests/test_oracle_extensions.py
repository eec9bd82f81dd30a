"""The opt-in extensions (real dielectric for Material::Glass, gamma 2 on output) in the CPU oracle.
The reference has neither (Source/main.cpp:594 returns flat white for Glass, :638-642 writes the plain
average), so there is no reference output to pin them to: these tests check that they are off by
default, and the properties that follow from their definition."""
import os

import numpy as np

import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
C3_CAM = ((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0)


def c3_scene():
    return np.fromfile(os.path.join(G, "scene_c3_500.bin"), dtype=O.SPHERE_DTYPE)


def test_flags_zero_is_the_reference_path():
    sc, cam = c3_scene(), O.camera(*C3_CAM)
    a, _, ra = O.trace_tile(sc, cam, 48, 27, 2, 8, O.RNG_COUNTER, seed=3)
    want = np.load(os.path.join(G, "c3_mt_64x36_spp16_d8.npy"))
    b, _, _ = O.trace_tile(sc, cam, 64, 36, 16, 8, O.RNG_MT19937)
    assert np.array_equal(b, want)  # the extended entry point still reproduces the reference image
    c, _, rc = O.trace_tile(sc, cam, 48, 27, 2, 8, O.RNG_COUNTER, seed=3, flags=0)
    assert np.array_equal(a, c) and ra == rc


def test_gamma2_table_and_frame():
    lut = np.array([O.lib().orc_gamma2(v) for v in range(256)], dtype=np.uint8)
    assert lut[0] == 0 and lut[255] == 255
    assert np.all(np.diff(lut.astype(int)) >= 0) and np.all(lut >= np.arange(256))
    # trunc(sqrt(v/255)*255) in double agrees except where float rounding sits on an integer
    ref = np.floor(np.sqrt(np.arange(256) / 255.0) * 255.0 + 1e-4).astype(int)
    assert np.max(np.abs(ref - lut.astype(int))) <= 1
    assert O.lib().orc_gamma2(-5) == 0 and O.lib().orc_gamma2(999) == 255
    sc, cam = O.default_scene(), O.default_camera()
    plain, _, _ = O.trace_tile(sc, cam, 40, 24, 3, 6, O.RNG_COUNTER, seed=1)
    gam, _, _ = O.trace_tile(sc, cam, 40, 24, 3, 6, O.RNG_COUNTER, seed=1, flags=O.FLAG_GAMMA2)
    assert np.array_equal(gam, lut[plain])


def test_dielectric_only_touches_glass():
    cam = O.default_camera()
    sc = O.default_scene()  # no glass: the flag must change nothing
    a, _, ra = O.trace_tile(sc, cam, 40, 24, 3, 6, O.RNG_COUNTER, seed=1)
    b, _, rb = O.trace_tile(sc, cam, 40, 24, 3, 6, O.RNG_COUNTER, seed=1, flags=O.FLAG_DIELECTRIC)
    assert np.array_equal(a, b) and ra == rb
    sc[1]["material"], sc[1]["color"] = O.MAT_GLASS, (1.0, 1.0, 1.0)
    a, _, ra = O.trace_tile(sc, cam, 40, 24, 3, 6, O.RNG_COUNTER, seed=1)
    b, _, rb = O.trace_tile(sc, cam, 40, 24, 3, 6, O.RNG_COUNTER, seed=1, flags=O.FLAG_DIELECTRIC)
    assert np.all(a[12, 20] == 255)  # flat white in the reference path, :594
    assert rb > ra and not np.array_equal(a, b)
    # the glass ball is no longer a white disc: it shows the scene behind and below it
    assert not np.all(b[10:14, 18:22] == 255)


def test_dielectric_central_ray_goes_straight_through():
    """Normal incidence: refraction leaves the direction unchanged, so the pixel behind a centred
    glass ball shows what is behind it except for the Schlick-reflected share of the samples."""
    cam = O.default_camera()
    sc = np.zeros(2, dtype=O.SPHERE_DTYPE)
    sc[0] = ((0.0, 0.0, -1.0), (1.0, 1.0, 1.0), 0.5, O.MAT_GLASS, 0.0)
    sc[1] = ((0.0, 0.0, -4.0), (0.0, 0.0, 0.0), 1.0, O.MAT_DIFFUSE, 0.0)  # black ball behind
    W = H = 33
    _, sums, _ = O.trace_tile(sc, cam, W, H, 2048, 8, O.RNG_COUNTER, seed=2, tile=(16, 16, 1, 1),
                              flags=O.FLAG_DIELECTRIC, want_sums=True)
    # Fresnel at normal incidence for index 1.5: 4% reflected on entry, 4% of the rest on exit, so
    # about 7.8% of the samples end on the (white, horizon) sky and the others on the black ball
    frac = sums[16, 16].astype(np.float64) / (2048 * 255.0)
    assert np.all(frac > 0.055) and np.all(frac < 0.10)
    plain, _, _ = O.trace_tile(sc, cam, W, H, 4, 8, O.RNG_COUNTER, seed=2, tile=(16, 16, 1, 1))
    assert np.all(plain[16, 16] == 255)


def test_float_shading_is_brighter_or_equal_and_close():
    """Dropping the per-level truncation can only raise a sample (every truncation rounds down and
    all factors are non-negative), and by less than one byte per level before amplification; on
    the reference's scene the frames stay within a few counts of each other."""
    sc, cam = O.default_scene(), O.default_camera()
    a, sa, _ = O.trace_tile(sc, cam, 64, 36, 8, 10, O.RNG_COUNTER, seed=3, want_sums=True)
    b, sb, _ = O.trace_tile(sc, cam, 64, 36, 8, 10, O.RNG_COUNTER, seed=3, want_sums=True, flags=O.FLAG_FLOAT_SHADING)
    assert np.all(sb.astype(np.int64) >= sa.astype(np.int64))
    assert not np.array_equal(a, b)
    sky = a[35]  # top row: sky only, one truncation either way
    assert np.array_equal(sky, b[35])
