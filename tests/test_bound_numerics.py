"""Numerical check of the inequality the box hierarchy's padding rests on (rt_bvh_walk.h): whenever
the reference's float arithmetic (Source/main.cpp:469-500) accepts a sphere, the point O + T D of the
accepted root lies within  r_eff = sqrt(r^2 (1 + 64u) + 64u |O - C|^2),  u = 2^-24,  of the centre.
The reference's operations are repeated in float32 with numpy (same order, no fused multiply-add),
the point is then evaluated in float64.  Rays are aimed to graze: that is where the rounded
discriminant is least reliable and the root most sensitive."""
import numpy as np

F = np.float32
U = 2.0 ** -24


def reference_roots(o, d, c, r):
    """RaySphereIntersectionTest on float32 arrays: accepted mask and root T (tmin 0.001, tmax inf)."""
    with np.errstate(all="ignore"):  # a few of the scaled-up rays overflow; they are not accepted
        return _reference_roots(o, d, c, r)


def _reference_roots(o, d, c, r):
    oc = (o - c).astype(F)
    A = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(F)
    dot = ((oc[:, 0] * d[:, 0] + oc[:, 1] * d[:, 1]) + oc[:, 2] * d[:, 2]).astype(F)
    B = (F(2.0) * dot).astype(F)
    C = (((oc[:, 0] * oc[:, 0] + oc[:, 1] * oc[:, 1]) + oc[:, 2] * oc[:, 2]).astype(F) - (r * r).astype(F)).astype(F)
    disc = ((B * B).astype(F) - ((F(4.0) * A).astype(F) * C).astype(F)).astype(F)
    ok = ~(disc < 0)
    sq = np.sqrt(np.where(ok, disc, F(0))).astype(F)
    two_a = (F(2.0) * A).astype(F)
    t1 = ((-B - sq).astype(F) / two_a).astype(F)
    t2 = ((-B + sq).astype(F) / two_a).astype(F)
    use1 = ~((t1 < F(0.001)) | ~np.isfinite(t1))
    T = np.where(use1, t1, t2)
    acc = ok & ~(T < F(0.001)) & np.isfinite(T)
    return acc, T, oc


def grazing_batch(rs, m, scale, offset):
    c = (rs.uniform(-10, 10, size=(m, 3)) * scale + offset).astype(F)
    r = (10.0 ** rs.uniform(-2, 0.5, size=m) * scale).astype(F)
    cd, rd = c.astype(np.float64), r.astype(np.float64)
    dist = rd * 10.0 ** rs.uniform(0.05, 4.0, size=m)
    u = rs.normal(size=(m, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    o = cd + u * dist[:, None]
    v = np.cross(u, rs.normal(size=(m, 3))); v /= np.linalg.norm(v, axis=1, keepdims=True)
    eps = rs.choice([0.0, 3e-8, -3e-8, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-3, -0.5], size=m)
    foot = cd + v * (rd * (1.0 + eps))[:, None]
    d = (foot - o) * (10.0 ** rs.uniform(-3, 3, size=m))[:, None]
    return o.astype(F), d.astype(F), c, r


def test_accepted_root_lies_within_the_effective_radius():
    rs = np.random.RandomState(7)
    worst = 0.0
    accepted = 0
    for scale, offset in ((1.0, 0.0), (1e-3, 0.0), (30.0, 2e4), (1e4, -3e6), (1.0, 1e5)):
        o, d, c, r = grazing_batch(rs, 400000, scale, offset)
        acc, T, oc = reference_roots(o, d, c, r)
        accepted += int(acc.sum())
        ocd, dd, Td, rd = oc.astype(np.float64), d.astype(np.float64), T.astype(np.float64), r.astype(np.float64)
        P = ocd + Td[:, None] * dd                      # relative to the centre, origin C + oc
        dist2 = (P * P).sum(axis=1)
        q = (ocd * ocd).sum(axis=1)
        excess = (dist2 - rd * rd) / (U * (q + rd * rd))  # in units of u (q + r^2)
        worst = max(worst, float(excess[acc].max()))
        assert np.all(dist2[acc] <= rd[acc] ** 2 * (1 + 64 * U) + 64 * U * q[acc])
    assert accepted > 400000
    # the analysis gives 41 (q) / 25 (r^2) and the padding assumes 64 for both; measured: about 9
    assert worst < 20.0, worst


def test_accepted_line_passes_within_the_discriminant_bound():
    """The weaker statement used for boxes the ray only passes: acceptance implies the line is within
    sqrt(r^2 + u (19.8 q + 5.2 r^2)) of the centre."""
    rs = np.random.RandomState(8)
    for scale, offset in ((1.0, 0.0), (1e3, 5e5)):
        o, d, c, r = grazing_batch(rs, 400000, scale, offset)
        acc, T, oc = reference_roots(o, d, c, r)
        ocd, dd, rd = oc.astype(np.float64), d.astype(np.float64), r.astype(np.float64)
        q = (ocd * ocd).sum(axis=1)
        h = (ocd * dd).sum(axis=1)
        line2 = q - h * h / (dd * dd).sum(axis=1)
        assert np.all(line2[acc] <= rd[acc] ** 2 + U * (19.8 * q[acc] + 5.2 * rd[acc] ** 2))
