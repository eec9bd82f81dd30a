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


def _fma(a, b, c):
    """float32 fused multiply-add: the product of two float32 is exact in float64."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def slab_hit(o, d, lo, hi, center, radius2, rmin, rmin2w, t_hi, rcp_bias):
    """The padded slab test of the hierarchy walk (rt_bvh_walk.h), operation by operation in float32.
    rcp_bias perturbs the hardware reciprocal and square root by one unit in the last place."""
    with np.errstate(all="ignore"):
        ox, oy, oz = [(o[:, k] - center[k]).astype(F) for k in range(3)]
        reach2 = (F(2.0) * _fma(oz, oz, _fma(oy, oy, _fma(ox, ox, np.full_like(ox, radius2))))).astype(F)
        k = F(2.0 ** -18 * (1.0 + 2.0 ** -10))
        reff = (np.sqrt(_fma(reach2, np.full_like(ox, k), np.full_like(ox, rmin2w))) * F(1.0 - rcp_bias * 2.0 ** -23)).astype(F)
        pad = (_fma(reff, np.full_like(ox, F(1.00390625 * 2.0 ** -10)), (reff - F(rmin)).astype(F)) * F(1.001)).astype(F)
        dmax = np.maximum(np.maximum(np.abs(d[:, 0]), np.abs(d[:, 1])), np.abs(d[:, 2]))
        dlim = (dmax * F(2.0 ** -50)).astype(F)
        tn = np.full(len(o), -np.inf, dtype=F)
        tf = np.full(len(o), np.inf, dtype=F)
        for kk, ok in enumerate((ox, oy, oz)):
            dk = np.where(np.abs(d[:, kk]) < dlim, np.copysign(dlim, d[:, kk]), d[:, kk]).astype(F)
            inv = ((F(1.0) / dk) * F(1.0 + rcp_bias * 2.0 ** -23)).astype(F)
            nl = (-((ok + pad).astype(F) * inv)).astype(F)
            nh = (-((ok - pad).astype(F) * inv)).astype(F)
            pos = inv >= 0
            zero = np.zeros_like(inv)
            na = np.where(pos, inv, zero); nb = (inv - na).astype(F)
            nc = np.where(pos, nl, nh); fc = np.where(pos, nh, nl)
            t_in = _fma(lo[:, kk], na, _fma(hi[:, kk], nb, nc))
            t_out = _fma(lo[:, kk], nb, _fma(hi[:, kk], na, fc))
            tn = np.maximum(tn, t_in)
            tf = np.minimum(tf, t_out)
        tn = np.maximum(tn, F(0.001 * (1.0 - 2.0 ** -20)))
        tf = np.minimum(tf, t_hi)
        return tn <= tf


def test_padded_box_is_met_wherever_the_reference_accepts():
    """The sphere's own box (the tightest box the hierarchy can hold it in), padded as the walk pads
    it and tested with the walk's arithmetic over [tmin, accepted root]: never a miss."""
    rs = np.random.RandomState(9)
    checked = 0
    for scale, offset in ((1.0, 0.0), (1e-3, 0.0), (30.0, 2e4), (1e4, -3e6)):
        o, d, c, r = grazing_batch(rs, 300000, scale, offset)
        acc, T, _ = reference_roots(o, d, c, r)
        tame = (np.abs(o).sum(axis=1) + np.abs(d).sum(axis=1) < 2.0 ** 28) & ((d.astype(np.float64) ** 2).sum(axis=1) >= 2.0 ** -40)
        sel = acc & tame
        o, d, c, r, T = o[sel], d[sel], c[sel], r[sel], T[sel]
        center = np.array([offset, offset, offset], dtype=F) + F(0.37 * scale)   # any point serves
        rel = c.astype(np.float64) - center.astype(np.float64)
        radius = float(np.sqrt((rel * rel).sum(axis=1)).max()) * (1 + 1e-6)
        radius2 = np.nextafter(F(radius) * F(radius), F(np.inf))
        rmin = float(r.min())
        rmin2w = np.nextafter(F(rmin * rmin * (1 + 2.0 ** -18)), F(np.inf))
        lo = np.nextafter((rel - r[:, None].astype(np.float64)).astype(F), F(-np.inf))
        hi = np.nextafter((rel + r[:, None].astype(np.float64)).astype(F), F(np.inf))
        t_hi = (T * F(1.0 + 2.0 ** -20)).astype(F)
        for bias in (-1.0, 0.0, 1.0):
            hit = slab_hit(o, d, lo, hi, center, radius2, np.nextafter(F(rmin), F(-np.inf)), rmin2w, t_hi, bias)
            assert hit.all(), (scale, offset, bias, int((~hit).sum()))
        checked += len(o)
    assert checked > 500000


def _dot32(a, b):
    return ((a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]).astype(F)


def _tame_accepted(rs, m, scale, offset):
    o, d, c, r = grazing_batch(rs, m, scale, offset)
    acc, T, _ = reference_roots(o, d, c, r)
    tame = (np.abs(o).sum(axis=1) + np.abs(d).sum(axis=1) < 2.0 ** 28) & ((d.astype(np.float64) ** 2).sum(axis=1) >= 2.0 ** -40)
    sel = acc & tame
    return o[sel], d[sel], c[sel], r[sel]


def _widened_r2(r):
    rr = (r * r).astype(F)
    wide = (rr + (rr * F(2.0 ** -16)).astype(F)).astype(F)
    return np.nextafter(wide, F(np.inf))


def test_vector_filter_never_rejects_what_the_reference_accepts():
    """filter_sign (rt_list_scan.h) in float32 with emulated fused multiply-adds: s >= 0 for every
    (ray, sphere) pair the reference's own discriminant accepts."""
    rs = np.random.RandomState(10)
    n = 0
    for scale, offset in ((1.0, 0.0), (1e-3, 0.0), (30.0, 2e4), (1e4, -3e6)):
        o, d, c, r = _tame_accepted(rs, 300000, scale, offset)
        with np.errstate(all="ignore"):
            oc = (o - c).astype(F)
            A = _dot32(d, d)
            a_filter = _fma(np.full_like(A, F(-2.0 ** -17)), A, A)
            dd = _fma(oc[:, 2], d[:, 2], _fma(oc[:, 1], d[:, 1], (oc[:, 0] * d[:, 0]).astype(F)))
            cc = _fma(oc[:, 0], oc[:, 0], _fma(oc[:, 1], oc[:, 1], _fma(oc[:, 2], oc[:, 2], -_widened_r2(r))))
            s = _fma(dd, dd, -(a_filter * cc).astype(F))
        assert not np.signbit(s).any(), (scale, offset, int(np.signbit(s).sum()))
        n += len(o)
    assert n > 500000


def test_matrix_filter_never_rejects_what_the_reference_accepts():
    """The two 4-term products of the matrix-core scan as k-ordered fused chains (what
    v_mfma_f32_16x16x4_f32 computes), with the host's {C - T, K} records: d*d + e >= 0 for every
    accepted pair, for shifts T inside and at the edge of the scene."""
    rs = np.random.RandomState(11)
    n = 0
    for scale, offset in ((1.0, 0.0), (1e-3, 0.0), (30.0, 2e4), (1e4, -3e6)):
        o, d, c, r = _tame_accepted(rs, 300000, scale, offset)
        for shift_at in (0.0, 9.0):
            T = (np.array([offset] * 3) + shift_at * scale).astype(F)
            with np.errstate(all="ignore"):
                cs = (c - T).astype(F)                                    # host: fl(C - shift)
                cs64 = cs.astype(np.float64)
                K = ((cs64 * cs64).sum(axis=1) * (1.0 - 2.0 ** -18) - _widened_r2(r).astype(np.float64)).astype(F)
                os_ = (o - T).astype(F)                                   # device: matrix_operands
                od = _fma(os_[:, 2], d[:, 2], _fma(os_[:, 1], d[:, 1], (os_[:, 0] * d[:, 0]).astype(F)))
                oo = _fma(os_[:, 2], os_[:, 2], _fma(os_[:, 1], os_[:, 1], (os_[:, 0] * os_[:, 0]).astype(F)))
                A = _dot32(d, d)
                ap = _fma(np.full_like(A, F(-2.0 ** -17)), A, A)
                ap2 = (ap + ap).astype(F)
                c2 = (-(ap * oo).astype(F) * F(1.0 - 2.0 ** -18)).astype(F)
                dacc = np.zeros_like(A)
                for k in range(3):
                    dacc = _fma(cs[:, k], -d[:, k], dacc)
                dacc = _fma(np.ones_like(A), od, dacc)
                eacc = c2
                for k in range(3):
                    eacc = _fma(cs[:, k], (ap2 * os_[:, k]).astype(F), eacc)
                eacc = _fma(K, -ap, eacc)
                s = _fma(dacc, dacc, eacc)
            assert not np.signbit(s).any(), (scale, offset, shift_at, int(np.signbit(s).sum()))
            n += len(o)
    assert n > 1000000
