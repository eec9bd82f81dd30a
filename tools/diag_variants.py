"""Builds timing-only variants of the library (tools/librt_diagN.so), each with one part of the
kernel cheapened or doubled, to see what that part costs.  Their images are NOT the reference's;
they are never shipped or tested.  Usage: python tools/diag_variants.py [N ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ray-tracer_amd", "csrc")
FLAGS = ("-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt "
         "-fno-fast-math -fno-slp-vectorize").split()


SOURCES = ["rt_kernels.hip", "rt_arith.h", "rt_list_scan.h", "rt_bvh_walk.h"]


def rep(s, a, b):
    """s: {file name: text}; replaces a by b in the one file that holds it."""
    hits = [k for k in s if a in s[k]]
    assert len(hits) == 1, (a[:60], hits)
    out = dict(s)
    out[hits[0]] = s[hits[0]].replace(a, b)
    return out


def variant(n, s):
    if n == 1:  # a three-multiply hash instead of Philox
        return rep(s, "    uint32_t c3 = 0u;\n#pragma unroll\n    for (int r = 0; r < 10; ++r) {", """    uint32_t c3 = 0u;
    {
        uint32_t h = (c0 * 0x9E3779B1u) ^ (c1 * 0x85EBCA77u) ^ (c2 * 0xC2B2AE3Du) ^ k0;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
        out[0] = h; out[1] = h * 0x9E3779B1u + 0x7F4A7C15u; out[2] = (h ^ (h >> 7)) * 0x85EBCA77u; out[3] = h + k1;
        out[1] ^= out[1] >> 13; out[2] ^= out[2] >> 11;
        return;
    }
#pragma unroll
    for (int r = 0; r < 10; ++r) {""")
    if n == 3:  # no replay of the bounce stack
        return rep(s, "for (int l = level - 1; l >= 0; --l) {", "for (int l = level - 1; l >= level; --l) {")
    if n == 4:  # every ray walks the hierarchy twice
        return rep(s, "                    bvh_begin(a.bvh, ray, 0.001f, w_node, w_sp, w_closest, w_best);", """                    bvh_begin(a.bvh, ray, 0.001f, w_node, w_sp, w_closest, w_best);
                    { int n2 = w_node; uint32_t s2 = 0u; float c2 = w_closest; int b2 = w_best; RayState r2 = ray;
                      uint32_t *stk2 = reinterpret_cast<uint32_t *>(stage) + tid;
                      BvhArgs bb = a.bvh; bb.suspend_lanes = 0u;
                      bvh_walk<false>(bb, nullptr, r2, stk2, n2 != kBvhDone, 0.001f, n2, s2, c2, b2);
                      if (b2 == 12345678) w_closest = c2; }""")
    if n == 7:  # the first point drawn is always accepted
        return rep(s, "        if (((x * x) + (y * y) + (z * z)) < 1.0f)\n            return;", "        return;")
    if n == 8:  # no refill arithmetic: every lane keeps its first pixel (timing only)
        return rep(s, "                const uint32_t s = item / (uint32_t)ch_npx;\n                const uint32_t p = item - s * (uint32_t)ch_npx;\n                const int ly = (int)(p / (uint32_t)ch_tw);\n                const int lx = (int)p - ly * ch_tw;",
                   "                const uint32_t s = item >> 6;\n                const uint32_t p = item & 63u;\n                const int ly = (int)(p >> 3);\n                const int lx = (int)(p & 7u);")
    if n == 9:  # the always-tested spheres are tested twice
        return rep(s, "                    bvh_begin(a.bvh, ray, 0.001f, w_node, w_sp, w_closest, w_best);", """                    { int n2; uint32_t s2; float c2; int b2;
                      bvh_begin(a.bvh, ray, 0.001f, n2, s2, c2, b2);
                      if (b2 == 12345678) w_sp = s2; }
                    bvh_begin(a.bvh, ray, 0.001f, w_node, w_sp, w_closest, w_best);""")
    if n == 11:  # the camera ray of a new path is formed twice
        return rep(s, "                depth = a.max_depth; // >= 1", """                { RayState r2; camera_ray(a.cam, a.width, a.height, a.inv_width, a.inv_height, px + 1, py, (double)unit_float(w[1]),
                             (double)unit_float(w[0]), r2);
                  if (r2.four_a == 12345.0f) ray = r2; }
                depth = a.max_depth; // >= 1""")
    if n == 13:  # the random point is drawn twice
        return rep(s, "        if (__any(wants_point))\n            unit_sphere_points_wave(", """        if (__any(wants_point)) {
            uint32_t c2 = call; float x2 = 0.f, y2 = 0.f, z2 = 0.f;
            unit_sphere_points_wave(wants_point, pixel, sample ^ 0x5555u, c2, a.key0, a.key1, coop, lane, x2, y2, z2);
            if (x2 == 12345.0f) rx = y2 + z2 + (float)c2;
        }
        if (__any(wants_point))
            unit_sphere_points_wave(""")
    if n == 14:  # no per-sample LDS atomics (timing only: the frame stays black)
        return rep(s, "                atomicAdd(acc + acc_at + 0u, (uint32_t)cr);\n                atomicAdd(acc + acc_at + 1u, (uint32_t)cg);\n                atomicAdd(acc + acc_at + 2u, (uint32_t)cb);",
                "                if (cr == 12345) atomicAdd(acc + acc_at + 0u, (uint32_t)(cr + cg + cb));")
    if n == 15:  # hit frame and scatter arithmetic twice
        return rep(s, "                    float ndx, ndy, ndz;\n                    if (material == RT_MAT_DIFFUSE) {", """                    { float p2x, p2y, p2z, n2x, n2y, n2z; bool in2;
                      hit_frame(ray, closest * 0.5f, sp->center[1], sp->center[0], sp->center[2], sp->radius, p2x, p2y, p2z, n2x, n2y, n2z, in2);
                      if (n2x == 12345.0f) px = p2x + p2y + p2z + n2y + n2z; }
                    float ndx, ndy, ndz;
                    if (material == RT_MAT_DIFFUSE) {""")
    raise SystemExit(f"no variant {n}")


def main():
    src = {k: open(os.path.join(CSRC, k)).read() for k in SOURCES}
    procs = []
    for n in [int(x) for x in sys.argv[1:]] or [1, 3, 4, 11, 13, 14, 15]:
        d = f"/tmp/diag/v{n}"
        os.makedirs(d, exist_ok=True)
        for k, text in variant(n, src).items():  # the patched copies shadow the originals (-I order)
            open(os.path.join(d, k), "w").write(text)
        out = os.path.join(ROOT, "tools", f"librt_diag{n}.so")
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", *FLAGS, f"-I{ROOT}/include", f"-I{d}", f"-I{CSRC}", "-shared", "-o", out,
                                       os.path.join(d, "rt_kernels.hip"), "-x", "hip", os.path.join(CSRC, "rt_capi.cpp"),
                                       os.path.join(CSRC, "rt_bvh.cpp")]))
    sys.exit(max(p.wait() for p in procs))


if __name__ == "__main__":
    main()
