// bvh_walk_check.cpp -- CPU restatement of the hierarchy scan, for the tests only.
//
// THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT (like everything under oracle/).  It builds the
// hierarchy with the library's own host code (ray-tracer_amd/csrc/rt_bvh.cpp, compiled into this
// checker) and walks it on the CPU with the arithmetic of the device walk (rt_bvh_walk.h): the
// same per-ray padding, the same plane offsets, the same order-free closest-hit rule.  The tests
// compare what it selects with what the reference's brute-force loop selects
// (IntersectSceneSpheres, Source/main.cpp:528-549, restated in rt_oracle.c), so the tree, the
// padding and the traversal logic are checked without a GPU.
#include <math.h>
#include <stdint.h>

#include <vector>

#include "rt_bvh.h"

using namespace rtamd;

namespace {

struct Ray {
    float ox, oy, oz, dx, dy, dz, four_a, two_a;
    bool tame;
};

// the reference's arithmetic for one sphere (Source/main.cpp:475-499), order-free update
void resolve(const Ray &r, const Geom &s, int idx, float tmin, float &closest, int &best)
{
    const float ocx = r.ox - s.cx, ocy = r.oy - s.cy, ocz = r.oz - s.cz;
    const float d = ocx * r.dx + ocy * r.dy + ocz * r.dz;
    const float B = d + d;
    const float C = (ocx * ocx + ocy * ocy + ocz * ocz) - s.r2;
    const float disc = B * B - r.four_a * C;
    if (disc < 0.0f)
        return;
    const float sq = sqrtf(disc);
    float root = (-B - sq) / r.two_a;
    if (root < tmin)
        root = (-B + sq) / r.two_a;
    if (root >= tmin && (root < closest || (root == closest && idx > best))) {
        closest = root;
        best = idx;
    }
}

int walk(const BvhHost &b, const Ray &r, float tmin, float *t_out)
{
    float closest = INFINITY;
    int best = -1;
    for (uint32_t i = 0; i < b.n_large; ++i)
        resolve(r, b.leaf_geom[i], (int)b.leaf_index[i], tmin, closest, best);
    if (!b.nodes.empty()) {
        const float radius2 = nextafterf(b.radius * b.radius, INFINITY);
        const float ox = r.ox - b.center[0], oy = r.oy - b.center[1], oz = r.oz - b.center[2];
        const float reach2 = 2.0f * fmaf(oz, oz, fmaf(oy, oy, fmaf(ox, ox, radius2)));
        const float reff = sqrtf(fmaf(reach2, 0x1p-18f * (1.0f + 0x1p-10f), b.rmin2w));
        const float pad = fmaf(reff, 0x1.01p-10f, reff - b.rmin) * 1.001f;
        const float dmax = fmaxf(fmaxf(fabsf(r.dx), fabsf(r.dy)), fabsf(r.dz));
        const float dlim = dmax * 0x1p-50f;
        const float dd[3] = { copysignf(fmaxf(fabsf(r.dx), dlim), r.dx), copysignf(fmaxf(fabsf(r.dy), dlim), r.dy),
                              copysignf(fmaxf(fabsf(r.dz), dlim), r.dz) };
        const float o[3] = { ox, oy, oz };
        float na[3], nb[3], nc[3], fc[3];
        for (int k = 0; k < 3; ++k) {
            const float inv = 1.0f / dd[k];
            const float m = -(o[k] * inv);
            na[k] = fmaxf(inv, 0.0f);
            nb[k] = inv - na[k];
            nc[k] = fmaf(-pad, fabsf(inv), m);
            fc[k] = fmaf(pad, fabsf(inv), m);
        }
        const float t_lo = tmin * (1.0f - 0x1p-20f);
        std::vector<int> stack;
        int node = 0;
        for (;;) {
            if (node >= 0) {
                const BvhNode &nd = b.nodes[(size_t)node];
                const float t_hi = closest * (1.0f + 0x1p-20f);
                float n0 = t_lo, f0 = t_hi, n1 = t_lo, f1 = t_hi;
                for (int k = 0; k < 3; ++k) {
                    n0 = fmaxf(n0, fmaf(nd.lo0[k], na[k], fmaf(nd.hi0[k], nb[k], nc[k])));
                    f0 = fminf(f0, fmaf(nd.lo0[k], nb[k], fmaf(nd.hi0[k], na[k], fc[k])));
                    n1 = fmaxf(n1, fmaf(nd.lo1[k], na[k], fmaf(nd.hi1[k], nb[k], nc[k])));
                    f1 = fminf(f1, fmaf(nd.lo1[k], nb[k], fmaf(nd.hi1[k], na[k], fc[k])));
                }
                const bool hit0 = n0 <= f0, hit1 = n1 <= f1;
                if (hit0 && hit1) {
                    const bool first0 = n0 <= n1;
                    stack.push_back(first0 ? nd.child1 : nd.child0);
                    node = first0 ? nd.child0 : nd.child1;
                    continue;
                }
                if (hit0) { node = nd.child0; continue; }
                if (hit1) { node = nd.child1; continue; }
            } else {
                const uint32_t code = (uint32_t)~node, first = code >> 4, count = code & 15u;
                for (uint32_t i = first; i < first + count; ++i)
                    resolve(r, b.leaf_geom[i], (int)b.leaf_index[i], tmin, closest, best);
            }
            if (stack.empty())
                break;
            node = stack.back();
            stack.pop_back();
        }
    }
    if (t_out)
        *t_out = closest;
    return best;
}

} // namespace

// rays = n_rays x {origin[3], direction[3]}; index[i] = the sphere the walk selects or -1, t[i] its
// root (may be NULL); index[i] = -2 for rays outside the walk's preconditions (the device scans
// those with the reference's in-order loop instead).  Returns the number of nodes of the tree.
extern "C" int orc_bvh_intersect(const rt_sphere *spheres, uint32_t n, int leaf_spheres, const float *rays,
                                 uint32_t n_rays, int32_t *index, float *t)
{
    std::vector<Geom> exact(n);
    for (uint32_t i = 0; i < n; ++i) {
        volatile float rr = spheres[i].radius * spheres[i].radius;
        exact[i] = Geom{ spheres[i].center[0], spheres[i].center[1], spheres[i].center[2], rr };
    }
    BvhHost b;
    bvh_build(spheres, exact.data(), n, leaf_spheres, b);
    for (uint32_t i = 0; i < n_rays; ++i) {
        const float *p = rays + (size_t)i * 6u;
        Ray r;
        r.ox = p[0]; r.oy = p[1]; r.oz = p[2];
        r.dx = p[3]; r.dy = p[4]; r.dz = p[5];
        const float A = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
        r.four_a = 4.0f * A;
        r.two_a = 2.0f * A;
        const float m = ((fabsf(r.ox) + fabsf(r.oy)) + (fabsf(r.oz) + fabsf(r.dx))) + (fabsf(r.dy) + fabsf(r.dz));
        r.tame = (m < 0x1p28f) && (A >= 0x1p-30f);
        if (!r.tame) {
            index[i] = -2;
            if (t) t[i] = 0.0f;
            continue;
        }
        index[i] = walk(b, r, 0.001f, t ? t + i : nullptr);
    }
    return (int)b.nodes.size();
}
