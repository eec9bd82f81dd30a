// rt_bvh.cpp -- host-side build of the sphere hierarchy (see rt_bvh.h).
//
// Top-down, binned surface-area heuristic on the sphere centres (16 bins per
// axis), falling back to an even split in list order along the widest axis when
// the bins cannot separate the set or the tree gets deep; so depth is bounded by
// kSahDepth + log2(n) whatever the input.  Spheres far larger than the typical
// one (a ground sphere of radius 1000 among marbles) get no box: they are put on
// a short list that every ray tests directly.
#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace rtamd {
namespace {

constexpr int kBins = 16;
constexpr int kSahDepth = 28;    // below this depth: even splits only
constexpr uint32_t kMaxLarge = 16;
constexpr double kLargeRatio = 4.0; // "large" = radius above 4 x the median radius (measured: 8 is 5 % slower)

struct Box {
    float lo[3], hi[3];
    void clear()
    {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::numeric_limits<float>::infinity();
            hi[k] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const Box &b)
    {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    double half_area() const
    {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return x * y + y * z + z * x;
    }
};

float round_down(double v)
{
    float f = (float)v;
    if ((double)f > v)
        f = std::nextafter(f, -std::numeric_limits<float>::infinity());
    return f;
}

float round_up(double v)
{
    float f = (float)v;
    if ((double)f < v)
        f = std::nextafter(f, std::numeric_limits<float>::infinity());
    return f;
}

struct Item {
    uint32_t index;   // list index of the sphere
    Box box;          // relative to the tree's centre, rounded outward
    double cen[3];    // centre relative to the tree's centre
};

struct Builder {
    const Geom *exact;
    int leaf_max;
    std::vector<Item> items;
    BvhHost *out;

    int32_t emit_leaf(uint32_t first, uint32_t count, uint32_t depth)
    {
        const uint32_t at = (uint32_t)out->leaf_geom.size();
        for (uint32_t i = 0; i < count; ++i) {
            out->leaf_geom.push_back(exact[items[first + i].index]);
            out->leaf_index.push_back(items[first + i].index);
        }
        out->levels = std::max(out->levels, depth);
        return ~(int32_t)((at << 4) | count);
    }

    // Returns the child reference of the subtree over items[first, first+count) and its box.
    int32_t build(uint32_t first, uint32_t count, uint32_t depth, Box &box)
    {
        box.clear();
        Box cb; // bounds of the centres
        cb.clear();
        for (uint32_t i = first; i < first + count; ++i) {
            box.grow(items[i].box);
            for (int k = 0; k < 3; ++k) {
                cb.lo[k] = std::min(cb.lo[k], (float)items[i].cen[k]);
                cb.hi[k] = std::max(cb.hi[k], (float)items[i].cen[k]);
            }
        }
        if (count <= (uint32_t)leaf_max)
            return emit_leaf(first, count, depth);

        uint32_t mid = 0;
        if (depth < (uint32_t)kSahDepth)
            mid = sah_partition(first, count, cb);
        if (mid == 0 || mid == count) {
            // even split along the widest axis of the centres
            int axis = 0;
            for (int k = 1; k < 3; ++k)
                if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis])
                    axis = k;
            mid = count / 2;
            std::nth_element(items.begin() + first, items.begin() + first + mid, items.begin() + first + count,
                             [axis](const Item &a, const Item &b) {
                                 return a.cen[axis] < b.cen[axis] || (a.cen[axis] == b.cen[axis] && a.index < b.index);
                             });
        }
        const uint32_t me = (uint32_t)out->nodes.size();
        out->nodes.emplace_back();
        Box b0, b1;
        const int32_t c0 = build(first, mid, depth + 1, b0);
        const int32_t c1 = build(first + mid, count - mid, depth + 1, b1);
        BvhNode &nd = out->nodes[me];
        memset(&nd, 0, sizeof nd);
        for (int k = 0; k < 3; ++k) {
            nd.lo0[k] = b0.lo[k]; nd.hi0[k] = b0.hi[k];
            nd.lo1[k] = b1.lo[k]; nd.hi1[k] = b1.hi[k];
        }
        nd.child0 = c0;
        nd.child1 = c1;
        return (int32_t)me;
    }

    // Binned SAH: reorders items[first, first+count) and returns the size of the
    // left part, or 0 when no bin boundary separates the set.
    // What a side of a split costs is its box's area times its spheres -- or, where leaves hold more than
    // four, times its groups of four: the walk tests a leaf's records four at a time, so a leaf of five
    // costs what a leaf of eight does (10 000 spheres in leaves of up to 8: 13.25 -> 12.87 ms per
    // 1080p / 64 spp frame; leaves of up to 4 are one group whatever they hold, and keep the count).
    uint32_t sah_partition(uint32_t first, uint32_t count, const Box &cb)
    {
        const bool by_groups = leaf_max > 4;
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1, best_bin = -1;
        for (int axis = 0; axis < 3; ++axis) {
            const double lo = cb.lo[axis], ext = (double)cb.hi[axis] - lo;
            if (!(ext > 0.0))
                continue;
            Box bb[kBins];
            uint32_t bn[kBins];
            for (int b = 0; b < kBins; ++b) { bb[b].clear(); bn[b] = 0; }
            const double scale = kBins / ext;
            for (uint32_t i = first; i < first + count; ++i) {
                int b = (int)((items[i].cen[axis] - lo) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                bb[b].grow(items[i].box);
                ++bn[b];
            }
            double right_area[kBins];
            uint32_t right_n[kBins];
            Box acc;
            acc.clear();
            uint32_t n_acc = 0;
            for (int b = kBins - 1; b > 0; --b) {
                if (bn[b]) acc.grow(bb[b]);
                n_acc += bn[b];
                right_area[b] = n_acc ? acc.half_area() : 0.0;
                right_n[b] = n_acc;
            }
            acc.clear();
            n_acc = 0;
            for (int b = 0; b < kBins - 1; ++b) { // split between bin b and b + 1
                if (bn[b]) acc.grow(bb[b]);
                n_acc += bn[b];
                if (n_acc == 0 || right_n[b + 1] == 0)
                    continue;
                const double cost = by_groups ? acc.half_area() * ((n_acc + 3u) / 4u) + right_area[b + 1] * ((right_n[b + 1] + 3u) / 4u)
                                              : acc.half_area() * n_acc + right_area[b + 1] * right_n[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_bin = b;
                }
            }
        }
        if (best_axis < 0)
            return 0;
        const double lo = cb.lo[best_axis], ext = (double)cb.hi[best_axis] - lo, scale = kBins / ext;
        const int axis = best_axis, bin = best_bin;
        auto it = std::partition(items.begin() + first, items.begin() + first + count, [=](const Item &a) {
            int b = (int)((a.cen[axis] - lo) * scale);
            b = std::min(std::max(b, 0), kBins - 1);
            return b <= bin;
        });
        return (uint32_t)(it - (items.begin() + first));
    }
};

} // namespace

void bvh_build(const rt_sphere *spheres, const Geom *exact, uint32_t n, int leaf_max, BvhHost &out)
{
    out = BvhHost();
    if (n == 0)
        return;
    leaf_max = std::min(std::max(leaf_max, 1), kBvhLeafMax);

    // spheres without a box: the largest ones, if they dwarf the median
    std::vector<float> radii(n);
    for (uint32_t i = 0; i < n; ++i)
        radii[i] = std::fabs(spheres[i].radius);
    std::vector<float> sorted(radii);
    std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
    const double big = kLargeRatio * (double)sorted[n / 2];
    std::vector<uint32_t> large;
    for (uint32_t i = 0; i < n; ++i)
        if ((double)radii[i] > big)
            large.push_back(i);
    if (large.size() > kMaxLarge) { // keep the biggest, ties by list order
        std::stable_sort(large.begin(), large.end(), [&](uint32_t a, uint32_t b) { return radii[a] > radii[b]; });
        large.resize(kMaxLarge);
    }
    std::sort(large.begin(), large.end());
    std::vector<char> is_large(n, 0);
    for (uint32_t i : large) {
        is_large[i] = 1;
        out.leaf_geom.push_back(exact[i]);
        out.leaf_index.push_back(i);
    }
    out.n_large = (uint32_t)large.size();
    const uint32_t n_tree = n - out.n_large;
    if (n_tree == 0)
        return;

    // centre of the tree's coordinates: middle of the centres' bounds, as a float
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    for (uint32_t i = 0; i < n; ++i) {
        if (is_large[i]) continue;
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], (double)spheres[i].center[k]);
            hi[k] = std::max(hi[k], (double)spheres[i].center[k]);
        }
    }
    for (int k = 0; k < 3; ++k)
        out.center[k] = (float)(0.5 * (lo[k] + hi[k]));

    Builder b;
    b.exact = exact;
    b.leaf_max = leaf_max;
    b.out = &out;
    b.items.reserve(n_tree);
    double rad2 = 0.0, rmin = 1e300;
    for (uint32_t i = 0; i < n; ++i) {
        if (is_large[i]) continue;
        Item it;
        it.index = i;
        const double r = radii[i];
        double d2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            it.cen[k] = (double)spheres[i].center[k] - (double)out.center[k]; // exact in double
            it.box.lo[k] = round_down(it.cen[k] - r);
            it.box.hi[k] = round_up(it.cen[k] + r);
            d2 += it.cen[k] * it.cen[k];
        }
        rad2 = std::max(rad2, d2);
        rmin = std::min(rmin, r);
        b.items.push_back(it);
    }
    out.radius = round_up(std::sqrt(rad2) * (1.0 + 1e-9));
    out.rmin = round_down(rmin);
    out.rmin2w = round_up(rmin * rmin * (1.0 + 0x1p-18));

    Box root_box;
    const int32_t root = b.build(0, n_tree, 0, root_box);
    if (root < 0) {
        // the whole tree is one leaf: give it an inner node above (second child: an empty leaf)
        BvhNode nd;
        memset(&nd, 0, sizeof nd);
        for (int k = 0; k < 3; ++k) {
            nd.lo0[k] = nd.lo1[k] = root_box.lo[k];
            nd.hi0[k] = nd.hi1[k] = root_box.hi[k];
        }
        nd.child0 = root;
        nd.child1 = ~(int32_t)0;
        out.nodes.push_back(nd);
        out.levels = 1;
    }
}

std::string bvh_validate(const rt_sphere *spheres, uint32_t n, const BvhHost &b)
{
    if (b.leaf_geom.size() != n || b.leaf_index.size() != n)
        return "record count differs from the sphere count";
    std::vector<char> seen(n, 0);
    for (uint32_t i = 0; i < b.n_large; ++i) {
        if (b.leaf_index[i] >= n || seen[b.leaf_index[i]]++)
            return "always-tested list holds a bad or repeated index";
    }
    if (b.nodes.empty())
        return b.n_large == n ? "" : "no tree although spheres remain";

    struct Frame { int32_t ref; Box box; uint32_t depth; bool has_box; };
    std::vector<Frame> todo;
    Frame root;
    root.ref = 0; root.depth = 0; root.has_box = false;
    root.box.clear();
    todo.push_back(root);
    uint32_t deepest = 0, visited_nodes = 0;
    while (!todo.empty()) {
        const Frame f = todo.back();
        todo.pop_back();
        if (f.ref >= 0) {
            if ((size_t)f.ref >= b.nodes.size())
                return "child index out of range";
            if (++visited_nodes > b.nodes.size())
                return "node reachable twice";
            const BvhNode &nd = b.nodes[f.ref];
            Frame c0, c1;
            c0.ref = nd.child0; c1.ref = nd.child1;
            c0.depth = c1.depth = f.depth + 1;
            c0.has_box = c1.has_box = true;
            for (int k = 0; k < 3; ++k) {
                c0.box.lo[k] = nd.lo0[k]; c0.box.hi[k] = nd.hi0[k];
                c1.box.lo[k] = nd.lo1[k]; c1.box.hi[k] = nd.hi1[k];
                if (f.has_box && (nd.lo0[k] < f.box.lo[k] || nd.hi0[k] > f.box.hi[k] ||
                                  nd.lo1[k] < f.box.lo[k] || nd.hi1[k] > f.box.hi[k]))
                    return "child box sticks out of its parent's";
            }
            todo.push_back(c0);
            todo.push_back(c1);
        } else {
            const uint32_t code = (uint32_t)~f.ref, first = code >> 4, count = code & 15u;
            if (count && (first < b.n_large || (size_t)first + count > b.leaf_geom.size()))
                return "leaf range out of bounds";
            deepest = std::max(deepest, f.depth);
            for (uint32_t i = first; i < first + count; ++i) {
                const uint32_t idx = b.leaf_index[i];
                if (idx >= n || seen[idx]++)
                    return "sphere missing from or repeated in the leaves";
                const rt_sphere &s = spheres[idx];
                if (memcmp(&b.leaf_geom[i].cx, s.center, 3 * sizeof(float)) != 0)
                    return "leaf record does not match its sphere";
                const double r = std::fabs((double)s.radius);
                if (r < (double)b.rmin)
                    return "rmin above a radius";
                double d2 = 0.0;
                for (int k = 0; k < 3; ++k) {
                    const double c = (double)s.center[k] - (double)b.center[k];
                    d2 += c * c;
                    if ((double)f.box.lo[k] > c - r || (double)f.box.hi[k] < c + r)
                        return "sphere sticks out of its leaf's box";
                }
                if (std::sqrt(d2) > (double)b.radius)
                    return "radius of the tree below a centre's distance";
            }
        }
    }
    for (uint32_t i = 0; i < n; ++i)
        if (!seen[i])
            return "sphere in no leaf";
    if (visited_nodes != b.nodes.size())
        return "unreachable node";
    if (deepest != b.levels)
        return "levels does not equal the deepest leaf's depth";
    return "";
}

} // namespace rtamd
