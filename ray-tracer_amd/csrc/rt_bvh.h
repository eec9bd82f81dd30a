// rt_bvh.h -- bounding-volume hierarchy over the sphere list (host-side build).
//
// The reference tests every sphere for every ray (IntersectSceneSpheres,
// Source/main.cpp:534).  The hierarchy only decides which spheres a ray CANNOT
// be accepted for; the survivors go through the reference's own arithmetic, so
// the image stays bit-identical.  Everything here is plain C++ (no HIP).
#pragma once

#include <stdint.h>

#include <string>
#include <vector>

#include "rt_device.h"

namespace rtamd {

// One inner node: the boxes of its two children, in coordinates relative to
// BvhHost::center, rounded outward.  A child reference >= 0 is a node index; a
// negative one is ~((first << 4) | count): `count` (0..15) consecutive records
// of leaf_geom / leaf_index starting at `first`.
struct BvhNode {
    float lo0[3]; int32_t child0;
    float hi0[3]; int32_t child1;
    float lo1[3]; uint32_t unused0;
    float hi1[3]; uint32_t unused1;
};
static_assert(sizeof(BvhNode) == 64, "four 16-byte loads per node");

struct BvhHost {
    std::vector<BvhNode> nodes;       // nodes[0] is the root; empty if no sphere is in the tree
    std::vector<Geom> leaf_geom;      // n_large always-tested records, then the leaves' records
    std::vector<uint32_t> leaf_index; // list index of each record
    uint32_t n_large = 0;             // spheres too big for a useful box: tested for every ray
    uint32_t levels = 0;              // depth of the deepest leaf = traversal stack entries needed
    float center[3] = { 0.f, 0.f, 0.f };
    float radius = 0.f;               // >= |C - center| for every sphere in the tree
    float rmin = 0.f;                 // <= |R| for every sphere in the tree
    float rmin2w = 0.f;               // >= rmin^2 (1 + 2^-18)
};

constexpr int kBvhLeafMax = 15;

// Builds the hierarchy for spheres[0..n) (`exact` holds their {C, fl(R*R)} records).
// The scene must be tame (finite, magnitudes below 2^28).  leaf_max in 1..15.
void bvh_build(const rt_sphere *spheres, const Geom *exact, uint32_t n, int leaf_max, BvhHost &out);

// Structural check used by the self-test entry point: every sphere is in exactly
// one leaf or in the always-tested list, every child box contains the boxes of the
// spheres below it, depth matches `levels`.  Returns an empty string when sound.
std::string bvh_validate(const rt_sphere *spheres, uint32_t n, const BvhHost &b);

} // namespace rtamd
