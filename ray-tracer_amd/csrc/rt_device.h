// rt_device.h -- kernel argument blocks and launch wrappers shared by the HIP
// kernels (rt_kernels.hip) and the C ABI (rt_capi.cpp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_mi355x.h"

namespace rtamd {

// Threads per workgroup of the trace kernel: 4 waves, one per SIMD of a CU.
constexpr int kTraceThreads = 256;
// Atomics on one address complete at only 40 to 80 million a second over the whole chip, and a launch
// has thousands of waves.  So the chunk queue has kQueues heads in separate 256-byte blocks (head q
// hands out chunks q, q + n_queues, ...; a launch uses n_queues of them, a power of two not above its
// number of workgroups, so that every head has workgroups of its own), a wave that finds its head
// dry tries only the next kQueueProbes - 1 heads before it retires (every head is drained by its own
// waves; chunks are dealt to heads round robin, so the heads carry equal samples of the frame), and
// the counters the waves add to at their end are kCounterSlots copies, summed on the host.
constexpr uint32_t kQueues = 64;
constexpr uint32_t kQueueStride = 64; // words between heads
constexpr uint32_t kQueueProbes = 4;
constexpr uint32_t kCounterSlots = 64;
constexpr uint32_t kCounterWords = 64; // 64-bit counters per slot
// Pixel tile one wave draws its samples from (also rt_params' tile unit).
constexpr int kTileW = 8;
constexpr int kTileH = 8;
// Spheres tested per candidate mask word.
constexpr int kBlock = 32;

// Geometry record streamed through LDS: centre and a squared radius (the exact
// fl(r*r) for the exact filter, r*r*(1 + 2^-16) for the conservative one).
struct Geom {
    float cx, cy, cz, r2;
};

// How the scan decides which spheres need the exact root evaluation.
enum {
    kFilterExact = 0,        // the reference's own discriminant, unfused, on the vector ALU
    kFilterConservative = 1, // fused, widened test on the vector ALU: superset of the exact candidates
    kFilterMatrix = 2,       // the widened test as two 4-term products on the FP32 matrix cores
    kFilterBvh = 3,          // per-ray walk of a box hierarchy; the spheres in the leaves reached are candidates
    // Not selectable: what kFilterConservative becomes for a list of at most kShortList spheres (the
    // reference's own scene has four) -- the reference's loop as it stands, Source/main.cpp:534-546,
    // over records read with scalar loads, in a kernel of its own that holds no tile and none of the
    // block filter's registers, so that seven waves per SIMD fit instead of four.
    kFilterShort = 4
};
constexpr uint32_t kShortList = 12u;

// Most samples of a pixel in one work-queue chunk: a wave keeps a chunk's per-pixel sums in 16 bits per
// channel, and a sample adds at most 255.
constexpr uint32_t kSegmentSamplesMax = 256;

// A tree is walked in LDS only if its records number at most this many: the walk's stack then holds
// child references in two bytes, and a leaf reference ~(first << 4 | count) must fit 16 signed bits.
constexpr uint32_t kBvhLdsRecords = 2048;

// Device view of the hierarchy (see rt_bvh.h for the encoding).
struct BvhArgs {
    const float4 *nodes;        // four float4 per node
    const Geom *leaf_geom;      // n_large always-tested records, then the leaves' records, then 3 of padding
    const uint32_t *leaf_index; // list index of each record
    uint32_t n_nodes, n_large;
    uint32_t levels;            // traversal stack entries per ray (>= 1)
    uint32_t lds_nodes;         // n_nodes if the kernel keeps the nodes in LDS, else 0
    uint32_t suspend_lanes;     // a wave leaves the walk when at most this many lanes are still in it
    float center[3];            // origin of the boxes' coordinates
    float radius2, rmin, rmin2w; // see BvhHost (radius2 >= radius^2)
};

// Most jobs one trace launch carries (rt_accumulate_batch_device).
constexpr uint32_t kMaxJobs = 8;

// One job of a trace launch: a camera, a frame, a range of samples and a set of tiles of a region, added
// into the job's own sums.  The jobs of a launch share the scene, the scan, the generator's key and the
// flags; their chunks form one work queue, job after job, so that waves which find a job's chunks all
// handed out go on with the next job's while that job's last paths are still bouncing elsewhere.
struct TraceJob {
    rt_camera cam;
    uint32_t width, height;
    double inv_width, inv_height; // 1/(double)(float)width, 1/(double)(float)height, correctly rounded
    double width_d, height_d;     // (double)(float)width, (double)(float)height
    uint32_t sample_begin, sample_count;
    int32_t max_depth;
    // The chunks' sums are added to `sums` with atomics: the samples are cut into segments, or another
    // job of the launch adds into the same memory.  Else with plain loads and stores (the launch's
    // only writer of those pixels is the wave that owns the chunk).
    uint32_t atomic_flush;
    int32_t x0, y0, w, h;      // traced region
    uint32_t tiles_x, tiles_y; // 8x8 tiles of the region
    uint32_t tile_first, tile_stride; // the job's tiles: tile_first + k*tile_stride
    // Work queue: a tile is cut into `subtiles` strips of 8/subtiles rows (1, 2 or 4; strips touch
    // the same memory lines of `sums` as the whole tile) and its sample range into `segments` runs of seg_samples samples (the last
    // may be shorter, none is empty).  The job's chunks are chunk_first .. chunk_first + n_chunks - 1 of
    // the launch; its chunk c is tile k = c / (subtiles*segments), strip (c / segments) % subtiles,
    // segment c % segments.  A strip clipped away by the region's edge is an empty chunk.
    uint32_t chunk_first, n_chunks;
    uint32_t subtiles, segments, seg_samples;
    uint32_t reserved;
    uint32_t *sums;            // full frame uint32 x3
};

struct TraceArgs {
    const Geom *geom;          // phase-1 records for `filter`, padded to a multiple of 128
    const Geom *exact;         // {centre, fl(r*r)}, same padding
    const rt_sphere *spheres;  // n records, the caller's list
    uint32_t n;
    int32_t filter;
    float shift[3];            // origin of the matrix filter's coordinates
    uint32_t tile_cap;         // spheres per LDS tile, multiple of kBlock (0: no tiles)
    uint32_t stack_entry_bytes; // bounce stack entry: 2 while n <= 65536, else 4
    uint32_t aux_bytes;        // LDS between the tiles and the bounce stack: operand staging or traversal stack
    BvhArgs bvh;               // kFilterBvh only
    int32_t max_depth;         // the largest of the jobs' (sizes the bounce stack in LDS)
    uint32_t flags;            // RT_FLAG_*
    uint32_t count_work;       // launch the counting build: tallies of executed work go to rays[2..4]
    uint32_t key0, key1;
    uint32_t n_jobs;           // 1 .. kMaxJobs
    uint32_t n_chunks;         // of all jobs
    uint32_t n_queues;         // heads of the work queue in use: a power of two <= min(kQueues, workgroups)
    // Chunks of very few samples: every path adds its colour to `sums` itself (three atomics) instead of
    // through the wave's two per-chunk accumulators, so a wave never waits for a chunk to drain
    // before it may pull the next one.  Single-job launches only.
    uint32_t direct_add;
    unsigned long long *rays;  // kCounterSlots x kCounterWords counters; a workgroup adds to slot blockIdx % kCounterSlots:
                               // [0] rays, [1] wave passes over the sphere list; counting build: [2] hierarchy
                               // nodes visited, [3] exact discriminants, [4] root evaluations, [8..] section tallies
    // Work queue heads, zeroed before launch: kQueues counters kQueueStride words apart; queue q
    // hands out chunks q, q + kQueues, q + 2 kQueues, ...
    uint32_t *chunk_counter;
    TraceJob jobs[kMaxJobs];
};

struct ResolveArgs {
    const uint32_t *sums;
    uint8_t *rgb;
    uint32_t width, height, spp_div, flags;
    int32_t x0, y0, w, h;
};

// Tile-major exchange buffers of a tile partition (pack_tiles_kernel, unpack_tiles_kernel).
struct TilesArgs {
    const uint32_t *sums;      // pack: full-frame sums
    uint8_t *tiles;            // pack: n_own x 192 bytes out
    const uint8_t *tiles_in;   // unpack: n_blocks blocks of block_bytes
    uint8_t *rgb;              // unpack: the frame
    uint32_t width, height, tiles_x, tiles_y;
    uint32_t spp_div, flags;
    uint32_t tile_first, tile_stride, n_own; // pack: this device's tiles
    uint32_t n_blocks;         // unpack: devices
    size_t block_bytes;        // unpack: bytes between two devices' blocks
};

// dst[i] += src[0][i] + ... + src[n_src - 1][i]: the first device of a sample-sliced frame adds the
// other devices' sums, read in place (peer access or the same device).
struct AddArgs {
    uint32_t *dst;
    const uint32_t *src[15];
    uint32_t n_src;
    size_t count;
};

struct HitsArgs {
    const Geom *geom;          // conservative filter records, or null for the exact scan
    const Geom *exact;
    uint32_t n;
    rt_camera cam;
    uint32_t width, height;
    double inv_width, inv_height; // as in TraceArgs
    int32_t *hits;
    uint32_t use_bvh;          // walk `bvh` instead of scanning the list
    BvhArgs bvh;
};

struct IntersectArgs {
    const Geom *geom;          // conservative filter records, or null for the exact scan
    const Geom *exact;
    const rt_sphere *spheres;
    uint32_t n;
    const float *rays;
    uint32_t n_rays;
    uint8_t *out; // 32 bytes per ray
    uint32_t use_bvh;          // walk `bvh` instead of scanning the list
    BvhArgs bvh;
};

// rt_render_device_static's scratch: a file-scope device array (rt_kernels.hip) holding the
// colour sums of a frame of up to kStaticMaxPixels pixels, then the work queue's heads.
constexpr size_t kStaticMaxPixels = RT_STATIC_MAX_PIXELS;
constexpr size_t kStaticSumsBytes = kStaticMaxPixels * 3u * sizeof(uint32_t);
constexpr size_t kStaticQueueBytes = kQueues * kQueueStride * sizeof(uint32_t);
// Device address of that array on the CURRENT device (hipGetSymbolAddress).
hipError_t static_scratch_address(void **ptr);

// Bytes of LDS between the tiles and the bounce stack for `filter`.
uint32_t trace_aux_bytes(int filter, uint32_t bvh_levels, uint32_t bvh_lds_nodes);
// Bytes of dynamic LDS the trace kernel needs.
size_t trace_lds_bytes(int filter, uint32_t tile_cap, uint32_t aux_bytes, int32_t max_depth, uint32_t stack_entry_bytes);

hipError_t launch_trace(const TraceArgs &a, uint32_t n_workgroups, hipStream_t stream);
hipError_t launch_resolve(const ResolveArgs &a, hipStream_t stream);
// Zeroes `bytes` bytes at `dst` (both multiples of 16) with a kernel: what a captured stream uses instead of a memset.
hipError_t launch_zero(void *dst, size_t bytes, hipStream_t stream);
hipError_t launch_pack_tiles(const TilesArgs &a, hipStream_t stream);
hipError_t launch_unpack_tiles(const TilesArgs &a, hipStream_t stream);
hipError_t launch_add_sums(const AddArgs &a, hipStream_t stream);
hipError_t launch_primary_hits(const HitsArgs &a, hipStream_t stream);
hipError_t launch_intersect(const IntersectArgs &a, hipStream_t stream);
// rt_set_scene for a context of a multi-device object: when `donor` (may be null) already holds the
// hierarchy for this very list under the same options, its host copy is reused instead of built again.
int set_scene_sharing_hierarchy(rt_context *ctx, const rt_sphere *spheres, uint32_t n, rt_context *donor);

// Resident workgroups per CU of the trace kernel at this LDS size.
hipError_t trace_occupancy(int filter, bool counting, size_t lds_bytes, int *workgroups_per_cu);

} // namespace rtamd
