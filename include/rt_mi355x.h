/*
 * rt_mi355x.h -- C ABI of the MI355X (gfx950) replacement for the per-pixel
 * ray-tracing path of swr06/Ray-Tracer.
 *
 * The reference has no plugin or FFI interface; its hot path is entered through
 *     void TraceThreadFunction(int xstart, int ystart, int xsize, int ysize)
 * (Source/main.cpp:613), called by TraceScene() (Source/main.cpp:647-659), with
 * the scene, camera, sizes and framebuffer passed as globals.  This header turns
 * that seam into explicit arguments.  Every entry point cites what it replaces.
 *
 * Conventions
 *   - plain C, no C++ or torch types; all structs are PODs with fixed layout;
 *   - every function returns 0 on success or a negative rt_status; nothing
 *     throws across the boundary (the reference's path has no error convention:
 *     its functions are void and PutPixel silently drops out-of-range pixels,
 *     Source/main.cpp:275);
 *   - one rt_context drives ONE device; use one context (one process) per GPU;
 *   - pointers named d_* are device memory of the context's device, all others
 *     are host memory; `stream` is a hipStream_t passed as void* (NULL = the
 *     default stream).  Entry points ending in _device enqueue work and return
 *     without synchronising; the others return when the output is in host memory;
 *   - images are tightly packed RGB8, index (x + y*width)*3, row 0 = BOTTOM of
 *     the picture, exactly g_PixelData's layout (Source/main.cpp:76, 273-281);
 *   - the reference calls its seam from four threads at once on disjoint tiles
 *     (Source/main.cpp:654): entry points taking a context lock it, so concurrent
 *     calls on one context are safe and run one after another (the _device entry
 *     points still only enqueue; rt_destroy must not race with anything).  Trace
 *     launches of one context run one after another on the device as well, even
 *     when they are enqueued on different streams (they share the work queue);
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 2

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_HIP = -2,          /* a HIP runtime call failed; see rt_last_error() */
    RT_ERR_NO_DEVICE = -3,    /* no usable gfx950 device / library built without it */
    RT_ERR_NO_SCENE = -4,     /* rt_set_scene has not been called */
    RT_ERR_OUT_OF_MEMORY = -5
} rt_status;

/* enum class Material, Source/main.cpp:354-360 (same values). */
enum { RT_MAT_GLASS = 0, RT_MAT_DIFFUSE = 1, RT_MAT_METAL = 2, RT_MAT_FUZZYMETAL = 3 };

/* class Sphere, Source/main.cpp:362-391: same field order, 36 bytes.  Diffuse
 * colours are on a 0..255 scale and metal colours on 0..1, as in the reference's
 * own scene (Source/main.cpp:520-526). */
typedef struct rt_sphere {
    float center[3];
    float color[3];
    float radius;
    int32_t material;
    float fuzz;
} rt_sphere;

/* The members Camera::GetRay reads (Source/main.cpp:442, 447-456). */
typedef struct rt_camera {
    float origin[3];
    float bottom_left[3];
    float horizontal[3];
    float vertical[3];
} rt_camera;

/* g_Width, g_Height (Source/main.cpp:74-75), SPP, RAY_DEPTH (:610-611) and the
 * partition of one frame over several devices. */
typedef struct rt_params {
    uint32_t width;        /* frame size in pixels */
    uint32_t height;
    uint32_t spp;          /* samples per pixel of the WHOLE frame (the divisor) */
    int32_t  max_depth;    /* RAY_DEPTH */
    uint64_t seed;         /* key of the per-(pixel, sample) counter generator */
    uint32_t sample_begin; /* this call traces samples [sample_begin, */
    uint32_t sample_count; /*   sample_begin + sample_count); 0 = all spp */
    uint32_t tile_first;   /* of the 8x8-pixel tiles of the region, numbered row */
    uint32_t tile_stride;  /*   by row, trace tiles first, first+stride, ...; 0 or 1 = all */
    uint32_t flags;        /* RT_FLAG_* extensions; 0 = the reference's behaviour */
    uint32_t reserved;     /* must be 0 */
} rt_params;

/* Opt-in extensions.  The reference implements neither: its Glass spheres are flat white
 * (Source/main.cpp:594) and it writes sum / SPP without a transfer curve (:638-642).  With
 * flags == 0 results are the reference's; with a flag set they are checked against the same
 * extension in the CPU oracle only. */
enum {
    /* Material::Glass scatters as a dielectric of index 1.5: Schlick reflectance against one
     * RandomFloat() draw, glm::reflect / glm::refract (glm/detail/func_geometric.inl:110-127)
     * on the normalised direction, colour = Color * child colour like Metal. */
    RT_FLAG_DIELECTRIC = 1u,
    /* rt_resolve / rt_render apply gamma 2: byte = trunc(sqrt((sum / spp) / 255) * 255). */
    RT_FLAG_GAMMA2 = 2u,
    /* The reference truncates the colour to bytes at every recursion level (ToRGB inside
     * GetRayColor, Source/main.cpp:569-578, 587-591).  With this flag it stays in float between
     * the levels (Diffuse halves by * 0.5f), still limited to [0, 255] per level, and each sample
     * is truncated to bytes once. */
    RT_FLAG_FLOAT_SHADING = 4u
};

typedef struct rt_stats {
    uint64_t rays;            /* IntersectSceneSpheres calls (Source/main.cpp:562) since the last reset */
    uint64_t sphere_tests;    /* rays x sphere count: the tests the reference's loop performs for these rays
                                 (Source/main.cpp:534), not the number the chosen scan executed */
    uint64_t wave_passes;     /* passes of a 64-lane wave over the sphere list; rays / (64 * wave_passes)
                                 is the fraction of lanes that carried a ray */
    double   last_kernel_ms;  /* device time of the last synchronous call's kernels */
    double   last_wall_ms;    /* host wall time of the last synchronous call */
    /* Work the chosen scan EXECUTED for these rays, tallied only while RT_OPT_COUNT_WORK is 1 (a
     * slower build of the trace kernel, same results; the rt_params.flags extensions are not
     * tallied), 0 otherwise: */
    uint64_t node_visits;     /* inner nodes of the hierarchy visited, two box tests each (RT_OPT_FILTER 3) */
    uint64_t exact_tests;     /* evaluations of the reference's discriminant (Source/main.cpp:475-479);
                                 equals sphere_tests for RT_OPT_FILTER 0 */
    uint64_t root_evaluations; /* of those, the ones that went on to sqrt and divide (Source/main.cpp:490-499) */
} rt_stats;

typedef struct rt_context rt_context;

/* Library / device ------------------------------------------------------- */

uint32_t rt_abi_version(void);
const char *rt_status_string(int status);
/* Message of the last failure on this context ("" if none). */
const char *rt_last_error(const rt_context *ctx);

/* Creates a context on HIP device `device_id`.  No counterpart in the reference
 * (it runs on the calling thread). */
int rt_create(int device_id, rt_context **ctx);
int rt_destroy(rt_context *ctx);

/* Scene and camera ------------------------------------------------------- */

/* Replaces the global `std::vector<Sphere> Spheres` (Source/main.cpp:520-526):
 * copies `n` spheres to the device (list order is significant: on equal hit
 * distance the later sphere wins, Source/main.cpp:492-499, 539-545). */
int rt_set_scene(rt_context *ctx, const rt_sphere *spheres, uint32_t n);

/* Camera::Camera(lookfrom, lookat, up, fov), Source/main.cpp:420-438 (aspect is
 * the reference's hard-coded 16:9).  Host-only arithmetic. */
int rt_camera_init(const float lookfrom[3], const float lookat[3], const float up[3],
                   float fov_degrees, rt_camera *cam);

/* The path ---------------------------------------------------------------- */

/* TraceThreadFunction(xstart, ystart, xsize, ysize), Source/main.cpp:613-645:
 * traces the tile and writes its pixels into the caller's full-frame RGB8 host
 * buffer `rgb` (width*height*3 bytes; pixels outside the tile are untouched). */
int rt_render(rt_context *ctx, const rt_camera *cam, const rt_params *params, int xstart,
              int ystart, int xsize, int ysize, uint8_t *rgb);

/* The reference shows the frame while it fills in (DoRenderLoop uploads g_PixelData every 15th
 * display frame, Source/main.cpp:302-305).  This variant traces `samples_per_pass` samples at a
 * time; after each pass the tile in `rgb` holds the average of the samples so far and `progress`
 * (may be NULL) is called; a non-zero return stops the render (RT_OK, partial average kept).
 * After the last pass `rgb` equals rt_render's output.  The next passes are traced while a pass's
 * frame travels to the host and `progress` runs -- one pass ahead, or one launch of several small
 * passes ahead (RT_OPT_PROGRESSIVE_PASSES_PER_LAUNCH) -- so a stop takes effect that much late: what
 * was enqueued is traced (and counted in rt_stats) but never shown.  `rgb` in pinned host memory
 * (hipHostMalloc) is fetched three times as fast as pageable memory (0.12 against 0.35 ms for a
 * 1920x1080 frame), which is what bounds passes of one sample per pixel. */
typedef int (*rt_progress_fn)(void *user, uint32_t samples_done, uint32_t samples_total);
int rt_render_progressive(rt_context *ctx, const rt_camera *cam, const rt_params *params, int xstart,
                          int ystart, int xsize, int ysize, uint8_t *rgb, uint32_t samples_per_pass,
                          rt_progress_fn progress, void *user);

/* Same as rt_render, output left in device memory (`d_rgb`: full frame, width*height*3). */
int rt_render_device(rt_context *ctx, const rt_camera *cam, const rt_params *params, int xstart,
                     int ystart, int xsize, int ysize, uint8_t *d_rgb, void *stream);

/* rt_render_device for callers that cannot allocate.  It has no counterpart in the reference; it
 * is for code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.
 * Same arguments, same result, but the per-pixel colour sums and the work queue's heads live in a
 * file-scope device array of the library instead of buffers the context grows on demand, and the
 * call allocates and frees nothing (no hipMalloc, no hipFree).  Limits and rules:
 *   - width * height <= RT_STATIC_MAX_PIXELS, else RT_ERR_INVALID_ARGUMENT;
 *   - each GPU the library runs on holds its own copy of the array (24 MiB + 16 KiB) until the
 *     process ends, and ALL contexts on one GPU share it.  The library puts these calls in order per
 *     GPU, whatever context and stream they come through -- except on a stream under graph capture:
 *     there the call is captured as it stands (kernels only; it neither waits for nor records the
 *     library's ordering events), and a replay of the graph races with any other use of the array on
 *     that GPU that overlaps it in time;
 *   - the scan's device data must exist already: rt_set_scene and rt_set_option put the hierarchy
 *     on the device when the options select it, so this holds unless an earlier error prevented it. */
#define RT_STATIC_MAX_PIXELS (1u << 21) /* 2 097 152: 1920 x 1080 fits */
int rt_render_device_static(rt_context *ctx, const rt_camera *cam, const rt_params *params, int xstart,
                            int ystart, int xsize, int ysize, uint8_t *d_rgb, void *stream);

/* The sample loop of Source/main.cpp:623-636 for samples [sample_begin,
 * sample_begin+sample_count) and the tiles selected by tile_first/tile_stride:
 * ADDS each sample's byte colour into `d_sums` (full frame, width*height*3
 * uint32, index (x + y*width)*3 + channel).  The caller zeroes d_sums.  Calls of ONE context on
 * the same d_sums are ordered by the library; calls of DIFFERENT contexts (or devices) into the same
 * d_sums must not overlap in time: a chunk's sums are added with plain loads and stores wherever the
 * launch owns the chunk's pixels. */
int rt_accumulate_device(rt_context *ctx, const rt_camera *cam, const rt_params *params,
                         int xstart, int ystart, int xsize, int ysize, uint32_t *d_sums,
                         void *stream);

/* Several rt_accumulate_device calls in ONE launch.  TraceScene starts its four strips one after
 * the other and never waits for any of them (Source/main.cpp:647-659), so the threads of one strip
 * finish while the others still run; a launch of its own per strip (or per frame, per pass, per
 * device share) would instead end with the device draining that call's last paths -- up to
 * max_depth bounces with ever fewer lanes at work, about 0.2 ms -- before the next begins.  Here
 * the jobs' work follows each other through one work queue: waves that find job k handed out go
 * on with job k + 1 while job k's last paths still bounce.  Every job is what one
 * rt_accumulate_device call takes (own camera, frame, sample range, tile set, region, d_sums) and
 * its sums are exactly what that call would add.  The jobs must share params.seed and
 * params.flags; two jobs may add into the same d_sums (same frame) or into buffers that do not
 * overlap.  n_jobs <= RT_MAX_JOBS. */
#define RT_MAX_JOBS 8
typedef struct rt_job {
    rt_camera cam;
    rt_params params;
    int32_t xstart, ystart, xsize, ysize;
    uint32_t *d_sums;
} rt_job;
int rt_accumulate_batch_device(rt_context *ctx, const rt_job *jobs, uint32_t n_jobs, void *stream);

/* FinalColor / SPP, ToRGB, PutPixel (Source/main.cpp:638-642) over the tile:
 * d_rgb = clamp(d_sums / spp_div, 0, 255); `flags` may carry RT_FLAG_GAMMA2. */
int rt_resolve_device(rt_context *ctx, const uint32_t *d_sums, uint32_t width, uint32_t height,
                      uint32_t spp_div, uint32_t flags, int xstart, int ystart, int xsize, int ysize,
                      uint8_t *d_rgb, void *stream);

/* Tile partition of one frame over several devices (the reference's partition is TraceScene's
 * four column strips, Source/main.cpp:647-659; here a device takes the 8x8 tiles tile_first,
 * tile_first + tile_stride, ... of the whole frame, numbered row by row, which balances sky
 * against ground).  rt_pack_tiles_device is rt_resolve_device for exactly those tiles, written one
 * after the other into `d_tiles` (rt_tile_block_bytes() bytes: 192 per tile, pixel (lx, ly) of a
 * tile at (ly*8 + lx)*3; pixels of an edge tile outside the frame are 0) -- the block a device
 * contributes to a gather.  rt_unpack_tiles_device is PutPixel (Source/main.cpp:273-281) for the
 * gathered blocks of all `n_blocks` devices (block b = what the device with tile_first b and
 * tile_stride n_blocks packed; consecutive blocks block_bytes apart) into the RGB8 frame. */
uint64_t rt_tile_block_bytes(uint32_t width, uint32_t height, uint32_t tile_first, uint32_t tile_stride);
int rt_pack_tiles_device(rt_context *ctx, const uint32_t *d_sums, uint32_t width, uint32_t height,
                         uint32_t spp_div, uint32_t flags, uint32_t tile_first, uint32_t tile_stride,
                         uint8_t *d_tiles, void *stream);
int rt_unpack_tiles_device(rt_context *ctx, const uint8_t *d_blocks, uint32_t n_blocks, uint64_t block_bytes,
                           uint32_t width, uint32_t height, uint8_t *d_rgb, void *stream);

/* Deterministic primary-ray map: for every pixel, the index of the sphere that
 * IntersectSceneSpheres (Source/main.cpp:528-549) selects for the ray through
 * the pixel centre (jitter 0.5, tmin 0.001f, tmax +inf), or -1.  hits[x + y*width]. */
int rt_primary_hits(rt_context *ctx, const rt_camera *cam, uint32_t width, uint32_t height,
                    int32_t *hits);
int rt_primary_hits_device(rt_context *ctx, const rt_camera *cam, uint32_t width, uint32_t height,
                           int32_t *d_hits, void *stream);

/* IntersectSceneSpheres for caller-supplied rays (Source/main.cpp:528-549):
 * rays = n x {origin[3], direction[3]} floats; out = n x {int32 index, float T,
 * float point[3], float normal[3]} (32 bytes, zero where there is no hit). */
int rt_intersect(rt_context *ctx, const float *rays, uint32_t n, void *out);

/* Several devices of one node, one host process ------------------------------ */

/* TraceScene (Source/main.cpp:647-659) fans one frame out over THREAD_SPAWN_COUNT host threads, a
 * column strip each.  An rt_multi fans it out over GPUs instead: it owns one rt_context per entry of
 * `device_ids`, splits every frame over them, and assembles the picture on the FIRST device with one
 * exchange step per frame:
 *   RT_PARTITION_SAMPLES  device i traces samples [i*spp/n, (i+1)*spp/n) of every pixel into its own
 *                         integer sums, which are sum-reduced to the first device (12 bytes per
 *                         pixel), which divides and packs;
 *   RT_PARTITION_TILES    device i traces all samples of the 8x8 tiles i, i + n, ..., divides and
 *                         packs them, and the tile blocks are gathered (3/n bytes per pixel and device).
 * The picture is bit-identical to rt_render's for any n and either partition.  `exchange` selects
 * the transport: RT_EXCHANGE_RCCL (ncclReduce / grouped ncclSend+ncclRecv; librccl is loaded with
 * dlopen by this call), RT_EXCHANGE_PEER_COPY (the first device adds the others' sums in place over
 * xGMI peer access, tile blocks are copied device to device), RT_EXCHANGE_AUTO (peer copies where the
 * first device can address every other one, else RCCL if it loads and all ids are distinct; creation
 * fails with RT_ERR_HIP when neither transport exists between the first device and another one).  With
 * one device RT_EXCHANGE_RCCL runs the same calls as with several (the rank reduces in place and sends
 * its block to itself).  The same id may appear several times (ranks sharing a device exchange by peer
 * copies; meant for testing on one GPU).  At most 16 entries.
 * Frames are double-buffered: rt_multi_render_device returns when the frame is enqueued, and the
 * exchange of one frame (or of one call's frames) runs beside the tracing of the next. */
typedef struct rt_multi rt_multi;
enum { RT_PARTITION_SAMPLES = 0, RT_PARTITION_TILES = 1 };
enum { RT_EXCHANGE_AUTO = 0, RT_EXCHANGE_RCCL = 1, RT_EXCHANGE_PEER_COPY = 2 };

int rt_multi_create(const int *device_ids, uint32_t n_devices, int exchange, rt_multi **multi);
int rt_multi_destroy(rt_multi *multi);
const char *rt_multi_last_error(const rt_multi *multi);
uint32_t rt_multi_device_count(const rt_multi *multi);
int rt_multi_uses_rccl(const rt_multi *multi);                   /* 1 if the exchange runs over RCCL */
rt_context *rt_multi_context(rt_multi *multi, uint32_t index);   /* the index-th device's context (owned by multi) */
int rt_multi_set_scene(rt_multi *multi, const rt_sphere *spheres, uint32_t n);   /* on every device */
int rt_multi_set_option(rt_multi *multi, int option, int64_t value);             /* on every device */
/* The whole frame (params->sample_* and tile_* must be 0) into host memory `rgb`; returns when it is there. */
int rt_multi_render(rt_multi *multi, const rt_camera *cam, const rt_params *params, int partition, uint8_t *rgb);
/* The same into `d_rgb`, memory of the FIRST device; only enqueues (frames complete in the order of the
 * calls; rt_multi_synchronize waits for all of them). */
int rt_multi_render_device(rt_multi *multi, const rt_camera *cam, const rt_params *params, int partition,
                           uint8_t *d_rgb);
/* `n_frames` (<= RT_MAX_JOBS) frames in one call: cams[f], params[f] -> d_rgb[f] (first device's memory),
 * the same pictures as n_frames calls of rt_multi_render_device.  Every device traces its shares of all
 * the frames in ONE launch (rt_accumulate_batch_device: a device's share of a frame is small, and the
 * end of a launch weighs on it), then the frames are exchanged and finished one by one.  The frames
 * must share seed and flags, as the jobs of one launch do; their sizes may differ.  Only enqueues.
 * Stands where a host would call TraceScene (Source/main.cpp:647-659) once per frame of a sequence. */
int rt_multi_render_frames_device(rt_multi *multi, const rt_camera *cams, const rt_params *params, uint32_t n_frames,
                                  int partition, uint8_t *const *d_rgb);
int rt_multi_synchronize(rt_multi *multi);
int rt_multi_get_stats(rt_multi *multi, rt_stats *stats);        /* counters summed over the devices */
int rt_multi_reset_stats(rt_multi *multi);

/* Tuning (no counterpart in the reference) ---------------------------------- */

typedef enum rt_option {
    /* How the sphere scan finds the spheres that need the reference's exact root
     * evaluation: 2 a conservatively widened discriminant test as two 4-term
     * products on the FP32 matrix cores; 1 the same kind of test with fused
     * multiply-adds on the vector ALU; 0 the reference's own unfused discriminant
     * for every sphere; 3 a per-ray walk of a bounding-box hierarchy built over the list
     * (the reference has none: it tests every sphere, Source/main.cpp:534), whose boxes are
     * padded so that no sphere the reference accepts is skipped; -1 (default) automatic:
     * 3 from RT_OPT_BVH_MIN_SPHERES spheres up (13), else 2, or 1 below 96 spheres.  All give
     * bit-identical results; forcing one exists to prove it. */
    RT_OPT_FILTER = 0,
    RT_OPT_TILE_SPHERES = 1,      /* spheres per LDS tile, 128..4096 (default 512) */
    RT_OPT_WORKGROUPS_PER_CU = 2, /* cap on resident workgroups per CU, 0 = occupancy limit */
    RT_OPT_BVH_LEAF_SPHERES = 3,  /* most spheres per leaf of the hierarchy, 1..15, or 0 (default) = chosen when the tree is
                                     built: the smallest of 4, 8, 12, 15 whose tree, copied to LDS, leaves room for seven
                                     workgroups per CU, else the largest that LDS may hold, or 8 for trees walked in memory */
    RT_OPT_BVH_MIN_SPHERES = 4,   /* scene size from which the automatic choice is the hierarchy (default 13) */
    RT_OPT_BVH_LDS_NODES = 5,     /* hierarchies of up to this many nodes (and at most 2048 sphere records) are walked
                                     in LDS, 0..2048 (default 512) */
    RT_OPT_BVH_SUSPEND_LANES = 6, /* a 64-lane wave stops walking the hierarchy when at most this many of its rays
                                     are still at it; they resume in the wave's next pass (0..63, 0 = never) */
    RT_OPT_CHUNKS_PER_WAVE = 7,   /* work queue granularity: a call whose 8x8-pixel tiles number fewer than this many
                                     per resident wave cuts each tile into 2, 4 or 8 strips of rows and then its samples
                                     into segments, chunks of at least 256 paths; segments add into the frame with
                                     atomics (0..256, default 32, 0 = never cut, except that a chunk never holds more than 256 samples
                                     of a pixel).  Results do not depend on it. */
    RT_OPT_SUBTILES = 8,          /* most strips of rows a tile is cut into before its samples are: 1, 2, 4 or 8 (default:
                                     single rows; strips own their pixels and add into the frame with plain stores) */
    RT_OPT_DIRECT_ADD_SAMPLES = 10, /* calls (or cut chunks) of at most this many samples per pixel add every path's colour to the
                                     frame with atomics of its own instead of through per-chunk sums in LDS, so that a wave
                                     never waits for a chunk to drain (0..64, default 1, 0 = never).  Results do not depend on it. */
    RT_OPT_COUNT_WORK = 9,        /* 1: trace with the counting build, which fills rt_stats' node_visits, exact_tests and
                                     root_evaluations (slower; for measurement); 0 (default): those stay 0 */
    RT_OPT_MAX_CHUNKS_PER_LAUNCH = 11, /* a launch's work queue counts its chunks in 31 bits: a call whose 8x8 tiles times
                                     256-sample runs outnumber this (default and most 2^30 - 1) has its sample range traced
                                     by several launches, one after the other.  Results do not depend on it; lowering it
                                     exists to test that */
    RT_OPT_PROGRESSIVE_PASSES_PER_LAUNCH = 12 /* rt_render_progressive traces up to this many passes per launch, 1..8, after the
                                     first pass, which goes alone; 0 (default) = as many as make 16 samples per pixel.
                                     Frames are delivered one per pass and in order either way, those of a launch right
                                     after each other; 1 delivers every pass as soon as it is traced */
} rt_option;

int rt_set_option(rt_context *ctx, int option, int64_t value);
/* The option's current value as rt_set_option takes it (a fresh context reports the defaults). */
int rt_get_option(rt_context *ctx, int option, int64_t *value);

/* Builds the hierarchy of RT_OPT_FILTER 3 for `spheres` on the host and checks its structure
 * (every sphere in exactly one leaf or on the always-tested list, every box contains what is
 * below it).  Needs no device.  info (may be NULL) = {nodes, levels, always-tested spheres,
 * records}.  No counterpart in the reference. */
int rt_accel_selfcheck(const rt_sphere *spheres, uint32_t n, int leaf_spheres, uint32_t info[4]);

/* Counters ---------------------------------------------------------------- */

/* Waits for the context's enqueued work, then reports. */
int rt_get_stats(rt_context *ctx, rt_stats *stats);
int rt_reset_stats(rt_context *ctx);
/* Measurement aid of the counting build (RT_OPT_COUNT_WORK 1): for the first `n` (<= 12) sections of the
 * trace kernel -- 0 wave pass, 1 refill, 2 walk start (always-tested spheres), 3 hierarchy node, 4 group of
 * up to four leaf records, 5 root evaluation, 6 first random-point attempt, 7 helper attempt, 8 scatter,
 * 9 replay of one recursion level, 10 walk set-up, 11 shading -- out[2i] = how many times a 64-lane wave
 * executed the section and out[2i+1] = with how many lanes enabled in total, since the last reset.
 * Pairs 12..16 hold shader-clock cycles instead, summed over the waves: out[24] wave lifetime, [25] waiting
 * for the work queue, [26] refill (queue included), [27] sphere scan, [28] random points, [29] shading and
 * replay, [30] flush of chunk sums, [32] the number of waves (n <= 28 pairs). */
int rt_get_work_profile(rt_context *ctx, uint64_t *out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
