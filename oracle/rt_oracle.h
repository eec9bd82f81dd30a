/*
 * rt_oracle.h -- CPU restatement of the per-pixel ray-tracing path of
 * swr06/Ray-Tracer (Source/main.cpp:64-70, 84-165, 273-293, 323-645).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, the smoke check
 * and the cpu_baseline leg of bench.py may link or call it; the shipped
 * library never does.
 *
 * Parity status: PINNED.  The restatement is checked pixel-exactly against the
 * reference's own code (compiled from /root/reference by oracle/Makefile into
 * oracle/_ref/) and against the golden vectors committed under tests/golden/.
 *
 * Two random-number sources drive the same arithmetic:
 *   ORC_RNG_MT19937  one sequential std::mt19937-compatible stream, as the
 *                    reference uses (Source/main.cpp:87-92);
 *   ORC_RNG_COUNTER  Philox4x32-10 blocks keyed on (seed, pixel, sample,
 *                    call number), the generator the HIP kernels use.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirror of class Sphere, Source/main.cpp:362-391 (36 bytes, same field order). */
typedef struct {
    float center[3];
    float color[3];
    float radius;
    int32_t material; /* enum class Material, Source/main.cpp:354-360 */
    float fuzz;
} orc_sphere;

enum { ORC_MAT_GLASS = 0, ORC_MAT_DIFFUSE = 1, ORC_MAT_METAL = 2, ORC_MAT_FUZZYMETAL = 3 };

/* The four members Camera::GetRay reads, Source/main.cpp:440-456. */
typedef struct {
    float origin[3];
    float bottom_left[3];
    float horizontal[3];
    float vertical[3];
} orc_camera;

enum { ORC_RNG_MT19937 = 0, ORC_RNG_COUNTER = 1 };

/* std::mt19937-compatible engine state (default seed 5489). */
typedef struct {
    uint32_t mt[624];
    int idx;
} orc_mt;

void orc_mt_seed(orc_mt *g, uint32_t seed);
uint32_t orc_mt_next(orc_mt *g);

/* RandomFloat(), Source/main.cpp:87-92, as libstdc++ evaluates
 * uniform_real_distribution<float>(0,1) on a 32-bit engine word. */
float orc_u32_to_unit_float(uint32_t w);
void orc_random_floats(uint32_t seed, int n, float *out);

/* Philox4x32-10 (Salmon et al., SC'11), one block. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Camera::Camera, Source/main.cpp:420-438. */
void orc_camera_init(const float lookfrom[3], const float lookat[3], const float up[3],
                     float fov_degrees, orc_camera *cam);

/* Camera::GetRay, Source/main.cpp:440-444. */
void orc_get_ray(const orc_camera *cam, float u, float v, float origin[3], float dir[3]);

/* IntersectSceneSpheres, Source/main.cpp:528-549.  Returns the index of the
 * sphere the reference would copy out, or -1.  Any of t/point/normal may be NULL. */
int32_t orc_intersect(const orc_sphere *spheres, uint32_t n, const float origin[3],
                      const float dir[3], float tmin, float tmax, float *t, float point[3],
                      float normal[3]);

/* GetGradientColorAtRay, Source/main.cpp:461-467. */
void orc_sky(const float dir[3], uint8_t rgb[3]);

/* ToRGB(vec3), Source/main.cpp:117-133, one channel. */
uint8_t orc_to_byte(float v);

/*
 * TraceThreadFunction, Source/main.cpp:613-645, with the compile-time constants
 * (g_Width, g_Height, SPP, RAY_DEPTH) turned into arguments.
 *
 * rgb      full-frame RGB8 buffer, index (x + y*width)*3, row 0 = bottom; only
 *          the tile's pixels are written.  May be NULL.
 * sums     full-frame uint32 x3 per-pixel sums of the per-sample bytes over
 *          samples [s0, s1); may be NULL.  When given, rgb (if given) receives
 *          sums / spp_div.
 * mt       engine for ORC_RNG_MT19937 (state carries across calls like the
 *          reference's function-static generator); ignored in counter mode.
 * rays     incremented by the number of IntersectSceneSpheres calls.
 */
void orc_trace_tile(const orc_sphere *spheres, uint32_t n, const orc_camera *cam,
                    uint32_t width, uint32_t height, uint32_t s0, uint32_t s1, uint32_t spp_div,
                    int32_t max_depth, int rng_mode, uint64_t seed, orc_mt *mt, int xstart,
                    int ystart, int xsize, int ysize, uint8_t *rgb, uint32_t *sums,
                    uint64_t *rays);

/* Extensions the reference does not have (default off everywhere): a real dielectric for
 * Material::Glass, gamma 2 on output, and colour kept in float between the recursion levels.
 * Parity for them is CPU oracle <-> GPU only. */
enum { ORC_FLAG_DIELECTRIC = 1u, ORC_FLAG_GAMMA2 = 2u, ORC_FLAG_FLOAT_SHADING = 4u };
void orc_trace_tile_ext(const orc_sphere *spheres, uint32_t n, const orc_camera *cam,
                        uint32_t width, uint32_t height, uint32_t s0, uint32_t s1, uint32_t spp_div,
                        int32_t max_depth, int rng_mode, uint64_t seed, orc_mt *mt, int xstart,
                        int ystart, int xsize, int ysize, uint32_t flags, uint8_t *rgb, uint32_t *sums,
                        uint64_t *rays);
uint8_t orc_gamma2(int32_t v);

/* One sample's colour, Source/main.cpp:627-631; counter mode only. */
void orc_sample_color(const orc_sphere *spheres, uint32_t n, const orc_camera *cam,
                      uint32_t width, uint32_t height, int32_t max_depth, uint64_t seed, int x,
                      int y, uint32_t sample, uint8_t rgb[3], uint64_t *rays);

/* Deterministic primary-ray closest-hit map: jitter == 0.5 (double),
 * tmin = 0.001f, tmax = +inf; out[x + y*width] = sphere index or -1. */
void orc_primary_hits(const orc_sphere *spheres, uint32_t n, const orc_camera *cam,
                      uint32_t width, uint32_t height, int32_t *out);

/* TraceScene, Source/main.cpp:647-659, joined: nthreads column strips, each
 * with its own mt19937 seeded 5489 + strip index.  Timing baseline only. */
void orc_trace_scene_threads(const orc_sphere *spheres, uint32_t n, const orc_camera *cam,
                             uint32_t width, uint32_t height, uint32_t spp, int32_t max_depth,
                             int xstart, int xsize, int nthreads, uint8_t *rgb, uint64_t *rays);

uint64_t orc_fnv1a64(const void *data, uint64_t nbytes);

#ifdef __cplusplus
}
#endif
#endif
