/*
 * rt_oracle.c -- CPU restatement of swr06/Ray-Tracer's per-pixel path.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Parity status: pinned against
 * the reference's own code and against tests/golden/.
 *
 * Every function cites the lines of /root/reference/Source/main.cpp it follows.
 * Arithmetic is written operation by operation in the reference's order:
 * single-precision multiply and add are never fused (build with
 * -ffp-contract=off), products of a dot() are rounded individually and summed
 * left to right (glm/detail/func_geometric.inl:54-61), and double precision is
 * used exactly where the reference's `double`-returning RandomFloat() promotes
 * an expression.
 */
#include "rt_oracle.h"

#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* std::mt19937 (32-bit Mersenne Twister, Matsumoto & Nishimura 1998)  */
/* ------------------------------------------------------------------ */

void orc_mt_seed(orc_mt *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

uint32_t orc_mt_next(orc_mt *g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = g->mt[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u)
                v ^= 0x9908b0dfu;
            g->mt[i] = v;
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/*
 * RandomFloat(), Source/main.cpp:87-92: uniform_real_distribution<float>(0,1)
 * over a 32-bit engine.  libstdc++'s generate_canonical<float,24> takes ONE
 * engine word, converts it to float (round to nearest even), divides by 2^32
 * and replaces a result of 1.0f by the float just below it.
 */
float orc_u32_to_unit_float(uint32_t w)
{
    float sum = (float)w;
    float r = sum / 4294967296.0f;
    if (r >= 1.0f)
        r = 0.99999994f; /* nextafterf(1, 0) */
    return r;
}

void orc_random_floats(uint32_t seed, int n, float *out)
{
    orc_mt g;
    orc_mt_seed(&g, seed);
    for (int i = 0; i < n; i++)
        out[i] = orc_u32_to_unit_float(orc_mt_next(&g));
}

/* ------------------------------------------------------------------ */
/* Philox4x32-10                                                       */
/* ------------------------------------------------------------------ */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------ */
/* Random source shared by the sample loop and the scatter functions   */
/* ------------------------------------------------------------------ */

typedef struct {
    int mode;
    orc_mt *mt;
    uint32_t key[2];
    uint32_t pixel, sample, call;
    uint32_t flags; /* ORC_FLAG_*: extensions the reference does not have */
} rng_t;

/*
 * k consecutive engine words.  MT mode: the next k outputs of the shared
 * stream.  Counter mode: words 0..k-1 of block (pixel, sample, call++).
 */
static void rng_words(rng_t *r, int k, uint32_t *w)
{
    if (r->mode == ORC_RNG_MT19937) {
        for (int i = 0; i < k; i++)
            w[i] = orc_mt_next(r->mt);
    } else {
        uint32_t ctr[4] = { r->pixel, r->sample, r->call, 0u };
        uint32_t blk[4];
        orc_philox4x32_10(ctr, r->key, blk);
        r->call++;
        for (int i = 0; i < k; i++)
            w[i] = blk[i];
    }
}

/* RandomFloat(min,max), Source/main.cpp:94-97: float + float*double -> double -> float. */
static float random_range(float mn, float mx, uint32_t w)
{
    double r = (double)orc_u32_to_unit_float(w);
    return (float)((double)mn + (double)(mx - mn) * r);
}

/* ------------------------------------------------------------------ */
/* vec3 helpers in glm 0.9.8.5's operation order                       */
/* ------------------------------------------------------------------ */

static float dot3(const float a[3], const float b[3])
{
    /* glm/detail/func_geometric.inl:54-61: tmp = a*b; tmp.x + tmp.y + tmp.z */
    float px = a[0] * b[0], py = a[1] * b[1], pz = a[2] * b[2];
    return px + py + pz;
}

static void normalize3(const float v[3], float out[3])
{
    /* func_geometric.inl:88-95: v * inversesqrt(dot(v,v)); inversesqrt = 1/sqrt
     * (func_exponential.inl:130-133) */
    float inv = 1.0f / sqrtf(dot3(v, v));
    out[0] = v[0] * inv; out[1] = v[1] * inv; out[2] = v[2] * inv;
}

static void cross3(const float x[3], const float y[3], float out[3])
{
    /* func_geometric.inl:74-85 */
    out[0] = x[1] * y[2] - y[1] * x[2];
    out[1] = x[2] * y[0] - y[2] * x[0];
    out[2] = x[0] * y[1] - y[0] * x[1];
}

/*
 * static_cast<int>(float) as the reference's x86-64 build executes it
 * (cvttss2si): truncation, and INT_MIN for NaN or out-of-range input.
 */
static int float_to_int_x86(float v)
{
    return (v < 2147483648.0f && v >= -2147483648.0f) ? (int)v : INT_MIN;
}

static int clampi(int x, int lo, int hi)
{
    /* glm::clamp = min(max(x, lo), hi), func_common.inl:543-547 */
    int m = x > lo ? x : lo;
    return m < hi ? m : hi;
}

/* ToRGB(vec3), Source/main.cpp:117-133 (one channel). */
uint8_t orc_to_byte(float v)
{
    return (uint8_t)clampi(float_to_int_x86(v), 0, 255);
}

/* ------------------------------------------------------------------ */
/* Camera                                                              */
/* ------------------------------------------------------------------ */

/* Camera::Camera, Source/main.cpp:420-438. */
void orc_camera_init(const float lookfrom[3], const float lookat[3], const float up[3],
                     float fov_degrees, orc_camera *cam)
{
    const float aspect = 16.0f / 9.0f; /* :448 */
    float theta = fov_degrees * 0.01745329251994329576923690768489f; /* glm::radians */
    float H = tanf(theta / 2.0f);
    float vh = 2.0f * H;
    float vw = vh * aspect;

    float d[3] = { lookfrom[0] - lookat[0], lookfrom[1] - lookat[1], lookfrom[2] - lookat[2] };
    float w[3], c[3], u[3], v[3];
    normalize3(d, w);
    cross3(up, w, c);
    normalize3(c, u);
    cross3(w, u, v);

    for (int i = 0; i < 3; i++) {
        cam->origin[i] = lookfrom[i];
        cam->horizontal[i] = vw * u[i];
        cam->vertical[i] = vh * v[i];
    }
    for (int i = 0; i < 3; i++) {
        /* m_Origin - (m_Horizontal / 2.0f) - (m_Vertical / 2.0f) - w */
        float a = cam->origin[i] - (cam->horizontal[i] / 2.0f);
        float b = a - (cam->vertical[i] / 2.0f);
        cam->bottom_left[i] = b - w[i];
    }
}

/* Camera::GetRay, Source/main.cpp:440-444: direction is NOT normalised. */
void orc_get_ray(const orc_camera *cam, float u, float v, float origin[3], float dir[3])
{
    for (int i = 0; i < 3; i++) {
        float a = cam->bottom_left[i] + (cam->horizontal[i] * u);
        float b = a + (v * cam->vertical[i]);
        dir[i] = b - cam->origin[i];
        origin[i] = cam->origin[i];
    }
}

/* ------------------------------------------------------------------ */
/* Intersection                                                        */
/* ------------------------------------------------------------------ */

typedef struct {
    float point[3];
    float normal[3];
    float t;
    int inside; /* normal was flipped: the ray travels inside the sphere (:510-514) */
} hit_t;

/* RaySphereIntersectionTest, Source/main.cpp:469-518. */
static int sphere_test(const orc_sphere *s, const float O[3], const float D[3], float tmin,
                       float tmax, hit_t *rec)
{
    float oc[3] = { O[0] - s->center[0], O[1] - s->center[1], O[2] - s->center[2] };
    float A = dot3(D, D);
    float B = (float)(2.0 * (double)dot3(oc, D));
    float C = dot3(oc, oc) - s->radius * s->radius;
    float fourA = 4.0f * A; /* `4 * A * C` groups as (4*A)*C */
    float disc = B * B - fourA * C;

    if (disc < 0)
        return 0;

    float root = (-B - sqrtf(disc)) / (2.0f * A);
    if (root < tmin || root > tmax) {
        root = (-B + sqrtf(disc)) / (2.0f * A);
        if (root < tmin || root > tmax)
            return 0;
    }

    rec->t = root;
    /* Ray::GetAt, :343-346: origin + direction * vec3(scale) */
    for (int i = 0; i < 3; i++)
        rec->point[i] = O[i] + (D[i] * root);
    for (int i = 0; i < 3; i++)
        rec->normal[i] = (rec->point[i] - s->center[i]) / s->radius;
    rec->inside = 0;
    if (dot3(D, rec->normal) > 0.0f) {
        rec->normal[0] = -rec->normal[0];
        rec->normal[1] = -rec->normal[1];
        rec->normal[2] = -rec->normal[2];
        rec->inside = 1;
    }
    return 1;
}

/* IntersectSceneSpheres, Source/main.cpp:528-549: list order, shrinking tmax,
 * an equal T is accepted, so the later sphere wins a tie. */
static int32_t intersect_scene(const orc_sphere *spheres, uint32_t n, const float O[3],
                               const float D[3], float tmin, float tmax, hit_t *closest)
{
    hit_t tmp;
    int32_t hit = -1;
    float closest_t = tmax;
    for (uint32_t i = 0; i < n; i++) {
        if (sphere_test(&spheres[i], O, D, tmin, closest_t, &tmp)) {
            hit = (int32_t)i;
            closest_t = tmp.t;
            *closest = tmp;
        }
    }
    return hit;
}

int32_t orc_intersect(const orc_sphere *spheres, uint32_t n, const float origin[3],
                      const float dir[3], float tmin, float tmax, float *t, float point[3],
                      float normal[3])
{
    hit_t rec;
    memset(&rec, 0, sizeof rec);
    int32_t idx = intersect_scene(spheres, n, origin, dir, tmin, tmax, &rec);
    if (idx >= 0) {
        if (t) *t = rec.t;
        if (point) memcpy(point, rec.point, sizeof rec.point);
        if (normal) memcpy(normal, rec.normal, sizeof rec.normal);
    }
    return idx;
}

/* ------------------------------------------------------------------ */
/* Shading                                                             */
/* ------------------------------------------------------------------ */

/* GetGradientColorAtRay + Lerp, Source/main.cpp:461-467, 157-160. */
void orc_sky(const float dir[3], uint8_t rgb[3])
{
    static const float v1[3] = { 255.0f, 255.0f, 255.0f };
    static const float v2[3] = { 128.0f, 178.0f, 255.0f };
    float t = dir[1] * 1.8f;
    for (int i = 0; i < 3; i++) {
        float v = (1.0f - t) * v1[i] + t * v2[i];
        rgb[i] = (uint8_t)clampi(float_to_int_x86(v), 0, 255); /* ivec3(v), ToRGB(ivec3) */
    }
}

/* GeneratePointInUnitSphere, Source/main.cpp:398-414. */
static void point_in_unit_sphere(rng_t *rng, float p[3])
{
    for (;;) {
        uint32_t w[3];
        rng_words(rng, 3, w);
        p[0] = random_range(-1.0f, 1.0f, w[0]);
        p[1] = random_range(-1.0f, 1.0f, w[1]);
        p[2] = random_range(-1.0f, 1.0f, w[2]);
        /* PointIsInSphere, :393-396 */
        if (((p[0] * p[0]) + (p[1] * p[1]) + (p[2] * p[2])) < (1.0f * 1.0f))
            return;
    }
}

/* The scattering half of GetRayColor's material branches, Source/main.cpp:563-594: fills the
 * child ray's direction R and returns how the child's colour enters the parent's:
 * 0 no child (flat white), 1 Color * (child / 2) (Diffuse), 2 Color * child. */
static int scatter(const orc_sphere *s, const hit_t *rec, const float D[3], rng_t *rng, float R[3])
{
    if (s->material == ORC_MAT_DIFFUSE) {
        float r[3];
        point_in_unit_sphere(rng, r);
        for (int i = 0; i < 3; i++)
            R[i] = rec->normal[i] + r[i];
        return 1;
    }
    if (s->material == ORC_MAT_METAL) {
        /* glm::reflect: I - N * dot(N, I) * 2, func_geometric.inl:110-116 */
        float d = dot3(rec->normal, D);
        float r[3];
        for (int i = 0; i < 3; i++)
            R[i] = D[i] - (rec->normal[i] * d) * 2.0f;
        point_in_unit_sphere(rng, r);
        for (int i = 0; i < 3; i++)
            R[i] = R[i] + (s->fuzz * r[i]);
        return 2;
    }
    if (s->material == ORC_MAT_GLASS && (rng->flags & ORC_FLAG_DIELECTRIC)) {
        /* EXTENSION (not in the reference, whose Glass is flat white, :594): a dielectric of
         * index 1.5 in the style of the book the reference follows.  One RandomFloat() draw
         * decides between reflection and refraction; glm::reflect / glm::refract
         * (func_geometric.inl:110-127) act on the normalised direction; every float operation
         * is written out so that the GPU kernel can repeat it bit for bit. */
        float inv = 1.0f / sqrtf(dot3(D, D));
        float U[3] = { D[0] * inv, D[1] * inv, D[2] * inv };
        float ratio = rec->inside ? 1.5f : (1.0f / 1.5f);
        float nu = dot3(rec->normal, U); /* dot(N, I) <= 0: N opposes the ray */
        float cos_theta = -nu;
        if (cos_theta > 1.0f) cos_theta = 1.0f;
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        float r0 = (1.0f - ratio) / (1.0f + ratio);
        r0 = r0 * r0;
        float m = 1.0f - cos_theta;
        float m5 = ((m * m) * (m * m)) * m;
        float reflectance = r0 + (1.0f - r0) * m5;
        uint32_t w[1];
        rng_words(rng, 1, w);
        if (ratio * sin_theta > 1.0f || reflectance > orc_u32_to_unit_float(w[0])) {
            for (int i = 0; i < 3; i++)
                R[i] = U[i] - (rec->normal[i] * nu) * 2.0f;
        } else {
            float k = 1.0f - (ratio * ratio) * (1.0f - nu * nu);
            float f = ratio * nu + sqrtf(k);
            float keep = k >= 0.0f ? 1.0f : 0.0f;
            for (int i = 0; i < 3; i++)
                R[i] = (ratio * U[i] - f * rec->normal[i]) * keep;
        }
        return 2;
    }
    return 0; /* :594 */
}

/* GetRayColor, Source/main.cpp:551-598. */
static void ray_color(const orc_sphere *spheres, uint32_t n, const float O[3], const float D[3],
                      int depth, rng_t *rng, uint64_t *rays, uint8_t out[3])
{
    if (depth <= 0) {
        out[0] = out[1] = out[2] = 0;
        return;
    }

    hit_t rec;
    (*rays)++;
    int32_t idx = intersect_scene(spheres, n, O, D, 0.001f, INFINITY, &rec);
    if (idx < 0) {
        orc_sky(D, out);
        return;
    }

    const orc_sphere *s = &spheres[idx];
    float R[3];
    const int kind = scatter(s, &rec, D, rng, R);
    if (kind == 0) {
        out[0] = out[1] = out[2] = 255; /* :594 */
        return;
    }
    uint8_t c[3];
    ray_color(spheres, n, rec.point, R, depth - 1, rng, rays, c);
    for (int i = 0; i < 3; i++) {
        uint8_t h = kind == 1 ? (uint8_t)(c[i] / 2) : c[i]; /* :569-571 */
        out[i] = orc_to_byte(s->color[i] * (float)h);       /* :573-578, :587-591 */
    }
}

/* EXTENSION (ORC_FLAG_FLOAT_SHADING): the same recursion with the colour kept in float between
 * the levels -- no truncation to bytes, Diffuse halves by * 0.5f -- but still limited to
 * [0, 255] at every level like ToRGB does (the reference's diffuse colours are on a 0..255
 * scale and rely on that limit); NaN counts as 0.  The sample is truncated to bytes once. */
static float limit_255(float v)
{
    return v > 0.0f ? (v < 255.0f ? v : 255.0f) : 0.0f; /* false for NaN: 0 */
}

static void ray_color_float(const orc_sphere *spheres, uint32_t n, const float O[3], const float D[3],
                            int depth, rng_t *rng, uint64_t *rays, float out[3])
{
    if (depth <= 0) {
        out[0] = out[1] = out[2] = 0.0f;
        return;
    }
    hit_t rec;
    (*rays)++;
    int32_t idx = intersect_scene(spheres, n, O, D, 0.001f, INFINITY, &rec);
    if (idx < 0) { /* GetGradientColorAtRay, :461-467, before its conversion to integers */
        static const float v1[3] = { 255.0f, 255.0f, 255.0f };
        static const float v2[3] = { 128.0f, 178.0f, 255.0f };
        float t = D[1] * 1.8f;
        for (int i = 0; i < 3; i++)
            out[i] = limit_255((1.0f - t) * v1[i] + t * v2[i]);
        return;
    }
    const orc_sphere *s = &spheres[idx];
    float R[3];
    const int kind = scatter(s, &rec, D, rng, R);
    if (kind == 0) {
        out[0] = out[1] = out[2] = 255.0f;
        return;
    }
    float c[3];
    ray_color_float(spheres, n, rec.point, R, depth - 1, rng, rays, c);
    for (int i = 0; i < 3; i++) {
        float h = kind == 1 ? c[i] * 0.5f : c[i];
        out[i] = limit_255(s->color[i] * h);
    }
}

/* One sample: Source/main.cpp:627-631. */
static void sample_color(const orc_sphere *spheres, uint32_t n, const orc_camera *cam,
                         uint32_t width, uint32_t height, int32_t max_depth, int i, int j,
                         rng_t *rng, uint64_t *rays, uint8_t out[3])
{
    uint32_t w[2];
    rng_words(rng, 2, w);
    double ru = (double)orc_u32_to_unit_float(w[0]);
    double rv = (double)orc_u32_to_unit_float(w[1]);
    float u = (float)(((double)(float)i + ru) / (double)(float)width);
    float v = (float)(((double)(float)j + rv) / (double)(float)height);
    float O[3], D[3];
    orc_get_ray(cam, u, v, O, D);
    if (rng->flags & ORC_FLAG_FLOAT_SHADING) {
        float f[3];
        ray_color_float(spheres, n, O, D, max_depth, rng, rays, f);
        for (int k = 0; k < 3; k++)
            out[k] = orc_to_byte(f[k]);
    } else {
        ray_color(spheres, n, O, D, max_depth, rng, rays, out);
    }
}

void orc_sample_color(const orc_sphere *spheres, uint32_t n, const orc_camera *cam,
                      uint32_t width, uint32_t height, int32_t max_depth, uint64_t seed, int x,
                      int y, uint32_t sample, uint8_t rgb[3], uint64_t *rays)
{
    uint64_t local = 0;
    rng_t rng;
    rng.flags = 0;
    rng.mode = ORC_RNG_COUNTER;
    rng.mt = NULL;
    rng.key[0] = (uint32_t)seed;
    rng.key[1] = (uint32_t)(seed >> 32);
    rng.pixel = (uint32_t)x + (uint32_t)y * width;
    rng.sample = sample;
    rng.call = 0;
    sample_color(spheres, n, cam, width, height, max_depth, x, y, &rng, &local, rgb);
    if (rays) *rays += local;
}

void orc_trace_tile(const orc_sphere *spheres, uint32_t n, const orc_camera *cam,
                    uint32_t width, uint32_t height, uint32_t s0, uint32_t s1, uint32_t spp_div,
                    int32_t max_depth, int rng_mode, uint64_t seed, orc_mt *mt, int xstart,
                    int ystart, int xsize, int ysize, uint8_t *rgb, uint32_t *sums,
                    uint64_t *rays)
{
    orc_trace_tile_ext(spheres, n, cam, width, height, s0, s1, spp_div, max_depth, rng_mode, seed, mt,
                       xstart, ystart, xsize, ysize, 0u, rgb, sums, rays);
}

/* EXTENSION: gamma 2 on the averaged byte (the reference writes sum / SPP as is, :638-642). */
uint8_t orc_gamma2(int32_t v)
{
    int32_t c = clampi(v, 0, 255);
    return (uint8_t)clampi(float_to_int_x86(sqrtf((float)c / 255.0f) * 255.0f), 0, 255);
}

/* TraceThreadFunction, Source/main.cpp:613-645, plus the ORC_FLAG_* extensions. */
void orc_trace_tile_ext(const orc_sphere *spheres, uint32_t n, const orc_camera *cam,
                        uint32_t width, uint32_t height, uint32_t s0, uint32_t s1, uint32_t spp_div,
                        int32_t max_depth, int rng_mode, uint64_t seed, orc_mt *mt, int xstart,
                        int ystart, int xsize, int ysize, uint32_t flags, uint8_t *rgb, uint32_t *sums,
                        uint64_t *rays)
{
    uint64_t local = 0;
    rng_t rng;
    rng.flags = flags;
    rng.mode = rng_mode;
    rng.mt = mt;
    rng.key[0] = (uint32_t)seed;
    rng.key[1] = (uint32_t)(seed >> 32);

    for (int i = xstart; i < xstart + xsize; i++) {
        for (int j = ystart; j < ystart + ysize; j++) {
            int32_t fin[3] = { 0, 0, 0 }; /* glm::ivec3 zero-initialises, type_vec3.inl:34-41 */
            for (uint32_t s = s0; s < s1; s++) {
                uint8_t c[3];
                rng.pixel = (uint32_t)i + (uint32_t)j * width;
                rng.sample = s;
                rng.call = 0;
                sample_color(spheres, n, cam, width, height, max_depth, i, j, &rng, &local, c);
                fin[0] += c[0]; fin[1] += c[1]; fin[2] += c[2];
            }
            /* PutPixel, :273-281: silently drops out-of-range pixels */
            if ((uint32_t)i >= width || (uint32_t)j >= height)
                continue;
            size_t loc = ((size_t)i + (size_t)j * width) * 3;
            if (sums) {
                sums[loc + 0] = (uint32_t)fin[0];
                sums[loc + 1] = (uint32_t)fin[1];
                sums[loc + 2] = (uint32_t)fin[2];
            }
            if (rgb) {
                for (int c = 0; c < 3; c++) {
                    int32_t q = fin[c] / (int32_t)spp_div;
                    rgb[loc + c] = (flags & ORC_FLAG_GAMMA2) ? orc_gamma2(q) : (uint8_t)clampi(q, 0, 255);
                }
            }
        }
    }
    if (rays) *rays += local;
}

void orc_primary_hits(const orc_sphere *spheres, uint32_t n, const orc_camera *cam,
                      uint32_t width, uint32_t height, int32_t *out)
{
    for (uint32_t j = 0; j < height; j++) {
        for (uint32_t i = 0; i < width; i++) {
            float u = (float)(((double)(float)i + 0.5) / (double)(float)width);
            float v = (float)(((double)(float)j + 0.5) / (double)(float)height);
            float O[3], D[3];
            hit_t rec;
            orc_get_ray(cam, u, v, O, D);
            out[(size_t)i + (size_t)j * width] =
                intersect_scene(spheres, n, O, D, 0.001f, INFINITY, &rec);
        }
    }
}

/* ------------------------------------------------------------------ */
/* Threaded strips (timing baseline)                                   */
/* ------------------------------------------------------------------ */

typedef struct {
    const orc_sphere *spheres; uint32_t n; const orc_camera *cam;
    uint32_t width, height, spp; int32_t max_depth;
    int xstart, xsize; uint32_t seed; uint8_t *rgb; uint64_t rays;
} strip_t;

static void *strip_main(void *p)
{
    strip_t *s = (strip_t *)p;
    orc_mt g;
    orc_mt_seed(&g, s->seed);
    orc_trace_tile(s->spheres, s->n, s->cam, s->width, s->height, 0, s->spp, s->spp,
                   s->max_depth, ORC_RNG_MT19937, 0, &g, s->xstart, 0, s->xsize,
                   (int)s->height, s->rgb, NULL, &s->rays);
    return NULL;
}

/* TraceScene, Source/main.cpp:647-659, joined, one generator per strip. */
void orc_trace_scene_threads(const orc_sphere *spheres, uint32_t n, const orc_camera *cam,
                             uint32_t width, uint32_t height, uint32_t spp, int32_t max_depth,
                             int xstart, int xsize, int nthreads, uint8_t *rgb, uint64_t *rays)
{
    if (nthreads < 1) nthreads = 1;
    strip_t *st = (strip_t *)calloc((size_t)nthreads, sizeof *st);
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof *th);
    int sizex = xsize / nthreads; /* remainder columns are not rendered, as in :649 */
    for (int t = 0; t < nthreads; t++) {
        strip_t s = { spheres, n, cam, width, height, spp, max_depth,
                      xstart + sizex * t, sizex, 5489u + (uint32_t)t, rgb, 0 };
        st[t] = s;
        pthread_create(&th[t], NULL, strip_main, &st[t]);
    }
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        if (rays) *rays += st[t].rays;
    }
    free(th);
    free(st);
}

uint64_t orc_fnv1a64(const void *data, uint64_t nbytes)
{
    const uint8_t *p = (const uint8_t *)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < nbytes; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}
