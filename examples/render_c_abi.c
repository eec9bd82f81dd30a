/* render_c_abi.c -- the reference's shipped frame (Source/main.cpp:520-526, 605-611) through the C
 * ABI from plain C: what a binding in any language does.
 *
 *   gcc -std=c99 -I include examples/render_c_abi.c -L ray-tracer_amd -lrt_mi355x \
 *       -Wl,-rpath,$PWD/ray-tracer_amd -o render_c_abi && ./render_c_abi out.ppm
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_mi355x.h"

static int check(rt_context *ctx, int status, const char *what)
{
    if (status != RT_OK)
        fprintf(stderr, "%s: %s (%s)\n", what, rt_status_string(status), rt_last_error(ctx));
    return status;
}

int main(int argc, char **argv)
{
    /* std::vector<Sphere> Spheres, Source/main.cpp:520-526 */
    const rt_sphere spheres[4] = {
        { { -1.0f, 0.0f, -1.0f }, { 0.8f, 0.6f, 0.2f }, 0.5f, RT_MAT_METAL, 0.65f },
        { { 0.0f, 0.0f, -1.0f }, { 255.0f, 0.0f, 0.0f }, 0.5f, RT_MAT_DIFFUSE, 0.0f },
        { { 1.0f, 0.0f, -1.0f }, { 0.8f, 0.8f, 0.8f }, 0.5f, RT_MAT_METAL, 0.0f },
        { { 0.0f, -100.5f, -1.0f }, { 255.0f, 215.0f, 10.0f }, 100.0f, RT_MAT_DIFFUSE, 0.0f },
    };
    const float from[3] = { 0, 0, 0 }, at[3] = { 0, 0, -1 }, up[3] = { 0, 1, 0 };
    const uint32_t W = 1024, H = 576;
    rt_context *ctx = NULL;
    rt_camera cam;
    rt_params p;
    rt_stats st;
    uint8_t *rgb;
    FILE *f;
    uint32_t y;

    if (check(NULL, rt_create(0, &ctx), "rt_create") != RT_OK)
        return 1;
    if (check(ctx, rt_set_scene(ctx, spheres, 4), "rt_set_scene") != RT_OK)
        return 1;
    rt_camera_init(from, at, up, 90.0f, &cam); /* g_SceneCamera, :605-608 */
    memset(&p, 0, sizeof p);
    p.width = W; p.height = H; p.spp = 100; p.max_depth = 10; /* g_Width, g_Height, SPP, RAY_DEPTH */
    rgb = (uint8_t *)malloc((size_t)W * H * 3);
    memset(rgb, 255, (size_t)W * H * 3);       /* g_PixelData's memset, :175 */
    if (check(ctx, rt_render(ctx, &cam, &p, 0, 0, (int)W, (int)H, rgb), "rt_render") != RT_OK)
        return 1;
    rt_get_stats(ctx, &st);
    printf("%llu rays in %.2f ms\n", (unsigned long long)st.rays, st.last_wall_ms);
    if (argc > 1 && (f = fopen(argv[1], "wb")) != NULL) { /* row 0 is the bottom of the picture */
        fprintf(f, "P6\n%u %u\n255\n", W, H);
        for (y = 0; y < H; ++y)
            fwrite(rgb + (size_t)(H - 1 - y) * W * 3, 1, (size_t)W * 3, f);
        fclose(f);
    }
    free(rgb);
    rt_destroy(ctx);
    return 0;
}
