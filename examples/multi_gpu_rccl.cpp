// multi_gpu_rccl.cpp -- one frame over all GPUs of a node from a C++ host, without Python:
// one rt_context per device, the frame's samples sliced over the devices, the per-pixel integer
// sums reduced to device 0 with RCCL (ncclReduce over xGMI), device 0 divides and packs the frame.
// This is the C++ twin of ray-tracer_amd/python/dist_render.py (which runs one process per GPU);
// the frame is bit-identical to the single-GPU one for any number of devices.
//
//   hipcc -O2 -std=c++17 -I include examples/multi_gpu_rccl.cpp -L ray-tracer_amd -lrt_mi355x -lrccl \
//         -Wl,-rpath,$PWD/ray-tracer_amd -o multi_gpu_rccl && ./multi_gpu_rccl [n_gpus] [out.ppm]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_mi355x.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)
#define CHECK_RT(ctx, x) do { int s_ = (x); if (s_ != RT_OK) { std::fprintf(stderr, "%s: %s (%s)\n", #x, rt_status_string(s_), rt_last_error(ctx)); return 1; } } while (0)

int main(int argc, char **argv)
{
    int n_dev = 0;
    CHECK_HIP(hipGetDeviceCount(&n_dev));
    if (argc > 1 && std::atoi(argv[1]) > 0 && std::atoi(argv[1]) < n_dev)
        n_dev = std::atoi(argv[1]);
    if (n_dev < 1) { std::fprintf(stderr, "no HIP device\n"); return 1; }

    // the reference's scene and camera (Source/main.cpp:520-526, 605-608), 1080p, 64 spp per GPU
    const rt_sphere spheres[4] = {
        { { -1.0f, 0.0f, -1.0f }, { 0.8f, 0.6f, 0.2f }, 0.5f, RT_MAT_METAL, 0.65f },
        { { 0.0f, 0.0f, -1.0f }, { 255.0f, 0.0f, 0.0f }, 0.5f, RT_MAT_DIFFUSE, 0.0f },
        { { 1.0f, 0.0f, -1.0f }, { 0.8f, 0.8f, 0.8f }, 0.5f, RT_MAT_METAL, 0.0f },
        { { 0.0f, -100.5f, -1.0f }, { 255.0f, 215.0f, 10.0f }, 100.0f, RT_MAT_DIFFUSE, 0.0f },
    };
    const float from[3] = { 0, 0, 0 }, at[3] = { 0, 0, -1 }, up[3] = { 0, 1, 0 };
    rt_camera cam;
    rt_camera_init(from, at, up, 90.0f, &cam);
    const uint32_t W = 1920, H = 1080, spp_per_gpu = 64, spp = spp_per_gpu * (uint32_t)n_dev;
    const size_t n_sums = (size_t)W * H * 3;

    std::vector<int> devs(n_dev);
    for (int d = 0; d < n_dev; ++d) devs[d] = d;
    std::vector<ncclComm_t> comms(n_dev);
    CHECK_NCCL(ncclCommInitAll(comms.data(), n_dev, devs.data()));

    std::vector<rt_context *> ctx(n_dev, nullptr);
    std::vector<hipStream_t> stream(n_dev);
    std::vector<uint32_t *> sums(n_dev, nullptr);
    for (int d = 0; d < n_dev; ++d) {
        CHECK_HIP(hipSetDevice(d));
        CHECK_RT(nullptr, rt_create(d, &ctx[d]));
        CHECK_RT(ctx[d], rt_set_scene(ctx[d], spheres, 4));
        CHECK_HIP(hipStreamCreate(&stream[d]));
        CHECK_HIP(hipMalloc((void **)&sums[d], n_sums * sizeof(uint32_t)));
    }

    hipEvent_t t0, t1;
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipEventCreate(&t0));
    CHECK_HIP(hipEventCreate(&t1));
    uint8_t *d_rgb = nullptr;
    CHECK_HIP(hipMalloc((void **)&d_rgb, n_sums));

    for (int frame = 0; frame < 2; ++frame) { // the second frame is the timed one
        CHECK_HIP(hipSetDevice(0));
        CHECK_HIP(hipEventRecord(t0, stream[0]));
        for (int d = 0; d < n_dev; ++d) { // trace: every device its slice of the samples
            CHECK_HIP(hipSetDevice(d));
            CHECK_HIP(hipMemsetAsync(sums[d], 0, n_sums * sizeof(uint32_t), stream[d]));
            rt_params p;
            std::memset(&p, 0, sizeof p);
            p.width = W; p.height = H; p.spp = spp; p.max_depth = 10; p.seed = 1;
            p.sample_begin = (uint32_t)d * spp_per_gpu;
            p.sample_count = spp_per_gpu;
            CHECK_RT(ctx[d], rt_accumulate_device(ctx[d], &cam, &p, 0, 0, (int)W, (int)H, sums[d], stream[d]));
        }
        CHECK_NCCL(ncclGroupStart()); // the frame's only exchange: integer sums to device 0
        for (int d = 0; d < n_dev; ++d)
            CHECK_NCCL(ncclReduce(sums[d], sums[d], n_sums, ncclUint32, ncclSum, 0, comms[d], stream[d]));
        CHECK_NCCL(ncclGroupEnd());
        CHECK_HIP(hipSetDevice(0));
        CHECK_RT(ctx[0], rt_resolve_device(ctx[0], sums[0], W, H, spp, 0u, 0, 0, (int)W, (int)H, d_rgb, stream[0]));
        CHECK_HIP(hipEventRecord(t1, stream[0]));
        for (int d = 0; d < n_dev; ++d) {
            CHECK_HIP(hipSetDevice(d));
            CHECK_HIP(hipStreamSynchronize(stream[d]));
        }
    }
    float ms = 0.f;
    CHECK_HIP(hipEventElapsedTime(&ms, t0, t1));
    unsigned long long rays = 0;
    for (int d = 0; d < n_dev; ++d) {
        rt_stats st;
        CHECK_RT(ctx[d], rt_get_stats(ctx[d], &st));
        rays += st.rays / 2; // two frames were traced
    }
    std::printf("%d GPU(s): %ux%u, %u spp, %llu rays, %.2f ms on device 0's clock -> %.0f Mrays/s\n", n_dev, W, H, spp, rays,
                ms, rays / (ms * 1e-3) / 1e6);

    if (argc > 2) {
        std::vector<uint8_t> rgb(n_sums);
        CHECK_HIP(hipSetDevice(0));
        CHECK_HIP(hipMemcpy(rgb.data(), d_rgb, n_sums, hipMemcpyDeviceToHost));
        if (FILE *f = std::fopen(argv[2], "wb")) {
            std::fprintf(f, "P6\n%u %u\n255\n", W, H);
            for (uint32_t y = 0; y < H; ++y) // row 0 is the bottom of the picture
                std::fwrite(rgb.data() + (size_t)(H - 1 - y) * W * 3, 1, (size_t)W * 3, f);
            std::fclose(f);
        }
    }
    for (int d = 0; d < n_dev; ++d) {
        CHECK_HIP(hipSetDevice(d));
        (void)hipFree(sums[d]);
        (void)hipStreamDestroy(stream[d]);
        rt_destroy(ctx[d]);
        ncclCommDestroy(comms[d]);
    }
    return 0;
}
