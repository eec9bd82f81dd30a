// multi_gpu_rccl.cpp -- one frame over the GPUs of a node from a C++ host, through rt_multi_*:
// the reference's TraceScene (Source/main.cpp:647-659) with devices in place of its four threads.
// The library splits the frame (sample slices, reduced with ncclReduce; or interleaved 8x8 tiles,
// gathered with ncclSend/ncclRecv), runs every device from this one thread and assembles the
// picture on the first device; the frame is bit-identical to the single-GPU one.
//
//   make -C examples     (g++ -std=c++17 -I include ... -L ray-tracer_amd -lrt_mi355x)
//   ./multi_gpu_rccl [n_gpus] [samples|tiles] [rccl|peer] [out.ppm]
//
// n_gpus larger than the node's device count reuses devices round robin (ranks that share a device
// exchange by device-local copies), which is how this runs on a one-GPU machine.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_mi355x.h"

#define CHECK_RT(x) do { int s_ = (x); if (s_ != RT_OK) { std::fprintf(stderr, "%s: %s (%s)\n", #x, rt_status_string(s_), rt_multi_last_error(multi)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int n_ranks = argc > 1 && std::atoi(argv[1]) > 0 ? std::atoi(argv[1]) : 1;
    const int partition = argc > 2 && std::strcmp(argv[2], "tiles") == 0 ? RT_PARTITION_TILES : RT_PARTITION_SAMPLES;
    const int exchange = argc > 3 && std::strcmp(argv[3], "peer") == 0 ? RT_EXCHANGE_PEER_COPY
                       : argc > 3 && std::strcmp(argv[3], "rccl") == 0 ? RT_EXCHANGE_RCCL : RT_EXCHANGE_AUTO;

    // how many devices are there?  rt_create fails with RT_ERR_INVALID_ARGUMENT past the last one
    int n_dev = 0;
    for (rt_context *probe = nullptr; n_dev < 64 && rt_create(n_dev, &probe) == RT_OK; ++n_dev)
        rt_destroy(probe);
    if (n_dev == 0) { std::fprintf(stderr, "no HIP device\n"); return 1; }
    std::vector<int> ids(n_ranks);
    for (int i = 0; i < n_ranks; ++i) ids[i] = i % n_dev;

    rt_multi *multi = nullptr;
    const int st = rt_multi_create(ids.data(), (uint32_t)n_ranks, exchange, &multi);
    if (st != RT_OK) { std::fprintf(stderr, "rt_multi_create: %s\n", rt_status_string(st)); return 1; }

    // the reference's scene and camera (Source/main.cpp:520-526, 605-608) at 1080p, 64 spp
    const rt_sphere spheres[4] = {
        { { -1.0f, 0.0f, -1.0f }, { 0.8f, 0.6f, 0.2f }, 0.5f, RT_MAT_METAL, 0.65f },
        { { 0.0f, 0.0f, -1.0f }, { 255.0f, 0.0f, 0.0f }, 0.5f, RT_MAT_DIFFUSE, 0.0f },
        { { 1.0f, 0.0f, -1.0f }, { 0.8f, 0.8f, 0.8f }, 0.5f, RT_MAT_METAL, 0.0f },
        { { 0.0f, -100.5f, -1.0f }, { 255.0f, 215.0f, 10.0f }, 100.0f, RT_MAT_DIFFUSE, 0.0f },
    };
    const float from[3] = { 0, 0, 0 }, at[3] = { 0, 0, -1 }, up[3] = { 0, 1, 0 };
    rt_camera cam;
    rt_camera_init(from, at, up, 90.0f, &cam);
    CHECK_RT(rt_multi_set_scene(multi, spheres, 4));
    rt_params p;
    std::memset(&p, 0, sizeof p);
    p.width = 1920; p.height = 1080; p.spp = 64; p.max_depth = 10; p.seed = 1;
    std::vector<uint8_t> rgb((size_t)p.width * p.height * 3);

    CHECK_RT(rt_multi_render(multi, &cam, &p, partition, rgb.data())); // warm-up
    CHECK_RT(rt_multi_reset_stats(multi));
    const auto t0 = std::chrono::steady_clock::now();
    CHECK_RT(rt_multi_render(multi, &cam, &p, partition, rgb.data()));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    rt_stats stats;
    CHECK_RT(rt_multi_get_stats(multi, &stats));
    unsigned long long h = 0xcbf29ce484222325ull; // FNV-1a-64 of the frame, to compare runs
    for (uint8_t b : rgb) h = (h ^ b) * 0x100000001b3ull;
    std::printf("%d rank(s) on %d device(s), %s, %s: %ux%u, %u spp, %llu rays, %.2f ms host wall -> %.0f Mrays/s, frame fnv %016llx\n",
                n_ranks, n_ranks < n_dev ? n_ranks : n_dev, partition == RT_PARTITION_TILES ? "tiles gathered" : "sample slices reduced",
                rt_multi_uses_rccl(multi) ? "RCCL" : "peer copies", p.width, p.height, p.spp,
                (unsigned long long)stats.rays, ms, stats.rays / (ms * 1e-3) / 1e6, h);

    if (argc > 4) {
        if (FILE *f = std::fopen(argv[4], "wb")) {
            std::fprintf(f, "P6\n%u %u\n255\n", p.width, p.height);
            for (uint32_t y = 0; y < p.height; ++y) // row 0 is the bottom of the picture
                std::fwrite(rgb.data() + (size_t)(p.height - 1 - y) * p.width * 3, 1, (size_t)p.width * 3, f);
            std::fclose(f);
        }
    }
    rt_multi_destroy(multi);
    return 0;
}
