// main.cpp -- the reference's program (Source/main.cpp:668-679) without the
// window: set up the scene and camera, trace the frame on the GPU through
// TraceScene(), write the picture to a PPM file.
//
//   rt_render_cli [--width W --height H --spp N --depth D --seed S]
//                 [--scene spheres.bin] [--camera fx fy fz ax ay az ux uy uz fov]
//                 [--out image.ppm] [--progressive SAMPLES_PER_PASS]
//                 [--devices 0,1,2,... [--partition tiles|samples] [--exchange auto|rccl|peer]]
//                 [--dielectric] [--gamma2] [--float-shading]   (extensions, off by default)
// Defaults are the reference's shipped constants: 1024x576, 100 spp, depth 10,
// its four spheres and its camera.
#include "RayTracerHost.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

namespace RayTracer {

bool WritePPM(const std::string &path, const RenderState &s)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f << "P6\n" << s.Width << " " << s.Height << "\n255\n";
    for (unsigned y = 0; y < s.Height; ++y) {
        const size_t row = (size_t)(s.Height - 1 - y) * s.Width * 3;
        f.write(reinterpret_cast<const char *>(s.PixelData.data() + row), (std::streamsize)s.Width * 3);
    }
    return (bool)f;
}

} // namespace RayTracer

using namespace RayTracer;

int main(int argc, char **argv)
{
    RenderState state;
    std::string out = "render.ppm";
    unsigned per_pass = 0;
    std::vector<int> devices;
    int partition = RT_PARTITION_TILES, exchange = RT_EXCHANGE_AUTO;
    try {
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            auto need = [&](int n) { if (i + n >= argc) throw std::invalid_argument("missing value for " + a); };
            if (a == "--width") { need(1); state.Width = (unsigned)std::atoi(argv[++i]); }
            else if (a == "--height") { need(1); state.Height = (unsigned)std::atoi(argv[++i]); }
            else if (a == "--spp") { need(1); state.SPP = std::atoi(argv[++i]); }
            else if (a == "--depth") { need(1); state.RAY_DEPTH = std::atoi(argv[++i]); }
            else if (a == "--seed") { need(1); state.Seed = std::strtoull(argv[++i], nullptr, 0); }
            else if (a == "--out") { need(1); out = argv[++i]; }
            else if (a == "--progressive") { need(1); per_pass = (unsigned)std::atoi(argv[++i]); }
            else if (a == "--devices") {
                need(1);
                for (const char *c = argv[++i]; *c;) {
                    char *end = nullptr;
                    devices.push_back((int)std::strtol(c, &end, 10));
                    if (end == c) throw std::invalid_argument("--devices takes a comma-separated list of device ids");
                    c = *end == ',' ? end + 1 : end;
                }
            }
            else if (a == "--partition") {
                need(1);
                const std::string v = argv[++i];
                if (v != "tiles" && v != "samples") throw std::invalid_argument("--partition takes tiles or samples");
                partition = v == "tiles" ? RT_PARTITION_TILES : RT_PARTITION_SAMPLES;
            }
            else if (a == "--exchange") {
                need(1);
                const std::string v = argv[++i];
                if (v != "auto" && v != "rccl" && v != "peer") throw std::invalid_argument("--exchange takes auto, rccl or peer");
                exchange = v == "rccl" ? RT_EXCHANGE_RCCL : v == "peer" ? RT_EXCHANGE_PEER_COPY : RT_EXCHANGE_AUTO;
            }
            else if (a == "--dielectric") state.Flags |= RT_FLAG_DIELECTRIC;
            else if (a == "--gamma2") state.Flags |= RT_FLAG_GAMMA2;
            else if (a == "--float-shading") state.Flags |= RT_FLAG_FLOAT_SHADING;
            else if (a == "--scene") {
                need(1);
                std::ifstream f(argv[++i], std::ios::binary | std::ios::ate);
                if (!f) throw std::invalid_argument("cannot open scene file");
                const std::streamsize bytes = f.tellg();
                f.seekg(0);
                state.Spheres.assign((size_t)bytes / sizeof(Sphere), Sphere());
                f.read(reinterpret_cast<char *>(state.Spheres.data()), bytes);
            } else if (a == "--camera") {
                need(10);
                float v[10];
                for (float &x : v) x = std::strtof(argv[++i], nullptr);
                state.SceneCamera = Camera(vec3(v[0], v[1], v[2]), vec3(v[3], v[4], v[5]), vec3(v[6], v[7], v[8]), v[9]);
            } else {
                throw std::invalid_argument("unknown option " + a);
            }
        }

        state.AllocatePixels();
        if (!devices.empty()) { // the frame over several devices, assembled on the first
            if (per_pass > 0) throw std::invalid_argument("--progressive renders on one device");
            MultiDeviceTracer multi(devices, exchange);
            std::cout << std::endl << "Writing Pixel Data.." << std::endl;
            std::cout << "Ray Tracing.." << std::endl;
            const auto t0 = std::chrono::steady_clock::now();
            multi.TraceScene(state, partition);
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const rt_stats st = multi.Stats();
            std::printf("%ux%u, %d spp, depth %d, %zu spheres on %zu rank(s), %s, %s: %llu rays in %.3f s (%.1f Mrays/s)\n",
                        state.Width, state.Height, state.SPP, state.RAY_DEPTH, state.Spheres.size(), devices.size(),
                        partition == RT_PARTITION_TILES ? "tiles gathered" : "sample slices reduced",
                        multi.UsesRccl() ? "RCCL" : "peer copies", (unsigned long long)st.rays, sec, st.rays / sec / 1e6);
            if (!WritePPM(out, state)) {
                std::fprintf(stderr, "cannot write %s\n", out.c_str());
                return 2;
            }
            std::printf("wrote %s\n", out.c_str());
            return 0;
        }
        DeviceTracer tracer(0);
        std::cout << std::endl << "Writing Pixel Data.." << std::endl; // Source/main.cpp:663-664
        std::cout << "Ray Tracing.." << std::endl;
        const auto t0 = std::chrono::steady_clock::now();
        if (per_pass > 0) {
            // the reference re-uploads g_PixelData every 20 frames while tracing (:302-305);
            // here each pass leaves a complete picture, written over the same file
            tracer.TraceSceneProgressive(state, per_pass, [&](uint32_t done, uint32_t total) {
                std::printf("  %u / %u samples\n", done, total);
                return WritePPM(out, state);
            });
        } else {
            tracer.TraceScene(state);
        }
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const rt_stats st = tracer.Stats();
        std::printf("%ux%u, %d spp, depth %d, %zu spheres: %llu rays in %.3f s (%.1f Mrays/s, kernels %.3f ms)\n",
                    state.Width, state.Height, state.SPP, state.RAY_DEPTH, state.Spheres.size(),
                    (unsigned long long)st.rays, sec, st.rays / sec / 1e6, st.last_kernel_ms);
        if (!WritePPM(out, state)) {
            std::fprintf(stderr, "cannot write %s\n", out.c_str());
            return 2;
        }
        std::printf("wrote %s\n", out.c_str());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
