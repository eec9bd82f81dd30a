// RayTracerHost.h -- host-side mirror of the reference's ray-tracing interfaces
// (Source/main.cpp), kept so that host code written against the reference reads
// the same here: Ray (:323-352), Material (:354-360), Sphere (:362-391), Camera
// (:416-459), RGB (:45-62), the global scene `Spheres` (:520), g_SceneCamera
// (:605), PutPixel/GetPixel (:273-293) and TraceScene (:647-659).
//
// The difference is below TraceScene: instead of four detached CPU threads
// running TraceThreadFunction (:613-645), the frame is traced on an MI355X
// through the C ABI of include/rt_mi355x.h.  glm is replaced by a three-float
// struct; no host arithmetic other than the camera set-up (done inside the
// library by rt_camera_init) takes part in the image.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "rt_mi355x.h"

namespace RayTracer {

typedef unsigned char byte;

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float X, float Y, float Z) : x(X), y(Y), z(Z) {}
};

struct ivec2 {
    int x, y;
};

class RGB {
public:
    RGB(byte R, byte G, byte B) : r(R), g(G), b(B) {}
    RGB() : r(0), g(0), b(0) {}
    byte r, g, b;
};

class Ray {
public:
    Ray(const vec3 &origin, const vec3 &direction) : m_Origin(origin), m_Direction(direction) {}
    const vec3 &GetOrigin() const noexcept { return m_Origin; }
    const vec3 &GetDirection() const noexcept { return m_Direction; }

private:
    vec3 m_Origin, m_Direction;
};

enum class Material { Glass = 0, Diffuse, Metal, FuzzyMetal };

// Same members, order and size (36 bytes) as the reference's Sphere, hence as rt_sphere.
class Sphere {
public:
    vec3 Center;
    vec3 Color;
    float Radius;
    Material SphereMaterial;
    float FuzzLevel;

    Sphere(const vec3 &center, const vec3 &color, float radius, Material mat, float fuzz = 0.0f)
        : Center(center), Color(color), Radius(radius), SphereMaterial(mat), FuzzLevel(fuzz) {}
    Sphere() : Center(0.0f), Color(0.0f), Radius(0.0f), SphereMaterial(Material::Diffuse), FuzzLevel(0.0f) {}
};
static_assert(sizeof(Sphere) == sizeof(rt_sphere), "Sphere must match rt_sphere");

class Camera {
public:
    Camera(const vec3 &lookfrom, const vec3 &lookat, const vec3 &up, float fov) : m_FOV(fov)
    {
        const float f[3] = { lookfrom.x, lookfrom.y, lookfrom.z }, a[3] = { lookat.x, lookat.y, lookat.z },
                    u[3] = { up.x, up.y, up.z };
        if (rt_camera_init(f, a, u, fov, &m_Pod) != RT_OK)
            throw std::invalid_argument("rt_camera_init");
    }

    // Camera::GetRay, Source/main.cpp:440-444 (un-normalised direction, left-to-right sums).
    Ray GetRay(float u, float v) const
    {
        const rt_camera &c = m_Pod;
        vec3 d;
        d.x = ((c.bottom_left[0] + (c.horizontal[0] * u)) + (v * c.vertical[0])) - c.origin[0];
        d.y = ((c.bottom_left[1] + (c.horizontal[1] * u)) + (v * c.vertical[1])) - c.origin[1];
        d.z = ((c.bottom_left[2] + (c.horizontal[2] * u)) + (v * c.vertical[2])) - c.origin[2];
        return Ray(vec3(c.origin[0], c.origin[1], c.origin[2]), d);
    }

    const rt_camera &Pod() const { return m_Pod; }

private:
    rt_camera m_Pod;
    float m_FOV;
};

// The reference's globals (Source/main.cpp:74-76, 520, 605, 610-611) gathered in one object.
struct RenderState {
    unsigned Width = 1024, Height = 576; // g_Width, g_Height
    int SPP = 100, RAY_DEPTH = 10;       // :610-611
    uint64_t Seed = 0;
    uint32_t Flags = 0;                  // RT_FLAG_*: 0 = the reference's shading and output
    std::vector<Sphere> Spheres = {      // :520-526
        Sphere(vec3(-1.0f, 0.0f, -1.0f), vec3(0.8f, 0.6f, 0.2f), 0.5f, Material::Metal, 0.65f),
        Sphere(vec3(0.0f, 0.0f, -1.0f), vec3(255, 0, 0), 0.5f, Material::Diffuse),
        Sphere(vec3(1.0f, 0.0f, -1.0f), vec3(0.8f, 0.8f, 0.8f), 0.5f, Material::Metal, 0.0f),
        Sphere(vec3(0.0f, -100.5f, -1.0f), vec3(255, 215, 10), 100.0f, Material::Diffuse),
    };
    Camera SceneCamera = Camera(vec3(0.0f), vec3(0.0f, 0.0f, -1.0f), vec3(0.0f, 1.0f, 0.0f), 90.0f); // :605-608
    std::vector<byte> PixelData;         // g_PixelData: RGB8, row 0 = bottom

    void AllocatePixels() { PixelData.assign((size_t)Width * Height * 3, 255); } // memset 255, :175

    // PutPixel / GetPixel, Source/main.cpp:273-293
    void PutPixel(const ivec2 &loc, const RGB &col) noexcept
    {
        if (loc.x < 0 || loc.y < 0 || (unsigned)loc.x >= Width || (unsigned)loc.y >= Height) return;
        const size_t i = ((size_t)loc.x + (size_t)loc.y * Width) * 3;
        PixelData[i] = col.r; PixelData[i + 1] = col.g; PixelData[i + 2] = col.b;
    }
    RGB GetPixel(const ivec2 &loc) const
    {
        const size_t i = ((size_t)loc.x + (size_t)loc.y * Width) * 3;
        return RGB(PixelData[i], PixelData[i + 1], PixelData[i + 2]);
    }
};

// Owns the rt_context; TraceScene() is the reference's entry (Source/main.cpp:647-659).
class DeviceTracer {
public:
    explicit DeviceTracer(int device = 0)
    {
        const int st = rt_create(device, &m_Ctx);
        if (st != RT_OK)
            throw std::runtime_error(std::string("rt_create: ") + rt_status_string(st));
    }
    ~DeviceTracer() { if (m_Ctx) rt_destroy(m_Ctx); }
    DeviceTracer(const DeviceTracer &) = delete;
    DeviceTracer &operator=(const DeviceTracer &) = delete;

    // TraceThreadFunction(xstart, ystart, xsize, ysize), Source/main.cpp:613.
    void TraceThreadFunction(RenderState &s, int xstart, int ystart, int xsize, int ysize)
    {
        Check(rt_set_scene(m_Ctx, reinterpret_cast<const rt_sphere *>(s.Spheres.data()), (uint32_t)s.Spheres.size()));
        rt_params p;
        std::memset(&p, 0, sizeof p);
        p.width = s.Width; p.height = s.Height; p.spp = (uint32_t)s.SPP; p.max_depth = s.RAY_DEPTH; p.seed = s.Seed;
        p.flags = s.Flags;
        if (s.PixelData.size() != (size_t)s.Width * s.Height * 3) s.AllocatePixels();
        Check(rt_render(m_Ctx, &s.SceneCamera.Pod(), &p, xstart, ystart, xsize, ysize, s.PixelData.data()));
    }

    // TraceScene + DoRenderLoop's periodic upload (Source/main.cpp:295-321): PixelData holds the
    // average of the samples so far after every pass; OnPass returning false stops the trace.
    template <class F> void TraceSceneProgressive(RenderState &s, uint32_t SamplesPerPass, F &&OnPass)
    {
        Check(rt_set_scene(m_Ctx, reinterpret_cast<const rt_sphere *>(s.Spheres.data()), (uint32_t)s.Spheres.size()));
        rt_params p;
        std::memset(&p, 0, sizeof p);
        p.width = s.Width; p.height = s.Height; p.spp = (uint32_t)s.SPP; p.max_depth = s.RAY_DEPTH; p.seed = s.Seed;
        p.flags = s.Flags;
        if (s.PixelData.size() != (size_t)s.Width * s.Height * 3) s.AllocatePixels();
        auto thunk = [](void *user, uint32_t done, uint32_t total) -> int {
            return (*static_cast<typename std::remove_reference<F>::type *>(user))(done, total) ? 0 : 1;
        };
        Check(rt_render_progressive(m_Ctx, &s.SceneCamera.Pod(), &p, 0, 0, (int)s.Width, (int)s.Height,
                                    s.PixelData.data(), SamplesPerPass, thunk, &OnPass));
    }

    // TraceScene: the reference splits the frame into THREAD_SPAWN_COUNT column strips; one device
    // call covers them all (and, unlike the reference, the remainder columns of Width % 4).
    void TraceScene(RenderState &s) { TraceThreadFunction(s, 0, 0, (int)s.Width, (int)s.Height); }

    rt_stats Stats()
    {
        rt_stats st;
        Check(rt_get_stats(m_Ctx, &st));
        return st;
    }

private:
    void Check(int st)
    {
        if (st != RT_OK)
            throw std::runtime_error(std::string(rt_status_string(st)) + ": " + rt_last_error(m_Ctx));
    }
    rt_context *m_Ctx = nullptr;
};

// TraceScene over several devices: the reference starts THREAD_SPAWN_COUNT threads on column strips
// (Source/main.cpp:647-659); this starts the frame on every device of the list (rt_multi_*), by sample
// slices or by interleaved tiles, and the picture is assembled on the first device.
class MultiDeviceTracer {
public:
    explicit MultiDeviceTracer(const std::vector<int> &devices, int exchange = RT_EXCHANGE_AUTO)
    {
        const int st = rt_multi_create(devices.data(), (uint32_t)devices.size(), exchange, &m_Multi);
        if (st != RT_OK)
            throw std::runtime_error(std::string("rt_multi_create: ") + rt_status_string(st));
    }
    ~MultiDeviceTracer() { if (m_Multi) rt_multi_destroy(m_Multi); }
    MultiDeviceTracer(const MultiDeviceTracer &) = delete;
    MultiDeviceTracer &operator=(const MultiDeviceTracer &) = delete;

    void TraceScene(RenderState &s, int partition = RT_PARTITION_TILES)
    {
        Check(rt_multi_set_scene(m_Multi, reinterpret_cast<const rt_sphere *>(s.Spheres.data()), (uint32_t)s.Spheres.size()));
        rt_params p;
        std::memset(&p, 0, sizeof p);
        p.width = s.Width; p.height = s.Height; p.spp = (uint32_t)s.SPP; p.max_depth = s.RAY_DEPTH; p.seed = s.Seed;
        p.flags = s.Flags;
        if (s.PixelData.size() != (size_t)s.Width * s.Height * 3) s.AllocatePixels();
        Check(rt_multi_render(m_Multi, &s.SceneCamera.Pod(), &p, partition, s.PixelData.data()));
    }

    rt_stats Stats()
    {
        rt_stats st;
        Check(rt_multi_get_stats(m_Multi, &st));
        return st;
    }
    bool UsesRccl() const { return rt_multi_uses_rccl(m_Multi) != 0; }

private:
    void Check(int st)
    {
        if (st != RT_OK)
            throw std::runtime_error(std::string(rt_status_string(st)) + ": " + rt_multi_last_error(m_Multi));
    }
    rt_multi *m_Multi = nullptr;
};

// Binary PPM (P6), top row first: g_PixelData's row 0 is the BOTTOM of the picture
// (Source/main.cpp:273-281 with the quad's UVs, :220-225), so rows are flipped on output.
bool WritePPM(const std::string &path, const RenderState &s);

} // namespace RayTracer
