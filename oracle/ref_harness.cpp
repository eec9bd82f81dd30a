// ref_harness.cpp -- drives the REAL reference implementation headless.
//
// TEST INFRASTRUCTURE ONLY.  This translation unit includes the reference's
// single source file from where it lies (/root/reference/Source/main.cpp; the
// include path is given by oracle/Makefile) and calls its hot-path functions
// (Camera::GetRay :440, IntersectSceneSpheres :528, GetRayColor :551,
// TraceThreadFunction :613).  Nothing of the reference is copied into this
// repository; the built binary goes to oracle/_ref/ (git-ignored).  It is used
// to (1) pin oracle/rt_oracle.c, (2) regenerate tests/golden/, and (3) serve as
// the "reference" CPU baseline of bench.py.
//
// The reference's OpenGL/GLFW/ImGui symbols are never called; the link leaves
// them unresolved (see oracle/Makefile).
#define main reference_main
#include "main.cpp"
#undef main

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>

static uint64_t fnv1a64(const void* data, size_t n)
{
	const unsigned char* p = (const unsigned char*)data;
	uint64_t h = 0xcbf29ce484222325ull;
	for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
	return h;
}

static void write_file(const char* path, const void* data, size_t n)
{
	FILE* f = fopen(path, "wb");
	if (!f) { perror(path); _exit(2); }
	fwrite(data, 1, n, f);
	fclose(f);
}

static std::vector<unsigned char> read_file(const char* path)
{
	FILE* f = fopen(path, "rb");
	if (!f) { perror(path); _exit(2); }
	std::vector<unsigned char> buf;
	unsigned char tmp[65536];
	size_t k;
	while ((k = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + k);
	fclose(f);
	return buf;
}

// Scene files are arrays of the reference's own Sphere layout (36 bytes).
static void load_scene(const char* path)
{
	std::vector<unsigned char> buf = read_file(path);
	size_t n = buf.size() / sizeof(Sphere);
	Spheres.clear();
	Spheres.resize(n);
	memcpy((void*)Spheres.data(), buf.data(), n * sizeof(Sphere));
}

static int sphere_index(const Sphere& s)
{
	for (int i = (int)Spheres.size() - 1; i >= 0; i--)
		if (memcmp(&Spheres[i], &s, sizeof(Sphere)) == 0) return i;
	return -2;
}

// Camera has const members (no assignment), so parsed cameras live on the heap.
static const Camera* parse_camera(char** a)
{
	float f[10];
	for (int i = 0; i < 10; i++) f[i] = strtof(a[i], nullptr);
	return new Camera(glm::vec3(f[0], f[1], f[2]), glm::vec3(f[3], f[4], f[5]), glm::vec3(f[6], f[7], f[8]), f[9]);
}

// Ray count of the reference's own loops.  The reference keeps no counter, and its source is not
// touched: the Makefile's second build of this file (_ref/ref_harness_count) adds gcc's
// -finstrument-functions, which calls the two hooks below around every function of this
// translation unit (also where a function was inlined); entries of IntersectSceneSpheres are the
// rays (one per GetRayColor call with depth > 0, Source/main.cpp:562).
#ifdef RT_COUNT_CALLS
static unsigned long long g_IntersectCalls = 0;
extern "C" __attribute__((no_instrument_function)) void __cyg_profile_func_enter(void* fn, void*)
{
	if (fn == (void*)&IntersectSceneSpheres) g_IntersectCalls++;
}
extern "C" __attribute__((no_instrument_function)) void __cyg_profile_func_exit(void*, void*) {}
#endif

[[noreturn]] static void done()
{
	fflush(stdout);
	_exit(0); // skip ~Application (it would call into ImGui/GLFW)
}

int main(int argc, char** argv)
{
	if (argc < 2) {
		fprintf(stderr,
			"usage:\n"
			"  kat                                  sizes, RandomFloat draws, centre ray\n"
			"  c1 <xsize> <out.rgb>                 TraceThreadFunction(0,0,xsize,576)\n"
			"  hits <W> <H> <out.i32> [scene cam*10] primary closest-hit map, jitter 0.5\n"
			"  render <scene|-> <cam*10|-> <W> <H> <spp> <depth> <x0> <xs> <out.rgb|->  sample loop of :613-645\n"
			"  camera <cam*10>                      print origin/bottom_left/horizontal/vertical via GetRay\n"
			"  sky <n> <in.f32x3> <out.u8x3>        GetGradientColorAtRay\n"
			"  isect <scene|-> <n> <rays.f32x6> <out>  IntersectSceneSpheres per ray\n");
		return 1;
	}
	std::string mode = argv[1];

	if (mode == "kat") {
		printf("sizeof Sphere %zu Ray %zu RayHitRecord %zu RGB %zu\n", sizeof(Sphere), sizeof(Ray), sizeof(RayHitRecord), sizeof(RGB));
		Ray r = g_SceneCamera.GetRay(0.5f, 0.5f);
		RayHitRecord rec; Sphere s;
		bool hit = IntersectSceneSpheres(r, 0.001f, _INFINITY, rec, s);
		printf("centre dir %.9g %.9g %.9g hit %d index %d T %.9g N %.9g %.9g %.9g\n",
			r.GetDirection().x, r.GetDirection().y, r.GetDirection().z, (int)hit, sphere_index(s), rec.T,
			rec.Normal.x, rec.Normal.y, rec.Normal.z);
		for (int i = 0; i < 16; i++) {
			float f = (float)RandomFloat();
			uint32_t b; memcpy(&b, &f, 4);
			printf("draw %d %.9g 0x%08x\n", i, f, b);
		}
		done();
	}

	if (mode == "c1") {
		int xsize = atoi(argv[2]);
		auto t0 = std::chrono::steady_clock::now();
		TraceThreadFunction(0, 0, xsize, g_Height);
		double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		size_t nbytes = (size_t)g_Width * g_Height * 3;
		printf("c1 xsize %d fnv %016llx seconds %.3f\n", xsize, (unsigned long long)fnv1a64(g_PixelData, nbytes), sec);
#ifdef RT_COUNT_CALLS
		printf("rays %llu\n", g_IntersectCalls);
#endif
		if (argc > 3 && strcmp(argv[3], "-") != 0) write_file(argv[3], g_PixelData, nbytes);
		done();
	}

	if (mode == "hits") {
		int W = atoi(argv[2]), H = atoi(argv[3]);
		const Camera* camp = &g_SceneCamera;
		if (argc > 5) { load_scene(argv[5]); camp = parse_camera(argv + 6); }
		const Camera& cam = *camp;
		std::vector<int32_t> map((size_t)W * H);
		for (int j = 0; j < H; j++)
			for (int i = 0; i < W; i++) {
				float u = (float)(((double)(float)i + 0.5) / (double)(float)W);
				float v = (float)(((double)(float)j + 0.5) / (double)(float)H);
				Ray ray = cam.GetRay(u, v);
				RayHitRecord rec; Sphere s;
				bool hit = IntersectSceneSpheres(ray, 0.001f, _INFINITY, rec, s);
				map[(size_t)i + (size_t)j * W] = hit ? sphere_index(s) : -1;
			}
		printf("hits %dx%d fnv %016llx\n", W, H, (unsigned long long)fnv1a64(map.data(), map.size() * 4));
		if (strcmp(argv[4], "-") != 0) write_file(argv[4], map.data(), map.size() * 4);
		done();
	}

	if (mode == "render") {
		// argv: 2 scene, 3..12 camera (or a single "-"), then W H spp depth x0 xs out
		int a = 2;
		if (strcmp(argv[a], "-") != 0) load_scene(argv[a]);
		a++;
		const Camera* camp = &g_SceneCamera;
		if (strcmp(argv[a], "-") != 0) { camp = parse_camera(argv + a); a += 10; } else a++;
		const Camera& cam = *camp;
		int W = atoi(argv[a]), H = atoi(argv[a + 1]), spp = atoi(argv[a + 2]), depth = atoi(argv[a + 3]);
		int x0 = atoi(argv[a + 4]), xs = atoi(argv[a + 5]);
		const char* out = argv[a + 6];
		std::vector<unsigned char> img((size_t)W * H * 3, 255);
		auto t0 = std::chrono::steady_clock::now();
		// The sample loop of TraceThreadFunction (Source/main.cpp:613-645) with its
		// compile-time sizes turned into variables; everything it calls is the reference's.
		for (int i = x0; i < x0 + xs; i++)
			for (int j = 0; j < H; j++) {
				glm::ivec3 FinalColor;
				for (int s = 0; s < spp; s++) {
					float u = ((float)i + RandomFloat()) / (float)(uint)W;
					float v = ((float)j + RandomFloat()) / (float)(uint)H;
					Ray ray = cam.GetRay(u, v);
					RGB c = GetRayColor(ray, depth);
					FinalColor.r += c.r; FinalColor.g += c.g; FinalColor.b += c.b;
				}
				FinalColor.r = FinalColor.r / spp; FinalColor.g = FinalColor.g / spp; FinalColor.b = FinalColor.b / spp;
				RGB px = ToRGB(FinalColor);
				size_t loc = ((size_t)i + (size_t)j * W) * 3;
				img[loc] = px.r; img[loc + 1] = px.g; img[loc + 2] = px.b;
			}
		double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		printf("render %dx%d spp %d depth %d x0 %d xs %d spheres %zu fnv %016llx seconds %.6f\n", W, H, spp, depth, x0, xs,
			Spheres.size(), (unsigned long long)fnv1a64(img.data(), img.size()), sec);
		if (strcmp(out, "-") != 0) write_file(out, img.data(), img.size());
		done();
	}

	if (mode == "camera") {
		const Camera& cam = *parse_camera(argv + 2);
		// origin, and GetRay at the corners recover the members exactly enough for a KAT print
		Ray r00 = cam.GetRay(0.0f, 0.0f), r10 = cam.GetRay(1.0f, 0.0f), r01 = cam.GetRay(0.0f, 1.0f), r55 = cam.GetRay(0.5f, 0.5f);
		const Ray* rs[4] = { &r00, &r10, &r01, &r55 };
		for (int k = 0; k < 4; k++) {
			glm::vec3 d = rs[k]->GetDirection(), o = rs[k]->GetOrigin();
			uint32_t b[6]; memcpy(b, &o, 12); memcpy(b + 3, &d, 12);
			printf("ray %d %08x %08x %08x %08x %08x %08x\n", k, b[0], b[1], b[2], b[3], b[4], b[5]);
		}
		done();
	}

	if (mode == "sky") {
		int n = atoi(argv[2]);
		std::vector<unsigned char> in = read_file(argv[3]);
		std::vector<unsigned char> out((size_t)n * 3);
		const float* f = (const float*)in.data();
		for (int i = 0; i < n; i++) {
			Ray ray(glm::vec3(0.0f), glm::vec3(f[i * 3], f[i * 3 + 1], f[i * 3 + 2]));
			RGB c = GetGradientColorAtRay(ray);
			out[i * 3] = c.r; out[i * 3 + 1] = c.g; out[i * 3 + 2] = c.b;
		}
		write_file(argv[4], out.data(), out.size());
		done();
	}

	if (mode == "isect") {
		if (strcmp(argv[2], "-") != 0) load_scene(argv[2]);
		int n = atoi(argv[3]);
		std::vector<unsigned char> in = read_file(argv[4]);
		const float* f = (const float*)in.data();
		// per ray: int32 index, float T, float P[3], float N[3]  (32 bytes)
		std::vector<unsigned char> out((size_t)n * 32, 0);
		for (int i = 0; i < n; i++) {
			Ray ray(glm::vec3(f[i * 6], f[i * 6 + 1], f[i * 6 + 2]), glm::vec3(f[i * 6 + 3], f[i * 6 + 4], f[i * 6 + 5]));
			RayHitRecord rec; Sphere s;
			bool hit = IntersectSceneSpheres(ray, 0.001f, _INFINITY, rec, s);
			int32_t idx = hit ? sphere_index(s) : -1;
			unsigned char* o = out.data() + (size_t)i * 32;
			memcpy(o, &idx, 4);
			if (hit) { memcpy(o + 4, &rec.T, 4); memcpy(o + 8, &rec.Point, 12); memcpy(o + 20, &rec.Normal, 12); }
		}
		write_file(argv[5], out.data(), out.size());
		done();
	}

	fprintf(stderr, "unknown mode %s\n", argv[1]);
	return 1;
}
