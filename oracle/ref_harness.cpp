// TEST INFRASTRUCTURE -- the "real reference" oracle.  Never shipped, never linked into the product.
//
// This translation unit #includes the UNMODIFIED reference headers from /root/reference/src
// (renderer.h, hittables/*.h, lights/*.h, utils/threadPool.h) at build time in the authoring
// container only; the binary lands in oracle/_ref/ (git-ignored).  No reference source is copied here.
//
// What it adds around the reference code:
//   * time() is defined in this executable and returns --seed-base, which pins rand.h:7
//     (srand(time(NULL) + seed)) so that a single-threaded run is deterministic;
//   * rand() is defined here as a counting forwarder to random() (glibc rand() is random());
//   * counting subclasses of HittablesList / Sphere / Plane / Cube tally queries and primitive tests;
//   * Cube is abstract in the reference (cube.h has no hasIntersect); the subclass below supplies one that
//     calls intersect(), which is what HittablesList::hasIntersect does anyway (hittablesList.h:46);
//   * PointLight is NOT a reference class (the reference only has DirectionalLight); it is this repo's
//     definition for BASELINE config C4 and is called by the reference's LightsList/TraceRay through
//     the virtual Light interface;
//   * the per-pixel driver follows main.cpp:298-319 (u,v mapping, per-pixel seed x+y*W, SAMPLES loop,
//     clamp, average, pixel::Set / emissionPixel::Set).
//
// Modes:  --threads 0  : single thread, counters on, deterministic (the oracle proper)
//         --threads N  : the reference's own ThreadPool with 16x16 tiles (main.cpp:268-331), timing only
//                        (shared libc RNG => image not reproducible; that is the reference's behaviour).
//         --probe      : per-pixel MaxRayDistance (tests/singlePixel.cpp:82) instead of a render.

#include <ctime>
#include <cstdlib>
#include <cstdint>
#include <cfloat>
#include <cstring>

static long   g_seed_base   = 0;
static bool   g_count       = true;
static unsigned long long g_draws = 0, g_closest = 0, g_shadow = 0, g_tests = 0;
static int    g_in_shadow   = 0;

extern "C" time_t time(time_t* t) noexcept { if (t) *t = (time_t)g_seed_base; return (time_t)g_seed_base; }
extern "C" int rand(void) noexcept { if (g_count) ++g_draws; return (int)random(); }

#include "renderer.h"
#include "lights/directionalLight.h"
#include "lights/lightsList.h"
#include "hittables/hittablesList.h"
#include "hittables/sphere.h"
#include "hittables/plane.h"
#include "hittables/cube.h"
#include "utils/threadPool.h"

#include <chrono>
#include <vector>
#include <atomic>

extern "C" {
#include "scene_text.h"
}

struct CSphere : Sphere {
    using Sphere::Sphere;
    bool intersect(const Ray& r, RayHit& h) const override { if (g_count) ++g_tests; return Sphere::intersect(r, h); }
};
struct CPlane : Plane {
    using Plane::Plane;
    bool intersect(const Ray& r, RayHit& h) const override { if (g_count) ++g_tests; return Plane::intersect(r, h); }
};
struct CCube : Cube {
    using Cube::Cube;
    bool intersect(const Ray& r, RayHit& h) const override { if (g_count) ++g_tests; return Cube::intersect(r, h); }
    bool hasIntersect(const Ray& r) const override { RayHit h; return intersect(r, h); }
};
struct CList : HittablesList {
    using HittablesList::HittablesList;
    bool intersect(const Ray& r, RayHit& h) const
    {
        if (g_count) { if (g_in_shadow) ++g_shadow; else ++g_closest; }
        return HittablesList::intersect(r, h);
    }
    bool hasIntersect(const Ray& r) const override { if (g_count) ++g_shadow; return HittablesList::hasIntersect(r); }
};

// Repo-defined point light (config C4).  In light <=> no hittable lies strictly closer than the light
// along normalize(Lpos - p); intensity is Lambert max(0, n.l).
struct PointLight : Light {
    glm::vec3 m_Pos;
    explicit PointLight(const glm::vec3& p) : m_Pos(p) {}
    bool IsInLight(Hittable** world, const glm::vec3& position) const override
    {
        glm::vec3 d = m_Pos - position;
        float dist = glm::length(d);
        glm::vec3 l = glm::normalize(d);
        RayHit hit;
        ++g_in_shadow;
        bool h = (*world)->intersect(Ray{ position, l }, hit);
        --g_in_shadow;
        return !(h && hit.distance < dist);
    }
    void GetLightIntensity(Hittable**, const glm::vec3& position, const glm::vec3& normal, float& intensity) const override
    {
        glm::vec3 l = glm::normalize(m_Pos - position);
        float d = glm::dot(normal, l);
        intensity = d > 0 ? d : 0;
    }
};

static void write_ppm(const char* path, const pixel* img, int w, int h)
{
    FILE* f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    fwrite(img, sizeof(pixel), (size_t)w * h, f);
    fclose(f);
}

static inline void render_pixel(int x, int y, int W, int H, int samples, int maxdepth, Camera* camera,
                                Hittable** world, Light** lights, Material* materials, pixel* px, emissionPixel* em)
{
    float u = ((float)x / (float)W) * 2.0f - 1.0f;
    float v = ((float)y / (float)H) * 2.0f - 1.0f;
    curandState randState(x + y * W);
    float offX = 0.5f / W, offY = 0.5f / H;
    HitColorGlow acc;
    for (int i = 0; i < samples; ++i) {
        HitColorGlow s = AntiAliasing(u, v, offX, offY, camera, world, lights, materials, &randState, maxdepth);
        acc.color            += glm::clamp(s.color,    glm::vec3(0.0f), glm::vec3(1.0f));
        acc.emission         += glm::clamp(s.emission, glm::vec3(0.0f), glm::vec3(1.0f));
        acc.emissionStrenght += s.emissionStrenght;
    }
    if (px) px->Set(acc.color / glm::vec3(samples));
    if (em) em->Set(acc.emission / glm::vec3(samples), acc.emissionStrenght / samples);
}

int main(int argc, char** argv)
{
    const char* scene_path = nullptr; const char* out_ppm = nullptr; const char* out_em = nullptr;
    const char* out_cam = nullptr; const char* out_probe = nullptr;
    int W = 256, H = 128, samples = 5, maxdepth = 10, threads = 0, probe = 0, reps = 1;
    int row0 = 0, rows = -1, row_step = 1;
    for (int i = 1; i < argc; ++i) {
        auto is = [&](const char* k) { return !strcmp(argv[i], k) && i + 1 < argc; };
        if      (is("--scene"))     scene_path = argv[++i];
        else if (is("--out"))       out_ppm = argv[++i];
        else if (is("--emission"))  out_em = argv[++i];
        else if (is("--camera"))    out_cam = argv[++i];
        else if (is("--probe-out")) out_probe = argv[++i];
        else if (is("--w"))         W = atoi(argv[++i]);
        else if (is("--h"))         H = atoi(argv[++i]);
        else if (is("--samples"))   samples = atoi(argv[++i]);
        else if (is("--depth"))     maxdepth = atoi(argv[++i]);
        else if (is("--seed-base")) g_seed_base = atol(argv[++i]);
        else if (is("--threads"))   threads = atoi(argv[++i]);
        else if (is("--reps"))      reps = atoi(argv[++i]);
        else if (is("--row0"))      row0 = atoi(argv[++i]);
        else if (is("--rows"))      rows = atoi(argv[++i]);
        else if (is("--row-step"))  row_step = atoi(argv[++i]);   // sampled rows y = row0 + i*row_step (timing samples)
        else if (!strcmp(argv[i], "--probe")) probe = 1;
        else { fprintf(stderr, "unknown arg %s\n", argv[i]); return 2; }
    }
    if (!scene_path) { fprintf(stderr, "usage: ref_oracle --scene S [--w --h --samples --depth --seed-base --out --emission --threads]\n"); return 2; }
    if (rows < 0) rows = (H - row0 + row_step - 1) / row_step;

    st_scene sc;
    if (!st_load(scene_path, &sc)) return 2;

    Camera camera(sc.fov, (float)W, (float)H, sc.znear, sc.zfar);
    if (out_cam) {
        // 3 + 16 + 16 floats: position, inverse projection (column major), inverse view (column major)
        FILE* f = fopen(out_cam, "wb");
        fwrite(&camera.GetPosition()[0], 4, 3, f);
        fwrite(&camera.GetInverseProjectionMatrix()[0][0], 4, 16, f);
        fwrite(&camera.GetInverseViewMatrix()[0][0], 4, 16, f);
        fclose(f);
    }

    Material* materials = new Material[sc.n_materials > 0 ? sc.n_materials : 1];
    for (int i = 0; i < sc.n_materials; ++i) {
        const st_material& m = sc.materials[i];
        materials[i] = Material{ { m.color[0], m.color[1], m.color[2] }, m.roughness, m.reflection, m.refraction,
                                 { m.emission[0], m.emission[1], m.emission[2] }, m.glow };
    }
    Hittable** l_world = new Hittable*[sc.n_hittables > 0 ? sc.n_hittables : 1];
    for (int i = 0; i < sc.n_hittables; ++i) {
        const st_hittable& h = sc.hittables[i];
        glm::vec3 a{ h.a[0], h.a[1], h.a[2] }, b{ h.b[0], h.b[1], h.b[2] };
        if (h.type == ST_SPHERE)     l_world[i] = new CSphere(a, h.radius, (uint8_t)h.material);
        else if (h.type == ST_PLANE) l_world[i] = new CPlane(a, b, (uint8_t)h.material);
        else                         l_world[i] = new CCube(a, b, (uint32_t)h.material);
    }
    Hittable* world = new CList(l_world, sc.n_hittables);
    Light** l_light = new Light*[sc.n_lights > 0 ? sc.n_lights : 1];
    for (int i = 0; i < sc.n_lights; ++i) {
        glm::vec3 v{ sc.lights[i].v[0], sc.lights[i].v[1], sc.lights[i].v[2] };
        if (sc.lights[i].type == ST_DIRLIGHT) l_light[i] = new DirectionalLight(v);
        else                                  l_light[i] = new PointLight(v);
    }
    Light* lights = new LightsList(l_light, sc.n_lights);

    std::vector<pixel> image((size_t)W * H);
    std::vector<emissionPixel> emission((size_t)W * H);

    double ms = 0.0;
    if (probe) {
        // tests/singlePixel.cpp:65-90: one centre ray per pixel, flag pixels whose maxDepth reached the cap.
        std::vector<unsigned char> flags((size_t)W * H);
        int any = 0;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float u = ((float)x / (float)W) * 2.0f - 1.0f;
                float v = ((float)y / (float)H) * 2.0f - 1.0f;
                curandState randState(x + y * W);
                bool m = MaxRayDistance(u, v, &camera, &world, &lights, materials, &randState, maxdepth);
                flags[x + (size_t)y * W] = m; any |= m;
            }
        if (out_probe) { FILE* f = fopen(out_probe, "wb"); fwrite(flags.data(), 1, flags.size(), f); fclose(f); }
        printf("{\"mode\": \"probe\", \"w\": %d, \"h\": %d, \"depth\": %d, \"seed_base\": %ld, \"any_max_depth\": %d, \"rand_draws\": %llu}\n",
               W, H, maxdepth, g_seed_base, any, g_draws);
        return 0;
    }

    if (threads <= 0) {
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < reps; ++rep)
            for (int i = 0; i < rows; ++i)
                for (int x = 0; x < W; ++x) {
                    int y = row0 + i * row_step;
                    render_pixel(x, y, W, H, samples, maxdepth, &camera, &world, &lights, materials,
                                 &image[x + (size_t)y * W], &emission[x + (size_t)y * W]);
                }
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    } else {
        g_count = false;
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < reps; ++rep) {
            const int tile = 16;
            int splitH = (W + tile - 1) / tile, splitV = (rows + tile - 1) / tile;
            std::atomic<int> done{ 0 };
            {
                ThreadPool pool(threads);
                for (int sY = 0; sY < splitV; ++sY)
                    for (int sX = 0; sX < splitH; ++sX)
                        pool.enqueue([&, sX, sY]() {
                            for (int nY = 0; nY < tile; ++nY)
                                for (int nX = 0; nX < tile; ++nX) {
                                    int x = nX + sX * tile, i = nY + sY * tile;
                                    if (x >= W || i >= rows) continue;
                                    int y = row0 + i * row_step;
                                    render_pixel(x, y, W, H, samples, maxdepth, &camera, &world, &lights, materials,
                                                 &image[x + (size_t)y * W], &emission[x + (size_t)y * W]);
                                }
                            ++done;
                        });
            }   // pool destructor joins the workers (threadPool.h:72-84)
        }
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    }

    if (out_ppm) write_ppm(out_ppm, image.data(), W, H);
    if (out_em) { FILE* f = fopen(out_em, "wb"); fwrite(emission.data(), sizeof(emissionPixel), emission.size(), f); fclose(f); }

    unsigned long long primary = (unsigned long long)W * rows * samples * 4ull;
    printf("{\"mode\": \"render\", \"w\": %d, \"h\": %d, \"samples\": %d, \"depth\": %d, \"seed_base\": %ld, \"row0\": %d, \"rows\": %d, "
           "\"threads\": %d, \"ms\": %.3f, \"primary\": %llu, \"closest\": %llu, \"shadow\": %llu, \"tests\": %llu, \"rand_draws\": %llu}\n",
           W, H, samples, maxdepth, g_seed_base, row0, rows, threads, ms, primary,
           g_closest / reps, g_shadow / reps, g_tests / reps, g_draws / reps);
    return 0;
}
