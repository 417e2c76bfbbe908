// renderer.h -- host side of the MI355X backend, the counterpart of the reference's src/renderer.h.
//
// The reference's renderer.h holds the per-ray CPU functions (UVToDirection :21, TraceRay :28, AntiAliasing :138,
// MaxRayDistance :204) that every driver calls from its own pixel loop (src/main.cpp:295-320,
// src/tests/speedUp.cpp:91-113).  Here that loop nest runs on the GPU: the functions below take the same scene
// objects (Camera, Hittable* world, Light* lights, Material* materials) and hand back the filled pixel[] /
// emissionPixel[] buffers, by flattening the scene and calling the C-ABI of include/rt_amd.h.
//
//   RenderImage()      = the tile loop + recompose of src/main.cpp:285-395 (image and emission out)
//   Render()           = double Render(Camera&, Hittable*, Light*, Material*, int threads) of
//                        src/tests/speedUp.cpp:26,69 (same signature, returns elapsed ms, discards the pixels)
//   MaxRayDistanceAny()= the per-pixel MaxRayDistance sweep of src/tests/singlePixel.cpp:65-120
//   RenderFrames()     = a sequence of frames with several under way at once: the reference's producer / consumer
//                        shape (src/main.cpp:268-331 workers, :375-395 consumer) applied to whole frames
//
// Errors: the reference signals none on this path; these functions return the rt_status of the failing C-ABI call
// (RenderImage, MaxRayDistanceAny) or a negative time (Render), and never fall back to the CPU.
#pragma once

#include "core.h"
#include "camera.h"
#include "structs.h"
#include "material.h"
#include "lights/light.h"
#include "hittables/hittable.h"

#include "rt_amd.h"

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

struct RenderSettings
{
    int width = 1024, height = 512;   // WIDTH, HEIGHT  (src/main.cpp:20-21)
    int samples = 5;                  // SAMPLES        (src/main.cpp:23)
    int maxDepth = 10;                // MAXDEPTH       (src/main.cpp:27)
    long long seedBase = 0;           // stands in for time(NULL) in rand.h:7
    int device = 0;                   // HIP device ordinal
};

namespace rt_host {

inline void camera35(const Camera& camera, rt_camera35f& out)
{
    out.position[0] = camera.GetPosition().x; out.position[1] = camera.GetPosition().y; out.position[2] = camera.GetPosition().z;
    std::memcpy(out.inv_proj, camera.GetInverseProjectionMatrix().m, sizeof(float) * 16);
    std::memcpy(out.inv_view, camera.GetInverseViewMatrix().m, sizeof(float) * 16);
}

// Material* carries no length in the reference's interfaces; the table must cover every index the scene uses.
inline int materials_needed(const RtFlatScene& flat)
{
    int n = 0;
    for (size_t i = 0; i < flat.h_material.size(); ++i) {
        int m = flat.h_material[i];
        if (flat.h_type[i] != RT_CUBE) m = (uint8_t)m;
        if (m + 1 > n) n = m + 1;
    }
    return n;
}

// One context per call keeps these helpers stateless like the reference's free functions; callers that render many
// frames should hold an rt_ctx themselves.
struct Session {
    rt_ctx* ctx = nullptr;
    int status = RT_OK;
    Session(const Hittable* world, const Light* lights, const Material* materials, int materialsCount, int device)
    {
        RtFlatScene flat;
        if (world) world->flatten(flat);
        if (lights) lights->flatten(flat);
        if (materialsCount < 0) materialsCount = materials_needed(flat);
        status = rt_create(&ctx, device);
        if (status != RT_OK) return;
        rt_scene_soa s;
        s.n_materials = materialsCount; s.materials = reinterpret_cast<const float*>(materials);
        s.n_hittables = (int32_t)flat.h_type.size();
        s.h_type = flat.h_type.data(); s.h_a = flat.h_a.data(); s.h_b = flat.h_b.data();
        s.h_radius = flat.h_radius.data(); s.h_material = flat.h_material.data();
        s.n_lights = (int32_t)flat.l_type.size(); s.l_type = flat.l_type.data(); s.l_v = flat.l_v.data();
        status = rt_upload_scene(ctx, &s);
    }
    ~Session() { rt_destroy(ctx); }
    Session(const Session&) = delete;
    Session& operator=(const Session&) = delete;
};

inline rt_params params(const RenderSettings& st)
{
    rt_params p;
    std::memset(&p, 0, sizeof p);
    p.width = st.width; p.height = st.height; p.samples = st.samples; p.max_depth = st.maxDepth;
    p.seed_base = st.seedBase; p.row0 = 0; p.rows = st.height;
    return p;
}

}  // namespace rt_host

// Fills image[width*height] (and emission[width*height] unless null), row-major, y = 0 at the top, exactly what the
// reference's recompose loop leaves in `image` / `emission` (src/main.cpp:375-395).  materialsCount < 0: inferred.
inline int RenderImage(Camera& camera, Hittable* world, Light* lights, Material* materials, int materialsCount,
                       const RenderSettings& settings, pixel* image, emissionPixel* emission, double* kernelMs = nullptr)
{
    rt_host::Session s(world, lights, materials, materialsCount, settings.device);
    if (s.status != RT_OK) return s.status;
    rt_camera35f cam;
    rt_host::camera35(camera, cam);
    rt_params p = rt_host::params(settings);
    int rc = rt_render_rows(s.ctx, &cam, &p, reinterpret_cast<uint8_t*>(image), reinterpret_cast<float*>(emission), nullptr);
    if (kernelMs) *kernelMs = rt_last_device_ms(s.ctx);
    return rc;
}

// applyGlow(image, emission, width, height) of the reference (src/main.cpp:163-229), on the GPU: image is modified in
// place; unlike the reference's, `emission` is not freed.  Returns RT_OK or an rt_status (RT_ERR_UNSUPPORTED when a
// pyramid level's sizes do not divide the frame's, where the reference reads out of bounds).
inline int ApplyGlow(pixel* image, const emissionPixel* emission, int width, int height, int device = 0, int* levels = nullptr)
{
    rt_ctx* ctx = nullptr;
    int rc = rt_create(&ctx, device);
    if (rc != RT_OK) return rc;
    int32_t n = 0;
    rc = rt_apply_glow(ctx, reinterpret_cast<uint8_t*>(image), reinterpret_cast<const float*>(emission), width, height, &n);
    if (levels) *levels = n;
    rt_destroy(ctx);
    return rc;
}

// One frame over `nranks` processes, one per GPU (this process is `rank`, on GPU settings.device): every rank renders
// its cyclic 16-row bands, RCCL carries them to rank 0, which leaves the frame in image[width*height] -- the reference's
// worker threads, Redis list and recompose loop (src/main.cpp:268-395) across GPUs.  commId: RT_COMM_ID_BYTES made by
// rt_comm_unique_id() on one rank and carried to the others by the host (rt_render --ranks uses pipes).  image: rank 0.
inline int RenderImageRanks(Camera& camera, Hittable* world, Light* lights, Material* materials, int materialsCount,
                            const RenderSettings& settings, int rank, int nranks, const unsigned char* commId, pixel* image,
                            int frames = 1, double* msPerFrame = nullptr)
{
    rt_host::Session s(world, lights, materials, materialsCount, settings.device);
    if (s.status != RT_OK) return s.status;
    void* comm = nullptr;
    if (nranks > 1) {
        int rc = rt_comm_create(&comm, s.ctx, commId, rank, nranks);
        if (rc != RT_OK) return rc;
    }
    rt_camera35f cam;
    rt_host::camera35(camera, cam);
    rt_params p = rt_host::params(settings);
    int rc = RT_OK;
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < frames && rc == RT_OK; ++k)
        rc = rt_render_gather(s.ctx, &cam, &p, comm, rank, nranks, 16, reinterpret_cast<uint8_t*>(image));
    if (msPerFrame) *msPerFrame = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (frames > 0 ? frames : 1);
    rt_comm_destroy(comm);
    return rc;
}

// The producer side of the reference's tile list (src/main.cpp:283-340): every whole tileSize x tileSize tile of the frame
// is handed to `send` twice, with the arguments the reference gives Redis::SendImage (src/redis.h:27) -- first the
// `pixel` payload, then the `emissionPixel` payload -- and `progress` (if set) receives the percentage the reference
// prints (src/main.cpp:337) after every chunk of `rowsPerChunk` rows.  A consumer written against Redis::ReceiveImage
// sees the same records.
//   send(const unsigned char* image, int x, int y, int width, int height, int elementSize)
template <class Send, class Progress>
inline int RenderTiles(Camera& camera, Hittable* world, Light* lights, Material* materials, int materialsCount,
                       const RenderSettings& settings, int tileSize, int rowsPerChunk, Send send, Progress progress)
{
    rt_host::Session s(world, lights, materials, materialsCount, settings.device);
    if (s.status != RT_OK) return s.status;
    rt_camera35f cam;
    rt_host::camera35(camera, cam);
    rt_params p = rt_host::params(settings);
    struct State { Send* send; Progress* progress; int tile, delivered, total, per_chunk; } st;
    st.send = &send; st.progress = &progress; st.tile = tileSize; st.delivered = 0;
    st.total = 2 * (settings.width / tileSize) * (settings.height / tileSize);
    const int chunk_tile_rows = rowsPerChunk <= 0 ? settings.height / tileSize : (rowsPerChunk / tileSize > 0 ? rowsPerChunk / tileSize : 1);
    st.per_chunk = 2 * (settings.width / tileSize) * chunk_tile_rows;
    rt_tile_sink sink = [](void* user, const uint8_t* record, size_t bytes) -> int {
        State& t = *static_cast<State*>(user);
        int x, y;
        std::memcpy(&x, record, sizeof(int)); std::memcpy(&y, record + sizeof(int), sizeof(int));
        const int elementSize = (int)((bytes - 2 * sizeof(int)) / ((size_t)t.tile * t.tile));
        (*t.send)(record + 2 * sizeof(int), x, y, t.tile, t.tile, elementSize);
        ++t.delivered;
        if (t.delivered % t.per_chunk == 0 || t.delivered == t.total) (*t.progress)((float)(t.delivered * 100) / (float)t.total);
        return 0;
    };
    return rt_render_tiles(s.ctx, &cam, &p, tileSize, rowsPerChunk, sink, &st);
}

// A sequence of `frames` frames with up to `inFlight` of them under way on the GPU at once.  One frame on its own is
// bound by the latency of its longest pixels (a pixel's samples are serial through its generator, src/main.cpp:304-318)
// while most of the GPU idles; a sequence hides that (the glass 1080p / 64 spp frame takes 169 ms
// alone and 50 ms per frame with six in flight).  `inFlight` producer threads, each with a context of its own, take
// frame numbers in order and render into a ring of 2 * inFlight host images; the calling thread hands the images out in
// frame order:
//   seedOf(k)           -> long long: frame k's seed base (what time(NULL) is to one run of the reference, rand.h:7);
//                          called from the producer threads
//   deliver(k, image)   called on the calling thread, frames in order; `image` (width*height pixels) is reused after
//                          the call returns
// Every frame's bytes are those of RenderImage() with settings.seedBase = seedOf(k).  Set GPU_MAX_HW_QUEUES=16 in the
// environment before the first HIP call of the process.  Returns the first failing rt_status.
template <class SeedOf, class Deliver>
inline int RenderFrames(Camera& camera, Hittable* world, Light* lights, Material* materials, int materialsCount,
                        const RenderSettings& settings, int frames, int inFlight, SeedOf seedOf, Deliver deliver)
{
    if (frames <= 0) return RT_OK;
    const int F = inFlight < 1 ? 1 : (inFlight > frames ? frames : inFlight);
    const int B = 2 * F;
    std::vector<std::unique_ptr<rt_host::Session>> sessions;
    for (int w = 0; w < F; ++w) {
        sessions.emplace_back(new rt_host::Session(world, lights, materials, materialsCount, settings.device));
        if (sessions.back()->status != RT_OK) return sessions.back()->status;
    }
    rt_camera35f cam;
    rt_host::camera35(camera, cam);
    const size_t npix = (size_t)settings.width * settings.height;
    std::vector<std::vector<pixel>> images(B, std::vector<pixel>(npix));
    enum { FREE, BUSY, READY };
    std::vector<int> state(B, FREE);
    std::mutex m;
    std::condition_variable cv;
    int next = 0, status = RT_OK;

    auto producer = [&](int w) {
        for (;;) {
            int k, b;
            {
                std::unique_lock<std::mutex> lock(m);
                k = next++;                                  // frames are taken in order; frame k lives in image k % B,
                if (k >= frames) return;                     // which is free once frame k - B was delivered
                b = k % B;
                cv.wait(lock, [&] { return state[b] == FREE || status != RT_OK; });
                if (status != RT_OK) return;
                state[b] = BUSY;
            }
            rt_params p = rt_host::params(settings);
            p.seed_base = seedOf(k);
            if (F > 1) p.flags |= RT_FLAG_SEQUENCE;
            const int rc = rt_render_rows(sessions[w]->ctx, &cam, &p, reinterpret_cast<uint8_t*>(images[b].data()), nullptr, nullptr);
            {
                std::lock_guard<std::mutex> lock(m);
                if (rc != RT_OK && status == RT_OK) status = rc;
                state[b] = READY;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> threads;
    for (int w = 0; w < F; ++w) threads.emplace_back(producer, w);
    for (int k = 0; k < frames; ++k) {
        const int b = k % B;
        {
            std::unique_lock<std::mutex> lock(m);
            cv.wait(lock, [&] { return state[b] == READY || status != RT_OK; });
            if (status != RT_OK) break;
        }
        deliver(k, images[b].data());
        {
            std::lock_guard<std::mutex> lock(m);
            state[b] = FREE;
        }
        cv.notify_all();
    }
    cv.notify_all();                            // after a failure: producers waiting for an image leave
    for (auto& t : threads) t.join();
    return status;
}

// The reference's benchmark entry point (src/tests/speedUp.cpp:69-134): renders the frame, discards it, returns the
// elapsed wall-clock milliseconds.  WIDTH / HEIGHT / SAMPLES / MAXDEPTH are compile-time macros of the including
// file there; define them before including this header to get the same spelling, or use RenderTimed().  `threads`
// is accepted for signature compatibility: the GPU grid replaces the ThreadPool.  Returns < 0 on failure.
inline double RenderTimed(Camera& camera, Hittable* world, Light* lights, Material* materials, const RenderSettings& settings)
{
    rt_host::Session s(world, lights, materials, -1, settings.device);
    if (s.status != RT_OK) { std::fprintf(stderr, "Render: %s\n", rt_strerror(s.status)); return -1.0; }
    rt_camera35f cam;
    rt_host::camera35(camera, cam);
    rt_params p = rt_host::params(settings);
    // like the reference, pixels are produced and dropped: render into the library's own staging buffer
    std::vector<uint8_t> sink((size_t)settings.width * settings.height * 3);
    auto t0 = std::chrono::steady_clock::now();
    int rc = rt_render_rows(s.ctx, &cam, &p, sink.data(), nullptr, nullptr);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc != RT_OK) { std::fprintf(stderr, "Render: %s\n", rt_strerror(rc)); return -1.0; }
    return ms;
}

#if defined(WIDTH) && defined(HEIGHT) && defined(SAMPLES) && defined(MAXDEPTH)
inline double Render(Camera& camera, Hittable* world, Light* lights, Material* materials, int /*threads*/)
{
    RenderSettings st;
    st.width = WIDTH; st.height = HEIGHT; st.samples = SAMPLES; st.maxDepth = MAXDEPTH;
    return RenderTimed(camera, world, lights, materials, st);
}
#endif

// src/tests/singlePixel.cpp:65-120: true if any pixel's centre ray reached maxDepth ("Status: Failed").
// flags (optional): width*height bytes, one per pixel.  Returns < 0 (rt_status) on failure.
inline int MaxRayDistanceAny(Camera& camera, Hittable* world, Light* lights, Material* materials, int materialsCount,
                             const RenderSettings& settings, unsigned char* flags = nullptr)
{
    rt_host::Session s(world, lights, materials, materialsCount, settings.device);
    if (s.status != RT_OK) return s.status;
    rt_camera35f cam;
    rt_host::camera35(camera, cam);
    rt_params p = rt_host::params(settings);
    p.samples = 1;
    std::vector<unsigned char> own;
    if (!flags) { own.resize((size_t)settings.width * settings.height); flags = own.data(); }
    int rc = rt_probe_max_depth(s.ctx, &cam, &p, flags);
    if (rc != RT_OK) return rc;
    int any = 0;
    for (size_t i = 0; i < (size_t)settings.width * settings.height; ++i) any |= flags[i];
    return any;
}

// src/main.cpp:413-428
inline void writePPM(const char* path, const pixel* img, int width, int height)
{
    FILE* file = std::fopen(path, "wb");
    if (!file) { std::fprintf(stderr, "Failed to open file\n"); return; }
    std::fprintf(file, "P6\n%d %d\n255\n", width, height);
    std::fwrite(img, sizeof(pixel), (size_t)width * height, file);
    std::fclose(file);
}
