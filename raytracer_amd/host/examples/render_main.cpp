// Drop-in demonstration: a driver written like the reference's (src/main.cpp:231-411 without Redis / Postgres / glow,
// src/main.cu:458-572 without CUDA) on top of the API-compatible host headers.  The scene is either the reference's
// default rows (src/postgres.h:207-210,224-227,191 == src/main.cu:74,83-86,522-525), built with the same `new Sphere`
// / `new HittablesList` / `new DirectionalLight` / `new LightsList` calls, or an rtscene file.
//
//   rt_render [--scene file] [--w 1024 --h 512 --samples 5 --depth 10 --seed-base 0] [--out output.ppm]
//             [--emission out.f32] [--glow] [--probe] [--speedup] [--tiles 16 [--chunk-rows 128]]
//             [--frames 8 [--in-flight 4]]      frame k is seeded seed-base + 1000 k and written to <out>.<k>.ppm
//             [--ranks N]                       one process per GPU (ranks 1..N-1 forked before anything touches a GPU), cyclic
//                                               16-row bands, RCCL gather to rank 0 (GPU r = --device + r)
#include "../renderer.h"

#include "../lights/directionalLight.h"
#include "../lights/lightsList.h"
#include "../hittables/hittablesList.h"
#include "../hittables/sphere.h"
#include "../scene_io.h"

#include <cstdlib>
#include <cstring>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>

int main(int argc, char** argv)
{
    RenderSettings st;
    const char* scenePath = nullptr; const char* out = "output.ppm"; const char* outEm = nullptr;
    bool probe = false, speedup = false, glow = false;
    int tiles = 0, chunkRows = 0, frames = 0, inFlight = 4, ranks = 0;
    for (int i = 1; i < argc; ++i) {
        auto is = [&](const char* k) { return !std::strcmp(argv[i], k) && i + 1 < argc; };
        if      (is("--scene"))     scenePath = argv[++i];
        else if (is("--w"))         st.width = std::atoi(argv[++i]);
        else if (is("--h"))         st.height = std::atoi(argv[++i]);
        else if (is("--samples"))   st.samples = std::atoi(argv[++i]);
        else if (is("--depth"))     st.maxDepth = std::atoi(argv[++i]);
        else if (is("--seed-base")) st.seedBase = std::atoll(argv[++i]);
        else if (is("--device"))    st.device = std::atoi(argv[++i]);
        else if (is("--out"))       out = argv[++i];
        else if (is("--emission"))  outEm = argv[++i];
        else if (is("--tiles"))     tiles = std::atoi(argv[++i]);
        else if (is("--chunk-rows")) chunkRows = std::atoi(argv[++i]);
        else if (is("--frames"))    frames = std::atoi(argv[++i]);
        else if (is("--in-flight")) inFlight = std::atoi(argv[++i]);
        else if (is("--ranks"))     ranks = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--probe"))   probe = true;
        else if (!std::strcmp(argv[i], "--speedup")) speedup = true;
        else if (!std::strcmp(argv[i], "--glow"))    glow = true;
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }

    // --ranks N: ranks 1 .. N-1 are forked here, before anything touches a GPU.  Then a handshake over pipes, so that no
    // rank enters RCCL unless every rank can: each rank opens (and closes) a context on its GPU and reports to rank 0;
    // rank 0 makes the communicator's id and answers every rank with go / no-go and the id.  A rank without a GPU, or a
    // failing rt_comm_unique_id, ends the job with exit code 1 on every rank instead of leaving the others waiting in
    // ncclCommInitRank.
    int rank = 0;
    unsigned char commId[RT_COMM_ID_BYTES] = { 0 };
    std::vector<pid_t> kids;
    if (ranks > 1) {
        std::vector<int> down_rd((size_t)ranks, -1), down_wr((size_t)ranks, -1), up_rd((size_t)ranks, -1), up_wr((size_t)ranks, -1);
        for (int r = 1; r < ranks; ++r) {
            int d[2], u[2];
            if (pipe(d) != 0 || pipe(u) != 0) { std::perror("pipe"); return 2; }
            down_rd[(size_t)r] = d[0]; down_wr[(size_t)r] = d[1]; up_rd[(size_t)r] = u[0]; up_wr[(size_t)r] = u[1];
        }
        for (int r = 1; r < ranks && rank == 0; ++r) {
            pid_t pid = fork();
            if (pid < 0) { std::perror("fork"); return 2; }
            if (pid == 0) { rank = r; kids.clear(); }     // (a child inherits the list of its elder siblings: not its to wait for)
            else kids.push_back(pid);
        }
        auto reap = [&kids]() {
            bool ok = true;
            for (pid_t pid : kids) { int wst = 0; if (waitpid(pid, &wst, 0) < 0 || !WIFEXITED(wst) || WEXITSTATUS(wst) != 0) ok = false; }
            kids.clear();
            return ok;
        };
        st.device += rank;
        rt_ctx* probe_ctx = nullptr;
        const int have = rt_create(&probe_ctx, st.device);
        if (have == RT_OK) rt_destroy(probe_ctx);
        else std::fprintf(stderr, "rank %d: no context on GPU %d: %s\n", rank, st.device, rt_strerror(have));
        unsigned char msg[1 + RT_COMM_ID_BYTES] = { 0 };
        if (rank == 0) {
            bool go = have == RT_OK;
            for (int r = 1; r < ranks; ++r) {
                close(down_rd[(size_t)r]); close(up_wr[(size_t)r]);
                unsigned char ok = 0;
                if (read(up_rd[(size_t)r], &ok, 1) != 1 || ok != 1) go = false;         // (a child that died reads as 0 bytes)
                close(up_rd[(size_t)r]);
            }
            if (go) {
                const int rc = rt_comm_unique_id(commId);
                if (rc != RT_OK) { std::fprintf(stderr, "rt_comm_unique_id failed: %s\n", rt_strerror(rc)); go = false; }
            }
            msg[0] = go ? 1 : 0;
            std::memcpy(msg + 1, commId, sizeof commId);
            for (int r = 1; r < ranks; ++r) {
                if (write(down_wr[(size_t)r], msg, sizeof msg) != (ssize_t)sizeof msg) std::perror("write");
                close(down_wr[(size_t)r]);
            }
            if (!go) { (void)reap(); return 1; }
        } else {
            for (int r = 1; r < ranks; ++r) {
                close(down_wr[(size_t)r]); close(up_rd[(size_t)r]);
                if (r != rank) { close(down_rd[(size_t)r]); close(up_wr[(size_t)r]); }
            }
            const unsigned char ok = have == RT_OK ? 1 : 0;
            if (write(up_wr[(size_t)rank], &ok, 1) != 1) std::perror("write");
            close(up_wr[(size_t)rank]);
            const bool got = read(down_rd[(size_t)rank], msg, sizeof msg) == (ssize_t)sizeof msg;
            close(down_rd[(size_t)rank]);
            if (!got || msg[0] != 1) { if (!got) std::fprintf(stderr, "rank %d: no answer from rank 0\n", rank); return 1; }
            std::memcpy(commId, msg + 1, sizeof commId);
        }
    }

    LoadedScene loaded;
    Material* materials = nullptr; int materialsCount = 0;
    Hittable* world = nullptr; Light* lights = nullptr;
    float fov = 60.0f, znear = 0.01f, zfar = 1000.0f;
    Material defaults[4];
    std::vector<Hittable*> l_world; std::vector<Light*> l_light;
    if (scenePath) {
        if (!LoadScene(scenePath, loaded)) return 2;
        materials = loaded.materials.data(); materialsCount = (int)loaded.materials.size();
        world = loaded.world; lights = loaded.lights;
        fov = loaded.fov; znear = loaded.znear; zfar = loaded.zfar;
    } else {
        // -- Init Lights
        l_light.push_back(new DirectionalLight({ -0.25f, -0.75f, 0.45f }));
        lights = new LightsList(l_light.data(), (uint32_t)l_light.size());
        // -- Init Materials
        defaults[0] = Material{ { 0.8f, 0.8f, 0.0f }, 0.0f,  0.0f,  0.0f,  { 0.0f, 0.0f, 0.0f }, 0.0f };
        defaults[1] = Material{ { 0.8f, 0.2f, 0.1f }, 0.08f, 0.02f, 0.0f,  { 1.0f, 0.0f, 0.0f }, 4.5f };
        defaults[2] = Material{ { 0.8f, 0.8f, 0.8f }, 0.9f,  0.75f, 0.0f,  { 0.0f, 0.0f, 0.0f }, 0.0f };
        defaults[3] = Material{ { 0.0f, 0.0f, 0.0f }, 0.0f,  0.0f,  1.85f, { 0.0f, 0.0f, 0.0f }, 0.0f };
        materials = defaults; materialsCount = 4;
        // -- Init world
        l_world.push_back(new Sphere({  0.0f, -1000.0f, -4.0f }, 1000.0f, 0));
        l_world.push_back(new Sphere({  0.0f,  1.0f,    -4.0f }, 1.0f,    1));
        l_world.push_back(new Sphere({ -3.0f,  1.0f,    -4.0f }, 1.0f,    2));
        l_world.push_back(new Sphere({  3.0f,  1.0f,    -4.0f }, 1.0f,    3));
        world = new HittablesList(l_world.data(), (uint32_t)l_world.size());
    }

    Camera camera(fov, (float)st.width, (float)st.height, znear, zfar);
    int rc = 0;
    if (probe) {
        int any = MaxRayDistanceAny(camera, world, lights, materials, materialsCount, st);
        if (any < 0) { std::fprintf(stderr, "probe failed: %s\n", rt_strerror(any)); rc = 1; }
        else std::printf("Status: %s!\n", any ? "Failed" : "Success");                     // src/tests/singlePixel.cpp:120
    } else if (speedup) {
        double ms = RenderTimed(camera, world, lights, materials, st);
        if (ms < 0) rc = 1; else std::printf("Ended in: %lf\n", ms);                       // src/tests/speedUp.cpp:48
    } else if (ranks > 0) {
        // one frame (or --frames of the same frame, timed) over --ranks GPUs
        std::vector<pixel> image(rank == 0 ? (size_t)st.width * st.height : 0);
        double ms = 0.0;
        int status = RenderImageRanks(camera, world, lights, materials, materialsCount, st, rank, ranks, commId,
                                      rank == 0 ? image.data() : nullptr, frames > 0 ? frames : 1, &ms);
        if (status != RT_OK) { std::fprintf(stderr, "rank %d: RenderImageRanks failed: %s\n", rank, rt_strerror(status)); rc = 1; }
        else if (rank == 0) {
            std::printf("Ended in: %lf ms per frame (%d ranks)\n", ms, ranks);
            writePPM(out, image.data(), st.width, st.height);
        }
        for (pid_t pid : kids) { int wst = 0; if (waitpid(pid, &wst, 0) < 0 || !WIFEXITED(wst) || WEXITSTATUS(wst) != 0) rc = 1; }
    } else if (frames > 0) {
        // a sequence of frames, several in flight; delivered in order
        auto t0 = std::chrono::steady_clock::now();
        int delivered = 0;
        int status = RenderFrames(camera, world, lights, materials, materialsCount, st, frames, inFlight,
                                  [&](int k) { return st.seedBase + 1000LL * k; },
                                  [&](int k, const pixel* image) {
                                      if (k != delivered++) std::fprintf(stderr, "frame %d out of order\n", k);
                                      char path[1024];
                                      std::snprintf(path, sizeof path, "%s.%d.ppm", out, k);
                                      writePPM(path, image, st.width, st.height);
                                  });
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (status != RT_OK) { std::fprintf(stderr, "RenderFrames failed: %s\n", rt_strerror(status)); rc = 1; }
        else std::printf("Ended in: %lf ms, %lf ms per frame (%d frames, %d in flight)\n", ms, ms / frames, frames, inFlight);
    } else if (tiles > 0) {
        // the reference's producer / consumer pair around its Redis list (src/main.cpp:283-395), without the list: tiles
        // arrive as SendImage arguments and are recomposed the way the RPOP loop does it
        std::vector<pixel> image((size_t)st.width * st.height);
        std::vector<emissionPixel> emission((size_t)st.width * st.height);
        auto send = [&](const unsigned char* data, int sX, int sY, int w, int h, int elementSize) {
            for (int nY = 0; nY < h; ++nY) {
                const size_t dst = (size_t)sX * w + ((size_t)sY * h + nY) * st.width;
                if (elementSize == (int)sizeof(pixel)) std::memcpy(&image[dst], data + (size_t)nY * w * elementSize, (size_t)w * elementSize);
                else                                   std::memcpy(&emission[dst], data + (size_t)nY * w * elementSize, (size_t)w * elementSize);
            }
        };
        auto progress = [](float percent) { std::printf("Progress: %f%%\n", percent); };           // src/main.cpp:337
        auto t0 = std::chrono::steady_clock::now();
        int status = RenderTiles(camera, world, lights, materials, materialsCount, st, tiles, chunkRows, send, progress);
        std::printf("Ended in: %lf ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (status != RT_OK) { std::fprintf(stderr, "RenderTiles failed: %s\n", rt_strerror(status)); rc = 1; }
        else {
            writePPM(out, image.data(), st.width, st.height);
            if (outEm) { FILE* f = std::fopen(outEm, "wb"); std::fwrite(emission.data(), sizeof(emissionPixel), emission.size(), f); std::fclose(f); }
        }
    } else {
        std::vector<pixel> image((size_t)st.width * st.height);
        std::vector<emissionPixel> emission((outEm || glow) ? (size_t)st.width * st.height : 0);
        double ms = 0.0;
        std::printf("Running Raytracing on MI355X...\n");
        int status = RenderImage(camera, world, lights, materials, materialsCount, st, image.data(),
                                 (outEm || glow) ? emission.data() : nullptr, &ms);
        if (status == RT_OK && glow) {                                                   // src/main.cpp:400-404
            std::printf("Running glow...\n");
            int levels = 0;
            status = ApplyGlow(image.data(), emission.data(), st.width, st.height, st.device, &levels);
            if (status == RT_OK) std::printf("Glow: %d levels\n", levels);
        }
        if (status != RT_OK) { std::fprintf(stderr, "RenderImage failed: %s\n", rt_strerror(status)); rc = 1; }
        else {
            std::printf("Ended in: %lf ms\n", ms);
            writePPM(out, image.data(), st.width, st.height);
            if (outEm) { FILE* f = std::fopen(outEm, "wb"); std::fwrite(emission.data(), sizeof(emissionPixel), emission.size(), f); std::fclose(f); }
        }
    }
    if (!scenePath) { delete lights; delete world; }
    return rc;
}
