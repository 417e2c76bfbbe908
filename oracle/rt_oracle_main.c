/* TEST INFRASTRUCTURE -- command-line front end of the CPU restatement; same flags and JSON line as
 * ref_harness.cpp so the two can be diffed directly (tests/test_oracle_vs_ref.py). */
#include "rt_oracle.h"
#include "scene_text.h"

#include <time.h>

static double now_ms(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char** argv)
{
    const char* scene_path = NULL; const char* out_ppm = NULL; const char* out_em = NULL;
    const char* out_cam = NULL; const char* out_probe = NULL;
    int W = 256, H = 128, samples = 5, maxdepth = 10, threads = 0, probe = 0, reps = 1, row0 = 0, rows = -1, row_step = 1;
    long seed_base = 0;
    for (int i = 1; i < argc; ++i) {
#define IS(k) (!strcmp(argv[i], k) && i + 1 < argc)
        if      (IS("--scene"))     scene_path = argv[++i];
        else if (IS("--out"))       out_ppm = argv[++i];
        else if (IS("--emission"))  out_em = argv[++i];
        else if (IS("--camera"))    out_cam = argv[++i];
        else if (IS("--probe-out")) out_probe = argv[++i];
        else if (IS("--w"))         W = atoi(argv[++i]);
        else if (IS("--h"))         H = atoi(argv[++i]);
        else if (IS("--samples"))   samples = atoi(argv[++i]);
        else if (IS("--depth"))     maxdepth = atoi(argv[++i]);
        else if (IS("--seed-base")) seed_base = atol(argv[++i]);
        else if (IS("--threads"))   threads = atoi(argv[++i]);
        else if (IS("--reps"))      reps = atoi(argv[++i]);
        else if (IS("--row0"))      row0 = atoi(argv[++i]);
        else if (IS("--rows"))      rows = atoi(argv[++i]);
        else if (IS("--row-step"))  row_step = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--probe")) probe = 1;
        else { fprintf(stderr, "unknown arg %s\n", argv[i]); return 2; }
    }
    if (!scene_path) { fprintf(stderr, "usage: rt_oracle --scene S [--w --h --samples --depth --seed-base --out --emission --threads]\n"); return 2; }
    if (rows < 0) rows = (H - row0 + row_step - 1) / row_step;

    st_scene sc;
    if (!st_load(scene_path, &sc)) return 2;
    ro_scene scene = { (const ro_material*)sc.materials, sc.n_materials,
                       (const ro_hittable*)sc.hittables, sc.n_hittables,
                       (const ro_light*)sc.lights, sc.n_lights };
    ro_camera cam;
    ro_camera_init(&cam, sc.fov, (float)W, (float)H, sc.znear, sc.zfar);
    if (out_cam) { FILE* f = fopen(out_cam, "wb"); fwrite(&cam, 4, 35, f); fclose(f); }

    if (probe) {
        unsigned char* flags = (unsigned char*)malloc((size_t)W * H);
        if (ro_probe_max_depth(&scene, &cam, W, H, maxdepth, seed_base, flags)) return 2;
        int any = 0; for (size_t i = 0; i < (size_t)W * H; ++i) any |= flags[i];
        if (out_probe) { FILE* f = fopen(out_probe, "wb"); fwrite(flags, 1, (size_t)W * H, f); fclose(f); }
        printf("{\"mode\": \"probe\", \"w\": %d, \"h\": %d, \"depth\": %d, \"seed_base\": %ld, \"any_max_depth\": %d}\n",
               W, H, maxdepth, seed_base, any);
        return 0;
    }

    unsigned char* rgb = (unsigned char*)calloc((size_t)W * H, 3);
    float* em = (float*)calloc((size_t)W * H, 16);
    ro_counters cnt;
    double t0 = now_ms();
    for (int r = 0; r < reps; ++r)
        if (ro_render_strided(&scene, &cam, W, H, samples, maxdepth, seed_base, row0, rows, row_step, threads,
                              row_step == 1 ? rgb + (size_t)row0 * W * 3 : rgb, row_step == 1 ? em + (size_t)row0 * W * 4 : em, &cnt)) { fprintf(stderr, "ro_render failed\n"); return 2; }
    double ms = (now_ms() - t0) / reps;

    if (out_ppm) {
        FILE* f = fopen(out_ppm, "wb");
        fprintf(f, "P6\n%d %d\n255\n", W, H);          /* main.cpp:423 */
        fwrite(rgb, 3, (size_t)W * H, f);
        fclose(f);
    }
    if (out_em) { FILE* f = fopen(out_em, "wb"); fwrite(em, 16, (size_t)W * H, f); fclose(f); }
    printf("{\"mode\": \"render\", \"w\": %d, \"h\": %d, \"samples\": %d, \"depth\": %d, \"seed_base\": %ld, \"row0\": %d, \"rows\": %d, "
           "\"threads\": %d, \"ms\": %.3f, \"primary\": %llu, \"closest\": %llu, \"shadow\": %llu, \"tests\": %llu, \"rand_draws\": %llu}\n",
           W, H, samples, maxdepth, seed_base, row0, rows, threads, ms,
           cnt.primary, cnt.closest, cnt.shadow, cnt.tests, cnt.draws);
    st_free(&sc);
    free(rgb); free(em);
    return 0;
}
