/* TEST INFRASTRUCTURE -- CPU restatement ("port" oracle) of the reference's TraceRay hot path in plain C.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.  The product
 * (raytracer_amd/csrc, include/rt_amd.h) never links, imports or calls it.
 *
 * Pinned against oracle/_ref/ref_oracle (the unmodified reference headers compiled in the authoring
 * container) and against the golden fixtures in tests/golden/ that binary produced: PPMs must be
 * byte-identical, emission buffers bit-identical, counters equal.  The reference's own tests hold no
 * numeric fixtures for this path, so those goldens are the pin.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { RO_SPHERE = 0, RO_PLANE = 1, RO_CUBE = 2 };
enum { RO_DIRLIGHT = 0, RO_POINTLIGHT = 1 };

/* material.h:5-11 (40 bytes) */
typedef struct { float color[3], roughness, reflection, refraction, emission[3], glow; } ro_material;
/* sphere: a = centre, radius; plane: a = centre, b = normal (normalised on load, plane.h:11);
 * cube: a = min corner, b = size (cube.h:10-11). */
typedef struct { int type; float a[3]; float b[3]; float radius; int material; } ro_hittable;
/* dirlight: v = direction as given to the ctor (directionalLight.h:8-9 stores normalize(-v));
 * pointlight: v = position (repo-defined, see ref_harness.cpp PointLight). */
typedef struct { int type; float v[3]; } ro_light;

typedef struct {
    const ro_material* materials; int n_materials;
    const ro_hittable* hittables; int n_hittables;
    const ro_light*    lights;    int n_lights;
} ro_scene;

/* camera.h:11-22: position, inverse projection and inverse view, column-major like glm::mat4 */
typedef struct { float pos[3]; float inv_proj[16]; float inv_view[16]; } ro_camera;

typedef struct { unsigned long long primary, closest, shadow, tests, draws; } ro_counters;

void ro_camera_init(ro_camera* cam, float fov_deg, float width, float height, float znear, float zfar);

/* Renders rows [row0, row0+rows) of a W x H frame.  rgb: rows*W*3 bytes; emission: rows*W*4 floats or NULL.
 * threads <= 1: single thread.  Every pixel owns its RNG (seeded seed_base + x + y*W like rand.h:7 with
 * time() pinned), so the image does not depend on the thread count.  Returns 0, or -1 on bad arguments. */
int ro_render(const ro_scene* scene, const ro_camera* cam, int W, int H, int samples, int maxdepth,
              long seed_base, int row0, int rows, int threads,
              unsigned char* rgb, float* emission, ro_counters* counters);

/* Same, for the sampled rows y = row0 + i*row_step, i < rows (packed output); used to time a bounded sample. */
int ro_render_strided(const ro_scene* scene, const ro_camera* cam, int W, int H, int samples, int maxdepth,
                      long seed_base, int row0, int rows, int row_step, int threads,
                      unsigned char* rgb, float* emission, ro_counters* counters);

/* tests/singlePixel.cpp:82 -> renderer.h:204-219: flags[x + y*W] = (maxDepth >= maxdepth) for one centre ray. */
int ro_probe_max_depth(const ro_scene* scene, const ro_camera* cam, int W, int H, int maxdepth,
                       long seed_base, unsigned char* flags);

/* Glow post-process (src/main.cpp:32-229, rt_glow.c: PARITY UNPINNED, see its header).  image: width*height `pixel`
 * (3 bytes), modified in place; emission: width*height emissionPixel (4 floats), read only.  Returns the number of
 * pyramid levels applied, -1 if a level's sizes do not divide the frame's (the reference reads out of bounds there). */
int ro_apply_glow(unsigned char* image, const float* emission, int width, int height);
/* the blur's normalised weights (src/main.cpp:101-123), kernel[(x + size) * (2 size + 1) + (y + size)] */
void ro_glow_kernel(float sigma, int size, float* kernel);

/* glibc srand()/rand() TYPE_3 restatement (rand.h:7,9). state: 31 words + 2 indices. */
typedef struct { unsigned int r[31]; int f, b; } ro_rng;
void ro_srand(ro_rng* g, unsigned int seed);
int  ro_rand(ro_rng* g);

/* pixel::Set (structs.h:12-17) for one channel */
unsigned char ro_quantise(float c);

#ifdef __cplusplus
}
#endif
#endif
