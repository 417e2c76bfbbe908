/* rt_amd.h -- C-ABI of the MI355X (gfx950) rendering backend for DevDuoItaly/Raytracer's TraceRay hot path.
 *
 * The reference has no plugin/FFI interface; its boundary is a source-level convention: whoever owns the pixel
 * loop calls AntiAliasing() per sample and fills pixel[] / emissionPixel[] (reference src/main.cpp:295-320,
 * src/tests/speedUp.cpp:91-113, the function-shaped instance being
 *   double Render(Camera&, Hittable* world, Light* lights, Material* materials, int threads)
 * at src/tests/speedUp.cpp:26,69).  This header is what a binding for that loop nest links against.  Each entry
 * point names the reference code it replaces.  Plain C, plain pointers and sizes; no exceptions cross it.
 *
 * Conventions: every function returns RT_OK (0) or a negative rt_status; the caller owns every buffer it passes
 * (the library copies what it keeps); a context is bound to one GPU and is not thread-safe; calls are synchronous.
 * The library works on HIP streams of its own: device buffers passed in must have no work pending on other streams.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RT_OK               =  0,
    RT_ERR_INVALID_ARG  = -1,  /* null pointer, non-positive size, rows outside the frame ...               */
    RT_ERR_NO_DEVICE    = -2,  /* no such GPU / not a gfx950 device                                          */
    RT_ERR_HIP          = -3,  /* a HIP runtime call failed; rt_last_hip_error() has the code                */
    RT_ERR_SCENE        = -4,  /* material index out of range, unknown hittable / light type                 */
    RT_ERR_UNSUPPORTED  = -5,  /* material.reflection > 1 (unbounded recursion in the reference too),
                                  max_depth above RT_MAX_DEPTH, scene larger than the LDS staging budget     */
    RT_ERR_NO_SCENE     = -6,  /* rt_render_rows before rt_upload_scene                                      */
    RT_ERR_DEVICE_FAULT = -7,  /* the kernel reported an internal bound violation (bounce stack overflow)    */
    RT_ERR_TOO_LARGE    = -8,  /* rt_render_rows_static: rows * width above RT_STATIC_MAX_PIXELS              */
    RT_ERR_RCCL         = -9   /* librccl could not be loaded, or an RCCL call failed; rt_last_rccl_error()   */
} rt_status;

enum { RT_SPHERE = 0, RT_PLANE = 1, RT_CUBE = 2 };        /* reference src/hittables/{sphere,plane,cube}.h   */
enum { RT_DIRLIGHT = 0, RT_POINTLIGHT = 1 };              /* src/lights/directionalLight.h; point: repo-defined */

#define RT_MAX_DEPTH     64      /* largest accepted rt_params.max_depth (MAXDEPTH, src/main.cpp:27)          */
#define RT_MAX_HITTABLES 4096    /* LDS staging budget: 32 B per hittable                                     */
#define RT_MAX_LIGHTS    64
#define RT_MAX_MATERIALS 256     /* Sphere/Plane take a uint8_t material index (sphere.h:10, plane.h:10)      */

typedef struct rt_ctx rt_ctx;

/* Flattened scene, host memory, structure of arrays.  Replaces the device-side `new Sphere/new HittablesList/
 * new DirectionalLight/new LightsList` of src/main.cu:69-90 and the Material upload of src/main.cu:517-530.
 * Hittables stay in list order (closest-hit ties go to the earlier one, src/hittables/hittablesList.h:27).
 *   materials : n_materials x 10 floats, exactly the layout of `struct Material` (src/material.h:5-11):
 *               color.rgb, roughness, reflection, refraction, emissionColor.rgb, glowStrength
 *   h_type    : RT_SPHERE / RT_PLANE / RT_CUBE
 *   h_a       : n x 3  sphere centre | plane centre | cube min corner
 *   h_b       : n x 3  (unused)      | plane normal (normalised by the library like plane.h:11) | cube size
 *   h_radius  : n      sphere radius
 *   h_material: n      index into materials (spheres and planes are truncated to uint8_t like the reference ctors)
 *   l_type    : RT_DIRLIGHT / RT_POINTLIGHT
 *   l_v       : n x 3  direction as passed to DirectionalLight's ctor (the library stores normalize(-v),
 *               directionalLight.h:9) | point-light position */
typedef struct {
    int32_t        n_materials;
    const float*   materials;
    int32_t        n_hittables;
    const int32_t* h_type;
    const float*   h_a;
    const float*   h_b;
    const float*   h_radius;
    const int32_t* h_material;
    int32_t        n_lights;
    const int32_t* l_type;
    const float*   l_v;
} rt_scene_soa;

/* What the kernels consume of `class Camera` (src/camera.h:11-47): m_Position and the two inverse matrices,
 * column-major like glm::mat4. */
typedef struct {
    float position[3];
    float inv_proj[16];
    float inv_view[16];
} rt_camera35f;

/* The reference's compile-time knobs (src/main.cpp:20-27) as run-time parameters, plus the row selection.
 * Rows rendered by one call: local row r (0 <= r < rows) is frame row
 *     y = row0 + (r / band_rows) * band_stride + (r % band_rows)        when band_rows > 0
 *     y = row0 + r                                                      when band_rows == 0
 * so a rank of an N-GPU job renders every N-th band (cyclic bands, DESIGN.md "multi-GPU").  Outputs are packed:
 * local row r starts at rgb_out + r*width*3 and emission_out + r*width*4. */
typedef struct {
    int32_t width, height;       /* WIDTH, HEIGHT                                                            */
    int32_t samples;             /* SAMPLES; each sample traces 4 primary rays (src/renderer.h:167-195)       */
    int32_t max_depth;           /* MAXDEPTH, passed to AntiAliasing as maxTotalDepth                        */
    int64_t seed_base;           /* stands in for time(NULL) in rand.h:7: pixel seed = seed_base + x + y*W   */
    int32_t row0, rows;
    int32_t band_rows, band_stride;
    int32_t flags;               /* RT_FLAG_* */
} rt_params;

#define RT_FLAG_COUNT_LIKE_REFERENCE  1  /* tally, issuing every query the reference would issue (one general kernel
                                           over every pixel; slower)                                              */
#define RT_FLAG_COUNT_EXECUTED        2  /* tally the general kernel with its shadow-query pruning (queries that cannot
                                           change the pixel are skipped)                      */
#define RT_FLAG_SINGLE_KERNEL         4  /* A/B: one general kernel over every pixel instead of the direct + listed +
                                           heavy kernels (DESIGN.md "execution model")                            */
#define RT_FLAG_FORCE_SAMPLE_PARALLEL 8  /* test hook: every bounce pixel takes the sample-parallel route that only
                                           long pixels take otherwise (DESIGN.md "execution model", 3)            */
#define RT_FLAG_SEQUENCE              16 /* hint: the caller keeps several frames in flight on this GPU, one context
                                           each.  Same bytes; the library then favours how many
                                           workgroups are resident over the latency of one frame (the listed kernels
                                           keep their generators in scratch rather than LDS; one launch instead of two
                                           side by side; the calling thread naps between polls while it waits)      */

/* (32 and 64 belonged to two alternative routes for bounce pixels -- a lane per ray, a wave per unsteady pixel -- that were
 * measured in round 2, had no default user and were removed in round 3: DESIGN.md section 5, 3b / 3c) */
#define RT_FLAG_RELAXED               128 /* opt-in: pixels whose primary rays cannot bounce (the direct kernel's, which draw
                                           no random numbers) are rendered with relaxed float arithmetic -- fused
                                           multiply-adds, the hardware's 1-ulp sqrt / reciprocal -- instead of the
                                           reference's operation for operation.  NOT bit-identical: within the tolerance
                                           the port is specified to (>= 99.9 % of the 8-bit pixels within 1 LSB of the
                                           reference's, >= 50 dB); tests/test_gpu_relaxed.py has the measured margins.
                                           Bounce pixels and every other flag keep the exact arithmetic.  Ignored by the
                                           tallying builds and RT_FLAG_SINGLE_KERNEL (they run the exact general kernel)  */

#define RT_FLAG_FORCE_WAVE_GROUPS     256 /* test hook, with RT_FLAG_SEQUENCE: the listed kernel runs with one wave per
                                           workgroup whatever the context measured on its last frame (DESIGN.md section 9,
                                           "what frames in flight wait for"; without the hook a context chooses that
                                           launch shape for frames whose listed kernel took 20 ms or more).  Same bytes  */

/* Query tallies with the reference's meaning: primary = W*rows*samples*4; closest = TraceRay closest-hit queries
 * (src/renderer.h:40); shadow = hasIntersect / point-light occlusion queries (directionalLight.h:13); tests =
 * primitive intersect() calls; rand_draws = RANDOM_UNIT calls; shaded = closest-hit queries that hit (one
 * GetLightIntensity each, src/renderer.h:58).  Filled only when one of the RT_FLAG_COUNT_* flags is set. */
typedef struct {
    uint64_t primary, closest, shadow, tests, rand_draws, shaded;
} rt_counters;

/* Creates a context on GPU `device_ordinal` (HIP ordinal).  Replaces the cudaMalloc/cudaDeviceSetLimit set-up of
 * src/main.cu:469-483. */
int rt_create(rt_ctx** out, int device_ordinal);
void rt_destroy(rt_ctx* ctx);

/* Validates and uploads a scene (see rt_scene_soa). */
int rt_upload_scene(rt_ctx* ctx, const rt_scene_soa* scene);

/* Camera(fov, width, height, near, far) of src/camera.h:11-22 (eye (-3,6,10) -> (0,3,0), GLM perspectiveFov /
 * lookAt / inverse in GLM's operation order), host arithmetic only. */
int rt_camera_init(rt_camera35f* cam, float fov_deg, float width, float height, float z_near, float z_far);

/* The hot path: replaces the tile lambda of src/main.cpp:289-330 (== the `kernel` of src/main.cu:38-67) for the
 * selected rows.  rgb_out: rows*width*3 bytes of `pixel` (src/structs.h:8-25), emission_out: rows*width*4 floats
 * of `emissionPixel` (src/structs.h:28-44) or NULL, counters: NULL unless a RT_FLAG_COUNT_* flag is set.
 * rgb_out / emission_out may be host or device pointers (device pointers must belong to the context's GPU). */
int rt_render_rows(rt_ctx* ctx, const rt_camera35f* cam, const rt_params* params,
                   uint8_t* rgb_out, float* emission_out, rt_counters* counters);

/* rt_render_rows for callers that cannot allocate.  It has no counterpart in the reference: it is for callers that
 * may not call hipMalloc where they render -- tight serving loops, harnesses that forbid allocation, code that sets up
 * a HIP graph capture elsewhere in the process and must not disturb the allocator meanwhile.  The call itself is
 * synchronous (it waits on the context's stream and reads a few words back, like rt_render_rows) and therefore cannot
 * itself be captured into a graph.  Same kernels, same bytes as rt_render_rows, but the call allocates nothing (no
 * hipMalloc / hipFree, no stream or event creation): the per-frame device state and the pixel classes (a class per pixel and the list of
 * bounce pixels, 6 bytes per pixel) live in a static device array of RT_STATIC_MAX_PIXELS * 6 + 4096 bytes that every
 * GPU the library runs on holds until the process ends.
 *   - rgb_dev / emission_dev must be DEVICE pointers on the context's GPU (host outputs would need a staging buffer);
 *     emission_dev may be NULL; RT_FLAG_COUNT_* flags are rejected (RT_ERR_INVALID_ARG);
 *   - params->rows * params->width above RT_STATIC_MAX_PIXELS is rejected with RT_ERR_TOO_LARGE: render the frame in
 *     several calls over row ranges (rows are independent, rt_params.row0 / rows);
 *   - every call on a GPU shares that GPU's array, whatever context it goes through: calls that overlap in time on one
 *     GPU race.  One call at a time per GPU (calls through rt_render_rows do not touch the array and may overlap). */
#define RT_STATIC_MAX_PIXELS (1 << 20)
int rt_render_rows_static(rt_ctx* ctx, const rt_camera35f* cam, const rt_params* params,
                          uint8_t* rgb_dev, float* emission_dev);

/* Depth probe of src/tests/singlePixel.cpp:65-90 (MaxRayDistance, src/renderer.h:204-219): one centre ray per
 * pixel; flags_out[x + y*width] = (maxDepth >= max_depth).  flags_out: host or device, width*height bytes. */
int rt_probe_max_depth(rt_ctx* ctx, const rt_camera35f* cam, const rt_params* params, uint8_t* flags_out);

/* Reassembles a frame from per-rank packed cyclic bands after the gather (the role of the Redis RPOP recompose
 * loop, src/main.cpp:375-395).  gathered: nranks blocks of rows_per_rank*width*bytes_per_pixel (rank-major, each
 * block packed as rt_render_rows writes it for row0 = rank*band_rows, band_stride = nranks*band_rows); frame_out:
 * height*width*bytes_per_pixel.  Both device pointers on the context's GPU. */
int rt_deinterleave_bands(rt_ctx* ctx, const void* gathered, void* frame_out, int32_t width, int32_t height,
                          int32_t bytes_per_pixel, int32_t band_rows, int32_t nranks, int32_t rows_per_rank);

/* Multi-GPU: the gather of the ranks' band blocks to rank 0 over RCCL (xGMI), one process per GPU.  Replaces the
 * consumer side of the reference's tile transport: workers LPUSH their tiles, the main thread RPOPs and recomposes
 * them (src/redis.h:27-48, src/main.cpp:322-326, 375-395); the reference has no collective of its own.
 *
 * rt_gather_rows: every rank calls it with its packed block (rt_render_rows output for row0 = rank*band_rows,
 * band_rows, band_stride = nranks*band_rows; rows_per_rank*width*bytes_per_pixel bytes, the same size on every rank);
 * one group of ncclSend (every rank -> 0) / ncclRecv (rank 0 <- every rank) on the context's stream delivers them
 * into gathered_on_rank0 (nranks blocks, rank-major), and rank 0 reassembles frame_on_rank0 (height*width*
 * bytes_per_pixel; NULL: leave the blocks as gathered).  nccl_comm: the caller's ncclComm_t for these ranks on these
 * GPUs (one made by rt_comm_create, or its own); NULL with nranks == 1 (the block is reassembled in place, no RCCL).
 * Device pointers on the context's GPU.  Synchronous: on return my_block may be rendered into again and the frame is
 * complete.  Ranks must issue their gathers in the same order.  A communicator serves one gather at a time.
 *
 * rt_comm_unique_id / rt_comm_create / rt_comm_destroy: a communicator without linking RCCL into the host --
 * ncclGetUniqueId on one rank, the RT_COMM_ID_BYTES carried to the others by whatever the host has (a pipe, MPI,
 * torch.distributed), ncclCommInitRank on the context's GPU.  librccl is loaded when one of these is first called. */
#define RT_COMM_ID_BYTES 128
int  rt_comm_unique_id(unsigned char* id /* RT_COMM_ID_BYTES */);
int  rt_comm_create(void** nccl_comm_out, rt_ctx* ctx, const unsigned char* id, int32_t rank, int32_t nranks);
void rt_comm_destroy(void* nccl_comm);
int  rt_gather_rows(rt_ctx* ctx, void* nccl_comm, int32_t rank, int32_t nranks, const void* my_block, void* gathered_on_rank0,
                    void* frame_on_rank0, int32_t width, int32_t height, int32_t bytes_per_pixel, int32_t band_rows,
                    int32_t rows_per_rank);
int  rt_last_rccl_error(const rt_ctx* ctx);          /* ncclResult_t of the last failing RCCL call of the context */
/* Observability of a multi-GPU run (the reference prints its recompose time, src/main.cpp:369-397): the number of ranks
 * the communicator spans as RCCL sees it (ncclCommCount), and the device time of the context's last rt_gather_rows -- its
 * RCCL group and the reassembly on rank 0, from HIP events on the context's stream. */
int   rt_comm_count(void* nccl_comm, int32_t* nranks_out);
float rt_last_gather_ms(const rt_ctx* ctx);

/* One frame over the ranks of a job, for hosts that hold no device memory of their own (the C++ host of
 * raytracer_amd/host): renders this rank's cyclic band_rows-row bands of the params->width x params->height frame
 * (params->row0 / rows / band_* are ignored), gathers them with rt_gather_rows and leaves the frame's `pixel` bytes in
 * rgb_out_rank0 (height*width*3, host or device) on rank 0.  What src/main.cpp:275-395 does with a thread pool and a
 * Redis list, over GPUs.  The context keeps the band block and the gather buffer between calls. */
int  rt_render_gather(rt_ctx* ctx, const rt_camera35f* cam, const rt_params* params, void* nccl_comm, int32_t rank, int32_t nranks,
                      int32_t band_rows, uint8_t* rgb_out_rank0);

/* Glow post-process on the emission buffer: applyGlow() of the reference's CPU path (src/main.cpp:163-229 over
 * downsample :32, gaussianBlur :97 and upscale :77), the step between the recomposed frame and writePPM
 * (src/main.cpp:402-406).  rgb: height*width `pixel` bytes, modified in place; emission: height*width emissionPixel
 * (4 floats), as rt_render_rows wrote them, read only (the reference frees its copy).  Host or device pointers.
 * *levels_out (nullable): pyramid levels applied.  The CPU form is the specification -- a two-dimensional gaussian of
 * (2 size + 1)^2 clamped taps, sigma 1000, size 8 doubling per level -- not the CUDA file's separable sigma-20 filter.
 * RT_ERR_UNSUPPORTED (levels before it are applied): a level whose sizes times its factor are not the frame's -- the
 * reference indexes the upscaled level with the frame's row length there and reads out of bounds (src/main.cpp:191-198
 * over :77-95) -- i.e. width and height must be divisible by 2^levels; or a window above 257 x 257 (a fifth level). */
int rt_apply_glow(rt_ctx* ctx, uint8_t* rgb, const float* emission, int32_t width, int32_t height, int32_t* levels_out);

/* Tile streaming: replaces the producer side of the reference's Redis tile list (src/main.cpp:283-326 pushing through
 * Redis::SendImage, src/redis.h:27-36).  The frame is rendered in chunks of whole tile rows; after each chunk every
 * tile of it is handed to `sink` as the two records the reference pushes per tile, in this order:
 *     [int32 tile_x][int32 tile_y][tile*tile x 3 bytes  of `pixel`]          8 + tile*tile*3  bytes
 *     [int32 tile_x][int32 tile_y][tile*tile x 4 floats of `emissionPixel`]  8 + tile*tile*16 bytes
 * (rows of a tile top to bottom, host byte order: exactly what Redis::ReceiveImage unpacks, redis.h:38-48).  Like the
 * reference, only whole tiles are produced: width / tile columns and height / tile rows of tiles (src/main.cpp:276).
 * Tiles arrive row by row, left to right (the reference's order depends on its thread pool).  `rows_per_chunk` is
 * rounded down to whole tile rows (0: the whole frame in one chunk); progress = records delivered * 100 /
 * (2 * tiles), the figure src/main.cpp:337 prints.  A non-zero return of `sink` stops the stream (RT_ERR_INVALID_ARG).
 * params->row0 / rows / band_* are ignored: the call covers the frame. */
typedef int (*rt_tile_sink)(void* user, const uint8_t* record, size_t bytes);
int rt_render_tiles(rt_ctx* ctx, const rt_camera35f* cam, const rt_params* params, int32_t tile, int32_t rows_per_chunk,
                    rt_tile_sink sink, void* user);

/* Durations in milliseconds of the last rt_render_rows, from HIP events on the context's stream: from the launch of
 * the direct kernel to the end of all render kernels of the call (direct, listed, heavy, resume: they overlap), and
 * all device work of the call (pixel classification included). */
float rt_last_kernel_ms(const rt_ctx* ctx);
float rt_last_device_ms(const rt_ctx* ctx);
/* hipError_t of the last failing HIP call (0 if none). */
int rt_last_hip_error(const rt_ctx* ctx);
const char* rt_strerror(int status);

/* Unit-test hooks: run single device functions of the hot path over arrays (host pointers).
 *   rt_debug_rand     : out[i*n_draws + k] = k-th rand() after srand(seeds[i])              (src/rand.h:7,9)
 *   rt_debug_quantise : out[i] = pixel::Set channel value for in[i]                          (src/structs.h:12-17)
 *   rt_debug_intersect: one ray against hittable `index` of the uploaded scene; out = {hit, distance,
 *                       position.xyz, normal.xyz} per ray (rays: n x 6 floats origin, direction)
 *   rt_debug_cull     : out[i] = 1 if the large-scene kernels would visit the cluster of hittable `index` for ray i,
 *                       0 if they would skip it -- which they may only when rt_debug_intersect says "no hit" for
 *                       every hittable of that cluster (no counterpart in the reference: the skip rule's own test)
 *   rt_debug_cull_enter: enter[i] = the ray parameter at which ray i enters that cluster's box as the large-scene
 *                       kernels' distance rule sees it (-FLT_MAX where the rule does not apply): a closest-hit query
 *                       drops the cluster once its best distance b has enter > max(1.001 b, 0), which it may only when
 *                       no hittable of the cluster is hit at a distance below b (src/hittables/hittablesList.h:27)
 *   rt_debug_exact_math: the kernels' short correctly-rounded sqrt, reciprocal and 1/sqrt sequences against the
 *                       compiler's IEEE sqrtf and division over all 2^32 float bit patterns; mismatches3[k] = inputs on
 *                       which sequence k differs (0, 0, 0 is the claim; the reference's sqrt / division are IEEE,
 *                       src/hittables/sphere.h:25-27, glm normalize) */
int rt_debug_exact_math(rt_ctx* ctx, uint64_t* mismatches3);
int rt_debug_rand(rt_ctx* ctx, const uint32_t* seeds, int32_t n_seeds, int32_t n_draws, int32_t* out);
int rt_debug_quantise(rt_ctx* ctx, const float* in, int32_t n, uint8_t* out);
int rt_debug_intersect(rt_ctx* ctx, int32_t index, const float* rays, int32_t n_rays, float* out8);
int rt_debug_cull(rt_ctx* ctx, int32_t index, const float* rays, int32_t n_rays, uint8_t* out);
int rt_debug_cull_enter(rt_ctx* ctx, int32_t index, const float* rays, int32_t n_rays, float* enter);

#ifdef __cplusplus
}
#endif
#endif
