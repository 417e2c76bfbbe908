// Interface between the C-ABI layer (rt_abi.cpp) and the kernels (rt_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Bounce-stack capacities.  A path's frames are its ancestors: at most `max_depth` reflection parents (each adds
// one to depth, renderer.h:68,82) plus the refraction parents, whose count is bounded by multiplier *= 0.9f until
// < 0.001f (renderer.h:30,118): 66 in float arithmetic (rt_abi.cpp recomputes it).  Shallow serves max_depth <= 14,
// deep serves RT_MAX_DEPTH (64).
// pixel classes: 0 = direct, 1 + material index (RT_MAX_MATERIALS = 256)
#define RT_MAX_KEYS 257
#define RT_HEAVY_REC_WORDS 16
#define RT_HEAVY_PROBE 4      /* samples the listed kernel does before a pixel can be handed over */
// the listed render_kernel runs twice over the same list when long pixels are handed over (heavy_cap > 0)
#define RT_MODE_ALL    0   /* owns every pixel of the list: hands the steady long ones over, keeps the rest */
#define RT_MODE_STEADY 1   /* owns the pixels whose first samples draw equal counts and are long: hands them over */
#define RT_MODE_REST   2   /* owns all other pixels */

#define RT_STACK_SHALLOW 80
#define RT_PEND_SHALLOW  15
#define RT_STACK_DEEP    132
#define RT_PEND_DEEP     65

struct KernelArgs {
    float cam[35];               // position, inverse projection, inverse view (rt_camera35f)
    int width, height, samples, max_depth;
    unsigned int seed_base;      // low 32 bits: srand() takes an unsigned int (rand.h:7)
    int row0, rows, band_rows, band_stride;
    int tiles_x, tiles_y;
    int n_hit, n_lights, n_mat;
    const float4* hrec;          // 2 * n_hit
    const float4* lights;        // n_lights
    const float* mats;           // 10 * n_mat
    const float4* prim;          // n_hit: per sphere (camera - centre, |camera - centre|^2 - r^2) for primary rays
    const unsigned int* order;   // n_hit: hittable indices sorted by type, the order coop_kernel stages them in
    const unsigned int* slot_of; // n_hit: the inverse (place of hittable i in `order`)
    const float* clusters;       // ceil(n_hit / 64) x 8 floats (centre.xyz, has-cubes, half.xyz, 0): boxes of the rounds of 64 in `order`
    int n_clusters;
    unsigned char* rgb;          // rows * width * 3 (probe: width * height flags)
    float4* emission;            // rows * width, or null
    unsigned long long* counters;// 5 words (closest, shadow, tests, draws, shaded), or null
    int* fault;                  // set to 1 by a kernel whose bounce stack would overflow
    unsigned int* queue;         // coop_kernel: next entry of the pixel list to hand out (zero at launch)
    unsigned short* keys;        // rows * width pixel classes (0 = direct, 1 + material = general), or null
    unsigned int* hist;          // RT_MAX_KEYS pixel counts per key
    const unsigned int* list;    // pixel list of the listed render_kernel (indices into the packed output)
    int list_count;
    // sample-parallel pixels (listed render_kernel -> heavy_kernel); heavy_cap == 0 switches the hand-over off
    unsigned int* heavy_count;   // pixels handed over (may run past heavy_cap: those were not handed over)
    unsigned int* heavy_rec;     // heavy_cap x RT_HEAVY_REC_WORDS (heavy_handover in rt_kernels.hip lists the fields)
    unsigned int* heavy_pool;    // stream pool: each handed-over pixel's rand() sequence, written out
    unsigned long long* heavy_pool_used;   // words of the pool given out (may run past heavy_pool_words: those got none)
    unsigned int heavy_pool_words;
    int heavy_cap;
    unsigned int heavy_min_work; // hand over when draws(sample 0) * (samples left) reaches this
    unsigned int* fallback_count;// pixels heavy_kernel gave up on ...
    unsigned int* fallback_list; // ... their record indices, for resume_kernel (heavy_cap entries)
    int listed_mode;             // RT_MODE_*: which pixels of the list this launch of the listed kernel owns
    int coop_lds_stack;          // coop_kernel: colour frames of a wave's bounce stack that live in LDS (set by rt_launch_coop;
                                 // the stack's capacity or more: all of it, emission frames and pending steps included)
};

struct LaunchTraits {
    bool emission;   // emission buffer requested
    bool mixed;      // some material has reflection > 0 and refraction > 0
    bool nonsphere;  // scene holds planes or cubes
    bool deep;       // max_depth needs the deep stack
    bool count;      // tally queries
    bool listed;     // render_kernel over KernelArgs::list instead of over all tiles
    bool prune;      // with count: tally the production kernel's pruned shadow queries instead of the reference's
    bool rng_scratch;// listed render_kernel: keep the generators in scratch although LDS has room (rt_abi.cpp says when)
    bool wave_groups;// listed render_kernel with its generators in scratch: one wave per workgroup (rt_abi.cpp says when)
};

size_t rt_launch_lds_bytes(int n_hit, int n_lights, int n_mat);
hipError_t rt_launch_render(const KernelArgs& a, const LaunchTraits& t, hipStream_t stream);
hipError_t rt_launch_classify(const KernelArgs& a, const LaunchTraits& t, unsigned int* cursor, unsigned int* list, hipStream_t stream);
hipError_t rt_launch_direct(const KernelArgs& a, const LaunchTraits& t, hipStream_t stream);
hipError_t rt_launch_direct_relaxed(const KernelArgs& a, const LaunchTraits& t, hipStream_t stream);   // rt_kernels_relaxed.hip
bool rt_launch_heavy_fits(int n_hit, int n_lights, int n_mat);
bool rt_launch_coop_fits(int n_hit, int n_lights, int n_mat);
hipError_t rt_launch_coop(const KernelArgs& a, const LaunchTraits& t, hipStream_t stream);
hipError_t rt_launch_heavy(const KernelArgs& a, const LaunchTraits& t, hipStream_t stream);
hipError_t rt_launch_probe(const KernelArgs& a, const LaunchTraits& t, hipStream_t stream);
hipError_t rt_launch_deinterleave(const void* gathered, void* frame, int row_bytes, int height, int band_rows,
                                  int nranks, int rows_per_rank, hipStream_t stream);
hipError_t rt_launch_tile_pack(const void* rgb, const void* emission, int width, int tile, int tiles_x, int tile_y0, int n_tiles,
                               void* img_rec, void* em_rec, hipStream_t stream);
hipError_t rt_launch_debug_rand(const unsigned int* seeds, int n_seeds, int n_draws, int* out, hipStream_t stream);
hipError_t rt_launch_debug_exact_math(unsigned long long* out3, hipStream_t stream);
hipError_t rt_launch_debug_quantise(const float* in, int n, unsigned char* out, hipStream_t stream);
hipError_t rt_launch_debug_intersect(const float4* hrec, int index, const float* rays, int n, float* out8, hipStream_t stream);
hipError_t rt_launch_debug_cull(const float* clusters, const unsigned int* slot_of, int index, const float* rays, int n, unsigned char* out, hipStream_t stream);
hipError_t rt_launch_debug_cull_enter(const float* clusters, const unsigned int* slot_of, int index, const float* rays, int n, float* enter, hipStream_t stream);
// glow post-process (rt_glow.hip)
hipError_t rt_launch_glow_downsample(const void* src, int srcW, void* dst, int dW, int dH, hipStream_t stream);
hipError_t rt_launch_glow_blur(const void* src, void* dst, int w, int h, int size, const float* kernel, hipStream_t stream);
hipError_t rt_launch_glow_add(void* rgb, const void* down, int dW, int up, int startW, int startH, hipStream_t stream);
hipError_t rt_launch_glow_filter(void* down, int n, unsigned int* max_bits, hipStream_t stream);
