// HIP kernels of the TraceRay hot path for gfx950 and their launchers.  See DESIGN.md for the execution model.
//
// One thread owns one pixel (its RNG stream is strictly sequential: rand.h seeds per pixel, main.cpp:304), one wave
// owns an 8x8 pixel tile, one 256-thread workgroup a 16x16 tile (the reference's CPU tile, main.cpp:275).  The scene
// is staged once per workgroup into LDS; the recursion of renderer.h:28-135 is an explicit per-lane stack machine
// whose frames live in scratch, so colour and emission are still combined child-to-parent in the reference's order.
#include "rt_device.hpp"
#include "rt_launch.h"

namespace rt {

constexpr int BLOCK = 256;
// waves per SIMD the register allocator must leave room for (72 VGPRs at 7); measured in DESIGN.md "occupancy"
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 7
#endif

struct PathOut { v3 color, emission; float strength; int maxDepth; };

enum { TAG_REFL = 0, TAG_REFR = 1 };

// TraceRay (renderer.h:28-135) for one primary ray, iteratively.
//   frame (pushed when a node spawns a child) = float4(colour so far, meta) with
//   meta = material | tag << 8 | mixed << 9; tag says which branch the child is, mixed = material has both branches.
//   A non-mixed parent's emission / strength are still the material's own values when its only child returns, so
//   they are re-read instead of stored.  Mixed parents also keep (emission, strength) in stk_em and, while their
//   reflection subtree runs, what the refraction step needs (hit position, normal, incoming direction, multiplier,
//   depth) in `pend`.
template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE, bool COUNT, bool PRUNE, int MAXSP, int MAXPEND>
__device__ __forceinline__ PathOut trace_path(const SceneLds& sc, Rng& rng, v3 ro, v3 rd, int maxTotalDepth,
                                              Counters& cnt, int* fault)
{
    float4 stk[MAXSP];
    float4 stk_em[(MIXED && EMISSION) ? MAXSP : 1];
    float4 pend[MIXED ? 3 * MAXPEND : 1];
    int sp = 0, pp = 0;

    float mult = 1.0f;
    int depth = 1;
    int maxDepth = 0;

    // the node being evaluated
    v3 color = mk(0, 0, 0), emission = mk(0, 0, 0);
    float strength = 0.0f;
    int mat = 0;
    v3 hpos = mk(0, 0, 0), hnorm = mk(0, 0, 0), inDir = mk(0, 0, 0);   // inDir: the ray that hit (un-normalised)
    bool primary = true;          // wave-uniform: every lane's first query is its primary ray

    for (;;) {
        bool need_refr = false;

        // ---- TraceRay entry for (ro, rd, mult, depth)
        if (mult < 0.001f) {                                                     // renderer.h:30-34
            maxDepth = depth;
            color = mk(0, 0, 0); emission = mk(0, 0, 0); strength = 0.0f;
        } else {
            if (COUNT) ++cnt.closest;
            float dist;
            int bi;
            if (!HAS_NONSPHERE && primary) bi = closest_hit_primary<COUNT>(sc, rd, dist, cnt);
            else                           bi = closest_hit<HAS_NONSPHERE, COUNT>(sc, ro, rd, dist, cnt);
            primary = false;
            if (bi < 0) {                                                        // renderer.h:40-50
                maxDepth = depth;
                float a = (rd.y + 1.0f) * 0.5f;
                float w = 1.0f - a;
                color = mk(w + a * 0.2f, w + a * 0.3f, w + a * 0.8f) * mult;
                emission = mk(0, 0, 0); strength = 0.0f;
            } else {
                Hit h = hit_record(sc, bi, ro, rd, dist);
                v3 offPosition = h.position + h.normal * 0.005f;                 // renderer.h:55
                if (COUNT) ++cnt.shaded;
                float intensity = light_intensity<HAS_NONSPHERE, COUNT, PRUNE>(sc, offPosition, h.normal, cnt);
                mat = h.material;
                const float* m = sc.mats + 10 * mat;
                color = (mk(m[0], m[1], m[2]) * intensity) * mult;               // renderer.h:64
                if (EMISSION) { emission = mk(m[6], m[7], m[8]); strength = m[9]; }
                if (depth <= maxTotalDepth) {                                    // renderer.h:68
                    inDir = rd;                                                  // normalised only if refraction runs
                    hpos = h.position; hnorm = h.normal;
                    float reflection = m[4];
                    bool spawned = false;
                    if (reflection > 0) {                                        // renderer.h:74-91
                        v3 d = reflect(rd, hnorm);
                        v3 hemi = random_unit_hemisphere<COUNT>(rng, hnorm, cnt);
                        d = normalize(d + hemi * m[3]);
                        if (dot(d, hnorm) > 0) {
                            if (sp >= MAXSP || (MIXED && pp >= MAXPEND)) { *fault = 1; break; }
                            bool mixed = MIXED && m[5] > 0;
                            stk[sp] = make_float4(color.x, color.y, color.z,
                                                  __int_as_float(mat | (TAG_REFL << 8) | ((int)mixed << 9)));
                            if (MIXED && mixed) {
                                if (EMISSION) stk_em[sp] = make_float4(emission.x, emission.y, emission.z, strength);
                                pend[3 * pp + 0] = make_float4(hpos.x, hpos.y, hpos.z, mult);
                                pend[3 * pp + 1] = make_float4(hnorm.x, hnorm.y, hnorm.z, __int_as_float(depth));
                                pend[3 * pp + 2] = make_float4(inDir.x, inDir.y, inDir.z, 0.0f);
                                ++pp;
                            }
                            ++sp;
                            ro = offPosition; rd = d;
                            mult = mult * reflection; depth = depth + 1;
                            spawned = true;
                        }
                    }
                    if (spawned) continue;
                    need_refr = true;
                } else {
                    maxDepth = depth;                                            // renderer.h:131-132
                }
            }
        }

        // ---- refraction step of the current node, or return to the parent(s)
        bool resumed = false;
        for (;;) {
            if (need_refr) {
                need_refr = false;
                const float* m = sc.mats + 10 * mat;
                float reflection = m[4], refraction = m[5];
                if (refraction > 0) {                                            // renderer.h:93-126
                    v3 rayDir = normalize(inDir);                                // renderer.h:72
                    v3 outNorm; float ir;
                    if (dot(rayDir, hnorm) > 0) { outNorm = -hnorm; ir = refraction; }
                    else                        { outNorm = hnorm;  ir = 1.0f / refraction; }
                    v3 d;
                    if (!refract_dir(rayDir, outNorm, ir, d)) d = reflect(rayDir, hnorm);
                    v3 hemi = random_unit_hemisphere<COUNT>(rng, hnorm, cnt);
                    d = normalize(d + hemi * m[3]);
                    if (sp >= MAXSP) { *fault = 1; sp = 0; break; }
                    bool mixed = MIXED && reflection > 0;
                    stk[sp] = make_float4(color.x, color.y, color.z,
                                          __int_as_float(mat | (TAG_REFR << 8) | ((int)mixed << 9)));
                    if (MIXED && EMISSION && mixed) stk_em[sp] = make_float4(emission.x, emission.y, emission.z, strength);
                    ++sp;
                    ro = hpos + d * 0.01f; rd = d;                               // renderer.h:116
                    mult = mult * 0.9f;                                          // renderer.h:118 (same depth)
                    resumed = true;
                    break;
                }
                if (reflection == 0 && refraction == 0) maxDepth = depth;                         // renderer.h:128-129
            }
            // the current node is complete: hand (color, emission, strength) to its parent
            if (sp == 0) break;
            --sp;
            float4 fr = stk[sp];
            int meta = __float_as_int(fr.w);
            int pmat = meta & 255;
            int tag = (meta >> 8) & 1;
            bool mixed = MIXED && ((meta >> 9) & 1);
            const float* pm = sc.mats + 10 * pmat;
            color = mk(fr.x, fr.y, fr.z) + color;                                // renderer.h:83,119
            if (EMISSION) {
                v3 pem; float pst;
                if (MIXED && mixed) { float4 e = stk_em[sp]; pem = mk(e.x, e.y, e.z); pst = e.w; }
                else                { pem = mk(pm[6], pm[7], pm[8]); pst = pm[9]; }
                if (strength > 0) {                                              // renderer.h:85-89,121-125
                    pem = normalize(pem * pst + emission * strength);
                    float scaled = (tag == TAG_REFL) ? strength * pm[4] * 1.25f : strength * 0.95f;
                    pst = gmax(pst, scaled);
                }
                emission = pem; strength = pst;
            }
            mat = pmat;
            if (MIXED && mixed && tag == TAG_REFL) {
                --pp;
                float4 p0 = pend[3 * pp + 0], p1 = pend[3 * pp + 1], p2 = pend[3 * pp + 2];
                hpos = mk(p0.x, p0.y, p0.z); mult = p0.w;
                hnorm = mk(p1.x, p1.y, p1.z); depth = __float_as_int(p1.w);
                inDir = mk(p2.x, p2.y, p2.z);
                need_refr = true;
            }
        }
        if (!resumed) break;
    }

    PathOut o;
    o.color = color; o.emission = emission; o.strength = strength; o.maxDepth = maxDepth;
    return o;
}

__device__ __forceinline__ void stage_scene(const KernelArgs& a, float4* lds, SceneLds& sc, float*& lds_free)
{
    const int nrec = 2 * a.n_hit;
    float4* l_hrec = lds;
    float4* l_lights = l_hrec + nrec;
    float* l_mats = (float*)(l_lights + a.n_lights);
    const int nm = 10 * a.n_mat;
    for (int i = threadIdx.x; i < nrec; i += BLOCK) l_hrec[i] = a.hrec[i];
    for (int i = threadIdx.x; i < a.n_lights; i += BLOCK) l_lights[i] = a.lights[i];
    for (int i = threadIdx.x; i < nm; i += BLOCK) l_mats[i] = a.mats[i];
    // per-thread words start on the next 16-byte boundary
    lds_free = l_mats + ((nm + 3) & ~3);
    sc.hrec = l_hrec; sc.lights = l_lights; sc.mats = l_mats;
    sc.n_hit = a.n_hit; sc.n_lights = a.n_lights;
    sc.khrec = (kptr4)a.hrec; sc.klights = (kptr4)a.lights; sc.kprim = (kptr4)a.prim;
    __syncthreads();
}

__device__ __forceinline__ void flush_counters(const KernelArgs& a, const Counters& cnt)
{
    atomicAdd(&a.counters[0], (unsigned long long)cnt.closest);
    atomicAdd(&a.counters[1], (unsigned long long)cnt.shadow);
    atomicAdd(&a.counters[2], (unsigned long long)cnt.tests);
    atomicAdd(&a.counters[3], (unsigned long long)cnt.draws);
    atomicAdd(&a.counters[4], (unsigned long long)cnt.shaded);
}

// The per-pixel driver (main.cpp:298-319 == main.cu:38-67) over AntiAliasing (renderer.h:138-201), for pixel
// (x, y) of the frame = column x of packed output row r.
template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE, bool COUNT, bool PRUNE, int MAXSP, int MAXPEND>
__device__ __forceinline__ void render_pixel(const KernelArgs& a, const SceneLds& sc, float* dirs, int x, int y, int r, Counters& cnt)
{
    Camera cam;
#pragma unroll
    for (int i = 0; i < 3; ++i) cam.pos[i] = a.cam[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) { cam.ip[i] = a.cam[3 + i]; cam.iv[i] = a.cam[19 + i]; }

    const float u = ((float)x / (float)a.width) * 2.0f - 1.0f;              // main.cpp:301-302
    const float v = -(((float)y / (float)a.height) * 2.0f - 1.0f);          // renderer.h:140
    const float offX = 0.5f / a.width, offY = 0.5f / a.height;              // main.cpp:306-307
    const v3 origin = mk(cam.pos[0], cam.pos[1], cam.pos[2]);

    unsigned int rng_words[31];
    Rng rng;
    rng.init(rng_words, a.seed_base + (unsigned int)(x + y * a.width));                // rand.h:7, main.cpp:304

    // AntiAliasing is called with the same (u, v) for every sample (main.cpp:312), so the four primary directions
    // are per-pixel constants: computed once, parked in LDS (12 words per thread, word j of thread t at [j*BLOCK + t]).
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        float uu = (k & 1) ? u + offX : u - offX;
        float vv = (k & 2) ? v + offY : v - offY;
        v3 d = uv_to_direction(cam, uu, vv);
        dirs[(3 * k + 0) * BLOCK] = d.x; dirs[(3 * k + 1) * BLOCK] = d.y; dirs[(3 * k + 2) * BLOCK] = d.z;
    }

    v3 acc_color = mk(0, 0, 0), acc_em = mk(0, 0, 0);
    float acc_st = 0.0f;
    for (int s = 0; s < a.samples; ++s) {
        v3 res_color = mk(0, 0, 0), res_em = mk(0, 0, 0);
        float res_st = 0.0f;
#pragma unroll 1
        for (int k = 0; k < a.rays_per_sample; ++k) {                        // renderer.h:167-195 (4; 1 in the cost pre-pass)
            v3 dir = mk(dirs[(3 * k + 0) * BLOCK], dirs[(3 * k + 1) * BLOCK], dirs[(3 * k + 2) * BLOCK]);
            PathOut p = trace_path<EMISSION, MIXED, HAS_NONSPHERE, COUNT, PRUNE, MAXSP, MAXPEND>(sc, rng, origin, dir, a.max_depth, cnt, a.fault);
            // renderer.h:169: color / vec3(maxDepth); x / 1.0f == x, and most paths end at depth 1
            v3 pc = p.color;
            if (p.maxDepth != 1) { float md = (float)p.maxDepth; pc = mk(pc.x / md, pc.y / md, pc.z / md); }
            res_color = res_color + pc;
            if (EMISSION) {
                res_em = res_em + p.emission;
                res_st = (k == 0) ? p.strength : gmax(res_st, p.strength);
            }
        }
        res_color = res_color * 0.25f;
        acc_color = acc_color + mk(clamp01(res_color.x), clamp01(res_color.y), clamp01(res_color.z));   // main.cpp:313
        if (EMISSION) {
            res_em = res_em * 0.25f;
            acc_em = acc_em + mk(clamp01(res_em.x), clamp01(res_em.y), clamp01(res_em.z));
            acc_st += res_st;
        }
    }

    const float n = (float)a.samples;
    const size_t o = (size_t)x + (size_t)r * a.width;
    if (a.rgb) {                                                             // main.cpp:318 -> structs.h:12-17
        a.rgb[o * 3 + 0] = quantise(acc_color.x / n);
        a.rgb[o * 3 + 1] = quantise(acc_color.y / n);
        a.rgb[o * 3 + 2] = quantise(acc_color.z / n);
    }
    if (EMISSION && a.emission)                                              // main.cpp:319 -> structs.h:39-43
        a.emission[o] = make_float4(acc_em.x / n, acc_em.y / n, acc_em.z / n, acc_st / n);
}

// One workgroup = one 16x16 tile (four 8x8 waves).  a.order (optional) maps launch order to tile index so that
// expensive tiles start first; a.tile_cost (counting builds only) receives the tile's primitive-test count instead of
// the global tallies -- that is the cost pre-pass (DESIGN.md "tile order").
template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE, bool COUNT, bool PRUNE, int MAXSP, int MAXPEND>
__global__ __launch_bounds__(BLOCK, RT_MIN_WAVES) void render_kernel(KernelArgs a)
{
    extern __shared__ float4 lds[];
    SceneLds sc;
    float* lds_free;
    stage_scene(a, lds, sc, lds_free);

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int tile = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int x = tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int r = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool active = x < a.width && r < a.rows;
    const int y = a.band_rows > 0 ? a.row0 + (r / a.band_rows) * a.band_stride + (r % a.band_rows) : a.row0 + r;

    Counters cnt = { 0, 0, 0, 0, 0 };
    if (active)
        render_pixel<EMISSION, MIXED, HAS_NONSPHERE, COUNT, PRUNE, MAXSP, MAXPEND>(a, sc, lds_free + tid, x, y, r, cnt);

    if (COUNT) {
        if (a.tile_cost) {
            unsigned int* acc = (unsigned int*)(lds_free + 12 * BLOCK);
            if (tid == 0) *acc = 0;
            __syncthreads();
            if (active) atomicAdd(acc, cnt.tests);
            __syncthreads();
            if (tid == 0) a.tile_cost[tile] = *acc;
        } else if (active) {
            flush_counters(a, cnt);
        }
    }
}

// Launch order from tile costs: counting sort by a 7-bit log-scale key, most expensive first.  One workgroup.
__global__ __launch_bounds__(1024) void order_kernel(const unsigned int* __restrict__ cost, int n, int* __restrict__ order)
{
    __shared__ unsigned int hist[128];
    __shared__ unsigned int start[128];
    const int tid = threadIdx.x;
    if (tid < 128) hist[tid] = 0;
    __syncthreads();
    auto key = [](unsigned int c) -> int {
        // descending cost -> ascending key: 4 buckets per octave
        if (c == 0) return 127;
        int e = 31 - __clz(c);
        int m = e >= 2 ? (int)((c >> (e - 2)) & 3) : 0;
        int k = e * 4 + m;                 // 0 .. 127
        return 127 - (k > 127 ? 127 : k);
    };
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[key(cost[i])], 1u);
    __syncthreads();
    if (tid == 0) {
        unsigned int s = 0;
        for (int k = 0; k < 128; ++k) { start[k] = s; s += hist[k]; }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) order[atomicAdd(&start[key(cost[i])], 1u)] = i;
}

// MaxRayDistance (renderer.h:204-219) per pixel, as tests/singlePixel.cpp:74-83 drives it.
template <bool MIXED, bool HAS_NONSPHERE, int MAXSP, int MAXPEND>
__global__ __launch_bounds__(BLOCK) void probe_kernel(KernelArgs a)
{
    extern __shared__ float4 lds[];
    SceneLds sc;
    float* lds_free;
    stage_scene(a, lds, sc, lds_free);

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    const int x = tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (x >= a.width || y >= a.height) return;

    Camera cam;
#pragma unroll
    for (int i = 0; i < 3; ++i) cam.pos[i] = a.cam[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) { cam.ip[i] = a.cam[3 + i]; cam.iv[i] = a.cam[19 + i]; }
    const float u = ((float)x / (float)a.width) * 2.0f - 1.0f;
    const float v = -(((float)y / (float)a.height) * 2.0f - 1.0f);
    unsigned int rng_words[31];
    Rng rng;
    rng.init(rng_words, a.seed_base + (unsigned int)(x + y * a.width));
    Counters cnt = { 0, 0, 0, 0, 0 };
    v3 dir = uv_to_direction(cam, u, v);
    PathOut p = trace_path<false, MIXED, HAS_NONSPHERE, false, true, MAXSP, MAXPEND>(
        sc, rng, mk(cam.pos[0], cam.pos[1], cam.pos[2]), dir, a.max_depth, cnt, a.fault);
    a.rgb[(size_t)x + (size_t)y * a.width] = p.maxDepth >= a.max_depth;
}

// Frame reassembly after the band gather: out row y comes from rank (y / band_rows) % nranks, local row
// (y / (band_rows * nranks)) * band_rows + y % band_rows.  16 bytes per thread where alignment allows.
__global__ void deinterleave_kernel(const unsigned char* __restrict__ gathered, unsigned char* __restrict__ frame,
                                    int row_bytes, int height, int band_rows, int nranks, int rows_per_rank)
{
    const int y = blockIdx.y;
    const int band = y / band_rows;
    const int rank = band % nranks;
    const int local = (band / nranks) * band_rows + (y % band_rows);
    const unsigned char* src = gathered + ((size_t)rank * rows_per_rank + local) * row_bytes;
    unsigned char* dst = frame + (size_t)y * row_bytes;
    const bool vec = ((((size_t)src | (size_t)dst) & 15) == 0) && (row_bytes % 16 == 0);
    if (vec) {
        const int n16 = row_bytes / 16;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x)
            ((uint4*)dst)[i] = ((const uint4*)src)[i];
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes; i += gridDim.x * blockDim.x)
            dst[i] = src[i];
    }
}

// ---- unit-test kernels (rt_debug_*) -------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void debug_rand_kernel(const unsigned int* seeds, int n_seeds, int n_draws, int* out)
{
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_seeds) return;
    unsigned int rng_words[31];
    Rng rng;
    rng.init(rng_words, seeds[i]);
    rng.do_seed();
    // mix the one-at-a-time and the three-at-a-time forms
    int k = 0;
    for (; k + 4 <= n_draws; k += 4) {
        int* o = out + (size_t)i * n_draws + k;
        o[0] = rng.next();
        rng.next3(o[1], o[2], o[3]);
    }
    for (; k < n_draws; ++k) out[(size_t)i * n_draws + k] = rng.next();
}

__global__ void debug_quantise_kernel(const float* in, int n, unsigned char* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = quantise(in[i]);
}

__global__ void debug_intersect_kernel(const float4* hrec, int index, const float* rays, int n, float* out8)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SceneLds sc; sc.hrec = hrec; sc.lights = nullptr; sc.mats = nullptr; sc.n_hit = index + 1; sc.n_lights = 0; sc.khrec = (kptr4)hrec; sc.klights = nullptr; sc.kprim = nullptr;
    v3 ro = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), rd = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    float4 r0 = hrec[2 * index], r1 = hrec[2 * index + 1];
    float fa = 4.0f * dot(rd, rd);
    v3 inv = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    float d = -1.0f;
    bool hit = hittable_hit<true>(r0, r1, ro, rd, fa, inv, d);
    float* o = out8 + 8 * i;
    o[0] = hit ? 1.0f : 0.0f;
    if (hit) {
        Hit h = hit_record(sc, index, ro, rd, d);
        o[1] = d; o[2] = h.position.x; o[3] = h.position.y; o[4] = h.position.z;
        o[5] = h.normal.x; o[6] = h.normal.y; o[7] = h.normal.z;
    } else {
        for (int k = 1; k < 8; ++k) o[k] = 0.0f;
    }
}

}  // namespace rt

// ---- launchers ------------------------------------------------------------------------------------------------
using namespace rt;

template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE, bool COUNT, bool PRUNE, int MAXSP, int MAXPEND>
static hipError_t launch_one(const KernelArgs& a, int tiles, size_t lds_bytes, hipStream_t stream)
{
    auto k = render_kernel<EMISSION, MIXED, HAS_NONSPHERE, COUNT, PRUNE, MAXSP, MAXPEND>;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(BLOCK), lds_bytes, stream, a);
    return hipGetLastError();
}

template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE>
static hipError_t launch_depth(const KernelArgs& a, bool deep, int tiles, size_t lds, hipStream_t s)
{
    if (deep) return launch_one<EMISSION, MIXED, HAS_NONSPHERE, false, true, RT_STACK_DEEP, RT_PEND_DEEP>(a, tiles, lds, s);
    return launch_one<EMISSION, MIXED, HAS_NONSPHERE, false, true, RT_STACK_SHALLOW, RT_PEND_SHALLOW>(a, tiles, lds, s);
}

size_t rt_launch_lds_bytes(int n_hit, int n_lights, int n_mat)
{
    size_t nm = (size_t)((10 * n_mat + 3) & ~3);
    return (size_t)n_hit * 32 + (size_t)n_lights * 16 + nm * 4 + (size_t)12 * BLOCK * 4 + 16;
}

hipError_t rt_launch_render(const KernelArgs& a, const LaunchTraits& t, hipStream_t stream)
{
    const int tiles = a.tiles_x * a.tiles_y;
    if (tiles <= 0) return hipSuccess;
    const size_t lds = rt_launch_lds_bytes(a.n_hit, a.n_lights, a.n_mat);
    // the tally builds are the most general ones: emission, mixed, all primitive types, deep stack
    if (t.count && !t.prune) return launch_one<true, true, true, true, false, RT_STACK_DEEP, RT_PEND_DEEP>(a, tiles, lds, stream);
    if (t.count && !t.cost) return launch_one<true, true, true, true, true, RT_STACK_DEEP, RT_PEND_DEEP>(a, tiles, lds, stream);
    if (t.count) {   // cost pre-pass: tallies primitive tests per tile, no emission bookkeeping
        if (!t.mixed && !t.nonsphere && !t.deep) return launch_one<false, false, false, true, true, RT_STACK_SHALLOW, RT_PEND_SHALLOW>(a, tiles, lds, stream);
        return launch_one<false, true, true, true, true, RT_STACK_DEEP, RT_PEND_DEEP>(a, tiles, lds, stream);
    }
#define RT_DISPATCH(E, M, N) if (t.emission == E && t.mixed == M && t.nonsphere == N) return launch_depth<E, M, N>(a, t.deep, tiles, lds, stream);
    RT_DISPATCH(false, false, false) RT_DISPATCH(false, false, true)
    RT_DISPATCH(false, true, false)  RT_DISPATCH(false, true, true)
    RT_DISPATCH(true, false, false)  RT_DISPATCH(true, false, true)
    RT_DISPATCH(true, true, false)   RT_DISPATCH(true, true, true)
#undef RT_DISPATCH
    return hipErrorInvalidValue;
}

hipError_t rt_launch_order(const unsigned int* cost, int n, int* order, hipStream_t stream)
{
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, stream, cost, n, order);
    return hipGetLastError();
}

hipError_t rt_launch_probe(const KernelArgs& a, const LaunchTraits& t, hipStream_t stream)
{
    const int tiles = a.tiles_x * a.tiles_y;
    if (tiles <= 0) return hipSuccess;
    const size_t lds = rt_launch_lds_bytes(a.n_hit, a.n_lights, a.n_mat);
    auto k = probe_kernel<true, true, RT_STACK_DEEP, RT_PEND_DEEP>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(BLOCK), lds, stream, a);
    return hipGetLastError();
}

hipError_t rt_launch_deinterleave(const void* gathered, void* frame, int row_bytes, int height, int band_rows,
                                  int nranks, int rows_per_rank, hipStream_t stream)
{
    if (height <= 0 || row_bytes <= 0) return hipSuccess;
    int bx = (row_bytes / 16 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(deinterleave_kernel, dim3(bx, height), dim3(256), 0, stream, (const unsigned char*)gathered,
                       (unsigned char*)frame, row_bytes, height, band_rows, nranks, rows_per_rank);
    return hipGetLastError();
}

hipError_t rt_launch_debug_rand(const unsigned int* seeds, int n_seeds, int n_draws, int* out, hipStream_t stream)
{
    hipLaunchKernelGGL(debug_rand_kernel, dim3((n_seeds + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, seeds, n_seeds, n_draws, out);
    return hipGetLastError();
}

hipError_t rt_launch_debug_quantise(const float* in, int n, unsigned char* out, hipStream_t stream)
{
    hipLaunchKernelGGL(debug_quantise_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, in, n, out);
    return hipGetLastError();
}

hipError_t rt_launch_debug_intersect(const float4* hrec, int index, const float* rays, int n, float* out8, hipStream_t stream)
{
    hipLaunchKernelGGL(debug_intersect_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, hrec, index, rays, n, out8);
    return hipGetLastError();
}
