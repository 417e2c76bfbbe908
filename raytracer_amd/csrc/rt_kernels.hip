// HIP kernels of the TraceRay hot path for gfx950 and their launchers.  DESIGN.md section 5 has the execution model.
//
// One thread owns one pixel (its RNG stream is strictly sequential: rand.h seeds per pixel, main.cpp:304) unless noted.
//   classify / scan / scatter -- which pixels can bounce (one closest-hit query per primary direction), compacted into
//                     a material-grouped list;
//   direct_kernel  -- all other pixels: every ray is closest hit + one shading, no RNG, no bounce stack, no scratch;
//                     8x8-pixel waves, 16x16 tiles, per-wave candidate lists;
//   render_kernel  -- the general per-pixel driver (iterative TraceRay with its stack in scratch): over the list, twice
//                     (one launch owns the long pixels whose draw counts repeat and hands them over, one the rest), or
//                     over every pixel for tallying builds and RT_FLAG_SINGLE_KERNEL;
//   heavy_kernel   -- handed-over pixels, one wave per pixel, one lane per sample, draws read from the pixel's written-
//                     out rand() stream at guessed indices; resume_kernel finishes what it gives up on;
//   coop_kernel    -- listed pixels of large scenes, one wave per pixel, the lanes share every scan of the scene;
//   tile_pack / deinterleave / probe / debug kernels.
#include "rt_launch.h"
#include "rt_device.hpp"
#include <type_traits>
#include <cstdio>
#include <cstdlib>

namespace RT_NS {

constexpr int BLOCK = 256;
// waves per SIMD the register allocator must leave room for in the tiled general kernel (72 VGPRs at 7)
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 7
#endif
// the listed render_kernel (DESIGN.md "execution model"): generator words in LDS, scene records fetched in batches
#ifndef RT_LISTED_LDS_RNG
#define RT_LISTED_LDS_RNG 1
#endif
#ifndef RT_LISTED_BATCH
#define RT_LISTED_BATCH 4
#endif
#ifndef RT_DIRECT_CULL
#define RT_DIRECT_CULL 1
#endif
#ifndef RT_DIRECT_BATCH
#define RT_DIRECT_BATCH 4
#endif
// waves per SIMD the sphere-only direct kernel's register allocation must leave room for (80 VGPRs at 6; left alone it
// takes 85 and runs at 4.7 instead of 5.4 resident waves per SIMD; the variants with planes / cubes or an emission
// buffer: below)
#ifndef RT_DIRECT_WAVES
#define RT_DIRECT_WAVES 6
#endif
// ... and the variants with planes / cubes or an emission buffer: 96 registers at 5 (left alone: 100 - 157; 4 - 83 of
// them spilled then).  Not for the kernel's own speed (a frame alone takes the same time) but for frames in flight whose
// listed waves live long: a SIMD that holds five of those (96 registers each) has room for a direct wave of this size as
// soon as one of them ends, and the direct kernels of glass 1080p no longer wait ~ 85 ms each for room (22.7 -> 20.3 -
// 21.3 ms per frame with 16 in flight; 4, 6 or 8 waves: 21.8 / 20.9 - 22.2 / 20.6 - 20.9)
#ifndef RT_DIRECT_NS_WAVES
#define RT_DIRECT_NS_WAVES 5
#endif
// scenes with at least this many hittables: direct_kernel<.., BIG>
#ifndef RT_DIRECT_BIG
#define RT_DIRECT_BIG 256
#endif
#ifndef RT_LISTED_WAVES
#define RT_LISTED_WAVES 4
#endif
// the listed render_kernel of a sequence's long frames (LaunchTraits::wave_groups): threads per workgroup, and the waves
// per SIMD its registers must leave room for (5: 96 registers and 7 - 51 of them spilled, where 4 leaves 100 - 117 and
// spills nothing -- the frames this variant is for are short of slots: glass 1080p in flight 23.4 -> 22.3 ms per frame;
// 6 is slower again)
#ifndef RT_LISTED_SEQ_BLOCK
#define RT_LISTED_SEQ_BLOCK 64
#endif
#ifndef RT_LISTED_SEQ_WAVES
#define RT_LISTED_SEQ_WAVES 5
#endif
// the listed render_kernel's per-wave candidate lists for the primary segment (listed_wave_lists)
#ifndef RT_LISTED_LISTS
#define RT_LISTED_LISTS 1
#endif
// coop_kernel: the path state in scalar registers (rt_device.hpp uni()): the kernel's 64 lanes hold one pixel's path, and
// telling the compiler so takes the register spills at 80 VGPRs from 45 to 19 and the kernel's HBM traffic on C4 from
// 215 to 96 GB per frame, for 1.4 % more instructions and the same time (547 against 551 ms; round 2 measured it 1.5 %
// slower and left it off -- the spills were not, and are not, what the frame waits for, but they were most of its traffic).
#ifndef RT_COOP_UNIFORM
#define RT_COOP_UNIFORM 1
#endif
// how many of the first four samples must have drawn the same count for a pixel to be handed over (measured: with 3
// or 2, C5 gets slower -- 186 / 210 instead of 175 ms -- and nothing gets faster)
#ifndef RT_HEAVY_AGREE
#define RT_HEAVY_AGREE 4
#endif
static_assert(RT_HEAVY_PROBE == 4, "render_pixel keeps the counts of exactly four first samples");

struct PathOut { v3 color, emission; float strength; int maxDepth; };

enum { TAG_REFL = 0, TAG_REFR = 1 };

// TraceRay (renderer.h:28-135) for one primary ray, iteratively.
//   frame (pushed when a node spawns a child) = float4(colour so far, meta) with
//   meta = material | tag << 8 | mixed << 9; tag says which branch the child is, mixed = material has both branches.
//   A non-mixed parent's emission / strength are still the material's own values when its only child returns, so
//   they are re-read instead of stored.  Mixed parents also keep (emission, strength) in stk_em and, while their
//   reflection subtree runs, what the refraction step needs (hit position, normal, incoming direction, multiplier,
//   depth) in `pend`.
template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE, bool COUNT, bool PRUNE, int MAXSP, int MAXPEND, class RNG = Rng, int BATCH = 1>
__device__ __forceinline__ PathOut trace_path(const SceneLds& sc, RNG& rng, v3 ro, v3 rd, int maxTotalDepth,
                                              Counters& cnt, int* fault)
{
    float4 stk[MAXSP];
    float4 stk_em[(MIXED && EMISSION) ? MAXSP : 1];
    float4 pend[MIXED ? 3 * MAXPEND : 1];
    int sp = 0, pp = 0;
    // coop_kernel: the 64 lanes of the wave hold one path, so one copy of its frames is enough -- lane 0's; the others
    // neither write theirs nor use what they read back
    // Where LDS has room for it (sc.xstack: this wave's 2 MAXSP + 3 MAXPEND frames) that copy lives there: a pop then
    // costs an LDS read instead of a trip to the L2.
    // Where LDS has room for only part of it (large scenes: sc.xstack_n < MAXSP), the first xstack_n colour frames
    // live there -- paths rarely go deeper -- and the rest, like the emission frames and the pending refraction steps,
    // in lane 0's scratch, which only lane 0 touches: a pop there is one lane's read, broadcast.
    constexpr bool ONE_COPY = BATCH == RT_COOP;
    const bool keeper = !ONE_COPY || (threadIdx.x & 63) == 0;
    float4* const xs = ONE_COPY ? sc.xstack : nullptr;
    const int xn = ONE_COPY ? sc.xstack_n : 0;
    auto in_lds = [&](const float4* arr, int i) { return xs != nullptr && (xn >= MAXSP || (arr == stk && i < xn)); };
    auto put = [&](float4* arr, int i, float4 v) {
        if (ONE_COPY && in_lds(arr, i)) { if (keeper) xs[(arr == stk ? 0 : arr == stk_em ? MAXSP : 2 * MAXSP) + i] = v; }
        else if (keeper) arr[i] = v;
    };
    auto get = [&](const float4* arr, int i) -> float4 {
        if (ONE_COPY && in_lds(arr, i)) return xs[(arr == stk ? 0 : arr == stk_em ? MAXSP : 2 * MAXSP) + i];       // (every lane reads the one copy)
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (keeper) v = arr[i];
        if (ONE_COPY) {
            v.x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.x)));
            v.y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.y)));
            v.z = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.z)));
            v.w = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.w)));
        }
        return v;
    };

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
#if RT_COOP_UNIFORM
        if (ONE_COPY) {                                  // (rt_device.hpp uni(): the path state into scalar registers)
            make_uniform<ONE_COPY>(ro); make_uniform<ONE_COPY>(rd); make_uniform<ONE_COPY>(mult); make_uniform<ONE_COPY>(depth);
            make_uniform<ONE_COPY>(maxDepth); make_uniform<ONE_COPY>(sp); make_uniform<ONE_COPY>(pp);
        }
#endif

        // ---- TraceRay entry for (ro, rd, mult, depth)
        if (mult < 0.001f) {                                                     // renderer.h:30-34
            maxDepth = depth;
            color = mk(0, 0, 0); emission = mk(0, 0, 0); strength = 0.0f;
        } else {
            if (COUNT) ++cnt.closest;
            float dist;
            int bi;
            const bool listed_query = primary && sc.cl_nprim >= 0;      // a wave's (coop_kernel: a pixel's) candidate lists
            if (listed_query)                                  { RT_CSTAT(9, 1); bi = closest_hit_ldslist<HAS_NONSPHERE>(sc, sc.cl_prim, sc.cl_nprim, ro, rd, dist); }
            else if (!HAS_NONSPHERE && primary && BATCH != RT_COOP) bi = closest_hit_primary<COUNT>(sc, rd, dist, cnt);
            else                                               bi = closest_hit<HAS_NONSPHERE, COUNT, BATCH>(sc, ro, rd, dist, cnt);
            primary = false;
            if (bi < 0) {                                                        // renderer.h:40-50
                maxDepth = depth;
                float a = (rd.y + 1.0f) * 0.5f;
                float w = 1.0f - a;
                color = mk(w + a * 0.2f, w + a * 0.3f, w + a * 0.8f) * mult;
                emission = mk(0, 0, 0); strength = 0.0f;
            } else {
                Hit h = hit_record<!HAS_NONSPHERE>(sc, bi, ro, rd, dist);
                v3 offPosition = h.position + h.normal * 0.005f;                 // renderer.h:55
                if (COUNT) ++cnt.shaded;
                mat = h.material;
                const float* m = sc.mats + 10 * mat;
                // A material whose colour is zero in every channel -- the reference's glass, postgres.h:210 -- shades to
                // (0 * intensity) * multiplier = 0 whatever the lights say: GetLightIntensity is max(0.085, Lambert terms),
                // a positive finite number (a NaN term fails `intens > intensity`, lightsList.h:37), and the product keeps
                // the zero's sign.  Its shadow queries decide nothing and are left out (not by the tallying builds).
                const bool black = PRUNE && !COUNT && m[0] == 0.0f && m[1] == 0.0f && m[2] == 0.0f;
                float intensity = 0.085f;
                if (!black)
                    intensity = (listed_query && sc.cl_noccl >= 0)
                        ? light_intensity_ldslist<HAS_NONSPHERE>(sc, sc.cl_occl, sc.cl_noccl, offPosition, h.normal)
                        : light_intensity<HAS_NONSPHERE, COUNT, PRUNE, BATCH>(sc, offPosition, h.normal, cnt);
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
                            put(stk, sp, make_float4(color.x, color.y, color.z,
                                                     __int_as_float(mat | (TAG_REFL << 8) | ((int)mixed << 9))));
                            if (MIXED && mixed) {
                                if (EMISSION) put(stk_em, sp, make_float4(emission.x, emission.y, emission.z, strength));
                                put(pend, 3 * pp + 0, make_float4(hpos.x, hpos.y, hpos.z, mult));
                                put(pend, 3 * pp + 1, make_float4(hnorm.x, hnorm.y, hnorm.z, __int_as_float(depth)));
                                put(pend, 3 * pp + 2, make_float4(inDir.x, inDir.y, inDir.z, 0.0f));
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
#if RT_COOP_UNIFORM
            if (ONE_COPY) {
                make_uniform<ONE_COPY>(color); make_uniform<ONE_COPY>(mat); make_uniform<ONE_COPY>(sp); make_uniform<ONE_COPY>(pp);
                make_uniform<ONE_COPY>(hpos); make_uniform<ONE_COPY>(hnorm); make_uniform<ONE_COPY>(inDir);
                make_uniform<ONE_COPY>(mult); make_uniform<ONE_COPY>(depth); make_uniform<ONE_COPY>(maxDepth);
                if (EMISSION) { make_uniform<ONE_COPY>(emission); make_uniform<ONE_COPY>(strength); }
            }
#endif
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
                    put(stk, sp, make_float4(color.x, color.y, color.z,
                                             __int_as_float(mat | (TAG_REFR << 8) | ((int)mixed << 9))));
                    if (MIXED && EMISSION && mixed) put(stk_em, sp, make_float4(emission.x, emission.y, emission.z, strength));
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
            if (!MIXED && !EMISSION && BATCH > 1) {
                // every remaining frame only adds its colour (renderer.h:83,119): fetch four frames per round trip,
                // add them in the same order
                while (sp >= 4) {
                    float4 f0 = get(stk, sp - 1), f1 = get(stk, sp - 2), f2 = get(stk, sp - 3), f3 = get(stk, sp - 4);
                    sp -= 4;
                    color = mk(f0.x, f0.y, f0.z) + color;
                    color = mk(f1.x, f1.y, f1.z) + color;
                    color = mk(f2.x, f2.y, f2.z) + color;
                    color = mk(f3.x, f3.y, f3.z) + color;
                }
                if (sp == 0) break;
            }
            --sp;
            float4 fr = get(stk, sp);
            int meta = __float_as_int(fr.w);
            int pmat = meta & 255;
            int tag = (meta >> 8) & 1;
            bool mixed = MIXED && ((meta >> 9) & 1);
            const float* pm = sc.mats + 10 * pmat;
            color = mk(fr.x, fr.y, fr.z) + color;                                // renderer.h:83,119
            if (EMISSION) {
                v3 pem; float pst;
                if (MIXED && mixed) { float4 e = get(stk_em, sp); pem = mk(e.x, e.y, e.z); pst = e.w; }
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
                float4 p0 = get(pend, 3 * pp + 0), p1 = get(pend, 3 * pp + 1), p2 = get(pend, 3 * pp + 2);
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

// HREC false (direct_kernel on large scenes): the hittables stay in global memory -- wave-uniform reads go through
// scalar loads anyway (urec), the few per-lane ones (the winner's record) through the vector cache.
template <bool HREC = true>
__device__ __forceinline__ void stage_scene(const KernelArgs& a, float4* lds, SceneLds& sc, float*& lds_free)
{
    const int nrec = HREC ? 2 * a.n_hit : 0;
    float4* l_hrec = lds;
    float4* l_lights = l_hrec + nrec;
    float* l_mats = (float*)(l_lights + a.n_lights);
    const int nm = 10 * a.n_mat;
    const int nthreads = (int)blockDim.x;
    for (int i = threadIdx.x; i < nrec; i += nthreads) l_hrec[i] = a.hrec[i];
    for (int i = threadIdx.x; i < a.n_lights; i += nthreads) l_lights[i] = a.lights[i];
    for (int i = threadIdx.x; i < nm; i += nthreads) l_mats[i] = a.mats[i];
    // per-thread words start on the next 16-byte boundary
    lds_free = l_mats + ((nm + 3) & ~3);
    sc.hrec = HREC ? l_hrec : a.hrec; sc.lights = l_lights; sc.mats = l_mats;
    sc.n_hit = a.n_hit; sc.n_lights = a.n_lights;
    sc.khrec = (kptr4)a.hrec; sc.klights = (kptr4)a.lights; sc.kprim = (kptr4)a.prim;
    sc.orig = nullptr; sc.slot_of = nullptr; sc.clusters = nullptr; sc.n_clusters = 0; sc.xstack = nullptr; sc.xstack_n = 0;
    sc.cl_prim = nullptr; sc.cl_nprim = -1; sc.cl_occl = nullptr; sc.cl_noccl = -1;
    __syncthreads();
}

// coop_kernel's staging: the hittables in the order of a.order -- by type, so that lane l of a cooperative scan reads
// slot base + l and the 64 tests of a round mostly take one branch, and within a type along a space-filling curve, so
// that a round's hittables sit together in space and whole rounds can be skipped (rt_device.hpp cluster_mask) -- with
// l_orig[slot] = index in the scene and the rounds' boxes behind them.
__device__ __forceinline__ void stage_scene_sorted(const KernelArgs& a, float4* lds, SceneLds& sc, float*& lds_free)
{
    stage_scene(a, lds, sc, lds_free);
    float4* l_hrec = lds;
    unsigned short* l_orig = (unsigned short*)lds_free;
    float* l_clusters = lds_free + (((size_t)a.n_hit * 2 + 15) & ~(size_t)15) / 4;
    const int nthreads = (int)blockDim.x;
    __syncthreads();                                   // (stage_scene's copy is replaced: nobody reads it in between)
    for (int j = threadIdx.x; j < a.n_hit; j += nthreads) {
        const unsigned int i = a.order[j];
        l_hrec[2 * j] = a.hrec[2 * i]; l_hrec[2 * j + 1] = a.hrec[2 * i + 1];
        l_orig[j] = (unsigned short)i;
    }
    for (int j = threadIdx.x; j < 8 * a.n_clusters; j += nthreads) l_clusters[j] = a.clusters[j];
    lds_free = l_clusters + 8 * a.n_clusters;
    sc.orig = l_orig; sc.slot_of = a.slot_of;
    sc.clusters = l_clusters; sc.n_clusters = a.n_clusters;
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

// Pixel -> frame coordinates.  Tiled: workgroup = 16x16 tile, wave = 8x8 block.  Listed: lane i of the grid takes
// entry i of the pixel list (index into the packed output, row-major).
struct PixelId { int x, y, r; bool active; };

__device__ __forceinline__ int frame_row(const KernelArgs& a, int r)
{
    return a.band_rows > 0 ? a.row0 + (r / a.band_rows) * a.band_stride + (r % a.band_rows) : a.row0 + r;
}

__device__ __forceinline__ PixelId tiled_pixel(const KernelArgs& a, int tid = (int)threadIdx.x)
{
    const int wave = tid >> 6, lane = tid & 63;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    PixelId p;
    p.x = tx * 16 + (wave & 1) * 8 + (lane & 7);
    p.r = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    p.active = p.x < a.width && p.r < a.rows;
    p.y = frame_row(a, p.r);
    return p;
}

template <int WG = BLOCK>
__device__ __forceinline__ PixelId listed_pixel(const KernelArgs& a)
{
    const int i = blockIdx.x * WG + threadIdx.x;
    PixelId p;
    p.active = (unsigned int)i < (unsigned int)a.list_count;
    const unsigned int o = p.active ? a.list[i] : 0u;
    p.x = (int)(o % (unsigned int)a.width);
    p.r = (int)(o / (unsigned int)a.width);
    p.y = frame_row(a, p.r);
    return p;
}

// The four primary directions of a pixel (renderer.h:140,167,175,183,191 over main.cpp:301-307) into LDS words.
template <int STRIDE = BLOCK>
__device__ __forceinline__ void primary_directions(const KernelArgs& a, float* dirs, int x, int y)
{
    Camera cam;
#pragma unroll
    for (int i = 0; i < 3; ++i) cam.pos[i] = a.cam[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) { cam.ip[i] = a.cam[3 + i]; cam.iv[i] = a.cam[19 + i]; }
    const float u = ((float)x / (float)a.width) * 2.0f - 1.0f;
    const float v = -(((float)y / (float)a.height) * 2.0f - 1.0f);
    const float offX = 0.5f / a.width, offY = 0.5f / a.height;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        float uu = (k & 1) ? u + offX : u - offX;
        float vv = (k & 2) ? v + offY : v - offY;
        v3 d = uv_to_direction(cam, uu, vv);
        dirs[(3 * k + 0) * STRIDE] = d.x; dirs[(3 * k + 1) * STRIDE] = d.y; dirs[(3 * k + 2) * STRIDE] = d.z;
    }
}

// One AntiAliasing call (renderer.h:138-201) and the clamps the pixel driver applies to its result (main.cpp:313-315).
// dirs: the pixel's four primary directions, word j at dirs[j * STRIDE].
struct SampleOut { v3 color, emission; float strength; };

template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE, bool COUNT, bool PRUNE, int MAXSP, int MAXPEND, class RNG, int BATCH, int STRIDE = BLOCK>
__device__ __forceinline__ SampleOut render_sample(const KernelArgs& a, const SceneLds& sc, RNG& rng, const float* dirs,
                                                   v3 origin, Counters& cnt)
{
    v3 res_color = mk(0, 0, 0), res_em = mk(0, 0, 0);
    float res_st = 0.0f;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {                                            // renderer.h:167-195
        v3 dir = mk(dirs[(3 * k + 0) * STRIDE], dirs[(3 * k + 1) * STRIDE], dirs[(3 * k + 2) * STRIDE]);
        PathOut p = trace_path<EMISSION, MIXED, HAS_NONSPHERE, COUNT, PRUNE, MAXSP, MAXPEND, RNG, BATCH>(sc, rng, origin, dir, a.max_depth, cnt, a.fault);
        // renderer.h:169: color / vec3(maxDepth); x / 1.0f == x, and most paths end at depth 1
        v3 pc = p.color;
        if (p.maxDepth != 1) { float md = (float)p.maxDepth; pc = mk(pc.x / md, pc.y / md, pc.z / md); }
        res_color = res_color + pc;
        if (EMISSION) {
            res_em = res_em + p.emission;
            res_st = (k == 0) ? p.strength : gmax(res_st, p.strength);
        }
#if RT_COOP_UNIFORM
        make_uniform<BATCH == RT_COOP>(res_color);
        if (EMISSION) { make_uniform<BATCH == RT_COOP>(res_em); make_uniform<BATCH == RT_COOP>(res_st); }
#endif
    }
    SampleOut o;
    res_color = res_color * 0.25f;
    o.color = mk(clamp01(res_color.x), clamp01(res_color.y), clamp01(res_color.z));               // main.cpp:313
    o.emission = mk(0, 0, 0); o.strength = 0.0f;
    if (EMISSION) {
        res_em = res_em * 0.25f;
        o.emission = mk(clamp01(res_em.x), clamp01(res_em.y), clamp01(res_em.z));
        o.strength = res_st;
    }
    return o;
}

// pixel::Set / emissionPixel::Set of the sample sums (main.cpp:318-319 -> structs.h:12-17,39-43)
template <bool EMISSION>
__device__ __forceinline__ void store_pixel(const KernelArgs& a, size_t o, v3 acc_color, v3 acc_em, float acc_st)
{
    const float n = (float)a.samples;
    if (a.rgb) {
        a.rgb[o * 3 + 0] = quantise(acc_color.x / n);
        a.rgb[o * 3 + 1] = quantise(acc_color.y / n);
        a.rgb[o * 3 + 2] = quantise(acc_color.z / n);
    }
    if (EMISSION && a.emission)
        a.emission[o] = make_float4(acc_em.x / n, acc_em.y / n, acc_em.z / n, acc_st / n);
}

// The `pixel` bytes of a wave of the direct kernel (8 x 8 pixels, lane = x + 8 y).  Four neighbours in a row own 12
// consecutive bytes: where all four lanes are alive (none of the pixels is the listed kernels', which write theirs at
// their own time, none is off the frame) and the 12 bytes start on a 4-byte boundary, three of the lanes store one dword
// each -- a row of the block is then two runs of 12 bytes in three stores instead of 24 single bytes.  Otherwise every
// lane stores its own three bytes.  `mine`: this pixel's first byte.
__device__ __forceinline__ void store_rgb_quad(unsigned char* mine, unsigned char b0, unsigned char b1, unsigned char b2)
{
    const int lane = (int)(threadIdx.x & 63), i = lane & 3;
    const unsigned int p = (unsigned int)b0 | ((unsigned int)b1 << 8) | ((unsigned int)b2 << 16);
    const unsigned long long alive = __ballot(true);
    const bool whole = ((alive >> (lane & 60)) & 0xFull) == 0xFull && ((size_t)(mine - 3 * i) & 3) == 0;
    // the right-hand neighbour's bytes: quad_perm [1, 2, 3, 3] (read only where the whole quad is alive)
    const unsigned int q = (unsigned int)__builtin_amdgcn_update_dpp((int)p, (int)p, 0xF9, 0xF, 0xF, false);
    if (whole) {
        if (i < 3) *(unsigned int*)(mine + i) = (p >> (8 * i)) | (q << (24 - 8 * i));     // bytes 4 i .. 4 i + 3 of the twelve
    } else {
        mine[0] = b0; mine[1] = b1; mine[2] = b2;
    }
}

// Hands a long pixel over to heavy_kernel (DESIGN.md "sample-parallel pixels") after its first RT_HEAVY_PROBE
// samples.  The pixel gets a stretch of the stream pool that starts with the generator's 31 state words in the order
// they were written (oldest first: the stream continues them, rt_device.hpp RngStreamT), and a record: pixel, the draw
// count to guess per sample, the sums over the samples done, the next sample, where its stream lies and how many draws
// it holds, the draw index the next sample starts at, draws written so far.  Returns false when the pool or the record
// table is full (the caller then carries on itself).
template <class RNG>
__device__ __forceinline__ bool heavy_handover(const KernelArgs& a, RNG& rng, unsigned int pixel, unsigned int guess,
                                               v3 acc_color, v3 acc_em, float acc_st)
{
    // room for twice the guessed number of draws of the remaining samples, in whole turns of 31
    unsigned int cap = (unsigned int)(a.samples - RT_HEAVY_PROBE) * guess * 2u + 124u;
    cap -= cap % 31u;
    const unsigned int need = 31u + cap;
    const unsigned long long base = atomicAdd(a.heavy_pool_used, (unsigned long long)need);
    if (base + need > (unsigned long long)a.heavy_pool_words) return false;
    const unsigned int h = atomicAdd(a.heavy_count, 1u);
    if (h >= (unsigned int)a.heavy_cap) return false;
    unsigned int* st = a.heavy_pool + base;
    int f = rng.f;
    for (int k = 0; k < 31; ++k) { st[k] = rng.r[f * RNG::stride]; f = RNG::wrap(f + 1); }
    unsigned int* rec = a.heavy_rec + (size_t)h * RT_HEAVY_REC_WORDS;
    rec[0] = pixel; rec[1] = guess;
    rec[2] = __float_as_uint(acc_color.x); rec[3] = __float_as_uint(acc_color.y); rec[4] = __float_as_uint(acc_color.z);
    rec[5] = __float_as_uint(acc_em.x); rec[6] = __float_as_uint(acc_em.y); rec[7] = __float_as_uint(acc_em.z);
    rec[8] = __float_as_uint(acc_st);
    rec[9] = RT_HEAVY_PROBE; rec[10] = (unsigned int)base; rec[11] = cap; rec[12] = 0u; rec[13] = 0u;
    return true;
}

// The per-pixel driver (main.cpp:298-319 == main.cu:38-67) over AntiAliasing (renderer.h:138-201), for pixel
// (x, y) of the frame = column x of packed output row r.
// LDS_RNG: the generator's 31 words live in LDS at rng_lds (slot-major) instead of in scratch.  HEAVY: pixels whose
// first sample shows a long path are handed to heavy_kernel.
template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE, bool COUNT, bool PRUNE, int MAXSP, int MAXPEND, bool LDS_RNG, int BATCH, bool HEAVY, int STRIDE = BLOCK>
__device__ __forceinline__ void render_pixel(const KernelArgs& a, const SceneLds& sc, float* dirs, unsigned int* rng_lds,
                                             int x, int y, int r, Counters& cnt, bool writer = true, bool dirs_ready = false)
{
    const v3 origin = mk(a.cam[0], a.cam[1], a.cam[2]);
    typedef RngT<LDS_RNG ? STRIDE : 1> RNG;
    unsigned int rng_words[LDS_RNG ? 1 : 31];
    RNG rng;
    rng.init(LDS_RNG ? rng_lds : rng_words, a.seed_base + (unsigned int)(x + y * a.width));   // rand.h:7, main.cpp:304

    // AntiAliasing is called with the same (u, v) for every sample (main.cpp:312), so the four primary directions
    // are per-pixel constants: computed once, parked in LDS (12 words per thread, word j of thread t at [j*BLOCK + t]).
    if (!dirs_ready) primary_directions<STRIDE>(a, dirs, x, y);

    const size_t o = (size_t)x + (size_t)r * a.width;
    v3 acc_color = mk(0, 0, 0), acc_em = mk(0, 0, 0);
    float acc_st = 0.0f;
    unsigned int c0 = 0, c1 = 0, c2 = 0;          // draws of the first samples
    for (int s = 0; s < a.samples; ++s) {
        const unsigned int before = rng.ndraws;
        SampleOut so = render_sample<EMISSION, MIXED, HAS_NONSPHERE, COUNT, PRUNE, MAXSP, MAXPEND, RNG, BATCH, STRIDE>(a, sc, rng, dirs, origin, cnt);
        acc_color = acc_color + so.color;
        if (EMISSION) { acc_em = acc_em + so.emission; acc_st += so.strength; }
#if RT_COOP_UNIFORM
        make_uniform<BATCH == RT_COOP>(acc_color);
        if (EMISSION) { make_uniform<BATCH == RT_COOP>(acc_em); make_uniform<BATCH == RT_COOP>(acc_st); }
        make_uniform<BATCH == RT_COOP>(rng.f); make_uniform<BATCH == RT_COOP>(rng.ndraws);
#endif
        if (HEAVY && a.heavy_cap > 0 && s < RT_HEAVY_PROBE) {
            // The first four samples tell whether the pixel's draw count repeats (a path whose shape does not hang on
            // the numbers drawn, mostly): at least RT_HEAVY_AGREE of them drew the same count, which becomes the guess.
            // Two launches of this kernel share the list: one owns the long pixels of that kind and hands them over,
            // the other owns the rest; both learn which is which from the same first samples.
            const unsigned int c3 = rng.ndraws - before;
            if (s == 0) c0 = c3;
            if (s == 1) c1 = c3;
            if (s == 2) c2 = c3;
            if (s == RT_HEAVY_PROBE - 1) {
                unsigned int guess = c3;
                int agree = 1 + (c0 == c3) + (c1 == c3) + (c2 == c3);
                if (agree < 3) { guess = c0; agree = (c1 == c0) + (c2 == c0) + 1; }      // three of four: c0 == c1 == c2
                // (the test hook, threshold 0, also hands over pixels whose counts differ: the re-guess path)
                const bool eligible = guess > 0 && (agree >= RT_HEAVY_AGREE || a.heavy_min_work == 0)
                    && (unsigned long long)guess * (unsigned int)(a.samples - RT_HEAVY_PROBE) >= (unsigned long long)a.heavy_min_work;
                if (a.listed_mode == RT_MODE_REST) { if (eligible) return; }
                else if (eligible) { if (heavy_handover(a, rng, (unsigned int)o, guess, acc_color, acc_em, acc_st)) return; }
                else if (a.listed_mode == RT_MODE_STEADY) return;
            }
        }
    }
    if (writer) store_pixel<EMISSION>(a, o, acc_color, acc_em, acc_st);
}

template <bool HAS_NONSPHERE, int BATCH = 1>
__device__ __forceinline__ int primary_closest(const SceneLds& sc, v3 ro, v3 rd, float& dist)
{
    Counters none = { 0, 0, 0, 0, 0 };
    if (!HAS_NONSPHERE) return closest_hit_primary<false, BATCH>(sc, rd, dist, none);
    return closest_hit<HAS_NONSPHERE, false, BATCH>(sc, ro, rd, dist, none);
}

// Large scenes: the hittables some primary ray of the wave hits, found from the clusters some ray of the wave may hit
// (rt_device.hpp cluster_mask: a skipped cluster holds only tests that certainly return false), in slot order -- the
// caller sorts the list back into scene order (wavelist_sort) before closest-hit queries walk it.
template <bool HAS_NONSPHERE>
__device__ __forceinline__ void wave_primary_candidates(const KernelArgs& a, const SceneLds& sc, const float* dirs, v3 origin, bool mine, WaveList& prim)
{
    typedef const float __attribute__((address_space(4)))* kptrf;
    typedef const unsigned int __attribute__((address_space(4)))* kptru;
    const kptrf kcl = (kptrf)a.clusters;
    const kptru korder = (kptru)a.order;
    for (int c = 0; c < a.n_clusters; ++c) {
        const float4 cb = make_float4(kcl[8 * c + 0], kcl[8 * c + 1], kcl[8 * c + 2], kcl[8 * c + 3]);
        const float4 ch = make_float4(kcl[8 * c + 4], kcl[8 * c + 5], kcl[8 * c + 6], kcl[8 * c + 7]);
        bool may = false;
#pragma unroll 1
        for (int k = 0; k < 4; ++k)
            may = may || cluster_may_hit(cb, ch, origin, mk(dirs[(3 * k + 0) * BLOCK], dirs[(3 * k + 1) * BLOCK], dirs[(3 * k + 2) * BLOCK]));
        if (__ballot(may && mine) == 0ull) continue;
        const int end = 64 * c + 64 < sc.n_hit ? 64 * c + 64 : sc.n_hit;
        for (int slot = 64 * c; slot < end; ++slot) {
            const int i = (int)korder[slot];
            float4 r0 = sc.urec(2 * i), r1 = sc.urec(2 * i + 1);
            bool any = false;
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                v3 rd = mk(dirs[(3 * k + 0) * BLOCK], dirs[(3 * k + 1) * BLOCK], dirs[(3 * k + 2) * BLOCK]);
                float fa = 4.0f * dot(rd, rd);
                v3 inv = mk(0, 0, 0);
                if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
                float t;
                if (!HAS_NONSPHERE) {
                    f32x4 q = sc.kprim[i];
                    any = any || sphere_hit_oc(mk(q.x, q.y, q.z), q.w, rd, fa, t);
                } else {
                    any = any || hittable_hit<HAS_NONSPHERE>(r0, r1, origin, rd, fa, inv, t);
                }
            }
            wavelist_offer(prim, i, any && mine);
        }
    }
}

// classify_kernel: key[pixel] = 0 when none of the pixel's four primary rays hits a material that bounces
// (reflection > 0 or refraction > 0, renderer.h:74,93; bounces need depth 1 <= max_depth, renderer.h:68), else
// 1 + the first such material.  Primary rays are the same for every sample (same u, v; no draw happens before the
// first hit), so the key holds for the whole pixel.  hist[key] counts pixels per key.
// BIG (scenes of the cooperative route's size): hittables not staged, queries over the wave's candidate list.
template <bool HAS_NONSPHERE, bool BIG = false>
__global__ __launch_bounds__(BLOCK) void classify_kernel(KernelArgs a)
{
    extern __shared__ float4 lds[];
    __shared__ unsigned int l_hist[RT_MAX_KEYS];
    SceneLds sc;
    float* lds_free;
    stage_scene<!BIG>(a, lds, sc, lds_free);
    for (int i = threadIdx.x; i < RT_MAX_KEYS; i += BLOCK) l_hist[i] = 0;
    __syncthreads();

    const PixelId p = tiled_pixel(a);
    float* dirs = lds_free + threadIdx.x;
    const v3 origin = mk(a.cam[0], a.cam[1], a.cam[2]);
    WaveList prim;
    wavelist_init(prim);
    prim.full = true;                                    // (no list: every query over the scene)
    if (BIG) {                                           // (whole waves: lanes off the frame come along)
        primary_directions(a, dirs, p.x, p.y);
        prim.full = false;
        wave_primary_candidates<HAS_NONSPHERE>(a, sc, dirs, origin, p.active, prim);
        if (!prim.full) wavelist_sort(prim);
    }
    if (p.active) {
        if (!BIG) primary_directions(a, dirs, p.x, p.y);
        int key = 0;
#pragma unroll 1
        for (int k = 0; k < 4 && key == 0; ++k) {
            v3 rd = mk(dirs[(3 * k + 0) * BLOCK], dirs[(3 * k + 1) * BLOCK], dirs[(3 * k + 2) * BLOCK]);
            float dist;
            int bi = (BIG && !prim.full) ? closest_hit_listed<HAS_NONSPHERE, true>(sc, prim, origin, rd, dist)
                                         : primary_closest<HAS_NONSPHERE>(sc, origin, rd, dist);
            if (bi >= 0) {
                int m = rec_material(sc.hrec[2 * bi + 1]);
                if (sc.mats[10 * m + 4] > 0 || sc.mats[10 * m + 5] > 0) key = m + 1;
            }
        }
        a.keys[(size_t)p.x + (size_t)p.r * a.width] = (unsigned short)key;
        if (key) atomicAdd(&l_hist[key], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < RT_MAX_KEYS; i += BLOCK)
        if (l_hist[i]) atomicAdd(&a.hist[i], l_hist[i]);
}

// Exclusive scan of hist[1..] into cursor[] (one workgroup; at most 257 keys); cursor[0] receives the total.
__global__ void scan_kernel(const unsigned int* __restrict__ hist, unsigned int* __restrict__ cursor)
{
    if (threadIdx.x == 0) {
        unsigned int s = 0;
        for (int k = 1; k < RT_MAX_KEYS; ++k) { cursor[k] = s; s += hist[k]; }
        cursor[0] = s;
    }
}

// list[cursor[key]++] = pixel, for every pixel with key != 0: the list comes out grouped by material, so a wave of
// render_kernel works on pixels whose paths look alike.
__global__ __launch_bounds__(BLOCK) void scatter_kernel(const unsigned short* __restrict__ keys, unsigned int n,
                                                        unsigned int* __restrict__ cursor, unsigned int* __restrict__ list)
{
    // ranks within the workgroup's chunk come from LDS counters; one global reservation per key and workgroup
    __shared__ unsigned int l_cnt[RT_MAX_KEYS], l_base[RT_MAX_KEYS];
    for (int i = threadIdx.x; i < RT_MAX_KEYS; i += BLOCK) l_cnt[i] = 0;
    __syncthreads();
    const unsigned int chunk = blockIdx.x * (BLOCK * 8);
    unsigned int k[8], rank[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned int i = chunk + j * BLOCK + threadIdx.x;
        k[j] = i < n ? keys[i] : 0u;
        rank[j] = k[j] ? atomicAdd(&l_cnt[k[j]], 1u) : 0u;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < RT_MAX_KEYS; i += BLOCK)
        if (l_cnt[i]) l_base[i] = atomicAdd(&cursor[i], l_cnt[i]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (k[j]) list[l_base[k[j]] + rank[j]] = chunk + j * BLOCK + threadIdx.x;
}

// direct_kernel: the per-pixel driver for pixels whose primary rays never bounce.  TraceRay then is: closest hit ->
// sky (renderer.h:40-50) or one shading (renderer.h:55-66) with multiplier 1 and maxDepth = depth = 1
// (renderer.h:128-132), so color / vec3(maxDepth) is color / 1 and no random number is drawn.
// BIG (scenes of the cooperative route's size): the hittables are not staged into LDS (more waves per CU), and the
// candidate lists are built from the clusters some ray of the wave may hit (rt_device.hpp cluster_mask) instead of from
// the whole scene.
template <bool EMISSION, bool HAS_NONSPHERE, bool CULL = (RT_DIRECT_CULL != 0), bool BIG = false>
__global__ __launch_bounds__(BLOCK, (HAS_NONSPHERE || EMISSION) ? RT_DIRECT_NS_WAVES : RT_DIRECT_WAVES) void direct_kernel(KernelArgs a)
{
    extern __shared__ float4 lds[];
    SceneLds sc;
    float* lds_free;
    stage_scene<!BIG>(a, lds, sc, lds_free);

    const PixelId p = tiled_pixel(a);
    const size_t o = (size_t)p.x + (size_t)p.r * a.width;
    // lanes off the frame, and lanes whose pixel render_kernel has, stay until the candidate lists are built
    const bool mine = p.active && !(a.keys && a.keys[o] != 0);

    float* dirs = lds_free + threadIdx.x;
    primary_directions(a, dirs, p.x, p.y);
    const v3 origin = mk(a.cam[0], a.cam[1], a.cam[2]);
    Counters none = { 0, 0, 0, 0, 0 };

    // Candidate lists of this wave (rt_device.hpp "wave-level candidate lists"): hittables some primary ray of the wave
    // hits, and hittables that occlude some light from some primary hit point.  One pass over the scene, then every
    // sample's queries run over the lists.
    WaveList prim, occl;
    wavelist_init(prim); wavelist_init(occl);
    if (CULL && BIG) {
        if (__ballot(mine) == 0ull) return;
        typedef const float __attribute__((address_space(4)))* kptrf;
        typedef const unsigned int __attribute__((address_space(4)))* kptru;
        const kptrf kcl = (kptrf)a.clusters;
        const kptru korder = (kptru)a.order;
        auto box = [&](int c, float4& cb, float4& ch) {
            cb = make_float4(kcl[8 * c + 0], kcl[8 * c + 1], kcl[8 * c + 2], kcl[8 * c + 3]);
            ch = make_float4(kcl[8 * c + 4], kcl[8 * c + 5], kcl[8 * c + 6], kcl[8 * c + 7]);
        };
        wave_primary_candidates<HAS_NONSPHERE>(a, sc, dirs, origin, mine, prim);
        if (!prim.full) {
            wavelist_sort(prim);
            WaveBits listed; listed.bits = 0ull;
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                v3 rd = mk(dirs[(3 * k + 0) * BLOCK], dirs[(3 * k + 1) * BLOCK], dirs[(3 * k + 2) * BLOCK]);
                float dist;
                int bi = closest_hit_listed<HAS_NONSPHERE, true>(sc, prim, origin, rd, dist);
                const bool hit = mine && bi >= 0;
                if (__ballot(hit) == 0ull) continue;
                v3 pos = mk(0, 0, 0);
                if (hit) { Hit h = hit_record<!HAS_NONSPHERE>(sc, bi, origin, rd, dist); pos = h.position + h.normal * 0.005f; }
                for (int j = 0; j < sc.n_lights; ++j) {
                    int type; float ldist;
                    v3 l = light_direction(sc, j, pos, type, ldist);
                    float fa = 4.0f * dot(l, l);
                    v3 inv = mk(0, 0, 0);
                    if (HAS_NONSPHERE) inv = mk(rt_rcp(l.x), rt_rcp(l.y), rt_rcp(l.z));
                    for (int c = 0; c < a.n_clusters; ++c) {
                        float4 cb, ch;
                        box(c, cb, ch);
                        if (__ballot(hit && cluster_may_hit(cb, ch, pos, l)) == 0ull) continue;
                        const int end = 64 * c + 64 < sc.n_hit ? 64 * c + 64 : sc.n_hit;
                        for (int slot = 64 * c; slot < end; ++slot) {
                            const int i = (int)korder[slot];
                            if (wavebits_test(listed, i)) continue;                  // one list for all rays and lights
                            float4 r0 = sc.urec(2 * i), r1 = sc.urec(2 * i + 1);
                            bool o = hit && ((type == L_DIR) ? occluder_hit<HAS_NONSPHERE, false, false>(r0, r1, pos, l, fa, inv, 0.0f)
                                                             : occluder_hit<HAS_NONSPHERE, false, true>(r0, r1, pos, l, fa, inv, ldist));
                            if (__ballot(o) != 0ull) { wavelist_offer(occl, i, true); wavebits_set(listed, i); }
                        }
                    }
                }
            }
        }
    } else if (CULL) {
        for (int i = 0; i < sc.n_hit; ++i) {
            float4 r0 = sc.urec(2 * i), r1 = sc.urec(2 * i + 1);
            bool any = false;
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                v3 rd = mk(dirs[(3 * k + 0) * BLOCK], dirs[(3 * k + 1) * BLOCK], dirs[(3 * k + 2) * BLOCK]);
                float fa = 4.0f * dot(rd, rd);
                v3 inv = mk(0, 0, 0);
                if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
                float t;
                if (!HAS_NONSPHERE) {
                    // the same operands as closest_hit_primary (camera-relative terms)
                    f32x4 q = sc.kprim[i];
                    any = any || sphere_hit_oc(mk(q.x, q.y, q.z), q.w, rd, fa, t);
                } else {
                    any = any || hittable_hit<HAS_NONSPHERE>(r0, r1, origin, rd, fa, inv, t);
                }
            }
            wavelist_offer(prim, i, any && mine);
        }
        if (!prim.full) {
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                v3 rd = mk(dirs[(3 * k + 0) * BLOCK], dirs[(3 * k + 1) * BLOCK], dirs[(3 * k + 2) * BLOCK]);
                float dist;
                int bi = closest_hit_listed<HAS_NONSPHERE, true>(sc, prim, origin, rd, dist);
                const bool hit = mine && bi >= 0;
                v3 pos = mk(0, 0, 0);
                if (hit) { Hit h = hit_record<!HAS_NONSPHERE>(sc, bi, origin, rd, dist); pos = h.position + h.normal * 0.005f; }
                for (int j = 0; j < sc.n_lights; ++j) {
                    int type; float ldist;
                    v3 l = light_direction(sc, j, pos, type, ldist);
                    float fa = 4.0f * dot(l, l);
                    v3 inv = mk(0, 0, 0);
                    if (HAS_NONSPHERE) inv = mk(rt_rcp(l.x), rt_rcp(l.y), rt_rcp(l.z));
                    for (int i = 0; i < sc.n_hit; ++i) {
                        float4 r0 = sc.urec(2 * i), r1 = sc.urec(2 * i + 1);
                        bool o = hit && ((type == L_DIR) ? occluder_hit<HAS_NONSPHERE, false, false>(r0, r1, pos, l, fa, inv, 0.0f)
                                                         : occluder_hit<HAS_NONSPHERE, false, true>(r0, r1, pos, l, fa, inv, ldist));
                        // one list for all rays and lights: offer only what is not on it yet
                        bool known = false;
                        for (int e = 0; e < occl.n; ++e) known = known || __builtin_amdgcn_readlane(occl.items, e) == i;
                        if (!known) wavelist_offer(occl, i, o);
                    }
                }
            }
        }
    }
    const bool culled = CULL && !prim.full && !occl.full;
    if (!mine) return;

    // the four directions and 4 * dot(d, d) of each (`a` of sphere.h:17,24's quadratic) are the pixel's constants: in
    // registers for all samples
    v3 dk[4]; float fak[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        dk[k] = mk(dirs[(3 * k + 0) * BLOCK], dirs[(3 * k + 1) * BLOCK], dirs[(3 * k + 2) * BLOCK]);
        fak[k] = 4.0f * dot(dk[k], dk[k]);
    }

    v3 acc_color = mk(0, 0, 0), acc_em = mk(0, 0, 0);
    float acc_st = 0.0f;
    // MODE 0: every query over the scene; 1: over the wave's candidate lists; 2: both lists hold at most RT_SMALL_LIST hittables,
    // whose records stay in scalar registers
    auto samples = [&](auto mode) {
        constexpr int MODE = decltype(mode)::value;
        SmallList sp, so;
        if (MODE == 2) { smalllist_load<HAS_NONSPHERE, true>(sc, prim, sp); smalllist_load<HAS_NONSPHERE, false>(sc, occl, so); }
        for (int s = 0; s < a.samples; ++s) {
            v3 res_color = mk(0, 0, 0), res_em = mk(0, 0, 0);
            float res_st = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                    // renderer.h:167-195
                const v3 rd = dk[k];
                float dist;
                int bi = MODE == 2 ? closest_hit_small<HAS_NONSPHERE, true>(sp, origin, rd, fak[k], dist)
                       : MODE == 1 ? closest_hit_listed<HAS_NONSPHERE, true>(sc, prim, origin, rd, dist, fak[k])
                                   : primary_closest<HAS_NONSPHERE, RT_DIRECT_BATCH>(sc, origin, rd, dist);
                v3 color, em = mk(0, 0, 0);
                float st = 0.0f;
                if (bi < 0) {                                                // renderer.h:47-49
                    float aa = (rd.y + 1.0f) * 0.5f;
                    float w = 1.0f - aa;
                    color = mk(w + aa * 0.2f, w + aa * 0.3f, w + aa * 0.8f) * 1.0f;
                } else {
                    Hit h = hit_record<!HAS_NONSPHERE>(sc, bi, origin, rd, dist);
                    v3 offPosition = h.position + h.normal * 0.005f;         // renderer.h:55
                    float intensity = MODE == 2 ? light_intensity_small<HAS_NONSPHERE>(sc, so, offPosition, h.normal)
                                    : MODE == 1 ? light_intensity_listed<HAS_NONSPHERE>(sc, occl, offPosition, h.normal)
                                                : light_intensity<HAS_NONSPHERE, false, true, RT_DIRECT_BATCH>(sc, offPosition, h.normal, none);
                    const float* m = sc.mats + 10 * h.material;
                    color = (mk(m[0], m[1], m[2]) * intensity) * 1.0f;       // renderer.h:64
                    if (EMISSION) { em = mk(m[6], m[7], m[8]); st = m[9]; }
                }
                res_color = res_color + color;                               // renderer.h:169, maxDepth == 1
                if (EMISSION) {
                    res_em = res_em + em;
                    res_st = (k == 0) ? st : gmax(res_st, st);
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
    };
    if (culled && prim.n <= RT_SMALL_LIST && occl.n <= RT_SMALL_LIST) samples(std::integral_constant<int, 2>());
    else if (culled) samples(std::integral_constant<int, 1>());
    else        samples(std::integral_constant<int, 0>());
    const float n = (float)a.samples;
    // (the pixel's place in the output, worked out again from a thread index the compiler cannot connect with the first
    // one: kept across the sample loop it cost the sphere-only kernel, at its 80 registers, a 64-bit spill per lane --
    // 512 B per wave written to scratch and read back, 65 MB of the kernel's 120 MB of HBM writes per 4K frame; four
    // more registers are still spilled inside the sample loop, which the L2 absorbs: 55 MB written for 24.9 MB of
    // pixels.  At 5 waves per SIMD nothing spills -- 33 MB -- and a frame takes 4 % longer.)
    int tid2 = (int)threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const PixelId p2 = tiled_pixel(a, tid2);
    const size_t o2 = (size_t)p2.x + (size_t)p2.r * a.width;
    if (a.rgb)                                                               // main.cpp:318 -> structs.h:12-17
        store_rgb_quad(a.rgb + o2 * 3, quantise(acc_color.x / n), quantise(acc_color.y / n), quantise(acc_color.z / n));
    if (EMISSION && a.emission)                                              // main.cpp:319 -> structs.h:39-43
        a.emission[o2] = make_float4(acc_em.x / n, acc_em.y / n, acc_em.z / n, acc_st / n);
}

// The listed kernel's waves: candidate lists for the primary segment of every ray, as in the direct kernel (rt_device.hpp
// "wave-level candidate lists") -- a listed pixel's four primary rays and the shadow rays of their hit points are the same
// for every sample too, only what follows the first hit hangs on the numbers drawn.  One pass over the scene per wave
// (all 64 lanes alive: lanes past the end of the pixel list come along with nothing to offer), the hittables some primary
// ray of the wave hits / that occlude some light from some primary hit point in two short lists in LDS, in scene order;
// trace_path's first query and its shading then walk those instead of the scene.  Scenes of at most LISTED_LIST
// hittables (larger ones: no lists, every query over the scene).
constexpr int LISTED_LIST = 32;

template <bool HAS_NONSPHERE, int STRIDE = BLOCK>
__device__ __forceinline__ void listed_wave_lists(const KernelArgs& a, SceneLds& sc, const float* dirs, bool mine,
                                                  unsigned short* lp, unsigned short* lo)
{
    const int lane = (int)(threadIdx.x & 63);
    const v3 origin = mk(a.cam[0], a.cam[1], a.cam[2]);
    int nprim = 0, noccl = 0;
    for (int i = 0; i < sc.n_hit; ++i) {
        const float4 r0 = sc.urec(2 * i), r1 = sc.urec(2 * i + 1);
        bool any = false;
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const v3 rd = mk(dirs[(3 * k + 0) * STRIDE], dirs[(3 * k + 1) * STRIDE], dirs[(3 * k + 2) * STRIDE]);
            const float fa = 4.0f * dot(rd, rd);
            v3 inv = mk(0, 0, 0);
            if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
            float t;
            any = any || hittable_hit<HAS_NONSPHERE>(r0, r1, origin, rd, fa, inv, t);
        }
        if (__ballot(any && mine) != 0ull) { if (lane == 0) lp[nprim] = (unsigned short)i; ++nprim; }
    }
    __builtin_amdgcn_wave_barrier();
    unsigned int listed = 0u;                                  // hittables on the occluder list, one bit each
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const v3 rd = mk(dirs[(3 * k + 0) * STRIDE], dirs[(3 * k + 1) * STRIDE], dirs[(3 * k + 2) * STRIDE]);
        float dist;
        const int bi = closest_hit_ldslist<HAS_NONSPHERE>(sc, lp, nprim, origin, rd, dist);
        const bool hit = mine && bi >= 0;
        if (__ballot(hit) == 0ull) continue;
        v3 pos = mk(0, 0, 0);
        if (hit) { const Hit h = hit_record<!HAS_NONSPHERE>(sc, bi, origin, rd, dist); pos = h.position + h.normal * 0.005f; }
        for (int j = 0; j < sc.n_lights; ++j) {
            int type; float ldist;
            const v3 l = light_direction(sc, j, pos, type, ldist);
            const float fa = 4.0f * dot(l, l);
            v3 inv = mk(0, 0, 0);
            if (HAS_NONSPHERE) inv = mk(rt_rcp(l.x), rt_rcp(l.y), rt_rcp(l.z));
            for (int i = 0; i < sc.n_hit; ++i) {
                if ((listed >> i) & 1u) continue;
                const float4 r0 = sc.urec(2 * i), r1 = sc.urec(2 * i + 1);
                const bool o = hit && ((type == L_DIR) ? occluder_hit<HAS_NONSPHERE, false, false>(r0, r1, pos, l, fa, inv, 0.0f)
                                                       : occluder_hit<HAS_NONSPHERE, false, true>(r0, r1, pos, l, fa, inv, ldist));
                if (__ballot(o) != 0ull) listed |= 1u << i;
            }
        }
    }
    for (int i = 0; i < sc.n_hit; ++i)                          // in scene order
        if ((listed >> i) & 1u) { if (lane == 0) lo[noccl] = (unsigned short)i; ++noccl; }
    __builtin_amdgcn_wave_barrier();
    sc.cl_prim = lp; sc.cl_nprim = nprim; sc.cl_occl = lo; sc.cl_noccl = noccl;
}

// render_kernel: the general per-pixel driver.  LISTED: over the pixel list built by classify / scatter, at raised
// wave priority (these are the long pixels; the direct kernel's waves fill in around them).  Otherwise over 16x16
// tiles of every pixel (tallying builds).
#if defined(RT_WAVE_TIMES) && !defined(RT_RELAXED_TU)
// (experiments, -DRT_WAVE_TIMES: every wave of the listed kernel leaves when it started (100 MHz clock), how long it ran,
// HW_ID and XCC_ID | workgroup << 4; tools/wave_times.py reads them back and draws the occupancy of frames in flight)
#define RT_WAVE_TIMES_CAP (1 << 20)
__device__ uint4 g_wave_times[RT_WAVE_TIMES_CAP];
__device__ unsigned int g_wave_times_n;
extern "C" __attribute__((visibility("default"))) int rt_debug_wave_times(void* out, int cap)
{
    unsigned int n = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wave_times_n), 4) != hipSuccess) return -1;
    if (n > RT_WAVE_TIMES_CAP) n = RT_WAVE_TIMES_CAP;
    if ((int)n > cap) n = (unsigned int)cap;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), (size_t)n * 16) != hipSuccess) return -1;
    const unsigned int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wave_times_n), &zero, 4) != hipSuccess) return -1;
    return (int)n;
}
#endif

// WG: threads of a workgroup.  BLOCK, or RT_LISTED_SEQ_BLOCK for the listed kernel of frames in flight whose listed waves
// live long (rt_abi.cpp): the kernel holds 4 waves per SIMD (100 - 117 registers), a workgroup holds its slots until its
// last wave has finished, and with a wave per workgroup every finished wave's slot goes to the next frame's waves
// (glass 1080p, 16 frames in flight: 2.5 of the 4 slots hold a live wave with 256 threads; 27.5 -> 22.3 ms per frame).
// Short listed kernels keep BLOCK: the dispatcher packs single waves onto SIMDs that hold listed waves already, and a
// listed wave in company is slower.
template <bool LISTED, bool LDS_RNG, bool EMISSION, bool MIXED, bool HAS_NONSPHERE, bool COUNT, bool PRUNE, int MAXSP, int MAXPEND, int WG = BLOCK>
__global__ __launch_bounds__(WG, LISTED ? (WG == BLOCK ? RT_LISTED_WAVES : RT_LISTED_SEQ_WAVES) : RT_MIN_WAVES) void render_kernel(KernelArgs a)
{
    extern __shared__ float4 lds[];
#if defined(RT_WAVE_TIMES) && !defined(RT_RELAXED_TU)
    const unsigned long long wt0 = wall_clock64();
#endif
    SceneLds sc;
    float* lds_free;
    stage_scene(a, lds, sc, lds_free);
    if (LISTED) __builtin_amdgcn_s_setprio(3);

    const PixelId p = LISTED ? listed_pixel<WG>(a) : tiled_pixel(a);
    Counters cnt = { 0, 0, 0, 0, 0 };
    // the listed kernel has few, long waves: it trades LDS and registers for shorter dependent chains
    constexpr int BATCH = LISTED ? RT_LISTED_BATCH : 1;
    unsigned int* rng_lds = (unsigned int*)(lds_free + 12 * WG) + threadIdx.x;
    bool dirs_ready = false;
#if RT_LISTED_LISTS
    if (LISTED && !COUNT && PRUNE && sc.n_hit <= LISTED_LIST) {
        // (the lists sit behind the per-thread words: 2 x LISTED_LIST entries per wave; rt_launch_lds_bytes has the room)
        unsigned short* lp = (unsigned short*)(lds_free + 12 * WG + (LDS_RNG ? 32 * WG : 0)) + (threadIdx.x >> 6) * (2 * LISTED_LIST);
        primary_directions<WG>(a, lds_free + threadIdx.x, p.x, p.y);
        dirs_ready = true;
        listed_wave_lists<HAS_NONSPHERE, WG>(a, sc, lds_free + threadIdx.x, p.active, lp, lp + LISTED_LIST);
    }
#endif
    if (p.active)
        render_pixel<EMISSION, MIXED, HAS_NONSPHERE, COUNT, PRUNE, MAXSP, MAXPEND, LDS_RNG, BATCH, LISTED, WG>(a, sc, lds_free + threadIdx.x, rng_lds, p.x, p.y, p.r, cnt, true, dirs_ready);
    if (COUNT && p.active) flush_counters(a, cnt);
#if defined(RT_WAVE_TIMES) && !defined(RT_RELAXED_TU)
    if (LISTED && (threadIdx.x & 63) == 0) {
        const unsigned int i = atomicAdd(&g_wave_times_n, 1u);
        if (i < RT_WAVE_TIMES_CAP)
            g_wave_times[i] = make_uint4((unsigned int)wt0, (unsigned int)(wall_clock64() - wt0), __builtin_amdgcn_s_getreg(63492),
                                         (__builtin_amdgcn_s_getreg(63508) & 0xFu) | (blockIdx.x << 4));
    }
#endif
}

// heavy_kernel: one wave per handed-over pixel, one lane per sample (DESIGN.md "sample-parallel pixels").  A pixel's
// samples share one rand() sequence, so sample s can only start where sample s-1 stopped.  The wave writes the
// sequence out as a stream (lane 0, ahead of need) and every lane traces its sample from a *guessed* draw index:
// `guess` draws per sample.  Up to and including the first lane whose draw count is not `guess` the indices were
// right; those samples stand and are added, in sample order, to the pixel's sums.  The next sample's true index is then
// known, the later lanes re-guess from it and trace again -- a round costs one sample's path and settles at least one
// sample, more when the draw counts repeat.  A pixel whose rounds settle too few samples (the more pixels wait, the
// more is asked) is handed on to resume_kernel with its progress.
template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE, int MAXSP, int MAXPEND>
__device__ __forceinline__ void heavy_pixel(const KernelArgs& a, const SceneLds& sc, float* lds_free, unsigned int h, int need_rate)
{
    const int lane = threadIdx.x & 63;
    unsigned int* rec = a.heavy_rec + (size_t)h * RT_HEAVY_REC_WORDS;
    const unsigned int o = rec[0], guess = rec[1], cap = rec[11];
    unsigned int* stream = a.heavy_pool + rec[10];
    const int x = (int)(o % (unsigned int)a.width), r = (int)(o / (unsigned int)a.width);
    float* dirs = lds_free + threadIdx.x;
    primary_directions(a, dirs, x, frame_row(a, r));                         // every lane: the same pixel
    const v3 origin = mk(a.cam[0], a.cam[1], a.cam[2]);
    unsigned int* win = (unsigned int*)(lds_free + 12 * BLOCK) + threadIdx.x;

    v3 acc_color = mk(__uint_as_float(rec[2]), __uint_as_float(rec[3]), __uint_as_float(rec[4]));
    v3 acc_em = mk(__uint_as_float(rec[5]), __uint_as_float(rec[6]), __uint_as_float(rec[7]));
    float acc_st = __uint_as_float(rec[8]);
    int first = (int)rec[9];                  // samples below `first` stand
    unsigned int first_draw = rec[12];        // draw index sample `first` starts at
    unsigned int written = rec[13];           // draws in the stream so far
    const int first0 = first;
    unsigned int extra = 0;                   // more stream than the guesses ask for, after a sample ran out of it

    Counters cnt = { 0, 0, 0, 0, 0 };
    typedef RngStreamT<BLOCK> RNG;
    for (int round = 1;; ++round) {
        const bool mine = lane >= first && lane < a.samples;
        const unsigned int my_draw = first_draw + (unsigned int)(mine ? lane - first : 0) * guess;
        // the stream must cover every guessed sample and some slack for samples that draw more
        unsigned int want = first_draw + (unsigned int)(a.samples - first) * guess + 2u * guess + extra;
        if (want > cap) want = cap;
        if (written < want) {
            const int turns = (int)((want - written + 30u) / 31u);           // cap is a whole number of turns
            if (lane == 0) stream_turns(stream + 31 + written, turns);
            written += 31u * (unsigned int)turns;
            __threadfence();
        }
        RNG rng;
        rng.start(stream, win, my_draw, written, cap);
        SampleOut so;
        so.color = mk(0, 0, 0); so.emission = mk(0, 0, 0); so.strength = 0.0f;
        if (mine) so = render_sample<EMISSION, MIXED, HAS_NONSPHERE, false, true, MAXSP, MAXPEND, RNG, RT_LISTED_BATCH>(a, sc, rng, dirs, origin, cnt);

        // lanes up to the first one that is off started at their true index; that one stands too unless it ran out of
        // stream (its numbers were not the generator's)
        const unsigned long long off = __ballot(mine && (rng.overflow || rng.ndraws != guess));
        int last = a.samples - 1;
        bool starved = false;
        if (off != 0ull) {
            const int sb = __ffsll((long long)off) - 1;
            starved = __shfl((int)rng.overflow, sb) != 0;
            last = starved ? sb - 1 : sb;
        }
        for (int s = first; s <= last; ++s) {                                // main.cpp:313-315: sums in sample order
            acc_color = acc_color + mk(__shfl(so.color.x, s), __shfl(so.color.y, s), __shfl(so.color.z, s));
            if (EMISSION) {
                acc_em = acc_em + mk(__shfl(so.emission.x, s), __shfl(so.emission.y, s), __shfl(so.emission.z, s));
                acc_st += __shfl(so.strength, s);
            }
        }
        if (last >= first) first_draw = (unsigned int)__shfl((int)(my_draw + rng.ndraws), last);
        first = last + 1;
        if (first >= a.samples) break;

        bool give_up = round >= 3 && first - first0 < need_rate * round;
        if (starved) {
            if (written >= cap) give_up = true;          // the pixel draws more than its stream holds
            extra += 4u * guess;
        }
        if (give_up) {
            if (lane == 0) {
                rec[2] = __float_as_uint(acc_color.x); rec[3] = __float_as_uint(acc_color.y); rec[4] = __float_as_uint(acc_color.z);
                rec[5] = __float_as_uint(acc_em.x); rec[6] = __float_as_uint(acc_em.y); rec[7] = __float_as_uint(acc_em.z);
                rec[8] = __float_as_uint(acc_st);
                rec[9] = (unsigned int)first; rec[12] = first_draw; rec[13] = written;
                a.fallback_list[atomicAdd(a.fallback_count, 1u)] = h;
            }
            return;
        }
    }
    if (lane == 0) store_pixel<EMISSION>(a, (size_t)o, acc_color, acc_em, acc_st);
}

template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE, int MAXSP, int MAXPEND>
__global__ __launch_bounds__(BLOCK, RT_LISTED_WAVES) void heavy_kernel(KernelArgs a)
{
    extern __shared__ float4 lds[];
    unsigned int count = *a.heavy_count;
    if (count > (unsigned int)a.heavy_cap) count = (unsigned int)a.heavy_cap;
    if (blockIdx.x * (BLOCK / 64) >= count) return;                         // whole workgroup: before any barrier
    SceneLds sc;
    float* lds_free;
    stage_scene(a, lds, sc, lds_free);
    __builtin_amdgcn_s_setprio(3);
    // samples a round must settle on average: 2 while the pixels fit on the GPU at once, more when they queue
    int need_rate = (int)(count / 512u);
    need_rate = need_rate < 2 ? 2 : need_rate > 16 ? 16 : need_rate;
    // The grid is no larger than fits the GPU at once (rt_launch_heavy) -- the table's capacity is the length of the
    // pixel list, the count usually a small part of it -- and every wave takes records from a queue until they are
    // used up (a wave takes at most `count` of them: the bound is never reached, and no wave can spin).
    for (unsigned int turn = 0; turn <= count; ++turn) {
        unsigned int h = 0;
        if ((threadIdx.x & 63) == 0) h = atomicAdd(a.queue, 1u);
        h = (unsigned int)__builtin_amdgcn_readfirstlane((int)h);
        if (h >= count) break;
        heavy_pixel<EMISSION, MIXED, HAS_NONSPHERE, MAXSP, MAXPEND>(a, sc, lds_free, h, need_rate);
    }
}

// coop_kernel: large scenes.  One wave per listed pixel; all 64 lanes run the pixel's path code on the same values and
// share every scan of the scene (rt_device.hpp "cooperative scans"), so a bounce costs n/64 tests per lane instead of
// n tests in one lane's instruction stream.  Lane 0 writes the pixel.  What the lanes of a wave hold in common -- the
// pixel's primary directions, the generator's words -- exists once per wave in LDS (every lane reads and writes the
// same words with the same values), so a workgroup of COOP_WAVES waves needs little more than the staged scene and
// two of them fit a CU next to a 2049-hittable scene.
#ifndef RT_COOP_WAVES
#define RT_COOP_WAVES 12
#endif
#ifndef RT_COOP_MIN_WAVES
#define RT_COOP_MIN_WAVES 6
#endif
constexpr int COOP_WAVES = RT_COOP_WAVES;
constexpr int COOP_LIST = 64;     // entries of a wave's candidate list (more candidates: no list, every scan covers the scene)

template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE, int MAXSP, int MAXPEND>
__global__ __launch_bounds__(COOP_WAVES * 64, RT_COOP_MIN_WAVES) void coop_kernel(KernelArgs a)
{
    extern __shared__ float4 lds[];
    SceneLds sc;
    float* lds_free;
    stage_scene_sorted(a, lds, sc, lds_free);
    __builtin_amdgcn_s_setprio(3);

    const int wave = (int)(threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63);
    Counters cnt = { 0, 0, 0, 0, 0 };
    float* dirs = lds_free + wave;                                           // word j of wave w at [j * COOP_WAVES + w]
    unsigned int* rng_lds = (unsigned int*)(lds_free + 12 * COOP_WAVES) + wave;
    // Pixels differ in cost by orders of magnitude, and a workgroup holds its copy of the scene until its last wave is
    // done: every wave therefore takes pixels from a queue (a counter over the list) until the list is used up, and
    // the grid is only as large as fits the GPU at once (rt_launch_coop).
    // (a wave takes at most list_count pixels: the bound is never reached, and no wave can spin for ever)
    const unsigned int n_list = (unsigned int)a.list_count;
    for (unsigned int turn = 0; turn <= n_list; ++turn) {
    unsigned int i = 0;
    if (lane == 0) i = atomicAdd(a.queue, 1u);
    i = (unsigned int)__builtin_amdgcn_readfirstlane((int)i);
    if (i >= n_list) break;
    const unsigned int o = a.list[i];
    const int x = (int)(o % (unsigned int)a.width), r = (int)(o / (unsigned int)a.width);
    const int y = frame_row(a, r);
    sc.cl_prim = nullptr; sc.cl_nprim = -1; sc.cl_occl = nullptr; sc.cl_noccl = -1;

    // The pixel's primary rays and the shadow rays of their hit points are the same for every sample (the argument of
    // rt_device.hpp "wave-level candidate lists", for one pixel): one cooperative pass finds the hittables they hit /
    // that occlude them, and the primary segment of every sample runs over those two lists instead of the scene.
    unsigned short* lp = (unsigned short*)(lds_free + (12 + 32) * COOP_WAVES) + wave * (2 * COOP_LIST);
    unsigned short* lo = lp + COOP_LIST;
    // (behind the lists, 16-byte aligned: the waves' bounce stacks, when the launch made room for them)
    // (a.coop_lds_stack: colour frames per wave in LDS; MAXSP and more: the whole stack, 2 MAXSP + 3 MAXPEND frames)
    const int xframes = a.coop_lds_stack >= MAXSP ? 2 * MAXSP + 3 * MAXPEND : a.coop_lds_stack;
    sc.xstack = a.coop_lds_stack > 0 ? (float4*)(lds_free + (12 + 32) * COOP_WAVES + COOP_WAVES * COOP_LIST) + (size_t)wave * xframes : nullptr;
    sc.xstack_n = a.coop_lds_stack;
    const unsigned long long below = (1ull << lane) - 1ull;
    const v3 origin = mk(a.cam[0], a.cam[1], a.cam[2]);
    primary_directions<COOP_WAVES>(a, dirs, x, y);
    int nprim = 0, noccl = 0;
    bool fits = true;
    unsigned long long rounds = 0ull;                  // rounds that can hold a hit of one of the four directions
#pragma unroll 1
    for (int k = 0; k < 4; ++k)
        rounds |= cluster_mask(sc, origin, mk(dirs[(3 * k + 0) * COOP_WAVES], dirs[(3 * k + 1) * COOP_WAVES], dirs[(3 * k + 2) * COOP_WAVES]));
    while (rounds != 0ull && fits) {
        const int i = 64 * __builtin_ctzll(rounds) + lane;
        rounds &= rounds - 1ull;
        bool any = false;
        if (i < sc.n_hit) {
            const float4 r0 = sc.hrec[2 * i], r1 = sc.hrec[2 * i + 1];
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const v3 rd = mk(dirs[(3 * k + 0) * COOP_WAVES], dirs[(3 * k + 1) * COOP_WAVES], dirs[(3 * k + 2) * COOP_WAVES]);
                const float fa = 4.0f * dot(rd, rd);
                v3 inv = mk(0, 0, 0);
                if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
                float t;
                any = any || hittable_hit<HAS_NONSPHERE, false>(r0, r1, origin, rd, fa, inv, t);
            }
        }
        const unsigned long long m = __ballot(any);
        const int add = __popcll(m);
        if (nprim + add > COOP_LIST) fits = false;
        else if (any) lp[nprim + __popcll(m & below)] = (unsigned short)i;           // ascending slots
        nprim += add;
    }
    if (fits && nprim > 1) {
        // closest-hit queries walk this list with the sequential rule: it has to be in scene order, not slot order
        unsigned short slot = 0;
        int rank = 0;
        if (lane < nprim) {
            slot = lp[lane];
            const unsigned short mine = sc.orig[slot];
            for (int f = 0; f < nprim; ++f) rank += sc.orig[lp[f]] < mine;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < nprim) lp[rank] = slot;
        __builtin_amdgcn_wave_barrier();
    }
    if (fits) {
#pragma unroll 1
        for (int k = 0; k < 4 && fits; ++k) {
            const v3 rd = mk(dirs[(3 * k + 0) * COOP_WAVES], dirs[(3 * k + 1) * COOP_WAVES], dirs[(3 * k + 2) * COOP_WAVES]);
            float dist;
            const int bi = closest_hit_ldslist<HAS_NONSPHERE>(sc, lp, nprim, origin, rd, dist);
            if (bi < 0) continue;
            const Hit h = hit_record<!HAS_NONSPHERE>(sc, bi, origin, rd, dist);
            const v3 pos = h.position + h.normal * 0.005f;
            for (int j = 0; j < sc.n_lights && fits; ++j) {
                int type; float ldist;
                const v3 l = light_direction(sc, j, pos, type, ldist);
                const float fa = 4.0f * dot(l, l);
                v3 inv = mk(0, 0, 0);
                if (HAS_NONSPHERE) inv = mk(rt_rcp(l.x), rt_rcp(l.y), rt_rcp(l.z));
                unsigned long long lrounds = cluster_mask(sc, pos, l);
                while (lrounds != 0ull && fits) {
                    const int i = 64 * __builtin_ctzll(lrounds) + lane;
                    lrounds &= lrounds - 1ull;
                    bool o = false;
                    if (i < sc.n_hit) {
                        const float4 r0 = sc.hrec[2 * i], r1 = sc.hrec[2 * i + 1];
                        o = (type == L_DIR) ? occluder_hit<HAS_NONSPHERE, false, false, false>(r0, r1, pos, l, fa, inv, 0.0f)
                                            : occluder_hit<HAS_NONSPHERE, false, true, false>(r0, r1, pos, l, fa, inv, ldist);
                        for (int e = 0; o && e < noccl; ++e) o = lo[e] != (unsigned short)i;   // already listed
                    }
                    const unsigned long long m = __ballot(o);
                    const int add = __popcll(m);
                    if (noccl + add > COOP_LIST) fits = false;
                    else if (o) lo[noccl + __popcll(m & below)] = (unsigned short)i;
                    noccl += add;
                }
            }
        }
    }
    if (fits) { sc.cl_prim = lp; sc.cl_nprim = nprim; sc.cl_occl = lo; sc.cl_noccl = noccl; }

    render_pixel<EMISSION, MIXED, HAS_NONSPHERE, false, true, MAXSP, MAXPEND, true, RT_COOP, false, COOP_WAVES>(
        a, sc, dirs, rng_lds, x, y, r, cnt, lane == 0);
    }
}

// resume_kernel: the serial pixel loop for what heavy_kernel gave up on, one lane per pixel, from the sample its record
// names and the generator state found in the pixel's stream.
template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE, int MAXSP, int MAXPEND>
__global__ __launch_bounds__(BLOCK, RT_LISTED_WAVES) void resume_kernel(KernelArgs a)
{
    extern __shared__ float4 lds[];
    const unsigned int count = *a.fallback_count;
    if (blockIdx.x * BLOCK >= count) return;                                 // whole workgroup: before any barrier
    SceneLds sc;
    float* lds_free;
    stage_scene(a, lds, sc, lds_free);
    __builtin_amdgcn_s_setprio(3);

    const unsigned int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const unsigned int h = a.fallback_list[i];
    const unsigned int* rec = a.heavy_rec + (size_t)h * RT_HEAVY_REC_WORDS;
    const unsigned int o = rec[0];
    const int next = (int)rec[9];
    const int x = (int)(o % (unsigned int)a.width), r = (int)(o / (unsigned int)a.width);
    float* dirs = lds_free + threadIdx.x;
    primary_directions(a, dirs, x, frame_row(a, r));
    const v3 origin = mk(a.cam[0], a.cam[1], a.cam[2]);

    unsigned int* slots = (unsigned int*)(lds_free + 12 * BLOCK) + threadIdx.x;
    typedef RngT<BLOCK> RNG;
    RNG rng;
    rng.init(slots, 0u);
    rng.seeded = true;
    // the 31 words before the next sample's first draw, oldest first, are the generator's state (front slot 0)
    const unsigned int* e = a.heavy_pool + rec[10] + rec[12];
    for (int k = 0; k < 31; ++k) slots[k * BLOCK] = e[k];
    rng.f = 0;

    Counters cnt = { 0, 0, 0, 0, 0 };
    v3 acc_color = mk(__uint_as_float(rec[2]), __uint_as_float(rec[3]), __uint_as_float(rec[4]));
    v3 acc_em = mk(__uint_as_float(rec[5]), __uint_as_float(rec[6]), __uint_as_float(rec[7]));
    float acc_st = __uint_as_float(rec[8]);
    for (int s = next; s < a.samples; ++s) {
        SampleOut so = render_sample<EMISSION, MIXED, HAS_NONSPHERE, false, true, MAXSP, MAXPEND, RNG, RT_LISTED_BATCH>(a, sc, rng, dirs, origin, cnt);
        acc_color = acc_color + so.color;
        if (EMISSION) { acc_em = acc_em + so.emission; acc_st += so.strength; }
    }
    store_pixel<EMISSION>(a, (size_t)o, acc_color, acc_em, acc_st);
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

// Tile records for a streaming consumer: what the reference's Redis::SendImage pushes per tile (redis.h:27-36 as called
// from main.cpp:322-326): [int tile x][int tile y][tile*tile pixels], once with `pixel` (3 B) and once with
// `emissionPixel` (16 B) payloads.  rgb / emission: `rows` packed rows of the frame starting at a tile boundary.
// One thread per pixel of the covered tiles; record t of the chunk = tile (t % tiles_x, tile_y0 + t / tiles_x).
__global__ __launch_bounds__(BLOCK) void tile_pack_kernel(const unsigned char* __restrict__ rgb, const float4* __restrict__ emission,
                                                          int width, int tile, int tiles_x, int tile_y0, int n_tiles,
                                                          unsigned char* __restrict__ img_rec, unsigned char* __restrict__ em_rec)
{
    const int per_tile = tile * tile;
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (long long)n_tiles * per_tile) return;
    const int t = (int)(i / per_tile), q = (int)(i % per_tile);
    const int tx = t % tiles_x, ty = t / tiles_x;
    const int x = tx * tile + q % tile, r = ty * tile + q / tile;           // r: row within the chunk
    const size_t src = (size_t)x + (size_t)r * width;
    const size_t img_bytes = 8 + (size_t)per_tile * 3, em_bytes = 8 + (size_t)per_tile * 16;
    unsigned char* ir = img_rec + (size_t)t * img_bytes;
    unsigned char* er = em_rec + (size_t)t * em_bytes;
    if (q == 0) {
        // image records are 8 + 3*tile*tile bytes long: not a multiple of 4 for odd tiles, so the coordinates go in bytewise
        const unsigned int cx = (unsigned int)tx, cy = (unsigned int)(tile_y0 + ty);
        for (int b = 0; b < 4; ++b) { ir[b] = (unsigned char)(cx >> (8 * b)); ir[4 + b] = (unsigned char)(cy >> (8 * b)); }
        int* ec = (int*)er; ec[0] = (int)cx; ec[1] = (int)cy;
    }
    ir[8 + (size_t)q * 3 + 0] = rgb[src * 3 + 0];
    ir[8 + (size_t)q * 3 + 1] = rgb[src * 3 + 1];
    ir[8 + (size_t)q * 3 + 2] = rgb[src * 3 + 2];
    // emission records start 8 bytes into a 16-byte-aligned buffer: 8-byte stores
    const float4 e = emission[src];
    float2* ep = (float2*)(er + 8 + (size_t)q * 16);
    ep[0] = make_float2(e.x, e.y);
    ep[1] = make_float2(e.z, e.w);
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

// rt_sqrt / rt_rcp / rt_inv_sqrt (rt_device.hpp) against the compiler's correctly rounded sqrtf and division, over every
// float bit pattern: out[k] counts the inputs on which sequence k differs (both NaN counts as equal).
__device__ __forceinline__ bool same_float(float a, float b) { return (a != a && b != b) || __float_as_uint(a) == __float_as_uint(b); }

__global__ __launch_bounds__(BLOCK) void debug_exact_math_kernel(unsigned long long* out)
{
    unsigned int bad0 = 0, bad1 = 0, bad2 = 0;
    const unsigned long long step = (unsigned long long)gridDim.x * BLOCK;
    for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < (1ull << 32); i += step) {
        const float x = __uint_as_float((unsigned int)i);
        bad0 += !same_float(rt_sqrt(x), sqrtf(x));
        bad1 += !same_float(rt_rcp(x), 1.0f / x);
        bad2 += !same_float(rt_inv_sqrt(x), 1.0f / sqrtf(x));
    }
    if (bad0) atomicAdd(&out[0], (unsigned long long)bad0);
    if (bad1) atomicAdd(&out[1], (unsigned long long)bad1);
    if (bad2) atomicAdd(&out[2], (unsigned long long)bad2);
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
    SceneLds sc; sc.hrec = hrec; sc.lights = nullptr; sc.mats = nullptr; sc.n_hit = index + 1; sc.n_lights = 0; sc.khrec = (kptr4)hrec; sc.klights = nullptr; sc.kprim = nullptr; sc.orig = nullptr; sc.slot_of = nullptr; sc.clusters = nullptr; sc.n_clusters = 0; sc.xstack = nullptr; sc.xstack_n = 0; sc.cl_prim = nullptr; sc.cl_nprim = -1; sc.cl_occl = nullptr; sc.cl_noccl = -1;
    v3 ro = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), rd = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    float4 r0 = hrec[2 * index], r1 = hrec[2 * index + 1];
    float fa = 4.0f * dot(rd, rd);
    v3 inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
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

}  // namespace RT_NS

// ---- launchers ------------------------------------------------------------------------------------------------
using namespace RT_NS;

// kernel variants by (emission buffer, reflect-and-refract material, planes or cubes).  RT_DEV_FAST (experiments only):
// just the three that the bench workloads use, for a quick build.
#ifdef RT_DEV_FAST
#define RT_ALL_VARIANTS(X) X(false, false, false) X(false, false, true) X(false, true, true)
#else
#define RT_ALL_VARIANTS(X) X(false, false, false) X(false, false, true) X(false, true, false) X(false, true, true) \
                           X(true, false, false)  X(true, false, true)  X(true, true, false)  X(true, true, true)
#endif

template <class K>
static hipError_t launch_with_lds(K k, const KernelArgs& a, int blocks, size_t lds_bytes, hipStream_t stream, int threads = BLOCK)
{
    if (blocks <= 0) return hipSuccess;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds_bytes, stream, a);
    return hipGetLastError();
}

// The direct kernel's launcher.  This file is compiled twice: as it is, and through rt_kernels_relaxed.hip with relaxed
// float arithmetic (RT_RELAXED_TU), where it yields the direct kernel and this launcher under another name and nothing else.
#ifdef RT_RELAXED_TU
#define RT_LAUNCH_DIRECT rt_launch_direct_relaxed
#else
#define RT_LAUNCH_DIRECT rt_launch_direct
#endif
hipError_t RT_LAUNCH_DIRECT(const KernelArgs& a, const LaunchTraits& t, hipStream_t stream)
{
    const int tiles = a.tiles_x * a.tiles_y;
    if (RT_DIRECT_CULL && a.n_hit >= RT_DIRECT_BIG && a.clusters && a.order) {
        // large scenes: no hittables in LDS, candidate lists from the clusters
        const size_t lds = rt_launch_lds_bytes(0, a.n_lights, a.n_mat);
        if (t.emission) return t.nonsphere ? launch_with_lds(direct_kernel<true, true, true, true>, a, tiles, lds, stream) : launch_with_lds(direct_kernel<true, false, true, true>, a, tiles, lds, stream);
        return t.nonsphere ? launch_with_lds(direct_kernel<false, true, true, true>, a, tiles, lds, stream) : launch_with_lds(direct_kernel<false, false, true, true>, a, tiles, lds, stream);
    }
    const size_t lds = rt_launch_lds_bytes(a.n_hit, a.n_lights, a.n_mat);
    if (t.emission) return t.nonsphere ? launch_with_lds(direct_kernel<true, true>, a, tiles, lds, stream) : launch_with_lds(direct_kernel<true, false>, a, tiles, lds, stream);
    return t.nonsphere ? launch_with_lds(direct_kernel<false, true>, a, tiles, lds, stream) : launch_with_lds(direct_kernel<false, false>, a, tiles, lds, stream);
}

#ifndef RT_RELAXED_TU
size_t rt_launch_lds_bytes(int n_hit, int n_lights, int n_mat)
{
    size_t nm = (size_t)((10 * n_mat + 3) & ~3);
    // (+ 512 bytes: the listed kernel's per-wave candidate lists, 4 waves x 2 x 32 entries)
    return (size_t)n_hit * 32 + (size_t)n_lights * 16 + nm * 4 + (size_t)12 * BLOCK * 4 + 16 + 512;
}

// the listed render_kernel also keeps the generators there: 31 words per thread
static size_t listed_lds_bytes(int n_hit, int n_lights, int n_mat)
{
    return rt_launch_lds_bytes(n_hit, n_lights, n_mat) + (size_t)32 * BLOCK * 4;
}

template <bool LISTED, bool LDS_RNG, bool EMISSION, bool MIXED, bool HAS_NONSPHERE, int WG = BLOCK>
static hipError_t launch_general(const KernelArgs& a, bool deep, int blocks, size_t lds, hipStream_t s)
{
    if (deep) return launch_with_lds(render_kernel<LISTED, LDS_RNG, EMISSION, MIXED, HAS_NONSPHERE, false, true, RT_STACK_DEEP, RT_PEND_DEEP, WG>, a, blocks, lds, s, WG);
    return launch_with_lds(render_kernel<LISTED, LDS_RNG, EMISSION, MIXED, HAS_NONSPHERE, false, true, RT_STACK_SHALLOW, RT_PEND_SHALLOW, WG>, a, blocks, lds, s, WG);
}

// The general per-pixel kernel: over the pixel list (a.list / a.list_count) when t.listed, else over all tiles.
hipError_t rt_launch_render(const KernelArgs& a, const LaunchTraits& t, hipStream_t stream)
{
    int blocks = t.listed ? (a.list_count + BLOCK - 1) / BLOCK : a.tiles_x * a.tiles_y;
    size_t lds = rt_launch_lds_bytes(a.n_hit, a.n_lights, a.n_mat);
    // large scenes leave no room for the generators next to the staged records: they stay in scratch then
    const bool lds_rng = t.listed && RT_LISTED_LDS_RNG && listed_lds_bytes(a.n_hit, a.n_lights, a.n_mat) <= 160 * 1024 && !t.rng_scratch;
    if (lds_rng) lds = listed_lds_bytes(a.n_hit, a.n_lights, a.n_mat);
    // a wave per workgroup (render_kernel's WG; the per-thread words shrink with the workgroup)
    const bool wave_groups = t.listed && !lds_rng && t.wave_groups;
    if (wave_groups) {
        blocks = (a.list_count + RT_LISTED_SEQ_BLOCK - 1) / RT_LISTED_SEQ_BLOCK;
        lds -= (size_t)12 * (BLOCK - RT_LISTED_SEQ_BLOCK) * 4;
    }
    // the tally builds are the most general ones: emission, mixed, all primitive types, deep stack
    if (t.count && !t.prune) return launch_with_lds(render_kernel<false, false, true, true, true, true, false, RT_STACK_DEEP, RT_PEND_DEEP>, a, blocks, lds, stream);
    if (t.count)             return launch_with_lds(render_kernel<false, false, true, true, true, true, true, RT_STACK_DEEP, RT_PEND_DEEP>, a, blocks, lds, stream);
#define RT_DISPATCH(E, M, N) if (t.emission == E && t.mixed == M && t.nonsphere == N) \
        return !t.listed ? launch_general<false, false, E, M, N>(a, t.deep, blocks, lds, stream) \
             : lds_rng   ? launch_general<true, true, E, M, N>(a, t.deep, blocks, lds, stream)   \
             : wave_groups ? launch_general<true, false, E, M, N, RT_LISTED_SEQ_BLOCK>(a, t.deep, blocks, lds, stream) \
                         : launch_general<true, false, E, M, N>(a, t.deep, blocks, lds, stream);
    RT_ALL_VARIANTS(RT_DISPATCH)
#undef RT_DISPATCH
    return hipErrorInvalidValue;
}

// classify (keys + histogram) -> scan -> scatter (the pixel list).  a.keys / a.hist / cursor / list are device buffers.
hipError_t rt_launch_classify(const KernelArgs& a, const LaunchTraits& t, unsigned int* cursor, unsigned int* list, hipStream_t stream)
{
    const int tiles = a.tiles_x * a.tiles_y;
    const size_t lds = rt_launch_lds_bytes(a.n_hit, a.n_lights, a.n_mat);
    const bool big = RT_DIRECT_CULL && a.n_hit >= RT_DIRECT_BIG && a.clusters && a.order;
    hipError_t e = big ? (t.nonsphere ? launch_with_lds(classify_kernel<true, true>, a, tiles, rt_launch_lds_bytes(0, a.n_lights, a.n_mat), stream)
                                      : launch_with_lds(classify_kernel<false, true>, a, tiles, rt_launch_lds_bytes(0, a.n_lights, a.n_mat), stream))
                 : t.nonsphere ? launch_with_lds(classify_kernel<true>, a, tiles, lds, stream)
                               : launch_with_lds(classify_kernel<false>, a, tiles, lds, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(64), 0, stream, (const unsigned int*)a.hist, cursor);
    const unsigned int n = (unsigned int)a.width * (unsigned int)a.rows;
    int blocks = (int)((n + BLOCK * 8 - 1) / (BLOCK * 8));
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(BLOCK), 0, stream, (const unsigned short*)a.keys, n, cursor, list);
    return hipGetLastError();
}

bool rt_launch_heavy_fits(int n_hit, int n_lights, int n_mat)
{
    return listed_lds_bytes(n_hit, n_lights, n_mat) <= 160 * 1024;
}

// heavy_kernel: one wave per pixel handed over by the listed render_kernel; the grid covers the table, workgroups past
// the count the listed kernel left in a.heavy_count leave at once.  Then resume_kernel for what it gave up on.
hipError_t rt_launch_heavy(const KernelArgs& a, const LaunchTraits& t, hipStream_t stream)
{
    int blocks = (a.heavy_cap + BLOCK / 64 - 1) / (BLOCK / 64);
    const int rblocks = (a.heavy_cap + BLOCK - 1) / BLOCK;
    const size_t lds = listed_lds_bytes(a.n_hit, a.n_lights, a.n_mat);
    // heavy_kernel's waves take records from a queue: no more workgroups than are resident at once (44 KB of LDS each)
    int dev = 0, cus = 0;
    hipError_t q = hipGetDevice(&dev);
    if (q == hipSuccess) q = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (q != hipSuccess) return q;
    const int per_cu = (int)((size_t)160 * 1024 / (lds > 0 ? lds : 1));
    const int resident = (per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu) * (cus > 0 ? cus : 1);
    if (blocks > resident) blocks = resident;
#define RT_DISPATCH(E, M, N) if (t.emission == E && t.mixed == M && t.nonsphere == N) {                                  \
        hipError_t e = t.deep ? launch_with_lds(heavy_kernel<E, M, N, RT_STACK_DEEP, RT_PEND_DEEP>, a, blocks, lds, stream)       \
                              : launch_with_lds(heavy_kernel<E, M, N, RT_STACK_SHALLOW, RT_PEND_SHALLOW>, a, blocks, lds, stream); \
        if (e != hipSuccess) return e;                                                                                    \
        return t.deep ? launch_with_lds(resume_kernel<E, M, N, RT_STACK_DEEP, RT_PEND_DEEP>, a, rblocks, lds, stream)             \
                      : launch_with_lds(resume_kernel<E, M, N, RT_STACK_SHALLOW, RT_PEND_SHALLOW>, a, rblocks, lds, stream); }
    RT_ALL_VARIANTS(RT_DISPATCH)
#undef RT_DISPATCH
    return hipErrorInvalidValue;
}

static size_t coop_lds_bytes(int n_hit, int n_lights, int n_mat)
{
    // (+ the slot -> scene index table and the clusters' boxes of stage_scene_sorted)
    size_t nm = (size_t)((10 * n_mat + 3) & ~3);
    return (size_t)n_hit * 32 + (size_t)n_lights * 16 + nm * 4 + (((size_t)n_hit * 2 + 15) & ~(size_t)15) + (size_t)((n_hit + 63) / 64) * 32
         + (size_t)(12 + 32) * COOP_WAVES * 4 + (size_t)COOP_WAVES * 2 * COOP_LIST * 2 + 16;
}

bool rt_launch_coop_fits(int n_hit, int n_lights, int n_mat)
{
    return coop_lds_bytes(n_hit, n_lights, n_mat) <= 160 * 1024;
}

#ifdef RT_COOP_STATS
// (experiments, -DRT_COOP_STATS: the tallies of rt_device.hpp's RT_CSTAT after a frame's coop_kernel)
__global__ void coop_stats_kernel()
{
    printf("[coop stats] closest: %llu queries, rounds near %llu rest %llu (scanned of rest: %llu); any-hit: %llu queries, rounds listed %llu scanned %llu, occluded %llu; light loops: %llu; closest over a pixel's candidate list %llu\n",
           g_coop_stats[0], g_coop_stats[1], g_coop_stats[2], g_coop_stats[3], g_coop_stats[4], g_coop_stats[5], g_coop_stats[6], g_coop_stats[7], g_coop_stats[8], g_coop_stats[9]);
    for (int k = 0; k < 12; ++k) g_coop_stats[k] = 0ull;
}
#endif

template <class K>
static hipError_t launch_coop(K k, const KernelArgs& a, int blocks, size_t lds_bytes, hipStream_t stream)
{
    if (blocks <= 0) return hipSuccess;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    // as many workgroups as are resident at once: their waves take pixels from the queue until it is empty
    int per_cu = 0, dev = 0, cus = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k, COOP_WAVES * 64, lds_bytes);
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if (const char* dbg = getenv("RT_DEBUG_COOP")) {
        fprintf(stderr, "[rt] coop launch: lds %zu B, %d workgroups of %d waves per CU by the occupancy query, %d CUs\n", lds_bytes, per_cu, COOP_WAVES, cus);
        if (atoi(dbg) > 1) per_cu = atoi(dbg);
    }
    const int resident = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
    if (blocks > resident) blocks = resident;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(COOP_WAVES * 64), lds_bytes, stream, a);
#ifdef RT_COOP_STATS
    hipLaunchKernelGGL(coop_stats_kernel, dim3(1), dim3(1), 0, stream);
#endif
    return hipGetLastError();
}

// coop_kernel over the pixel list: one wave per pixel.
hipError_t rt_launch_coop(const KernelArgs& a0, const LaunchTraits& t, hipStream_t stream)
{
    KernelArgs a = a0;
    const int blocks = (a.list_count + COOP_WAVES - 1) / COOP_WAVES;
    size_t lds = coop_lds_bytes(a.n_hit, a.n_lights, a.n_mat);
    // The waves' bounce stacks go into LDS when two workgroups still fit a CU with them, or when the scene is small
    // (the route of long pixels of small scenes: few waves, each a long serial chain of pushes and pops).
    const int maxsp = t.deep ? RT_STACK_DEEP : RT_STACK_SHALLOW;
    const size_t stack = (size_t)COOP_WAVES * 16 * (t.deep ? 2 * RT_STACK_DEEP + 3 * RT_PEND_DEEP : 2 * RT_STACK_SHALLOW + 3 * RT_PEND_SHALLOW);
    a.coop_lds_stack = 0;
    // workgroups the register bound lets a CU hold, and the LDS each may take without lowering that
    constexpr int per_cu = (4 * RT_COOP_MIN_WAVES) / COOP_WAVES > 0 ? (4 * RT_COOP_MIN_WAVES) / COOP_WAVES : 1;
    const size_t share = (size_t)160 * 1024 / per_cu;
    if (lds + stack <= share || (a.n_hit < 256 && lds + stack <= 160 * 1024)) {
        a.coop_lds_stack = maxsp;
        lds += stack;
    } else if (lds <= share) {
        // large scenes: as many colour frames per wave as the workgroups of a CU leave room for (the scratch copy costs a
        // 64-byte write per push and four lines per pop: 430 GB of HBM traffic per C4 frame before this)
        const int frames = (int)((share - lds) / ((size_t)COOP_WAVES * 16));
        a.coop_lds_stack = frames >= maxsp ? maxsp - 1 : frames >= 4 ? frames : 0;
        lds += (size_t)COOP_WAVES * 16 * (size_t)a.coop_lds_stack;
    }
#define RT_DISPATCH(E, M, N) if (t.emission == E && t.mixed == M && t.nonsphere == N)                                \
        return t.deep ? launch_coop(coop_kernel<E, M, N, RT_STACK_DEEP, RT_PEND_DEEP>, a, blocks, lds, stream)       \
                      : launch_coop(coop_kernel<E, M, N, RT_STACK_SHALLOW, RT_PEND_SHALLOW>, a, blocks, lds, stream);
    RT_ALL_VARIANTS(RT_DISPATCH)
#undef RT_DISPATCH
    return hipErrorInvalidValue;
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

hipError_t rt_launch_tile_pack(const void* rgb, const void* emission, int width, int tile, int tiles_x, int tile_y0, int n_tiles,
                               void* img_rec, void* em_rec, hipStream_t stream)
{
    const long long n = (long long)n_tiles * tile * tile;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_pack_kernel, dim3((unsigned int)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream,
                       (const unsigned char*)rgb, (const float4*)emission, width, tile, tiles_x, tile_y0, n_tiles,
                       (unsigned char*)img_rec, (unsigned char*)em_rec);
    return hipGetLastError();
}

hipError_t rt_launch_debug_rand(const unsigned int* seeds, int n_seeds, int n_draws, int* out, hipStream_t stream)
{
    hipLaunchKernelGGL(debug_rand_kernel, dim3((n_seeds + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, seeds, n_seeds, n_draws, out);
    return hipGetLastError();
}

hipError_t rt_launch_debug_exact_math(unsigned long long* out3, hipStream_t stream)
{
    hipLaunchKernelGGL(debug_exact_math_kernel, dim3(256 * 16), dim3(BLOCK), 0, stream, out3);
    return hipGetLastError();
}

hipError_t rt_launch_debug_quantise(const float* in, int n, unsigned char* out, hipStream_t stream)
{
    hipLaunchKernelGGL(debug_quantise_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, in, n, out);
    return hipGetLastError();
}

// out[i] = 1 if the large-scene kernels would visit hittable `index`'s cluster for ray i (rt_device.hpp cluster_mask)
__global__ void debug_cull_kernel(const float* clusters, const unsigned int* slot_of, int index, const float* rays, int n, unsigned char* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(slot_of[index] / 64u);
    const float4 cb = make_float4(clusters[8 * c + 0], clusters[8 * c + 1], clusters[8 * c + 2], clusters[8 * c + 3]);
    const float4 ch = make_float4(clusters[8 * c + 4], clusters[8 * c + 5], clusters[8 * c + 6], clusters[8 * c + 7]);
    v3 ro = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), rd = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    out[i] = cluster_may_hit(cb, ch, ro, rd) ? 1 : 0;
}

// enter[i] = where ray i enters the grown box of hittable `index`'s cluster as the distance rule of the cooperative
// scans sees it (rt_device.hpp cluster_rounds); -FLT_MAX where the rule does not apply to the ray or the box
__global__ void debug_cull_enter_kernel(const float* clusters, const unsigned int* slot_of, int index, const float* rays, int n, float* enter)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(slot_of[index] / 64u);
    const float4 cb = make_float4(clusters[8 * c + 0], clusters[8 * c + 1], clusters[8 * c + 2], clusters[8 * c + 3]);
    const float4 ch = make_float4(clusters[8 * c + 4], clusters[8 * c + 5], clusters[8 * c + 6], clusters[8 * c + 7]);
    v3 ro = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), rd = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    bool inside;
    enter[i] = ray_by_dist(rd) ? cluster_enter(cb, ch, ro, rd, inside) : -3.402823466e+38f;
}

hipError_t rt_launch_debug_cull_enter(const float* clusters, const unsigned int* slot_of, int index, const float* rays, int n, float* enter, hipStream_t stream)
{
    hipLaunchKernelGGL(debug_cull_enter_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, clusters, slot_of, index, rays, n, enter);
    return hipGetLastError();
}

hipError_t rt_launch_debug_cull(const float* clusters, const unsigned int* slot_of, int index, const float* rays, int n, unsigned char* out, hipStream_t stream)
{
    hipLaunchKernelGGL(debug_cull_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, clusters, slot_of, index, rays, n, out);
    return hipGetLastError();
}

hipError_t rt_launch_debug_intersect(const float4* hrec, int index, const float* rays, int n, float* out8, hipStream_t stream)
{
    hipLaunchKernelGGL(debug_intersect_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, hrec, index, rays, n, out8);
    return hipGetLastError();
}
#endif  // RT_RELAXED_TU
