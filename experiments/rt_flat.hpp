// The lane-level state machine form of the per-pixel driver + AntiAliasing + TraceRay (main.cpp:298-319,
// renderer.h:28-201).  Every lane owns one pixel and walks its 4*SAMPLES primary rays and their bounce trees on its
// own; the wave's loop body is "one scene query per lane" -- a closest-hit query (renderer.h:40) or a shadow query
// (directionalLight.h:13 / the point light) -- over the shared, wave-uniform object loop, followed by the lane's own
// bookkeeping (shade, next light, bounce, return to parents, next primary ray).  Lanes whose path ended start their
// next ray at once instead of waiting for the longest path of the wave, which is what glass / mirror scenes need
// (DESIGN.md "execution model").  Arithmetic, draw order and unwind order are those of trace_path().
#pragma once

#include "rt_device.hpp"
#include "rt_launch.h"

namespace rt {

template <bool EMISSION, bool MIXED, bool HAS_NONSPHERE, bool COUNT, bool PRUNE, int MAXSP, int MAXPEND, int BLOCK>
__device__ __forceinline__ void render_pixel_flat(const KernelArgs& a, const SceneLds& sc, float* dirs, int x, int y, int r,
                                                  Counters& cnt)
{
    enum { TAG_REFL_ = 0, TAG_REFR_ = 1 };

    // ---- pixel set-up (main.cpp:298-307, renderer.h:140-146)
    {
        Camera cam;
#pragma unroll
        for (int i = 0; i < 3; ++i) cam.pos[i] = a.cam[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) { cam.ip[i] = a.cam[3 + i]; cam.iv[i] = a.cam[19 + i]; }
        const float u = ((float)x / (float)a.width) * 2.0f - 1.0f;
        const float v = -(((float)y / (float)a.height) * 2.0f - 1.0f);
        const float offX = 0.5f / a.width, offY = 0.5f / a.height;
        // the four primary directions are per-pixel constants (AntiAliasing gets the same u, v every sample)
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            float uu = (k & 1) ? u + offX : u - offX;
            float vv = (k & 2) ? v + offY : v - offY;
            v3 d = uv_to_direction(cam, uu, vv);
            dirs[(3 * k + 0) * BLOCK] = d.x; dirs[(3 * k + 1) * BLOCK] = d.y; dirs[(3 * k + 2) * BLOCK] = d.z;
        }
    }
    const v3 origin = mk(a.cam[0], a.cam[1], a.cam[2]);

    unsigned int rng_words[31];
    Rng rng;
    rng.init(rng_words, a.seed_base + (unsigned int)(x + y * a.width));                // rand.h:7, main.cpp:304

    float4 stk[MAXSP];
    float4 stk_em[(MIXED && EMISSION) ? MAXSP : 1];
    float4 pend[MIXED ? 3 * MAXPEND : 1];

    // pixel / sample accumulators
    v3 acc_color = mk(0, 0, 0), acc_em = mk(0, 0, 0);
    float acc_st = 0.0f;
    v3 res_color = mk(0, 0, 0), res_em = mk(0, 0, 0);
    float res_st = 0.0f;
    int s = 0, k = 0;

    // path state
    v3 ro = origin, rd = mk(dirs[0], dirs[BLOCK], dirs[2 * BLOCK]);
    float mult = 1.0f;
    int depth = 1, maxDepth = 0, sp = 0, pp = 0;

    // node state (the hit being shaded)
    v3 color = mk(0, 0, 0), emission = mk(0, 0, 0);
    float strength = 0.0f;
    int mat = 0;
    v3 hpos = mk(0, 0, 0), hnorm = mk(0, 0, 0);

    for (;;) {
        // -------- TraceRay for (ro, rd, mult, depth): one closest-hit query per lane per trip
        bool have_ray = false;     // the lane has set up its next ray
        bool need_refr = false;
        if (mult < 0.001f) {                                                     // renderer.h:30-34
            maxDepth = depth;
            color = mk(0, 0, 0); emission = mk(0, 0, 0); strength = 0.0f;
        } else {
            if (COUNT) ++cnt.closest;
            float best;
            int bi = closest_hit<HAS_NONSPHERE, COUNT>(sc, ro, rd, best, cnt);
            if (bi < 0) {                                                        // renderer.h:40-50
                maxDepth = depth;
                float aa = (rd.y + 1.0f) * 0.5f;
                float w = 1.0f - aa;
                color = mk(w + aa * 0.2f, w + aa * 0.3f, w + aa * 0.8f) * mult;
                emission = mk(0, 0, 0); strength = 0.0f;
            } else {
                Hit h = hit_record(sc, bi, ro, rd, best);
                hpos = h.position; hnorm = h.normal; mat = h.material;
                ro = hpos + hnorm * 0.005f;                                      // renderer.h:55,70 (offPosition)
                if (COUNT) ++cnt.shaded;
                float intensity = light_intensity<HAS_NONSPHERE, COUNT, PRUNE>(sc, ro, hnorm, cnt);
                const float* m = sc.mats + 10 * mat;
                color = (mk(m[0], m[1], m[2]) * intensity) * mult;               // renderer.h:64
                if (EMISSION) { emission = mk(m[6], m[7], m[8]); strength = m[9]; }
                if (depth <= a.max_depth) {                                      // renderer.h:68
                    float reflection = m[4];
                    if (reflection > 0) {                                        // renderer.h:74-91
                        v3 d = reflect(rd, hnorm);
                        v3 hemi = random_unit_hemisphere<COUNT>(rng, hnorm, cnt);
                        d = normalize(d + hemi * m[3]);
                        if (dot(d, hnorm) > 0) {
                            if (sp >= MAXSP || (MIXED && pp >= MAXPEND)) { *a.fault = 1; return; }
                            bool mixed = MIXED && m[5] > 0;
                            stk[sp] = make_float4(color.x, color.y, color.z,
                                                  __int_as_float(mat | (TAG_REFL_ << 8) | ((int)mixed << 9)));
                            if (MIXED && mixed) {
                                if (EMISSION) stk_em[sp] = make_float4(emission.x, emission.y, emission.z, strength);
                                pend[3 * pp + 0] = make_float4(hpos.x, hpos.y, hpos.z, mult);
                                pend[3 * pp + 1] = make_float4(hnorm.x, hnorm.y, hnorm.z, __int_as_float(depth));
                                pend[3 * pp + 2] = make_float4(rd.x, rd.y, rd.z, 0.0f);
                                ++pp;
                            }
                            ++sp;
                            rd = d;                                              // ro is already offPosition
                            mult = mult * reflection; depth = depth + 1;
                            have_ray = true;
                        }
                    }
                    if (!have_ray) need_refr = true;
                } else {
                    maxDepth = depth;                                            // renderer.h:131-132
                }
            }
        }
        if (have_ray) continue;

        // -------- refraction step of the current node / return to the parent(s)   (rd = the ray that hit)
        bool path_done = false;
        for (;;) {
            if (need_refr) {
                need_refr = false;
                const float* m = sc.mats + 10 * mat;
                float reflection = m[4], refraction = m[5];
                if (refraction > 0) {                                            // renderer.h:93-126
                    v3 rayDir = normalize(rd);                                   // renderer.h:72
                    v3 outNorm; float ir;
                    if (dot(rayDir, hnorm) > 0) { outNorm = -hnorm; ir = refraction; }
                    else                        { outNorm = hnorm;  ir = 1.0f / refraction; }
                    v3 d;
                    if (!refract_dir(rayDir, outNorm, ir, d)) d = reflect(rayDir, hnorm);
                    v3 hemi = random_unit_hemisphere<COUNT>(rng, hnorm, cnt);
                    d = normalize(d + hemi * m[3]);
                    if (sp >= MAXSP) { *a.fault = 1; return; }
                    bool mixed = MIXED && reflection > 0;
                    stk[sp] = make_float4(color.x, color.y, color.z,
                                          __int_as_float(mat | (TAG_REFR_ << 8) | ((int)mixed << 9)));
                    if (MIXED && EMISSION && mixed) stk_em[sp] = make_float4(emission.x, emission.y, emission.z, strength);
                    ++sp;
                    ro = hpos + d * 0.01f; rd = d;                               // renderer.h:116
                    mult = mult * 0.9f;                                          // renderer.h:118 (same depth)
                    have_ray = true;
                    break;
                }
                if (reflection == 0 && refraction == 0) maxDepth = depth;        // renderer.h:128-129
            }
            if (sp == 0) { path_done = true; break; }
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
                    float scaled = (tag == TAG_REFL_) ? strength * pm[4] * 1.25f : strength * 0.95f;
                    pst = gmax(pst, scaled);
                }
                emission = pem; strength = pst;
            }
            mat = pmat;
            if (MIXED && mixed && tag == TAG_REFL_) {
                --pp;
                float4 p0 = pend[3 * pp + 0], p1 = pend[3 * pp + 1], p2 = pend[3 * pp + 2];
                hpos = mk(p0.x, p0.y, p0.z); mult = p0.w;
                hnorm = mk(p1.x, p1.y, p1.z); depth = __float_as_int(p1.w);
                rd = mk(p2.x, p2.y, p2.z);
                need_refr = true;
            }
        }
        if (!path_done) continue;

        // -------- the primary ray's tree is complete (renderer.h:168-195)
        {
            v3 pc = color;
            if (maxDepth != 1) { float md = (float)maxDepth; pc = mk(pc.x / md, pc.y / md, pc.z / md); }   // x / 1.0f == x
            res_color = res_color + pc;
            if (EMISSION) {
                res_em = res_em + emission;
                res_st = (k == 0) ? strength : gmax(res_st, strength);
            }
        }
        ++k;
        if (k == a.rays_per_sample) {
            // sample complete (renderer.h:197-198, main.cpp:313-315)
            res_color = res_color * 0.25f;
            acc_color = acc_color + mk(clamp01(res_color.x), clamp01(res_color.y), clamp01(res_color.z));
            if (EMISSION) {
                res_em = res_em * 0.25f;
                acc_em = acc_em + mk(clamp01(res_em.x), clamp01(res_em.y), clamp01(res_em.z));
                acc_st += res_st;
            }
            res_color = mk(0, 0, 0); res_em = mk(0, 0, 0); res_st = 0.0f;
            k = 0; ++s;
            if (s == a.samples) break;
        }
        // next primary ray of this pixel
        ro = origin;
        rd = mk(dirs[(3 * k + 0) * BLOCK], dirs[(3 * k + 1) * BLOCK], dirs[(3 * k + 2) * BLOCK]);
        mult = 1.0f; depth = 1; maxDepth = 0; sp = 0; pp = 0;
    }

    // ---- pixel complete (main.cpp:318-319)
    const float n = (float)a.samples;
    const size_t o = (size_t)x + (size_t)r * a.width;
    if (a.rgb) {
        a.rgb[o * 3 + 0] = quantise(acc_color.x / n);
        a.rgb[o * 3 + 1] = quantise(acc_color.y / n);
        a.rgb[o * 3 + 2] = quantise(acc_color.z / n);
    }
    if (EMISSION && a.emission)
        a.emission[o] = make_float4(acc_em.x / n, acc_em.y / n, acc_em.z / n, acc_st / n);
}

}  // namespace rt
