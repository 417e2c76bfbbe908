// Device-side building blocks of the TraceRay hot path for gfx950 (wave64).  Everything here is float32 in the
// reference's operation order (GLM 0.9.9.8 scalar mode); the translation unit is built with
// -ffp-contract=off and HIP's default correctly-rounded fp32 divide/sqrt, so results are bit-identical to the x86
// reference.  Reference citations are relative to the reference's src/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef RT_UNROLL
#define RT_UNROLL 1
#endif

#ifndef RT_NS
#define RT_NS rt
#endif
namespace RT_NS {

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 mk(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator-(v3 a) { return mk(-a.x, -a.y, -a.z); }
// glm dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z (vendor/glm/detail/func_geometric.inl:48-55)
__device__ __forceinline__ float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// Correctly rounded square root and reciprocal (IEEE-754, what sqrtf() and 1.0f / y give on the x86 reference) in
// fewer instructions than the compiler's general sequences (16 for sqrtf, 11 for a division).  Both were compared with
// those over all 2^32 bit patterns on an MI355X (tools/exp/exact_math.hip; rt_debug_exact_math runs the same sweep in
// the GPU suite): outside the guards below no input differs.
//   rt_sqrt: v_sqrt_f32 is within 1 ulp; the two fused residuals x - (s -/+ 1 ulp) * s say which neighbour is the
//   rounded root (the compiler's own sequence without its scaling of tiny inputs and its zero / infinity fix-up: +-0,
//   inf and NaN come out right by themselves; inputs below 2^-104 would lose the residual's sign to underflow).
//   rt_rcp: one Newton step on v_rcp_f32 (1 ulp) with fused residuals rounds correctly whenever y and 1 / y are both
//   normal numbers.
// RT_RELAXED (rt_kernels_relaxed.hip, the direct kernel behind RT_FLAG_RELAXED): not the reference's bits but within
// the north star's tolerance -- the hardware's 1-ulp square root, reciprocal and reciprocal square root as they are, and
// the translation unit is built with fused multiply-adds allowed.
#ifndef RT_RELAXED
#define RT_RELAXED 0
#endif
#ifndef RT_SHORT_EXACT
#define RT_SHORT_EXACT 1
#endif
// (a wave-level guard -- any lane outside sends the whole wave down the compiler's sequence, a scalar branch -- measured
// worse: 95 instead of 85 VGPRs in the direct kernel and 15 % more instructions)
#define RT_ANY_LANE(c) (c)
__device__ __forceinline__ float rt_sqrt(float x)
{
#if RT_RELAXED
    return __builtin_amdgcn_sqrtf(x);
#elif RT_SHORT_EXACT
    if (__builtin_expect(RT_ANY_LANE(x < 0x1p-96f), 0)) return sqrtf(x); // tiny, zero, negative
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    s = (rm <= 0.0f) ? sm : s;
    s = (rp > 0.0f) ? sp : s;
    return s;
#else
    return sqrtf(x);
#endif
}
__device__ __forceinline__ float rt_rcp(float y)
{
#if RT_RELAXED
    return __builtin_amdgcn_rcpf(y);
#elif RT_SHORT_EXACT
    const float a = fabsf(y);
    if (__builtin_expect(RT_ANY_LANE(!(a >= 0x1p-126f && a <= 0x1p126f)), 0)) return 1.0f / y;   // zero, denormal, huge, inf, NaN
    const float r = __builtin_amdgcn_rcpf(y);
    return __builtin_fmaf(__builtin_fmaf(-y, r, 1.0f), r, r);
#else
    return 1.0f / y;
#endif
}
// 1.0f / sqrtf(x) for x = a sum of squares: the root of a float in [2^-96, FLT_MAX] lies in [2^-48, 2^64], where
// rt_rcp's fast path holds, so one range test serves both
__device__ __forceinline__ float rt_inv_sqrt(float x)
{
#if RT_RELAXED
    return __builtin_amdgcn_rsqf(x);
#elif RT_SHORT_EXACT
    if (__builtin_expect(RT_ANY_LANE(!(x >= 0x1p-96f && x <= 3.402823466e+38f)), 0)) return 1.0f / sqrtf(x);
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    s = (rm <= 0.0f) ? sm : s;
    s = (rp > 0.0f) ? sp : s;
    const float r = __builtin_amdgcn_rcpf(s);
    return __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
#else
    return 1.0f / sqrtf(x);
#endif
}
// glm normalize = v * (1 / sqrt(dot(v,v))) (func_geometric.inl:88, func_exponential.inl:138)
__device__ __forceinline__ v3 normalize(v3 a) { return a * rt_inv_sqrt(dot(a, a)); }
// glm reflect = I - N * dot(N,I) * 2 (func_geometric.inl:108)
__device__ __forceinline__ v3 reflect(v3 I, v3 N) { return I - (N * dot(N, I)) * 2.0f; }
// glm min/max (func_common.inl:20,29): NaN handling follows the comparisons, so no fminf/fmaxf here
#if RT_RELAXED
__device__ __forceinline__ float gmin(float x, float y) { return fminf(x, y); }      // (one instruction; differs for NaNs only)
__device__ __forceinline__ float gmax(float x, float y) { return fmaxf(x, y); }
#else
__device__ __forceinline__ float gmin(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float gmax(float x, float y) { return (x < y) ? y : x; }
#endif
__device__ __forceinline__ float clamp01(float x) { return gmin(gmax(x, 0.0f), 1.0f); }

// coop_kernel: all 64 lanes of a wave hold one pixel's path, so its state is the same in every lane.  Passing a value
// through v_readfirstlane tells the compiler so, and it keeps the value in a scalar register: the path state (some 50
// values) then costs no vector registers, which the kernel's occupancy hangs on.  ON = false: the value as it is.
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned int uni(unsigned int v) { return (unsigned int)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ v3 uni(v3 v) { return mk(uni(v.x), uni(v.y), uni(v.z)); }
template <bool ON, class T> __device__ __forceinline__ void make_uniform(T& v) { if (ON) v = uni(v); }

// ---------------------------------------------------------------------------------------------------------------
// Scene image.  The host flattens the scene into these records (rt_abi.cpp), the kernel stages them into LDS once
// per workgroup.  One hittable = two float4:
//   r0 = (a.x, a.y, a.z, sphere: radius*radius)         a = sphere centre | plane centre | cube min corner
//   r1 = (b.x, b.y, b.z, bits(type | material << 2))    b = plane unit normal | cube max corner (a + size)
// radius*radius and a+size are the same float operations the reference performs per ray (sphere.h:19, cube.h:17).
// One light = one float4 (v.xyz, bits(type)): v = normalize(-direction) (directionalLight.h:9) | position.
// One material = 10 floats (material.h:5-11).
enum { T_SPHERE = 0, T_PLANE = 1, T_CUBE = 2 };
enum { L_DIR = 0, L_POINT = 1 };

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(4)))* kptr4;    // constant address space: uniform loads become s_load

struct SceneLds {
    const float4* hrec;   // 2 * n_hit
    const float4* lights; // n_lights
    const float*  mats;   // 10 * n_mat
    int n_hit, n_lights;
    kptr4 khrec, klights; // the same records in global memory, for wave-uniform (scalar) reads
    kptr4 kprim;          // per sphere (camera - centre, |camera - centre|^2 - r^2): primary rays share their origin
    // coop_kernel stages the records sorted by type: hrec[2 * slot], and every index it handles is a slot.  orig (LDS):
    // slot -> index in the scene, which decides ties; slot_of (global): the inverse, for walking in scene order.  Both
    // null when the order is the scene's own (sphere-only scenes, every other kernel).
    const unsigned short* orig;
    const unsigned int* slot_of;
    // coop_kernel: boxes (LDS, 8 floats each) of the rounds of 64 slots, for cluster_mask(); 0: none
    const float* clusters; int n_clusters;
    float4* xstack;       // coop_kernel: this wave's bounce stack in LDS (trace_path), or null: scratch
    int xstack_n;         // ... how many colour frames of it (fewer than the stack's capacity: the rest is in scratch)
    // coop_kernel: per-wave candidate lists for the pixel's primary rays and their shadow rays (LDS, ascending indices);
    // a negative count means there is no list
    const unsigned short* cl_prim; int cl_nprim;
    const unsigned short* cl_occl; int cl_noccl;
#ifdef RT_SCALAR_SCENE
    __device__ __forceinline__ float4 urec(int i) const { f32x4 v = khrec[i]; return make_float4(v.x, v.y, v.z, v.w); }
    __device__ __forceinline__ float4 ulight(int i) const { f32x4 v = klights[i]; return make_float4(v.x, v.y, v.z, v.w); }
#else
    __device__ __forceinline__ float4 urec(int i) const { return hrec[i]; }
    __device__ __forceinline__ float4 ulight(int i) const { return lights[i]; }
#endif
};

struct Counters { unsigned int closest, shadow, tests, draws, shaded; };

// ---------------------------------------------------------------------------------------------------------------
// RNG: glibc srand()/rand() TYPE_3 additive-feedback generator (rand.h:7,9), one generator per
// pixel: r[f] += r[b]; output r[f] >> 1; f, b step through 31 words, b three behind f.
// The 31 words live per lane either in scratch (STRIDE 1: a private array indexed at run time) or in LDS, slot-major
// (STRIDE = workgroup size: word k of thread t at [k*STRIDE + t], a wave's access to one slot is 64 consecutive
// banks).  Draws come three at a time (RANDOM_UNIT_VECTOR, core.h:24-25) and the three updates touch six distinct
// words, so their loads are issued together.  Seeding (30 Lehmer steps + 310 discarded outputs) runs in registers,
// fully unrolled and out of line, and is lazy: most pixels of a diffuse scene never draw.
template <int STRIDE>
struct RngT {
    static constexpr int stride = STRIDE;
    unsigned int* r;    // the lane's word 0
    int f;              // front slot; rear slot = f - 3 (mod 31)
    bool seeded;
    unsigned int seed;
    unsigned int ndraws;   // rand() calls so far

    __device__ __forceinline__ void init(unsigned int* words, unsigned int s) { r = words; seed = s; seeded = false; f = 3; ndraws = 0; }

    // srand(seed): out of line, so that its 31 live words never share a register budget with the hot loops
    __device__ __noinline__ static void seed_words(unsigned int* r, unsigned int seed)
    {
        unsigned int t[31];
        unsigned int s = seed == 0u ? 1u : seed;
        t[0] = s;
        int word = (int)s;
#pragma unroll
        for (int i = 1; i < 31; ++i) {
            // 16807 * word % 2147483647 without overflow (glibc random_r.c __srandom_r)
            int hi = word / 127773;
            int lo = word % 127773;
            word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            t[i] = (unsigned int)word;
        }
        // discard 310 outputs: 10 full turns of t[(k+3)%31] += t[k]
#pragma unroll 1
        for (int turn = 0; turn < 10; ++turn) {
#pragma unroll
            for (int k = 0; k < 31; ++k) t[(k + 3) % 31] += t[k];
        }
#pragma unroll
        for (int k = 0; k < 31; ++k) r[k * STRIDE] = t[k];
    }

    __device__ __forceinline__ void do_seed() { seed_words(r, seed); f = 3; seeded = true; }

    __device__ __forceinline__ static int wrap(int i) { return i >= 31 ? i - 31 : i; }

    // one rand()
    __device__ __forceinline__ int next()
    {
        int b = wrap(f + 28);
        unsigned int v = r[f * STRIDE] + r[b * STRIDE];
        r[f * STRIDE] = v;
        f = wrap(f + 1);
        ndraws += 1;
        return (int)(v >> 1);
    }

    // three consecutive rand(): words f, f+1, f+2 are updated from words f-3, f-2, f-1 -- no overlap
    __device__ __forceinline__ void next3(int& o0, int& o1, int& o2)
    {
        int f0 = f, f1 = wrap(f + 1), f2 = wrap(f + 2);
        int b0 = wrap(f + 28), b1 = wrap(f + 29), b2 = wrap(f + 30);
        unsigned int v0 = r[f0 * STRIDE] + r[b0 * STRIDE], v1 = r[f1 * STRIDE] + r[b1 * STRIDE], v2 = r[f2 * STRIDE] + r[b2 * STRIDE];
        r[f0 * STRIDE] = v0; r[f1 * STRIDE] = v1; r[f2 * STRIDE] = v2;
        f = wrap(f + 3);
        ndraws += 3;
        o0 = (int)(v0 >> 1); o1 = (int)(v1 >> 1); o2 = (int)(v2 >> 1);
    }
};

typedef RngT<1> Rng;   // words in a private (scratch) array

// The same generator read from a *stream*: the raw words r[i] = r[i-3] + r[i-31] written out in order (word 31 + n of
// the stream is draw n; the first 31 words are the state the stream starts from).  A lane that knows the index of its
// first draw needs no private generator state, so many lanes can draw from one pixel's sequence at guessed offsets
// (heavy_kernel).  Words come through a 32-word LDS window per lane (word k at win[k * STRIDE]).
template <int STRIDE>
struct RngStreamT {
    const unsigned int* src;   // the stream (31 state words, then the draws)
    unsigned int* win;
    unsigned int off;          // index of the next draw
    unsigned int lim;          // draws written so far
    unsigned int cap;          // draws the stream's buffer holds
    int pos;                   // next word of the window; 32 = empty
    bool seeded;               // always true: the stream is what srand() + earlier draws left
    bool overflow;             // drew past lim: the values were not the generator's
    unsigned int ndraws;

    __device__ __forceinline__ void start(const unsigned int* stream, unsigned int* window, unsigned int first_draw,
                                          unsigned int written, unsigned int capacity)
    {
        src = stream; win = window; off = first_draw; lim = written; cap = capacity;
        pos = 32; seeded = true; overflow = false; ndraws = 0;
    }
    __device__ __forceinline__ void do_seed() {}

    __device__ __forceinline__ void refill()
    {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            unsigned int i = off + (unsigned int)k;
            if (i >= cap) i = cap - 1;             // stay inside the buffer; such draws are flagged by `overflow`
            win[k * STRIDE] = src[31 + i];
        }
        pos = 0;
    }

    __device__ __forceinline__ void next3(int& o0, int& o1, int& o2)
    {
        if (pos > 29) refill();
        unsigned int v0 = win[pos * STRIDE], v1 = win[(pos + 1) * STRIDE], v2 = win[(pos + 2) * STRIDE];
        pos += 3; off += 3; ndraws += 3;
        if (off > lim) overflow = true;
        o0 = (int)(v0 >> 1); o1 = (int)(v1 >> 1); o2 = (int)(v2 >> 1);
    }
};

// Appends 31 * turns words to a stream whose last 31 words precede `out` (out[-31 .. -1]).
__device__ __noinline__ static void stream_turns(unsigned int* out, int turns)
{
    unsigned int t[31];
#pragma unroll
    for (int k = 0; k < 31; ++k) t[k] = out[k - 31];
#pragma unroll 1
    for (int turn = 0; turn < turns; ++turn) {
#pragma unroll
        for (int n = 0; n < 31; ++n) { t[n] += t[(n + 28) % 31]; out[n] = t[n]; }
        out += 31;
    }
}

// randomUnit() = (float)rand() / (float)RAND_MAX, (float)RAND_MAX == 2^31 (rand.h:9)
__device__ __forceinline__ float rand_unit(int r) { return (float)r * 4.656612873077392578125e-10f; }

// ---------------------------------------------------------------------------------------------------------------
// Primitive tests.  Each returns whether the reference's intersect() would return true and, if so, hit.distance.
// Position / normal are produced afterwards for the winning hittable only (same values: the reference overwrites
// them on every accepted candidate, hittablesList.h:27-31).

// sphere.h:13-44.  `a` = dot(dir,dir) and `fa` = 4*a are per-ray values; origin = ray.origin - centre and
// c = dot(origin, origin) - r*r depend on the ray's origin only.
__device__ __forceinline__ bool sphere_hit_oc(v3 origin, float c, v3 rd, float fa, float& dist)
{
    float b = 2.0f * dot(origin, rd);
    float discriminant = b * b - fa * c;
    if (discriminant < 0.0f) return false;
    float sqrtd = rt_sqrt(discriminant);
    float t1 = (-b - sqrtd) / 2.0f;
    float t2 = (-b + sqrtd) / 2.0f;
    if (t1 < 0 && t2 < 0) return false;
    if (t2 > t1) dist = (t1 >= 0) ? t1 : t2;
    else         dist = (t2 >= 0) ? t2 : t1;
    return true;
}

__device__ __forceinline__ bool sphere_hit(float4 r0, v3 ro, v3 rd, float fa, float& dist)
{
    v3 origin = ro - mk(r0.x, r0.y, r0.z);
    float c = dot(origin, origin) - r0.w;
    return sphere_hit_oc(origin, c, rd, fa, dist);
}

// The boolean of sphere.h:13-33 only (any-hit queries), without the square root where the outcome is already decided.
// With sqrtd = RN(sqrt(disc)) >= 0 and t2 = (-b + sqrtd)/2 >= t1, intersect() is true iff !(t2 < 0):
//   b <= 0 : -b + sqrtd >= 0                                                                    -> true
//   b > 0 and disc < bb*(1 - 1e-6), bb = RN(b*b): RN(sqrt(x)) is monotone and RN(sqrt(RN(b*b))) == b, and a disc
//            that far below bb has sqrt(disc) < b*(1 - 2^-22) < b - ulp(b)/2, so sqrtd < b strictly; -b + sqrtd is
//            then a negative number of magnitude > b*2^-23, far from the denormals where /2 could round to -0
//            (b > 1e-18 is required for that)                                                   -> false
//   anything else (disc within 1e-6 of bb, tiny b, NaNs)                                        -> the full test
__device__ __forceinline__ bool sphere_occludes(float4 r0, v3 ro, v3 rd, float fa)
{
    v3 origin = ro - mk(r0.x, r0.y, r0.z);
    float b = 2.0f * dot(origin, rd);
    float c = dot(origin, origin) - r0.w;
    float bb = b * b;
    float discriminant = bb - fa * c;
    if (discriminant < 0.0f) return false;
    if (b <= 0.0f) return true;
    if (b > 1e-18f && discriminant < bb * 0.999999f) return false;
    float sqrtd = rt_sqrt(discriminant);
    float t1 = (-b - sqrtd) / 2.0f;
    float t2 = (-b + sqrtd) / 2.0f;
    return !(t1 < 0 && t2 < 0);
}

// plane.h:13-23.  The reference compares |denom| against the double 1e-6; for a float x, (double)x <= 1e-6 is
// the same as x <= (float)1e-6 because (float)1e-6 < 1e-6 ("epsilon compares").
__device__ __forceinline__ bool plane_hit(float4 r0, float4 r1, v3 ro, v3 rd, float& dist)
{
    v3 n = mk(r1.x, r1.y, r1.z);
    float denom = dot(n, rd);
    if (fabsf(denom) <= 1e-6f) return false;
    float t = dot(ro - mk(r0.x, r0.y, r0.z), n) / denom;
    if (t < 0) return false;
    dist = t;
    return true;
}

// cube.h:13-28 (always tMin, even when negative).  inv = 1/dir per component, IEEE infinities included.
__device__ __forceinline__ bool cube_hit(float4 r0, float4 r1, v3 ro, v3 inv, float& dist)
{
    v3 t0 = (mk(r0.x, r0.y, r0.z) - ro) * inv;
    v3 t1 = (mk(r1.x, r1.y, r1.z) - ro) * inv;
    v3 tmin = mk(gmin(t0.x, t1.x), gmin(t0.y, t1.y), gmin(t0.z, t1.z));
    v3 tmax = mk(gmax(t0.x, t1.x), gmax(t0.y, t1.y), gmax(t0.z, t1.z));
    float tMin = gmax(gmax(tmin.x, tmin.y), tmin.z);
    float tMax = gmin(gmin(tmax.x, tmax.y), tmax.z);
    if (tMax < 0 || tMin > tMax) return false;
    dist = tMin;
    return true;
}

__device__ __forceinline__ int rec_type(float4 r1) { return __float_as_int(r1.w) & 3; }
__device__ __forceinline__ int rec_material(float4 r1) { return __float_as_int(r1.w) >> 2; }

template <bool HAS_NONSPHERE, bool UNIFORM = true>
__device__ __forceinline__ bool hittable_hit(float4 r0, float4 r1, v3 ro, v3 rd, float fa, v3 inv, float& dist)
{
    if (HAS_NONSPHERE) {
        // UNIFORM: the record index is wave-uniform, so the type is too: make the branch scalar
        int type = UNIFORM ? __builtin_amdgcn_readfirstlane(rec_type(r1)) : rec_type(r1);
        if (type == T_PLANE) return plane_hit(r0, r1, ro, rd, dist);
        if (type == T_CUBE)  return cube_hit(r0, r1, ro, inv, dist);
    }
    return sphere_hit(r0, ro, rd, fa, dist);
}

struct Hit { v3 position, normal; int material; };

// position / normal / material of hittable `idx` at distance `dist` (sphere.h:46-49, plane.h:24-26, cube.h:28-46)
// (SPHERES_ONLY: the scene holds nothing else, rt_abi.cpp's `nonsphere`)
template <bool SPHERES_ONLY = false>
__device__ __forceinline__ Hit hit_record(const SceneLds& sc, int idx, v3 ro, v3 rd, float dist)
{
    float4 r0 = sc.hrec[2 * idx], r1 = sc.hrec[2 * idx + 1];
    int type = SPHERES_ONLY ? (int)T_SPHERE : rec_type(r1);
    Hit h;
    h.material = rec_material(r1);
    v3 a = mk(r0.x, r0.y, r0.z);
    if (type == T_SPHERE) {
        v3 origin = ro - a;
        v3 p = origin + rd * dist;
        h.normal = normalize(p);
        h.position = p + a;
    } else if (type == T_PLANE) {
        h.position = ro + rd * dist;
        h.normal = mk(r1.x, r1.y, r1.z);
    } else {
        v3 hp = ro + rd * dist;
        // |x| < 1e-4 (double) <=> |x| <= (float)1e-4 for float x
        v3 n;
        if      (fabsf(hp.x - r0.x) <= 1e-4f) n = mk(-1, 0, 0);
        else if (fabsf(hp.x - r1.x) <= 1e-4f) n = mk(1, 0, 0);
        else if (fabsf(hp.y - r0.y) <= 1e-4f) n = mk(0, -1, 0);
        else if (fabsf(hp.y - r1.y) <= 1e-4f) n = mk(0, 1, 0);
        else if (fabsf(hp.z - r0.z) <= 1e-4f) n = mk(0, 0, -1);
        else                                  n = mk(0, 0, 1);
        h.position = hp;
        h.normal = n;
    }
    return h;
}

// ---------------------------------------------------------------------------------------------------------------
// Cooperative scans (BATCH == RT_COOP): all 64 lanes of the wave hold the *same* ray (one wave renders one pixel, every
// lane runs the same path code on the same values) and share the scan of the scene: lane l tests hittables l, l + 64,
// ... from the LDS copy, then the wave combines.  Without NaN distances the sequential rule (a hit replaces unless
// `dist >= best`, best starting at FLT_MAX) picks the smallest distance below FLT_MAX and, among equals, the lowest
// index; that is what each lane computes over its indices and what the butterfly computes over lanes.  A NaN distance
// anywhere makes the order of replacements matter (hittablesList.h:25): the wave then repeats the query sequentially.
#define RT_COOP (-1)
// (experiments: -DRT_COOP_STATS tallies the cooperative queries; rt_kernels.hip prints the sums after coop_kernel)
#ifdef RT_COOP_STATS
__device__ unsigned long long g_coop_stats[12];
#define RT_CSTAT(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_coop_stats[k], (unsigned long long)(v)); } while (0)
#else
#define RT_CSTAT(k, v) ((void)0)
#endif

template <bool HAS_NONSPHERE>
__device__ __forceinline__ int closest_hit_seq(const SceneLds& sc, v3 ro, v3 rd, float& best);
template <bool HAS_NONSPHERE, bool COUNT, bool LIMITED, bool UNIFORM = true>
__device__ __forceinline__ bool occluder_hit(float4 r0, float4 r1, v3 ro, v3 rd, float fa, v3 inv, float limit);

// Minimum / maximum of a value over the 64 lanes, the same in every lane afterwards (no NaNs among the inputs): an
// xor butterfly inside each row of 16 lanes by DPP (quad_perm [1,0,3,2], [2,3,0,1], row_ror 4 and 8 -- no LDS
// traffic, unlike __shfl_xor's ds_bpermute), then the four rows' results by readlane.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v)
{
    const int x = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false));
}
template <bool MAX>
__device__ __forceinline__ float wave_extreme(float v)
{
    auto pick = [](float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); };
    v = pick(v, dpp_move<0xB1>(v));
    v = pick(v, dpp_move<0x4E>(v));
    v = pick(v, dpp_move<0x124>(v));
    v = pick(v, dpp_move<0x128>(v));
    const int x = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(x, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(x, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(x, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(x, 48));
    return pick(pick(r0, r1), pick(r2, r3));
}

// Which rounds of a cooperative scan can hold a hit at all.  The staged order puts hittables that are close in space
// into the same round of 64 (rt_abi.cpp rt_upload_scene), and every round has an axis-aligned box around its members.
// A round is skipped when intersect() *certainly* returns false for every member -- skipping a test that returns false
// changes nothing (hittablesList.h:25-26 `continue`, :46) -- which is the case when the ray (t >= 0) misses the box
// grown by a margin m on every side, m = D / 128, D = the distance from the ray's origin to the box's farthest corner
// (D bounds every member's |ro - centre|, radius and |corner - ro|).  Then for every member either (a) the ray's line
// stays >= m away from it, or (b) the line meets the member grown by m only over an interval that ends before t = 0.
// Why the margin is enough (u = 2^-24):
//   sphere.h:15-33, o = ro - centre, p = the centre's distance from the line, s = o.d / |d|.  The float discriminant
//     b b - 4 a c is within 100 u D^2 d.d of 4 d.d (r^2 - p^2), b within 8 u D |d| of 2 s |d|, c within 7 u D^2 of
//     p^2 + s^2 - r^2.  If p >= r + m / 2: 4 d.d (p^2 - r^2) >= d.d m^2 = 6e-5 D^2 d.d, ten times the error, so
//     `discriminant < 0`; that covers (a).  Otherwise (b) says s > sqrt((r + m)^2 - p^2) > 0.86 m, so b > D |d| / 75 is
//     positive, c >= 2 r m + m^2 >= 6e-5 D^2 is positive and 4 a c / b^2 >= 6e-5 >> u: the discriminant is negative or
//     sqrtd < b strictly, both roots negative.
//   cube.h:15-26.  Every t = (corner - origin) * (1 / d) is within 3 u |t| of its value.  (a): at the point P where
//     the line leaves the first slab (tMax) it is >= m from the cube, so outside another slab k by >= m / sqrt 2,
//     which it enters at tMax + g, g >= m / (sqrt 2 |d_k|); |t_enter_k| <= D / |d_k| and |tMax| <= |t_enter_k| + g, so
//     the computed tMin - tMax >= g - 3 u (2 D / |d_k| + g) > 0.  (b): min_i (t_exit_i + m / |d_i|) < 0, so for that
//     i the computed t_exit_i, and with it tMax, is negative (|t_exit_i| > m / |d_i|: far from rounding to -0).
//     Both need 1 / d_i finite and the products neither NaN nor infinite (cube.h's comparisons come out `true` on
//     those): rounds with a cube in them are only skipped for rays with every |d_i| > 1e-24 -- with D <= 1e12 no
//     product then passes 1e36.  (The bounds above are ratios: they do not care how small a finite |d_i| is.  Round 2
//     had 1e-12 here, and C4 paid for it: a ray refracted into a glass cube comes out of refract() along an axis but
//     for components of 1e-12 .. 1e-13, is "hit" by its own cube at tMin = -0.01 (cube.h:28 takes tMin when the origin
//     is inside), steps back and repeats until the multiplier runs out -- 66 closest-hit queries per such ray, a fifth
//     of all closest-hit queries of the frame, every one of them scanning all 33 rounds.)
//   Overflow and underflow would break these bounds (a NaN discriminant *is* a hit, sphere.h:22,32): nothing is skipped
//   unless d.d is within [1e-6, 1e6] and D within [1e-6, 1e12] (NaNs fail that too).
//   Planes and hittables with non-finite parameters make their round's box infinite: always visited.
// The miss is proved by a separating axis (the box's three, for the half-line; d x e_i, for the line), in float, with
// the box (rounded outwards, rt_abi.cpp) grown by D / 100: every term of the comparisons is bounded by D |d_i|, the
// float evaluation is off by less than 9 u of that, 1e-4 of the D |d_i| / 128 the argument needs -- the difference
// between / 100 and / 128 is there for it.
// the margin as a fraction of D (the argument above needs >= 1 / 128; tests build with 0 to see that they notice)
#ifndef RT_CULL_MARGIN
#define RT_CULL_MARGIN 0.01f
#endif
__device__ __forceinline__ bool cluster_may_hit(float4 c, float4 h, v3 ro, v3 rd)
{
    const float ax = fabsf(rd.x), ay = fabsf(rd.y), az = fabsf(rd.z);
    const float ox = c.x - ro.x, oy = c.y - ro.y, oz = c.z - ro.z;
    const float fx = fabsf(ox) + h.x, fy = fabsf(oy) + h.y, fz = fabsf(oz) + h.z;
    const float dd = rd.x * rd.x + rd.y * rd.y + rd.z * rd.z, D2 = fx * fx + fy * fy + fz * fz;
    bool usable = dd >= 1e-6f && dd <= 1e6f && D2 >= 1e-12f && D2 <= 1e24f;
    if (c.w != 0.0f) usable = usable && ax > 1e-24f && ay > 1e-24f && az > 1e-24f;
    if (!usable) return true;
    const float m = rt_sqrt(D2) * RT_CULL_MARGIN;
    const float Hx = h.x + m, Hy = h.y + m, Hz = h.z + m;
    bool miss = (ox > Hx && rd.x <= 0.0f) || (ox < -Hx && rd.x >= 0.0f)
             || (oy > Hy && rd.y <= 0.0f) || (oy < -Hy && rd.y >= 0.0f)
             || (oz > Hz && rd.z <= 0.0f) || (oz < -Hz && rd.z >= 0.0f);
    miss = miss || fabsf(oy * rd.z - oz * rd.y) > Hy * az + Hz * ay
                || fabsf(oz * rd.x - ox * rd.z) > Hx * az + Hz * ax
                || fabsf(ox * rd.y - oy * rd.x) > Hx * ay + Hy * ax;
    return !miss;
}

// All lanes of the wave hold the same ray; lane k decides round k (at most 64 rounds: RT_MAX_HITTABLES = 4096).
__device__ __forceinline__ unsigned long long cluster_mask(const SceneLds& sc, v3 ro, v3 rd)
{
    const int lane = (int)(threadIdx.x & 63);
    // no boxes staged (stage_scene without stage_scene_sorted): every round of 64 is visited, nothing is skipped
    if (sc.clusters == nullptr || sc.n_clusters <= 0) {
        const int rounds = (sc.n_hit + 63) / 64;
        return rounds >= 64 ? ~0ull : (1ull << rounds) - 1ull;
    }
    bool visit = lane < sc.n_clusters;            // (n_clusters = ceil(n_hit / 64) wherever the cooperative scans run)
    if (visit) visit = cluster_may_hit(((const float4*)sc.clusters)[2 * lane], ((const float4*)sc.clusters)[2 * lane + 1], ro, rd);
    return __ballot(visit);
}

// The same, in two parts: `near`, the rounds whose grown box holds the ray's origin -- scanned first: the hittable a
// bounce ray leaves, its neighbours, the ground -- and `rest`; and for the rounds of `rest` where the ray enters lane k's
// box (`enter`, the parameter t by slabs), so that a query can drop rounds it has no use for any more:
//   closest hit: a round entered later than the best distance so far.  For rays with every |d_i| > 1e-24 and |d|^2
//     within 1e-4 of 1 (`by_dist`; the others skip nothing this way) the t at which the ray enters the box grown by
//     m = D / 100 (each product within 3 u of its value) is at least 0.9 m below the t of any point P of a member,
//     all of which lie inside the box proper.  A member that is hit, is hit at such a P -- an origin inside a member is
//     inside the box: a `near` round -- and its computed distance (sphere.h:26-27's root is t |d|^2 within 100 u D,
//     cube.h:28's tMin is t within 3 u) is at least t (1 - 2e-4) - 1e-5 D > enter.  So enter > 1.001 best (0 for a
//     negative best: the origin lies inside a cube, cube.h:28, and only hittables that hold the origin can beat it) puts
//     every member's distance above `best`, and a hit with `distance >= best` is a `continue` in hittablesList.h:27;
//   the point light's "something closer than the light": the same with the light's distance for `best`.
struct ClusterRounds { unsigned long long near, rest; float enter; bool by_dist; };

__device__ __forceinline__ bool ray_by_dist(v3 rd)
{
    const float dd = dot(rd, rd);
    return dd >= 0.9999f && dd <= 1.0001f && fabsf(rd.x) > 1e-24f && fabsf(rd.y) > 1e-24f && fabsf(rd.z) > 1e-24f;
}

// Where a ray (ray_by_dist) enters the box (c, h) grown by its margin; `inside`: the origin lies in the grown box.  A box
// that cannot be judged (non-finite, out of range) is entered at -FLT_MAX, inside.
__device__ __forceinline__ float cluster_enter(float4 c, float4 h, v3 ro, v3 rd, bool& inside)
{
    const float ox = c.x - ro.x, oy = c.y - ro.y, oz = c.z - ro.z;
    const float fx = fabsf(ox) + h.x, fy = fabsf(oy) + h.y, fz = fabsf(oz) + h.z;
    const float D2 = fx * fx + fy * fy + fz * fz;
    inside = true;
    if (!(D2 >= 1e-12f && D2 <= 1e24f)) return -3.402823466e+38f;
    const float m = rt_sqrt(D2) * RT_CULL_MARGIN;
    const float Hx = h.x + m, Hy = h.y + m, Hz = h.z + m;
    const float ix = rt_rcp(rd.x), iy = rt_rcp(rd.y), iz = rt_rcp(rd.z);
    const float x0 = (ox - Hx) * ix, x1 = (ox + Hx) * ix;
    const float y0 = (oy - Hy) * iy, y1 = (oy + Hy) * iy;
    const float z0 = (oz - Hz) * iz, z1 = (oz + Hz) * iz;
    inside = fabsf(ox) <= Hx && fabsf(oy) <= Hy && fabsf(oz) <= Hz;
    return fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
}

__device__ __forceinline__ ClusterRounds cluster_rounds(const SceneLds& sc, v3 ro, v3 rd)
{
    const int lane = (int)(threadIdx.x & 63);
    ClusterRounds cr;
    cr.enter = -3.402823466e+38f;
    cr.by_dist = false;
    if (sc.clusters == nullptr || sc.n_clusters <= 0) { cr.near = 0ull; cr.rest = cluster_mask(sc, ro, rd); return cr; }
    cr.by_dist = ray_by_dist(rd);
    bool visit = lane < sc.n_clusters, inside = false;
    if (visit) {
        const float4 c = ((const float4*)sc.clusters)[2 * lane], h = ((const float4*)sc.clusters)[2 * lane + 1];
        visit = cluster_may_hit(c, h, ro, rd);
        inside = visit;                            // (rays that skip nothing by distance: every round with the first ones)
        if (visit && cr.by_dist) cr.enter = cluster_enter(c, h, ro, rd, inside);
    }
    cr.near = __ballot(visit && inside);
    cr.rest = __ballot(visit && !inside);
    return cr;
}

template <bool HAS_NONSPHERE>
__device__ __forceinline__ int closest_hit_coop(const SceneLds& sc, v3 ro, v3 rd, float& best)
{
    const int lane = (int)(threadIdx.x & 63);
    const bool sorted = sc.orig != nullptr;
    float fa = 4.0f * dot(rd, rd);
    v3 inv = mk(0, 0, 0);
    if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
    best = 3.402823466e+38f;
    int bi = -1;
    bool nan_seen = false;
    auto scan = [&](unsigned long long rounds) {
        while (rounds != 0ull) {
            const int j = 64 * __builtin_ctzll(rounds) + lane;
            rounds &= rounds - 1ull;
            if (j >= sc.n_hit) continue;
            float4 r0 = sc.hrec[2 * j], r1 = sc.hrec[2 * j + 1];
            float t;
            if (hittable_hit<HAS_NONSPHERE, false>(r0, r1, ro, rd, fa, inv, t)) {
                nan_seen = nan_seen || (t != t);
                // slots ascend, scene indices need not: equal distances go to the lower scene index
                if (!(t >= best) || (sorted && t == best && bi >= 0 && sc.orig[j] < sc.orig[bi])) { best = t; bi = j; }
            }
        }
    };
    // the rounds that can hold a hit (all of them without clusters): first those around the origin, then what is left of
    // the others once the best distance so far is known
    const ClusterRounds cr = cluster_rounds(sc, ro, rd);
    RT_CSTAT(0, 1); RT_CSTAT(1, __popcll(cr.near)); RT_CSTAT(2, __popcll(cr.rest));
    scan(cr.near);
    unsigned long long rest = cr.rest;
    if (rest != 0ull && cr.by_dist && cr.near != 0ull) {
        const float b = wave_extreme<false>(best);                       // (FLT_MAX: no hit yet, nothing is dropped)
        rest = __ballot(((rest >> lane) & 1ull) != 0ull && !(cr.enter > fmaxf(b * 1.001f, 0.0f)));
    }
    RT_CSTAT(3, __popcll(rest));
    scan(rest);
    if (__ballot(nan_seen) != 0ull) return closest_hit_seq<HAS_NONSPHERE>(sc, ro, rd, best);
    // The winner over the lanes: the smallest distance (no-hit lanes carry FLT_MAX, which no hit reaches); two hits at the
    // same distance go to the lower index in the scene.
    const float tmin = wave_extreme<false>(best);
    unsigned long long cand = __ballot(bi >= 0 && best == tmin);
    if (cand == 0ull) { best = 3.402823466e+38f; return -1; }
    int wl = __builtin_ctzll(cand);
    cand &= cand - 1ull;
    if (cand != 0ull) {                                          // a tie between lanes (rare)
        const int bo = sorted ? (int)sc.orig[bi >= 0 ? bi : 0] : bi;
        int wo = __builtin_amdgcn_readlane(bo, wl);
        while (cand != 0ull) {
            const int l = __builtin_ctzll(cand);
            cand &= cand - 1ull;
            const int o = __builtin_amdgcn_readlane(bo, l);
            if (o < wo) { wo = o; wl = l; }
        }
    }
    best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best), wl));       // (the winner's own: -0 and +0 compare equal)
    return __builtin_amdgcn_readlane(bi, wl);
}

template <bool HAS_NONSPHERE, bool LIMITED>
__device__ __forceinline__ bool any_hit_coop(const SceneLds& sc, v3 ro, v3 rd, float limit)
{
    const int lane = (int)(threadIdx.x & 63);
    float fa = 4.0f * dot(rd, rd);
    v3 inv = mk(0, 0, 0);
    if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
    // 64 hittables per round; the first round with an occluder ends the query (any order gives the same boolean): the
    // rounds around the origin first -- what shadows a point is mostly next to it
    const ClusterRounds cr = cluster_rounds(sc, ro, rd);
    unsigned long long rest = cr.rest;
    if (LIMITED && cr.by_dist && limit >= 0.0f)
        rest = __ballot(((rest >> lane) & 1ull) != 0ull && !(cr.enter > limit * 1.001f));
    RT_CSTAT(4, 1); RT_CSTAT(5, __popcll(cr.near) + __popcll(rest));
    for (int part = 0; part < 2; ++part) {
        unsigned long long rounds = part == 0 ? cr.near : rest;
        while (rounds != 0ull) {
            const int j = 64 * __builtin_ctzll(rounds) + lane;
            rounds &= rounds - 1ull;
            bool o = false;
            if (j < sc.n_hit) {
                float4 r0 = sc.hrec[2 * j], r1 = sc.hrec[2 * j + 1];
                o = occluder_hit<HAS_NONSPHERE, false, LIMITED, false>(r0, r1, ro, rd, fa, inv, limit);
            }
            RT_CSTAT(6, 1);
            if (__ballot(o) != 0ull) { RT_CSTAT(7, 1); return true; }
        }
    }
    return false;
}

// HittablesList::intersect (hittablesList.h:17-37): brute force, strict '<' so the earlier hittable wins ties;
// a hit replaces the best unless `dist >= best` (NaN distances therefore replace, and once the best is a NaN every
// later hit replaces it, as in the reference; a miss never does).
// BATCH > 1 reads the records of BATCH hittables before testing the first: one memory round trip per BATCH tests,
// for kernels whose waves wait on latency rather than on issue slots.
template <bool HAS_NONSPHERE, bool COUNT, int BATCH = 1>
__device__ __forceinline__ int closest_hit(const SceneLds& sc, v3 ro, v3 rd, float& best, Counters& cnt)
{
    if (BATCH == RT_COOP) return closest_hit_coop<HAS_NONSPHERE>(sc, ro, rd, best);
    float fa = 4.0f * dot(rd, rd);
    v3 inv = mk(0, 0, 0);
    if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
    best = 3.402823466e+38f;   // FLT_MAX
    int bi = -1;
    int i = 0;
    if (BATCH > 1) {
        for (; i + BATCH <= sc.n_hit; i += BATCH) {
            float4 r[2 * (BATCH > 1 ? BATCH : 1)];
#pragma unroll
            for (int j = 0; j < 2 * BATCH; ++j) r[j] = sc.urec(2 * i + j);
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                if (COUNT) ++cnt.tests;
                float t;
                if (hittable_hit<HAS_NONSPHERE>(r[2 * j], r[2 * j + 1], ro, rd, fa, inv, t) && !(t >= best)) { best = t; bi = i + j; }
            }
        }
    }
#pragma unroll RT_UNROLL
    for (; i < sc.n_hit; ++i) {
        float4 r0 = sc.urec(2 * i), r1 = sc.urec(2 * i + 1);
        if (COUNT) ++cnt.tests;
        float t;
        if (hittable_hit<HAS_NONSPHERE>(r0, r1, ro, rd, fa, inv, t) && !(t >= best)) { best = t; bi = i; }
    }
    return bi;
}

// the sequential rule over the staged records, in scene order (the cooperative query's fallback when a distance is NaN)
template <bool HAS_NONSPHERE>
__device__ __forceinline__ int closest_hit_seq(const SceneLds& sc, v3 ro, v3 rd, float& best)
{
    float fa = 4.0f * dot(rd, rd);
    v3 inv = mk(0, 0, 0);
    if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
    best = 3.402823466e+38f;
    int bi = -1;
    for (int k = 0; k < sc.n_hit; ++k) {
        const int j = sc.slot_of ? (int)sc.slot_of[k] : k;
        float4 r0 = sc.hrec[2 * j], r1 = sc.hrec[2 * j + 1];
        float t;
        if (hittable_hit<HAS_NONSPHERE, false>(r0, r1, ro, rd, fa, inv, t) && !(t >= best)) { best = t; bi = j; }
    }
    return bi;
}

// The same query for a primary ray of a sphere-only scene: all primary rays start at the camera, so
// (origin - centre, dot(..) - r*r) come precomputed per sphere (rt_abi.cpp, the same float operations).
template <bool COUNT, int BATCH = 1>
__device__ __forceinline__ int closest_hit_primary(const SceneLds& sc, v3 rd, float& best, Counters& cnt)
{
    float fa = 4.0f * dot(rd, rd);
    best = 3.402823466e+38f;
    int bi = -1;
    int i = 0;
    if (BATCH > 1) {
        for (; i + BATCH <= sc.n_hit; i += BATCH) {
            f32x4 p[BATCH > 1 ? BATCH : 1];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) p[j] = sc.kprim[i + j];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                if (COUNT) ++cnt.tests;
                float t;
                if (sphere_hit_oc(mk(p[j].x, p[j].y, p[j].z), p[j].w, rd, fa, t) && !(t >= best)) { best = t; bi = i + j; }
            }
        }
    }
#pragma unroll RT_UNROLL
    for (; i < sc.n_hit; ++i) {
        f32x4 p = sc.kprim[i];
        if (COUNT) ++cnt.tests;
        float t;
        if (sphere_hit_oc(mk(p.x, p.y, p.z), p.w, rd, fa, t) && !(t >= best)) { best = t; bi = i; }
    }
    return bi;
}

// HittablesList::hasIntersect (hittablesList.h:39-50): first hittable whose intersect() is true, no distance bound.
// With `limit`: the point-light occlusion test "some hittable is hit strictly closer than the light".
template <bool HAS_NONSPHERE, bool COUNT, bool LIMITED, bool UNIFORM>
__device__ __forceinline__ bool occluder_hit(float4 r0, float4 r1, v3 ro, v3 rd, float fa, v3 inv, float limit)
{
    if (!LIMITED) {
        // directional light: only the boolean matters
        bool is_sphere = true;
        if (HAS_NONSPHERE) is_sphere = (UNIFORM ? __builtin_amdgcn_readfirstlane(rec_type(r1)) : rec_type(r1)) == T_SPHERE;
        if (is_sphere) return sphere_occludes(r0, ro, rd, fa);
    }
    float d;
    if (hittable_hit<HAS_NONSPHERE, UNIFORM>(r0, r1, ro, rd, fa, inv, d)) return !LIMITED || d < limit;
    return false;
}

template <bool HAS_NONSPHERE, bool COUNT, bool LIMITED, int BATCH = 1>
__device__ __forceinline__ bool any_hit(const SceneLds& sc, v3 ro, v3 rd, float limit, Counters& cnt)
{
    if (BATCH == RT_COOP) return any_hit_coop<HAS_NONSPHERE, LIMITED>(sc, ro, rd, limit);
    float fa = 4.0f * dot(rd, rd);
    v3 inv = mk(0, 0, 0);
    if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
    int i = 0;
    if (BATCH > 1) {
        for (; i + BATCH <= sc.n_hit; i += BATCH) {
            float4 r[2 * (BATCH > 1 ? BATCH : 1)];
#pragma unroll
            for (int j = 0; j < 2 * BATCH; ++j) r[j] = sc.urec(2 * i + j);
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                if (COUNT) ++cnt.tests;
                if (occluder_hit<HAS_NONSPHERE, COUNT, LIMITED>(r[2 * j], r[2 * j + 1], ro, rd, fa, inv, limit)) return true;
            }
        }
    }
    for (; i < sc.n_hit; ++i) {
        float4 r0 = sc.urec(2 * i), r1 = sc.urec(2 * i + 1);
        if (COUNT) ++cnt.tests;
        if (occluder_hit<HAS_NONSPHERE, COUNT, LIMITED>(r0, r1, ro, rd, fa, inv, limit)) return true;
    }
    return false;
}

// LightsList::GetLightIntensity for the cooperative kernel.  Its value is max(0.085, the largest Lambert term among the
// lights in sight) (lightsList.h:22-40): it does not depend on the order in which the lights are asked.  All lanes of
// the wave hold the same point, so lane j works out light j's direction and Lambert term (the same float operations),
// and the lights are asked for occlusion -- each a scan of the scene -- in descending order of their terms: the first
// one in sight ends it, no later one can raise the maximum.  (At most 64 lights, RT_MAX_LIGHTS.)
template <bool HAS_NONSPHERE>
__device__ __forceinline__ float light_intensity_coop(const SceneLds& sc, v3 position, v3 normal)
{
    const int lane = (int)(threadIdx.x & 63);
    float my = -1.0f, lx = 0.0f, ly = 0.0f, lz = 0.0f, ldist = 0.0f;
    int ltype = L_DIR;
    if (lane < sc.n_lights) {
        float4 L = sc.lights[lane];
        ltype = __float_as_int(L.w);
        v3 l;
        if (ltype == L_DIR) {
            l = mk(L.x, L.y, L.z);
        } else {
            v3 dv = mk(L.x, L.y, L.z) - position;
            ldist = rt_sqrt(dot(dv, dv));
            l = normalize(dv);
        }
        float d = dot(normal, l);
        my = d > 0 ? d : 0;
        lx = l.x; ly = l.y; lz = l.z;
    }
    float intensity = 0.085f;
    RT_CSTAT(8, 1);
    for (int turn = 0; turn < sc.n_lights; ++turn) {
        // the largest term left that could raise the intensity; among equals the first light
        const float best = wave_extreme<true>(my > intensity ? my : -1.0f);
        if (!(best > intensity)) break;                      // no light left that could raise it
        const int who = __builtin_ctzll(__ballot(my == best));
        auto from = [who](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), who)); };
        const v3 l = mk(from(lx), from(ly), from(lz));
        const int type = __builtin_amdgcn_readlane(ltype, who);
        const float dist = from(ldist);
        const bool occluded = (type == L_DIR) ? any_hit_coop<HAS_NONSPHERE, false>(sc, position, l, 0.0f)
                                              : any_hit_coop<HAS_NONSPHERE, true>(sc, position, l, dist);
        if (!occluded) { intensity = best; break; }
        if (lane == who) my = -1.0f;
    }
    return intensity;
}

// LightsList::GetLightIntensity (lightsList.h:22-40) over DirectionalLight (directionalLight.h:11-20) and the
// repo-defined point light.  The reference asks IsInLight first and the Lambert term second; the result only
// changes when the Lambert term exceeds the running maximum, so with PRUNE the occlusion query is skipped whenever
// it cannot matter (same value either way).  Without PRUNE every query the reference issues is issued and tallied,
// the point light's as the full closest-hit query the reference-side class makes.
template <bool HAS_NONSPHERE, bool COUNT, bool PRUNE, int BATCH = 1>
__device__ __forceinline__ float light_intensity(const SceneLds& sc, v3 position, v3 normal, Counters& cnt)
{
    if (BATCH == RT_COOP && PRUNE && !COUNT) return light_intensity_coop<HAS_NONSPHERE>(sc, position, normal);
    float intensity = 0.085f;
    for (int i = 0; i < sc.n_lights; ++i) {
        float4 L = sc.ulight(i);
        int type = __builtin_amdgcn_readfirstlane(__float_as_int(L.w));
        v3 l; float dist = 0.0f;
        if (type == L_DIR) {
            l = mk(L.x, L.y, L.z);
        } else {
            v3 dv = mk(L.x, L.y, L.z) - position;
            dist = rt_sqrt(dot(dv, dv));
            l = normalize(dv);
        }
        float d = dot(normal, l);
        float intens = d > 0 ? d : 0;
        if (PRUNE) {
            if (intens > intensity) {
                if (COUNT) ++cnt.shadow;
                bool occluded = (type == L_DIR) ? any_hit<HAS_NONSPHERE, COUNT, false, BATCH>(sc, position, l, 0.0f, cnt)
                                                : any_hit<HAS_NONSPHERE, COUNT, true, BATCH>(sc, position, l, dist, cnt);
                if (!occluded) intensity = intens;
            }
        } else {
            if (COUNT) ++cnt.shadow;
            bool occluded;
            if (type == L_DIR) {
                occluded = any_hit<HAS_NONSPHERE, COUNT, false>(sc, position, l, 0.0f, cnt);
            } else {
                float best;
                int bi = closest_hit<HAS_NONSPHERE, COUNT>(sc, position, l, best, cnt);
                occluded = bi >= 0 && best < dist;
            }
            if (!occluded && intens > intensity) intensity = intens;
        }
    }
    return intensity;
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-level candidate lists.  A pixel's four primary rays are the same for every sample (main.cpp:312 passes the same
// u, v), and so are the shadow rays of primary hits.  A hittable whose intersect() is false for every such ray of every
// lane of a wave stays false for all samples, and skipping a test that returns false changes nothing (the reference
// `continue`s, hittablesList.h:25).  So a wave tests all hittables once, keeps those that returned true for some lane,
// and runs its per-sample queries over that list, in list order (ties still go to the earlier hittable).
// Entry j lives in lane j of one VGPR (read back with v_readlane); more than 64 entries: `full`, and the wave keeps
// using the whole scene.  All 64 lanes must be alive while a list is built.
struct WaveList { int items; int n; bool full; };

__device__ __forceinline__ void wavelist_init(WaveList& wl) { wl.items = 0; wl.n = 0; wl.full = false; }

__device__ __forceinline__ void wavelist_offer(WaveList& wl, int i, bool wanted)
{
    if (__ballot(wanted) != 0ull) {
        if (wl.n < 64) { if ((int)(threadIdx.x & 63) == wl.n) wl.items = i; ++wl.n; }
        else wl.full = true;
    }
}

// The list in ascending order of its items (closest-hit queries walk it with the sequential rule: scene order).
__device__ __forceinline__ void wavelist_sort(WaveList& wl)
{
    const int lane = (int)(threadIdx.x & 63);
    int rank = 0;
    for (int e = 0; e < wl.n; ++e) rank += __builtin_amdgcn_readlane(wl.items, e) < wl.items;       // (items are distinct)
    int sorted = 0;
    for (int e = 0; e < wl.n; ++e) {
        const int v = __builtin_amdgcn_readlane(wl.items, e), r = __builtin_amdgcn_readlane(rank, e);
        if (lane == r) sorted = v;
    }
    wl.items = sorted;
}

// Which hittables a wave has listed already, one bit each: lane l holds hittables 64 l .. 64 l + 63 (RT_MAX_HITTABLES = 4096).
struct WaveBits { unsigned long long bits; };
__device__ __forceinline__ bool wavebits_test(const WaveBits& wb, int i)
{
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)wb.bits, i >> 6);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(wb.bits >> 32), i >> 6);
    return ((((unsigned long long)hi << 32) | lo) >> (i & 63)) & 1ull;
}
__device__ __forceinline__ void wavebits_set(WaveBits& wb, int i)
{
    if ((int)(threadIdx.x & 63) == (i >> 6)) wb.bits |= 1ull << (i & 63);
}

// closest_hit / closest_hit_primary over a candidate list
template <bool HAS_NONSPHERE, bool PRIMARY>
__device__ __forceinline__ int closest_hit_listed(const SceneLds& sc, const WaveList& wl, v3 ro, v3 rd, float& best, float fa_known = -1.0f)
{
    float fa = fa_known >= 0.0f ? fa_known : 4.0f * dot(rd, rd);
    v3 inv = mk(0, 0, 0);
    if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
    best = 3.402823466e+38f;
    int bi = -1;
    for (int j = 0; j < wl.n; ++j) {
        const int i = __builtin_amdgcn_readlane(wl.items, j);
        float t;
        bool hit;
        if (PRIMARY && !HAS_NONSPHERE) {
            f32x4 p = sc.kprim[i];
            hit = sphere_hit_oc(mk(p.x, p.y, p.z), p.w, rd, fa, t);
        } else {
            float4 r0 = sc.urec(2 * i), r1 = sc.urec(2 * i + 1);
            hit = hittable_hit<HAS_NONSPHERE>(r0, r1, ro, rd, fa, inv, t);
        }
        if (hit && !(t >= best)) { best = t; bi = i; }
    }
    return bi;
}

template <bool HAS_NONSPHERE, bool LIMITED>
__device__ __forceinline__ bool any_hit_listed(const SceneLds& sc, const WaveList& wl, v3 ro, v3 rd, float limit)
{
    float fa = 4.0f * dot(rd, rd);
    v3 inv = mk(0, 0, 0);
    if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
    for (int j = 0; j < wl.n; ++j) {
        const int i = __builtin_amdgcn_readlane(wl.items, j);
        float4 r0 = sc.urec(2 * i), r1 = sc.urec(2 * i + 1);
        if (occluder_hit<HAS_NONSPHERE, false, LIMITED>(r0, r1, ro, rd, fa, inv, limit)) return true;
    }
    return false;
}

// direction towards light i from `position` (directionalLight.h:9,17; point light: repo-defined), its type and distance
__device__ __forceinline__ v3 light_direction(const SceneLds& sc, int i, v3 position, int& type, float& dist)
{
    float4 L = sc.ulight(i);
    type = __builtin_amdgcn_readfirstlane(__float_as_int(L.w));
    dist = 0.0f;
    if (type == L_DIR) return mk(L.x, L.y, L.z);
    v3 dv = mk(L.x, L.y, L.z) - position;
    dist = rt_sqrt(dot(dv, dv));
    return normalize(dv);
}

// light_intensity<.., PRUNE = true> with the occlusion queries over a candidate list
template <bool HAS_NONSPHERE>
__device__ __forceinline__ float light_intensity_listed(const SceneLds& sc, const WaveList& wl, v3 position, v3 normal)
{
    float intensity = 0.085f;
    for (int i = 0; i < sc.n_lights; ++i) {
        int type; float dist;
        v3 l = light_direction(sc, i, position, type, dist);
        float d = dot(normal, l);
        float intens = d > 0 ? d : 0;
        if (intens > intensity) {
            bool occluded = (type == L_DIR) ? any_hit_listed<HAS_NONSPHERE, false>(sc, wl, position, l, 0.0f)
                                            : any_hit_listed<HAS_NONSPHERE, true>(sc, wl, position, l, dist);
            if (!occluded) intensity = intens;
        }
    }
    return intensity;
}

// The same queries over a candidate list of at most RT_SMALL_LIST hittables whose records sit in scalar registers (the list is
// the wave's, so its records are wave-uniform): what most waves of a frame have -- the ground, one sphere -- without a
// register read, a scalar load and a wait per candidate and ray.  List order, same arithmetic.
#ifndef RT_SMALL_LIST
#define RT_SMALL_LIST 2
#endif
struct SmallList { float4 r0[RT_SMALL_LIST], r1[RT_SMALL_LIST]; int idx[RT_SMALL_LIST]; int n; };

template <bool HAS_NONSPHERE, bool PRIMARY>
__device__ __forceinline__ void smalllist_load(const SceneLds& sc, const WaveList& wl, SmallList& sl)
{
    sl.n = wl.n;
#pragma unroll
    for (int j = 0; j < RT_SMALL_LIST; ++j) {
        const int i = j < wl.n ? __builtin_amdgcn_readlane(wl.items, j) : 0;
        sl.idx[j] = i;
        if (PRIMARY && !HAS_NONSPHERE) {
            f32x4 p = sc.kprim[i];
            sl.r0[j] = make_float4(p.x, p.y, p.z, p.w); sl.r1[j] = sl.r0[j];
        } else {
            sl.r0[j] = sc.urec(2 * i); sl.r1[j] = sc.urec(2 * i + 1);
        }
    }
}

template <bool HAS_NONSPHERE, bool PRIMARY>
__device__ __forceinline__ int closest_hit_small(const SmallList& sl, v3 ro, v3 rd, float fa, float& best)
{
    v3 inv = mk(0, 0, 0);
    if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
    best = 3.402823466e+38f;
    int bi = -1;
#pragma unroll
    for (int j = 0; j < RT_SMALL_LIST; ++j) {
        if (j < sl.n) {
            float t;
            bool hit;
            if (PRIMARY && !HAS_NONSPHERE) hit = sphere_hit_oc(mk(sl.r0[j].x, sl.r0[j].y, sl.r0[j].z), sl.r0[j].w, rd, fa, t);
            else                           hit = hittable_hit<HAS_NONSPHERE>(sl.r0[j], sl.r1[j], ro, rd, fa, inv, t);
            if (hit && !(t >= best)) { best = t; bi = sl.idx[j]; }
        }
    }
    return bi;
}

template <bool HAS_NONSPHERE>
__device__ __forceinline__ float light_intensity_small(const SceneLds& sc, const SmallList& sl, v3 position, v3 normal)
{
    float intensity = 0.085f;
    for (int i = 0; i < sc.n_lights; ++i) {
        int type; float dist;
        v3 l = light_direction(sc, i, position, type, dist);
        float d = dot(normal, l);
        float intens = d > 0 ? d : 0;
        if (intens > intensity) {
            bool occluded = false;
            if (sl.n > 0) {
                float fa = 4.0f * dot(l, l);
                v3 inv = mk(0, 0, 0);
                if (HAS_NONSPHERE) inv = mk(rt_rcp(l.x), rt_rcp(l.y), rt_rcp(l.z));
#pragma unroll
                for (int j = 0; j < RT_SMALL_LIST; ++j)
                    if (j < sl.n && !occluded)
                        occluded = (type == L_DIR) ? occluder_hit<HAS_NONSPHERE, false, false>(sl.r0[j], sl.r1[j], position, l, fa, inv, 0.0f)
                                                   : occluder_hit<HAS_NONSPHERE, false, true>(sl.r0[j], sl.r1[j], position, l, fa, inv, dist);
            }
            if (!occluded) intensity = intens;
        }
    }
    return intensity;
}

// The same queries over a short list of hittable indices in LDS (coop_kernel's primary segment: every lane of the wave
// holds the same ray and walks the same list, in list order -- scene order --, with the sequential replacement rule).
template <bool HAS_NONSPHERE>
__device__ __forceinline__ int closest_hit_ldslist(const SceneLds& sc, const unsigned short* list, int n, v3 ro, v3 rd, float& best)
{
    float fa = 4.0f * dot(rd, rd);
    v3 inv = mk(0, 0, 0);
    if (HAS_NONSPHERE) inv = mk(rt_rcp(rd.x), rt_rcp(rd.y), rt_rcp(rd.z));
    best = 3.402823466e+38f;
    int bi = -1;
    for (int j = 0; j < n; ++j) {
        const int i = list[j];
        float4 r0 = sc.hrec[2 * i], r1 = sc.hrec[2 * i + 1];
        float t;
        if (hittable_hit<HAS_NONSPHERE, false>(r0, r1, ro, rd, fa, inv, t) && !(t >= best)) { best = t; bi = i; }
    }
    return bi;
}

template <bool HAS_NONSPHERE>
__device__ __forceinline__ float light_intensity_ldslist(const SceneLds& sc, const unsigned short* list, int n, v3 position, v3 normal)
{
    float intensity = 0.085f;
    for (int i = 0; i < sc.n_lights; ++i) {
        int type; float dist;
        v3 l = light_direction(sc, i, position, type, dist);
        float d = dot(normal, l);
        float intens = d > 0 ? d : 0;
        if (intens > intensity) {
            float fa = 4.0f * dot(l, l);
            v3 inv = mk(0, 0, 0);
            if (HAS_NONSPHERE) inv = mk(rt_rcp(l.x), rt_rcp(l.y), rt_rcp(l.z));
            bool occluded = false;
            for (int j = 0; j < n && !occluded; ++j) {
                const int h = list[j];
                float4 r0 = sc.hrec[2 * h], r1 = sc.hrec[2 * h + 1];
                occluded = (type == L_DIR) ? occluder_hit<HAS_NONSPHERE, false, false, false>(r0, r1, position, l, fa, inv, 0.0f)
                                           : occluder_hit<HAS_NONSPHERE, false, true, false>(r0, r1, position, l, fa, inv, dist);
            }
            if (!occluded) intensity = intens;
        }
    }
    return intensity;
}

// RANDOM_UNIT_EMISPHERE (core.h:24-35): three draws in x, y, z order, normalise, flip into the normal's hemisphere
template <bool COUNT, class RNG>
__device__ __forceinline__ v3 random_unit_hemisphere(RNG& rng, v3 normal, Counters& cnt)
{
    if (!rng.seeded) rng.do_seed();
    int i0, i1, i2;
    rng.next3(i0, i1, i2);
    if (COUNT) cnt.draws += 3;
    v3 rnd = normalize(mk(rand_unit(i0), rand_unit(i1), rand_unit(i2)));
    if (dot(rnd, normal) > 0) return rnd;
    return -rnd;
}

// refract (core.h:38-48)
__device__ __forceinline__ bool refract_dir(v3 v, v3 n, float ir, v3& refracted)
{
    v3 uv = normalize(v);
    float dt = dot(uv, n);
    float discriminant = 1.0f - ir * ir * (1.0f - dt * dt);
    if (discriminant <= 0) return false;
    refracted = normalize((uv - n * dt) * ir - n * rt_sqrt(discriminant));
    return true;
}

struct Camera {
    float pos[3];
    float ip[16];   // inverse projection, column major
    float iv[16];   // inverse view, column major
};

// UVToDirection (renderer.h:21-25); mat4*vec4 = (m0*v0 + m1*v1) + (m2*v2 + m3*v3) (type_mat4x4.inl:561-572)
__device__ __forceinline__ v3 uv_to_direction(const Camera& c, float u, float v)
{
    float t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        t[r] = (c.ip[r] * u + c.ip[4 + r] * v) + (c.ip[8 + r] * 1.0f + c.ip[12 + r] * 1.0f);
    v3 n = normalize(mk(t[0] / t[3], t[1] / t[3], t[2] / t[3]));
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        d[r] = (c.iv[r] * n.x + c.iv[4 + r] * n.y) + (c.iv[8 + r] * n.z + c.iv[12 + r] * 0.0f);
    return mk(d[0], d[1], d[2]);
}

// pixel::Set for one channel (structs.h:12-17): the unqualified sqrt is the double overload, * 255.0f is a double
// multiply, the conversion truncates.  NaN -> 0 (what the x86 reference binary produces).
__device__ __forceinline__ unsigned char quantise(float c)
{
    double d = sqrt((double)c) * 255.0;
    if (!(d == d)) return 0;
    return (unsigned char)(int)d;
}

}  // namespace RT_NS
