/* TEST INFRASTRUCTURE -- CPU restatement of the reference's TraceRay hot path (see rt_oracle.h).
 *
 * Every function cites the reference file:line it follows.  All arithmetic is float32 in the reference's
 * operation order (GLM 0.9.9.8 pure-scalar mode); build with -ffp-contract=off.  Recursion is kept
 * (like renderer.h:28-135) so that colour and emission are combined on unwind exactly as the reference does.
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
/* glm compute_dot<vec3>: tmp = a*b; tmp.x + tmp.y + tmp.z  (vendor/glm/detail/func_geometric.inl:48-55) */
static inline float v3_dot(v3 a, v3 b) { v3 t = v3_mul(a, b); return t.x + t.y + t.z; }
/* glm normalize: v * inversesqrt(dot(v,v)), inversesqrt = 1/sqrt (func_geometric.inl:88, func_exponential.inl:138) */
static inline v3 v3_normalize(v3 a) { return v3_scale(a, 1.0f / sqrtf(v3_dot(a, a))); }
/* glm reflect: I - N * dot(N, I) * 2 (func_geometric.inl:108) */
static inline v3 v3_reflect(v3 I, v3 N) { return v3_sub(I, v3_scale(v3_scale(N, v3_dot(N, I)), 2.0f)); }
/* glm min/max (func_common.inl:20,29) */
static inline float f_min(float x, float y) { return (y < x) ? y : x; }
static inline float f_max(float x, float y) { return (x < y) ? y : x; }
static inline float f_clamp01(float x) { return f_min(f_max(x, 0.0f), 1.0f); }
static inline v3 v3_cross(v3 x, v3 y)
{
    return v3_make(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}

/* ------------------------------------------------------------------ RNG: glibc srand/rand, TYPE_3 (rand.h:7,9) */
void ro_srand(ro_rng* g, unsigned int seed)
{
    if (seed == 0) seed = 1;
    g->r[0] = seed;
    int word = (int)seed;
    for (int i = 1; i < 31; ++i) {
        long hi = word / 127773;
        long lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int)w;
        g->r[i] = (unsigned int)word;
    }
    g->f = 3; g->b = 0;
    for (int i = 0; i < 310; ++i) (void)ro_rand(g);
}

int ro_rand(ro_rng* g)
{
    unsigned int val = (g->r[g->f] += g->r[g->b]);
    int result = (int)(val >> 1);
    if (++g->f >= 31) { g->f = 0; ++g->b; }
    else if (++g->b >= 31) g->b = 0;
    return result;
}

/* ------------------------------------------------------------------ camera (camera.h:11-22 + GLM) */
static void mat4_inverse(const float m[4][4], float out[4][4])
{
    /* vendor/glm/detail/func_matrix.inl:294-351, m[col][row] */
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

    float Fac0[4] = { Coef00, Coef00, Coef02, Coef03 };
    float Fac1[4] = { Coef04, Coef04, Coef06, Coef07 };
    float Fac2[4] = { Coef08, Coef08, Coef10, Coef11 };
    float Fac3[4] = { Coef12, Coef12, Coef14, Coef15 };
    float Fac4[4] = { Coef16, Coef16, Coef18, Coef19 };
    float Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };
    float Vec0[4] = { m[1][0], m[0][0], m[0][0], m[0][0] };
    float Vec1[4] = { m[1][1], m[0][1], m[0][1], m[0][1] };
    float Vec2[4] = { m[1][2], m[0][2], m[0][2], m[0][2] };
    float Vec3[4] = { m[1][3], m[0][3], m[0][3], m[0][3] };
    static const float SignA[4] = { +1, -1, +1, -1 };
    static const float SignB[4] = { -1, +1, -1, +1 };
    float Inverse[4][4];
    for (int i = 0; i < 4; ++i) {
        float Inv0 = Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i] + Vec3[i] * Fac2[i];
        float Inv1 = Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i] + Vec3[i] * Fac4[i];
        float Inv2 = Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i] + Vec3[i] * Fac5[i];
        float Inv3 = Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i] + Vec2[i] * Fac5[i];
        Inverse[0][i] = Inv0 * SignA[i];
        Inverse[1][i] = Inv1 * SignB[i];
        Inverse[2][i] = Inv2 * SignA[i];
        Inverse[3][i] = Inv3 * SignB[i];
    }
    float Dot0[4] = { m[0][0] * Inverse[0][0], m[0][1] * Inverse[1][0], m[0][2] * Inverse[2][0], m[0][3] * Inverse[3][0] };
    float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            out[c][r] = Inverse[c][r] * OneOverDeterminant;
}

void ro_camera_init(ro_camera* cam, float fov_deg, float width, float height, float znear, float zfar)
{
    /* camera.h:12 */
    v3 eye = v3_make(-3.0f, 6.0f, 10.0f);
    cam->pos[0] = eye.x; cam->pos[1] = eye.y; cam->pos[2] = eye.z;

    /* camera.h:15: perspectiveFov(radians(fov), w, h, near, far) -> perspectiveFovRH_NO
     * (vendor/glm/ext/matrix_clip_space.inl:372-389, vendor/glm/detail/func_trigonometric.inl:13) */
    float rad = fov_deg * (float)0.01745329251994329576923690768489;
    float h = cosf(0.5f * rad) / sinf(0.5f * rad);
    float w = h * height / width;
    float P[4][4]; memset(P, 0, sizeof P);
    P[0][0] = w;
    P[1][1] = h;
    P[2][2] = -(zfar + znear) / (zfar - znear);
    P[2][3] = -1.0f;
    P[3][2] = -(2.0f * zfar * znear) / (zfar - znear);
    float iP[4][4];
    mat4_inverse(P, iP);

    /* camera.h:19-20: lookAt(eye, (0,3,0), (0,1,0)) -> lookAtRH (vendor/glm/ext/matrix_transform.inl:99-119) */
    v3 center = v3_make(0.0f, 3.0f, 0.0f), up = v3_make(0.0f, 1.0f, 0.0f);
    v3 f = v3_normalize(v3_sub(center, eye));
    v3 s = v3_normalize(v3_cross(f, up));
    v3 u = v3_cross(s, f);
    float V[4][4]; memset(V, 0, sizeof V);
    V[0][0] = 1.0f; V[1][1] = 1.0f; V[2][2] = 1.0f; V[3][3] = 1.0f;
    V[0][0] = s.x; V[1][0] = s.y; V[2][0] = s.z;
    V[0][1] = u.x; V[1][1] = u.y; V[2][1] = u.z;
    V[0][2] = -f.x; V[1][2] = -f.y; V[2][2] = -f.z;
    V[3][0] = -v3_dot(s, eye);
    V[3][1] = -v3_dot(u, eye);
    V[3][2] = v3_dot(f, eye);
    float iV[4][4];
    mat4_inverse(V, iV);

    memcpy(cam->inv_proj, iP, sizeof iP);
    memcpy(cam->inv_view, iV, sizeof iV);
}

/* ------------------------------------------------------------------ prepared scene */
typedef struct { int type; v3 a, b; float radius; int material; } hittable;
typedef struct { int type; v3 v; } light;

typedef struct {
    const ro_material* materials;
    hittable* hittables; int n_hittables;
    light* lights; int n_lights;
    const ro_camera* cam;
    int maxdepth;
} prepared;

typedef struct {
    const prepared* p;
    ro_rng rng;
    ro_counters cnt;
} ctx;

typedef struct { v3 position, normal; float distance; int material; int object; } rayhit;
typedef struct { v3 color, emission; float strength; } hcg;

static void prepare(prepared* p, const ro_scene* s, const ro_camera* cam, int maxdepth)
{
    p->materials = s->materials;
    p->n_hittables = s->n_hittables; p->n_lights = s->n_lights;
    p->hittables = (hittable*)malloc(sizeof(hittable) * (s->n_hittables > 0 ? s->n_hittables : 1));
    p->lights = (light*)malloc(sizeof(light) * (s->n_lights > 0 ? s->n_lights : 1));
    for (int i = 0; i < s->n_hittables; ++i) {
        const ro_hittable* h = &s->hittables[i];
        hittable* o = &p->hittables[i];
        o->type = h->type;
        o->a = v3_make(h->a[0], h->a[1], h->a[2]);
        o->b = v3_make(h->b[0], h->b[1], h->b[2]);
        o->radius = h->radius;
        o->material = h->material;
        if (h->type == RO_PLANE) o->b = v3_normalize(o->b);            /* plane.h:11 */
        if (h->type != RO_CUBE) o->material = (unsigned char)h->material; /* sphere.h:10, plane.h:10 take uint8_t */
    }
    for (int i = 0; i < s->n_lights; ++i) {
        p->lights[i].type = s->lights[i].type;
        v3 v = v3_make(s->lights[i].v[0], s->lights[i].v[1], s->lights[i].v[2]);
        p->lights[i].v = s->lights[i].type == RO_DIRLIGHT ? v3_normalize(v3_neg(v)) : v;  /* directionalLight.h:9 */
    }
    p->cam = cam;
    p->maxdepth = maxdepth;
}

static void unprepare(prepared* p) { free(p->hittables); free(p->lights); }

/* ------------------------------------------------------------------ intersections */
/* sphere.h:13-51 */
static int sphere_intersect(const hittable* s, v3 ro, v3 rd, rayhit* hit)
{
    v3 origin = v3_sub(ro, s->a);
    float a = v3_dot(rd, rd);
    float b = 2.0f * v3_dot(origin, rd);
    float c = v3_dot(origin, origin) - s->radius * s->radius;
    float discriminant = b * b - 4.0f * a * c;
    if (discriminant < 0.0f) return 0;
    float sqrtd = sqrtf(discriminant);
    float t1 = (-b - sqrtd) / 2.0f;
    float t2 = (-b + sqrtd) / 2.0f;
    if (t1 < 0 && t2 < 0) return 0;
    if (t2 > t1) { if (t1 >= 0) hit->distance = t1; else hit->distance = t2; }
    else         { if (t2 >= 0) hit->distance = t2; else hit->distance = t1; }
    hit->position = v3_add(origin, v3_scale(rd, hit->distance));
    hit->normal = v3_normalize(hit->position);
    hit->position = v3_add(hit->position, s->a);
    hit->material = s->material;
    return 1;
}

/* plane.h:13-28 */
static int plane_intersect(const hittable* pl, v3 ro, v3 rd, rayhit* hit)
{
    float denom = v3_dot(pl->b, rd);
    if ((double)fabsf(denom) <= 1e-6) return 0;
    float t = v3_dot(v3_sub(ro, pl->a), pl->b) / denom;
    if (t < 0) return 0;
    hit->distance = t;
    hit->position = v3_add(ro, v3_scale(rd, t));
    hit->normal = pl->b;
    hit->material = pl->material;
    return 1;
}

/* cube.h:13-49 */
static int cube_intersect(const hittable* cu, v3 ro, v3 rd, rayhit* hit)
{
    v3 invDir = v3_make(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    v3 t0 = v3_mul(v3_sub(cu->a, ro), invDir);
    v3 t1 = v3_mul(v3_sub(v3_add(cu->a, cu->b), ro), invDir);
    v3 tmin = v3_make(f_min(t0.x, t1.x), f_min(t0.y, t1.y), f_min(t0.z, t1.z));
    v3 tmax = v3_make(f_max(t0.x, t1.x), f_max(t0.y, t1.y), f_max(t0.z, t1.z));
    float tMin = f_max(f_max(tmin.x, tmin.y), tmin.z);
    float tMax = f_min(f_min(tmax.x, tmax.y), tmax.z);
    if (tMax < 0 || tMin > tMax) return 0;
    v3 hp = v3_add(ro, v3_scale(rd, tMin));
    v3 n;
    if      ((double)fabsf(hp.x - cu->a.x) < 1e-4)             n = v3_make(-1, 0, 0);
    else if ((double)fabsf(hp.x - (cu->a.x + cu->b.x)) < 1e-4) n = v3_make(1, 0, 0);
    else if ((double)fabsf(hp.y - cu->a.y) < 1e-4)             n = v3_make(0, -1, 0);
    else if ((double)fabsf(hp.y - (cu->a.y + cu->b.y)) < 1e-4) n = v3_make(0, 1, 0);
    else if ((double)fabsf(hp.z - cu->a.z) < 1e-4)             n = v3_make(0, 0, -1);
    else                                                       n = v3_make(0, 0, 1);
    hit->distance = tMin;
    hit->position = hp;
    hit->normal = n;
    hit->material = cu->material;
    return 1;
}

static inline int hittable_intersect(ctx* c, const hittable* h, v3 ro, v3 rd, rayhit* hit)
{
    ++c->cnt.tests;
    switch (h->type) {
    case RO_SPHERE: return sphere_intersect(h, ro, rd, hit);
    case RO_PLANE:  return plane_intersect(h, ro, rd, hit);
    default:        return cube_intersect(h, ro, rd, hit);
    }
}

/* hittablesList.h:17-37 */
static int world_intersect(ctx* c, v3 ro, v3 rd, rayhit* hit)
{
    const prepared* p = c->p;
    hit->distance = FLT_MAX;
    rayhit cur; memset(&cur, 0, sizeof cur); cur.distance = -1.0f; cur.material = -1; cur.object = -1;
    int has = 0;
    for (int i = 0; i < p->n_hittables; ++i) {
        if (!hittable_intersect(c, &p->hittables[i], ro, rd, &cur) || cur.distance >= hit->distance) continue;
        *hit = cur;
        hit->object = i;
        has = 1;
    }
    return has;
}

/* hittablesList.h:39-50 (calls intersect, not hasIntersect) */
static int world_has_intersect(ctx* c, v3 ro, v3 rd)
{
    const prepared* p = c->p;
    rayhit hit;
    for (int i = 0; i < p->n_hittables; ++i)
        if (hittable_intersect(c, &p->hittables[i], ro, rd, &hit)) return 1;
    return 0;
}

/* ------------------------------------------------------------------ lights */
/* lightsList.h:22-40, directionalLight.h:11-20, and the repo-defined point light (ref_harness.cpp) */
static float light_intensity(ctx* c, v3 position, v3 normal)
{
    const prepared* p = c->p;
    float intensity = 0.085f;
    for (int i = 0; i < p->n_lights; ++i) {
        const light* L = &p->lights[i];
        float intens;
        if (L->type == RO_DIRLIGHT) {
            ++c->cnt.shadow;
            if (world_has_intersect(c, position, L->v)) continue;
            float d = v3_dot(normal, L->v);
            intens = d > 0 ? d : 0;
        } else {
            v3 dv = v3_sub(L->v, position);
            float dist = sqrtf(v3_dot(dv, dv));
            v3 l = v3_normalize(dv);
            rayhit hit;
            ++c->cnt.shadow;
            int h = world_intersect(c, position, l, &hit);
            if (h && hit.distance < dist) continue;
            v3 l2 = v3_normalize(v3_sub(L->v, position));
            float d = v3_dot(normal, l2);
            intens = d > 0 ? d : 0;
        }
        if (intens > intensity) intensity = intens;
    }
    return intensity;
}

/* ------------------------------------------------------------------ random helpers (core.h:20-35, rand.h:9) */
static inline float random_unit(ctx* c)
{
    ++c->cnt.draws;
    return (float)ro_rand(&c->rng) / (float)2147483647;   /* (float)RAND_MAX == 2147483648.0f */
}

static v3 random_unit_hemisphere(ctx* c, v3 normal)
{
    float x = random_unit(c);
    float y = random_unit(c);
    float z = random_unit(c);
    v3 rnd = v3_normalize(v3_make(x, y, z));
    if (v3_dot(rnd, normal) > 0) return rnd;
    return v3_neg(rnd);
}

/* core.h:38-48 */
static int refract_dir(v3 v, v3 n, float ir, v3* refracted)
{
    v3 uv = v3_normalize(v);
    float dt = v3_dot(uv, n);
    float discriminant = 1.0f - ir * ir * (1.0f - dt * dt);
    if (discriminant <= 0) return 0;
    *refracted = v3_normalize(v3_sub(v3_scale(v3_sub(uv, v3_scale(n, dt)), ir), v3_scale(n, sqrtf(discriminant))));
    return 1;
}

/* ------------------------------------------------------------------ TraceRay (renderer.h:28-135) */
static hcg trace_ray(ctx* c, v3 ro, v3 rd, float multiplier, int depth, int* maxDepth)
{
    const prepared* p = c->p;
    hcg out; memset(&out, 0, sizeof out);
    if (multiplier < 0.001f) { *maxDepth = depth; return out; }

    rayhit hit;
    ++c->cnt.closest;
    if (!world_intersect(c, ro, rd, &hit)) {
        *maxDepth = depth;
        float a = (rd.y + 1.0f) * 0.5f;
        v3 white = v3_scale(v3_make(1.0f, 1.0f, 1.0f), 1.0f - a);
        v3 blue = v3_scale(v3_make(0.2f, 0.3f, 0.8f), a);
        out.color = v3_scale(v3_add(white, blue), multiplier);
        return out;
    }

    v3 offPosition = v3_add(hit.position, v3_scale(hit.normal, 0.005f));
    float intensity = light_intensity(c, offPosition, hit.normal);

    const ro_material* m = &p->materials[hit.material];
    v3 color = v3_scale(v3_scale(v3_make(m->color[0], m->color[1], m->color[2]), intensity), multiplier);
    v3 emission = v3_make(m->emission[0], m->emission[1], m->emission[2]);
    float strength = m->glow;

    if (depth <= p->maxdepth) {
        v3 rayDir = v3_normalize(rd);
        if (m->reflection > 0) {
            v3 d = v3_reflect(rd, hit.normal);
            d = v3_normalize(v3_add(d, v3_scale(random_unit_hemisphere(c, hit.normal), m->roughness)));
            rd = d;
            ro = offPosition;
            if (v3_dot(d, hit.normal) > 0) {
                hcg h = trace_ray(c, offPosition, d, multiplier * m->reflection, depth + 1, maxDepth);
                color = v3_add(color, h.color);
                if (h.strength > 0) {
                    v3 e = v3_add(v3_scale(emission, strength), v3_scale(h.emission, h.strength));
                    emission = v3_normalize(e);
                    strength = f_max(strength, h.strength * m->reflection * 1.25f);
                }
            }
        }
        if (m->refraction > 0) {
            v3 outNorm; float ir;
            if (v3_dot(rayDir, hit.normal) > 0) { outNorm = v3_neg(hit.normal); ir = m->refraction; }
            else                                { outNorm = hit.normal;         ir = 1.0f / m->refraction; }
            v3 d;
            if (!refract_dir(rayDir, outNorm, ir, &d)) d = v3_reflect(rayDir, hit.normal);
            d = v3_normalize(v3_add(d, v3_scale(random_unit_hemisphere(c, hit.normal), m->roughness)));
            v3 o = v3_add(hit.position, v3_scale(d, 0.01f));
            hcg h = trace_ray(c, o, d, multiplier * 0.9f, depth, maxDepth);
            color = v3_add(color, h.color);
            if (h.strength > 0) {
                v3 e = v3_add(v3_scale(emission, strength), v3_scale(h.emission, h.strength));
                emission = v3_normalize(e);
                strength = f_max(strength, h.strength * 0.95f);
            }
        }
        if (m->reflection == 0 && m->refraction == 0) *maxDepth = depth;
    } else {
        *maxDepth = depth;
    }
    out.color = color; out.emission = emission; out.strength = strength;
    return out;
}

/* renderer.h:21-25 with mat4*vec4 = (m0*v0 + m1*v1) + (m2*v2 + m3*v3) (vendor/glm/detail/type_mat4x4.inl:561-572) */
static void mat4_mul_vec4(const float* m, const float v[4], float out[4])
{
    for (int r = 0; r < 4; ++r) {
        float add0 = m[0 * 4 + r] * v[0] + m[1 * 4 + r] * v[1];
        float add1 = m[2 * 4 + r] * v[2] + m[3 * 4 + r] * v[3];
        out[r] = add0 + add1;
    }
}

static v3 uv_to_direction(const ro_camera* cam, float u, float v)
{
    float in[4] = { u, v, 1.0f, 1.0f }, target[4];
    mat4_mul_vec4(cam->inv_proj, in, target);
    v3 t = v3_make(target[0] / target[3], target[1] / target[3], target[2] / target[3]);
    v3 n = v3_normalize(t);
    float in2[4] = { n.x, n.y, n.z, 0.0f }, d[4];
    mat4_mul_vec4(cam->inv_view, in2, d);
    return v3_make(d[0], d[1], d[2]);
}

/* renderer.h:138-201 */
static hcg anti_aliasing(ctx* c, float u, float v, float offX, float offY)
{
    const ro_camera* cam = c->p->cam;
    v = -v;
    hcg result; memset(&result, 0, sizeof result);
    v3 origin = v3_make(cam->pos[0], cam->pos[1], cam->pos[2]);
    const float du[4] = { -offX, +offX, -offX, +offX };
    const float dv[4] = { -offY, -offY, +offY, +offY };
    for (int k = 0; k < 4; ++k) {
        int maxDepth = 0;
        v3 dir = uv_to_direction(cam, u + du[k], v + dv[k]);
        ++c->cnt.primary;
        hcg info = trace_ray(c, origin, dir, 1.0f, 1, &maxDepth);
        float md = (float)maxDepth;
        result.color = v3_add(result.color, v3_make(info.color.x / md, info.color.y / md, info.color.z / md));
        result.emission = v3_add(result.emission, info.emission);
        result.strength = k == 0 ? info.strength : f_max(result.strength, info.strength);
    }
    result.color = v3_scale(result.color, 0.25f);
    result.emission = v3_scale(result.emission, 0.25f);
    return result;
}

/* structs.h:12-17: unqualified sqrt(float) resolves to ::sqrt(double); * 255.0f is then a double multiply;
 * the cast truncates.  NaN -> 0 (what x86 cvttsd2si + low byte yields for the reference binary). */
unsigned char ro_quantise(float c)
{
    double d = sqrt((double)c) * 255.0;
    if (d != d) return 0;
    return (unsigned char)(int)d;
}

/* main.cpp:298-319 */
static void render_pixel(ctx* c, int x, int y, int W, int H, int samples, long seed_base,
                         unsigned char* rgb, float* em)
{
    float u = ((float)x / (float)W) * 2.0f - 1.0f;
    float v = ((float)y / (float)H) * 2.0f - 1.0f;
    ro_srand(&c->rng, (unsigned int)(seed_base + (long)(x + y * W)));
    float offX = 0.5f / W, offY = 0.5f / H;
    hcg acc; memset(&acc, 0, sizeof acc);
    for (int i = 0; i < samples; ++i) {
        hcg s = anti_aliasing(c, u, v, offX, offY);
        acc.color = v3_add(acc.color, v3_make(f_clamp01(s.color.x), f_clamp01(s.color.y), f_clamp01(s.color.z)));
        acc.emission = v3_add(acc.emission, v3_make(f_clamp01(s.emission.x), f_clamp01(s.emission.y), f_clamp01(s.emission.z)));
        acc.strength += s.strength;
    }
    float n = (float)samples;
    if (rgb) {
        rgb[0] = ro_quantise(acc.color.x / n);
        rgb[1] = ro_quantise(acc.color.y / n);
        rgb[2] = ro_quantise(acc.color.z / n);
    }
    if (em) {
        em[0] = acc.emission.x / n; em[1] = acc.emission.y / n; em[2] = acc.emission.z / n;
        em[3] = acc.strength / samples;
    }
}

/* ------------------------------------------------------------------ driver: 16x16 tiles on a small pthread pool */
typedef struct {
    const prepared* p;
    int W, H, samples, row0, rows, row_step; long seed_base;
    unsigned char* rgb; float* emission;
    atomic_int next_tile; int tiles_x, tiles_y;
    ro_counters total; pthread_mutex_t lock;
} job;

static void* worker(void* arg)
{
    job* j = (job*)arg;
    ctx c; memset(&c, 0, sizeof c); c.p = j->p;
    for (;;) {
        int t = atomic_fetch_add(&j->next_tile, 1);
        if (t >= j->tiles_x * j->tiles_y) break;
        int sX = t % j->tiles_x, sY = t / j->tiles_x;
        for (int nY = 0; nY < 16; ++nY)
            for (int nX = 0; nX < 16; ++nX) {
                int x = nX + sX * 16, yl = nY + sY * 16;
                if (x >= j->W || yl >= j->rows) continue;
                int y = j->row0 + yl * j->row_step;
                size_t o = (size_t)x + (size_t)yl * j->W;
                render_pixel(&c, x, y, j->W, j->H, j->samples, j->seed_base,
                             j->rgb ? j->rgb + o * 3 : NULL, j->emission ? j->emission + o * 4 : NULL);
            }
    }
    pthread_mutex_lock(&j->lock);
    j->total.primary += c.cnt.primary; j->total.closest += c.cnt.closest; j->total.shadow += c.cnt.shadow;
    j->total.tests += c.cnt.tests; j->total.draws += c.cnt.draws;
    pthread_mutex_unlock(&j->lock);
    return NULL;
}

int ro_render(const ro_scene* scene, const ro_camera* cam, int W, int H, int samples, int maxdepth,
              long seed_base, int row0, int rows, int threads,
              unsigned char* rgb, float* emission, ro_counters* counters)
{
    return ro_render_strided(scene, cam, W, H, samples, maxdepth, seed_base, row0, rows, 1, threads, rgb, emission, counters);
}

int ro_render_strided(const ro_scene* scene, const ro_camera* cam, int W, int H, int samples, int maxdepth,
                      long seed_base, int row0, int rows, int row_step, int threads,
                      unsigned char* rgb, float* emission, ro_counters* counters)
{
    if (!scene || !cam || W <= 0 || H <= 0 || samples <= 0 || row0 < 0 || rows < 0 || row_step < 1) return -1;
    if (rows > 0 && row0 + (rows - 1) * row_step >= H) return -1;
    for (int i = 0; i < scene->n_hittables; ++i) {
        int m = scene->hittables[i].material;
        if (scene->hittables[i].type != RO_CUBE) m = (unsigned char)m;
        if (m < 0 || m >= scene->n_materials) return -1;
    }
    prepared p; prepare(&p, scene, cam, maxdepth);
    job j; memset(&j, 0, sizeof j);
    j.p = &p; j.W = W; j.H = H; j.samples = samples; j.row0 = row0; j.rows = rows; j.row_step = row_step; j.seed_base = seed_base;
    j.rgb = rgb; j.emission = emission;
    j.tiles_x = (W + 15) / 16; j.tiles_y = (rows + 15) / 16;
    atomic_init(&j.next_tile, 0);
    pthread_mutex_init(&j.lock, NULL);
    if (threads <= 1) {
        worker(&j);
    } else {
        pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
        for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, worker, &j);
        for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
        free(th);
    }
    pthread_mutex_destroy(&j.lock);
    if (counters) *counters = j.total;
    unprepare(&p);
    return 0;
}

/* tests/singlePixel.cpp:65-90 -> renderer.h:204-219 */
int ro_probe_max_depth(const ro_scene* scene, const ro_camera* cam, int W, int H, int maxdepth,
                       long seed_base, unsigned char* flags)
{
    if (!scene || !cam || !flags || W <= 0 || H <= 0) return -1;
    prepared p; prepare(&p, scene, cam, maxdepth);
    ctx c; memset(&c, 0, sizeof c); c.p = &p;
    v3 origin = v3_make(cam->pos[0], cam->pos[1], cam->pos[2]);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float u = ((float)x / (float)W) * 2.0f - 1.0f;
            float v = ((float)y / (float)H) * 2.0f - 1.0f;
            ro_srand(&c.rng, (unsigned int)(seed_base + (long)(x + y * W)));
            v = -v;
            int maxDepth = 0;
            v3 dir = uv_to_direction(cam, u, v);
            (void)trace_ray(&c, origin, dir, 1.0f, 1, &maxDepth);
            flags[x + (size_t)y * W] = maxDepth >= maxdepth;
        }
    unprepare(&p);
    return 0;
}
