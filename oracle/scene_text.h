/* TEST INFRASTRUCTURE (oracle side) -- never linked into the product library.
 *
 * Plain-C reader of the "rtscene v1" text format shared by the two oracle
 * programs (ref_harness.cpp, which compiles the reference headers, and
 * rt_oracle.c, the CPU restatement).  The product has its own reader
 * (raytracer_amd/host/scene_io.h); the format itself is this repo's invention
 * and mirrors the reference's Postgres tables material / sphere /
 * directional_light (/root/reference/src/postgres.h:183-228).
 *
 *   material  r g b  roughness reflection refraction  er eg eb  glow
 *   sphere    x y z  radius  material
 *   plane     cx cy cz  nx ny nz  material
 *   cube      px py pz  sx sy sz  material
 *   dirlight  dx dy dz
 *   pointlight x y z
 *   camera    fov near far          (optional, default 60 0.01 1000)
 *
 * Numbers are parsed with strtof, so "%.9g" text round-trips float32 exactly.
 * Hittables keep their file order: closest-hit ties go to the earlier one
 * (/root/reference/src/hittables/hittablesList.h:27).
 */
#ifndef ORACLE_SCENE_TEXT_H
#define ORACLE_SCENE_TEXT_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { ST_SPHERE = 0, ST_PLANE = 1, ST_CUBE = 2 };
enum { ST_DIRLIGHT = 0, ST_POINTLIGHT = 1 };

typedef struct { float color[3], roughness, reflection, refraction, emission[3], glow; } st_material;
typedef struct { int type; float a[3]; float b[3]; float radius; int material; } st_hittable;
typedef struct { int type; float v[3]; } st_light;

typedef struct {
    st_material* materials; int n_materials;
    st_hittable* hittables; int n_hittables;
    st_light*    lights;    int n_lights;
    float fov, znear, zfar;
} st_scene;

static int st_read_floats(char** p, float* out, int n)
{
    for (int i = 0; i < n; ++i) {
        char* end;
        out[i] = strtof(*p, &end);
        if (end == *p) return 0;
        *p = end;
    }
    return 1;
}

static int st_load(const char* path, st_scene* s)
{
    memset(s, 0, sizeof(*s));
    s->fov = 60.0f; s->znear = 0.01f; s->zfar = 1000.0f;
    FILE* f = fopen(path, "r");
    if (!f) { fprintf(stderr, "scene: cannot open %s\n", path); return 0; }
    int cap_m = 0, cap_h = 0, cap_l = 0;
    char line[1024];
    int lineno = 0;
    while (fgets(line, sizeof line, f)) {
        ++lineno;
        char* p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (*p == '#' || *p == '\n' || *p == '\r' || *p == 0) continue;
        char kw[32]; int used = 0;
        if (sscanf(p, "%31s%n", kw, &used) != 1) continue;
        p += used;
        float v[10];
        int ok = 1;
        if (!strcmp(kw, "material")) {
            ok = st_read_floats(&p, v, 10);
            if (ok) {
                if (s->n_materials == cap_m) { cap_m = cap_m ? cap_m * 2 : 8; s->materials = (st_material*)realloc(s->materials, cap_m * sizeof(st_material)); }
                st_material* m = &s->materials[s->n_materials++];
                m->color[0] = v[0]; m->color[1] = v[1]; m->color[2] = v[2];
                m->roughness = v[3]; m->reflection = v[4]; m->refraction = v[5];
                m->emission[0] = v[6]; m->emission[1] = v[7]; m->emission[2] = v[8];
                m->glow = v[9];
            }
        } else if (!strcmp(kw, "sphere") || !strcmp(kw, "plane") || !strcmp(kw, "cube")) {
            int type = !strcmp(kw, "sphere") ? ST_SPHERE : !strcmp(kw, "plane") ? ST_PLANE : ST_CUBE;
            int nf = type == ST_SPHERE ? 5 : 7;
            ok = st_read_floats(&p, v, nf);
            if (ok) {
                if (s->n_hittables == cap_h) { cap_h = cap_h ? cap_h * 2 : 64; s->hittables = (st_hittable*)realloc(s->hittables, cap_h * sizeof(st_hittable)); }
                st_hittable* h = &s->hittables[s->n_hittables++];
                memset(h, 0, sizeof(*h));
                h->type = type;
                h->a[0] = v[0]; h->a[1] = v[1]; h->a[2] = v[2];
                if (type == ST_SPHERE) { h->radius = v[3]; h->material = (int)v[4]; }
                else { h->b[0] = v[3]; h->b[1] = v[4]; h->b[2] = v[5]; h->material = (int)v[6]; }
            }
        } else if (!strcmp(kw, "dirlight") || !strcmp(kw, "pointlight")) {
            ok = st_read_floats(&p, v, 3);
            if (ok) {
                if (s->n_lights == cap_l) { cap_l = cap_l ? cap_l * 2 : 8; s->lights = (st_light*)realloc(s->lights, cap_l * sizeof(st_light)); }
                st_light* l = &s->lights[s->n_lights++];
                l->type = !strcmp(kw, "dirlight") ? ST_DIRLIGHT : ST_POINTLIGHT;
                l->v[0] = v[0]; l->v[1] = v[1]; l->v[2] = v[2];
            }
        } else if (!strcmp(kw, "camera")) {
            ok = st_read_floats(&p, v, 3);
            if (ok) { s->fov = v[0]; s->znear = v[1]; s->zfar = v[2]; }
        } else {
            fprintf(stderr, "scene: %s:%d: unknown keyword '%s'\n", path, lineno, kw);
            fclose(f);
            return 0;
        }
        if (!ok) { fprintf(stderr, "scene: %s:%d: malformed '%s' record\n", path, lineno, kw); fclose(f); return 0; }
    }
    fclose(f);
    return 1;
}

static void st_free(st_scene* s)
{
    free(s->materials); free(s->hittables); free(s->lights);
    memset(s, 0, sizeof(*s));
}

#endif
