// "rtscene v1" text reader for the host side: builds the reference-style object graph (HittablesList of
// Sphere/Plane/Cube, LightsList, Material[]) from a file whose records mirror the reference's Postgres tables
// material / sphere / directional_light (src/postgres.h:183-228), replacing getMaterials / getWorld / getLights
// (src/postgres.h:54-129) without a database.  Format: see raytracer_amd/scenes.py; rows of those tables with their
// VEC3 columns in readVec3's "(x, y, z)" syntax are read too (rt_scan_table_row below).
#pragma once

#include "material.h"
#include "hittables/hittablesList.h"
#include "hittables/sphere.h"
#include "hittables/plane.h"
#include "hittables/cube.h"
#include "lights/lightsList.h"
#include "lights/directionalLight.h"
#include "lights/pointLight.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct LoadedScene
{
    std::vector<Material> materials;
    std::vector<Hittable*> hittables;   // owned by `world` once built
    std::vector<Light*> lightList;      // owned by `lights` once built
    Hittable* world = nullptr;
    Light* lights = nullptr;
    float fov = 60.0f, znear = 0.01f, zfar = 1000.0f;

    ~LoadedScene() { delete world; delete lights; }
};

// A row of the reference's tables (src/postgres.h:183-228) in table-column order, VEC3 columns written "(x, y, z)" as
// readVec3 (src/postgres.h:232-256) takes them, columns separated by blanks, '|' or ',' (quotes and ::VEC3 casts are
// skipped, so the reference's own INSERT rows and a psql dump both read):
//     material <id> (color) <roughness> <reflection> <refraction> (emission_color) <glow_strength>
//     sphere <id> (position) <radius> <material_id> <scene_id>
//     directional_light <id> (direction) <scene_id>
// cols receives the columns flattened (a VEC3 is three values), shape one char per column ('v' VEC3, 's' scalar).
inline bool rt_scan_table_row(const char* p, std::vector<float>& cols, std::string& shape)
{
    cols.clear(); shape.clear();
    for (;;) {
        while (*p == ' ' || *p == '\t' || *p == ',' || *p == '|' || *p == '\'' || *p == '\n' || *p == '\r') ++p;
        if (!std::strncmp(p, "::VEC3", 6)) { p += 6; continue; }
        if (*p == 0 || *p == '#') return true;
        if (*p == '(') {
            ++p;
            for (int k = 0; k < 3; ++k) {
                char* e; const float v = (float)std::strtod(p, &e);          // atof of the part, as readVec3
                if (e == p) return false;
                cols.push_back(v); p = e;
                while (*p == ' ') ++p;
                if (*p != (k < 2 ? ',' : ')')) return false;
                ++p;
            }
            shape += 'v';
        } else {
            char* e; const float v = std::strtof(p, &e);
            if (e == p) return false;
            cols.push_back(v); p = e; shape += 's';
        }
    }
}

// Returns false (and prints the offending line) on a malformed file.  sceneId: table rows of other scenes are skipped
// (getWorld / getLights select by scene_id, src/postgres.h:88,113).
inline bool LoadScene(const char* path, LoadedScene& out, int sceneId = 0)
{
    FILE* f = std::fopen(path, "r");
    if (!f) { std::fprintf(stderr, "scene: cannot open %s\n", path); return false; }
    char line[1024];
    int lineno = 0;
    bool ok = true;
    while (ok && std::fgets(line, sizeof line, f)) {
        ++lineno;
        char* p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (*p == '#' || *p == '\n' || *p == '\r' || *p == 0) continue;
        char kw[32]; int used = 0;
        if (std::sscanf(p, "%31s%n", kw, &used) != 1) continue;
        p += used;
        const std::string k(kw);
        if ((k == "material" || k == "sphere" || k == "directional_light") && std::strchr(p, '(')) {
            std::vector<float> c; std::string shape;
            if (!rt_scan_table_row(p, c, shape)) { std::fprintf(stderr, "scene: %s:%d: bad table row '%s'\n", path, lineno, kw); ok = false; }
            else if (k == "material" && shape == "svsssvs")
                out.materials.push_back(Material{ { c[1], c[2], c[3] }, c[4], c[5], c[6], { c[7], c[8], c[9] }, c[10] });
            else if (k == "sphere" && shape == "svsss") { if ((int)c[6] == sceneId) out.hittables.push_back(new Sphere({ c[1], c[2], c[3] }, c[4], (uint8_t)(int)c[5])); }
            else if (k == "directional_light" && shape == "svs") { if ((int)c[4] == sceneId) out.lightList.push_back(new DirectionalLight({ c[1], c[2], c[3] })); }
            else { std::fprintf(stderr, "scene: %s:%d: bad table row '%s'\n", path, lineno, kw); ok = false; }
            continue;
        }
        float v[10]; int n = 0;
        for (; n < 10; ++n) { char* e; v[n] = std::strtof(p, &e); if (e == p) break; p = e; }
        if (k == "material" && n == 10)
            out.materials.push_back(Material{ { v[0], v[1], v[2] }, v[3], v[4], v[5], { v[6], v[7], v[8] }, v[9] });
        else if (k == "sphere" && n == 5) out.hittables.push_back(new Sphere({ v[0], v[1], v[2] }, v[3], (uint8_t)(int)v[4]));
        else if (k == "plane" && n == 7)  out.hittables.push_back(new Plane({ v[0], v[1], v[2] }, { v[3], v[4], v[5] }, (uint8_t)(int)v[6]));
        else if (k == "cube" && n == 7)   out.hittables.push_back(new Cube({ v[0], v[1], v[2] }, { v[3], v[4], v[5] }, (uint32_t)(int)v[6]));
        else if (k == "dirlight" && n == 3)   out.lightList.push_back(new DirectionalLight({ v[0], v[1], v[2] }));
        else if (k == "pointlight" && n == 3) out.lightList.push_back(new PointLight({ v[0], v[1], v[2] }));
        else if (k == "camera" && n == 3) { out.fov = v[0]; out.znear = v[1]; out.zfar = v[2]; }
        else { std::fprintf(stderr, "scene: %s:%d: bad record '%s'\n", path, lineno, kw); ok = false; }
    }
    std::fclose(f);
    if (!ok) {
        for (Hittable* h : out.hittables) delete h;
        for (Light* l : out.lightList) delete l;
        out.hittables.clear(); out.lightList.clear();
        return false;
    }
    out.world = new HittablesList(out.hittables.data(), (uint32_t)out.hittables.size());
    out.lights = new LightsList(out.lightList.data(), (uint32_t)out.lightList.size());
    return true;
}
