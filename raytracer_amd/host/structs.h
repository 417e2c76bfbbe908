// The data side of the reference's API (src/structs.h:8-94) with the reference's names, layouts and member functions, so
// that a driver written against it -- the recompose loop of src/main.cpp:375-395, the glow's `image[i].Add(...)` of
// src/main.cpp:197, a per-pixel loop that ends in `image[i].Set(...)` (src/main.cpp:318-319) -- compiles unchanged.
// The GPU writes `pixel` / `emissionPixel` buffers directly (rt_device.hpp quantise() is pixel::Set in the kernel); these
// host versions are for code that post-processes or fills such buffers itself.  Nothing here traces a ray.
#pragma once

#include "core.h"

#include <cmath>

// 3 bytes, what writePPM dumps (src/main.cpp:425)
struct pixel
{
    unsigned char x = 0, y = 0, z = 0;

    // src/structs.h:12-17: the unqualified sqrt of a float picks the double overload and `* 255.0f` is then a double
    // multiply; the conversion truncates
    static unsigned char channel(float c) { return (unsigned char)(std::sqrt((double)c) * 255.0); }

    inline void Set(const glm::vec3& c) { x = channel(c.x); y = channel(c.y); z = channel(c.z); }

    // src/structs.h:19-24: the sum is formed in int and saturates at 255
    inline void Add(const glm::vec3& c)
    {
        x = (unsigned char)glm::min((int)x + (int)channel(c.x), 255);
        y = (unsigned char)glm::min((int)y + (int)channel(c.y), 255);
        z = (unsigned char)glm::min((int)z + (int)channel(c.z), 255);
    }
};
static_assert(sizeof(pixel) == 3, "pixel is what the PPM body is made of");

// 16 bytes: vec3 + strength (src/structs.h:28-44)
struct emissionPixel
{
    glm::vec3 emission{ 0.0f, 0.0f, 0.0f };
    float strenght = 0.0f;      // (sic) the reference's spelling

    void Set(const emissionPixel& o) { emission = o.emission; strenght = o.strenght; }
    void Set(const glm::vec3& e, float s) { emission = e; strenght = s; }
};
static_assert(sizeof(emissionPixel) == 16, "emissionPixel = vec3 + float");

// src/structs.h:47-51 (24 bytes)
struct Ray
{
    glm::vec3 origin   { 0.0f, 0.0f,  0.0f };
    glm::vec3 direction{ 0.0f, 0.0f, -1.0f };
};
static_assert(sizeof(Ray) == 24, "Ray = two vec3");

// src/structs.h:54-77 (36 bytes): what Hittable::intersect fills in the reference
struct RayHit
{
    RayHit() {}
    RayHit(float dist) : distance(dist) {}

    void copy(const RayHit& o)
    {
        position = o.position; normal = o.normal; distance = o.distance;
        materialIndx = o.materialIndx; objectIndx = o.objectIndx;
    }

    glm::vec3 position{ 0.0f, 0.0f, 0.0f };
    glm::vec3 normal  { 0.0f, 0.0f, 0.0f };
    float distance = -1.0f;
    int materialIndx = -1, objectIndx = -1;
};
static_assert(sizeof(RayHit) == 36, "RayHit = two vec3, a float, two ints");

// src/structs.h:80-86 (unused by the reference's renderer; kept for the name)
struct TraceInfo
{
    glm::vec3 position{ 0.0f, 0.0f, 0.0f };
    glm::vec3 normal  { 0.0f, 0.0f, 0.0f };
    glm::vec3 color   { 0.0f, 0.0f, 0.0f };
    float roughness = 0.0f;
};

// src/structs.h:89-94 (28 bytes): what AntiAliasing returns per sample
struct HitColorGlow
{
    glm::vec3 color   { 0.0f, 0.0f, 0.0f };
    glm::vec3 emission{ 0.0f, 0.0f, 0.0f };
    float emissionStrenght = 0.0f;
};
static_assert(sizeof(HitColorGlow) == 28, "HitColorGlow = two vec3 + float");
