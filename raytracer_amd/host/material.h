// Material: the reference's 40-byte aggregate (src/material.h:5-11).  Field order and types are the interface -- user
// code brace-initialises it and the array goes to rt_scene_soa::materials as it is -- so they are kept; nothing else is.
#pragma once

#include "core.h"

struct Material
{
    glm::vec3 color;            // floats 0-2: albedo, scaled by the light intensity (renderer.h:64)
    float roughness;            // float 3: weight of the random hemisphere vector added to a bounce direction
    float reflection;           // float 4: > 0 spawns the reflection ray; multiplier of its contribution
    float refraction;           // float 5: > 0 spawns the refraction ray; the index of refraction
    glm::vec3 emissionColor;    // floats 6-8
    float glowStrength;         // float 9
};
static_assert(sizeof(Material) == 10 * sizeof(float), "Material must stay 10 packed floats (rt_scene_soa::materials)");
