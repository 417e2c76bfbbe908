// Material: the reference's 40-byte POD (src/material.h:5-11), field for field, so a Material[] can be handed to
// rt_scene_soa::materials as is.
#pragma once

#include "core.h"

struct Material
{
    glm::vec3 color;
    float roughness, reflection, refraction;
    glm::vec3 emissionColor;
    float glowStrength;
};
static_assert(sizeof(Material) == 40, "Material must stay 10 packed floats (rt_scene_soa::materials)");
