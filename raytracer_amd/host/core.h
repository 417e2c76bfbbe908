// Host-side core include of the MI355X backend.  The reference's core.h (src/core.h) switches its headers between
// CPU and CUDA with PREFIX* macros and defines the RNG / refract helpers; here the helpers live in the HIP kernels
// (raytracer_amd/csrc/rt_device.hpp) and the host headers only describe the scene, so this file just provides the
// math types under the names user code already spells (glm::vec3, glm::mat4, glm::normalize, ...).
#pragma once

#include "rt_math.h"

namespace glm = rtm;

// Scene flattening target: every Hittable / Light appends itself in list order (hittablesList.h:27 tie rule).
#include <cstdint>
#include <vector>

struct RtFlatScene {
    std::vector<int32_t> h_type, h_material;
    std::vector<float> h_a, h_b, h_radius;
    std::vector<int32_t> l_type;
    std::vector<float> l_v;

    void hittable(int type, const rtm::vec3& a, const rtm::vec3& b, float radius, int material)
    {
        h_type.push_back(type); h_material.push_back(material); h_radius.push_back(radius);
        h_a.insert(h_a.end(), { a.x, a.y, a.z });
        h_b.insert(h_b.end(), { b.x, b.y, b.z });
    }
    void light(int type, const rtm::vec3& v)
    {
        l_type.push_back(type);
        l_v.insert(l_v.end(), { v.x, v.y, v.z });
    }
};
