// Light: base of the light sources (reference src/lights/light.h:8-14).  IsInLight / GetLightIntensity run in the HIP
// kernels (rt_device.hpp light_intensity); a Light appends its description to the flattened scene.  The reference's two
// virtuals keep their names and signatures (see hittables/hittable.h: no host implementation on purpose).
#pragma once

#include "../core.h"
#include "../hittables/hittable.h"

class Light
{
public:
    virtual ~Light() = default;
    virtual void flatten(RtFlatScene& out) const = 0;

    virtual bool IsInLight(Hittable**, const glm::vec3&) const { rt_host_call_not_available("Light::IsInLight"); }
    virtual void GetLightIntensity(Hittable**, const glm::vec3&, const glm::vec3&, float&) const { rt_host_call_not_available("Light::GetLightIntensity"); }
};
