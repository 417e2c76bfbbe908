// Light: base of the light sources (reference src/lights/light.h).  IsInLight / GetLightIntensity run in the HIP
// kernels (rt_device.hpp light_intensity); a Light only appends its description to the flattened scene.
#pragma once

#include "../core.h"

class Light
{
public:
    virtual ~Light() = default;
    virtual void flatten(RtFlatScene& out) const = 0;
};
