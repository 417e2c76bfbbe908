// Hittable: base of the scene objects (reference src/hittables/hittable.h).  In the reference it carries the
// virtual intersect()/hasIntersect() that TraceRay calls per ray; here the intersection code is in the HIP kernels
// and a Hittable only knows how to append its description to the flattened scene.
#pragma once

#include "../core.h"

class Hittable
{
public:
    virtual ~Hittable() = default;
    virtual void flatten(RtFlatScene& out) const = 0;
};
