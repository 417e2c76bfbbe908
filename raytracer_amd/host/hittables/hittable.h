// Hittable: base of the scene objects (reference src/hittables/hittable.h:8-14).  In the reference its virtual
// intersect() / hasIntersect() are what TraceRay calls per ray; here the intersection code is in the HIP kernels
// (rt_device.hpp sphere_hit / plane_hit / cube_hit) and a Hittable's job is to append its description to the flattened
// scene.  The two virtuals keep their names and signatures so that code which passes Hittable* around or overrides
// them compiles; there is deliberately NO host-side implementation behind them (that would be a CPU fallback): the
// defaults stop the program with a message.
#pragma once

#include "../core.h"
#include "../structs.h"

#include <cstdio>
#include <cstdlib>

[[noreturn]] inline void rt_host_call_not_available(const char* what)
{
    std::fprintf(stderr, "%s runs in the GPU kernels of this backend (librt_amd.so); there is no host implementation -- "
                         "render through Render() / RenderImage() / RenderTiles() / MaxRayDistanceAny() (renderer.h)\n", what);
    std::abort();
}

class Hittable
{
public:
    virtual ~Hittable() = default;
    virtual void flatten(RtFlatScene& out) const = 0;

    virtual bool intersect(const Ray&, RayHit&) const { rt_host_call_not_available("Hittable::intersect"); }
    virtual bool hasIntersect(const Ray&) const { rt_host_call_not_available("Hittable::hasIntersect"); }
};
