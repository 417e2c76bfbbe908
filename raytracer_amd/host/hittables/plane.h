#pragma once

#include "hittable.h"

// Plane(center, normal, materialIndx) -- reference src/hittables/plane.h:10-11.  The normal is passed on as given;
// the library normalises it exactly like the reference constructor does (rt_upload_scene).
class Plane : public Hittable
{
public:
    Plane(const glm::vec3& center, const glm::vec3& normal, uint8_t materialIndx)
        : m_Center(center), m_Normal(normal), m_MaterialIndx(materialIndx) {}

    void flatten(RtFlatScene& out) const override { out.hittable(1 /*RT_PLANE*/, m_Center, m_Normal, 0.0f, m_MaterialIndx); }

    const glm::vec3& center() const { return m_Center; }
    const glm::vec3& normal() const { return m_Normal; }
    int material() const { return m_MaterialIndx; }

private:
    glm::vec3 m_Center, m_Normal;
    int m_MaterialIndx;
};
