#pragma once

#include "light.h"

// PointLight(position) -- not in the reference (BASELINE config C4 asks for point lights).  In light <=> no hittable
// is hit strictly closer than the light along normalize(position - p); intensity max(0, n.l).  See DESIGN.md.
class PointLight : public Light
{
public:
    PointLight(const glm::vec3& position) : m_Position(position) {}

    void flatten(RtFlatScene& out) const override { out.light(1 /*RT_POINTLIGHT*/, m_Position); }

    const glm::vec3& position() const { return m_Position; }

private:
    glm::vec3 m_Position;
};
