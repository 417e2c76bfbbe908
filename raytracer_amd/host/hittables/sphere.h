#pragma once

#include "hittable.h"

// Sphere(position, radius, materialIndx) -- reference src/hittables/sphere.h:10 (material index is a uint8_t there too)
class Sphere : public Hittable
{
public:
    Sphere(const glm::vec3& position, const float radius, uint8_t materialIndx)
        : m_Position(position), m_Radius(radius), m_MaterialIndx(materialIndx) {}

    void flatten(RtFlatScene& out) const override { out.hittable(0 /*RT_SPHERE*/, m_Position, glm::vec3(0.0f), m_Radius, m_MaterialIndx); }

    const glm::vec3& position() const { return m_Position; }
    float radius() const { return m_Radius; }
    int material() const { return m_MaterialIndx; }

private:
    glm::vec3 m_Position;
    float m_Radius;
    int m_MaterialIndx;
};
