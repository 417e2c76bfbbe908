#pragma once

#include "hittable.h"

// Cube(position, size, materialIndx) -- reference src/hittables/cube.h:10-11: min corner + edge lengths.  The
// reference class is abstract (it never defines hasIntersect) and therefore unused there; here it is a regular
// scene object and the kernels implement its slab test (rt_device.hpp cube_hit).
class Cube : public Hittable
{
public:
    Cube(const glm::vec3& position, const glm::vec3& size, uint32_t materialIndx)
        : m_Position(position), m_Size(size), m_MaterialIndx(materialIndx) {}

    void flatten(RtFlatScene& out) const override { out.hittable(2 /*RT_CUBE*/, m_Position, m_Size, 0.0f, (int)m_MaterialIndx); }

    const glm::vec3& position() const { return m_Position; }
    const glm::vec3& size() const { return m_Size; }
    uint32_t material() const { return m_MaterialIndx; }

private:
    glm::vec3 m_Position, m_Size;
    uint32_t m_MaterialIndx;
};
