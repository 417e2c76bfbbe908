#pragma once

#include "light.h"

// DirectionalLight(direction) -- reference src/lights/directionalLight.h:8-9.  The constructor argument is passed on
// unchanged; the library stores normalize(-direction) like the reference constructor.
class DirectionalLight : public Light
{
public:
    DirectionalLight(const glm::vec3& direction) : m_Given(direction) {}

    void flatten(RtFlatScene& out) const override { out.light(0 /*RT_DIRLIGHT*/, m_Given); }

    const glm::vec3& direction() const { return m_Given; }

private:
    glm::vec3 m_Given;
};
