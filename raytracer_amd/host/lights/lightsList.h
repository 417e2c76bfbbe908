#pragma once

#include "light.h"

// LightsList(lights, length) -- reference src/lights/lightsList.h:8-15: owns the lights it is given.
class LightsList : public Light
{
public:
    LightsList(Light** lights, uint32_t length)
        : m_Lights(lights), m_Length((int)length) {}

    ~LightsList() override
    {
        for (int i = 0; i < m_Length; ++i)
            delete m_Lights[i];
    }

    void flatten(RtFlatScene& out) const override
    {
        for (int i = 0; i < m_Length; ++i)
            m_Lights[i]->flatten(out);
    }

public:
    Light** m_Lights = nullptr;
    int m_Length = -1;
};
