#pragma once

#include "hittable.h"

// HittablesList(hittables, length) -- reference src/hittables/hittablesList.h:8-15: takes ownership of the array's
// objects (deletes them in its destructor, like the reference; the array itself stays the caller's).
class HittablesList : public Hittable
{
public:
    HittablesList(Hittable** hittables, uint32_t length)
        : m_Hittables(hittables), m_Length((int)length) {}

    ~HittablesList() override
    {
        for (int i = 0; i < m_Length; ++i)
            delete m_Hittables[i];
    }

    // children in list order: closest-hit ties go to the earlier one (reference hittablesList.h:27)
    void flatten(RtFlatScene& out) const override
    {
        for (int i = 0; i < m_Length; ++i)
            m_Hittables[i]->flatten(out);
    }

public:
    Hittable** m_Hittables = nullptr;
    int m_Length = -1;
};
