// Output PODs with the reference's layouts (src/structs.h:8-44): `pixel` is 3 bytes (what writePPM dumps,
// src/main.cpp:425), `emissionPixel` is 16 bytes (vec3 + strength).  The GPU writes both buffers directly;
// pixel::Set's double-precision sqrt quantisation happens in the kernel (rt_device.hpp quantise()).
#pragma once

#include "core.h"

struct pixel
{
    unsigned char x = 0, y = 0, z = 0;
};
static_assert(sizeof(pixel) == 3, "pixel is what the PPM body is made of");

struct emissionPixel
{
    glm::vec3 emission{ 0.0f, 0.0f, 0.0f };
    float strenght = 0.0f;      // (sic) the reference's spelling
};
static_assert(sizeof(emissionPixel) == 16, "emissionPixel = vec3 + float");
