// curandState of the reference's CPU build (src/rand.h:4-10): srand(time(NULL) + seed) at construction, randomUnit() =
// rand() / RAND_MAX from the process-wide libc generator.  Host code that seeds per pixel the way src/main.cpp:304 does
// compiles against this; the GPU kernels do not use it -- they carry one glibc TYPE_3 generator per pixel, seeded
// seed_base + x + y * WIDTH with rt_params.seed_base standing in for time(NULL) (rt_device.hpp RngT).
#pragma once

#include <cstdlib>
#include <ctime>

class curandState
{
public:
    inline curandState(int seed) { std::srand((unsigned int)(std::time(NULL) + seed)); }

    inline float randomUnit() { return (float)std::rand() / (float)RAND_MAX; }
};
