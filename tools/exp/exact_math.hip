// Exhaustive check of short, correctly rounded sqrt / reciprocal sequences against the compiler's IEEE sqrtf / division
// over all 2^32 float bit patterns.  hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/exp/exact_math.hip -o /tmp/exact_math
// Prints, per variant, the number of mismatching inputs and the smallest / largest |x| among them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float cand_sqrt(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    s = (rm <= 0.0f) ? sm : s;
    s = (rp > 0.0f) ? sp : s;
    return s;
}

// one Newton step on v_rcp_f32
__device__ __forceinline__ float cand_rcp1(float y)
{
    float r = __builtin_amdgcn_rcpf(y);
    const float e = __builtin_fmaf(-y, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// ... and a residual correction with the refined value
__device__ __forceinline__ float cand_rcp2(float y)
{
    const float r0 = __builtin_amdgcn_rcpf(y);
    const float e = __builtin_fmaf(-y, r0, 1.0f);
    const float r1 = __builtin_fmaf(e, r0, r0);
    const float e1 = __builtin_fmaf(-y, r1, 1.0f);
    return __builtin_fmaf(e1, r1, r1);
}

// 1 / sqrt(x) as the reference computes it: RN(1 / RN(sqrt(x)))
__device__ __forceinline__ float cand_rsqrt(float x) { return cand_rcp2(cand_sqrt(x)); }

struct Stat { unsigned long long bad; unsigned int lo, hi; };

__device__ void note(Stat* st, unsigned int bits)
{
    atomicAdd(&st->bad, 1ull);
    const unsigned int a = bits & 0x7fffffffu;
    atomicMin(&st->lo, a);
    atomicMax(&st->hi, a);
}

__device__ bool same(float a, float b)
{
    if (a != a && b != b) return true;
    return __float_as_uint(a) == __float_as_uint(b);
}

__global__ void sweep(Stat* st)
{
    const unsigned long long n = 1ull << 32;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned int bits = (unsigned int)i;
        const float x = __uint_as_float(bits);
        if (!same(cand_sqrt(x), sqrtf(x))) note(&st[0], bits);
        if (!same(cand_rcp1(x), 1.0f / x)) note(&st[1], bits);
        if (!same(cand_rcp2(x), 1.0f / x)) note(&st[2], bits);
        if (!same(cand_rsqrt(x), 1.0f / sqrtf(x))) note(&st[3], bits);
    }
}

int main()
{
    Stat h[4];
    for (auto& s : h) { s.bad = 0; s.lo = 0xffffffffu; s.hi = 0; }
    Stat* d;
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sweep, dim3(256 * 32), dim3(256), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = { "sqrt (v_sqrt + two residual tests)", "rcp, one Newton step", "rcp, two steps", "1/sqrt from both" };
    for (int k = 0; k < 4; ++k) {
        float lo, hi;
        std::memcpy(&lo, &h[k].lo, 4); std::memcpy(&hi, &h[k].hi, 4);
        printf("%-36s mismatches %llu", names[k], h[k].bad);
        if (h[k].bad) printf("  |x| in [%a, %a]  (bits %08x .. %08x)", lo, hi, h[k].lo, h[k].hi);
        printf("\n");
    }
    return 0;
}
