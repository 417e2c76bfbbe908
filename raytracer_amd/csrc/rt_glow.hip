// Glow post-process on the emission buffer: the reference's CPU form, src/main.cpp:32-229 --
// downsample by 2, a (2 size + 1)^2 two-dimensional gaussian with sigma 1000 (size 8, doubling per level; edge pixels
// clamped), nearest-neighbour upscale, pixel::Add of emission * 0.1 * strength, strength *= 0.65 with emission zeroed
// below 1, until no strength reaches 1.  (The CUDA form, src/main.cu:100-456, is a different filter: separable, sigma
// 20, zero padding.  The CPU path is what this library stands in for.)
//
// Every output value is formed by the same float operations in the same order as the reference's loops, so the kernels
// are bit-identical to the CPU restatement in oracle/rt_glow.c; what is parallel is the pixel loop.  One thread per
// output pixel everywhere.  The blur reads its (2 size + 1)^2 taps straight from global memory: a level's image is at
// most a quarter of the frame and stays in L2, neighbouring lanes read neighbouring pixels, and the weights are
// wave-uniform scalar loads.  Work per level is about width * height * 72 taps whatever the level (the image shrinks
// by 4, the window grows by 4), a few hundred microseconds at 4K.
#include "rt_launch.h"

namespace rt {

typedef float f32x4g __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(4)))* kptrf;      // constant address space: uniform loads become s_load

// src/main.cpp:32-75 with scaleFactor 2
__global__ __launch_bounds__(256) void glow_downsample_kernel(const float4* __restrict__ src, int srcW, float4* __restrict__ dst, int dW, int dH)
{
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= dW || y >= dH) return;
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, strength = 0.0f;
    int count = 0;
#pragma unroll
    for (int nY = 0; nY < 2; ++nY)
#pragma unroll
        for (int nX = 0; nX < 2; ++nX) {
            const float4 p = src[(size_t)(x * 2 + nX) + (size_t)(y * 2 + nY) * srcW];
            c0 += p.x; c1 += p.y; c2 += p.z;
            if (p.w > 0) { ++count; strength += p.w; }
        }
    if (count > 0) strength /= count;
    const float percStep = 1.0f / (2 * 2);
    dst[(size_t)x + (size_t)y * dW] = make_float4(c0 * percStep, c1 * percStep, c2 * percStep, strength);
}

// src/main.cpp:125-156: taps in the reference's order (kX outer, kY inner), clamped coordinates
__global__ __launch_bounds__(256) void glow_blur_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int w, int h, int size,
                                                         const float* __restrict__ kernel)
{
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    const kptrf kk = (kptrf)kernel;
    const int n = 2 * size + 1;
    float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f, strength = 0.0f;
    int count = 0;
    for (int kX = -size; kX <= size; ++kX) {
        int newX = x + kX; newX = newX < 0 ? 0 : newX; newX = newX < w - 1 ? newX : w - 1;
        const kptrf krow = kk + (size_t)(kX + size) * n + size;
#pragma unroll 8
        for (int kY = -size; kY <= size; ++kY) {                 // (unrolled: eight taps' loads in flight, added in order)
            int newY = y + kY; newY = newY < 0 ? 0 : newY; newY = newY < h - 1 ? newY : h - 1;
            const float k = krow[kY];
            const float4 p = src[(size_t)newX + (size_t)newY * w];
            e0 += p.x * k; e1 += p.y * k; e2 += p.z * k;
            if (p.w > 0) { ++count; strength += p.w; }
        }
    }
    if (count > 0) strength /= count;
    dst[(size_t)x + (size_t)y * w] = make_float4(e0, e1, e2, strength);
}

// pixel::Add (src/structs.h:19-24): the conversion truncates through a 32-bit integer like the x86 binary, NaN -> 0
__device__ __forceinline__ unsigned char glow_add(unsigned char x, float c)
{
    const double d = sqrt((double)c) * 255.0;
    int q = 0;
    if (d == d && d < 2147483648.0 && d > -2147483649.0) q = (int)d;
    const int v = (int)x + (int)(unsigned char)q;
    return (unsigned char)(v < 255 ? v : 255);
}

// upscale (src/main.cpp:77-95) and combine (src/main.cpp:191-198) in one pass over the frame
__global__ __launch_bounds__(256) void glow_add_kernel(unsigned char* __restrict__ rgb, const float4* __restrict__ down, int dW, int up, int startW, int startH)
{
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= startW || y >= startH) return;
    const float4 p = down[(size_t)(x / up) + (size_t)(y / up) * dW];
    unsigned char* px = rgb + 3 * ((size_t)x + (size_t)y * startW);
    px[0] = glow_add(px[0], (p.x * 0.1f) * p.w);
    px[1] = glow_add(px[1], (p.y * 0.1f) * p.w);
    px[2] = glow_add(px[2], (p.z * 0.1f) * p.w);
}

// filter (src/main.cpp:203-218); *max_bits receives the largest strength above 0 as float bits (0 stands for the
// reference's starting value 0.0f; NaNs and negatives never replace it, as with std::max(max, p.strenght))
__global__ __launch_bounds__(256) void glow_filter_kernel(float4* __restrict__ down, int n, unsigned int* __restrict__ max_bits)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float s = 0.0f;
    if (i < n) {
        float4 p = down[i];
        p.w *= 0.65f;
        if (p.w < 1) { p.x = 0.0f; p.y = 0.0f; p.z = 0.0f; }
        down[i] = p;
        s = p.w;
    }
    unsigned int bits = (s > 0.0f) ? __float_as_uint(s) : 0u;        // positive floats order like their bit patterns
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const unsigned int o = (unsigned int)__shfl_xor((int)bits, m); bits = o > bits ? o : bits; }
    if ((threadIdx.x & 63) == 0 && bits) atomicMax(max_bits, bits);
}

}  // namespace rt

using namespace rt;

hipError_t rt_launch_glow_downsample(const void* src, int srcW, void* dst, int dW, int dH, hipStream_t stream)
{
    if (dW <= 0 || dH <= 0) return hipSuccess;
    hipLaunchKernelGGL(glow_downsample_kernel, dim3((dW + 15) / 16, (dH + 15) / 16), dim3(256), 0, stream, (const float4*)src, srcW, (float4*)dst, dW, dH);
    return hipGetLastError();
}

hipError_t rt_launch_glow_blur(const void* src, void* dst, int w, int h, int size, const float* kernel, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return hipSuccess;
    hipLaunchKernelGGL(glow_blur_kernel, dim3((w + 15) / 16, (h + 15) / 16), dim3(256), 0, stream, (const float4*)src, (float4*)dst, w, h, size, kernel);
    return hipGetLastError();
}

hipError_t rt_launch_glow_add(void* rgb, const void* down, int dW, int up, int startW, int startH, hipStream_t stream)
{
    hipLaunchKernelGGL(glow_add_kernel, dim3((startW + 15) / 16, (startH + 15) / 16), dim3(256), 0, stream, (unsigned char*)rgb, (const float4*)down, dW, up, startW, startH);
    return hipGetLastError();
}

hipError_t rt_launch_glow_filter(void* down, int n, unsigned int* max_bits, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(glow_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (float4*)down, n, max_bits);
    return hipGetLastError();
}
