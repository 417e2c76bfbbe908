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

// src/main.cpp:125-156: taps in the reference's order (kX outer, kY inner), clamped coordinates.  A T x T tile of
// level pixels per workgroup; the taps come from LDS: for a run of `cw` consecutive kX the workgroup stages the
// (T + 2 size) x (T + cw - 1) source pixels its threads will read, coordinates clamped as the reference clamps them,
// and every thread adds its taps of that run in the reference's order (runs ascending, kX ascending, kY ascending: the
// same sequence of float additions per output pixel).  One global load per staged pixel instead of one per tap: the
// 4K frame's four levels read 2.2e9 taps.
template <int T>
__global__ __launch_bounds__(T * T) void glow_blur_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int w, int h, int size,
                                                           const float* __restrict__ kernel, int cw)
{
    extern __shared__ float4 tile[];
    // the run's weights behind the tile (LDS like the taps: scalar loads would share the LDS reads' wait counter and
    // return out of order, so every wait would be a wait for everything)
    float* wts = (float*)(tile + (T + 2 * size) * (T + cw - 1));
    const int tx = threadIdx.x % T, ty = threadIdx.x / T;
    const int bx = blockIdx.x * T, by = blockIdx.y * T;
    const int x = bx + tx, y = by + ty;
    const kptrf kk = (kptrf)kernel;
    const int n = 2 * size + 1;
    const int rows = T + 2 * size, pitch = T + cw - 1;
    float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f, strength = 0.0f;
    int count = 0;
    for (int k0 = -size; k0 <= size; k0 += cw) {
        const int run = k0 + cw - 1 <= size ? cw : size - k0 + 1;
        const int cols = T + run - 1;
        __syncthreads();                                          // (the previous run's taps have been read)
        // (thread (tx, ty) stages columns tx, tx + T, .. of rows ty, ty + T, ..: four rows' loads in flight)
        for (int c = tx; c < cols; c += T) {
            int sx = bx + k0 + c; sx = sx < 0 ? 0 : sx; sx = sx < w - 1 ? sx : w - 1;
#pragma unroll 4
            for (int r = ty; r < rows; r += T) {
                int sy = by - size + r; sy = sy < 0 ? 0 : sy; sy = sy < h - 1 ? sy : h - 1;
                tile[r * pitch + c] = src[(size_t)sx + (size_t)sy * w];
            }
        }
        for (int i = threadIdx.x; i < run * n; i += T * T) wts[i] = kk[(size_t)(k0 + size) * n + i];
        __syncthreads();
        for (int kX = k0; kX < k0 + run; ++kX) {
            const float* krow = wts + (kX - k0) * n;
            const float4* col = tile + ty * pitch + tx + (kX - k0);
            // j = kY + size; eight taps' reads in flight, added in order
            int j = 0;
            for (; j + 8 <= n; j += 8) {
                float4 p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = col[(j + u) * pitch];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float k = krow[j + u];
                    e0 += p[u].x * k; e1 += p[u].y * k; e2 += p[u].z * k;
                    if (p[u].w > 0) { ++count; strength += p[u].w; }
                }
            }
            for (; j < n; ++j) {
                const float k = krow[j];
                const float4 p = col[j * pitch];
                e0 += p.x * k; e1 += p.y * k; e2 += p.z * k;
                if (p.w > 0) { ++count; strength += p.w; }
            }
        }
    }
    if (x >= w || y >= h) return;
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
    // small levels (large windows) in 8 x 8 tiles: four times the workgroups for the few pixels they have
    const int T = size >= 32 ? 8 : 16;
    const int rows = T + 2 * size, n = 2 * size + 1;
    // columns of kX per staged run: tile (rows x (T + cw - 1) pixels of 16 B) + weights (cw x n floats) within an LDS
    // budget that lets every workgroup of the level be resident at once where that is possible (a thread's taps are one
    // serial chain of additions: a second round of workgroups doubles the level's time), 12 .. 48 KiB
    const int blocks = ((w + T - 1) / T) * ((h + T - 1) / T);
    const int per_cu = (blocks + 255) / 256;
    int budget = (T != 8 || per_cu > 8) ? 48 * 1024 : 160 * 1024 / per_cu - 1024;   // (the large levels' 256-thread workgroups: fewer, longer runs measured faster)
    budget = budget < 12 * 1024 ? 12 * 1024 : budget > 48 * 1024 ? 48 * 1024 : budget;
    int cw = (budget - rows * (T - 1) * 16) / (rows * 16 + n * 4);
    cw = cw < 1 ? 1 : cw > n ? n : cw;
    const size_t lds = (size_t)rows * (size_t)(T + cw - 1) * 16 + (size_t)cw * n * 4;
    if (lds > 64 * 1024) return hipErrorInvalidValue;             // (size <= 128: rt_apply_glow refuses larger windows)
    if (T == 8) hipLaunchKernelGGL(glow_blur_kernel<8>, dim3((w + 7) / 8, (h + 7) / 8), dim3(64), lds, stream, (const float4*)src, (float4*)dst, w, h, size, kernel, cw);
    else        hipLaunchKernelGGL(glow_blur_kernel<16>, dim3((w + 15) / 16, (h + 15) / 16), dim3(256), lds, stream, (const float4*)src, (float4*)dst, w, h, size, kernel, cw);
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
