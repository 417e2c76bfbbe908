/* TEST INFRASTRUCTURE -- CPU restatement of the reference's glow post-process (src/main.cpp:32-229: downsample,
 * gaussianBlur, upscale, applyGlow), the consumer of the emission buffer the TraceRay path writes.
 *
 * PARITY UNPINNED: src/main.cpp cannot be compiled in the authoring container (it includes the PostgreSQL and Redis
 * client headers, which the image lacks), and the reference's tests hold no fixture for this step, so this file is a
 * restatement checked by reading only.  The reference also ships a CUDA variant (src/main.cu:100-456: separable blur,
 * sigma 20, zero padding) that gives different pixels; the CPU form is the specification here, because the drop-in
 * boundary is the CPU Render() path.
 *
 * Arithmetic as the reference writes it under g++: unqualified exp() / sqrt() on floats are the
 * double overloads; `int kernelSigma = 1000.0f` is the int 1000; glm vec3 operations are per-component float ops.
 * Conversions of out-of-range doubles to unsigned char are undefined in C++; the x86 binary truncates through a 32-bit
 * integer (cvttsd2si) and keeps the low byte, NaN becoming 0 -- the same convention as ro_quantise. */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float e[3]; float s; } epx;          /* emissionPixel (src/structs.h:28-44) */

static unsigned char to_u8(double d)
{
    if (d != d) return 0;
    if (d >= 2147483648.0 || d <= -2147483649.0) return 0;      /* cvttsd2si's "integer indefinite" 0x80000000: low byte 0 */
    return (unsigned char)(int)d;
}

/* pixel::Add (src/structs.h:19-24): x = min(x + (unsigned char)(sqrt(c) * 255.0f), 255), sqrt and the product in double */
static unsigned char px_add(unsigned char x, float c)
{
    int v = (int)x + (int)to_u8(sqrt((double)c) * 255.0);
    return (unsigned char)(v < 255 ? v : 255);
}

/* src/main.cpp:32-75 */
static epx* downsample(const epx* em, int width, int height, int sf)
{
    const int startWidth = width;
    width /= sf; height /= sf;
    epx* result = (epx*)malloc(sizeof(epx) * (size_t)(width > 0 ? width : 1) * (size_t)(height > 0 ? height : 1));
    const float percStep = 1.0f / (sf * sf);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, strength = 0.0f;
            int count = 0;
            for (int nY = 0; nY < sf; ++nY)
                for (int nX = 0; nX < sf; ++nX) {
                    const epx* p = &em[x * sf + nX + (size_t)(y * sf + nY) * startWidth];
                    c0 += p->e[0]; c1 += p->e[1]; c2 += p->e[2];
                    if (p->s > 0) { ++count; strength += p->s; }
                }
            if (count > 0) strength /= count;
            epx* r = &result[x + (size_t)y * width];
            r->e[0] = c0 * percStep; r->e[1] = c1 * percStep; r->e[2] = c2 * percStep; r->s = strength;
        }
    return result;
}

/* the blur's weights (src/main.cpp:101-123): kernel[x + size][y + size], normalised by 1 / sum */
void ro_glow_kernel(float sigma, int size, float* kernel /* (2 size + 1)^2, index (x + size) * (2 size + 1) + (y + size) */)
{
    const int n = 2 * size + 1;
    float sum = 0.0f;
    for (int y = -size; y <= size; ++y)
        for (int x = -size; x <= size; ++x) {
            float value = (float)exp((double)(-(x * x + y * y) / (2 * sigma * sigma)));
            kernel[(x + size) * n + (y + size)] = value;
            sum += value;
        }
    sum = 1.0f / sum;
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) kernel[x * n + y] *= sum;
}

/* src/main.cpp:97-161 */
static void gaussian_blur(epx* em, int width, int height, float sigma, int size)
{
    const int n = 2 * size + 1;
    float* kernel = (float*)malloc(sizeof(float) * (size_t)n * n);
    ro_glow_kernel(sigma, size, kernel);
    epx* tmp = (epx*)malloc(sizeof(epx) * (size_t)(width > 0 ? width : 1) * (size_t)(height > 0 ? height : 1));
    memcpy(tmp, em, sizeof(epx) * (size_t)width * height);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f, strength = 0.0f;
            int count = 0;
            for (int kX = -size; kX <= size; kX++)
                for (int kY = -size; kY <= size; kY++) {
                    int newX = x + kX; newX = newX < 0 ? 0 : newX; newX = newX < width - 1 ? newX : width - 1;      /* min(max(.., 0), width - 1) */
                    int newY = y + kY; newY = newY < 0 ? 0 : newY; newY = newY < height - 1 ? newY : height - 1;
                    const float k = kernel[(kX + size) * n + (kY + size)];
                    const epx* p = &em[newX + (size_t)newY * width];
                    e0 += p->e[0] * k; e1 += p->e[1] * k; e2 += p->e[2] * k;
                    if (p->s > 0) { ++count; strength += p->s; }
                }
            if (count > 0) strength /= count;
            epx* r = &tmp[x + (size_t)y * width];
            r->e[0] = e0; r->e[1] = e1; r->e[2] = e2; r->s = strength;
        }
    memcpy(em, tmp, sizeof(epx) * (size_t)width * height);
    free(tmp); free(kernel);
}

/* applyGlow (src/main.cpp:163-229) on image[width*height] (pixel, 3 bytes) and a copy of emission[width*height] (4 floats).
 * The reference indexes the upscaled level as upscaled[x + y*startW] although its rows are (width/2^l)*2^l long
 * (src/main.cpp:77-95,191-198): the two agree only while the frame's sizes are divisible by the level's factor, and the
 * reference reads out of bounds otherwise.  Returns the number of levels applied, or -1 when a level would need that
 * out-of-bounds read (nothing is written for that level or later ones). */
int ro_apply_glow(unsigned char* image, const float* emission, int width, int height)
{
    float max = 1.0f;
    const int downScaleFactor = 2;
    int upScaleFactor = downScaleFactor;
    const int kernelSigma = (int)1000.0f;
    int kernelSize = 8;
    const int startW = width, startH = height;
    epx* em = (epx*)malloc(sizeof(epx) * (size_t)(width > 0 ? width : 1) * (size_t)(height > 0 ? height : 1));
    memcpy(em, emission, sizeof(epx) * (size_t)width * height);
    int levels = 0;
    while (max >= 1 && width > 0 && height > 0) {
        epx* down = downsample(em, width, height, downScaleFactor);
        free(em);
        const int dW = width / downScaleFactor, dH = height / downScaleFactor;
        if (dW * upScaleFactor != startW || dH * upScaleFactor != startH) { free(down); return -1; }
        gaussian_blur(down, dW, dH, (float)kernelSigma, kernelSize);
        /* upscale (nearest, src/main.cpp:77-95) + combine (src/main.cpp:191-198) */
        for (int y = 0; y < startH; ++y)
            for (int x = 0; x < startW; ++x) {
                const epx* p = &down[(x / upScaleFactor) + (size_t)(y / upScaleFactor) * dW];
                unsigned char* px = &image[3 * ((size_t)x + (size_t)y * startW)];
                px[0] = px_add(px[0], (p->e[0] * 0.1f) * p->s);
                px[1] = px_add(px[1], (p->e[1] * 0.1f) * p->s);
                px[2] = px_add(px[2], (p->e[2] * 0.1f) * p->s);
            }
        /* filter (src/main.cpp:203-218) */
        max = 0.0f;
        for (int y = 0; y < dH; ++y)
            for (int x = 0; x < dW; ++x) {
                epx* p = &down[x + (size_t)y * dW];
                p->s *= 0.65f;
                if (p->s < 1) { p->e[0] = 0.0f; p->e[1] = 0.0f; p->e[2] = 0.0f; }
                max = (max < p->s) ? p->s : max;          /* std::max(max, p.strenght) */
            }
        width /= downScaleFactor; height /= downScaleFactor;
        em = down;
        upScaleFactor *= 2;
        kernelSize *= 2;
        ++levels;
    }
    free(em);
    return levels;
}
