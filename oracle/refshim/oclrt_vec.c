/*
 * oracle/refshim/oclrt_vec.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See oclrt.h.
 * Built-ins whose signatures carry OpenCL vector types: compiled with the same clang that compiles the .cl
 * sources, so that both sides agree on how int2 / float4 / uint4 travel.
 */
#include "oclrt.h"
#include <stdint.h>
typedef int int2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

/* sampler_t: the kernels build CLK_NORMALIZED_COORDS_FALSE | CLK_ADDRESS_CLAMP_TO_EDGE | CLK_FILTER_NEAREST only */
void *oclrt_translate_sampler(int v) __asm__("__translate_sampler_initializer");
void *oclrt_translate_sampler(int v) { return (void *)(intptr_t)v; }

static inline const void *img_at(const oclrt_image2d *im, int x, int y)
{
    x = x < 0 ? 0 : (x > im->width - 1 ? im->width - 1 : x);
    y = y < 0 ? 0 : (y > im->height - 1 ? im->height - 1 : y);
    return (const char *)im->data + (size_t)y * im->step + (size_t)x * (im->elem == 2 ? 1 : 4);
}
float4_t oclrt_read_imagef(const oclrt_image2d *im, void *smp, int2_t c) __asm__("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_i");
float4_t oclrt_read_imagef(const oclrt_image2d *im, void *smp, int2_t c)
{
    (void)smp;
    float4_t r = {0.f, 0.f, 0.f, 1.f};
    r.x = *(const float *)img_at(im, c.x, c.y);
    return r;
}
