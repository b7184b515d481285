// Per-pixel arithmetic of the INTER_LINEAR resize (device side), shared by k_resize (tvl1_kernels.hip) and the Farneback kernels that
// sample a coarser plane on the fly (farneback_kernels.hip).  Not part of the C-ABI.
#pragma once
#include "miflow/c_api.h"

namespace mi {
namespace tvl1 {

// One destination pixel = column side x row side.  CPU_REF: cv::resize INTER_LINEAR f32 (main repo imgproc/resize.cpp): half-pixel
// centres, coordinates in double -> float, horizontal pass then vertical pass in float.  CUDA_COMPAT: cudawarping/src/cuda/
// resize.cu:234-269.  Split so that a thread evaluates the (double precision) coordinate arithmetic of its four columns once for
// all the rows it produces.
struct RszX { int i0, i1; float w0, w1; };
template <int SEM>
__device__ __forceinline__ RszX resize_xside(int dx, int sw, double scale_x)
{
    RszX r;
    if (SEM == MI_SEM_CPU_REF) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
        r.i0 = sx; r.i1 = min(sx + 1, sw - 1); r.w0 = 1.f - fx; r.w1 = fx;
    } else {
        const float fx = (float)scale_x, src_x = (float)dx * fx;
        const int x1 = (int)floorf(src_x), x2 = x1 + 1;
        r.i0 = x1; r.i1 = min(x2, sw - 1); r.w0 = (float)x2 - src_x; r.w1 = src_x - (float)x1;
    }
    return r;
}
template <int SEM>
__device__ __forceinline__ RszX resize_yside(int dy, int sh, double scale_y)
{
    RszX r;
    if (SEM == MI_SEM_CPU_REF) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= (float)sy;
        r.i0 = min(max(sy, 0), sh - 1); r.i1 = min(max(sy + 1, 0), sh - 1); r.w0 = 1.f - fy; r.w1 = fy;
    } else {
        const float fy = (float)scale_y, src_y = (float)dy * fy;
        const int y1 = (int)floorf(src_y), y2 = y1 + 1;
        r.i0 = y1; r.i1 = min(y2, sh - 1); r.w0 = (float)y2 - src_y; r.w1 = src_y - (float)y1;
    }
    return r;
}
template <int SEM>
__device__ __forceinline__ float resize_combine(const float *R0, const float *R1, const RszX &X, const RszX &Y)
{
    if (SEM == MI_SEM_CPU_REF) {
        const float h0 = R0[X.i0] * X.w0 + R0[X.i1] * X.w1;
        const float h1 = R1[X.i0] * X.w0 + R1[X.i1] * X.w1;
        return h0 * Y.w0 + h1 * Y.w1;
    }
    float out = 0.f;
    out = out + R0[X.i0] * (X.w0 * Y.w0);
    out = out + R0[X.i1] * (X.w1 * Y.w0);
    out = out + R1[X.i0] * (X.w0 * Y.w1);
    out = out + R1[X.i1] * (X.w1 * Y.w1);
    return out;
}


}  // namespace tvl1
}  // namespace mi
