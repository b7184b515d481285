/*
 * oracle/refshim/cudashim/warping_cu_host.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Two kernels of modules/cudawarping run on the CPU through cudashim.h:
 *   resize_linear<T>  (cudawarping/src/cuda/resize.cu:234-269)  -- what cuda::resize(INTER_LINEAR) runs for 1- and 4-channel images,
 *       i.e. the pyramid of cv::cuda::OpticalFlowDual_TVL1 (cudaoptflow/src/tvl1flow.cpp:255-256) and of FarnebackOpticalFlow;
 *   pyrDown<T, B>     (cudawarping/src/cuda/pyr_down.cu:54-175) -- cuda::pyrDown (Farneback's fastPyramids, the 8-bit pyramid of
 *       SparsePyrLKOpticalFlow).
 * The two files instantiate their templates for every vector pixel type through main-repo device headers (vec_traits / vec_math /
 * filters) that are not under /root/reference, so cu2host.py --extract copies just these two function templates, verbatim, into
 * oracle/_ref/warping.gen.inc at build time and this driver instantiates them for T = float and uchar.  Stand-ins below: the
 * single-channel cases of VecTraits / TypeVec / saturate_cast (core/cuda/vec_traits.hpp, saturate_cast.hpp) and
 * __float2int_rd.  Launch geometry as the reference's callers use it (resize.cu:298-308: 32 x 8 threads; pyr_down.cu:178-190:
 * 256 threads, one block row per output row).
 */
#include "opencv2/core/cuda/common.hpp"
#include "opencv2/core/cuda/border_interpolate.hpp"
#include <cmath>

namespace cv { namespace cuda { namespace device {
template <typename T> struct VecTraits;
template <> struct VecTraits<float> { enum { cn = 1 }; typedef float elem_type; static float all(float v) { return v; } };
template <> struct VecTraits<uchar> { enum { cn = 1 }; typedef uchar elem_type; static uchar all(uchar v) { return v; } };
template <typename T, int CN> struct TypeVec;
template <> struct TypeVec<float, 1> { typedef float vec_type; };
template <typename T> static inline T saturate_cast(float v);
template <> inline float saturate_cast<float>(float v) { return v; }
template <> inline uchar saturate_cast<uchar>(float v)   // saturate_cast.hpp: cvt.rni.sat.u8.f32 = round to nearest even, saturate
{
    const float r = nearbyintf(v);
    return (uchar)(r < 0.f ? 0 : r > 255.f ? 255 : (int)r);
}
static inline int __float2int_rd(float v) { return (int)floorf(v); }

#include "warping.gen.inc"

namespace imgproc_host {
template <typename T> void resize_linear_host(const T *src, int sr, int sc, T *dst, int dr, int dc, float fy, float fx)
{
    PtrStepSz<T> s(sr, sc, (T *)src, (size_t)sc * sizeof(T)), d(dr, dc, dst, (size_t)dc * sizeof(T));
    const dim3 block(32, 8);
    const dim3 grid(divUp(dc, block.x), divUp(dr, block.y));
    cudashim::launch(grid, block, 0, [&]() { resize_linear<T>(s, d, fy, fx); });
}
template <typename T> void pyr_down_host(const T *src, int sr, int sc, T *dst, int dr, int dc)
{
    PtrStepSz<T> s(sr, sc, (T *)src, (size_t)sc * sizeof(T)), d(dr, dc, dst, (size_t)dc * sizeof(T));
    const dim3 block(256);
    const dim3 grid(divUp(sc, block.x), dr);
    BrdReflect101<T> b(sr, sc);
    cudashim::launch(grid, block, 0, [&]() { pyrDown<T>(s, (PtrStep<T>)d, b, dc); });
}
}
}}}

using namespace cv::cuda::device::imgproc_host;
extern "C" {
/* fy, fx = the INVERSE scale factors the reference passes (cudawarping/src/resize.cpp:95-96, 102: 1 / fx) */
void ref_cu_resize_linear_f32(const float *src, int sr, int sc, float *dst, int dr, int dc, float fy, float fx) { resize_linear_host<float>(src, sr, sc, dst, dr, dc, fy, fx); }
void ref_cu_resize_linear_u8(const unsigned char *src, int sr, int sc, unsigned char *dst, int dr, int dc, float fy, float fx) { resize_linear_host<unsigned char>(src, sr, sc, dst, dr, dc, fy, fx); }
void ref_cu_pyr_down_f32(const float *src, int sr, int sc, float *dst, int dr, int dc) { pyr_down_host<float>(src, sr, sc, dst, dr, dc); }
void ref_cu_pyr_down_u8(const unsigned char *src, int sr, int sc, unsigned char *dst, int dr, int dc) { pyr_down_host<unsigned char>(src, sr, sc, dst, dr, dc); }
}
