// Orientation and descriptors of the reference's CPU SURF class on the GPU (xfeatures2d::SURF_Impl, surf.cpp:568-866): the
// arithmetic the reference's known-answer vectors were produced with.  One workgroup per keypoint; the phases live in surfcpu_dev.h,
// which also compiles for the host where tests check it bit for bit against oracle/surfcpu_ref.c.  cv::cuda::SURF_CUDA's own
// sampling (surf_kernels.hip) stays the default; these entry points serve callers that need the CPU class's numbers.
#include "mi_common.h"
#include "surfcpu_dev.h"

namespace mi {
namespace surfcpu {

// keypoint matrix rows of cv::cuda::SURF_CUDA (xfeatures2d/cuda.hpp:89-99)
enum { X_ROW = 0, Y_ROW, LAPLACIAN_ROW, OCTAVE_ROW, SIZE_ROW, ANGLE_ROW, HESSIAN_ROW, ROWS_COUNT };

__global__ __launch_bounds__(ORI_T) void k_orientation(Integral S, int rows, int cols, float *kp, long long kstep, int upright, Tables T)
{
    __shared__ OriShared sm;
    const int k = blockIdx.x;
    float *size = kp + SIZE_ROW * kstep + k;
    const float sz = *size;
    if (!(sz > 0)) return;          // already erased
    orientation_block(S, rows, cols, kp[X_ROW * kstep + k], kp[Y_ROW * kstep + k], sz, upright, T, sm, kp + ANGLE_ROW * kstep + k, size);
}

__global__ __launch_bounds__(DESC_T) void k_descriptor(Image I, const float *kp, long long kstep, int upright, int extended, Tables T,
                                                       float *desc, long long dstep)
{
    __shared__ DescShared sm;
    const int k = blockIdx.x;
    float *out = desc + (long long)k * dstep;
    const float sz = kp[SIZE_ROW * kstep + k];
    if (!(sz > 0)) {                // erased keypoint: zero row
        if ((int)threadIdx.x < (extended ? 128 : 64)) out[threadIdx.x] = 0.f;
        return;
    }
    descriptor_block(I, kp[X_ROW * kstep + k], kp[Y_ROW * kstep + k], sz, kp[ANGLE_ROW * kstep + k], upright, extended, T, sm, out);
}

static const Tables &tables()
{
    static const Tables T = [] { Tables t; make_tables(t); return t; }();
    return T;
}

static int check_keypoints(const mi_mat *kp, int n)
{
    MI_REQUIRE(kp && kp->data && kp->type == MI_32FC1 && kp->rows == ROWS_COUNT && kp->cols >= n && n >= 0 && kp->step % 4 == 0, MI_ERR_BAD_ARG,
               "keypoints must be the CV_32FC1 7 x nFeatures matrix of SURF_CUDA");
    return MI_OK;
}

}  // namespace surfcpu
}  // namespace mi

using namespace mi;

extern "C" {

int mi_surfcpu_orientation(const mi_mat *sum, mi_mat *keypoints, int n, int upright, void *stream)
{
    MI_REQUIRE(sum && sum->data && sum->type == MI_32SC1 && sum->rows > 1 && sum->cols > 1 && sum->step % 4 == 0, MI_ERR_BAD_ARG,
               "sum must be the CV_32SC1 (rows + 1) x (cols + 1) integral image");
    if (int rc = surfcpu::check_keypoints(keypoints, n)) return rc;
    if (n == 0) return MI_OK;
    const surfcpu::Integral S = {(const int *)sum->data, (long long)(sum->step / 4)};
    hipLaunchKernelGGL(surfcpu::k_orientation, dim3(n), dim3(surfcpu::ORI_T), 0, (hipStream_t)stream, S, sum->rows - 1, sum->cols - 1,
                       (float *)keypoints->data, (long long)(keypoints->step / 4), upright != 0, surfcpu::tables());
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int mi_surfcpu_descriptors(const mi_mat *img, const mi_mat *keypoints, int n, int extended, int upright, mi_mat *descriptors, void *stream)
{
    MI_REQUIRE(img && img->data && img->type == MI_8UC1 && img->rows > 1 && img->cols > 1, MI_ERR_BAD_TYPE, "img must be CV_8UC1");
    if (int rc = surfcpu::check_keypoints(keypoints, n)) return rc;
    const int dsize = extended ? 128 : 64;
    MI_REQUIRE(descriptors && descriptors->data && descriptors->type == MI_32FC1 && descriptors->rows >= n && descriptors->cols == dsize &&
                   descriptors->step % 4 == 0, MI_ERR_BAD_SIZE, "descriptors must be CV_32FC1 nFeatures x 64 (128 when extended)");
    if (n == 0) return MI_OK;
    const surfcpu::Image I = {(const unsigned char *)img->data, (long long)img->step, img->rows, img->cols};
    hipLaunchKernelGGL(surfcpu::k_descriptor, dim3(n), dim3(surfcpu::DESC_T), 0, (hipStream_t)stream, I, (const float *)keypoints->data,
                       (long long)(keypoints->step / 4), upright != 0, extended != 0, surfcpu::tables(), (float *)descriptors->data,
                       (long long)(descriptors->step / 4));
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // extern "C"
