/*
 * oracle/refshim/cudashim/farneback_cu_host.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * C entry points over the reference's own Farneback CUDA source (modules/cudaoptflow/src/cuda/farneback.cu, rewritten only at
 * its launch sites by cu2host.py, run on the CPU through cudashim.h), called as FarnebackOpticalFlowImpl does
 * (modules/cudaoptflow/src/farneback.cpp:60-92, 209-312).  Dense f32 planes, step = cols * 4; 5-plane stacks are 5*rows x cols.
 */
#include "opencv2/core/cuda/common.hpp"

namespace cv { namespace cuda { namespace device { namespace optflow_farneback {
void setPolynomialExpansionConsts(int polyN, const float *g, const float *xg, const float *xxg, float ig11, float ig03, float ig33, float ig55);
void polynomialExpansionGpu(const PtrStepSzf &src, int polyN, PtrStepSzf dst, cudaStream_t stream);
void setUpdateMatricesConsts();
void updateMatricesGpu(const PtrStepSzf flowx, const PtrStepSzf flowy, const PtrStepSzf R0, const PtrStepSzf R1, PtrStepSzf M, cudaStream_t stream);
void updateFlowGpu(const PtrStepSzf M, PtrStepSzf flowx, PtrStepSzf flowy, cudaStream_t stream);
void boxFilter5Gpu(const PtrStepSzf src, int ksizeHalf, PtrStepSzf dst, cudaStream_t stream);
void setGaussianBlurKernel(const float *gKer, int ksizeHalf);
void gaussianBlurGpu(const PtrStepSzf src, int ksizeHalf, PtrStepSzf dst, int borderType, cudaStream_t stream);
void gaussianBlur5Gpu(const PtrStepSzf src, int ksizeHalf, PtrStepSzf dst, int borderType, cudaStream_t stream);
}}}}

using namespace cv::cuda;
namespace fb = cv::cuda::device::optflow_farneback;
static PtrStepSzf P(const float *p, int rows, int cols) { return PtrStepSzf(rows, cols, (float *)p, (size_t)cols * 4); }

extern "C" {

/* g, xg, xxg: polyN + 1 taps each (farneback.cpp:209-260 computes them on the host; the caller passes the oracle's) */
void ref_cu_fb_poly_exp(const float *src, int rows, int cols, int polyN, const float *g, const float *xg, const float *xxg, float ig11,
                        float ig03, float ig33, float ig55, float *dst5)
{
    fb::setPolynomialExpansionConsts(polyN, g, xg, xxg, ig11, ig03, ig33, ig55);
    fb::polynomialExpansionGpu(P(src, rows, cols), polyN, P(dst5, 5 * rows, cols), nullptr);
}
void ref_cu_fb_update_matrices(const float *flowx, const float *flowy, const float *R0, const float *R1, int rows, int cols, float *M5)
{
    fb::setUpdateMatricesConsts();
    fb::updateMatricesGpu(P(flowx, rows, cols), P(flowy, rows, cols), P(R0, 5 * rows, cols), P(R1, 5 * rows, cols), P(M5, 5 * rows, cols), nullptr);
}
void ref_cu_fb_update_flow(const float *M5, int rows, int cols, float *flowx, float *flowy)
{
    fb::updateFlowGpu(P(M5, 5 * rows, cols), P(flowx, rows, cols), P(flowy, rows, cols), nullptr);
}
void ref_cu_fb_box5(const float *M5, int rows, int cols, int ksizeHalf, float *dst5)
{
    fb::boxFilter5Gpu(P(M5, 5 * rows, cols), ksizeHalf, P(dst5, 5 * rows, cols), nullptr);
}
/* ker: ksizeHalf + 1 taps (centre first); borderType: cv::BorderTypes (1 replicate, 4 reflect101) */
void ref_cu_fb_gaussian_blur(const float *src, int rows, int cols, const float *ker, int ksizeHalf, int borderType, float *dst)
{
    fb::setGaussianBlurKernel(ker, ksizeHalf);
    fb::gaussianBlurGpu(P(src, rows, cols), ksizeHalf, P(dst, rows, cols), borderType, nullptr);
}
void ref_cu_fb_gaussian_blur5(const float *M5, int rows, int cols, const float *ker, int ksizeHalf, int borderType, float *dst5)
{
    fb::setGaussianBlurKernel(ker, ksizeHalf);
    fb::gaussianBlur5Gpu(P(M5, 5 * rows, cols), ksizeHalf, P(dst5, 5 * rows, cols), borderType, nullptr);
}

}  // extern "C"
