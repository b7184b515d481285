/*
 * oracle/refshim/cudashim/tvl1_dbf_cu_host.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * C entry points over two more reference CUDA sources run on the CPU through cudashim.h (launch sites rewritten by cu2host.py):
 *   modules/cudaoptflow/src/cuda/tvl1flow.cu  -- the kernels of cv::cuda::OpticalFlowDual_TVL1 themselves, called as
 *       OpticalFlowDual_TVL1_Impl::procOneScale does (cudaoptflow/src/tvl1flow.cpp:58-76, 304-382);
 *   modules/cudastereo/src/cuda/disparity_bilateral_filter.cu -- called as DispBilateralFilterImpl does, tables included
 *       (cudastereo/src/disparity_bilateral_filter.cpp:88-113, 136-160).
 * Dense planes, step = cols * elemSize.
 */
#include "opencv2/core/cuda/common.hpp"
#include <cmath>
#include <vector>

using namespace cv::cuda;

namespace tvl1flow {
void centeredGradient(PtrStepSzf src, PtrStepSzf dx, PtrStepSzf dy, cudaStream_t stream);
void warpBackward(PtrStepSzf I0, PtrStepSzf I1, PtrStepSzf I1x, PtrStepSzf I1y, PtrStepSzf u1, PtrStepSzf u2, PtrStepSzf I1w, PtrStepSzf I1wx,
                  PtrStepSzf I1wy, PtrStepSzf grad, PtrStepSzf rho, cudaStream_t stream);
void estimateU(PtrStepSzf I1wx, PtrStepSzf I1wy, PtrStepSzf grad, PtrStepSzf rho_c, PtrStepSzf p11, PtrStepSzf p12, PtrStepSzf p21, PtrStepSzf p22,
               PtrStepSzf p31, PtrStepSzf p32, PtrStepSzf u1, PtrStepSzf u2, PtrStepSzf u3, PtrStepSzf error, float l_t, float theta, float gamma,
               bool calcError, cudaStream_t stream);
void estimateDualVariables(PtrStepSzf u1, PtrStepSzf u2, PtrStepSzf u3, PtrStepSzf p11, PtrStepSzf p12, PtrStepSzf p21, PtrStepSzf p22, PtrStepSzf p31,
                           PtrStepSzf p32, float taut, float gamma, cudaStream_t stream);
}
namespace cv { namespace cuda { namespace device { namespace disp_bilateral_filter {
template <typename T>
void disp_bilateral_filter(PtrStepSz<T> disp, PtrStepSzb img, int channels, int iters, const float *, const float *, size_t, int radius, short edge_disc,
                           short max_disc, cudaStream_t stream);
}}}}

static PtrStepSzf P(const float *p, int rows, int cols) { return PtrStepSzf(rows, cols, (float *)p, (size_t)cols * 4); }

extern "C" {

void ref_cu_tvl1_centered_gradient(const float *src, int rows, int cols, float *dx, float *dy)
{
    tvl1flow::centeredGradient(P(src, rows, cols), P(dx, rows, cols), P(dy, rows, cols), nullptr);
}
void ref_cu_tvl1_warp(const float *I0, const float *I1, const float *I1x, const float *I1y, const float *u1, const float *u2, int rows, int cols,
                      float *I1w, float *I1wx, float *I1wy, float *grad, float *rho)
{
    tvl1flow::warpBackward(P(I0, rows, cols), P(I1, rows, cols), P(I1x, rows, cols), P(I1y, rows, cols), P(u1, rows, cols), P(u2, rows, cols),
                           P(I1w, rows, cols), P(I1wx, rows, cols), P(I1wy, rows, cols), P(grad, rows, cols), P(rho, rows, cols), nullptr);
}
/* u*, p* in place; u3 / p31 / p32 may be NULL when gamma == 0 (the class passes empty matrices then) */
void ref_cu_tvl1_estimate_u(const float *I1wx, const float *I1wy, const float *grad, const float *rho_c, const float *p11, const float *p12,
                            const float *p21, const float *p22, const float *p31, const float *p32, float *u1, float *u2, float *u3, float *error,
                            int rows, int cols, float l_t, float theta, float gamma, int calc_error)
{
    tvl1flow::estimateU(P(I1wx, rows, cols), P(I1wy, rows, cols), P(grad, rows, cols), P(rho_c, rows, cols), P(p11, rows, cols), P(p12, rows, cols),
                        P(p21, rows, cols), P(p22, rows, cols), P(p31, rows, cols), P(p32, rows, cols), P(u1, rows, cols), P(u2, rows, cols),
                        P(u3, rows, cols), P(error, rows, cols), l_t, theta, gamma, calc_error != 0, nullptr);
}
void ref_cu_tvl1_estimate_dual(const float *u1, const float *u2, const float *u3, float *p11, float *p12, float *p21, float *p22, float *p31, float *p32,
                               int rows, int cols, float taut, float gamma)
{
    tvl1flow::estimateDualVariables(P(u1, rows, cols), P(u2, rows, cols), P(u3, rows, cols), P(p11, rows, cols), P(p12, rows, cols), P(p21, rows, cols),
                                    P(p22, rows, cols), P(p31, rows, cols), P(p32, rows, cols), taut, gamma, nullptr);
}

/* DispBilateralFilterImpl::apply on a copy of disp.  disp_type 0 = CV_8U, 3 = CV_16S; channels 1 or 3 (interleaved u8 image). */
int ref_cu_dbf_apply(void *disp_inout, int disp_type, const unsigned char *img, int channels, int rows, int cols, int ndisp, int radius, int iters,
                     float edge_threshold, float max_disc_threshold, float sigma_range)
{
    using namespace cv::cuda::device::disp_bilateral_filter;
    // calc_color_weighted_table / calc_space_weighted_filter, disparity_bilateral_filter.cpp:88-113
    std::vector<float> table_color(255);
    for (int i = 0; i < 255; i++) table_color[i] = static_cast<float>(std::exp(-double(i * i) / (2 * sigma_range * sigma_range)));
    const int win_size = radius * 2 + 1, half = win_size >> 1;
    const float dist_space = radius + 1.0f;
    std::vector<float> table_space((size_t)(half + 1) * (half + 1));
    for (int y = 0; y <= half; ++y)
        for (int x = 0; x <= half; ++x) table_space[(size_t)y * (half + 1) + x] = exp(-sqrt(float(y * y) + float(x * x)) / dist_space);
    const short edge_disc = std::max<short>(short(1), short(ndisp * edge_threshold + 0.5));   // :143-144
    const short max_disc = short(ndisp * max_disc_threshold + 0.5);
    PtrStepSzb im(rows, cols, (unsigned char *)img, (size_t)cols * channels);
    if (disp_type == 0)
        disp_bilateral_filter<unsigned char>(PtrStepSz<unsigned char>(rows, cols, (unsigned char *)disp_inout, (size_t)cols), im, channels, iters,
                                             table_color.data(), table_space.data(), (size_t)(half + 1), radius, edge_disc, max_disc, nullptr);
    else if (disp_type == 3)
        disp_bilateral_filter<short>(PtrStepSz<short>(rows, cols, (short *)disp_inout, (size_t)cols * 2), im, channels, iters, table_color.data(),
                                     table_space.data(), (size_t)(half + 1), radius, edge_disc, max_disc, nullptr);
    else
        return -1;
    return 0;
}

}  // extern "C"
