/*
 * oracle/farneback_ref.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * CPU restatement of cv::cuda::FarnebackOpticalFlow (modules/cudaoptflow/src/farneback.cpp:167-482,
 * src/cuda/farneback.cu:66-651).  Dense row-major planes; 5-plane buffers are stacked vertically (5h x w)
 * like the reference's.  PARITY UNPINNED, see farneback_ref.c.
 */
#ifndef ORACLE_FARNEBACK_REF_H
#define ORACLE_FARNEBACK_REF_H
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_BORDER_REPLICATE = 1, ORC_BORDER_REFLECT101 = 4 };                 /* cv::BorderTypes */
enum { ORC_OPTFLOW_USE_INITIAL_FLOW = 4, ORC_OPTFLOW_FARNEBACK_GAUSSIAN = 256 }; /* cv::OPTFLOW_* (main repo video/tracking.hpp) */

typedef struct orc_fb_params {
    int num_levels;
    double pyr_scale;
    int fast_pyramids;
    int win_size, num_iters, poly_n;
    double poly_sigma;
    int flags;
} orc_fb_params;

void orc_fb_default_params(orc_fb_params *p);
void orc_fb_gaussian_kernel(int n, double sigma, float *k);
int orc_fb_prepare_gaussian(int n, double sigma, float *g, float *xg, float *xxg, float ig[4]);
void orc_fb_gaussian_blur(const float *src, float *dst, int w, int h, int ksizeHalf, const float *gker_center, int border);
void orc_fb_poly_exp(const float *src, int w, int h, int polyN, const float *g, const float *xg, const float *xxg,
                     const float ig[4], float *dst);
void orc_fb_update_matrices(const float *flowx, const float *flowy, const float *R0, const float *R1, float *M, int w, int h);
void orc_fb_update_flow(const float *M, float *flowx, float *flowy, int w, int h);
void orc_fb_blur5(const float *src, float *dst, int w, int h, int ksizeHalf, const float *gker_center /* NULL: box */);
void orc_fb_pyr_down(const float *src, int sw, int sh, float *dst, int dw, int dh);
/* returns 0; -1 bad argument (CV_Assert); -2 internal */
int orc_fb_calc(const orc_fb_params *P, const void *frame0, const void *frame1, int type, int w, int h, float *flow);

#ifdef __cplusplus
}
#endif
#endif
