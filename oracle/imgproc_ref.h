/*
 * oracle/imgproc_ref.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatements of the main-repo (opencv/opencv 4.x, NOT vendored under
 * /root/reference) image primitives that the reference hot path calls.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  PARITY UNPINNED: the reference's golden vectors live in
 * opencv_extra, which is absent; these functions restate the published
 * algorithm of cv::resize / cv::remap / cv::medianBlur from its documented
 * behaviour (call sites cited per function).
 *
 * All planes are dense row-major float (stride == width) unless noted.
 */
#ifndef ORACLE_IMGPROC_REF_H
#define ORACLE_IMGPROC_REF_H

#ifdef __cplusplus
extern "C" {
#endif

/* cv::resize(INTER_LINEAR) on CV_32FC1, CPU convention (half-pixel centres).
 * scale_x = 1/inv_scale_x as cv::resize computes it: when dsize is empty,
 * inv_scale = the fx the caller passed; otherwise inv_scale = dsize/ssize.
 * Call sites: optflow/src/tvl1flow.cpp:479-480, 522-524. */
void orc_resize_linear_cv(const float *src, int sw, int sh, float *dst, int dw, int dh,
                          double scale_x, double scale_y);

/* cv::cuda::resize(INTER_LINEAR) on CV_32FC1: src = dst * f, floor, 4 taps,
 * right/bottom clamp (cudawarping/src/cuda/resize.cu:234-269). fx,fy = 1/scale
 * as float (cudawarping/src/resize.cpp:107). */
void orc_resize_linear_cuda(const float *src, int sw, int sh, float *dst, int dw, int dh,
                            float fx, float fy);

/* dsize rule shared by both: saturate_cast<int>(ssize * f) = round-half-even
 * (cudawarping/src/resize.cpp:78). */
int orc_scaled_dim(int n, double f);

/* cv::remap(INTER_CUBIC, BORDER_CONSTANT 0) with CV_32FC1 maps:
 * coordinates quantised to 1/32 px, Keys a = -0.75 table, float weights.
 * Call site: optflow/src/tvl1flow.cpp:1372-1374. */
void orc_remap_cubic_cv(const float *src, int sw, int sh, const float *mapx, const float *mapy,
                        float *dst, int dw, int dh);

/* cv::medianBlur(ksize=5 or 3) on CV_32FC1, BORDER_REPLICATE.
 * Call site: optflow/src/tvl1flow.cpp:1381-1384. src != dst. */
void orc_median_blur(const float *src, float *dst, int w, int h, int ksize);

#ifdef __cplusplus
}
#endif
#endif
