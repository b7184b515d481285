/*
 * oracle/stereobm_ref.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * CPU restatement of cv::cuda::StereoBM (modules/cudastereo/src/stereobm.cpp:139-191,
 * src/cuda/stereobm.cu:66-711).  Dense row-major uint8 images (step == cols).
 * PARITY UNPINNED (golden PNGs live in opencv_extra, absent); see stereobm_ref.c.
 */
#ifndef ORACLE_STEREOBM_REF_H
#define ORACLE_STEREOBM_REF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_sbm_params {
    int num_disparities, block_size;
    int prefilter_type;      /* -1 none (default), 0 NORMALIZED_RESPONSE, 1 XSOBEL */
    int prefilter_cap, prefilter_size;
    float texture_threshold; /* avergeTexThreshold_, default 3 */
    int uniqueness_ratio;
    int emulate_edge;        /* 1: reproduce the truncated right half-window of the 128-wide CUDA block mapping */
} orc_sbm_params;

void orc_sbm_default_params(orc_sbm_params *p);
void orc_sbm_prefilter_xsobel(const uint8_t *src, uint8_t *dst, int rows, int cols, int cap);
void orc_sbm_prefilter_norm(const uint8_t *src, uint8_t *dst, int rows, int cols, int cap, int winsize);
/* returns 0, -1 bad argument, -3 image too small for the window/disparity range, -5 out of memory */
int orc_sbm_block_match(const uint8_t *left, const uint8_t *right, int rows, int cols, int ndisp, int winsz,
                        int uniqueness_ratio, int emulate_edge, uint8_t *disp, uint32_t *min_ssd /* may be NULL */);
void orc_sbm_textureness(const uint8_t *img, int rows, int cols, int winsz, float avg_threshold, uint8_t *disp);
int orc_sbm_compute(const orc_sbm_params *p, const uint8_t *left, const uint8_t *right, int rows, int cols, uint8_t *disp);

#ifdef __cplusplus
}
#endif
#endif
