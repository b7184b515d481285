/*
 * oracle/surf_ref.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * CPU restatement of cv::cuda::SURF_CUDA (modules/xfeatures2d/src/surf.cuda.cpp:134-255,
 * src/cuda/surf.cu:122-942; texture-free definitions from src/opencl/surf.cl).  PARITY UNPINNED, see surf_ref.c.
 */
#ifndef ORACLE_SURF_REF_H
#define ORACLE_SURF_REF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_surf_params {
    double hessian_threshold;
    int n_octaves, n_octave_layers, extended;
    float keypoints_ratio;
    int upright;
} orc_surf_params;

void orc_surf_default_params(orc_surf_params *p);
int orc_surf_calc_size(int octave, int layer);
void orc_surf_tables(float aptx[113], float apty[113], float aptw[113], float dw[400]);
void orc_surf_integral(const uint8_t *img, int rows, int cols, uint32_t *sum /* (rows+1)x(cols+1) */);
void orc_surf_det_trace(const uint32_t *sum, int rows, int cols, int octave, int nOctaveLayers, float *det, float *trace);
int orc_surf_find_maxima(const float *det, const float *trace, const uint32_t *mask_sum, int rows, int cols, int octave,
                         int nOctaveLayers, float hessianThreshold, int max_candidates, int *cand /* int4 each */);
int orc_surf_interpolate(const float *det, int rows, int cols, int octave, const int cand[4], float f[6]);
float orc_surf_orientation(const uint32_t *sum, int rows, int cols, float fx, float fy, float fsize, const float *aptx,
                           const float *apty, const float *aptw);
void orc_surf_descriptor(const uint8_t *img, int rows, int cols, float fx, float fy, float fsize, float fdir, int extended,
                         const float *dw, float *desc);
/* keypoints: 7 rows (X, Y, LAPLACIAN(int), OCTAVE(int), SIZE, ANGLE, HESSIAN) x kp_pitch; returns nFeatures (< 0: error) */
int orc_surf_detect_describe(const orc_surf_params *P, const uint8_t *img, const uint8_t *mask, int rows, int cols,
                             float *keypoints, int kp_pitch, float *descriptors, int want_desc);

#ifdef __cplusplus
}
#endif
#endif
