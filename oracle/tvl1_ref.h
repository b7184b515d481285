/*
 * oracle/tvl1_ref.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's Dual TV-L1 optical flow.
 *   semantics 0 (CPU_REF):     cv::optflow::DualTVL1OpticalFlow
 *        modules/optflow/src/tvl1flow.cpp:402-533 (calc), :1313-1408 (procOneScale)
 *   semantics 1 (CUDA_COMPAT): cv::cuda::OpticalFlowDual_TVL1
 *        modules/cudaoptflow/src/tvl1flow.cpp:185-382 + src/cuda/tvl1flow.cu:59-348
 * PARITY UNPINNED: the reference's golden data (RubberWhale, tvl1_flow.flo) is in
 * opencv_extra, absent here, and the reference cannot be built (no OpenCV core);
 * the restatement is anchored on the cited source lines only.
 */
#ifndef ORACLE_TVL1_REF_H
#define ORACLE_TVL1_REF_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_tvl1_params {
    double tau, lambda, theta, epsilon, scale_step, gamma;
    int nscales, warps;
    int inner_iterations;  /* CPU class only; cv::cuda equivalence: 1 */
    int outer_iterations;  /* CPU class; cv::cuda 'iterations' */
    int median_filtering;  /* CPU class only; cv::cuda equivalence: 1 (off) */
    int use_initial_flow;
    int semantics;         /* 0 CPU_REF, 1 CUDA_COMPAT */
} orc_tvl1_params;

#define ORC_TVL1_MAX_SCALES 32
#define ORC_TVL1_MAX_WARPS 64
typedef struct orc_tvl1_stats {
    int nscales_used;
    int level_w[ORC_TVL1_MAX_SCALES], level_h[ORC_TVL1_MAX_SCALES];
    int iters[ORC_TVL1_MAX_SCALES][ORC_TVL1_MAX_WARPS]; /* executed inner iterations */
} orc_tvl1_stats;

void orc_tvl1_default_params(orc_tvl1_params *p); /* optflow/src/tvl1flow.cpp:386-400 */

/* type: 0 = CV_8UC1, 1 = CV_32FC1 (values in [0,1], scaled x255 like the reference).
 * src_step in bytes.  flow: dense interleaved (u,v) float, w*h*2; read when
 * use_initial_flow.  Returns 0, or <0 on bad arguments (the reference CV_Asserts). */
int orc_tvl1_calc(const orc_tvl1_params *p, const void *I0, const void *I1, int type,
                  int w, int h, long src_step, float *flow, orc_tvl1_stats *stats);

/* Stage-level entry points (dense float planes, w*h) used by plane-by-plane tests. */
void orc_tvl1_centered_gradient(const float *src, int w, int h, float *dx, float *dy);
void orc_tvl1_warp(int semantics, const float *I0, const float *I1, const float *I1x,
                   const float *I1y, const float *u1, const float *u2, int w, int h,
                   float *I1w, float *I1wx, float *I1wy, float *grad, float *rho_c);
/* one inner iteration, in place on u*, p*; returns the error sum the reference would
 * compute (CPU_REF: serial float sum, :1085-1115). u3/p31/p32 may be NULL when gamma==0 */
float orc_tvl1_iteration(int semantics, const float *I1wx, const float *I1wy, const float *grad,
                         const float *rho_c, float *u1, float *u2, float *u3, float *p11,
                         float *p12, float *p21, float *p22, float *p31, float *p32, int w, int h,
                         float l_t, float theta, float taut, float gamma);

/* one pyramid level: gradient, `warps` x (warp + iteration loop with the class's convergence rule), in place on
 * u1, u2 (u3 when gamma != 0).  iters_out[warp] = executed inner iterations (may be NULL). */
void orc_tvl1_proc_one_scale(const orc_tvl1_params *p, const float *I0, const float *I1, float *u1, float *u2,
                             float *u3, int w, int h, int *iters_out);

#ifdef __cplusplus
}
#endif
#endif
