/*
 * oracle/refshim/tvl1_ocl_host.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Launches the reference's OWN Dual TV-L1 kernels -- modules/optflow/src/opencl/optical_flow_tvl1.cl,
 * compiled verbatim (oracle/Makefile.ref) -- on the CPU through oclrt.  Only the host glue is written
 * here: argument marshalling as cv_ocl_tvl1flow::centeredGradient / warpBackward / estimateU /
 * estimateDualVariables do it (modules/optflow/src/tvl1flow.cpp:245-383; dense planes, so every step
 * is the width and every ROI offset is 0) and the per-scale loop of
 * OpticalFlowDual_TVL1::procOneScale_ocl (tvl1flow.cpp:1224-1310), with cv::sum(diff)[0] as a double
 * accumulation of the float error plane.  The arithmetic executed is the reference's.
 */
#include "oclrt.h"
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* the compiled reference kernels (C calling convention on x86-64) */
extern void centeredGradientKernel(const float *src_ptr, int src_col, int src_row, int src_step, float *dx, float *dy, int d_step);
extern void warpBackwardKernel(const float *I0, int I0_step, int I0_col, int I0_row, const oclrt_image2d *tex_I1,
                               const oclrt_image2d *tex_I1x, const oclrt_image2d *tex_I1y, const float *u1, int u1_step,
                               const float *u2, float *I1w, float *I1wx, float *I1wy, float *grad, float *rho, int I1w_step,
                               int u2_step, int u1_offset_x, int u1_offset_y, int u2_offset_x, int u2_offset_y);
extern void estimateDualVariablesKernel(const float *u1, int u1_col, int u1_row, int u1_step, const float *u2, float *p11,
                                        int p11_step, float *p12, float *p21, float *p22, float taut, int u2_step,
                                        int u1_offset_x, int u1_offset_y, int u2_offset_x, int u2_offset_y);
extern void estimateUKernel(const float *I1wx, int I1wx_col, int I1wx_row, int I1wx_step, const float *I1wy, const float *grad,
                            const float *rho_c, const float *p11, const float *p12, const float *p21, const float *p22, float *u1,
                            int u1_step, float *u2, float *error, float l_t, float theta, int u2_step, int u1_offset_x,
                            int u1_offset_y, int u2_offset_x, int u2_offset_y, char calc_error);

typedef struct { const float *src; int w, h; float *dx, *dy; } grad_args;
static void grad_body(void *a_) { grad_args *a = (grad_args *)a_; centeredGradientKernel(a->src, a->w, a->h, a->w, a->dx, a->dy, a->w); }

/* cv_ocl_tvl1flow::centeredGradient, tvl1flow.cpp:245-262 */
void ref_ocl_tvl1_centered_gradient(const float *src, int w, int h, float *dx, float *dy)
{
    grad_args a = {src, w, h, dx, dy};
    const size_t g[2] = {(size_t)w, (size_t)h};
    oclrt_run(2, g, NULL, 0, 1, grad_body, &a);
}

typedef struct {
    const float *I0; int w, h; oclrt_image2d t1, tx, ty; const float *u1, *u2; float *I1w, *I1wx, *I1wy, *grad, *rho;
} warp_args;
static void warp_body(void *a_)
{
    warp_args *a = (warp_args *)a_;
    warpBackwardKernel(a->I0, a->w, a->w, a->h, &a->t1, &a->tx, &a->ty, a->u1, a->w, a->u2, a->I1w, a->I1wx, a->I1wy, a->grad,
                       a->rho, a->w, a->w, 0, 0, 0, 0);
}

/* cv_ocl_tvl1flow::warpBackward, tvl1flow.cpp:264-306 (ocl::Image2D of the three CV_32FC1 planes) */
void ref_ocl_tvl1_warp(const float *I0, const float *I1, const float *I1x, const float *I1y, const float *u1, const float *u2, int w,
                       int h, float *I1w, float *I1wx, float *I1wy, float *grad, float *rho_c)
{
    warp_args a;
    a.I0 = I0; a.w = w; a.h = h; a.u1 = u1; a.u2 = u2; a.I1w = I1w; a.I1wx = I1wx; a.I1wy = I1wy; a.grad = grad; a.rho = rho_c;
    const oclrt_image2d proto = {NULL, (long)w * 4, w, h, 0};
    a.t1 = proto; a.t1.data = I1; a.tx = proto; a.tx.data = I1x; a.ty = proto; a.ty.data = I1y;
    const size_t g[2] = {(size_t)w, (size_t)h};
    oclrt_run(2, g, NULL, 0, 1, warp_body, &a);
}

typedef struct {
    const float *I1wx, *I1wy, *grad, *rho_c; float *p11, *p12, *p21, *p22, *u1, *u2, *err; int w, h; float l_t, theta, taut;
    char calc_error;
} it_args;
static void estu_body(void *a_)
{
    it_args *a = (it_args *)a_;
    estimateUKernel(a->I1wx, a->w, a->h, a->w, a->I1wy, a->grad, a->rho_c, a->p11, a->p12, a->p21, a->p22, a->u1, a->w, a->u2,
                    a->err, a->l_t, a->theta, a->w, 0, 0, 0, 0, a->calc_error);
}
static void dual_body(void *a_)
{
    it_args *a = (it_args *)a_;
    estimateDualVariablesKernel(a->u1, a->w, a->h, a->w, a->u2, a->p11, a->w, a->p12, a->p21, a->p22, a->taut, a->w, 0, 0, 0, 0);
}

/* cv_ocl_tvl1flow::estimateU, tvl1flow.cpp:308-349.  In place on u1, u2; error plane written when calc_error. */
void ref_ocl_tvl1_estimate_u(const float *I1wx, const float *I1wy, const float *grad, const float *rho_c, const float *p11,
                             const float *p12, const float *p21, const float *p22, float *u1, float *u2, float *error, int w, int h,
                             float l_t, float theta, int calc_error)
{
    it_args a;
    memset(&a, 0, sizeof(a));
    a.I1wx = I1wx; a.I1wy = I1wy; a.grad = grad; a.rho_c = rho_c;
    a.p11 = (float *)p11; a.p12 = (float *)p12; a.p21 = (float *)p21; a.p22 = (float *)p22;
    a.u1 = u1; a.u2 = u2; a.err = error; a.w = w; a.h = h; a.l_t = l_t; a.theta = theta; a.calc_error = (char)(calc_error != 0);
    const size_t g[2] = {(size_t)w, (size_t)h};
    oclrt_run(2, g, NULL, 0, 1, estu_body, &a);
}

/* cv_ocl_tvl1flow::estimateDualVariables, tvl1flow.cpp:351-382.  In place on p. */
void ref_ocl_tvl1_estimate_dual(const float *u1, const float *u2, float *p11, float *p12, float *p21, float *p22, int w, int h, float taut)
{
    it_args a;
    memset(&a, 0, sizeof(a));
    a.u1 = (float *)u1; a.u2 = (float *)u2; a.p11 = p11; a.p12 = p12; a.p21 = p21; a.p22 = p22; a.w = w; a.h = h; a.taut = taut;
    const size_t g[2] = {(size_t)w, (size_t)h};
    oclrt_run(2, g, NULL, 0, 1, dual_body, &a);
}

/* OpticalFlowDual_TVL1::procOneScale_ocl, tvl1flow.cpp:1224-1310 (medianFiltering <= 1: no cv::medianBlur).
 * u1, u2 in place.  iters_out[warp] = executed inner iterations.  Parameters as the class holds them (doubles). */
void ref_ocl_tvl1_proc_one_scale(const float *I0, const float *I1, float *u1, float *u2, int w, int h, double tau, double lambda,
                                 double theta, double epsilon, int warps, int inner_iterations, int outer_iterations, int *iters_out)
{
    const size_t n = (size_t)w * h;
    const double scaledEpsilon = epsilon * epsilon * (double)(w * h);   /* I0.size().area() */
    float *buf = (float *)calloc(n * 12, sizeof(float));
    float *I1x = buf, *I1y = buf + n, *I1w = buf + 2 * n, *I1wx = buf + 3 * n, *I1wy = buf + 4 * n, *grad = buf + 5 * n,
          *rho_c = buf + 6 * n, *p11 = buf + 7 * n, *p12 = buf + 8 * n, *p21 = buf + 9 * n, *p22 = buf + 10 * n, *diff = buf + 11 * n;
    ref_ocl_tvl1_centered_gradient(I1, w, h, I1x, I1y);
    /* p11..p22.setTo(0): calloc */
    const float l_t = (float)(lambda * theta);
    const float taut = (float)(tau / theta);
    for (int warpings = 0; warpings < warps; ++warpings) {
        ref_ocl_tvl1_warp(I0, I1, I1x, I1y, u1, u2, w, h, I1w, I1wx, I1wy, grad, rho_c);
        double error = DBL_MAX;
        double prev_error = 0;
        int executed = 0;
        for (int n_outer = 0; error > scaledEpsilon && n_outer < outer_iterations; ++n_outer) {
            for (int n_inner = 0; error > scaledEpsilon && n_inner < inner_iterations; ++n_inner) {
                const int nn = n_inner + n_outer * inner_iterations;
                const char calc_error = (nn & 0x1) && (prev_error < scaledEpsilon);
                ref_ocl_tvl1_estimate_u(I1wx, I1wy, grad, rho_c, p11, p12, p21, p22, u1, u2, diff, w, h, l_t, (float)theta, calc_error);
                if (calc_error) {
                    double s = 0.0;   /* cv::sum(diff)[0]: double accumulator over the CV_32F plane */
                    for (size_t i = 0; i < n; ++i) s += (double)diff[i];
                    error = s;
                    prev_error = error;
                } else {
                    error = DBL_MAX;
                    prev_error -= scaledEpsilon;
                }
                ref_ocl_tvl1_estimate_dual(u1, u2, p11, p12, p21, p22, w, h, taut);
                ++executed;
            }
        }
        if (iters_out) iters_out[warpings] = executed;
    }
    free(buf);
}
