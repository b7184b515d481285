/*
 * oracle/tvl1_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See tvl1_ref.h.
 * Build with -ffp-contract=off (each float op separately rounded, reference order).
 */
#include "tvl1_ref.h"
#include "imgproc_ref.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_tvl1_default_params(orc_tvl1_params *p)
{
    /* optflow/src/tvl1flow.cpp:386-400 */
    p->tau = 0.25; p->lambda = 0.15; p->theta = 0.3; p->nscales = 5; p->warps = 5;
    p->epsilon = 0.01; p->gamma = 0.; p->inner_iterations = 30; p->outer_iterations = 10;
    p->use_initial_flow = 0; p->median_filtering = 5; p->scale_step = 0.8; p->semantics = 0;
}

static float *falloc(int w, int h) { return (float *)calloc((size_t)w * h, sizeof(float)); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* optflow/src/tvl1flow.cpp:688-770 (one-sided differences x0.5 at the borders ==
 * clamp form of cudaoptflow/src/cuda/tvl1flow.cu:59-69) */
void orc_tvl1_centered_gradient(const float *src, int w, int h, float *dx, float *dy)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float *r = src + (size_t)y * w;
        const float *rp = src + (size_t)imax(y - 1, 0) * w;
        const float *rn = src + (size_t)imin(y + 1, h - 1) * w;
        for (int x = 0; x < w; ++x) {
            dx[(size_t)y * w + x] = 0.5f * (r[imin(x + 1, w - 1)] - r[imax(x - 1, 0)]);
            dy[(size_t)y * w + x] = 0.5f * (rn[x] - rp[x]);
        }
    }
}

/* cudaoptflow/src/cuda/tvl1flow.cu:89-104 */
static inline float bicubic_coeff_cuda(float x_)
{
    const float x = fabsf(x_);
    if (x <= 1.0f) return x * x * (1.5f * x - 2.5f) + 1.0f;
    else if (x < 2.0f) return x * (x * (-0.5f * x + 2.5f) - 4.0f) + 2.0f;
    return 0.0f;
}

/* cudaoptflow/src/cuda/tvl1flow.cu:106-149; texture = point sampled, clamp addressed
 * (cudev/ptr2d/texture.hpp:228-232) */
static void warp_cuda_planes(const float *I1, const float *I1x, const float *I1y, const float *u1,
                             const float *u2, int w, int h, float *I1w, float *I1wx, float *I1wy)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const float wx = (float)x + u1[i], wy = (float)y + u2[i];
            const int xmin = (int)ceilf(wx - 2.0f), xmax = (int)floorf(wx + 2.0f);
            const int ymin = (int)ceilf(wy - 2.0f), ymax = (int)floorf(wy + 2.0f);
            float sum = 0.f, sumx = 0.f, sumy = 0.f, wsum = 0.f;
            for (int cy = ymin; cy <= ymax; ++cy)
                for (int cx = xmin; cx <= xmax; ++cx) {
                    const float wgt = bicubic_coeff_cuda(wx - (float)cx) * bicubic_coeff_cuda(wy - (float)cy);
                    const size_t j = (size_t)imin(imax(cy, 0), h - 1) * w + imin(imax(cx, 0), w - 1);
                    sum += wgt * I1[j];
                    sumx += wgt * I1x[j];
                    sumy += wgt * I1y[j];
                    wsum += wgt;
                }
            const float coeff = 1.0f / wsum;
            I1w[i] = sum * coeff;
            I1wx[i] = sumx * coeff;
            I1wy[i] = sumy * coeff;
        }
}

void orc_tvl1_warp(int semantics, const float *I0, const float *I1, const float *I1x,
                   const float *I1y, const float *u1, const float *u2, int w, int h, float *I1w,
                   float *I1wx, float *I1wy, float *grad, float *rho_c)
{
    if (semantics == 0) {
        /* optflow/src/tvl1flow.cpp:668-683 buildFlowMap, :1371-1374 3x remap(INTER_CUBIC) */
        float *m1 = falloc(w, h), *m2 = falloc(w, h);
#pragma omp parallel for schedule(static)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                m1[(size_t)y * w + x] = (float)x + u1[(size_t)y * w + x];
                m2[(size_t)y * w + x] = (float)y + u2[(size_t)y * w + x];
            }
        orc_remap_cubic_cv(I1, w, h, m1, m2, I1w, w, h);
        orc_remap_cubic_cv(I1x, w, h, m1, m2, I1wx, w, h);
        orc_remap_cubic_cv(I1y, w, h, m1, m2, I1wy, w, h);
        free(m1);
        free(m2);
    } else {
        warp_cuda_planes(I1, I1x, I1y, u1, u2, w, h, I1w, I1wx, I1wy);
    }
    /* optflow/src/tvl1flow.cpp:918-944 calcGradRho == tvl1flow.cu:151-163 */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const float Ix2 = I1wx[i] * I1wx[i];
            const float Iy2 = I1wy[i] * I1wy[i];
            grad[i] = Ix2 + Iy2;
            rho_c[i] = (I1w[i] - I1wx[i] * u1[i] - I1wy[i] * u2[i] - I0[i]);
        }
}

/* divergence with the first-row/col special cases: optflow/src/tvl1flow.cpp:857-899
 * == tvl1flow.cu:187-207 */
static inline float div_at(const float *v1, const float *v2, int w, int x, int y)
{
    const size_t i = (size_t)y * w + x;
    if (x > 0 && y > 0) {
        const float v1x = v1[i] - v1[i - 1];
        const float v2y = v2[i] - v2[i - w];
        return v1x + v2y;
    }
    if (y > 0) return v1[i] + v2[i] - v2[i - w];       /* x == 0 */
    if (x > 0) return v1[i] - v1[i - 1] + v2[i];       /* y == 0 */
    return v1[i] + v2[i];
}

float orc_tvl1_iteration(int semantics, const float *I1wx, const float *I1wy, const float *grad,
                         const float *rho_c, float *u1, float *u2, float *u3, float *p11,
                         float *p12, float *p21, float *p22, float *p31, float *p32, int w, int h,
                         float l_t, float theta, float taut, float gamma)
{
    const int use_gamma = gamma != 0.f;
    const size_t n = (size_t)w * h;
    float *errs = (float *)malloc(n * sizeof(float));

    /* estimateV (:989-1041) + divergence (:857-899) + estimateU (:1074-1116), per pixel.
     * Race-free in place: u(x,y) depends on u(x,y) and p at (x,y),(x-1,y),(x,y-1) only. */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const float ix = I1wx[i], iy = I1wy[i], g = grad[i];
            const float u1k = u1[i], u2k = u2[i], u3k = use_gamma ? u3[i] : 0.f;
            float rho;
            if (semantics == 0)
                rho = use_gamma ? rho_c[i] + (ix * u1k + iy * u2k) + gamma * u3k
                                : rho_c[i] + (ix * u1k + iy * u2k);
            else /* tvl1flow.cu:233 */
                rho = rho_c[i] + (ix * u1k + iy * u2k + gamma * u3k);
            float d1 = 0.f, d2 = 0.f, d3 = 0.f;
            if (rho < -l_t * g) {
                d1 = l_t * ix; d2 = l_t * iy; if (use_gamma) d3 = l_t * gamma;
            } else if (rho > l_t * g) {
                d1 = -l_t * ix; d2 = -l_t * iy; if (use_gamma) d3 = -l_t * gamma;
            } else if (g > FLT_EPSILON) {
                const float fi = -rho / g;
                d1 = fi * ix; d2 = fi * iy; if (use_gamma) d3 = fi * gamma;
            }
            const float v1 = u1k + d1, v2 = u2k + d2, v3 = u3k + d3;
            const float dp1 = div_at(p11, p12, w, x, y);
            const float dp2 = div_at(p21, p22, w, x, y);
            const float u1n = v1 + theta * dp1;
            const float u2n = v2 + theta * dp2;
            u1[i] = u1n;
            u2[i] = u2n;
            float e = (u1n - u1k) * (u1n - u1k) + (u2n - u2k) * (u2n - u2k);
            if (use_gamma) {
                const float dp3 = div_at(p31, p32, w, x, y);
                const float u3n = v3 + theta * dp3;
                u3[i] = u3n;
                if (semantics == 0) e = e + (u3n - u3k) * (u3n - u3k); /* :1110 */
            }
            errs[i] = e;
        }
    float error;
    if (semantics == 0) { /* serial float accumulation, row-major (:1085,1110) */
        float acc = 0.f;
        for (size_t i = 0; i < n; ++i) acc += errs[i];
        error = acc;
    } else { /* cuda::calcSum: float terms accumulated in double (cudaoptflow tvl1flow.cpp:366-370) */
        double acc = 0.0;
        for (size_t i = 0; i < n; ++i) acc += (double)errs[i];
        error = (float)acc;
    }
    free(errs);

    /* forwardGradient (:775-840) + estimateDualVariables (:1140-1181) == tvl1flow.cu:313-348.
     * In place: p(x,y) depends on p(x,y) and the *new* u at (x,y),(x+1,y),(x,y+1). */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const size_t ir = (size_t)y * w + imin(x + 1, w - 1);
            const size_t id = (size_t)imin(y + 1, h - 1) * w + x;
            const float u1x = u1[ir] - u1[i], u1y = u1[id] - u1[i];
            const float u2x = u2[ir] - u2[i], u2y = u2[id] - u2[i];
            float g1, g2;
            if (semantics == 0) { /* hypot in double, cast (:1161-1162) */
                g1 = (float)hypot((double)u1x, (double)u1y);
                g2 = (float)hypot((double)u2x, (double)u2y);
            } else {
                g1 = hypotf(u1x, u1y);
                g2 = hypotf(u2x, u2y);
            }
            const float ng1 = 1.0f + taut * g1, ng2 = 1.0f + taut * g2;
            p11[i] = (p11[i] + taut * u1x) / ng1;
            p12[i] = (p12[i] + taut * u1y) / ng1;
            p21[i] = (p21[i] + taut * u2x) / ng2;
            p22[i] = (p22[i] + taut * u2y) / ng2;
            if (use_gamma) {
                const float u3x = u3[ir] - u3[i], u3y = u3[id] - u3[i];
                const float g3 = semantics == 0 ? (float)hypot((double)u3x, (double)u3y) : hypotf(u3x, u3y);
                const float ng3 = 1.0f + taut * g3;
                p31[i] = (p31[i] + taut * u3x) / ng3;
                p32[i] = (p32[i] + taut * u3y) / ng3;
            }
        }
    return error;
}

typedef struct {
    int w, h;
    float *I0, *I1, *u1, *u2, *u3;
} level_t;

static void resize_level(int semantics, const float *src, int sw, int sh, float *dst, int dw, int dh,
                         double fx_given, int explicit_dsize)
{
    if (semantics == 0) {
        /* cv::resize: inv_scale = fx (dsize empty) or dsize/ssize; scale = 1/inv_scale */
        const double isx = explicit_dsize ? (double)dw / sw : fx_given;
        const double isy = explicit_dsize ? (double)dh / sh : fx_given;
        orc_resize_linear_cv(src, sw, sh, dst, dw, dh, 1. / isx, 1. / isy);
    } else {
        /* cudawarping/src/resize.cpp:76-107 */
        const double fx = explicit_dsize ? (double)dw / sw : fx_given;
        const double fy = explicit_dsize ? (double)dh / sh : fx_given;
        if (dw == sw && dh == sh) { memcpy(dst, src, sizeof(float) * (size_t)sw * sh); return; }
        orc_resize_linear_cuda(src, sw, sh, dst, dw, dh, (float)(1.0 / fx), (float)(1.0 / fy));
    }
}

static void scale_plane(float *p, size_t n, float s)
{
    for (size_t i = 0; i < n; ++i) p[i] = p[i] * s;
}

static void proc_one_scale(const orc_tvl1_params *P, const level_t *L, int *iters_out)
{
    const int w = L->w, h = L->h;
    const size_t n = (size_t)w * h;
    const int sem = P->semantics;
    const int use_gamma = P->gamma != 0.;
    float *I1x = falloc(w, h), *I1y = falloc(w, h);
    float *I1w = falloc(w, h), *I1wx = falloc(w, h), *I1wy = falloc(w, h);
    float *grad = falloc(w, h), *rho_c = falloc(w, h);
    float *p11 = falloc(w, h), *p12 = falloc(w, h), *p21 = falloc(w, h), *p22 = falloc(w, h);
    float *p31 = use_gamma ? falloc(w, h) : NULL, *p32 = use_gamma ? falloc(w, h) : NULL;
    float *tmp = falloc(w, h);

    orc_tvl1_centered_gradient(L->I1, w, h, I1x, I1y);

    const float l_t = (float)(P->lambda * P->theta);
    const float taut = (float)(P->tau / P->theta);
    const float theta = (float)P->theta;
    const float gamma = (float)P->gamma;

    for (int wp = 0; wp < P->warps; ++wp) {
        orc_tvl1_warp(sem, L->I0, L->I1, I1x, I1y, L->u1, L->u2, w, h, I1w, I1wx, I1wy, grad, rho_c);
        int executed = 0;
        if (sem == 0) {
            /* optflow/src/tvl1flow.cpp:1315, 1376-1406 */
            const float scaledEps = (float)(P->epsilon * P->epsilon * (double)(w * h));
            float error = FLT_MAX;
            for (int no = 0; error > scaledEps && no < P->outer_iterations; ++no) {
                if (P->median_filtering > 1) {
                    orc_median_blur(L->u1, tmp, w, h, P->median_filtering);
                    memcpy(L->u1, tmp, n * sizeof(float));
                    orc_median_blur(L->u2, tmp, w, h, P->median_filtering);
                    memcpy(L->u2, tmp, n * sizeof(float));
                }
                for (int ni = 0; error > scaledEps && ni < P->inner_iterations; ++ni) {
                    error = orc_tvl1_iteration(sem, I1wx, I1wy, grad, rho_c, L->u1, L->u2, L->u3, p11,
                                               p12, p21, p22, p31, p32, w, h, l_t, theta, taut, gamma);
                    ++executed;
                }
            }
        } else {
            /* cudaoptflow/src/tvl1flow.cpp:310, 357-380 (iterations == outer*inner) */
            const double scaledEps = P->epsilon * P->epsilon * (double)(w * h);
            const int iterations = P->outer_iterations * P->inner_iterations;
            double error = DBL_MAX, prevError = 0.0;
            for (int nn = 0; error > scaledEps && nn < iterations; ++nn) {
                const int calcError = (P->epsilon > 0) && (nn & 1) && (prevError < scaledEps);
                const float e = orc_tvl1_iteration(sem, I1wx, I1wy, grad, rho_c, L->u1, L->u2, L->u3,
                                                   p11, p12, p21, p22, p31, p32, w, h, l_t, theta,
                                                   taut, gamma);
                ++executed;
                if (calcError) { error = (double)e; prevError = error; }
                else { error = DBL_MAX; prevError -= scaledEps; }
            }
        }
        if (iters_out && wp < ORC_TVL1_MAX_WARPS) iters_out[wp] = executed;
    }
    free(I1x); free(I1y); free(I1w); free(I1wx); free(I1wy); free(grad); free(rho_c);
    free(p11); free(p12); free(p21); free(p22); free(p31); free(p32); free(tmp);
}

void orc_tvl1_proc_one_scale(const orc_tvl1_params *P, const float *I0, const float *I1, float *u1, float *u2,
                             float *u3, int w, int h, int *iters_out)
{
    level_t L;
    L.w = w; L.h = h; L.I0 = (float *)I0; L.I1 = (float *)I1; L.u1 = u1; L.u2 = u2; L.u3 = u3;
    proc_one_scale(P, &L, iters_out);
}

int orc_tvl1_calc(const orc_tvl1_params *P, const void *I0, const void *I1, int type, int w, int h,
                  long src_step, float *flow, orc_tvl1_stats *stats)
{
    if (!P || !I0 || !I1 || !flow) return -1;
    if (type != 0 && type != 1) return -2;               /* CV_Assert :417 */
    if (P->nscales <= 0 || P->nscales > ORC_TVL1_MAX_SCALES) return -3; /* :421 */
    if (w < 3 || h < 3) return -4;
    const int use_gamma = P->gamma != 0.;
    const int sem = P->semantics;
    int nscales = P->nscales;
    level_t *L = (level_t *)calloc((size_t)nscales, sizeof(level_t));

    /* convertTo(CV_32F, 8U ? 1 : 255)  (:428-429) */
    L[0].w = w; L[0].h = h;
    L[0].I0 = falloc(w, h); L[0].I1 = falloc(w, h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            if (type == 0) {
                L[0].I0[(size_t)y * w + x] = (float)((const unsigned char *)I0)[(size_t)y * src_step + x];
                L[0].I1[(size_t)y * w + x] = (float)((const unsigned char *)I1)[(size_t)y * src_step + x];
            } else {
                const float *r0 = (const float *)((const char *)I0 + (size_t)y * src_step);
                const float *r1 = (const float *)((const char *)I1 + (size_t)y * src_step);
                L[0].I0[(size_t)y * w + x] = r0[x] * 255.0f;
                L[0].I1[(size_t)y * w + x] = r1[x] * 255.0f;
            }
        }
    L[0].u1 = falloc(w, h); L[0].u2 = falloc(w, h);
    if (use_gamma) L[0].u3 = falloc(w, h);
    if (P->use_initial_flow) { /* :435-439 split */
        for (size_t i = 0; i < (size_t)w * h; ++i) { L[0].u1[i] = flow[2 * i]; L[0].u2[i] = flow[2 * i + 1]; }
    }

    /* create the scales (:473-500) */
    for (int s = 1; s < nscales; ++s) {
        const int pw = L[s - 1].w, ph = L[s - 1].h;
        const int cw = orc_scaled_dim(pw, P->scale_step), ch = orc_scaled_dim(ph, P->scale_step);
        if (cw < 1 || ch < 1) { nscales = s; break; }
        L[s].w = cw; L[s].h = ch;
        L[s].I0 = falloc(cw, ch); L[s].I1 = falloc(cw, ch);
        resize_level(sem, L[s - 1].I0, pw, ph, L[s].I0, cw, ch, P->scale_step, 0);
        resize_level(sem, L[s - 1].I1, pw, ph, L[s].I1, cw, ch, P->scale_step, 0);
        if (cw < 16 || ch < 16) { nscales = s; break; }
        L[s].u1 = falloc(cw, ch); L[s].u2 = falloc(cw, ch);
        if (P->use_initial_flow) {
            resize_level(sem, L[s - 1].u1, pw, ph, L[s].u1, cw, ch, P->scale_step, 0);
            resize_level(sem, L[s - 1].u2, pw, ph, L[s].u2, cw, ch, P->scale_step, 0);
            scale_plane(L[s].u1, (size_t)cw * ch, (float)P->scale_step);
            scale_plane(L[s].u2, (size_t)cw * ch, (float)P->scale_step);
        }
        if (use_gamma) L[s].u3 = falloc(cw, ch);
    }
    if (!P->use_initial_flow) {
        memset(L[nscales - 1].u1, 0, sizeof(float) * (size_t)L[nscales - 1].w * L[nscales - 1].h);
        memset(L[nscales - 1].u2, 0, sizeof(float) * (size_t)L[nscales - 1].w * L[nscales - 1].h);
    }
    if (use_gamma)
        memset(L[nscales - 1].u3, 0, sizeof(float) * (size_t)L[nscales - 1].w * L[nscales - 1].h);

    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->nscales_used = nscales;
        for (int s = 0; s < nscales; ++s) { stats->level_w[s] = L[s].w; stats->level_h[s] = L[s].h; }
    }

    /* coarse to fine (:503-529) */
    for (int s = nscales - 1; s >= 0; --s) {
        proc_one_scale(P, &L[s], stats ? stats->iters[s] : NULL);
        if (s == 0) break;
        const int fw = L[s - 1].w, fh = L[s - 1].h;
        resize_level(sem, L[s].u1, L[s].w, L[s].h, L[s - 1].u1, fw, fh, 0, 1);
        resize_level(sem, L[s].u2, L[s].w, L[s].h, L[s - 1].u2, fw, fh, 0, 1);
        if (use_gamma) resize_level(sem, L[s].u3, L[s].w, L[s].h, L[s - 1].u3, fw, fh, 0, 1);
        /* multiply by 1/scaleStep as float; u3 is not scaled (:526-528) */
        scale_plane(L[s - 1].u1, (size_t)fw * fh, (float)(1 / P->scale_step));
        scale_plane(L[s - 1].u2, (size_t)fw * fh, (float)(1 / P->scale_step));
    }

    /* merge (:531-532) */
    for (size_t i = 0; i < (size_t)w * h; ++i) { flow[2 * i] = L[0].u1[i]; flow[2 * i + 1] = L[0].u2[i]; }

    for (int s = 0; s < P->nscales; ++s) {
        free(L[s].I0); free(L[s].I1); free(L[s].u1); free(L[s].u2); free(L[s].u3);
    }
    free(L);
    return 0;
}
