/*
 * oracle/farneback_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of cv::cuda::FarnebackOpticalFlow:
 *   modules/cudaoptflow/src/farneback.cpp:167-207  calc (split/merge of the initial flow)
 *   modules/cudaoptflow/src/farneback.cpp:209-276  prepareGaussian / setPolynomialExpansionConsts
 *   modules/cudaoptflow/src/farneback.cpp:314-482  calcImpl (level loop)
 *   modules/cudaoptflow/src/cuda/farneback.cu:66-119   polynomialExpansion
 *   modules/cudaoptflow/src/cuda/farneback.cu:156-241  updateMatrices
 *   modules/cudaoptflow/src/cuda/farneback.cu:267-286  updateFlow
 *   modules/cudaoptflow/src/cuda/farneback.cu:357-412  boxFilter5
 *   modules/cudaoptflow/src/cuda/farneback.cu:455-492  gaussianBlur,  :539-595 gaussianBlur5
 *   modules/cudawarping/src/cuda/pyr_down.cu:54-175    pyrDown (fastPyramids)
 *   modules/cudawarping/src/cuda/resize.cu:234-269     cuda::resize INTER_LINEAR (oracle/imgproc_ref.c)
 * The CPU twin cv::calcOpticalFlowFarneback lives in the un-vendored main repo
 * (opencv/opencv modules/video/src/optflowgf.cpp; call site modules/optflow/src/interfaces.cpp:154-157);
 * the CUDA code above is the only in-tree statement of the arithmetic and was written to mirror it
 * (the reference's own test accepts |1-CCORR| <= 1e-4 between the two, cudaoptflow/test/test_optflow.cpp:349).
 * Main-repo helpers restated from their documented behaviour: cv::getGaussianKernel (CV_32F),
 * Mat::inv(DECOMP_CHOLESKY) on a 6x6 double matrix, cvRound (round half to even), BrdReflect101/BrdReplicate.
 * Every float operation is separately rounded in the reference's order (-ffp-contract=off).
 * PARITY UNPINNED: RubberWhale (opencv_extra) is absent; anchored on the cited lines + analytic-flow tests.
 */
#include "farneback_ref.h"
#include "imgproc_ref.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BORDER_SIZE 5 /* farneback.cu:55 */
#define MIN_SIZE 32   /* farneback.cpp:54 */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* BrdReflect101 (main repo core/cuda/border_interpolate.hpp): idx_low = |i| % n, idx_high = |last - |last - i|| % n */
static inline int reflect101(int i, int n)
{
    const int last = n - 1;
    int v = abs(last - abs(last - i)) % n;
    return abs(v) % n;
}
static inline int border_idx(int i, int n, int mode) { return mode == ORC_BORDER_REFLECT101 ? reflect101(i, n) : clampi(i, 0, n - 1); }

/* cv::getGaussianKernel(n, sigma, CV_32F): fixed tables for n <= 7 when sigma <= 0, otherwise
 * exp(-x^2 / (2 sigma^2)) normalised to sum 1 (computed in double, stored as float). */
void orc_fb_gaussian_kernel(int n, double sigma, float *k)
{
    static const float tab1[] = {1.f};
    static const float tab3[] = {0.25f, 0.5f, 0.25f};
    static const float tab5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    static const float tab7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    if (sigma <= 0 && n <= 7 && (n & 1)) {
        const float *t = n == 1 ? tab1 : n == 3 ? tab3 : n == 5 ? tab5 : tab7;
        memcpy(k, t, sizeof(float) * (size_t)n);
        return;
    }
    const double sx = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2 = -0.5 / (sx * sx);
    double sum = 0;
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) * 0.5;
        w[i] = exp(scale2 * x * x);
        sum += w[i];
    }
    for (int i = 0; i < n; ++i) k[i] = (float)(w[i] / sum);
    free(w);
}

/* 6x6 symmetric positive definite inverse (Cholesky), double */
static int spd_inverse6(const double A[6][6], double inv[6][6])
{
    double L[6][6] = {{0}};
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i][j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            if (i == j) { if (s <= 0) return -1; L[i][i] = sqrt(s); }
            else L[i][j] = s / L[j][j];
        }
    for (int c = 0; c < 6; ++c) {
        double y[6], x[6];
        for (int i = 0; i < 6; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
            y[i] = s / L[i][i];
        }
        for (int i = 5; i >= 0; --i) {
            double s = y[i];
            for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k];
            x[i] = s / L[i][i];
        }
        for (int i = 0; i < 6; ++i) inv[i][c] = x[i];
    }
    return 0;
}

/* farneback.cpp:209-276.  g/xg/xxg: polyN+1 entries (index 0..n, the non-negative half the kernels use). */
int orc_fb_prepare_gaussian(int n, double sigma, float *g, float *xg, float *xxg, float ig[4])
{
    if (sigma < FLT_EPSILON) sigma = n * 0.3;     /* :270-271 */
    float gb[2 * 8 + 1], xgb[2 * 8 + 1], xxgb[2 * 8 + 1];
    float *gg = gb + n, *xgg = xgb + n, *xxgg = xxgb + n;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        gg[x] = (float)exp(-x * x / (2 * sigma * sigma));
        s += gg[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        gg[x] = (float)(gg[x] * s);
        xgg[x] = (float)(x * gg[x]);
        xxgg[x] = (float)(x * x * gg[x]);
    }
    double G[6][6] = {{0}};
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0][0] += gg[y] * gg[x];
            G[1][1] += gg[y] * gg[x] * x * x;
            G[3][3] += gg[y] * gg[x] * x * x * x * x;
            G[5][5] += gg[y] * gg[x] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    double inv[6][6];
    if (spd_inverse6(G, inv)) return -1;
    ig[0] = (float)inv[1][1]; ig[1] = (float)inv[0][3]; ig[2] = (float)inv[3][3]; ig[3] = (float)inv[5][5];
    for (int k = 0; k <= n; ++k) { g[k] = gg[k]; xg[k] = xgg[k]; xxg[k] = xxgg[k]; }
    return 0;
}

/* farneback.cu:455-492: separable blur, vertical pass first, symmetric pairs, kernel half gker[0..ksizeHalf] */
void orc_fb_gaussian_blur(const float *src, float *dst, int w, int h, int ksizeHalf, const float *gker, int border)
{
#pragma omp parallel
    {
        float *row = (float *)malloc(sizeof(float) * (size_t)(w + 2 * ksizeHalf));
#pragma omp for schedule(static)
        for (int y = 0; y < h; ++y) {
            for (int i = 0; i < w + 2 * ksizeHalf; ++i) {
                const int xe = border_idx(i - ksizeHalf, w, border);
                float v = src[(size_t)y * w + xe] * gker[0];
                for (int j = 1; j <= ksizeHalf; ++j)
                    v += (src[(size_t)border_idx(y - j, h, border) * w + xe] + src[(size_t)border_idx(y + j, h, border) * w + xe]) * gker[j];
                row[i] = v;
            }
            for (int x = 0; x < w; ++x) {
                const float *r = row + x + ksizeHalf;
                float res = r[0] * gker[0];
                for (int i = 1; i <= ksizeHalf; ++i) res += (r[-i] + r[i]) * gker[i];
                dst[(size_t)y * w + x] = res;
            }
        }
        free(row);
    }
}

/* farneback.cu:66-119.  dst: 5 stacked planes (5h x w). */
void orc_fb_poly_exp(const float *src, int w, int h, int polyN, const float *g, const float *xg, const float *xxg,
                     const float ig[4], float *dst)
{
    const float ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];
#pragma omp parallel
    {
        float *r0 = (float *)malloc(sizeof(float) * 3 * (size_t)(w + 2 * polyN));
        float *r1 = r0 + (w + 2 * polyN), *r2 = r1 + (w + 2 * polyN);
#pragma omp for schedule(static)
        for (int y = 0; y < h; ++y) {
            for (int i = 0; i < w + 2 * polyN; ++i) {
                const int xw = clampi(i - polyN, 0, w - 1);
                float a = src[(size_t)y * w + xw] * g[0], b = 0.f, c = 0.f;
                for (int k = 1; k <= polyN; ++k) {
                    const float t0 = src[(size_t)(y - k > 0 ? y - k : 0) * w + xw];
                    const float t1 = src[(size_t)(y + k < h - 1 ? y + k : h - 1) * w + xw];
                    a += g[k] * (t0 + t1);
                    b += xg[k] * (t1 - t0);
                    c += xxg[k] * (t0 + t1);
                }
                r0[i] = a; r1[i] = b; r2[i] = c;
            }
            for (int x = 0; x < w; ++x) {
                const float *p0 = r0 + x + polyN, *p1 = r1 + x + polyN, *p2 = r2 + x + polyN;
                float b1 = g[0] * p0[0], b3 = g[0] * p1[0], b5 = g[0] * p2[0];
                float b2 = 0, b4 = 0, b6 = 0;
                for (int k = 1; k <= polyN; ++k) {
                    b1 += (p0[k] + p0[-k]) * g[k];
                    b4 += (p0[k] + p0[-k]) * xxg[k];
                    b2 += (p0[k] - p0[-k]) * xg[k];
                    b3 += (p1[k] + p1[-k]) * g[k];
                    b6 += (p1[k] - p1[-k]) * xg[k];
                    b5 += (p2[k] + p2[-k]) * g[k];
                }
                dst[(size_t)y * w + x] = b3 * ig11;
                dst[(size_t)(h + y) * w + x] = b2 * ig11;
                dst[(size_t)(2 * h + y) * w + x] = b1 * ig03 + b5 * ig33;
                dst[(size_t)(3 * h + y) * w + x] = b1 * ig03 + b4 * ig33;
                dst[(size_t)(4 * h + y) * w + x] = b6 * ig55;
            }
        }
        free(r0);
    }
}

/* farneback.cu:156-241 */
void orc_fb_update_matrices(const float *flowx, const float *flowy, const float *R0, const float *R1, float *M, int w, int h)
{
    static const float border[BORDER_SIZE + 1] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f, 1.f};  /* :246 */
#define R1P(k, yy, xx) R1[(size_t)((k) * h + (yy)) * w + (xx)]
#define R0P(k) R0[(size_t)((k) * h + y) * w + x]
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float dx = flowx[(size_t)y * w + x], dy = flowy[(size_t)y * w + x];
            float fx = x + dx, fy = y + dy;
            const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
            fx -= x1; fy -= y1;
            float r2, r3, r4, r5, r6;
            if (x1 >= 0 && y1 >= 0 && x1 < w - 1 && y1 < h - 1) {
                const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * R1P(0, y1, x1) + a01 * R1P(0, y1, x1 + 1) + a10 * R1P(0, y1 + 1, x1) + a11 * R1P(0, y1 + 1, x1 + 1);
                r3 = a00 * R1P(1, y1, x1) + a01 * R1P(1, y1, x1 + 1) + a10 * R1P(1, y1 + 1, x1) + a11 * R1P(1, y1 + 1, x1 + 1);
                r4 = a00 * R1P(2, y1, x1) + a01 * R1P(2, y1, x1 + 1) + a10 * R1P(2, y1 + 1, x1) + a11 * R1P(2, y1 + 1, x1 + 1);
                r5 = a00 * R1P(3, y1, x1) + a01 * R1P(3, y1, x1 + 1) + a10 * R1P(3, y1 + 1, x1) + a11 * R1P(3, y1 + 1, x1 + 1);
                r6 = a00 * R1P(4, y1, x1) + a01 * R1P(4, y1, x1 + 1) + a10 * R1P(4, y1 + 1, x1) + a11 * R1P(4, y1 + 1, x1 + 1);
                r4 = (R0P(2) + r4) * 0.5f;
                r5 = (R0P(3) + r5) * 0.5f;
                r6 = (R0P(4) + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0P(2);
                r5 = R0P(3);
                r6 = R0P(4) * 0.5f;
            }
            r2 = (R0P(0) - r2) * 0.5f;
            r3 = (R0P(1) - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            const int bx0 = x < BORDER_SIZE ? x : BORDER_SIZE, by0 = y < BORDER_SIZE ? y : BORDER_SIZE;
            const int bx1 = w - x - 1 < BORDER_SIZE ? w - x - 1 : BORDER_SIZE, by1 = h - y - 1 < BORDER_SIZE ? h - y - 1 : BORDER_SIZE;
            const float scale = border[bx0] * border[by0] * border[bx1] * border[by1];
            r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            M[(size_t)y * w + x] = r4 * r4 + r6 * r6;
            M[(size_t)(h + y) * w + x] = (r4 + r5) * r6;
            M[(size_t)(2 * h + y) * w + x] = r5 * r5 + r6 * r6;
            M[(size_t)(3 * h + y) * w + x] = r4 * r2 + r6 * r3;
            M[(size_t)(4 * h + y) * w + x] = r6 * r2 + r5 * r3;
        }
#undef R1P
#undef R0P
}

/* farneback.cu:267-286 */
void orc_fb_update_flow(const float *M, float *flowx, float *flowy, int w, int h)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float g11 = M[(size_t)y * w + x], g12 = M[(size_t)(h + y) * w + x], g22 = M[(size_t)(2 * h + y) * w + x];
            const float h1 = M[(size_t)(3 * h + y) * w + x], h2 = M[(size_t)(4 * h + y) * w + x];
            const float detInv = 1.f / (g11 * g22 - g12 * g12 + 1e-3f);
            flowx[(size_t)y * w + x] = (g11 * h2 - g12 * h1) * detInv;
            flowy[(size_t)y * w + x] = (g22 * h1 - g12 * h2) * detInv;
        }
}

/* farneback.cu:357-412 (gker == NULL: box, result * boxAreaInv) and :539-595 (gker: Gaussian, BORDER_REPLICATE) */
void orc_fb_blur5(const float *src, float *dst, int w, int h, int ksizeHalf, const float *gker)
{
    const float boxAreaInv = 1.f / ((1 + 2 * ksizeHalf) * (1 + 2 * ksizeHalf));
    const int smw = w + 2 * ksizeHalf;
#pragma omp parallel
    {
        float *row = (float *)malloc(sizeof(float) * 5 * (size_t)smw);
#pragma omp for schedule(static)
        for (int y = 0; y < h; ++y) {
            for (int i = 0; i < smw; ++i) {
                const int xe = clampi(i - ksizeHalf, 0, w - 1);
                for (int k = 0; k < 5; ++k) {
                    const float *P = src + (size_t)k * h * w;
                    float v = gker ? P[(size_t)y * w + xe] * gker[0] : P[(size_t)y * w + xe];
                    for (int j = 1; j <= ksizeHalf; ++j) {
                        const float s = P[(size_t)(y - j > 0 ? y - j : 0) * w + xe] + P[(size_t)(y + j < h - 1 ? y + j : h - 1) * w + xe];
                        v += gker ? s * gker[j] : s;
                    }
                    row[k * smw + i] = v;
                }
            }
            for (int x = 0; x < w; ++x)
                for (int k = 0; k < 5; ++k) {
                    const float *r = row + k * smw + x + ksizeHalf;
                    float res = gker ? r[0] * gker[0] : r[0];
                    for (int i = 1; i <= ksizeHalf; ++i) res += gker ? (r[-i] + r[i]) * gker[i] : r[-i] + r[i];
                    dst[(size_t)(k * h + y) * w + x] = gker ? res : res * boxAreaInv;
                }
        }
        free(row);
    }
}

/* cuda::pyrDown on CV_32FC1 (cudawarping/src/pyramids.cpp:66-94, src/cuda/pyr_down.cu:54-175), BrdReflect101 */
void orc_fb_pyr_down(const float *src, int sw, int sh, float *dst, int dw, int dh)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        const int sy = 2 * y;
        for (int dx = 0; dx < dw; ++dx) {
            float v[5];
            for (int j = 0; j < 5; ++j) {
                const int x = reflect101(2 * dx + j - 2, sw);
                float sum;
                sum = 0.0625f * src[(size_t)reflect101(sy - 2, sh) * sw + x];
                sum = sum + 0.25f * src[(size_t)reflect101(sy - 1, sh) * sw + x];
                sum = sum + 0.375f * src[(size_t)sy * sw + x];
                sum = sum + 0.25f * src[(size_t)reflect101(sy + 1, sh) * sw + x];
                sum = sum + 0.0625f * src[(size_t)reflect101(sy + 2, sh) * sw + x];
                v[j] = sum;
            }
            float sum;
            sum = 0.0625f * v[0];
            sum = sum + 0.25f * v[1];
            sum = sum + 0.375f * v[2];
            sum = sum + 0.25f * v[3];
            sum = sum + 0.0625f * v[4];
            dst[(size_t)y * dw + dx] = sum;
        }
    }
}

void orc_fb_default_params(orc_fb_params *p)
{
    /* cudaoptflow.hpp:285-293 */
    p->num_levels = 5; p->pyr_scale = 0.5; p->fast_pyramids = 0; p->win_size = 13; p->num_iters = 10;
    p->poly_n = 5; p->poly_sigma = 1.1; p->flags = 0;
}

static int cv_round(double v) { return (int)lrint(v); }

/* farneback.cpp:167-207 + :314-482.  type 0 = CV_8UC1, 1 = CV_32FC1 (convertTo(CV_32F), no scaling).
 * flow: interleaved (h x w x 2); read as the initial flow when OPTFLOW_USE_INITIAL_FLOW. */
int orc_fb_calc(const orc_fb_params *P, const void *frame0, const void *frame1, int type, int w, int h, float *flow)
{
    if (!(P->poly_n == 5 || P->poly_n == 7)) return -1;                          /* :316 */
    if (P->fast_pyramids && !(fabs(P->pyr_scale - 0.5) < 1e-6)) return -1;      /* :317 */
    if (P->num_levels < 0 || P->win_size < 1 || !(P->win_size & 1) || P->num_iters < 0 || !(P->pyr_scale > 0 && P->pyr_scale < 1)) return -1;
    const size_t n0 = (size_t)w * h;
    float *fr[2] = {(float *)malloc(sizeof(float) * n0), (float *)malloc(sizeof(float) * n0)};
    const void *in[2] = {frame0, frame1};
    for (int i = 0; i < 2; ++i)
        for (size_t k = 0; k < n0; ++k) fr[i][k] = type == 0 ? (float)((const unsigned char *)in[i])[k] : ((const float *)in[i])[k];
    float *flowx0 = (float *)malloc(sizeof(float) * n0), *flowy0 = (float *)malloc(sizeof(float) * n0);
    const int use_init = (P->flags & ORC_OPTFLOW_USE_INITIAL_FLOW) != 0;
    if (use_init) for (size_t k = 0; k < n0; ++k) { flowx0[k] = flow[2 * k]; flowy0[k] = flow[2 * k + 1]; }

    /* crop unnecessary levels :330-340 */
    double scale = 1;
    int levels = 0;
    for (; levels < P->num_levels; levels++) {
        scale *= P->pyr_scale;
        if (w * scale < MIN_SIZE || h * scale < MIN_SIZE) break;
    }
    /* fast pyramids :346-359 */
    float **pyr0 = NULL, **pyr1 = NULL;
    int *pw = NULL, *ph = NULL;
    if (P->fast_pyramids) {
        pyr0 = (float **)calloc((size_t)levels + 1, sizeof(float *));
        pyr1 = (float **)calloc((size_t)levels + 1, sizeof(float *));
        pw = (int *)malloc(sizeof(int) * ((size_t)levels + 1)); ph = (int *)malloc(sizeof(int) * ((size_t)levels + 1));
        pyr0[0] = fr[0]; pyr1[0] = fr[1]; pw[0] = w; ph[0] = h;
        for (int i = 1; i <= levels; ++i) {
            pw[i] = (pw[i - 1] + 1) / 2; ph[i] = (ph[i - 1] + 1) / 2;
            pyr0[i] = (float *)malloc(sizeof(float) * (size_t)pw[i] * ph[i]);
            pyr1[i] = (float *)malloc(sizeof(float) * (size_t)pw[i] * ph[i]);
            orc_fb_pyr_down(pyr0[i - 1], pw[i - 1], ph[i - 1], pyr0[i], pw[i], ph[i]);
            orc_fb_pyr_down(pyr1[i - 1], pw[i - 1], ph[i - 1], pyr1[i], pw[i], ph[i]);
        }
    }
    float g[8], xg[8], xxg[8], ig[4];
    if (orc_fb_prepare_gaussian(P->poly_n, P->poly_sigma, g, xg, xxg, ig)) return -2;

    float *prevx = NULL, *prevy = NULL;
    int prevw = 0, prevh = 0;
    float *blurred = (float *)malloc(sizeof(float) * n0);
    for (int k = levels; k >= 0; k--) {
        scale = 1;
        for (int i = 0; i < k; i++) scale *= P->pyr_scale;
        const double sigma = (1. / scale - 1) * 0.5;
        int smoothSize = cv_round(sigma * 5) | 1;
        smoothSize = smoothSize > 3 ? smoothSize : 3;
        int width = cv_round(w * scale), height = cv_round(h * scale);
        if (P->fast_pyramids) { width = pw[k]; height = ph[k]; }
        const size_t n = (size_t)width * height;
        float *curx, *cury;
        if (k > 0) { curx = (float *)malloc(sizeof(float) * n); cury = (float *)malloc(sizeof(float) * n); }
        else { curx = flowx0; cury = flowy0; }
        if (!prevx) {
            if (use_init) {   /* :398-404 (cuda::resize INTER_LINEAR, then * scale) */
                if (k > 0) {
                    orc_resize_linear_cuda(flowx0, w, h, curx, width, height, (float)(1.0 / ((double)width / w)), (float)(1.0 / ((double)height / h)));
                    orc_resize_linear_cuda(flowy0, w, h, cury, width, height, (float)(1.0 / ((double)width / w)), (float)(1.0 / ((double)height / h)));
                }
                /* GpuMat::convertTo(dst, depth, alpha): float -> float runs in binary32, alpha = saturate_cast<float>(alpha)
                 * (main repo core/src/cuda/gpu_mat.cu ConvertToScale<float,float,float>) */
                const float a = (float)scale;
                for (size_t i = 0; i < n; ++i) { curx[i] = a * curx[i]; cury[i] = a * cury[i]; }
            } else {
                memset(curx, 0, sizeof(float) * n);
                memset(cury, 0, sizeof(float) * n);
            }
        } else {              /* :412-417 */
            orc_resize_linear_cuda(prevx, prevw, prevh, curx, width, height, (float)(1.0 / ((double)width / prevw)), (float)(1.0 / ((double)height / prevh)));
            orc_resize_linear_cuda(prevy, prevw, prevh, cury, width, height, (float)(1.0 / ((double)width / prevw)), (float)(1.0 / ((double)height / prevh)));
            const float a = (float)(1. / P->pyr_scale);
            for (size_t i = 0; i < n; ++i) { curx[i] = a * curx[i]; cury[i] = a * cury[i]; }
        }
        float *M = (float *)malloc(sizeof(float) * 5 * n), *bufM = (float *)malloc(sizeof(float) * 5 * n);
        float *R[2] = {(float *)malloc(sizeof(float) * 5 * n), (float *)malloc(sizeof(float) * 5 * n)};
        if (P->fast_pyramids) {
            orc_fb_poly_exp(pyr0[k], width, height, P->poly_n, g, xg, xxg, ig, R[0]);
            orc_fb_poly_exp(pyr1[k], width, height, P->poly_n, g, xg, xxg, ig, R[1]);
        } else {              /* :434-454 */
            float *gk = (float *)malloc(sizeof(float) * (size_t)smoothSize);
            orc_fb_gaussian_kernel(smoothSize, sigma, gk);
            float *lvl = (float *)malloc(sizeof(float) * n);
            for (int i = 0; i < 2; i++) {
                orc_fb_gaussian_blur(fr[i], blurred, w, h, smoothSize / 2, gk + smoothSize / 2, ORC_BORDER_REFLECT101);
                if (width == w && height == h) memcpy(lvl, blurred, sizeof(float) * n);   /* resize to the same size = copy (resize.cpp:89-93) */
                else orc_resize_linear_cuda(blurred, w, h, lvl, width, height, (float)(1.0 / ((double)width / w)), (float)(1.0 / ((double)height / h)));
                orc_fb_poly_exp(lvl, width, height, P->poly_n, g, xg, xxg, ig, R[i]);
            }
            free(lvl);
            free(gk);
        }
        orc_fb_update_matrices(curx, cury, R[0], R[1], M, width, height);
        float *wk = NULL;
        if (P->flags & ORC_OPTFLOW_FARNEBACK_GAUSSIAN) {   /* :460-464 */
            wk = (float *)malloc(sizeof(float) * (size_t)P->win_size);
            orc_fb_gaussian_kernel(P->win_size, (double)(P->win_size / 2 * 0.3f), wk);
        }
        for (int i = 0; i < P->num_iters; i++) {           /* :465-471 -> :278-312 */
            orc_fb_blur5(M, bufM, width, height, P->win_size / 2, wk ? wk + P->win_size / 2 : NULL);
            float *t = M; M = bufM; bufM = t;
            orc_fb_update_flow(M, curx, cury, width, height);
            if (i < P->num_iters - 1) orc_fb_update_matrices(curx, cury, R[0], R[1], M, width, height);
        }
        free(wk); free(M); free(bufM); free(R[0]); free(R[1]);
        if (prevx && prevx != flowx0) { free(prevx); free(prevy); }
        prevx = curx; prevy = cury; prevw = width; prevh = height;
    }
    for (size_t k = 0; k < n0; ++k) { flow[2 * k] = flowx0[k]; flow[2 * k + 1] = flowy0[k]; }   /* cuda::merge :198 */
    free(blurred); free(flowx0); free(flowy0);
    if (P->fast_pyramids) {
        for (int i = 1; i <= levels; ++i) { free(pyr0[i]); free(pyr1[i]); }
        free(pyr0); free(pyr1); free(pw); free(ph);
    }
    free(fr[0]); free(fr[1]);
    return 0;
}
