/*
 * oracle/imgproc_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See imgproc_ref.h.
 * Build with -ffp-contract=off: every multiply/add below is a separately
 * rounded IEEE binary32 operation, in the order the restated algorithm uses.
 */
#include "imgproc_ref.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_scaled_dim(int n, double f) { return (int)lrint((double)n * f); }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------ */
/* cv::resize INTER_LINEAR, CV_32FC1: horizontal pass (alpha) then vertical
 * pass (beta), both in float.  fx/fy are computed in double, cast to float,
 * floor taken, fractional part formed in float. */
void orc_resize_linear_cv(const float *src, int sw, int sh, float *dst, int dw, int dh,
                          double scale_x, double scale_y)
{
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    float *alpha = (float *)malloc(sizeof(float) * 2 * (size_t)dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = 1.f - fx;
        alpha[2 * dx + 1] = fx;
    }
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        const float b0 = 1.f - fy, b1 = fy;
        const float *S0 = src + (size_t)clampi(sy, 0, sh - 1) * sw;
        const float *S1 = src + (size_t)clampi(sy + 1, 0, sh - 1) * sw;
        float *D = dst + (size_t)dy * dw;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx];
            const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            const float a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
            const float h0 = S0[sx] * a0 + S0[sx1] * a1;
            const float h1 = S1[sx] * a0 + S1[sx1] * a1;
            D[dx] = h0 * b0 + h1 * b1;
        }
    }
    free(xofs);
    free(alpha);
}

/* ------------------------------------------------------------------------ */
void orc_resize_linear_cuda(const float *src, int sw, int sh, float *dst, int dw, int dh,
                            float fx, float fy)
{
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < dh; ++dy) {
        for (int dx = 0; dx < dw; ++dx) {
            const float src_x = (float)dx * fx;
            const float src_y = (float)dy * fy;
            const int x1 = (int)floorf(src_x), y1 = (int)floorf(src_y);
            const int x2 = x1 + 1, y2 = y1 + 1;
            const int x2r = x2 < sw - 1 ? x2 : sw - 1;
            const int y2r = y2 < sh - 1 ? y2 : sh - 1;
            float out = 0.f;
            out = out + src[(size_t)y1 * sw + x1] * (((float)x2 - src_x) * ((float)y2 - src_y));
            out = out + src[(size_t)y1 * sw + x2r] * ((src_x - (float)x1) * ((float)y2 - src_y));
            out = out + src[(size_t)y2r * sw + x1] * (((float)x2 - src_x) * (src_y - (float)y1));
            out = out + src[(size_t)y2r * sw + x2r] * ((src_x - (float)x1) * (src_y - (float)y1));
            dst[(size_t)dy * dw + dx] = out;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* cv::remap bicubic.  Table: 32 sub-pixel phases, Keys a = -0.75, float. */
#define ORC_TAB 32
static float g_cubic_tab[ORC_TAB][4];
static int g_cubic_tab_ready = 0;

static void cubic_coeffs(float x, float *c)
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

static void init_cubic_tab(void)
{
    if (g_cubic_tab_ready) return;
    const float scale = 1.f / ORC_TAB;
    for (int i = 0; i < ORC_TAB; ++i) cubic_coeffs((float)i * scale, g_cubic_tab[i]);
    g_cubic_tab_ready = 1;
}

const float *orc_cubic_table(void) { init_cubic_tab(); return &g_cubic_tab[0][0]; }

static inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

void orc_remap_cubic_cv(const float *src, int sw, int sh, const float *mapx, const float *mapy,
                        float *dst, int dw, int dh)
{
    init_cubic_tab();
    const unsigned width1 = (unsigned)(sw - 3 > 0 ? sw - 3 : 0);
    const unsigned height1 = (unsigned)(sh - 3 > 0 ? sh - 3 : 0);
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < dh; ++dy) {
        for (int dx = 0; dx < dw; ++dx) {
            const size_t di = (size_t)dy * dw + dx;
            /* float map * 32 is exact; cvRound = round-half-even */
            const int qx = (int)lrintf(mapx[di] * (float)ORC_TAB);
            const int qy = (int)lrintf(mapy[di] * (float)ORC_TAB);
            const int sx = sat_short(qx >> 5) - 1;
            const int sy = sat_short(qy >> 5) - 1;
            const float *wx = g_cubic_tab[qx & (ORC_TAB - 1)];
            const float *wy = g_cubic_tab[qy & (ORC_TAB - 1)];
            float w[16];
            for (int k1 = 0; k1 < 4; ++k1)
                for (int k2 = 0; k2 < 4; ++k2) w[k1 * 4 + k2] = wy[k1] * wx[k2];

            if ((unsigned)sx < width1 && (unsigned)sy < height1) {
                const float *S = src + (size_t)sy * sw + sx;
                float sum = S[0] * w[0] + S[1] * w[1] + S[2] * w[2] + S[3] * w[3];
                S += sw;
                sum += S[0] * w[4] + S[1] * w[5] + S[2] * w[6] + S[3] * w[7];
                S += sw;
                sum += S[0] * w[8] + S[1] * w[9] + S[2] * w[10] + S[3] * w[11];
                S += sw;
                sum += S[0] * w[12] + S[1] * w[13] + S[2] * w[14] + S[3] * w[15];
                dst[di] = sum;
            } else {
                if (sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0) {
                    dst[di] = 0.f;
                    continue;
                }
                /* BORDER_CONSTANT, cval = 0: out-of-image taps contribute
                 * (0 - 0)*w, i.e. are skipped; sum starts at cval. */
                float sum = 0.f;
                for (int i = 0; i < 4; ++i) {
                    const int yi = sy + i;
                    if (yi < 0 || yi >= sh) continue;
                    const float *S = src + (size_t)yi * sw;
                    for (int j = 0; j < 4; ++j) {
                        const int xj = sx + j;
                        if (xj >= 0 && xj < sw) sum += (S[xj] - 0.f) * w[i * 4 + j];
                    }
                }
                dst[di] = sum;
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
static int cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

void orc_median_blur(const float *src, float *dst, int w, int h, int ksize)
{
    const int r = ksize / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        float win[25];
        for (int x = 0; x < w; ++x) {
            int n = 0;
            for (int j = -r; j <= r; ++j) {
                const int yy = clampi(y + j, 0, h - 1);
                for (int i = -r; i <= r; ++i) win[n++] = src[(size_t)yy * w + clampi(x + i, 0, w - 1)];
            }
            qsort(win, (size_t)n, sizeof(float), cmp_float);
            dst[(size_t)y * w + x] = win[n / 2];
        }
    }
}
