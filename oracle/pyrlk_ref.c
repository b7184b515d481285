/* CPU restatement of cv::cuda::DensePyrLKOpticalFlow -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Follows modules/cudaoptflow/src/pyrlk.cpp:238-299 (PyrLKOpticalFlowBase::dense: float pyramids by cuda::pyrDown, two
 * full-size zero-initialised (u, v) buffer pairs used alternately from the coarsest level down) and
 * modules/cudaoptflow/src/cuda/pyrlk.cu:709-847 (denseKernel: integer patch of I and its Scharr derivatives, 2x2 structure
 * tensor over the window, up to `iters` Newton steps sampling J bilinearly, early exits that leave (u, v) UNWRITTEN).
 *
 * parity unpinned: the reference samples through texture hardware (cudaFilterModeLinear, 1.8 fixed-point weights) and its
 * tests need opencv_extra images; the two texture reads are DEFINED here as
 *   texI(y + .5, x + .5)  = I[clamp(y)][clamp(x)]                                   (texel centre, clamp addressing)
 *   the window of J around nextPt = the integer lattice shifted by ONE common sub-pixel offset: bx = nextPt.x - halfWin.x,
 *   x0 = floor(bx), fx = bx - x0 (same in y); sample (i, j) = (T00 * (1 - fx) + T01 * fx) * (1 - fy) + (T10 * (1 - fx) + T11 * fx) * fy
 *   over the texels (y0 + i .. +1, x0 + j .. +1), clamp addressing, binary32, every operation separately rounded
 * (the reference adds i, j to the float coordinate first, which perturbs the fraction in the last bits), and the Scharr sums are
 * evaluated left to right without contraction.  Integer accumulations wrap modulo 2^32 like the device's.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void orc_fb_pyr_down(const float *src, int sw, int sh, float *dst, int dw, int dh);   /* oracle/farneback_ref.c: cuda::pyrDown */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float texel(const float *im, int rows, int cols, int y, int x) { return im[(size_t)clampi(y, 0, rows - 1) * cols + clampi(x, 0, cols - 1)]; }

static float tex_linear(const float *im, int rows, int cols, int y0, int x0, float fy, float fx)
{
    const float t00 = texel(im, rows, cols, y0, x0), t01 = texel(im, rows, cols, y0, x0 + 1);
    const float t10 = texel(im, rows, cols, y0 + 1, x0), t11 = texel(im, rows, cols, y0 + 1, x0 + 1);
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const float top = t00 * gx + t01 * fx, bot = t10 * gx + t11 * fx;
    return top * gy + bot * fy;
}

/* one pyramid level; u, v, prevU, prevV are rows0 x ld buffers (the reference's full-size GpuMats), level size rows x cols */
static void dense_level(const float *I, const float *J, int rows, int cols, float *u, float *v, const float *prevU, const float *prevV, int ld,
                        int wx, int wy, int iters)
{
    const int hx = (wx - 1) / 2, hy = (wy - 1) / 2;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int32_t A11i = 0, A12i = 0, A22i = 0;
            /* patch entries of the window (pyrlk.cu:727-741): int truncation of the float expressions */
            int *Ip = (int *)malloc(sizeof(int) * 3 * wx * wy), *dx = Ip + wx * wy, *dy = dx + wx * wy;
            for (int i = 0; i < wy; ++i)
                for (int j = 0; j < wx; ++j) {
                    const int yy = y - hy + i, xx = x - hx + j;
#define TI(a, b) texel(I, rows, cols, (a), (b))
                    Ip[i * wx + j] = (int)TI(yy, xx);
                    dx[i * wx + j] = (int)(3 * TI(yy - 1, xx + 1) + 10 * TI(yy, xx + 1) + 3 * TI(yy + 1, xx + 1) -
                                           (3 * TI(yy - 1, xx - 1) + 10 * TI(yy, xx - 1) + 3 * TI(yy + 1, xx - 1)));
                    dy[i * wx + j] = (int)(3 * TI(yy + 1, xx - 1) + 10 * TI(yy + 1, xx) + 3 * TI(yy + 1, xx + 1) -
                                           (3 * TI(yy - 1, xx - 1) + 10 * TI(yy - 1, xx) + 3 * TI(yy - 1, xx + 1)));
#undef TI
                    A11i = (int32_t)((uint32_t)A11i + (uint32_t)(dx[i * wx + j] * dx[i * wx + j]));
                    A12i = (int32_t)((uint32_t)A12i + (uint32_t)(dx[i * wx + j] * dy[i * wx + j]));
                    A22i = (int32_t)((uint32_t)A22i + (uint32_t)(dy[i * wx + j] * dy[i * wx + j]));
                }
            float A11 = (float)A11i, A12 = (float)A12i, A22 = (float)A22i;
            float D = A11 * A22 - A12 * A12;
            if (D < FLT_EPSILON) { free(Ip); continue; }   /* pyrlk.cu:777-782: returns without writing u, v */
            D = 1.f / D;
            A11 *= D; A12 *= D; A22 *= D;
            float nx = x + prevU[(size_t)(y / 2) * ld + x / 2] * 2.0f, ny = y + prevV[(size_t)(y / 2) * ld + x / 2] * 2.0f;
            int alive = 1;
            for (int k = 0; k < iters; ++k) {
                if (nx < 0 || nx >= cols || ny < 0 || ny >= rows) { alive = 0; break; }   /* :796-802: return, nothing written */
                int32_t b1 = 0, b2 = 0;
                const float bx = nx - hx, by = ny - hy;
                const float x0f = floorf(bx), y0f = floorf(by);
                const float fx = bx - x0f, fy = by - y0f;
                const int x0 = (int)x0f, y0 = (int)y0f;
                for (int i = 0; i < wy; ++i)
                    for (int j = 0; j < wx; ++j) {
                        const int Jv = (int)tex_linear(J, rows, cols, y0 + i, x0 + j, fy, fx);
                        const int diff = (Jv - Ip[i * wx + j]) * 32;
                        b1 = (int32_t)((uint32_t)b1 + (uint32_t)(diff * dx[i * wx + j]));
                        b2 = (int32_t)((uint32_t)b2 + (uint32_t)(diff * dy[i * wx + j]));
                    }
                const float ddx = A12 * b2 - A22 * b1, ddy = A12 * b1 - A11 * b2;
                nx += ddx; ny += ddy;
                if (fabsf(ddx) < 0.01f && fabsf(ddy) < 0.01f) break;
            }
            free(Ip);
            if (!alive) continue;
            u[(size_t)y * ld + x] = nx - x;
            v[(size_t)y * ld + x] = ny - y;
        }
}

/* prev, next: rows x cols uint8; flow: rows x cols x 2 float */
int orc_pyrlk_dense(const unsigned char *prev, const unsigned char *next, int rows, int cols, int wx, int wy, int max_level, int iters, float *flow)
{
    if (max_level < 0 || !(wx > 2 && wy > 2)) return -1;   /* CV_Assert, pyrlk.cpp:242-243 */
    const int nl = max_level + 1;
    float **P = (float **)malloc(sizeof(float *) * nl), **N = (float **)malloc(sizeof(float *) * nl);
    int *pw = (int *)malloc(sizeof(int) * nl), *ph = (int *)malloc(sizeof(int) * nl);
    pw[0] = cols; ph[0] = rows;
    P[0] = (float *)malloc(sizeof(float) * rows * cols); N[0] = (float *)malloc(sizeof(float) * rows * cols);
    for (size_t i = 0; i < (size_t)rows * cols; ++i) { P[0][i] = prev[i]; N[0][i] = next[i]; }   /* convertTo(CV_32F) */
    for (int l = 1; l < nl; ++l) {
        pw[l] = (pw[l - 1] + 1) / 2; ph[l] = (ph[l - 1] + 1) / 2;
        P[l] = (float *)malloc(sizeof(float) * pw[l] * ph[l]); N[l] = (float *)malloc(sizeof(float) * pw[l] * ph[l]);
        orc_fb_pyr_down(P[l - 1], pw[l - 1], ph[l - 1], P[l], pw[l], ph[l]);
        orc_fb_pyr_down(N[l - 1], pw[l - 1], ph[l - 1], N[l], pw[l], ph[l]);
    }
    const size_t n = (size_t)rows * cols;
    float *U[2], *V[2];
    for (int b = 0; b < 2; ++b) { U[b] = (float *)calloc(n, sizeof(float)); V[b] = (float *)calloc(n, sizeof(float)); }
    int idx = 0;
    for (int l = max_level; l >= 0; --l) {   /* pyrlk.cpp:284-295 */
        const int idx2 = (idx + 1) & 1;
        dense_level(P[l], N[l], ph[l], pw[l], U[idx], V[idx], U[idx2], V[idx2], cols, wx, wy, iters);
        if (l > 0) idx = idx2;
    }
    for (size_t i = 0; i < n; ++i) { flow[2 * i] = U[idx][i]; flow[2 * i + 1] = V[idx][i]; }
    for (int l = 0; l < nl; ++l) { free(P[l]); free(N[l]); }
    for (int b = 0; b < 2; ++b) { free(U[b]); free(V[b]); }
    free(P); free(N); free(pw); free(ph);
    return 0;
}
