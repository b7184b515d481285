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

/* ------------------------------------------------------------------------------------------------------------------------------
 * cv::cuda::SparsePyrLKOpticalFlow on CV_8UC1 frames.
 * Follows modules/cudaoptflow/src/pyrlk.cpp:134-231 (PyrLKOpticalFlowBase::buildImagePyramid: cuda::pyrDown of the 8-bit frames, so
 * every level is rounded to 8 bits; ::sparse: nextPts = (useInitialFlow ? nextPts : prevPts) / 2^maxLevel / 2, status = 1, level
 * loop from maxLevel down to 0, err only at level 0) and src/cuda/pyrlk.cu:148-340 (sparseKernel: normalised-float texture reads of
 * the patch and its Scharr derivatives, 2 x 2 structure tensor, Newton steps on J with (J - I) * 32, early exits that clear status
 * ONLY at level 0 and leave nextPts as it was, err = mean |J - I| * 255).
 * The texture reads are DEFINED like the dense path's above: a texel is value / 255 in binary32; a window is the integer lattice
 * shifted by one common sub-pixel offset (x0 = floor(p), f = p - x0), bilinear with separately rounded operations, clamp addressing.
 * The block-wide sums are DEFINED as: lane = element index mod 64 accumulates its elements in ascending order, then the 64 partial
 * sums fold by the tree s = 32, 16, 8, 4, 2, 1 (a[i] += a[i + s]) -- the order of the HIP kernel (one wave per point).
 * parity unpinned (the reference's test compares with cv::calcOpticalFlowPyrLK on opencv_extra images). */
static void pyr_down_u8(const unsigned char *src, int sw, int sh, unsigned char *dst, int dw, int dh)
{
    float *fs = (float *)malloc(sizeof(float) * (size_t)sw * sh), *fd = (float *)malloc(sizeof(float) * (size_t)dw * dh);
    for (size_t i = 0; i < (size_t)sw * sh; ++i) fs[i] = src[i];
    orc_fb_pyr_down(fs, sw, sh, fd, dw, dh);
    for (size_t i = 0; i < (size_t)dw * dh; ++i) {      /* saturate_cast<uchar>(float): round half to even, clamp */
        const long r = lrintf(fd[i]);
        dst[i] = (unsigned char)(r < 0 ? 0 : r > 255 ? 255 : r);
    }
    free(fs); free(fd);
}

/* the 8-bit pyramid step alone (cuda::pyrDown of CV_8UC1), for the pin against cudawarping/src/cuda/pyr_down.cu */
void orc_pyr_down_u8(const unsigned char *src, int sw, int sh, unsigned char *dst, int dw, int dh) { pyr_down_u8(src, sw, sh, dst, dw, dh); }

static inline float texel_u8(const unsigned char *im, int rows, int cols, int y, int x)
{
    return (float)im[(size_t)clampi(y, 0, rows - 1) * cols + clampi(x, 0, cols - 1)] / 255.0f;
}

static float tex_linear_u8(const unsigned char *im, int rows, int cols, int y0, int x0, float fy, float fx)
{
    const float t00 = texel_u8(im, rows, cols, y0, x0), t01 = texel_u8(im, rows, cols, y0, x0 + 1);
    const float t10 = texel_u8(im, rows, cols, y0 + 1, x0), t11 = texel_u8(im, rows, cols, y0 + 1, x0 + 1);
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const float top = t00 * gx + t01 * fx, bot = t10 * gx + t11 * fx;
    return top * gy + bot * fy;
}

static float fold64(float *a)      /* the wave's reduction tree */
{
    for (int s = 32; s >= 1; s >>= 1)
        for (int i = 0; i < s; ++i) a[i] += a[i + s];
    return a[0];
}

/* one point at one level: returns 0 when the kernel returned early (nextPt untouched), 1 when it wrote nextPt (and err) */
static int sparse_point(const unsigned char *I, const unsigned char *J, int rows, int cols, float px, float py, float *nx_io, float *ny_io,
                        int level, int wx, int wy, int iters, unsigned char *status, float *err)
{
    const int hx = (wx - 1) / 2, hy = (wy - 1) / 2, ne = wx * wy;
    px *= (1.0f / (1 << level));
    py *= (1.0f / (1 << level));
    if (px < 0 || px >= cols || py < 0 || py >= rows) { if (level == 0) *status = 0; return 0; }
    px -= hx; py -= hy;
    float *Ip = (float *)malloc(sizeof(float) * 3 * ne), *dx = Ip + ne, *dy = dx + ne;
    float p11[64] = {0}, p12[64] = {0}, p22[64] = {0};
    {
        const float x0f = floorf(px), y0f = floorf(py);
        const float fx = px - x0f, fy = py - y0f;
        const int x0 = (int)x0f, y0 = (int)y0f;
        for (int e = 0; e < ne; ++e) {
            const int i = e / wx, j = e % wx, l = e & 63;
#define S(a, b) tex_linear_u8(I, rows, cols, y0 + i + (a), x0 + j + (b), fy, fx)
            Ip[e] = S(0, 0);
            const float gx = 3.0f * S(-1, 1) + 10.0f * S(0, 1) + 3.0f * S(1, 1) - (3.0f * S(-1, -1) + 10.0f * S(0, -1) + 3.0f * S(1, -1));
            const float gy = 3.0f * S(1, -1) + 10.0f * S(1, 0) + 3.0f * S(1, 1) - (3.0f * S(-1, -1) + 10.0f * S(-1, 0) + 3.0f * S(-1, 1));
#undef S
            dx[e] = gx; dy[e] = gy;
            p11[l] += gx * gx; p12[l] += gx * gy; p22[l] += gy * gy;
        }
    }
    float A11 = fold64(p11), A12 = fold64(p12), A22 = fold64(p22);
    float D = A11 * A22 - A12 * A12;
    if (D < FLT_EPSILON) { if (level == 0) *status = 0; free(Ip); return 0; }
    D = 1.f / D;
    A11 *= D; A12 *= D; A22 *= D;
    float nx = *nx_io * 2.f, ny = *ny_io * 2.f;
    nx -= hx; ny -= hy;
    for (int k = 0; k < iters; ++k) {
        if (nx < -hx || nx >= cols || ny < -hy || ny >= rows) { if (level == 0) *status = 0; free(Ip); return 0; }
        float q1[64] = {0}, q2[64] = {0};
        const float x0f = floorf(nx), y0f = floorf(ny);
        const float fx = nx - x0f, fy = ny - y0f;
        const int x0 = (int)x0f, y0 = (int)y0f;
        for (int e = 0; e < ne; ++e) {
            const float Jv = tex_linear_u8(J, rows, cols, y0 + e / wx, x0 + e % wx, fy, fx);
            const float diff = (Jv - Ip[e]) * 32.0f;
            q1[e & 63] += diff * dx[e]; q2[e & 63] += diff * dy[e];
        }
        const float b1 = fold64(q1), b2 = fold64(q2);
        const float ddx = A12 * b2 - A22 * b1, ddy = A12 * b1 - A11 * b2;
        nx += ddx; ny += ddy;
        if (fabsf(ddx) < 0.01f && fabsf(ddy) < 0.01f) break;
    }
    if (err) {
        float q[64] = {0};
        const float x0f = floorf(nx), y0f = floorf(ny);
        const float fx = nx - x0f, fy = ny - y0f;
        const int x0 = (int)x0f, y0 = (int)y0f;
        for (int e = 0; e < ne; ++e) q[e & 63] += fabsf(tex_linear_u8(J, rows, cols, y0 + e / wx, x0 + e % wx, fy, fx) - Ip[e]);
        *err = fold64(q) / (wx * wy) * 255.0f;
    }
    nx += hx; ny += hy;
    *nx_io = nx; *ny_io = ny;
    free(Ip);
    return 1;
}

/* prev, next: rows x cols uint8; prev_pts / next_pts: n x 2 floats (next_pts read when use_initial_flow); status: n bytes; err: n or NULL */
int orc_pyrlk_sparse(const unsigned char *prev, const unsigned char *next, int rows, int cols, const float *prev_pts, float *next_pts, int n,
                     int wx, int wy, int max_level, int iters, int use_initial_flow, unsigned char *status, float *err)
{
    if (max_level < 0 || !(wx > 2 && wy > 2) || n < 0) return -1;      /* CV_Assert, pyrlk.cpp:156-157 */
    const int nl = max_level + 1;
    unsigned char **P = (unsigned char **)malloc(sizeof(void *) * nl), **N = (unsigned char **)malloc(sizeof(void *) * nl);
    int *pw = (int *)malloc(sizeof(int) * nl), *ph = (int *)malloc(sizeof(int) * nl);
    pw[0] = cols; ph[0] = rows;
    P[0] = (unsigned char *)prev; N[0] = (unsigned char *)next;
    for (int l = 1; l < nl; ++l) {
        pw[l] = (pw[l - 1] + 1) / 2; ph[l] = (ph[l - 1] + 1) / 2;
        P[l] = (unsigned char *)malloc((size_t)pw[l] * ph[l]); N[l] = (unsigned char *)malloc((size_t)pw[l] * ph[l]);
        pyr_down_u8(P[l - 1], pw[l - 1], ph[l - 1], P[l], pw[l], ph[l]);
        pyr_down_u8(N[l - 1], pw[l - 1], ph[l - 1], N[l], pw[l], ph[l]);
    }
    /* cuda::multiply(src, 1.0 / (1 << maxLevel) / 2.0, dst): the scale is a double, the product is rounded to float */
    const double sc = 1.0 / (1 << max_level) / 2.0;
    for (int i = 0; i < 2 * n; ++i) next_pts[i] = (float)((use_initial_flow ? next_pts[i] : prev_pts[i]) * sc);
    for (int i = 0; i < n; ++i) status[i] = 1;
    for (int l = max_level; l >= 0; --l) {
#pragma omp parallel for schedule(dynamic, 16)
        for (int i = 0; i < n; ++i)
            sparse_point(P[l], N[l], ph[l], pw[l], prev_pts[2 * i], prev_pts[2 * i + 1], &next_pts[2 * i], &next_pts[2 * i + 1], l, wx, wy, iters,
                         &status[i], l == 0 && err ? &err[i] : NULL);
    }
    for (int l = 1; l < nl; ++l) { free(P[l]); free(N[l]); }
    free(P); free(N); free(pw); free(ph);
    return 0;
}
