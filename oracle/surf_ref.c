/*
 * oracle/surf_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Sequential CPU restatement of cv::cuda::SURF_CUDA:
 *   modules/xfeatures2d/src/surf.cuda.cpp:134-255     SURF_CUDA_Invoker (limits, octave loop, orientation, descriptors)
 *   modules/xfeatures2d/src/cuda/surf.cu:122-152      icvCalcHaarPatternSum (double accumulation)
 *   modules/xfeatures2d/src/cuda/surf.cu:157-222      icvCalcLayerDetAndTrace
 *   modules/xfeatures2d/src/cuda/surf.cu:227-386      Mask::check, icvFindMaximaInLayer
 *   modules/xfeatures2d/src/cuda/surf.cu:391-511      icvInterpolateKeypoint
 *   modules/xfeatures2d/src/cuda/surf.cu:516-678      icvCalcOrientation
 *   modules/xfeatures2d/src/cuda/surf.cu:683-942      calc_dx_dy, compute_descriptors_64/128, normalize_descriptors
 * Texture-free definitions (point-sampled clamp-addressed reads, bilinear / area patch filters, 3x3 solve)
 * follow the reference's own OpenCL twin, which has no sampler dependence:
 *   modules/xfeatures2d/src/opencl/surf.cl:55-68 (read_sumTex_/read_imgTex_), :413-441 (solve3x3_float),
 *   :873-898 (linearFilter), :900-952 (areaFilter).
 * Sample tables are generated as the CPU class generates them (modules/xfeatures2d/src/surf.cpp:544-565):
 * apt = disc of radius 6 with getGaussianKernel(13, 2.5) weights, DW = outer product of getGaussianKernel(20, 3.3).
 * Deterministic order (the reference appends with atomicInc, surf.cu:341,476: order and, on overflow, subset are
 * run-dependent): candidates in (layer, row, column) scan order per octave, octaves ascending; the first
 * maxCandidates / maxFeatures are kept.  det/trace cells the reference leaves unwritten (stale memory) are 0.
 * Warp reductions (device::reduce<32>) are restated as the shfl_down tree 16,8,4,2,1.
 * Pinning: aloe.png (opencv_extra), which the reference's CUDA test reads, is absent.  The reference's known-answer vectors are
 * for its CPU class (misc/java/test/SURFFeatureDetectorTest.java:52-57, SURFDescriptorExtractorTest.java:43-66); oracle/surfcpu_ref.c
 * restates that class and reproduces them to the last digit, and this restatement is held to it the way the reference's own test
 * holds the CUDA class to the CPU class (test_surf.cuda.cpp:81-170) -- tests/test_zz_surf_cpu_class.py: on the golden cross the same
 * four keypoints (position, size, response, octave within 1e-3; orientation up to an exact two-way tie the two classes break
 * differently), on textured images matched-keypoint ratio 1.00 (> 0.95 required) and descriptor match ratio >= 0.98 (> 0.6 required).
 */
#include "surf_ref.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CV_PI_F 3.14159265f
#define PATCH_SZ 20
#define ORI_SAMPLES 113

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int rn(float v) { return (int)lrintf(v); }   /* __float2int_rn */

int orc_surf_calc_size(int octave, int layer) { return (9 + 6 * layer) << octave; }   /* surf.cu:161-173 */

/* The weight tables of surf.cu (c_aptW :522, c_DW :685-707) are LITERALS; tests/test_ref_pin_cuda.py parses them out of the reference and
 * asserts that this generator reproduces every one bit for bit: they are the outer products, in float, of cv::getGaussianKernel(n,
 * sigma, CV_32F) as OpenCV 2.4 computed it -- exp() in double ROUNDED TO FLOAT, the float terms summed in double, each float term times
 * (1 / sum) in double rounded to float again -- with sigma the float constants of the CPU class (surf.cpp: SURF_ORI_SIGMA = 2.5f,
 * SURF_DESC_SIGMA = 3.3f) promoted to double.  (Today's getGaussianKernel rounds once; 60 / 113 and 188 / 400 entries would differ by
 * 1-2 ulp.) */
static void gauss_tab(int n, float sigma_f, float *k)
{
    const double sigma = (double)sigma_f, scale2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; k[i] = (float)exp(scale2 * x * x); sum += k[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
}

void orc_surf_tables(float aptx[ORI_SAMPLES], float apty[ORI_SAMPLES], float aptw[ORI_SAMPLES], float dw[PATCH_SZ * PATCH_SZ])
{
    float g[13], G[PATCH_SZ];
    gauss_tab(13, 2.5f, g);            /* c_aptW, surf.cu:522 */
    int n = 0;
    for (int i = -6; i <= 6; i++)      /* surf.cpp:546-556 */
        for (int j = -6; j <= 6; j++)
            if (i * i + j * j <= 36) { aptx[n] = (float)i; apty[n] = (float)j; aptw[n++] = g[i + 6] * g[j + 6]; }
    gauss_tab(PATCH_SZ, 3.3f, G);      /* c_DW, surf.cu:685-707 */
    for (int i = 0; i < PATCH_SZ; i++)
        for (int j = 0; j < PATCH_SZ; j++) dw[i * PATCH_SZ + j] = G[i] * G[j];
}

/* cuda::integral: (rows+1) x (cols+1), first row/column 0, modulo 2^32 */
void orc_surf_integral(const uint8_t *img, int rows, int cols, uint32_t *sum)
{
    const int sc = cols + 1;
    memset(sum, 0, sizeof(uint32_t) * (size_t)sc);
    for (int y = 0; y < rows; ++y) {
        uint32_t r = 0;
        sum[(size_t)(y + 1) * sc] = 0;
        for (int x = 0; x < cols; ++x) {
            r += img[(size_t)y * cols + x];
            sum[(size_t)(y + 1) * sc + x + 1] = sum[(size_t)y * sc + x + 1] + r;
        }
    }
}

typedef struct { const uint32_t *s; int rows, cols; } SumTex;   /* image size; the sum is (rows+1)x(cols+1) */
static inline uint32_t tex(const SumTex *t, int y, int x)
{
    return t->s[(size_t)clampi(y, 0, t->rows) * (t->cols + 1) + clampi(x, 0, t->cols)];   /* surf.cl:55-60 */
}

/* surf.cu:122-152 */
static float haar(const SumTex *t, const float (*src)[5], int n, int oldSize, int newSize, int y, int x)
{
    const float ratio = (float)newSize / oldSize;
    double d = 0;
    for (int k = 0; k < n; ++k) {
        const int dx1 = rn(ratio * src[k][0]), dy1 = rn(ratio * src[k][1]), dx2 = rn(ratio * src[k][2]), dy2 = rn(ratio * src[k][3]);
        double tt = 0;
        tt += tex(t, y + dy1, x + dx1);
        tt -= tex(t, y + dy2, x + dx1);
        tt -= tex(t, y + dy1, x + dx2);
        tt += tex(t, y + dy2, x + dx2);
        d += tt * src[k][4] / ((dx2 - dx1) * (dy2 - dy1));
    }
    return (float)d;
}

static const float c_DX[3][5] = {{0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1}};
static const float c_DY[3][5] = {{2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1}};
static const float c_DXY[4][5] = {{1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1}};
static const float c_NX[2][5] = {{0, 0, 2, 4, -1}, {2, 0, 4, 4, 1}};
static const float c_NY[2][5] = {{0, 0, 4, 2, 1}, {0, 2, 4, 4, -1}};

/* surf.cu:175-203.  det/trace: (nOctaveLayers+2)*layer_rows x img_cols (row pitch img_cols), unwritten cells 0 */
void orc_surf_det_trace(const uint32_t *sum, int rows, int cols, int octave, int nOctaveLayers, float *det, float *trace)
{
    const SumTex t = {sum, rows, cols};
    const int layer_rows = rows >> octave;
    memset(det, 0, sizeof(float) * (size_t)(nOctaveLayers + 2) * layer_rows * cols);
    memset(trace, 0, sizeof(float) * (size_t)(nOctaveLayers + 2) * layer_rows * cols);
    for (int layer = 0; layer < nOctaveLayers + 2; ++layer) {
        const int size = orc_surf_calc_size(octave, layer);
        if (!(size <= rows && size <= cols)) continue;
        const int samples_i = 1 + ((rows - size) >> octave), samples_j = 1 + ((cols - size) >> octave);
        const int margin = (size >> 1) >> octave;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < samples_i; ++i)
            for (int j = 0; j < samples_j; ++j) {
                const float dx = haar(&t, c_DX, 3, 9, size, i << octave, j << octave);
                const float dy = haar(&t, c_DY, 3, 9, size, i << octave, j << octave);
                const float dxy = haar(&t, c_DXY, 4, 9, size, i << octave, j << octave);
                det[(size_t)(layer * layer_rows + i + margin) * cols + j + margin] = dx * dy - 0.81f * dxy * dxy;
                trace[(size_t)(layer * layer_rows + i + margin) * cols + j + margin] = dx + dy;
            }
    }
}

/* Mask::check surf.cu:229-261 */
static int mask_check(const SumTex *m, int sum_i, int sum_j, int size)
{
    const float ratio = (float)size / 9.0f;
    const int dx1 = rn(ratio * 0.f), dy1 = rn(ratio * 0.f), dx2 = rn(ratio * 9.f), dy2 = rn(ratio * 9.f);
    float tt = 0, d = 0;
    tt += (float)tex(m, sum_i + dy1, sum_j + dx1);
    tt -= (float)tex(m, sum_i + dy2, sum_j + dx1);
    tt -= (float)tex(m, sum_i + dy1, sum_j + dx2);
    tt += (float)tex(m, sum_i + dy2, sum_j + dx2);
    d += tt * 1.f / ((dx2 - dx1) * (dy2 - dy1));
    return d >= 0.5f;
}

/* surf.cu:263-355 in (layer, i, j) scan order.  cand: int4 {j, i, layer, laplacian}; returns the count (<= max) */
int orc_surf_find_maxima(const float *det, const float *trace, const uint32_t *mask_sum, int rows, int cols, int octave,
                         int nOctaveLayers, float hessianThreshold, int max_candidates, int *cand)
{
    const int layer_rows = rows >> octave, layer_cols = cols >> octave;
    const SumTex m = {mask_sum, rows, cols};
    int n = 0;
#define DET(l, ii, jj) det[(size_t)((l) * layer_rows + clampi(ii, 0, rows - 1)) * cols + clampi(jj, 0, cols - 1)]
    for (int layer = 1; layer <= nOctaveLayers; ++layer) {
        const int size = orc_surf_calc_size(octave, layer);
        const int margin = ((orc_surf_calc_size(octave, layer + 1) >> 1) >> octave) + 1;
        for (int i = margin; i < layer_rows - margin; ++i)
            for (int j = margin; j < layer_cols - margin; ++j) {
                const float v = DET(layer, i, j);
                if (!(v > hessianThreshold)) continue;
                const int sum_i = (i - ((size >> 1) >> octave)) << octave, sum_j = (j - ((size >> 1) >> octave)) << octave;
                if (mask_sum && !mask_check(&m, sum_i, sum_j, size)) continue;
                int ismax = 1;
                for (int dl = -1; dl <= 1 && ismax; ++dl)
                    for (int di = -1; di <= 1 && ismax; ++di)
                        for (int dj = -1; dj <= 1; ++dj) {
                            if (!dl && !di && !dj) continue;
                            if (!(v > DET(layer + dl, i + di, j + dj))) { ismax = 0; break; }
                        }
                if (!ismax) continue;
                if (n < max_candidates) {
                    const float tr = trace[(size_t)(layer * layer_rows + i) * cols + j];
                    cand[4 * n] = j; cand[4 * n + 1] = i; cand[4 * n + 2] = layer; cand[4 * n + 3] = (int)copysignf(1.0f, tr);
                    ++n;
                }
            }
    }
#undef DET
    return n;
}

/* core/cuda/utility.hpp solve3x3<float> (main repo; restated in oracle/refshim/cudashim/opencv2/core/cuda/utility.hpp): Cramer's rule with
 * the reciprocal of the determinant in DOUBLE, the components rounded back to float (surf.cl:413-441 is the all-float twin) */
static int solve3x3(const float A[3][3], const float b[3], float x[3])
{
    const float det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                      A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    if (det != 0) {
        const double invdet = 1.0 / det;
        x[0] = (float)(invdet * (b[0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (b[1] * A[2][2] - A[1][2] * b[2]) +
                         A[0][2] * (b[1] * A[2][1] - A[1][1] * b[2])));
        x[1] = (float)(invdet * (A[0][0] * (b[1] * A[2][2] - A[1][2] * b[2]) - b[0] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                         A[0][2] * (A[1][0] * b[2] - b[1] * A[2][0])));
        x[2] = (float)(invdet * (A[0][0] * (A[1][1] * b[2] - b[1] * A[2][1]) - A[0][1] * (A[1][0] * b[2] - b[1] * A[2][0]) +
                         b[0] * (A[1][0] * A[2][1] - A[1][1] * A[2][0])));
        return 1;
    }
    return 0;
}

/* surf.cu:391-493 for one candidate; returns 1 and fills f[6] = {x, y, laplacian, octave, size, hessian} when accepted */
int orc_surf_interpolate(const float *det, int rows, int cols, int octave, const int cand[4], float f[6])
{
    const int layer_rows = rows >> octave;
    float N9[3][3][3];
    for (int z = 0; z < 3; ++z)
        for (int y = 0; y < 3; ++y)
            for (int x = 0; x < 3; ++x)
                N9[z][y][x] = det[(size_t)(layer_rows * (cand[2] - 1 + z) + cand[1] - 1 + y) * cols + cand[0] - 1 + x];
    float dD[3], H[3][3], x[3];
    dD[0] = -0.5f * (N9[1][1][2] - N9[1][1][0]);
    dD[1] = -0.5f * (N9[1][2][1] - N9[1][0][1]);
    dD[2] = -0.5f * (N9[2][1][1] - N9[0][1][1]);
    H[0][0] = N9[1][1][0] - 2.0f * N9[1][1][1] + N9[1][1][2];
    H[0][1] = 0.25f * (N9[1][2][2] - N9[1][2][0] - N9[1][0][2] + N9[1][0][0]);
    H[0][2] = 0.25f * (N9[2][1][2] - N9[2][1][0] - N9[0][1][2] + N9[0][1][0]);
    H[1][0] = H[0][1];
    H[1][1] = N9[1][0][1] - 2.0f * N9[1][1][1] + N9[1][2][1];
    H[1][2] = 0.25f * (N9[2][2][1] - N9[2][0][1] - N9[0][2][1] + N9[0][0][1]);
    H[2][0] = H[0][2];
    H[2][1] = H[1][2];
    H[2][2] = N9[0][1][1] - 2.0f * N9[1][1][1] + N9[2][1][1];
    if (!solve3x3(H, dD, x)) return 0;
    if (!(fabsf(x[0]) <= 1.f && fabsf(x[1]) <= 1.f && fabsf(x[2]) <= 1.f)) return 0;
    const int size = orc_surf_calc_size(octave, cand[2]);
    const int sum_i = (cand[1] - ((size >> 1) >> octave)) << octave, sum_j = (cand[0] - ((size >> 1) >> octave)) << octave;
    const float center_i = sum_i + (float)(size - 1) / 2, center_j = sum_j + (float)(size - 1) / 2;
    const float px = center_j + x[0] * (1 << octave), py = center_i + x[1] * (1 << octave);
    const int ds = size - orc_surf_calc_size(octave, cand[2] - 1);
    const float psize = roundf(size + x[2] * ds);
    const float s = psize * 1.2f / 9.0f;
    const int grad_wav_size = 2 * rn(2.0f * s);
    if (!((rows + 1) >= grad_wav_size && (cols + 1) >= grad_wav_size)) return 0;
    f[0] = px; f[1] = py; f[2] = (float)cand[3]; f[3] = (float)octave; f[4] = psize; f[5] = N9[1][1][1];
    return 1;
}

/* device::reduce<32> with plus<float>: val += shfl_down(val, 16), 8, 4, 2, 1; result of lane 0 */
static float reduce32(float v[32])
{
    for (int off = 16; off >= 1; off >>= 1)
        for (int i = 0; i < off; ++i) v[i] = v[i] + v[i + off];
    return v[0];
}

/* surf.cu:527-658 */
float orc_surf_orientation(const uint32_t *sum, int rows, int cols, float fx, float fy, float fsize, const float *aptx,
                           const float *apty, const float *aptw)
{
    const SumTex t = {sum, rows, cols};
    const float s = fsize * 1.2f / 9.0f;
    const int grad_wav_size = 2 * rn(2.0f * s);
    if ((rows + 1) < grad_wav_size || (cols + 1) < grad_wav_size) return 0.f;   /* kernel returns without writing (row stays 0) */
    float sX[128], sY[128], sA[128];
    for (int tid = 0; tid < 128; ++tid) {
        float X = 0.f, Y = 0.f, angle = 0.f;
        if (tid < ORI_SAMPLES) {
            const float margin = (float)(grad_wav_size - 1) / 2.0f;
            const int x = rn(fx + aptx[tid] * s - margin), y = rn(fy + apty[tid] * s - margin);
            if (y >= 0 && y < (rows + 1) - grad_wav_size && x >= 0 && x < (cols + 1) - grad_wav_size) {
                X = aptw[tid] * haar(&t, c_NX, 2, 4, grad_wav_size, y, x);
                Y = aptw[tid] * haar(&t, c_NY, 2, 4, grad_wav_size, y, x);
                angle = atan2f(Y, X);
                if (angle < 0) angle += 2.0f * CV_PI_F;
                angle *= 180.0f / CV_PI_F;
            }
        }
        sX[tid] = X; sY[tid] = Y; sA[tid] = angle;
    }
    float bx[4], by[4], bm[4];
    for (int ty = 0; ty < 4; ++ty) {
        float bestx = 0, besty = 0, best_mod = 0;
        for (int i = 0; i < 18; ++i) {
            const int dir = (i * 4 + ty) * 5;
            float vx[32], vy[32];
            for (int tx = 0; tx < 32; ++tx) {
                float sumx = 0.f, sumy = 0.f;
                for (int q = 0; q < 4; ++q) {
                    const int d = abs(rn(sA[tx + 32 * q]) - dir);
                    if (d < 30 || d > 330) {
                        if (q == 0) { sumx = sX[tx]; sumy = sY[tx]; }
                        else { sumx += sX[tx + 32 * q]; sumy += sY[tx + 32 * q]; }
                    }
                }
                vx[tx] = sumx; vy[tx] = sumy;
            }
            const float sumx = reduce32(vx), sumy = reduce32(vy);
            const float temp_mod = sumx * sumx + sumy * sumy;
            if (temp_mod > best_mod) { best_mod = temp_mod; bestx = sumx; besty = sumy; }
        }
        bx[ty] = bestx; by[ty] = besty; bm[ty] = best_mod;
    }
    int bestIdx = 0;
    if (bm[1] > bm[bestIdx]) bestIdx = 1;
    if (bm[2] > bm[bestIdx]) bestIdx = 2;
    if (bm[3] > bm[bestIdx]) bestIdx = 3;
    float kp_dir = atan2f(by[bestIdx], bx[bestIdx]);
    if (kp_dir < 0) kp_dir += 2.0f * CV_PI_F;
    kp_dir *= 180.0f / CV_PI_F;
    kp_dir = 360.0f - kp_dir;
    if (fabsf(kp_dir - 360.f) < FLT_EPSILON) kp_dir = 0.f;
    return kp_dir;
}

/* ---- descriptor: rotated window reader + patch filters (surf.cu:709-778).  WinReader reads the 8-bit image through a point-sampled,
 * clamp-addressed texture: texel (floor(x), floor(y)) (CUDA programming guide, nearest-point sampling of unnormalised coordinates).  The two
 * patch filters are core/cuda/filters.hpp's (main repo; restated in oracle/refshim/cudashim/opencv2/core/cuda/filters.hpp): float
 * accumulation in the order below, then saturate_cast<elem_type> -- WinReader::elem_type is uchar (surf.cu:711), so every patch sample is
 * ROUNDED to an 8-bit value (cvt.rni.sat.u8.f32); LinearFilter takes floor(x), floor(y); AreaFilter normalises by
 * 1 / (min(s, width - fsx1) * min(s, height - fsy1)) with width = height = win_size (surf.cu:754) -- smaller than s * s in the last
 * patch row / column.  (The OpenCL twin, surf.cl:62-68,873-952, rounds the coordinates, keeps the samples in float and divides by s * s.) */
typedef struct { const uint8_t *img; int rows, cols, win; float cx, cy, off, c, s; } Win;
static inline float win_get(const Win *w, int i, int j)
{
    const float px = w->cx + (w->off + j) * w->c + (w->off + i) * w->s;
    const float py = w->cy - (w->off + j) * w->s + (w->off + i) * w->c;
    const int x = clampi((int)floorf(px), 0, w->cols - 1), y = clampi((int)floorf(py), 0, w->rows - 1);
    return (float)w->img[(size_t)y * w->cols + x];
}
static inline float sat_u8(float v)   /* saturate_cast<uchar>(float): round to nearest even, saturate */
{
    const float r = nearbyintf(v);
    return r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
}
static float linear_filter(const Win *w, float y, float x)
{
    float out = 0.0f;
    const int x1 = (int)floorf(x), y1 = (int)floorf(y), x2 = x1 + 1, y2 = y1 + 1;
    out = out + win_get(w, y1, x1) * ((x2 - x) * (y2 - y));
    out = out + win_get(w, y1, x2) * ((x - x1) * (y2 - y));
    out = out + win_get(w, y2, x1) * ((x2 - x) * (y - y1));
    out = out + win_get(w, y2, x2) * ((x - x1) * (y - y1));
    return sat_u8(out);
}
static float area_filter(const Win *w, float x, float y, float s)
{
    const float fsx1 = x * s, fsx2 = fsx1 + s;
    const int sx1 = (int)ceilf(fsx1), sx2 = (int)floorf(fsx2);
    const float fsy1 = y * s, fsy2 = fsy1 + s;
    const int sy1 = (int)ceilf(fsy1), sy2 = (int)floorf(fsy2);
    const float scale = 1.f / (fminf(s, w->win - fsx1) * fminf(s, w->win - fsy1));
    float out = 0.f;
    for (int dy = sy1; dy < sy2; ++dy) {
        for (int dx = sx1; dx < sx2; ++dx) out = out + win_get(w, dy, dx) * scale;
        if (sx1 > fsx1) out = out + win_get(w, dy, sx1 - 1) * ((sx1 - fsx1) * scale);
        if (sx2 < fsx2) out = out + win_get(w, dy, sx2) * ((fsx2 - sx2) * scale);
    }
    if (sy1 > fsy1) for (int dx = sx1; dx < sx2; ++dx) out = out + win_get(w, sy1 - 1, dx) * ((sy1 - fsy1) * scale);
    if (sy2 < fsy2) for (int dx = sx1; dx < sx2; ++dx) out = out + win_get(w, sy2, dx) * ((fsy2 - sy2) * scale);
    if ((sy1 > fsy1) && (sx1 > fsx1)) out = out + win_get(w, sy1 - 1, sx1 - 1) * ((sy1 - fsy1) * (sx1 - fsx1) * scale);
    if ((sy1 > fsy1) && (sx2 < fsx2)) out = out + win_get(w, sy1 - 1, sx2) * ((sy1 - fsy1) * (fsx2 - sx2) * scale);
    if ((sy2 < fsy2) && (sx2 < fsx2)) out = out + win_get(w, sy2, sx2) * ((fsy2 - sy2) * (fsx2 - sx2) * scale);
    if ((sy2 < fsy2) && (sx1 > fsx1)) out = out + win_get(w, sy2, sx1 - 1) * ((fsy2 - sy2) * (sx1 - fsx1) * scale);
    return sat_u8(out);
}

/* surf.cu:733-912: one descriptor (64 or 128 floats), normalised */
void orc_surf_descriptor(const uint8_t *img, int rows, int cols, float fx, float fy, float fsize, float fdir, int extended,
                         const float *dw, float *desc)
{
    Win w;
    w.img = img; w.rows = rows; w.cols = cols; w.cx = fx; w.cy = fy;
    const float s = fsize * 1.2f / 9.0f;
    const int win_size = (int)((PATCH_SZ + 1) * s);
    w.win = win_size;
    w.off = -(win_size - 1.0f) / 2.0f;
    float ddir = 360.0f - fdir;
    if (fabsf(ddir - 360.f) < FLT_EPSILON) ddir = 0.f;
    ddir *= CV_PI_F / 180.0f;
    w.s = sinf(ddir); w.c = cosf(ddir);
    float P[PATCH_SZ + 1][PATCH_SZ + 1];
    for (int yl = 0; yl <= PATCH_SZ; ++yl)
        for (int xl = 0; xl <= PATCH_SZ; ++xl)
            P[yl][xl] = s > 1 ? area_filter(&w, (float)xl, (float)yl, s) : linear_filter(&w, yl * s, xl * s);
    const int dsz = extended ? 128 : 64;
    for (int ty = 0; ty < 16; ++ty) {          /* one 5x5 sub-region per threadIdx.y */
        const int xb = ty % 4, yb = ty / 4;
        float vdx[32], vdy[32];
        for (int tx = 0; tx < 32; ++tx) {
            float dx = 0.f, dy = 0.f;
            const int xp = tx % 5, yp = tx / 5;
            if (yp < 5) {
                const int xi = xb * 5 + xp, yi = yb * 5 + yp;
                const float wgt = dw[yi * PATCH_SZ + xi];
                dx = (P[yi][xi + 1] - P[yi][xi] + P[yi + 1][xi + 1] - P[yi + 1][xi]) * wgt;
                dy = (P[yi + 1][xi] - P[yi][xi] + P[yi + 1][xi + 1] - P[yi][xi + 1]) * wgt;
            }
            vdx[tx] = dx; vdy[tx] = dy;
        }
        float a[32], b[32], c[32], d[32];
        if (!extended) {
            for (int k = 0; k < 32; ++k) { a[k] = vdx[k]; b[k] = vdy[k]; c[k] = fabsf(vdx[k]); d[k] = fabsf(vdy[k]); }
            desc[ty * 4 + 0] = reduce32(a); desc[ty * 4 + 1] = reduce32(b); desc[ty * 4 + 2] = reduce32(c); desc[ty * 4 + 3] = reduce32(d);
        } else {
            for (int k = 0; k < 32; ++k) {
                const int pos = vdy[k] >= 0;
                a[k] = pos ? vdx[k] : 0.f; b[k] = pos ? fabsf(vdx[k]) : 0.f; c[k] = pos ? 0.f : vdx[k]; d[k] = pos ? 0.f : fabsf(vdx[k]);
            }
            desc[ty * 8 + 0] = reduce32(a); desc[ty * 8 + 1] = reduce32(b); desc[ty * 8 + 2] = reduce32(c); desc[ty * 8 + 3] = reduce32(d);
            for (int k = 0; k < 32; ++k) {
                const int pos = vdx[k] >= 0;
                a[k] = pos ? vdy[k] : 0.f; b[k] = pos ? fabsf(vdy[k]) : 0.f; c[k] = pos ? 0.f : vdy[k]; d[k] = pos ? 0.f : fabsf(vdy[k]);
            }
            desc[ty * 8 + 4] = reduce32(a); desc[ty * 8 + 5] = reduce32(b); desc[ty * 8 + 6] = reduce32(c); desc[ty * 8 + 7] = reduce32(d);
        }
    }
    /* normalize_descriptors<N> surf.cu:891-912: len = device::reduce<N> of the squares, N = 64 / 128 threads = 2 / 4 warps
     * (core/cuda/detail/reduce.hpp GenericOptimized32 on sm_30+: the shfl_down tree in every warp, then the tree of width N / 32 over
     * the warps' partials in the first warp); val / sqrt(len) */
    float part[4];
    for (int wv = 0; wv < dsz / 32; ++wv) {
        float sq[32];
        for (int k = 0; k < 32; ++k) sq[k] = desc[wv * 32 + k] * desc[wv * 32 + k];
        part[wv] = reduce32(sq);
    }
    for (int off = dsz / 64; off >= 1; off >>= 1)
        for (int i = 0; i < off; ++i) part[i] = part[i] + part[i + off];
    const float len = sqrtf(part[0]);
    for (int k = 0; k < dsz; ++k) desc[k] = desc[k] / len;
}

void orc_surf_default_params(orc_surf_params *p)
{
    /* SURF_CUDA::create(thr, 4, 2, false, 0.01f, false), cuda.hpp:117-118 */
    p->hessian_threshold = 100; p->n_octaves = 4; p->n_octave_layers = 2; p->extended = 0; p->keypoints_ratio = 0.01f; p->upright = 0;
}

/* surf.cuda.cpp:137-236.  keypoints: 7 rows x max_features (row pitch max_features); returns nFeatures or < 0 */
int orc_surf_detect_describe(const orc_surf_params *P, const uint8_t *img, const uint8_t *mask, int rows, int cols,
                             float *keypoints, int kp_pitch, float *descriptors, int want_desc)
{
    if (!(P->n_octaves > 0 && P->n_octave_layers > 0)) return -1;                 /* :141 */
    const int min_size = orc_surf_calc_size(P->n_octaves - 1, 0);
    if (rows - min_size < 0 || cols - min_size < 0) return -3;                     /* :144-145 */
    const int lr = rows >> (P->n_octaves - 1), lc = cols >> (P->n_octaves - 1);
    const int min_margin = ((orc_surf_calc_size(P->n_octaves - 1, 2) >> 1) >> (P->n_octaves - 1)) + 1;
    if (!(lr - 2 * min_margin > 0 && lc - 2 * min_margin > 0)) return -3;          /* :150-151 */
    int maxFeatures = (int)((float)(rows * cols) * P->keypoints_ratio);            /* static_cast<int>(area * ratio) :153 */
    if (maxFeatures > 65535) maxFeatures = 65535;
    int maxCandidates = (int)(1.5 * maxFeatures);
    if (maxCandidates > 65535) maxCandidates = 65535;
    if (!(maxFeatures > 0)) return -1;                                              /* :156 */
    if (kp_pitch < maxFeatures) return -1;

    const size_t sn = (size_t)(rows + 1) * (cols + 1);
    uint32_t *sum = (uint32_t *)malloc(sizeof(uint32_t) * sn), *msum = NULL;
    orc_surf_integral(img, rows, cols, sum);
    if (mask) {
        uint8_t *m1 = (uint8_t *)malloc((size_t)rows * cols);
        for (size_t i = 0; i < (size_t)rows * cols; ++i) m1[i] = mask[i] < 1 ? mask[i] : 1;   /* cuda::min(mask, 1.0) :167 */
        msum = (uint32_t *)malloc(sizeof(uint32_t) * sn);
        orc_surf_integral(m1, rows, cols, msum);
        free(m1);
    }
    float aptx[ORI_SAMPLES], apty[ORI_SAMPLES], aptw[ORI_SAMPLES], dw[PATCH_SZ * PATCH_SZ];
    orc_surf_tables(aptx, apty, aptw, dw);
    float *det = (float *)malloc(sizeof(float) * (size_t)rows * (P->n_octave_layers + 2) * cols);
    float *trace = (float *)malloc(sizeof(float) * (size_t)rows * (P->n_octave_layers + 2) * cols);
    int *cand = (int *)malloc(sizeof(int) * 4 * (size_t)maxCandidates);
    memset(keypoints, 0, sizeof(float) * 7 * (size_t)kp_pitch);
    int nf = 0;
    for (int octave = 0; octave < P->n_octaves; ++octave) {
        orc_surf_det_trace(sum, rows, cols, octave, P->n_octave_layers, det, trace);
        const int nc = orc_surf_find_maxima(det, trace, msum, rows, cols, octave, P->n_octave_layers, (float)P->hessian_threshold,
                                            maxCandidates, cand);
        for (int c = 0; c < nc; ++c) {
            float f[6];
            if (!orc_surf_interpolate(det, rows, cols, octave, cand + 4 * c, f)) continue;
            if (nf < maxFeatures) {
                keypoints[0 * kp_pitch + nf] = f[0];
                keypoints[1 * kp_pitch + nf] = f[1];
                ((int *)keypoints)[2 * kp_pitch + nf] = (int)f[2];    /* LAPLACIAN/OCTAVE rows hold int bit patterns (cuda.hpp:89-99) */
                ((int *)keypoints)[3 * kp_pitch + nf] = octave;
                keypoints[4 * kp_pitch + nf] = f[4];
                keypoints[6 * kp_pitch + nf] = f[5];
                ++nf;
            }
        }
    }
    for (int i = 0; i < nf; ++i)
        keypoints[5 * kp_pitch + i] = P->upright ? 270.f
            : orc_surf_orientation(sum, rows, cols, keypoints[i], keypoints[kp_pitch + i], keypoints[4 * kp_pitch + i], aptx, apty, aptw);
    if (want_desc && descriptors) {
        const int dsz = P->extended ? 128 : 64;
#pragma omp parallel for schedule(dynamic, 8)
        for (int i = 0; i < nf; ++i)
            orc_surf_descriptor(img, rows, cols, keypoints[i], keypoints[kp_pitch + i], keypoints[4 * kp_pitch + i],
                                keypoints[5 * kp_pitch + i], P->extended, dw, descriptors + (size_t)i * dsz);
    }
    free(sum); free(msum); free(det); free(trace); free(cand);
    return nf;
}
