/* CPU restatement of the reference's CPU SURF class, xfeatures2d::SURF_Impl -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Why it exists: the product (and oracle/surf_ref.c) follow the CUDA class cv::cuda::SURF_CUDA.  The reference's own test of that
 * class (xfeatures2d/test/test_surf.cuda.cpp:81-170) accepts it by comparing with THIS class (keypoints matched > 95 %, descriptors
 * nearest-neighbour matched > 60 %), and the reference holds known-answer vectors for this class only
 * (misc/java/test/SURFFeatureDetectorTest.java:52-57 keypoints, SURFDescriptorExtractorTest.java:43-66 a 128-float descriptor).
 * tests/test_surf.py pins this restatement on those vectors (tolerance 1e-3, the Java tests' EPS) and then applies the
 * reference's acceptance comparison between the two classes.
 *
 * Follows modules/xfeatures2d/src/surf.cpp: resizeHaarPattern :145-162, calcHaarPattern :137-143, calcLayerDetAndTrace :168-213,
 * interpolateKeypoint :236-265, findMaximaInLayer :352-453, KeypointGreater :455-470, fastHessianDetector :473-527,
 * SURFInvoker :530-866 (orientation :612-667, window extraction :672-767, descriptor :768-849), detectAndCompute :881-1015.
 * Main-repository functions it needs (opencv/opencv, un-vendored) are restated from their published definitions: integral,
 * getGaussianKernel, cvRound (round half to even), fastAtan2 / phase (7th-order odd polynomial, core/src/mathfuncs_core),
 * Matx33f::solve(DECOMP_LU) (Cramer's rule with one reciprocal, core/operations.hpp), resize(INTER_AREA) of an 8-bit image
 * (imgproc/src/resize.cpp computeResizeAreaTab + ResizeArea_: float accumulation, horizontal then vertical, saturate_cast).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORI_RADIUS 6
#define ORI_WIN 60
#define PATCH_SZ 20
#define ORI_SEARCH_INC 5
#define HAAR_SIZE0 9
#define HAAR_SIZE_INC 6

typedef struct { int p0, p1, p2, p3; float w; } hf_t;

static int cv_round(double v) { return (int)nearbyint(v); }       /* default rounding mode: half to even */
static int cv_floor(double v) { return (int)floor(v); }
static int cv_ceil(double v) { return (int)ceil(v); }

/* cv::fastAtan2 / cv::phase(..., angleInDegrees = true) */
static float fast_atan2(float y, float x)
{
    const float k = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k, p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static void gauss_kernel(int n, double sigma, float *k)   /* getGaussianKernel(n, sigma, CV_32F) */
{
    const double scale2 = -0.5 / (sigma * sigma);
    double w[64], sum = 0;
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; w[i] = exp(scale2 * x * x); sum += w[i]; }
    sum = 1. / sum;   /* the main repo multiplies by the reciprocal (getGaussianKernelBitExact); so does refshim/cudahost's restatement */
    for (int i = 0; i < n; ++i) k[i] = (float)(w[i] * sum);
}

static void integral_u8(const uint8_t *img, int rows, int cols, int *sum)   /* (rows+1) x (cols+1), CV_32S */
{
    const int sw = cols + 1;
    memset(sum, 0, sizeof(int) * (size_t)sw);
    for (int y = 0; y < rows; ++y) {
        int run = 0;
        sum[(size_t)(y + 1) * sw] = 0;
        for (int x = 0; x < cols; ++x) {
            run += img[(size_t)y * cols + x];
            sum[(size_t)(y + 1) * sw + x + 1] = sum[(size_t)y * sw + x + 1] + run;
        }
    }
}

/* surf.cpp:145-162 */
static void resize_haar(const int src[][5], hf_t *dst, int n, int old_size, int new_size, int width_step)
{
    const float ratio = (float)new_size / old_size;
    for (int k = 0; k < n; ++k) {
        const int dx1 = cv_round(ratio * src[k][0]), dy1 = cv_round(ratio * src[k][1]);
        const int dx2 = cv_round(ratio * src[k][2]), dy2 = cv_round(ratio * src[k][3]);
        dst[k].p0 = dy1 * width_step + dx1;
        dst[k].p1 = dy2 * width_step + dx1;
        dst[k].p2 = dy1 * width_step + dx2;
        dst[k].p3 = dy2 * width_step + dx2;
        dst[k].w = src[k][4] / ((float)(dx2 - dx1) * (dy2 - dy1));
    }
}

/* surf.cpp:137-143: the box sum times its weight is a float product, accumulated in double */
static float haar(const int *origin, const hf_t *f, int n)
{
    double d = 0;
    for (int k = 0; k < n; ++k) d += (origin[f[k].p0] + origin[f[k].p3] - origin[f[k].p1] - origin[f[k].p2]) * f[k].w;
    return (float)d;
}

/* surf.cpp:168-213; det / trace: (rows / step) x (cols / step) */
static void layer_det_trace(const int *sum, int rows, int cols, int size, int step, float *det, float *trace)
{
    static const int dx_s[3][5] = {{0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1}};
    static const int dy_s[3][5] = {{2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1}};
    static const int dxy_s[4][5] = {{1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1}};
    hf_t Dx[3], Dy[3], Dxy[4];
    const int sw = cols + 1, lc = cols / step;
    if (size > rows || size > cols) return;
    resize_haar(dx_s, Dx, 3, 9, size, sw);
    resize_haar(dy_s, Dy, 3, 9, size, sw);
    resize_haar(dxy_s, Dxy, 4, 9, size, sw);
    const int samples_i = 1 + (rows - size) / step, samples_j = 1 + (cols - size) / step;
    const int margin = (size / 2) / step;
    for (int i = 0; i < samples_i; ++i) {
        const int *sp = sum + (size_t)(i * step) * sw;
        float *dp = det + (size_t)(i + margin) * lc + margin, *tp = trace + (size_t)(i + margin) * lc + margin;
        for (int j = 0; j < samples_j; ++j) {
            const float dx = haar(sp, Dx, 3), dy = haar(sp, Dy, 3), dxy = haar(sp, Dxy, 4);
            sp += step;
            dp[j] = dx * dy - 0.81f * dxy * dxy;
            tp[j] = dx + dy;
        }
    }
}

typedef struct { float x, y, size, angle, response; int octave, class_id; } kp_t;

/* surf.cpp:236-265 */
static int interpolate(float N9[3][9], int dx, int dy, int ds, kp_t *k)
{
    const float b0 = -(N9[1][5] - N9[1][3]) / 2, b1 = -(N9[1][7] - N9[1][1]) / 2, b2 = -(N9[2][4] - N9[0][4]) / 2;
    const float a00 = N9[1][3] - 2 * N9[1][4] + N9[1][5];
    const float a01 = (N9[1][8] - N9[1][6] - N9[1][2] + N9[1][0]) / 4;
    const float a02 = (N9[2][5] - N9[2][3] - N9[0][5] + N9[0][3]) / 4;
    const float a11 = N9[1][1] - 2 * N9[1][4] + N9[1][7];
    const float a12 = (N9[2][7] - N9[2][1] - N9[0][7] + N9[0][1]) / 4;
    const float a22 = N9[0][4] - 2 * N9[1][4] + N9[2][4];
    const float a10 = a01, a20 = a02, a21 = a12;
    /* Matx33f::solve(b, DECOMP_LU): 3 x 3 fast path = Cramer's rule */
    float d = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
    float x0 = 0, x1 = 0, x2 = 0;
    if (d != 0) {
        d = 1 / d;
        x0 = d * (b0 * (a11 * a22 - a12 * a21) - a01 * (b1 * a22 - a12 * b2) + a02 * (b1 * a21 - a11 * b2));
        x1 = d * (a00 * (b1 * a22 - a12 * b2) - b0 * (a10 * a22 - a12 * a20) + a02 * (a10 * b2 - b1 * a20));
        x2 = d * (a00 * (a11 * b2 - b1 * a21) - a01 * (a10 * b2 - b1 * a20) + b0 * (a10 * a21 - a11 * a20));
    }
    const int ok = (x0 != 0 || x1 != 0 || x2 != 0) && fabsf(x0) <= 1 && fabsf(x1) <= 1 && fabsf(x2) <= 1;
    if (ok) {
        k->x += x0 * dx;
        k->y += x1 * dy;
        k->size = (float)cv_round(k->size + x2 * ds);
    }
    return ok;
}

/* KeypointGreater, surf.cpp:455-470 (as a qsort comparator: "greater" first) */
static int kp_cmp(const void *pa, const void *pb)
{
    const kp_t *a = (const kp_t *)pa, *b = (const kp_t *)pb;
    if (a->response > b->response) return -1;
    if (a->response < b->response) return 1;
    if (a->size > b->size) return -1;
    if (a->size < b->size) return 1;
    if (a->octave > b->octave) return -1;
    if (a->octave < b->octave) return 1;
    if (a->y < b->y) return 1;
    if (a->y > b->y) return -1;
    return a->x < b->x ? -1 : (a->x > b->x ? 1 : 0);
}

/* fastHessianDetector + the mask handling of detectAndCompute (surf.cpp:473-527, 938-958).
 * kp: cap x 7 floats {x, y, size, angle = -1, response, octave, class_id}; returns the number found (may exceed cap: truncated). */
int orc_surfcpu_detect(const uint8_t *img, const uint8_t *mask, int rows, int cols, float hessian_threshold, int n_octaves,
                       int n_octave_layers, float *kp, int cap)
{
    if (!img || rows <= 0 || cols <= 0 || n_octaves <= 0 || n_octave_layers <= 0 || hessian_threshold < 0) return -1;
    const int sw = cols + 1;
    int *sum = (int *)malloc(sizeof(int) * (size_t)(rows + 1) * sw), *msum = NULL;
    integral_u8(img, rows, cols, sum);
    if (mask) {
        uint8_t *m1 = (uint8_t *)malloc((size_t)rows * cols);
        for (size_t i = 0; i < (size_t)rows * cols; ++i) m1[i] = mask[i] ? 1 : 0;      /* cv::min(mask, 1, mask1) */
        msum = (int *)malloc(sizeof(int) * (size_t)(rows + 1) * sw);
        integral_u8(m1, rows, cols, msum);
        free(m1);
    }
    const int n_total = (n_octave_layers + 2) * n_octaves;
    float **dets = (float **)calloc((size_t)n_total, sizeof(float *)), **traces = (float **)calloc((size_t)n_total, sizeof(float *));
    int *sizes = (int *)malloc(sizeof(int) * (size_t)n_total), *steps = (int *)malloc(sizeof(int) * (size_t)n_total);
    int index = 0, step = 1;      /* SAMPLE_STEP0 */
    for (int o = 0; o < n_octaves; ++o) {
        for (int l = 0; l < n_octave_layers + 2; ++l, ++index) {
            const size_t n = (size_t)(rows / step) * (cols / step);
            dets[index] = (float *)calloc(n ? n : 1, sizeof(float));      /* the reference leaves unwritten cells uninitialised */
            traces[index] = (float *)calloc(n ? n : 1, sizeof(float));
            sizes[index] = (HAAR_SIZE0 + HAAR_SIZE_INC * l) << o;
            steps[index] = step;
            layer_det_trace(sum, rows, cols, sizes[index], step, dets[index], traces[index]);
        }
        step *= 2;
    }
    int n = 0, alloc = 1024;
    kp_t *out = (kp_t *)malloc(sizeof(kp_t) * (size_t)alloc);
    for (int o = 0; o < n_octaves; ++o) {
        for (int ml = 1; ml <= n_octave_layers; ++ml) {          /* middle layers, surf.cpp:352-453 */
            const int layer = o * (n_octave_layers + 2) + ml;
            const int size = sizes[layer], st = steps[layer];
            const int lr = rows / st, lc = cols / st;
            const int margin = (sizes[layer + 1] / 2) / st + 1;
            static const int dm[1][5] = {{0, 0, 9, 9, 1}};
            hf_t Dm = {0, 0, 0, 0, 0.f};
            if (msum) resize_haar(dm, &Dm, 1, 9, size, sw);
            for (int i = margin; i < lr - margin; ++i) {
                for (int j = margin; j < lc - margin; ++j) {
                    const float val0 = dets[layer][(size_t)i * lc + j];
                    if (!(val0 > hessian_threshold)) continue;
                    const int sum_i = st * (i - (size / 2) / st), sum_j = st * (j - (size / 2) / st);
                    float N9[3][9];
                    for (int d = 0; d < 3; ++d) {
                        const float *p = dets[layer - 1 + d] + (size_t)i * lc + j;
                        N9[d][0] = p[-lc - 1]; N9[d][1] = p[-lc]; N9[d][2] = p[-lc + 1];
                        N9[d][3] = p[-1]; N9[d][4] = p[0]; N9[d][5] = p[1];
                        N9[d][6] = p[lc - 1]; N9[d][7] = p[lc]; N9[d][8] = p[lc + 1];
                    }
                    if (msum) {
                        const float mval = haar(msum + (size_t)sum_i * sw + sum_j, &Dm, 1);
                        if (mval < 0.5) continue;
                    }
                    int is_max = 1;
                    for (int d = 0; d < 3 && is_max; ++d)
                        for (int e = 0; e < 9; ++e)
                            if (!(d == 1 && e == 4) && !(val0 > N9[d][e])) { is_max = 0; break; }
                    if (!is_max) continue;
                    const float tr = traces[layer][(size_t)i * lc + j];
                    kp_t k = {sum_j + (size - 1) * 0.5f, sum_i + (size - 1) * 0.5f, (float)size, -1.f, val0, o, (tr > 0) - (tr < 0)};
                    if (!interpolate(N9, st, st, size - sizes[layer - 1], &k)) continue;
                    if (n == alloc) { alloc *= 2; out = (kp_t *)realloc(out, sizeof(kp_t) * (size_t)alloc); }
                    out[n++] = k;
                }
            }
        }
    }
    qsort(out, (size_t)n, sizeof(kp_t), kp_cmp);
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (mask) {      /* Point pt(keypoints[i].pt): rounds; mask.at<uchar>(pt.y, pt.x) == 0 -> erase */
            const int px = cv_round(out[i].x), py = cv_round(out[i].y);
            if (px >= 0 && px < cols && py >= 0 && py < rows && mask[(size_t)py * cols + px] == 0) continue;
        }
        if (m < cap) {
            float *r = kp + (size_t)m * 7;
            r[0] = out[i].x; r[1] = out[i].y; r[2] = out[i].size; r[3] = out[i].angle; r[4] = out[i].response;
            r[5] = (float)out[i].octave; r[6] = (float)out[i].class_id;
        }
        ++m;
    }
    for (int i = 0; i < n_total; ++i) { free(dets[i]); free(traces[i]); }
    free(dets); free(traces); free(sizes); free(steps); free(out); free(sum); free(msum);
    return m;
}

/* cv::resize(src (n x n, 8U), dst (m x m), INTER_AREA) for n >= m: imgproc resize.cpp, computeResizeAreaTab + ResizeArea_ */
typedef struct { int si, di; float alpha; } area_tab_t;

static int area_tab(int ssize, int dsize, double scale, area_tab_t *tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; ++dx) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = fmin(scale, ssize - fsx1);
        int sx1 = cv_ceil(fsx1), sx2 = cv_floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) { tab[k].di = dx; tab[k].si = sx1 - 1; tab[k++].alpha = (float)((sx1 - fsx1) / cell); }
        for (int sx = sx1; sx < sx2; ++sx) { tab[k].di = dx; tab[k].si = sx; tab[k++].alpha = (float)(1.0 / cell); }
        if (fsx2 - sx2 > 1e-3) { tab[k].di = dx; tab[k].si = sx2; tab[k++].alpha = (float)(fmin(fmin(fsx2 - sx2, 1.), cell) / cell); }
    }
    return k;
}

static int resize_area_u8(const uint8_t *src, int n, uint8_t *dst, int m)
{
    if (n < m) return -1;
    if (n == m) { memcpy(dst, src, (size_t)n * n); return 0; }
    const double scale = (double)n / m;
    area_tab_t *tab = (area_tab_t *)malloc(sizeof(area_tab_t) * (size_t)(n + 2 * m));
    const int nt = area_tab(n, m, scale, tab);      /* the same table serves x and y (square source and destination) */
    float *buf = (float *)malloc(sizeof(float) * (size_t)m), *acc = (float *)malloc(sizeof(float) * (size_t)m);
    int prev_dy = tab[0].di;
    for (int dx = 0; dx < m; ++dx) acc[dx] = 0;
    for (int j = 0; j < nt; ++j) {
        const float beta = tab[j].alpha;
        const int dy = tab[j].di;
        const uint8_t *S = src + (size_t)tab[j].si * n;
        for (int dx = 0; dx < m; ++dx) buf[dx] = 0;
        for (int k = 0; k < nt; ++k) buf[tab[k].di] += S[tab[k].si] * tab[k].alpha;
        if (dy != prev_dy) {
            for (int dx = 0; dx < m; ++dx) {
                int v = cv_round(acc[dx]);
                dst[(size_t)prev_dy * m + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
                acc[dx] = beta * buf[dx];
            }
            prev_dy = dy;
        } else {
            for (int dx = 0; dx < m; ++dx) acc[dx] += beta * buf[dx];
        }
    }
    for (int dx = 0; dx < m; ++dx) {
        int v = cv_round(acc[dx]);
        dst[(size_t)prev_dy * m + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
    free(tab); free(buf); free(acc);
    return 0;
}

/* SURFInvoker (surf.cpp:568-866) over n keypoints {x, y, size, angle, response, octave, class_id}: writes angle, and the
 * descriptors (n x 64 | 128) when desc != NULL; keypoints it cannot orient get size = -1 (detectAndCompute erases those, :984-1012;
 * the caller compacts).  Returns 0, or -2 when a window would have to be enlarged (size < 7.5, not restated). */
int orc_surfcpu_compute(const uint8_t *img, int rows, int cols, float *kp, int n, int extended, int upright, float *desc)
{
    if (!img || !kp || n < 0) return -1;
    const int sw = cols + 1, dsize = extended ? 128 : 64;
    int *sum = (int *)malloc(sizeof(int) * (size_t)(rows + 1) * sw);
    integral_u8(img, rows, cols, sum);
    float g13[13], g20[PATCH_SZ], aptw[169], DW[PATCH_SZ * PATCH_SZ];
    int aptx[169], apty[169], n_ori = 0;
    gauss_kernel(13, 2.5, g13);
    for (int i = -ORI_RADIUS; i <= ORI_RADIUS; ++i)
        for (int j = -ORI_RADIUS; j <= ORI_RADIUS; ++j)
            if (i * i + j * j <= ORI_RADIUS * ORI_RADIUS) { aptx[n_ori] = i; apty[n_ori] = j; aptw[n_ori++] = g13[i + ORI_RADIUS] * g13[j + ORI_RADIUS]; }
    gauss_kernel(PATCH_SZ, (double)3.3f, g20);   /* SURF_DESC_SIGMA is a FLOAT constant (surf.cpp:121) promoted to double: 3.2999999523.  Found by the
                                                  * verbatim build of the class (libref_surfcpu.so): 3.3 gave descriptors 1 ulp off in ~20 % of the entries */
    for (int i = 0; i < PATCH_SZ; ++i) for (int j = 0; j < PATCH_SZ; ++j) DW[i * PATCH_SZ + j] = g20[i] * g20[j];
    static const int dx_s[2][5] = {{0, 0, 2, 4, -1}, {2, 0, 4, 4, 1}}, dy_s[2][5] = {{0, 0, 4, 2, 1}, {0, 2, 4, 4, -1}};
    int rc = 0;
    for (int k = 0; k < n && rc == 0; ++k) {
        float *K = kp + (size_t)k * 7;
        const float size = K[2], cx = K[0], cy = K[1];
        const float s = size * 1.2f / 9.0f;
        const int gw = 2 * cv_round(2 * s);
        if (rows + 1 < gw || cols + 1 < gw) { K[2] = -1; continue; }
        float dir = 360.f - 90.f;
        if (!upright) {
            hf_t dxt[2], dyt[2];
            float X[169], Y[169], ang[169];
            int na = 0;
            resize_haar(dx_s, dxt, 2, 4, gw, sw);
            resize_haar(dy_s, dyt, 2, 4, gw, sw);
            for (int kk = 0; kk < n_ori; ++kk) {
                const int x = cv_round(cx + aptx[kk] * s - (float)(gw - 1) / 2), y = cv_round(cy + apty[kk] * s - (float)(gw - 1) / 2);
                if (y < 0 || y >= rows + 1 - gw || x < 0 || x >= cols + 1 - gw) continue;
                const int *p = sum + (size_t)y * sw + x;
                const float vx = haar(p, dxt, 2), vy = haar(p, dyt, 2);
                X[na] = vx * aptw[kk]; Y[na] = vy * aptw[kk];
                ++na;
            }
            if (na == 0) { K[2] = -1; continue; }
            for (int j = 0; j < na; ++j) ang[j] = fast_atan2(Y[j], X[j]);      /* phase(X, Y, angle, true) */
            float bestx = 0, besty = 0, best = 0;
            for (int i = 0; i < 360; i += ORI_SEARCH_INC) {
                float sx = 0, sy = 0;
                for (int j = 0; j < na; ++j) {
                    const int d = abs(cv_round(ang[j]) - i);
                    if (d < ORI_WIN / 2 || d > 360 - ORI_WIN / 2) { sx += X[j]; sy += Y[j]; }
                }
                const float mod = sx * sx + sy * sy;
                if (mod > best) { best = mod; bestx = sx; besty = sy; }
            }
            dir = fast_atan2(-besty, bestx);
        }
        K[3] = dir;
        if (!desc) continue;
        const int win = (int)((PATCH_SZ + 1) * s);
        if (win < PATCH_SZ + 1) { rc = -2; break; }
        uint8_t *WIN = (uint8_t *)malloc((size_t)win * win);
        if (!upright) {
            const float rad = dir * (float)(3.14159265358979323846 / 180);
            const float sin_dir = -sinf(rad), cos_dir = cosf(rad);
            const float off = -(float)(win - 1) / 2;
            float start_x = cx + off * cos_dir + off * sin_dir, start_y = cy - off * sin_dir + off * cos_dir;
            const int nc1 = cols - 1, nr1 = rows - 1;
            for (int i = 0; i < win; ++i, start_x += sin_dir, start_y += cos_dir) {
                double px = start_x, py = start_y;
                for (int j = 0; j < win; ++j, px += cos_dir, py -= sin_dir) {
                    const int ix = cv_floor(px), iy = cv_floor(py);
                    if ((unsigned)ix < (unsigned)nc1 && (unsigned)iy < (unsigned)nr1) {
                        const float a = (float)(px - ix), b = (float)(py - iy);
                        const uint8_t *p = img + (size_t)iy * cols + ix;
                        WIN[i * win + j] = (uint8_t)cv_round(p[0] * (1.f - a) * (1.f - b) + p[1] * a * (1.f - b) + p[cols] * (1.f - a) * b +
                                                             p[cols + 1] * a * b);
                    } else {
                        int x = cv_round(px), y = cv_round(py);
                        x = x < 0 ? 0 : x > nc1 ? nc1 : x;
                        y = y < 0 ? 0 : y > nr1 ? nr1 : y;
                        WIN[i * win + j] = img[(size_t)y * cols + x];
                    }
                }
            }
        } else {
            const float off = -(float)(win - 1) / 2;
            int start_x = cv_round(cx + off);
            const int start_y = cv_round(cy - off);
            for (int i = 0; i < win; ++i, ++start_x) {
                int py = start_y;
                for (int j = 0; j < win; ++j, --py) {
                    int x = start_x < 0 ? 0 : start_x, y = py < 0 ? 0 : py;
                    x = x > cols - 1 ? cols - 1 : x;
                    y = y > rows - 1 ? rows - 1 : y;
                    WIN[i * win + j] = img[(size_t)y * cols + x];
                }
            }
        }
        uint8_t PATCH[(PATCH_SZ + 1) * (PATCH_SZ + 1)];
        resize_area_u8(WIN, win, PATCH, PATCH_SZ + 1);
        free(WIN);
        float DX[PATCH_SZ][PATCH_SZ], DY[PATCH_SZ][PATCH_SZ];
#define P(i, j) ((int)PATCH[(i) * (PATCH_SZ + 1) + (j)])
        for (int i = 0; i < PATCH_SZ; ++i)
            for (int j = 0; j < PATCH_SZ; ++j) {
                const float dw = DW[i * PATCH_SZ + j];
                DX[i][j] = (P(i, j + 1) - P(i, j) + P(i + 1, j + 1) - P(i + 1, j)) * dw;
                DY[i][j] = (P(i + 1, j) - P(i, j) + P(i + 1, j + 1) - P(i, j + 1)) * dw;
            }
#undef P
        float *vec = desc + (size_t)k * dsize;
        for (int kk = 0; kk < dsize; ++kk) vec[kk] = 0;
        double sq = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                for (int y = i * 5; y < i * 5 + 5; ++y)
                    for (int x = j * 5; x < j * 5 + 5; ++x) {
                        const float tx = DX[y][x], ty = DY[y][x];
                        if (extended) {
                            if (ty >= 0) { vec[0] += tx; vec[1] += fabsf(tx); } else { vec[2] += tx; vec[3] += fabsf(tx); }
                            if (tx >= 0) { vec[4] += ty; vec[5] += fabsf(ty); } else { vec[6] += ty; vec[7] += fabsf(ty); }
                        } else {
                            vec[0] += tx; vec[1] += ty; vec[2] += fabsf(tx); vec[3] += fabsf(ty);
                        }
                    }
                const int nb = extended ? 8 : 4;
                for (int kk = 0; kk < nb; ++kk) sq += vec[kk] * vec[kk];
                vec += nb;
            }
        vec = desc + (size_t)k * dsize;
        const float scale = (float)(1. / (sqrt(sq) + FLT_EPSILON));
        for (int kk = 0; kk < dsize; ++kk) vec[kk] *= scale;
    }
    free(sum);
    return rc;
}

/* ---- the restated main-repo functions, exported for oracle/refshim/cvsurf (libref_surfcpu.so: the reference's own surf.cpp compiled
 * verbatim calls cv::integral / resize / getGaussianKernel / phase / cvRound, which are not under /root/reference) ---- */
float orc_cv_fast_atan2(float y, float x) { return fast_atan2(y, x); }
void orc_cv_gauss_kernel(int n, double sigma, float *k) { gauss_kernel(n, sigma, k); }
int orc_cv_round(double v) { return cv_round(v); }
int orc_cv_resize_area_u8(const uint8_t *src, int n, uint8_t *dst, int m) { return resize_area_u8(src, n, dst, m); }
void orc_cv_integral_u8(const uint8_t *img, long long step, int rows, int cols, int *sum)
{
    if (step == cols) { integral_u8(img, rows, cols, sum); return; }
    uint8_t *dense = (uint8_t *)malloc((size_t)rows * cols);
    for (int y = 0; y < rows; ++y) memcpy(dense + (size_t)y * cols, img + (size_t)y * step, (size_t)cols);
    integral_u8(dense, rows, cols, sum);
    free(dense);
}
