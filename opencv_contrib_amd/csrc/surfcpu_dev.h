// Orientation and descriptor of the reference's CPU SURF class (xfeatures2d::SURF_Impl, modules/xfeatures2d/src/surf.cpp:568-866)
// for given keypoints -- the arithmetic the reference's golden vectors were produced with (misc/java/test/SURF*Test.java).
//
// One workgroup per keypoint.  The code is written as PHASES separated by barriers, with every value that crosses a barrier held
// in the workgroup's shared struct: the same source compiles for the host (tests/cpp/surfcpu_emul.cpp: a phase becomes a loop over
// the thread index), where it is checked bit for bit against oracle/surfcpu_ref.c without a GPU.  Sequential dependences of the
// reference that decide rounding are kept sequential (one thread walks a window row: the double-precision position accumulates
// along it; one thread accumulates the per-row start positions; window sums and descriptor bins add in the reference's order).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MI_HD __host__ __device__
#else
#define MI_HD
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define MI_FOR_TID(T) for (int tid = (int)threadIdx.x, mi_once_ = 1; mi_once_; mi_once_ = 0)
#define MI_BARRIER() __syncthreads()
#else
#define MI_FOR_TID(T) for (int tid = 0; tid < (T); ++tid)
#define MI_BARRIER() ((void)0)
#endif

namespace mi {
namespace surfcpu {

constexpr int ORI_N = 113;        // grid points in the circle of radius 6 (surf.cpp:546-556)
constexpr int ORI_T = 64;         // threads per keypoint, orientation
constexpr int ORI_WINDOWS = 72;   // 360 / SURF_ORI_SEARCH_INC
constexpr int PATCH_SZ = 20, P1 = PATCH_SZ + 1;
constexpr int DESC_T = 256;       // threads per keypoint, descriptor
constexpr int MAX_WIN = 1024;     // window rows held per keypoint ((PATCH_SZ + 1) * s: 873 for the largest filter of 4 octaves x 4 layers)
constexpr int MAX_TAB = MAX_WIN + 2 * P1;

struct Tables {                   // generated on the host exactly as SURFInvoker's constructor does (surf.cpp:544-565)
    signed char ax[ORI_N], ay[ORI_N];
    float aw[ORI_N];
    float dw[PATCH_SZ * PATCH_SZ];
};

struct Image { const unsigned char *p; long long step; int rows, cols; };
struct Integral { const int *p; long long ld; };      // (rows + 1) x (cols + 1), ld in elements

struct Hf { int p0, p1, p2, p3; float w; };

MI_HD inline int cv_round(double v) { return (int)rint(v); }          // round half to even
MI_HD inline int cv_floor(double v) { return (int)floor(v); }
MI_HD inline int cv_ceil(double v) { return (int)ceil(v); }

// cv::fastAtan2 / cv::phase(..., angleInDegrees = true): 7th-order odd polynomial in the smaller ratio
MI_HD inline float fast_atan2(float y, float x)
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

// resizeHaarPattern, surf.cpp:145-162
MI_HD inline void resize_haar(const int (*src)[5], Hf *dst, int n, int old_size, int new_size, long long ld)
{
    const float ratio = (float)new_size / old_size;
    for (int k = 0; k < n; ++k) {
        const int dx1 = cv_round(ratio * src[k][0]), dy1 = cv_round(ratio * src[k][1]);
        const int dx2 = cv_round(ratio * src[k][2]), dy2 = cv_round(ratio * src[k][3]);
        dst[k].p0 = (int)(dy1 * ld + dx1);
        dst[k].p1 = (int)(dy2 * ld + dx1);
        dst[k].p2 = (int)(dy1 * ld + dx2);
        dst[k].p3 = (int)(dy2 * ld + dx2);
        dst[k].w = src[k][4] / ((float)(dx2 - dx1) * (dy2 - dy1));
    }
}

// calcHaarPattern, surf.cpp:137-143: float product of the box sum and its weight, accumulated in double
MI_HD inline float haar2(const int *o, const Hf *f)
{
    double d = 0;
    for (int k = 0; k < 2; ++k) d += (o[f[k].p0] + o[f[k].p3] - o[f[k].p1] - o[f[k].p2]) * f[k].w;
    return (float)d;
}

// ------------------------------------------------------------------------------------------------ orientation (surf.cpp:598-667)
struct OriShared {
    float X[ORI_N], Y[ORI_N];      // weighted responses of the samples inside the image, compacted in sample order
    int A[ORI_N];                  // cvRound of their angle
    float rx[ORI_N], ry[ORI_N];    // per sample, before compaction
    int ra[ORI_N], ok[ORI_N];
    int n;
    float mod[ORI_WINDOWS], sx[ORI_WINDOWS], sy[ORI_WINDOWS];
};

// Returns through *angle (degrees) and *size (-1: the reference erases the keypoint, surf.cpp:604-611,631-638).
MI_HD inline void orientation_block(const Integral &S, int rows, int cols, float cx, float cy, float size, int upright, const Tables &T,
                                    OriShared &sm, float *angle, float *size_out)
{
    const float s = size * 1.2f / 9.0f;
    const int gw = 2 * cv_round(2 * s);
    if (rows + 1 < gw || cols + 1 < gw) {      // uniform for the workgroup
        MI_FOR_TID(ORI_T) { if (tid == 0) *size_out = -1.f; }
        return;
    }
    if (upright) {
        MI_FOR_TID(ORI_T) { if (tid == 0) *angle = 360.f - 90.f; }
        return;
    }
    MI_FOR_TID(ORI_T) {
        const int dx_s[2][5] = {{0, 0, 2, 4, -1}, {2, 0, 4, 4, 1}}, dy_s[2][5] = {{0, 0, 4, 2, 1}, {0, 2, 4, 4, -1}};
        Hf dxt[2], dyt[2];
        resize_haar(dx_s, dxt, 2, 4, gw, S.ld);
        resize_haar(dy_s, dyt, 2, 4, gw, S.ld);
        for (int kk = tid; kk < ORI_N; kk += ORI_T) {
            const int x = cv_round(cx + T.ax[kk] * s - (float)(gw - 1) / 2), y = cv_round(cy + T.ay[kk] * s - (float)(gw - 1) / 2);
            const bool in = !(y < 0 || y >= rows + 1 - gw || x < 0 || x >= cols + 1 - gw);
            sm.ok[kk] = in;
            if (in) {
                const int *p = S.p + (long long)y * S.ld + x;
                const float vx = haar2(p, dxt), vy = haar2(p, dyt);
                sm.rx[kk] = vx * T.aw[kk];
                sm.ry[kk] = vy * T.aw[kk];
                sm.ra[kk] = cv_round(fast_atan2(sm.ry[kk], sm.rx[kk]));       // phase(X, Y, angle, true), rounded at :647
            }
        }
    }
    MI_BARRIER();
    MI_FOR_TID(ORI_T) {
        if (tid == 0) {
            int n = 0;
            for (int kk = 0; kk < ORI_N; ++kk)
                if (sm.ok[kk]) { sm.X[n] = sm.rx[kk]; sm.Y[n] = sm.ry[kk]; sm.A[n] = sm.ra[kk]; ++n; }
            sm.n = n;
        }
    }
    MI_BARRIER();
    if (sm.n == 0) {                            // uniform
        MI_FOR_TID(ORI_T) { if (tid == 0) *size_out = -1.f; }
        return;
    }
    MI_FOR_TID(ORI_T) {
        for (int w = tid; w < ORI_WINDOWS; w += ORI_T) {
            const int i = w * 5;                // SURF_ORI_SEARCH_INC
            float sumx = 0, sumy = 0;
            for (int j = 0; j < sm.n; ++j) {
                const int a = sm.A[j] - i, d = a < 0 ? -a : a;
                if (d < 30 || d > 330) { sumx += sm.X[j]; sumy += sm.Y[j]; }      // ORI_WIN / 2, 360 - ORI_WIN / 2
            }
            sm.sx[w] = sumx; sm.sy[w] = sumy; sm.mod[w] = sumx * sumx + sumy * sumy;
        }
    }
    MI_BARRIER();
    MI_FOR_TID(ORI_T) {
        if (tid == 0) {
            float bestx = 0, besty = 0, best = 0;
            for (int w = 0; w < ORI_WINDOWS; ++w)
                if (sm.mod[w] > best) { best = sm.mod[w]; bestx = sm.sx[w]; besty = sm.sy[w]; }
            *angle = fast_atan2(-besty, bestx);
        }
    }
}

// ------------------------------------------------------------------------------------------------ descriptor (surf.cpp:669-849)
struct AreaTab { short si, di; float alpha; };

struct DescShared {
    float start_x[MAX_WIN], start_y[MAX_WIN];     // rotated: float positions of the window rows; upright: integer column / row start
    AreaTab tab[MAX_TAB];
    int nt, win, bad;
    float sin_dir, cos_dir;
    float hs[DESC_T][P1];                         // horizontal pass of a chunk of window rows
    float acc[P1][P1];
    int cursor[P1], prev_dy[P1];
    unsigned char patch[P1][P1];
    float DX[PATCH_SZ][PATCH_SZ], DY[PATCH_SZ][PATCH_SZ];
    float vec[128];
    float scale;
};

// imgproc resize.cpp computeResizeAreaTab for ssize -> P1 cells (scale = ssize / P1 > 1)
MI_HD inline int area_tab(int ssize, double scale, AreaTab *tab)
{
    int k = 0;
    for (int dx = 0; dx < P1; ++dx) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = fmin(scale, ssize - fsx1);
        int sx1 = cv_ceil(fsx1), sx2 = cv_floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) { tab[k].di = (short)dx; tab[k].si = (short)(sx1 - 1); tab[k++].alpha = (float)((sx1 - fsx1) / cell); }
        for (int sx = sx1; sx < sx2; ++sx) { tab[k].di = (short)dx; tab[k].si = (short)sx; tab[k++].alpha = (float)(1.0 / cell); }
        if (fsx2 - sx2 > 1e-3) { tab[k].di = (short)dx; tab[k].si = (short)sx2; tab[k++].alpha = (float)(fmin(fmin(fsx2 - sx2, 1.), cell) / cell); }
    }
    return k;
}

// one pixel of the rotated window (surf.cpp:707-733): bilinear inside, clamped nearest at the border
MI_HD inline unsigned char window_pixel(const Image &I, double px, double py)
{
    const int nc1 = I.cols - 1, nr1 = I.rows - 1;
    const int ix = cv_floor(px), iy = cv_floor(py);
    if ((unsigned)ix < (unsigned)nc1 && (unsigned)iy < (unsigned)nr1) {
        const float a = (float)(px - ix), b = (float)(py - iy);
        const unsigned char *p = I.p + (long long)iy * I.step + ix;
        return (unsigned char)cv_round(p[0] * (1.f - a) * (1.f - b) + p[1] * a * (1.f - b) + p[I.step] * (1.f - a) * b + p[I.step + 1] * a * b);
    }
    int x = cv_round(px), y = cv_round(py);
    x = x < 0 ? 0 : x > nc1 ? nc1 : x;
    y = y < 0 ? 0 : y > nr1 ? nr1 : y;
    return I.p[(long long)y * I.step + x];
}

MI_HD inline unsigned char saturate_u8(float v)
{
    const int r = cv_round(v);
    return (unsigned char)(r < 0 ? 0 : r > 255 ? 255 : r);
}

// desc: 64 or 128 floats.  Keypoints whose window would have to be enlarged (s < 1) or exceeds MAX_WIN get a zero descriptor and
// sm.bad = 1 (the reference's resize would switch algorithms for s < 1; neither case occurs for filters of 9 ... 312 px).
MI_HD inline void descriptor_block(const Image &I, float cx, float cy, float size, float dir, int upright, int extended, const Tables &T,
                                   DescShared &sm, float *desc)
{
    const int dsize = extended ? 128 : 64;
    MI_FOR_TID(DESC_T) {
        if (tid == 0) {
            const float s = size * 1.2f / 9.0f;
            const int win = (int)((PATCH_SZ + 1) * s);
            sm.win = win;
            sm.bad = (win < P1 || win > MAX_WIN);
            if (!sm.bad) {
                const float off = -(float)(win - 1) / 2;
                if (!upright) {
                    const float rad = dir * (float)(3.14159265358979323846 / 180);
                    const float sin_dir = -sinf(rad), cos_dir = cosf(rad);
                    float sx = cx + off * cos_dir + off * sin_dir, sy = cy - off * sin_dir + off * cos_dir;
                    for (int i = 0; i < win; ++i, sx += sin_dir, sy += cos_dir) { sm.start_x[i] = sx; sm.start_y[i] = sy; }
                    sm.sin_dir = sin_dir; sm.cos_dir = cos_dir;
                } else {
                    sm.start_x[0] = (float)cv_round(cx + off);      // window row i reads image column start_x + i ...
                    sm.start_y[0] = (float)cv_round(cy - off);      // ... and window column j reads image row start_y - j (surf.cpp:743-760)
                }
                sm.nt = win == P1 ? 0 : area_tab(win, (double)win / P1, sm.tab);
            }
        }
        if (tid < P1) { sm.cursor[tid] = 0; sm.prev_dy[tid] = 0; for (int r = 0; r < P1; ++r) sm.acc[r][tid] = 0; }
    }
    MI_BARRIER();
    if (sm.bad) {
        MI_FOR_TID(DESC_T) { if (tid < dsize) desc[tid] = 0; }
        return;
    }
    const int win = sm.win;
    if (win == P1) {                                      // resize to the same size: a copy
        MI_FOR_TID(DESC_T) {
            for (int e = tid; e < P1 * P1; e += DESC_T) {
                const int i = e / P1, j = e % P1;
                unsigned char v;
                if (!upright) {
                    double px = sm.start_x[i], py = sm.start_y[i];
                    for (int q = 0; q < j; ++q) { px += sm.cos_dir; py -= sm.sin_dir; }
                    v = window_pixel(I, px, py);
                } else {
                    int x = (int)sm.start_x[0] + i, y = (int)sm.start_y[0] - j;
                    x = x < 0 ? 0 : x > I.cols - 1 ? I.cols - 1 : x;
                    y = y < 0 ? 0 : y > I.rows - 1 ? I.rows - 1 : y;
                    v = I.p[(long long)y * I.step + x];
                }
                sm.patch[i][j] = v;
            }
        }
        MI_BARRIER();
    } else {
        // resize(win, patch, INTER_AREA): horizontal pass per window row (table order), vertical pass in row order (resize.cpp ResizeArea_)
        for (int base = 0; base < win; base += DESC_T) {
            MI_FOR_TID(DESC_T) {
                const int i = base + tid;                 // window row
                if (i < win) {
                    float *h = sm.hs[tid];
                    for (int dx = 0; dx < P1; ++dx) h[dx] = 0;
                    int j = 0;
                    double px = 0, py = 0;
                    int ux = 0, uy = 0;
                    unsigned char cur;
                    if (!upright) { px = sm.start_x[i]; py = sm.start_y[i]; cur = window_pixel(I, px, py); }
                    else {
                        ux = (int)sm.start_x[0] + i; uy = (int)sm.start_y[0];
                        ux = ux < 0 ? 0 : ux > I.cols - 1 ? I.cols - 1 : ux;
                        const int y0 = uy < 0 ? 0 : uy > I.rows - 1 ? I.rows - 1 : uy;
                        cur = I.p[(long long)y0 * I.step + ux];
                    }
                    for (int k = 0; k < sm.nt; ++k) {
                        while (j < sm.tab[k].si) {        // advance along the window row: positions accumulate in double (surf.cpp:711-713)
                            ++j;
                            if (!upright) { px += sm.cos_dir; py -= sm.sin_dir; cur = window_pixel(I, px, py); }
                            else {
                                int y = uy - j;
                                y = y < 0 ? 0 : y > I.rows - 1 ? I.rows - 1 : y;
                                cur = I.p[(long long)y * I.step + ux];
                            }
                        }
                        h[sm.tab[k].di] += cur * sm.tab[k].alpha;
                    }
                }
            }
            MI_BARRIER();
            MI_FOR_TID(DESC_T) {
                if (tid < P1) {                           // one thread per destination column walks the table's rows in order
                    const int dx = tid, end = base + DESC_T;
                    int k = sm.cursor[dx], prev = sm.prev_dy[dx];
                    while (k < sm.nt && sm.tab[k].si < end) {
                        const float beta = sm.tab[k].alpha, b = sm.hs[sm.tab[k].si - base][dx];
                        const int dy = sm.tab[k].di;
                        if (dy != prev) {
                            sm.patch[prev][dx] = saturate_u8(sm.acc[prev][dx]);
                            sm.acc[dy][dx] = beta * b;
                            prev = dy;
                        } else {
                            sm.acc[dy][dx] += beta * b;
                        }
                        ++k;
                    }
                    sm.cursor[dx] = k; sm.prev_dy[dx] = prev;
                }
            }
            MI_BARRIER();
        }
        MI_FOR_TID(DESC_T) { if (tid < P1) sm.patch[sm.prev_dy[tid]][tid] = saturate_u8(sm.acc[sm.prev_dy[tid]][tid]); }
        MI_BARRIER();
    }
    // gradients with wavelets of size 2s (surf.cpp:771-780)
    MI_FOR_TID(DESC_T) {
        for (int e = tid; e < PATCH_SZ * PATCH_SZ; e += DESC_T) {
            const int i = e / PATCH_SZ, j = e % PATCH_SZ;
            const float dw = T.dw[e];
            const int p00 = sm.patch[i][j], p01 = sm.patch[i][j + 1], p10 = sm.patch[i + 1][j], p11 = sm.patch[i + 1][j + 1];
            sm.DX[i][j] = (p01 - p00 + p11 - p10) * dw;
            sm.DY[i][j] = (p10 - p00 + p11 - p01) * dw;
        }
    }
    MI_BARRIER();
    // 4 x 4 sub-regions of 5 x 5 samples, bins in the reference's order (surf.cpp:783-842)
    MI_FOR_TID(DESC_T) {
        if (tid < 16) {
            const int i = tid / 4, j = tid % 4, nb = extended ? 8 : 4;
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int y = i * 5; y < i * 5 + 5; ++y)
                for (int x = j * 5; x < j * 5 + 5; ++x) {
                    const float tx = sm.DX[y][x], ty = sm.DY[y][x];
                    if (extended) {
                        if (ty >= 0) { v[0] += tx; v[1] += fabsf(tx); } else { v[2] += tx; v[3] += fabsf(tx); }
                        if (tx >= 0) { v[4] += ty; v[5] += fabsf(ty); } else { v[6] += ty; v[7] += fabsf(ty); }
                    } else {
                        v[0] += tx; v[1] += ty; v[2] += fabsf(tx); v[3] += fabsf(ty);
                    }
                }
            for (int kk = 0; kk < nb; ++kk) sm.vec[tid * nb + kk] = v[kk];
        }
    }
    MI_BARRIER();
    MI_FOR_TID(DESC_T) {
        if (tid == 0) {
            double sq = 0;
            for (int kk = 0; kk < dsize; ++kk) sq += sm.vec[kk] * sm.vec[kk];
            sm.scale = (float)(1. / (sqrt(sq) + FLT_EPSILON));      // unit vector (surf.cpp:845-848)
        }
    }
    MI_BARRIER();
    MI_FOR_TID(DESC_T) { if (tid < dsize) desc[tid] = sm.vec[tid] * sm.scale; }
}

// the two tables, as the constructor of SURFInvoker fills them (surf.cpp:544-565); host only
inline void make_tables(Tables &T)
{
    auto gauss = [](int n, double sigma, float *k) {      // getGaussianKernel(n, sigma, CV_32F)
        const double scale2 = -0.5 / (sigma * sigma);
        double w[64], sum = 0;
        for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; w[i] = exp(scale2 * x * x); sum += w[i]; }
        sum = 1. / sum;   // the main repo multiplies by the reciprocal (getGaussianKernelBitExact)
        for (int i = 0; i < n; ++i) k[i] = (float)(w[i] * sum);
    };
    float g13[13], g20[PATCH_SZ];
    gauss(13, 2.5, g13);
    int n = 0;
    for (int i = -6; i <= 6; ++i)
        for (int j = -6; j <= 6; ++j)
            if (i * i + j * j <= 36) { T.ax[n] = (signed char)i; T.ay[n] = (signed char)j; T.aw[n++] = g13[i + 6] * g13[j + 6]; }
    gauss(PATCH_SZ, (double)3.3f, g20);   // SURF_DESC_SIGMA is a float constant (surf.cpp:121) promoted to double: 3.2999999523 (round 4: found by the
                                          // verbatim build of the reference class, oracle/_ref/libref_surfcpu.so -- 3.3 was 1 ulp off in ~20 % of the entries)
    for (int i = 0; i < PATCH_SZ; ++i) for (int j = 0; j < PATCH_SZ; ++j) T.dw[i * PATCH_SZ + j] = g20[i] * g20[j];
}

}  // namespace surfcpu
}  // namespace mi
