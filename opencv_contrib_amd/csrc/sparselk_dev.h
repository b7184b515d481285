// cv::cuda::SparsePyrLKOpticalFlow on CV_8UC1 frames: one wave (64 lanes) per tracked point and pyramid level
// (modules/cudaoptflow/src/cuda/pyrlk.cu:148-340, sparseKernel).  Written as barrier-separated PHASES over a shared struct, like
// surfcpu_dev.h: the same source builds for the host (tests/cpp/sparselk_emul.cpp), where tests hold it bit for bit to
// oracle/pyrlk_ref.c.  Window element e = k * 64 + lane lives in slot [k][lane]; a lane adds its elements in ascending order and the
// 64 partial sums fold by the tree 32, 16, ..., 1 -- the summation order the oracle defines for this class.
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
namespace slk {

constexpr int T = 64;
constexpr int MAX_K = 16;          // window elements per lane: windows of up to 32 x 32 (the reference allows up to 80 x 80)

struct Image { const unsigned char *p; long long step; int rows, cols; };

struct Shared {
    float Ip[MAX_K][T], dx[MAX_K][T], dy[MAX_K][T];
    float r1[T], r2[T], r3[T];
    float nx, ny;
    int done;
};

MI_HD inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// a texel of the normalised-float texture: value / 255, clamp addressing
MI_HD inline float texel(const Image &I, int y, int x)
{
    return (float)I.p[(long long)clampi(y, 0, I.rows - 1) * I.step + clampi(x, 0, I.cols - 1)] / 255.0f;
}

MI_HD inline float tex_linear(const Image &I, int y0, int x0, float fy, float fx)
{
    const float t00 = texel(I, y0, x0), t01 = texel(I, y0, x0 + 1), t10 = texel(I, y0 + 1, x0), t11 = texel(I, y0 + 1, x0 + 1);
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const float top = t00 * gx + t01 * fx, bot = t10 * gx + t11 * fx;
    return top * gy + bot * fy;
}

// r1 (and r2, r3 when n3) hold one partial sum per lane on entry, the totals in slot 0 on exit
MI_HD inline void fold(Shared &sm, bool three)
{
    for (int s = T / 2; s >= 1; s >>= 1) {
        MI_FOR_TID(T) {
            if (tid < s) {
                sm.r1[tid] += sm.r1[tid + s];
                sm.r2[tid] += sm.r2[tid + s];
                if (three) sm.r3[tid] += sm.r3[tid + s];
            }
        }
        MI_BARRIER();
    }
}

// One point at one level.  next_pt: this point's entry of nextPts (read x 2, written only when the kernel runs to its end);
// status: cleared on the early exits at level 0 only (pyrlk.cu:165-171, 236-242, 259-265); err: level 0 only, may be NULL.
MI_HD inline void point_block(const Image &I, const Image &J, float px, float py, float *next_pt, int level, int wx, int wy, int iters,
                              unsigned char *status, float *err, Shared &sm)
{
    const int hx = (wx - 1) / 2, hy = (wy - 1) / 2, ne = wx * wy;
    px *= (1.0f / (1 << level));
    py *= (1.0f / (1 << level));
    if (px < 0 || px >= I.cols || py < 0 || py >= I.rows) {      // uniform
        MI_FOR_TID(T) { if (tid == 0 && level == 0) *status = 0; }
        return;
    }
    px -= hx;
    py -= hy;
    {
        const float x0f = floorf(px), y0f = floorf(py);
        const float fx = px - x0f, fy = py - y0f;
        const int x0 = (int)x0f, y0 = (int)y0f;
        MI_FOR_TID(T) {
            float a11 = 0, a12 = 0, a22 = 0;
            for (int k = 0, e = tid; e < ne; ++k, e += T) {
                const int i = e / wx, j = e % wx;
                // patch value and its Scharr derivatives (pyrlk.cu:196-209), sums left to right
                const float s00 = tex_linear(I, y0 + i - 1, x0 + j - 1, fy, fx), s01 = tex_linear(I, y0 + i - 1, x0 + j, fy, fx);
                const float s02 = tex_linear(I, y0 + i - 1, x0 + j + 1, fy, fx), s10 = tex_linear(I, y0 + i, x0 + j - 1, fy, fx);
                const float s11 = tex_linear(I, y0 + i, x0 + j, fy, fx), s12 = tex_linear(I, y0 + i, x0 + j + 1, fy, fx);
                const float s20 = tex_linear(I, y0 + i + 1, x0 + j - 1, fy, fx), s21 = tex_linear(I, y0 + i + 1, x0 + j, fy, fx);
                const float s22 = tex_linear(I, y0 + i + 1, x0 + j + 1, fy, fx);
                const float gx = 3.0f * s02 + 10.0f * s12 + 3.0f * s22 - (3.0f * s00 + 10.0f * s10 + 3.0f * s20);
                const float gy = 3.0f * s20 + 10.0f * s21 + 3.0f * s22 - (3.0f * s00 + 10.0f * s01 + 3.0f * s02);
                sm.Ip[k][tid] = s11; sm.dx[k][tid] = gx; sm.dy[k][tid] = gy;
                a11 += gx * gx; a12 += gx * gy; a22 += gy * gy;
            }
            sm.r1[tid] = a11; sm.r2[tid] = a12; sm.r3[tid] = a22;
        }
        MI_BARRIER();
    }
    fold(sm, true);
    float A11 = sm.r1[0], A12 = sm.r2[0], A22 = sm.r3[0];
    float D = A11 * A22 - A12 * A12;
    if (D < FLT_EPSILON) {                                       // uniform
        MI_FOR_TID(T) { if (tid == 0 && level == 0) *status = 0; }
        return;
    }
    D = 1.f / D;
    A11 *= D; A12 *= D; A22 *= D;
    MI_BARRIER();                                                // everyone has read slot 0 before it is reused
    MI_FOR_TID(T) {
        if (tid == 0) { sm.nx = next_pt[0] * 2.f - hx; sm.ny = next_pt[1] * 2.f - hy; sm.done = 0; }
    }
    MI_BARRIER();
    for (int it = 0; it < iters; ++it) {
        const float nx = sm.nx, ny = sm.ny;
        if (nx < -hx || nx >= I.cols || ny < -hy || ny >= I.rows) {      // uniform (pyrlk.cu:236-242)
            MI_FOR_TID(T) { if (tid == 0 && level == 0) *status = 0; }
            return;
        }
        const float x0f = floorf(nx), y0f = floorf(ny);
        const float fx = nx - x0f, fy = ny - y0f;
        const int x0 = (int)x0f, y0 = (int)y0f;
        MI_FOR_TID(T) {
            float b1 = 0, b2 = 0;
            for (int k = 0, e = tid; e < ne; ++k, e += T) {
                const float Jv = tex_linear(J, y0 + e / wx, x0 + e % wx, fy, fx);
                const float diff = (Jv - sm.Ip[k][tid]) * 32.0f;
                b1 += diff * sm.dx[k][tid];
                b2 += diff * sm.dy[k][tid];
            }
            sm.r1[tid] = b1; sm.r2[tid] = b2;
        }
        MI_BARRIER();
        fold(sm, false);
        MI_FOR_TID(T) {
            if (tid == 0) {
                const float b1 = sm.r1[0], b2 = sm.r2[0];
                const float ddx = A12 * b2 - A22 * b1, ddy = A12 * b1 - A11 * b2;
                sm.nx = nx + ddx;
                sm.ny = ny + ddy;
                sm.done = fabsf(ddx) < 0.01f && fabsf(ddy) < 0.01f;
            }
        }
        MI_BARRIER();
        if (sm.done) break;                                      // uniform
    }
    if (err) {
        const float nx = sm.nx, ny = sm.ny;
        const float x0f = floorf(nx), y0f = floorf(ny);
        const float fx = nx - x0f, fy = ny - y0f;
        const int x0 = (int)x0f, y0 = (int)y0f;
        MI_FOR_TID(T) {
            float q = 0;
            for (int k = 0, e = tid; e < ne; ++k, e += T) q += fabsf(tex_linear(J, y0 + e / wx, x0 + e % wx, fy, fx) - sm.Ip[k][tid]);
            sm.r1[tid] = q; sm.r2[tid] = 0;
        }
        MI_BARRIER();
        fold(sm, false);
    }
    MI_FOR_TID(T) {
        if (tid == 0) {
            next_pt[0] = sm.nx + hx;
            next_pt[1] = sm.ny + hy;
            if (err) *err = sm.r1[0] / (wx * wy) * 255.0f;       // / (min(cn, 3) * winSize) * DenormalizationFactor<uchar> (pyrlk.cu:338)
        }
    }
}

// cuda::pyrDown of an 8-bit image (cudawarping/src/cuda/pyr_down.cu:54-175): 5 x 5 binomial taps in float, vertical sums per column
// first, BORDER_REFLECT_101, saturate_cast<uchar>
MI_HD inline int reflect101(int i, int n)      // BrdReflect101: idx_low = |i| % n, idx_high = |last - |last - i|| % n
{
    const int last = n - 1;
    int a = last - i;
    a = a < 0 ? -a : a;
    int v = last - a;
    v = (v < 0 ? -v : v) % n;
    return v;
}

MI_HD inline unsigned char pyr_down_pixel(const Image &S, int y, int dx)
{
    const int sy = 2 * y;
    float v[5];
    for (int j = 0; j < 5; ++j) {
        const int x = reflect101(2 * dx + j - 2, S.cols);
        float sum;
        sum = 0.0625f * S.p[(long long)reflect101(sy - 2, S.rows) * S.step + x];
        sum = sum + 0.25f * S.p[(long long)reflect101(sy - 1, S.rows) * S.step + x];
        sum = sum + 0.375f * S.p[(long long)reflect101(sy, S.rows) * S.step + x];
        sum = sum + 0.25f * S.p[(long long)reflect101(sy + 1, S.rows) * S.step + x];
        sum = sum + 0.0625f * S.p[(long long)reflect101(sy + 2, S.rows) * S.step + x];
        v[j] = sum;
    }
    float sum;
    sum = 0.0625f * v[0];
    sum = sum + 0.25f * v[1];
    sum = sum + 0.375f * v[2];
    sum = sum + 0.25f * v[3];
    sum = sum + 0.0625f * v[4];
    const int r = (int)rintf(sum);
    return (unsigned char)(r < 0 ? 0 : r > 255 ? 255 : r);
}

}  // namespace slk
}  // namespace mi
