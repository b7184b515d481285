// Farneback dense optical flow (cv::cuda::FarnebackOpticalFlow) -- HIP kernels for gfx950, wave64.
//
// Reference: modules/cudaoptflow/src/cuda/farneback.cu:66-651 driven by src/farneback.cpp:314-482.
// Arithmetic (operation order, separately rounded binary32 ops, -ffp-contract=off) follows those kernels
// so the HIP path can be compared tightly with the oracle (oracle/farneback_ref.c).
// What is different for MI355X:
//   * parameters (Gaussian taps, inverse moment matrix entries, border table) travel as KERNEL ARGUMENTS:
//     the reference keeps them in __constant__ memory written per call (farneback.cu:60-63,154,453), so two
//     concurrent calc()s race; here handles/streams are independent.
//   * one inner iteration = ONE kernel: 5-plane box/Gaussian blur (vertical pass into LDS, horizontal from
//     LDS) + 2x2 flow solve + matrix update with the bilinear warp of R1, instead of the reference's
//     boxFilter5 -> updateFlow -> updateMatrices (3 launches, M written and re-read twice): 40 + 28 + 68 B/px
//     of the reference's traffic become 20 (M) + 20 (R0) + gathered R1 + 8 (flow) + 20 (M') B/px.
//   * no 5-stream fan-out + host waitForCompletion per level (farneback.cpp:319-324,366,456): everything is
//     ordered on the caller's stream.
// Planes are dense f32, `ld` floats per row (multiple of 64); 5-plane buffers are stacked vertically
// (plane k at rows [k*h, (k+1)*h)) exactly like the reference's 5H x W matrices.
#include "farneback_dev.h"
#include "resize_dev.h"
#include <cfloat>

namespace mi {
namespace fb {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
// BrdReflect101 (main repo core/cuda/border_interpolate.hpp)
__device__ __forceinline__ int reflect101(int i, int n)
{
    const int last = n - 1;
    return abs(abs(last - abs(last - i)) % n) % n;
}
template <int BORDER>
__device__ __forceinline__ int bidx(int i, int n) { return BORDER == MI_BORDER_REFLECT101 ? reflect101(i, n) : clampi(i, 0, n - 1); }

// ------------------------------------------------------------------ convert / flow split+merge
__global__ __launch_bounds__(256) void k_convert(const void *a, long long sa, const void *b, long long sb, int type, float *A,
                                                 float *B, int w, int h, int ld)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const long long o = (long long)y * ld + x;
    if (type == MI_8UC1) {   // convertTo(CV_32F), farneback.cpp:342-345 (no scaling)
        A[o] = (float)((const unsigned char *)a)[(long long)y * sa + x];
        B[o] = (float)((const unsigned char *)b)[(long long)y * sb + x];
    } else {
        A[o] = ((const float *)((const char *)a + (long long)y * sa))[x];
        B[o] = ((const float *)((const char *)b + (long long)y * sb))[x];
    }
}

__global__ __launch_bounds__(256) void k_split_flow(const void *flow, long long sf, float *fx, float *fy, int w, int h, int ld)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float2 f = ((const float2 *)((const char *)flow + (long long)y * sf))[x];   // cuda::split, farneback.cpp:184-187
    fx[(long long)y * ld + x] = f.x;
    fy[(long long)y * ld + x] = f.y;
}

__global__ __launch_bounds__(256) void k_merge_flow(const float *fx, const float *fy, void *flow, long long sf, int w, int h, int ld)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const long long o = (long long)y * ld + x;
    ((float2 *)((char *)flow + (long long)y * sf))[x] = make_float2(fx[o], fy[o]);   // cuda::merge, farneback.cpp:197-198
}

// ------------------------------------------------------------------ single-plane Gaussian blur
// farneback.cu:455-492.  One block = one row segment of 256 columns.
// FAST: the taps reach less than one image size past the border (kh < w, kh < h), so BrdReflect101's double modulo -- a 30-instruction
// integer division each, 2 (2 kh + 1) of them per output -- reduces to one mirror step with the same result.
template <int BORDER, bool FAST>
__device__ __forceinline__ int gidx(int i, int n)
{
    if (!FAST) return bidx<BORDER>(i, n);
    if (BORDER == MI_BORDER_REFLECT101) {
        i = abs(i);
        return i >= n ? 2 * n - 2 - i : i;
    }
    return clampi(i, 0, n - 1);
}
template <int BORDER, bool FAST>
__global__ __launch_bounds__(256) void k_gaussian_blur(const float *src, float *dst, int w, int h, int ld, int kh, Taps K, long long bs, int nf,
                                                       long long fs)
{
    extern __shared__ float row[];
    // blockIdx.z = pair * nf + frame: nf = 2 runs both frames of every pair in one launch (frame f of a pair lies fs floats behind frame 0)
    { const long long o = (long long)(blockIdx.z / nf) * bs + (long long)(blockIdx.z % nf) * fs; src += o; dst += o; }
    const int tx = threadIdx.x, y = blockIdx.y, x = blockIdx.x * 256 + tx;
    for (int i = tx; i < 256 + 2 * kh; i += 256) {
        // columns past w + kh feed no output (x < w below) but are loaded: keep them inside the range one mirror step covers
        const int xe = gidx<BORDER, FAST>(min((int)(blockIdx.x * 256) + i - kh, w - 1 + kh), w);
        float v = src[(long long)y * ld + xe] * K.k[0];
        // unrolled so that the loads of four tap pairs are in flight together (the sum itself stays in the reference's order)
#pragma unroll 4
        for (int j = 1; j <= kh; ++j)
            v += (src[(long long)gidx<BORDER, FAST>(y - j, h) * ld + xe] + src[(long long)gidx<BORDER, FAST>(y + j, h) * ld + xe]) * K.k[j];
        row[i] = v;
    }
    __syncthreads();
    if (x < w) {
        const float *r = row + tx + kh;
        float res = r[0] * K.k[0];
#pragma unroll 4
        for (int i = 1; i <= kh; ++i) res += (r[-i] + r[i]) * K.k[i];
        dst[(long long)y * ld + x] = res;
    }
}

// The same blur, tiled (round 5), for compile-time half sizes: a workgroup owns RB rows x 256 columns.  The one-row kernel above
// fetches the 2 kh + 1 rows of a column per OUTPUT row, a chain of dependent loads for 256 outputs per workgroup: 147 us per 64 planes
// of 640 x 480 whatever the kernel size, against ~35 us of plane traffic.  Here a thread loads the RB + 2 KH rows of its column ONCE,
// all in flight together (RB in 8..14 so that RB + 2 KH is a multiple of eight), forms the RB vertical sums from registers and
// leaves them in the LDS for the horizontal pass; the 2 KH columns beside the tile are a second, short round of wave 0.  Rows and
// columns go through the border rule when they are loaded and the sums are formed in the same order: bit-identical planes.
template <int KH> struct BlurTile { static constexpr int RB = (2 * KH) % 8 ? 16 - (2 * KH) % 8 : 8; };
// (the tile's arithmetic; `at(row, column)` reads a source pixel as float)
template <int BORDER, int KH, class At>
__device__ __forceinline__ void blur_tile(At at, float *dst, int w, int h, int ld, const Taps &K)
{
    constexpr int RB = BlurTile<KH>::RB, W2 = 256 + 2 * KH, NL = RB + 2 * KH;
    __shared__ float vs[RB][W2];
    const int tx = threadIdx.x, y0 = blockIdx.y * RB, x0 = blockIdx.x * 256;
    float k[KH + 1];
#pragma unroll
    for (int j = 0; j <= KH; ++j) k[j] = K.k[j];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int i = tx + 256 * pass;
        if (i >= W2) break;
        const int xc = gidx<BORDER, true>(min(x0 + i - KH, w - 1 + KH), w);
        float c[NL];
#pragma unroll
        for (int r = 0; r < NL; ++r) c[r] = at(gidx<BORDER, true>(min(y0 - KH + r, h - 1 + KH), h), xc);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            float v = c[r + KH] * k[0];
#pragma unroll
            for (int j = 1; j <= KH; ++j) v += (c[r + KH - j] + c[r + KH + j]) * k[j];
            vs[r][i] = v;
        }
    }
    __syncthreads();
    const int x = x0 + tx;
    if (x >= w) return;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        if (y0 + r >= h) break;
        const float *q = &vs[r][tx + KH];
        float res = q[0] * k[0];
#pragma unroll
        for (int i = 1; i <= KH; ++i) res += (q[-i] + q[i]) * k[i];
        dst[(long long)(y0 + r) * ld + x] = res;
    }
}
template <int BORDER, int KH>
__global__ __launch_bounds__(256) void k_gaussian_blur_t(const float *src, float *dst, int w, int h, int ld, Taps K, long long bs, int nf, long long fs)
{
    { const long long o = (long long)(blockIdx.z / nf) * bs + (long long)(blockIdx.z % nf) * fs; src += o; dst += o; }
    blur_tile<BORDER, KH>([&](int r, int c) { return src[(long long)r * ld + c]; }, dst, w, h, ld, K);
}
// ... reading the CALLER's matrices (round 5): blockIdx.z = pair * 2 + frame, frame 0 / 1 of pair p = T.a[p] / T.b[p] with row steps T.sa / T.sb
// in bytes; CV_8UC1 converted on the fly (convertTo(CV_32F), farneback.cpp:342-345: exact) or CV_32FC1.  The conversion pass and its
// two planes per pair are gone, and the four blurs of an 8-bit pair read a quarter of the bytes.
template <int BORDER, int KH, bool U8>
__global__ __launch_bounds__(256) void k_gaussian_blur_tab(FmtTab T, float *dst, int w, int h, int ld, Taps K, long long bs, long long fs)
{
    const int p = blockIdx.z >> 1, f = blockIdx.z & 1;
    dst += (long long)p * bs + (long long)f * fs;
    const char *base = (const char *)(f ? T.b[p] : T.a[p]);
    const long long step = f ? T.sb[p] : T.sa[p];
    if (U8) blur_tile<BORDER, KH>([&](int r, int c) { return (float)((const unsigned char *)(base + (long long)r * step))[c]; }, dst, w, h, ld, K);
    else blur_tile<BORDER, KH>([&](int r, int c) { return ((const float *)(base + (long long)r * step))[c]; }, dst, w, h, ld, K);
}

// ------------------------------------------------------------------ polynomial expansion
// farneback.cu:66-119: vertical pass (g, xg, xxg) into 3 LDS rows, horizontal pass -> 5 coefficient planes.
// RS: `src` is the FULL-SIZE blurred frame (sw x sh, pitch sld) and the level image the expansion reads is its cuda::resize
// (INTER_LINEAR, cudawarping/src/cuda/resize.cu:234-269) to w x h, sampled on the fly with the arithmetic of k_resize -- the same
// bits as resizing into a plane first, one launch (and one plane round trip) less per level: what a single pair needs (round 4).
struct RsSrc { int sw, sh, sld; double scx, scy; };
template <int N, bool RS>
__global__ __launch_bounds__(256) void k_poly_exp(const float *src, float *dst, int w, int h, int ld, PolyC C, long long bs, int nf, long long fs_src,
                                                  long long fs_dst, RsSrc rs)
{
    __shared__ float smem[3 * 256];
    src += (long long)(blockIdx.z / nf) * bs + (long long)(blockIdx.z % nf) * fs_src;   // blockIdx.z = pair * nf + frame
    dst += (long long)(blockIdx.z / nf) * bs + (long long)(blockIdx.z % nf) * fs_dst;
    const int tx = threadIdx.x, y = blockIdx.y;
    const int x = blockIdx.x * (256 - 2 * N) + tx - N;
    float *row = smem + tx;
    const int xw = clampi(x, 0, w - 1);
    tvl1::RszX X;
    if (RS) X = tvl1::resize_xside<MI_SEM_CUDA_COMPAT>(xw, rs.sw, rs.scx);
    const auto at = [&](int yy) -> float {
        if (!RS) return src[(long long)yy * ld + xw];
        const tvl1::RszX Y = tvl1::resize_yside<MI_SEM_CUDA_COMPAT>(yy, rs.sh, rs.scy);
        return tvl1::resize_combine<MI_SEM_CUDA_COMPAT>(src + (long long)Y.i0 * rs.sld, src + (long long)Y.i1 * rs.sld, X, Y);
    };
    {
        float a = at(y) * C.g[0], b = 0.f, c = 0.f;
#pragma unroll
        for (int k = 1; k <= N; ++k) {
            const float t0 = at(max(y - k, 0));
            const float t1 = at(min(y + k, h - 1));
            a += C.g[k] * (t0 + t1);
            b += C.xg[k] * (t1 - t0);
            c += C.xxg[k] * (t0 + t1);
        }
        row[0] = a; row[256] = b; row[512] = c;
    }
    __syncthreads();
    if (tx >= N && tx + N < 256 && x < w) {
        float b1 = C.g[0] * row[0], b3 = C.g[0] * row[256], b5 = C.g[0] * row[512];
        float b2 = 0, b4 = 0, b6 = 0;
#pragma unroll
        for (int k = 1; k <= N; ++k) {
            b1 += (row[k] + row[-k]) * C.g[k];
            b4 += (row[k] + row[-k]) * C.xxg[k];
            b2 += (row[k] - row[-k]) * C.xg[k];
            b3 += (row[k + 256] + row[-k + 256]) * C.g[k];
            b6 += (row[k + 256] - row[-k + 256]) * C.xg[k];
            b5 += (row[k + 512] + row[-k + 512]) * C.g[k];
        }
        const long long ps = (long long)ld * h, o = (long long)y * ld + x;
        dst[o] = b3 * C.ig11;
        dst[ps + o] = b2 * C.ig11;
        dst[2 * ps + o] = b1 * C.ig03 + b5 * C.ig33;
        dst[3 * ps + o] = b1 * C.ig03 + b4 * C.ig33;
        dst[4 * ps + o] = b6 * C.ig55;
    }
}

// The expansion on RB-row tiles (round 5): the one-row kernel reads the 2 N + 1 rows of a column per output row (r16g: 160 us for the 64
// frames of a 32-pair 640 x 480 level, 2.9 TB/s of its 24 B/px).  A thread loads the RB + 2 N rows of its column once, all in flight,
// and forms the RB triples of vertical sums from registers; the horizontal pass is the one-row kernel's.  Same sums in the same order.
template <int N, int RB>
__global__ __launch_bounds__(256) void k_poly_exp_t(const float *src, float *dst, int w, int h, int ld, PolyC C, long long bs, int nf, long long fs_src,
                                                    long long fs_dst)
{
    __shared__ float smem[RB][3 * 256];
    src += (long long)(blockIdx.z / nf) * bs + (long long)(blockIdx.z % nf) * fs_src;   // blockIdx.z = pair * nf + frame
    dst += (long long)(blockIdx.z / nf) * bs + (long long)(blockIdx.z % nf) * fs_dst;
    const int tx = threadIdx.x, y0 = blockIdx.y * RB;
    const int x = blockIdx.x * (256 - 2 * N) + tx - N;
    const float *P = src + clampi(x, 0, w - 1);
    float c[RB + 2 * N];
#pragma unroll
    for (int r = 0; r < RB + 2 * N; ++r) c[r] = P[(long long)clampi(y0 + r - N, 0, h - 1) * ld];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        float a = c[r + N] * C.g[0], b = 0.f, cc = 0.f;
#pragma unroll
        for (int k = 1; k <= N; ++k) {
            const float t0 = c[r + N - k], t1 = c[r + N + k];
            a += C.g[k] * (t0 + t1);
            b += C.xg[k] * (t1 - t0);
            cc += C.xxg[k] * (t0 + t1);
        }
        smem[r][tx] = a; smem[r][256 + tx] = b; smem[r][512 + tx] = cc;
    }
    __syncthreads();
    if (!(tx >= N && tx + N < 256 && x < w)) return;
    const long long ps = (long long)ld * h;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int y = y0 + r;
        if (y >= h) break;
        const float *row = &smem[r][tx];
        float b1 = C.g[0] * row[0], b3 = C.g[0] * row[256], b5 = C.g[0] * row[512];
        float b2 = 0, b4 = 0, b6 = 0;
#pragma unroll
        for (int k = 1; k <= N; ++k) {
            b1 += (row[k] + row[-k]) * C.g[k];
            b4 += (row[k] + row[-k]) * C.xxg[k];
            b2 += (row[k] - row[-k]) * C.xg[k];
            b3 += (row[k + 256] + row[-k + 256]) * C.g[k];
            b6 += (row[k + 256] - row[-k + 256]) * C.xg[k];
            b5 += (row[k + 512] + row[-k + 512]) * C.g[k];
        }
        const long long o = (long long)y * ld + x;
        dst[o] = b3 * C.ig11;
        dst[ps + o] = b2 * C.ig11;
        dst[2 * ps + o] = b1 * C.ig03 + b5 * C.ig33;
        dst[3 * ps + o] = b1 * C.ig03 + b4 * C.ig33;
        dst[4 * ps + o] = b6 * C.ig55;
    }
}

// ------------------------------------------------------------------ matrix update (per pixel)
// farneback.cu:156-241; border attenuation table of :246 inlined.
__device__ __forceinline__ float border_w(int d)
{
    return d < 2 ? 0.14f : (d < 5 ? 0.4472f : 1.f);
}

// The per-pixel update in two steps so that a caller can put the loads of SEVERAL pixels in flight before it uses any of them (round 5:
// a thread of the tiled iteration walked its four rows one after the other, each with three dependent round trips -- three planes of R0,
// the gather, two more planes of R0).  um_load issues the 5 + 20 loads of a pixel unconditionally: the 2 x 2 window's origin is clamped
// into the plane, so the addresses are valid whether the displaced position is inside (then they are the reference's) or not (then the
// values are not used).  um_finish is the arithmetic of farneback.cu:176-239 on those values, operation for operation.
struct UmTaps { float r0[5], p00[5], p01[5], p10[5], p11[5]; };
__device__ __forceinline__ void um_load(int x, int y, int w, int h, int ld, float dx, float dy, const float *R0, const float *R1, UmTaps &T)
{
    const long long ps = (long long)ld * h;
    const int x1 = (int)floorf(x + dx), y1 = (int)floorf(y + dy);
    // column + 1 of a one-column plane stays inside its row (ld is a multiple of 64, plane_of); row + 1 of a one-row plane would not.
    // The two columns of a window row are adjacent words: one 8-byte load, half the L1 tag lookups of two.
    const long long oy = h > 1 ? ld : 0;
    const float *P = R1 + (long long)clampi(y1, 0, max(h - 2, 0)) * ld + clampi(x1, 0, max(w - 2, 0));
    const float *Q = R0 + (long long)y * ld + x;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float *Pb = P + oy;
        T.p00[k] = P[0]; T.p01[k] = P[1]; T.p10[k] = Pb[0]; T.p11[k] = Pb[1];
        T.r0[k] = Q[0];
        P += ps; Q += ps;
    }
}
__device__ __forceinline__ void um_finish(int x, int y, int w, int h, float dx, float dy, const UmTaps &T, float (&m)[5])
{
    float fx = x + dx, fy = y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    fx -= x1; fy -= y1;
    float r2, r3, r4, r5, r6;
    if (x1 >= 0 && y1 >= 0 && x1 < w - 1 && y1 < h - 1) {
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        r2 = a00 * T.p00[0] + a01 * T.p01[0] + a10 * T.p10[0] + a11 * T.p11[0];
        r3 = a00 * T.p00[1] + a01 * T.p01[1] + a10 * T.p10[1] + a11 * T.p11[1];
        r4 = a00 * T.p00[2] + a01 * T.p01[2] + a10 * T.p10[2] + a11 * T.p11[2];
        r5 = a00 * T.p00[3] + a01 * T.p01[3] + a10 * T.p10[3] + a11 * T.p11[3];
        r6 = a00 * T.p00[4] + a01 * T.p01[4] + a10 * T.p10[4] + a11 * T.p11[4];
        r4 = (T.r0[2] + r4) * 0.5f;
        r5 = (T.r0[3] + r5) * 0.5f;
        r6 = (T.r0[4] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = T.r0[2];
        r5 = T.r0[3];
        r6 = T.r0[4] * 0.5f;
    }
    r2 = (T.r0[0] - r2) * 0.5f;
    r3 = (T.r0[1] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    const float scale = border_w(min(x, 5)) * border_w(min(y, 5)) * border_w(min(w - x - 1, 5)) * border_w(min(h - y - 1, 5));
    r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}
__device__ __forceinline__ void update_matrices_vals(int x, int y, int w, int h, int ld, float dx, float dy, const float *R0,
                                                     const float *R1, float (&m)[5])
{
    UmTaps T;
    um_load(x, y, w, h, ld, dx, dy, R0, R1, T);
    um_finish(x, y, w, h, dx, dy, T, m);
}
__device__ __forceinline__ void update_matrices_px(int x, int y, int w, int h, int ld, float dx, float dy, const float *R0,
                                                   const float *R1, float *M)
{
    float m[5];
    update_matrices_vals(x, y, w, h, ld, dx, dy, R0, R1, m);
    const long long ps = (long long)ld * h, o = (long long)y * ld + x;
    M[o] = m[0]; M[ps + o] = m[1]; M[2 * ps + o] = m[2]; M[3 * ps + o] = m[3]; M[4 * ps + o] = m[4];
}


__global__ __launch_bounds__(256) void k_update_matrices(const float *flowx, const float *flowy, const float *R0, const float *R1,
                                                         float *M, int w, int h, int ld, long long bs)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const long long po = (long long)blockIdx.z * bs;
    const long long o = (long long)y * ld + x;
    // flowx == nullptr: the zero flow the coarsest level starts from (farneback.cpp:405-409) without a cleared plane to read it from
    update_matrices_px(x, y, w, h, ld, flowx ? flowx[po + o] : 0.f, flowx ? flowy[po + o] : 0.f, R0 + po, R1 + po, M + po);
}

// ------------------------------------------------------------------ fused inner iteration
// boxFilter5 (farneback.cu:357-412) or gaussianBlur5<BrdReplicate> (:539-595) of M, then updateFlow (:267-286)
// and, when `update`, updateMatrices (:156-241) into Mout (a different buffer: other blocks still read M).
template <bool GAUSS>
__global__ __launch_bounds__(256) void k_iterate(const float *M, const float *R0, const float *R1, float *flowx, float *flowy,
                                                 float *Mout, int w, int h, int ld, int kh, float boxAreaInv, int update, Taps K, long long bs)
{
    extern __shared__ float smem[];
    {
        const long long po = (long long)blockIdx.z * bs;
        M += po; R0 += po; R1 += po; flowx += po; flowy += po; Mout += po;
    }
    const int tx = threadIdx.x, y = blockIdx.y, x = blockIdx.x * 256 + tx;
    const int smw = 256 + 2 * kh;
    const long long ps = (long long)ld * h;
    for (int i = tx; i < smw; i += 256) {
        const int xe = clampi((int)(blockIdx.x * 256) + i - kh, 0, w - 1);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float *P = M + k * ps;
            float v = GAUSS ? P[(long long)y * ld + xe] * K.k[0] : P[(long long)y * ld + xe];
#pragma unroll 4
            for (int j = 1; j <= kh; ++j) {
                const float s = P[(long long)max(y - j, 0) * ld + xe] + P[(long long)min(y + j, h - 1) * ld + xe];
                v += GAUSS ? s * K.k[j] : s;
            }
            smem[k * smw + i] = v;
        }
    }
    __syncthreads();
    if (x >= w) return;
    float res[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float *r = smem + k * smw + tx + kh;
        float v = GAUSS ? r[0] * K.k[0] : r[0];
        for (int i = 1; i <= kh; ++i) v += GAUSS ? (r[-i] + r[i]) * K.k[i] : r[-i] + r[i];
        res[k] = GAUSS ? v : v * boxAreaInv;
    }
    const float g11 = res[0], g12 = res[1], g22 = res[2], h1 = res[3], h2 = res[4];
    const float detInv = 1.f / (g11 * g22 - g12 * g12 + 1e-3f);
    const float fx = (g11 * h2 - g12 * h1) * detInv;
    const float fy = (g22 * h1 - g12 * h2) * detInv;
    const long long o = (long long)y * ld + x;
    flowx[o] = fx;
    flowy[o] = fy;
    if (update) update_matrices_px(x, y, w, h, ld, fx, fy, R0, R1, Mout);
}

// The same iteration for compile-time window half sizes, tiled: a workgroup owns R rows x 256 columns.  The vertical pass loads
// the R + 2 KH rows of a column ONCE per plane into registers -- independent loads, one L2 round trip instead of KH dependent
// ones, and (R + 2 KH) / R instead of 2 KH + 1 loads per output -- and forms the R sums in the reference's order (centre, then
// the symmetric pairs outwards), so results are bit-identical to k_iterate.  (column, plane) tasks are dealt round-robin to the
// threads: 5 x (256 + 2 KH) tasks in ceil(./256) rounds.
// TW = columns of a tile.  256: a thread owns one column and walks the R rows (large grids).  64 (R = 4: 64 x 4 outputs = one per
// thread; round 4): for grids that do not fill the device -- a single 640 x 480 pair, the coarse levels of any pyramid -- where a launch
// costs its LATENCY, not its work: the vertical pass is 2 rounds of independent loads instead of 6 and the horizontal pass + flow
// solve + matrix update run once per thread instead of four times in a row (r08h: 9-10 us per launch whatever the level size).
// The sums are formed in the same order: bit-identical planes.
#ifndef MI_FB_VB
#define MI_FB_VB 2   // rounds of the vertical pass in flight together
#endif
#ifndef MI_FB_UB
#define MI_FB_UB 2   // rows of a thread whose matrix-update loads are in flight together
#endif
template <bool GAUSS, int KH, int R, int TW = 256>
__global__ __launch_bounds__(256) void k_iterate_t(const float *M, const float *R0, const float *R1, float *flowx, float *flowy,
                                                   float *Mout, int w, int h, int ld, float boxAreaInv, int update, Taps K, long long bs, int swz,
                                                   void *merged, long long merged_step)
{
    static_assert(TW == 256 || TW * R == 256, "narrow tiles: one output per thread");
    constexpr int SMW = TW + 2 * KH;
    __shared__ float smem[5][R][SMW];
    // workgroup -> (column tile, row tile, pair).  Consecutive workgroup ids go to different XCDs (8, each with its own L2), so
    // with the plain mapping the R + 2 KH rows a tile shares with its vertical neighbours are fetched from memory by several
    // XCDs (r02s: 1.5 GB fetched per launch against 0.59 GB of compulsory reads).  swz: every XCD takes a contiguous run of tiles.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (swz) {
        const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
        const unsigned orig = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned xcd = orig & 7u, qq = nwg >> 3, rr = nwg & 7u;
        const unsigned lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
        by = lid % gridDim.y; bx = (lid / gridDim.y) % gridDim.x; bz = lid / (gridDim.x * gridDim.y);   // rows of one column tile first
    }
    {
        const long long po = (long long)bz * bs;
        M += po; R0 += po; R1 += po; flowx += po; flowy += po; Mout += po;
    }
    const int tx = threadIdx.x, y0 = by * R, x = bx * TW + (TW == 256 ? tx : tx % TW);
    const long long ps = (long long)ld * h;
    // vertical pass: the (column, plane) tasks in rounds of 256, VB rounds' loads in flight together (round 5: the rounds used to wait for
    // each other, six dependent round trips per 256-column tile).  A thread past the last task repeats it (same value to the same LDS word).
    constexpr int NT = 5 * SMW, NRND = (NT + 255) / 256, VB = MI_FB_VB < NRND ? MI_FB_VB : NRND;
#pragma unroll
    for (int q0 = 0; q0 < NRND; q0 += VB) {
        float c[VB][R + 2 * KH];
#pragma unroll
        for (int v = 0; v < VB; ++v) {
            if (q0 + v >= NRND) break;
            const int t = min(tx + 256 * (q0 + v), NT - 1);
            const int k = t / SMW, i = t - k * SMW;
            const float *P = M + k * ps + clampi(bx * TW + i - KH, 0, w - 1);
#pragma unroll
            for (int r = 0; r < R + 2 * KH; ++r) c[v][r] = P[(long long)clampi(y0 + r - KH, 0, h - 1) * ld];
        }
#pragma unroll
        for (int v = 0; v < VB; ++v) {
            if (q0 + v >= NRND) break;
            const int t = min(tx + 256 * (q0 + v), NT - 1);   // (a conditional store would pull this round's loads into the branch, behind the others)
            const int k = t / SMW, i = t - k * SMW;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float u = GAUSS ? c[v][r + KH] * K.k[0] : c[v][r + KH];
#pragma unroll
                for (int j = 1; j <= KH; ++j) {
                    const float sj = c[v][r + KH - j] + c[v][r + KH + j];
                    u += GAUSS ? sj * K.k[j] : sj;
                }
                smem[k][r][i] = u;
            }
        }
    }
    __syncthreads();
    if (x >= w) return;
    // horizontal pass + 2 x 2 solve of every row of the thread, then the matrix update of those rows with all their loads in flight together
    constexpr int NR = TW == 256 ? R : 1;
    float fxs[NR], fys[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int r = TW == 256 ? rr : tx / TW;
        const int y = y0 + r;
        float res[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float *q = &smem[k][r][(TW == 256 ? tx : tx % TW) + KH];
            float v = GAUSS ? q[0] * K.k[0] : q[0];
#pragma unroll
            for (int i = 1; i <= KH; ++i) v += GAUSS ? (q[-i] + q[i]) * K.k[i] : q[-i] + q[i];
            res[k] = GAUSS ? v : v * boxAreaInv;
        }
        const float g11 = res[0], g12 = res[1], g22 = res[2], h1 = res[3], h2 = res[4];
        const float detInv = 1.f / (g11 * g22 - g12 * g12 + 1e-3f);
        const float fx = (g11 * h2 - g12 * h1) * detInv;
        const float fy = (g22 * h1 - g12 * h2) * detInv;
        fxs[rr] = fx; fys[rr] = fy;
        if (y < h) {
            const long long o = (long long)y * ld + x;
            flowx[o] = fx;
            flowy[o] = fy;
            // the last iteration of the finest level of a single pair also writes the caller's CV_32FC2 flow (cuda::merge, farneback.cpp:197-198)
            if (merged) ((float2 *)((char *)merged + (long long)y * merged_step))[x] = make_float2(fx, fy);
        }
    }
    if (!update) return;
    constexpr int UB = MI_FB_UB < NR ? MI_FB_UB : NR;   // rows whose gathers are in flight together
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += UB) {
        UmTaps T[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (r0 + u >= NR) break;
            const int y = min(y0 + (TW == 256 ? r0 + u : tx / TW), h - 1);   // a row past the image (last tile) loads the last row's and stores nothing
            um_load(x, y, w, h, ld, fxs[r0 + u], fys[r0 + u], R0, R1, T[u]);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (r0 + u >= NR) break;
            const int y = y0 + (TW == 256 ? r0 + u : tx / TW);
            if (y >= h) continue;
            float m[5];
            um_finish(x, y, w, h, fxs[r0 + u], fys[r0 + u], T[u], m);
            const long long o = (long long)y * ld + x;
#pragma unroll
            for (int k = 0; k < 5; ++k) Mout[k * ps + o] = m[k];
        }
    }
}

// TWO inner iterations in one launch (round 4; a single pair is a chain of ~5 us launches, 40 of them iterations): a workgroup still
// owns a 64 x 4 tile of the SECOND iteration's output and recomputes what that needs of the first -- the flow and the updated
// matrices M' on the tile grown by KH pixels on every side (E1: (64 + 2 KH) x (4 + 2 KH)), from M on the tile grown by 2 KH.  M' stays in
// LDS; the second iteration blurs it exactly as the one-iteration kernel blurs the stored plane (replicated borders = clamped indices
// into E1), so flow and Mout are BIT-IDENTICAL to two launches of k_iterate_t.  4.75 x the first iteration's work per tile: only for
// levels whose grid underfills the device, where a launch costs its latency.  Input M and output Mout must be different buffers
// (other workgroups still read M): the host swaps ONCE per fused pair.  1024 threads per workgroup: with 256 the E1 pass walked 4.75
// pixels per thread one after the other, each with its own gather round trip, and the launch took longer than the two it replaces (r08i).
template <bool GAUSS, int KH>
__global__ __launch_bounds__(1024) void k_iterate2_t(const float *M, const float *R0, const float *R1, float *flowx, float *flowy, float *Mout, int w,
                                                    int h, int ld, float boxAreaInv, int update, Taps K, long long bs, void *merged, long long merged_step)
{
    constexpr int TW = 64, TR = 4, EW = TW + 2 * KH, EH = TR + 2 * KH, SW = TW + 4 * KH, SH = TR + 4 * KH;
    __shared__ float smemA[5][EH][SW];   // vertical sums of M for the rows of E1 (later reused for the second iteration's vertical sums)
    __shared__ float smemM[5][EH][EW];   // M' on E1
    static_assert(sizeof(float) * 5 * (EH * SW + EH * EW) <= 65536 && TR * EW <= EH * SW, "two-iteration tile must fit the static LDS");
    {
        const long long po = (long long)blockIdx.z * bs;
        M += po; R0 += po; R1 += po; flowx += po; flowy += po; Mout += po;
    }
    const int tx = threadIdx.x, x0 = blockIdx.x * TW, y0 = blockIdx.y * TR;
    const long long ps = (long long)ld * h;
    // ---- first iteration, vertical pass: one (plane, column) task loads its SH rows once
    for (int t = tx; t < 5 * SW; t += 1024) {
        const int k = t / SW, i = t - k * SW;
        const float *P = M + k * ps + clampi(x0 - 2 * KH + i, 0, w - 1);
        float c[SH];
#pragma unroll
        for (int r = 0; r < SH; ++r) c[r] = P[(long long)clampi(y0 - 2 * KH + r, 0, h - 1) * ld];
#pragma unroll
        for (int e = 0; e < EH; ++e) {
            float v = GAUSS ? c[e + KH] * K.k[0] : c[e + KH];
#pragma unroll
            for (int j = 1; j <= KH; ++j) {
                const float sj = c[e + KH - j] + c[e + KH + j];
                v += GAUSS ? sj * K.k[j] : sj;
            }
            smemA[k][e][i] = v;
        }
    }
    __syncthreads();
    // ---- first iteration, horizontal pass + flow + matrix update on E1 (pixels inside the image only)
    for (int idx = tx; idx < EW * EH; idx += 1024) {
        const int e = idx / EW, ci = idx - e * EW;
        const int xc = x0 - KH + ci, ye = y0 - KH + e;
        if (xc < 0 || xc >= w || ye < 0 || ye >= h) continue;
        float res[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float *q = &smemA[k][e][ci + KH];
            float v = GAUSS ? q[0] * K.k[0] : q[0];
#pragma unroll
            for (int i = 1; i <= KH; ++i) v += GAUSS ? (q[-i] + q[i]) * K.k[i] : q[-i] + q[i];
            res[k] = GAUSS ? v : v * boxAreaInv;
        }
        const float g11 = res[0], g12 = res[1], g22 = res[2], h1 = res[3], h2 = res[4];
        const float detInv = 1.f / (g11 * g22 - g12 * g12 + 1e-3f);
        const float fx = (g11 * h2 - g12 * h1) * detInv;
        const float fy = (g22 * h1 - g12 * h2) * detInv;
        float m[5];
        update_matrices_vals(xc, ye, w, h, ld, fx, fy, R0, R1, m);
#pragma unroll
        for (int k = 0; k < 5; ++k) smemM[k][e][ci] = m[k];
    }
    __syncthreads();
    // ---- second iteration, vertical pass over M' (clamped rows / columns = the replicated border of the stored plane)
    float (*smemB)[TR][EW] = reinterpret_cast<float (*)[TR][EW]>(&smemA[0][0][0]);
    for (int t = tx; t < 5 * EW; t += 1024) {
        const int k = t / EW, ci = t - k * EW;
        const int cc = clampi(x0 - KH + ci, 0, w - 1) - (x0 - KH);
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            const int y = y0 + r;
            const auto at = [&](int yy) { return smemM[k][clampi(yy, 0, h - 1) - (y0 - KH)][cc]; };
            float v = GAUSS ? at(y) * K.k[0] : at(y);
#pragma unroll
            for (int j = 1; j <= KH; ++j) {
                const float sj = at(y - j) + at(y + j);
                v += GAUSS ? sj * K.k[j] : sj;
            }
            smemB[k][r][ci] = v;
        }
    }
    __syncthreads();
    // ---- second iteration, horizontal pass + flow (+ matrix update into Mout)
    const int r = tx / TW, x = x0 + tx % TW, y = y0 + r;
    if (tx >= TW * TR || x >= w || y >= h) return;
    float res[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float *q = &smemB[k][r][tx % TW + KH];
        float v = GAUSS ? q[0] * K.k[0] : q[0];
#pragma unroll
        for (int i = 1; i <= KH; ++i) v += GAUSS ? (q[-i] + q[i]) * K.k[i] : q[-i] + q[i];
        res[k] = GAUSS ? v : v * boxAreaInv;
    }
    const float g11 = res[0], g12 = res[1], g22 = res[2], h1 = res[3], h2 = res[4];
    const float detInv = 1.f / (g11 * g22 - g12 * g12 + 1e-3f);
    const float fx = (g11 * h2 - g12 * h1) * detInv;
    const float fy = (g22 * h1 - g12 * h2) * detInv;
    const long long o = (long long)y * ld + x;
    flowx[o] = fx;
    flowy[o] = fy;
    if (merged) ((float2 *)((char *)merged + (long long)y * merged_step))[x] = make_float2(fx, fy);
    if (update) update_matrices_px(x, y, w, h, ld, fx, fy, R0, R1, Mout);
}

// 5-plane blur only / flow solve only (stage-level entry points and tests)
template <bool GAUSS>
__global__ __launch_bounds__(256) void k_blur5(const float *M, float *dst, int w, int h, int ld, int kh, float boxAreaInv, Taps K)
{
    extern __shared__ float smem[];
    const int tx = threadIdx.x, y = blockIdx.y, x = blockIdx.x * 256 + tx;
    const int smw = 256 + 2 * kh;
    const long long ps = (long long)ld * h;
    for (int i = tx; i < smw; i += 256) {
        const int xe = clampi((int)(blockIdx.x * 256) + i - kh, 0, w - 1);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float *P = M + k * ps;
            float v = GAUSS ? P[(long long)y * ld + xe] * K.k[0] : P[(long long)y * ld + xe];
#pragma unroll 4
            for (int j = 1; j <= kh; ++j) {
                const float s = P[(long long)max(y - j, 0) * ld + xe] + P[(long long)min(y + j, h - 1) * ld + xe];
                v += GAUSS ? s * K.k[j] : s;
            }
            smem[k * smw + i] = v;
        }
    }
    __syncthreads();
    if (x >= w) return;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float *r = smem + k * smw + tx + kh;
        float v = GAUSS ? r[0] * K.k[0] : r[0];
        for (int i = 1; i <= kh; ++i) v += GAUSS ? (r[-i] + r[i]) * K.k[i] : r[-i] + r[i];
        dst[k * ps + (long long)y * ld + x] = GAUSS ? v : v * boxAreaInv;
    }
}

__global__ __launch_bounds__(256) void k_update_flow(const float *M, float *flowx, float *flowy, int w, int h, int ld)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const long long ps = (long long)ld * h, o = (long long)y * ld + x;
    const float g11 = M[o], g12 = M[ps + o], g22 = M[2 * ps + o], h1 = M[3 * ps + o], h2 = M[4 * ps + o];
    const float detInv = 1.f / (g11 * g22 - g12 * g12 + 1e-3f);
    flowx[o] = (g11 * h2 - g12 * h1) * detInv;
    flowy[o] = (g22 * h1 - g12 * h2) * detInv;
}

// ------------------------------------------------------------------ pyrDown (fastPyramids)
// cudawarping/src/cuda/pyr_down.cu:54-175, BrdReflect101, CV_32FC1
__global__ __launch_bounds__(256) void k_pyr_down(const float *src, int sw, int sh, int sld, float *dst, int dw, int dh, int dld, long long bs)
{
    src += (long long)blockIdx.z * bs; dst += (long long)blockIdx.z * bs;
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || y >= dh) return;
    const int sy = 2 * y;
    float v[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int x = reflect101(2 * dx + j - 2, sw);
        float sum;
        sum = 0.0625f * src[(long long)reflect101(sy - 2, sh) * sld + x];
        sum = sum + 0.25f * src[(long long)reflect101(sy - 1, sh) * sld + x];
        sum = sum + 0.375f * src[(long long)sy * sld + x];
        sum = sum + 0.25f * src[(long long)reflect101(sy + 1, sh) * sld + x];
        sum = sum + 0.0625f * src[(long long)reflect101(sy + 2, sh) * sld + x];
        v[j] = sum;
    }
    float sum;
    sum = 0.0625f * v[0];
    sum = sum + 0.25f * v[1];
    sum = sum + 0.375f * v[2];
    sum = sum + 0.25f * v[3];
    sum = sum + 0.0625f * v[4];
    dst[(long long)y * dld + dx] = sum;
}

// ------------------------------------------------------------------ host launchers
static inline dim3 grid2d(int w, int h) { return dim3(div_up(w, 64), div_up(h, 4)); }

// ---- the per-pair format steps of a BATCH in one launch each (round 5: a 32-pair batch spent 64 launches of ~5 us on them, 7 % of its
// time): blockIdx.z = pair, the caller's matrices arrive as a by-value table of up to kFmtPairs entries per launch
__global__ __launch_bounds__(256) void k_convert_batch(FmtTab T, int type, float *A, float *B, long long bs, int w, int h, int ld)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int p = blockIdx.z;
    const void *a = T.a[p], *b = T.b[p];
    const long long sa = T.sa[p], sb = T.sb[p];
    const long long o = (long long)p * bs + (long long)y * ld + x;
    if (type == MI_8UC1) {   // convertTo(CV_32F), farneback.cpp:342-345 (no scaling)
        A[o] = (float)((const unsigned char *)a)[(long long)y * sa + x];
        B[o] = (float)((const unsigned char *)b)[(long long)y * sb + x];
    } else {
        A[o] = ((const float *)((const char *)a + (long long)y * sa))[x];
        B[o] = ((const float *)((const char *)b + (long long)y * sb))[x];
    }
}
__global__ __launch_bounds__(256) void k_merge_flow_batch(const float *fx, const float *fy, FmtTab T, long long bs, int w, int h, int ld)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int p = blockIdx.z;
    const long long o = (long long)p * bs + (long long)y * ld + x;
    ((float2 *)((char *)const_cast<void *>(T.a[p]) + (long long)y * T.sa[p]))[x] = make_float2(fx[o], fy[o]);   // cuda::merge, farneback.cpp:197-198
}
int convert_batch(const FmtTab &T, int n, int type, float *A, float *B, long long bs, const Plane &g, hipStream_t s)
{
    dim3 grid = grid2d(g.w, g.h);
    grid.z = n;
    hipLaunchKernelGGL(k_convert_batch, grid, dim3(256), 0, s, T, type, A, B, bs, g.w, g.h, g.ld);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}
int merge_flow_batch(const float *fx, const float *fy, const FmtTab &T, int n, long long bs, const Plane &g, hipStream_t s)
{
    dim3 grid = grid2d(g.w, g.h);
    grid.z = n;
    hipLaunchKernelGGL(k_merge_flow_batch, grid, dim3(256), 0, s, fx, fy, T, bs, g.w, g.h, g.ld);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int convert(const void *a, long long sa, const void *b, long long sb, int type, float *A, float *B, const Plane &g, hipStream_t s)
{
    hipLaunchKernelGGL(k_convert, grid2d(g.w, g.h), dim3(256), 0, s, a, sa, b, sb, type, A, B, g.w, g.h, g.ld);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}
int split_flow(const void *flow, long long sf, float *fx, float *fy, const Plane &g, hipStream_t s)
{
    hipLaunchKernelGGL(k_split_flow, grid2d(g.w, g.h), dim3(256), 0, s, flow, sf, fx, fy, g.w, g.h, g.ld);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}
int merge_flow(const float *fx, const float *fy, void *flow, long long sf, const Plane &g, hipStream_t s)
{
    hipLaunchKernelGGL(k_merge_flow, grid2d(g.w, g.h), dim3(256), 0, s, fx, fy, flow, sf, g.w, g.h, g.ld);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int gaussian_blur(const float *src, float *dst, const Plane &g, int kh, const Taps &K, int border, hipStream_t s, int nf, long long fs)
{
    MI_REQUIRE(kh >= 0 && kh <= MI_FB_MAX_KSIZE_HALF, MI_ERR_BAD_ARG, "Gaussian kernel half size out of range");
    const dim3 grid(div_up(g.w, 256), g.h, g.batch * nf);
    const size_t lds = sizeof(float) * (256 + 2 * kh);
    const bool fast = kh < g.w && kh < g.h && g.w >= 2 && g.h >= 2;
    if (border != MI_BORDER_REFLECT101 && border != MI_BORDER_REPLICATE) { set_error("unsupported border mode %d", border); return MI_ERR_BAD_ARG; }   // farneback.cu:510-517: only these two
    // tiled form for the half sizes the pyramid of pyrScale 0.5 (1, 1, 4, 9, 19) and its neighbours use
    if (fast && tuning().fb_blur_tiled && (kh == 1 || kh == 2 || kh == 3 || kh == 4 || kh == 9 || kh == 19)) {
#define MI_FB_BLUR(KH) case KH: { const dim3 tg(div_up(g.w, 256), div_up(g.h, BlurTile<KH>::RB), g.batch * nf);                                  \
        if (border == MI_BORDER_REFLECT101) hipLaunchKernelGGL((k_gaussian_blur_t<MI_BORDER_REFLECT101, KH>), tg, dim3(256), 0, s, src, dst, g.w, g.h, g.ld, K, g.bs, nf, fs); \
        else hipLaunchKernelGGL((k_gaussian_blur_t<MI_BORDER_REPLICATE, KH>), tg, dim3(256), 0, s, src, dst, g.w, g.h, g.ld, K, g.bs, nf, fs); } break;
        switch (kh) { MI_FB_BLUR(1) MI_FB_BLUR(2) MI_FB_BLUR(3) MI_FB_BLUR(4) MI_FB_BLUR(9) MI_FB_BLUR(19) }
#undef MI_FB_BLUR
    } else if (border == MI_BORDER_REFLECT101) {
        if (fast) hipLaunchKernelGGL((k_gaussian_blur<MI_BORDER_REFLECT101, true>), grid, dim3(256), lds, s, src, dst, g.w, g.h, g.ld, kh, K, g.bs, nf, fs);
        else hipLaunchKernelGGL((k_gaussian_blur<MI_BORDER_REFLECT101, false>), grid, dim3(256), lds, s, src, dst, g.w, g.h, g.ld, kh, K, g.bs, nf, fs);
    } else
        hipLaunchKernelGGL((k_gaussian_blur<MI_BORDER_REPLICATE, true>), grid, dim3(256), lds, s, src, dst, g.w, g.h, g.ld, kh, K, g.bs, nf, fs);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

// the tiled blur straight from the caller's matrices (both frames of g.batch <= kFmtPairs pairs); false: no instantiation for this case
bool gaussian_blur_tab_ok(const Plane &g, int kh)
{
    const bool fast = kh < g.w && kh < g.h && g.w >= 2 && g.h >= 2;
    return fast && tuning().fb_blur_tiled && (kh == 1 || kh == 2 || kh == 3 || kh == 4 || kh == 9 || kh == 19);
}
int gaussian_blur_tab(const FmtTab &T, int type, float *dst, const Plane &g, int kh, const Taps &K, hipStream_t s, long long fs)
{
    MI_REQUIRE(gaussian_blur_tab_ok(g, kh) && g.batch <= kFmtPairs && (type == MI_8UC1 || type == MI_32FC1), MI_ERR_BAD_ARG, "no direct-source blur for this case");
#define MI_FB_BLUR(KH) case KH: { const dim3 tg(div_up(g.w, 256), div_up(g.h, BlurTile<KH>::RB), g.batch * 2);                                  \
        if (type == MI_8UC1) hipLaunchKernelGGL((k_gaussian_blur_tab<MI_BORDER_REFLECT101, KH, true>), tg, dim3(256), 0, s, T, dst, g.w, g.h, g.ld, K, g.bs, fs); \
        else hipLaunchKernelGGL((k_gaussian_blur_tab<MI_BORDER_REFLECT101, KH, false>), tg, dim3(256), 0, s, T, dst, g.w, g.h, g.ld, K, g.bs, fs); } break;
    switch (kh) { MI_FB_BLUR(1) MI_FB_BLUR(2) MI_FB_BLUR(3) MI_FB_BLUR(4) MI_FB_BLUR(9) MI_FB_BLUR(19) }
#undef MI_FB_BLUR
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int poly_exp(const float *src, float *dst5, const Plane &g, int polyN, const PolyC &C, hipStream_t s, int nf, long long fs_src, long long fs_dst,
             const Plane *resize_from)
{
    RsSrc rs;
    memset(&rs, 0, sizeof(rs));
    if (resize_from) {   // the scale factors exactly as tvl1::resize forms them for cv::cuda's semantics (cudawarping/src/resize.cpp:107)
        rs.sw = resize_from->w; rs.sh = resize_from->h; rs.sld = resize_from->ld;
        rs.scx = (double)(float)(1.0 / ((double)g.w / resize_from->w));
        rs.scy = (double)(float)(1.0 / ((double)g.h / resize_from->h));
    }
#define MI_PE(N) do { if (!resize_from && tuning().fb_poly_tiled && (long long)div_up(g.w, 256 - 2 * N) * div_up(g.h, 8) * g.batch * nf >= 1024) hipLaunchKernelGGL((k_poly_exp_t<N, 8>), dim3(div_up(g.w, 256 - 2 * N), div_up(g.h, 8), g.batch * nf), dim3(256), 0, s, src, dst5, g.w, g.h, g.ld, C, g.bs, nf, fs_src, fs_dst); \
                     else if (resize_from) hipLaunchKernelGGL((k_poly_exp<N, true>), dim3(div_up(g.w, 256 - 2 * N), g.h, g.batch * nf), dim3(256), 0, s, src, dst5, g.w, g.h, g.ld, C, g.bs, nf, fs_src, fs_dst, rs); \
                     else hipLaunchKernelGGL((k_poly_exp<N, false>), dim3(div_up(g.w, 256 - 2 * N), g.h, g.batch * nf), dim3(256), 0, s, src, dst5, g.w, g.h, g.ld, C, g.bs, nf, fs_src, fs_dst, rs); } while (0)
    if (polyN == 5) MI_PE(5);
    else if (polyN == 7) MI_PE(7);
    else { set_error("polyN must be 5 or 7"); return MI_ERR_BAD_ARG; }   // CV_Assert, farneback.cpp:316
#undef MI_PE
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

// The zoom of the coarser level's flow (cuda::resize + the 1 / pyrScale multiply, farneback.cpp:412-417) and the first matrix update
// of the level (:458) in one launch: the flow of a pixel is sampled with k_resize's arithmetic, stored, and used at once.
__global__ __launch_bounds__(256) void k_update_matrices_rs(const float *px, const float *py, RsSrc rs, float alpha, float *flowx, float *flowy,
                                                            const float *R0, const float *R1, float *M, int w, int h, int ld, long long bs)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const long long po = (long long)blockIdx.z * bs;
    const tvl1::RszX X = tvl1::resize_xside<MI_SEM_CUDA_COMPAT>(x, rs.sw, rs.scx), Y = tvl1::resize_yside<MI_SEM_CUDA_COMPAT>(y, rs.sh, rs.scy);
    float fx = tvl1::resize_combine<MI_SEM_CUDA_COMPAT>(px + po + (long long)Y.i0 * rs.sld, px + po + (long long)Y.i1 * rs.sld, X, Y);
    float fy = tvl1::resize_combine<MI_SEM_CUDA_COMPAT>(py + po + (long long)Y.i0 * rs.sld, py + po + (long long)Y.i1 * rs.sld, X, Y);
    if (alpha != 1.0f) { fx = fx * alpha; fy = fy * alpha; }
    const long long o = (long long)y * ld + x;
    flowx[po + o] = fx;
    flowy[po + o] = fy;
    update_matrices_px(x, y, w, h, ld, fx, fy, R0 + po, R1 + po, M + po);
}
int update_matrices_resized(const float *prevx, const float *prevy, const Plane &gprev, float alpha, float *flowx, float *flowy, const float *R0,
                            const float *R1, float *M, const Plane &g, hipStream_t s)
{
    RsSrc rs;
    rs.sw = gprev.w; rs.sh = gprev.h; rs.sld = gprev.ld;
    rs.scx = (double)(float)(1.0 / ((double)g.w / gprev.w));
    rs.scy = (double)(float)(1.0 / ((double)g.h / gprev.h));
    dim3 grid = grid2d(g.w, g.h);
    grid.z = g.batch;
    hipLaunchKernelGGL(k_update_matrices_rs, grid, dim3(256), 0, s, prevx, prevy, rs, alpha, flowx, flowy, R0, R1, M, g.w, g.h, g.ld, g.bs);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int update_matrices(const float *flowx, const float *flowy, const float *R0, const float *R1, float *M, const Plane &g, hipStream_t s)
{
    dim3 grid = grid2d(g.w, g.h);
    grid.z = g.batch;
    hipLaunchKernelGGL(k_update_matrices, grid, dim3(256), 0, s, flowx, flowy, R0, R1, M, g.w, g.h, g.ld, g.bs);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int iterate(const float *M, const float *R0, const float *R1, float *flowx, float *flowy, float *Mout, const Plane &g, int ksize,
            const Taps *gauss, bool update, hipStream_t s, void *merged, long long merged_step, bool *did_merge)
{
    if (did_merge) *did_merge = false;
    const int kh = ksize / 2;
    MI_REQUIRE(kh >= 0 && kh <= MI_FB_MAX_KSIZE_HALF, MI_ERR_BAD_ARG, "winSize out of range");
    const dim3 grid(div_up(g.w, 256), g.h, g.batch);
    const size_t lds = sizeof(float) * 5 * (256 + 2 * kh);
    const float inv = 1.f / ((1 + 2 * kh) * (1 + 2 * kh));
    Taps none;
    memset(&none, 0, sizeof(none));
    const int R = tuning().fb_rows, swz = tuning().fb_swz;
    dim3 tgrid(div_up(g.w, 256), div_up(g.h, R), g.batch);
    const Taps &K = gauss ? *gauss : none;
    const int upd = update ? 1 : 0;
    // narrow tiles where the 256-column grid would leave most of the device idle (fewer workgroups than two per CU): the launch then
    // costs its dependent steps, and a 64 x 4 tile has a third of them.  MIFLOW_FB_NARROW=0 / 1 forces the choice (tuning, tests).
    const int narrow_env = tuning().fb_narrow;
    const bool narrow = narrow_env >= 0 ? narrow_env != 0 : (long long)tgrid.x * tgrid.y * tgrid.z < 2LL * (device_simds() / 4);
    if (narrow && R == 4) tgrid = dim3(div_up(g.w, 64), div_up(g.h, 4), g.batch);
#define MI_FB_LAUNCH(G, KH, RR, TW) hipLaunchKernelGGL((k_iterate_t<G, KH, RR, TW>), tgrid, dim3(256), 0, s, M, R0, R1, flowx, flowy, Mout, g.w, g.h, g.ld, inv, upd, K, g.bs, swz, mg, merged_step)
#ifdef MIFLOW_EXPERIMENTS   // 8-row tiles (MIFLOW_FB_ROWS=8) lost their A/B: experiments build only
#define MI_FB_R8(G, KH) if (R == 8) MI_FB_LAUNCH(G, KH, 8, 256); else
#else
#define MI_FB_R8(G, KH)
#endif
#define MI_FB_TILED(KH)                                                                  \
    case KH:                                                                             \
        if (gauss) { MI_FB_R8(true, KH) if (narrow) MI_FB_LAUNCH(true, KH, 4, 64); else MI_FB_LAUNCH(true, KH, 4, 256); }   \
        else { MI_FB_R8(false, KH) if (narrow) MI_FB_LAUNCH(false, KH, 4, 64); else MI_FB_LAUNCH(false, KH, 4, 256); }       \
        break;
    void *mg = (merged && g.batch == 1 && tuning().fb_tiled && (kh == 4 || kh == 6 || kh == 7 || kh == 10)) ? merged : nullptr;
    if (did_merge) *did_merge = mg != nullptr;
    switch (tuning().fb_tiled ? kh : -1) {
        MI_FB_TILED(4) MI_FB_TILED(6) MI_FB_TILED(7) MI_FB_TILED(10)   // winSize 9, 13 (the default), 15, 21
    default:
        if (gauss) hipLaunchKernelGGL(k_iterate<true>, grid, dim3(256), lds, s, M, R0, R1, flowx, flowy, Mout, g.w, g.h, g.ld, kh, inv, upd, K, g.bs);
        else hipLaunchKernelGGL(k_iterate<false>, grid, dim3(256), lds, s, M, R0, R1, flowx, flowy, Mout, g.w, g.h, g.ld, kh, inv, upd, K, g.bs);
    }
#undef MI_FB_TILED
#undef MI_FB_R8
#undef MI_FB_LAUNCH
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

// two fused iterations (k_iterate2_t); false when the window size has no instantiation (the caller then launches twice)
bool iterate2_supported(int ksize) { const int kh = ksize / 2; return tuning().fb_tiled && (kh == 4 || kh == 6 || kh == 7); }
int iterate2(const float *M, const float *R0, const float *R1, float *flowx, float *flowy, float *Mout, const Plane &g, int ksize, const Taps *gauss,
             bool update, hipStream_t s, void *merged, long long merged_step, bool *did_merge)
{
    const int kh = ksize / 2;
    MI_REQUIRE(iterate2_supported(ksize) && M != Mout, MI_ERR_BAD_ARG, "no two-iteration kernel for this window size");
    const float inv = 1.f / ((1 + 2 * kh) * (1 + 2 * kh));
    Taps none;
    memset(&none, 0, sizeof(none));
    const Taps &K = gauss ? *gauss : none;
    void *mg = (merged && g.batch == 1) ? merged : nullptr;
    if (did_merge) *did_merge = mg != nullptr;
    const dim3 grid(div_up(g.w, 64), div_up(g.h, 4), g.batch);
#define MI_FB2(KH) case KH: if (gauss) hipLaunchKernelGGL((k_iterate2_t<true, KH>), grid, dim3(1024), 0, s, M, R0, R1, flowx, flowy, Mout, g.w, g.h, g.ld, inv, update ? 1 : 0, K, g.bs, mg, merged_step); \
                            else hipLaunchKernelGGL((k_iterate2_t<false, KH>), grid, dim3(1024), 0, s, M, R0, R1, flowx, flowy, Mout, g.w, g.h, g.ld, inv, update ? 1 : 0, K, g.bs, mg, merged_step); break;
    switch (kh) { MI_FB2(4) MI_FB2(6) MI_FB2(7) }
#undef MI_FB2
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int blur5(const float *M, float *dst, const Plane &g, int ksize, const Taps *gauss, hipStream_t s)
{
    const int kh = ksize / 2;
    MI_REQUIRE(kh >= 0 && kh <= MI_FB_MAX_KSIZE_HALF, MI_ERR_BAD_ARG, "winSize out of range");
    const dim3 grid(div_up(g.w, 256), g.h);
    const size_t lds = sizeof(float) * 5 * (256 + 2 * kh);
    const float inv = 1.f / ((1 + 2 * kh) * (1 + 2 * kh));
    Taps none;
    memset(&none, 0, sizeof(none));
    if (gauss) hipLaunchKernelGGL(k_blur5<true>, grid, dim3(256), lds, s, M, dst, g.w, g.h, g.ld, kh, inv, *gauss);
    else hipLaunchKernelGGL(k_blur5<false>, grid, dim3(256), lds, s, M, dst, g.w, g.h, g.ld, kh, inv, none);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int update_flow(const float *M, float *flowx, float *flowy, const Plane &g, hipStream_t s)
{
    hipLaunchKernelGGL(k_update_flow, grid2d(g.w, g.h), dim3(256), 0, s, M, flowx, flowy, g.w, g.h, g.ld);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int pyr_down(const float *src, const Plane &gs, float *dst, const Plane &gd, hipStream_t s)
{
    dim3 grid = grid2d(gd.w, gd.h);
    grid.z = gd.batch;
    hipLaunchKernelGGL(k_pyr_down, grid, dim3(256), 0, s, src, gs.w, gs.h, gs.ld, dst, gd.w, gd.h, gd.ld, gd.bs);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace fb
}  // namespace mi
