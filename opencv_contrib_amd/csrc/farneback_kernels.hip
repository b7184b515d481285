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
__global__ __launch_bounds__(256) void k_gaussian_blur(const float *src, float *dst, int w, int h, int ld, int kh, Taps K, long long bs)
{
    extern __shared__ float row[];
    src += (long long)blockIdx.z * bs; dst += (long long)blockIdx.z * bs;   // pair of the batch
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

// ------------------------------------------------------------------ polynomial expansion
// farneback.cu:66-119: vertical pass (g, xg, xxg) into 3 LDS rows, horizontal pass -> 5 coefficient planes.
template <int N>
__global__ __launch_bounds__(256) void k_poly_exp(const float *src, float *dst, int w, int h, int ld, PolyC C, long long bs)
{
    __shared__ float smem[3 * 256];
    src += (long long)blockIdx.z * bs; dst += (long long)blockIdx.z * bs;
    const int tx = threadIdx.x, y = blockIdx.y;
    const int x = blockIdx.x * (256 - 2 * N) + tx - N;
    float *row = smem + tx;
    const int xw = clampi(x, 0, w - 1);
    {
        float a = src[(long long)y * ld + xw] * C.g[0], b = 0.f, c = 0.f;
#pragma unroll
        for (int k = 1; k <= N; ++k) {
            const float t0 = src[(long long)max(y - k, 0) * ld + xw];
            const float t1 = src[(long long)min(y + k, h - 1) * ld + xw];
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

// ------------------------------------------------------------------ matrix update (per pixel)
// farneback.cu:156-241; border attenuation table of :246 inlined.
__device__ __forceinline__ float border_w(int d)
{
    return d < 2 ? 0.14f : (d < 5 ? 0.4472f : 1.f);
}

__device__ __forceinline__ void update_matrices_px(int x, int y, int w, int h, int ld, float dx, float dy, const float *R0,
                                                   const float *R1, float *M)
{
    const long long ps = (long long)ld * h;
    float fx = x + dx, fy = y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    fx -= x1; fy -= y1;
    const long long o = (long long)y * ld + x;
    float r2, r3, r4, r5, r6;
    if (x1 >= 0 && y1 >= 0 && x1 < w - 1 && y1 < h - 1) {
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        const float *P = R1 + (long long)y1 * ld + x1;
        r2 = a00 * P[0] + a01 * P[1] + a10 * P[ld] + a11 * P[ld + 1]; P += ps;
        r3 = a00 * P[0] + a01 * P[1] + a10 * P[ld] + a11 * P[ld + 1]; P += ps;
        r4 = a00 * P[0] + a01 * P[1] + a10 * P[ld] + a11 * P[ld + 1]; P += ps;
        r5 = a00 * P[0] + a01 * P[1] + a10 * P[ld] + a11 * P[ld + 1]; P += ps;
        r6 = a00 * P[0] + a01 * P[1] + a10 * P[ld] + a11 * P[ld + 1];
        r4 = (R0[2 * ps + o] + r4) * 0.5f;
        r5 = (R0[3 * ps + o] + r5) * 0.5f;
        r6 = (R0[4 * ps + o] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = R0[2 * ps + o];
        r5 = R0[3 * ps + o];
        r6 = R0[4 * ps + o] * 0.5f;
    }
    r2 = (R0[o] - r2) * 0.5f;
    r3 = (R0[ps + o] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    const float scale = border_w(min(x, 5)) * border_w(min(y, 5)) * border_w(min(w - x - 1, 5)) * border_w(min(h - y - 1, 5));
    r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    M[o] = r4 * r4 + r6 * r6;
    M[ps + o] = (r4 + r5) * r6;
    M[2 * ps + o] = r5 * r5 + r6 * r6;
    M[3 * ps + o] = r4 * r2 + r6 * r3;
    M[4 * ps + o] = r6 * r2 + r5 * r3;
}

__global__ __launch_bounds__(256) void k_update_matrices(const float *flowx, const float *flowy, const float *R0, const float *R1,
                                                         float *M, int w, int h, int ld, long long bs)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const long long po = (long long)blockIdx.z * bs;
    const long long o = (long long)y * ld + x;
    update_matrices_px(x, y, w, h, ld, flowx[po + o], flowy[po + o], R0 + po, R1 + po, M + po);
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
template <bool GAUSS, int KH, int R>
__global__ __launch_bounds__(256) void k_iterate_t(const float *M, const float *R0, const float *R1, float *flowx, float *flowy,
                                                   float *Mout, int w, int h, int ld, float boxAreaInv, int update, Taps K, long long bs, int swz)
{
    constexpr int SMW = 256 + 2 * KH;
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
    const int tx = threadIdx.x, y0 = by * R, x = bx * 256 + tx;
    const long long ps = (long long)ld * h;
    for (int t = tx; t < 5 * SMW; t += 256) {
        const int k = t / SMW, i = t - k * SMW;
        const int xe = clampi(bx * 256 + i - KH, 0, w - 1);
        const float *P = M + k * ps + xe;
        float c[R + 2 * KH];
#pragma unroll
        for (int r = 0; r < R + 2 * KH; ++r) c[r] = P[(long long)clampi(y0 + r - KH, 0, h - 1) * ld];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float v = GAUSS ? c[r + KH] * K.k[0] : c[r + KH];
#pragma unroll
            for (int j = 1; j <= KH; ++j) {
                const float sj = c[r + KH - j] + c[r + KH + j];
                v += GAUSS ? sj * K.k[j] : sj;
            }
            smem[k][r][i] = v;
        }
    }
    __syncthreads();
    if (x >= w) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int y = y0 + r;
        if (y >= h) break;
        float res[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float *q = &smem[k][r][tx + KH];
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
        if (update) update_matrices_px(x, y, w, h, ld, fx, fy, R0, R1, Mout);
    }
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

int gaussian_blur(const float *src, float *dst, const Plane &g, int kh, const Taps &K, int border, hipStream_t s)
{
    MI_REQUIRE(kh >= 0 && kh <= MI_FB_MAX_KSIZE_HALF, MI_ERR_BAD_ARG, "Gaussian kernel half size out of range");
    const dim3 grid(div_up(g.w, 256), g.h, g.batch);
    const size_t lds = sizeof(float) * (256 + 2 * kh);
    const bool fast = kh < g.w && kh < g.h && g.w >= 2 && g.h >= 2;
    if (border == MI_BORDER_REFLECT101) {
        if (fast) hipLaunchKernelGGL((k_gaussian_blur<MI_BORDER_REFLECT101, true>), grid, dim3(256), lds, s, src, dst, g.w, g.h, g.ld, kh, K, g.bs);
        else hipLaunchKernelGGL((k_gaussian_blur<MI_BORDER_REFLECT101, false>), grid, dim3(256), lds, s, src, dst, g.w, g.h, g.ld, kh, K, g.bs);
    } else if (border == MI_BORDER_REPLICATE)
        hipLaunchKernelGGL((k_gaussian_blur<MI_BORDER_REPLICATE, true>), grid, dim3(256), lds, s, src, dst, g.w, g.h, g.ld, kh, K, g.bs);
    else { set_error("unsupported border mode %d", border); return MI_ERR_BAD_ARG; }   // farneback.cu:510-517: only these two
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int poly_exp(const float *src, float *dst5, const Plane &g, int polyN, const PolyC &C, hipStream_t s)
{
    if (polyN == 5)
        hipLaunchKernelGGL(k_poly_exp<5>, dim3(div_up(g.w, 256 - 10), g.h, g.batch), dim3(256), 0, s, src, dst5, g.w, g.h, g.ld, C, g.bs);
    else if (polyN == 7)
        hipLaunchKernelGGL(k_poly_exp<7>, dim3(div_up(g.w, 256 - 14), g.h, g.batch), dim3(256), 0, s, src, dst5, g.w, g.h, g.ld, C, g.bs);
    else { set_error("polyN must be 5 or 7"); return MI_ERR_BAD_ARG; }   // CV_Assert, farneback.cpp:316
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
            const Taps *gauss, bool update, hipStream_t s)
{
    const int kh = ksize / 2;
    MI_REQUIRE(kh >= 0 && kh <= MI_FB_MAX_KSIZE_HALF, MI_ERR_BAD_ARG, "winSize out of range");
    const dim3 grid(div_up(g.w, 256), g.h, g.batch);
    const size_t lds = sizeof(float) * 5 * (256 + 2 * kh);
    const float inv = 1.f / ((1 + 2 * kh) * (1 + 2 * kh));
    Taps none;
    memset(&none, 0, sizeof(none));
    const int R = tuning().fb_rows, swz = tuning().fb_swz;
    const dim3 tgrid(div_up(g.w, 256), div_up(g.h, R), g.batch);
    const Taps &K = gauss ? *gauss : none;
    const int upd = update ? 1 : 0;
#define MI_FB_LAUNCH(G, KH, RR) hipLaunchKernelGGL((k_iterate_t<G, KH, RR>), tgrid, dim3(256), 0, s, M, R0, R1, flowx, flowy, Mout, g.w, g.h, g.ld, inv, upd, K, g.bs, swz)
#define MI_FB_TILED(KH)                                                                  \
    case KH:                                                                             \
        if (gauss) { if (R == 8) MI_FB_LAUNCH(true, KH, 8); else MI_FB_LAUNCH(true, KH, 4); }   \
        else { if (R == 8) MI_FB_LAUNCH(false, KH, 8); else MI_FB_LAUNCH(false, KH, 4); }       \
        break;
    switch (tuning().fb_tiled ? kh : -1) {
        MI_FB_TILED(4) MI_FB_TILED(6) MI_FB_TILED(7) MI_FB_TILED(10)   // winSize 9, 13 (the default), 15, 21
    default:
        if (gauss) hipLaunchKernelGGL(k_iterate<true>, grid, dim3(256), lds, s, M, R0, R1, flowx, flowy, Mout, g.w, g.h, g.ld, kh, inv, upd, K, g.bs);
        else hipLaunchKernelGGL(k_iterate<false>, grid, dim3(256), lds, s, M, R0, R1, flowx, flowy, Mout, g.w, g.h, g.ld, kh, inv, upd, K, g.bs);
    }
#undef MI_FB_TILED
#undef MI_FB_LAUNCH
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
