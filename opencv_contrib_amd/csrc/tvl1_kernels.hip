// Dual TV-L1 optical flow -- HIP kernels for gfx950 (MI355X, CDNA4), wave64.
//
// What the reference runs as 4 kernels + cuda::resize/multiply/setTo/merge/calcSum
// (modules/cudaoptflow/src/cuda/tvl1flow.cu:59-363, src/tvl1flow.cpp:185-382) is
// restructured here for an HBM-bound machine:
//   * estimateU + estimateDualVariables are ONE pass (64 B/px instead of 88), written as a
//     register row-streaming kernel: a wave owns a 252-px-wide column strip (dwordx4 per
//     lane), walks down its row band carrying the previous row's p12/p22 and the next
//     row's new u in VGPRs; x-neighbours come from the adjacent lane (cross-lane move), so
//     there is no LDS, no barrier and every global access is a coalesced 16-B load/store.
//   * the convergence test is evaluated on the device (per-launch control slots), so the
//     whole calc() is stream-ordered -- the reference syncs the host at every check
//     (src/tvl1flow.cpp:366-368).
//   * parameters travel as kernel arguments (no __constant__ state): handles never race.
// Compiled with -ffp-contract=off: the exact-math variants perform the same separately
// rounded binary32 operations, in the same order, as the CPU reference
// (modules/optflow/src/tvl1flow.cpp); fast-math variants use explicit fmaf/rcp.
#include "tvl1_dev.h"
#include "resize_dev.h"
#include "tvl1_warp_dev.h"
#include "tvl1_tb_dev.h"
#include <cfloat>
#include <cstdlib>
#include <climits>

namespace mi {
namespace tvl1 {

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ float4 ld4(const float *p, long long off, bool ok)
{
    return ok ? *reinterpret_cast<const float4 *>(p + off) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void st4(float *p, long long off, bool ok, const float v[4])
{
    if (ok) *reinterpret_cast<float4 *>(p + off) = make_float4(v[0], v[1], v[2], v[3]);
}
#define UNPACK4(dst, v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
__device__ __forceinline__ float lane_prev(float v) { return __shfl_up(v, 1); }   // lane n <- n-1
__device__ __forceinline__ float lane_next(float v) { return __shfl_down(v, 1); } // lane n <- n+1


// ------------------------------------------------------------------ convert / pack
__global__ __launch_bounds__(256) void k_convert(const PtrTab *tab, int type, float *I0, float *I1, Geo g)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (x >= g.w || y >= g.h) return;
    const PtrTab t = tab[b];
    const long long o = (long long)b * g.ps + (long long)y * g.ld + x;
    if (type == MI_8UC1) {
        // convertTo(CV_32F, 1.0)  cudaoptflow/src/tvl1flow.cpp:200-201
        I0[o] = (float)((const unsigned char *)t.a)[(long long)y * t.step_a + x];
        I1[o] = (float)((const unsigned char *)t.b)[(long long)y * t.step_b + x];
    } else {
        I0[o] = ((const float *)((const char *)t.a + (long long)y * t.step_a))[x] * 255.0f;
        I1[o] = ((const float *)((const char *)t.b + (long long)y * t.step_b))[x] * 255.0f;
    }
}

__global__ __launch_bounds__(256) void k_unpack_flow(const PtrTab *tab, float *u1, float *u2, Geo g)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (x >= g.w || y >= g.h) return;
    const PtrTab t = tab[b];
    const float2 f = ((const float2 *)((const char *)t.out + (long long)y * t.step_out))[x];
    const long long o = (long long)b * g.ps + (long long)y * g.ld + x;
    u1[o] = f.x;
    u2[o] = f.y;
}

__global__ __launch_bounds__(256) void k_pack_flow(const PtrTab *tab, const float *u1a, const float *u1b,
                                                   const float *u2a, const float *u2b, Geo g, CtlK ctl, int cur_host)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (x >= g.w || y >= g.h) return;
    const int cur = resolve_cur_k(ctl, b, cur_host);
    const float *u1 = cur ? u1b : u1a, *u2 = cur ? u2b : u2a;
    const PtrTab t = tab[b];
    const long long o = (long long)b * g.ps + (long long)y * g.ld + x;
    // cuda::merge -> CV_32FC2  cudaoptflow/src/tvl1flow.cpp:181-182
    ((float2 *)((char *)t.out + (long long)y * t.step_out))[x] = make_float2(u1[o], u2[o]);
}

// ------------------------------------------------------------------ median filter of the flow (CPU class only)
// cv::medianBlur(u, u, medianFiltering) before every outer iteration (optflow/src/tvl1flow.cpp:1381-1384): ksize 3 or 5,
// replicate border.  Loop control: the filter runs only while the outer loop is still active, evaluated on the device
// from the previous iteration launch's slot exactly like k_iterate does.
__device__ __forceinline__ bool resolve_active_k(const CtlK &c, int b, int cur_host, int &cur)
{
    if (!c.S) { cur = cur_host; return true; }
    if (c.q_prev < 0) { cur = 0; return true; }
    const long long sp = (long long)b * c.Q + c.q_prev;
    const int2 s = c.S[sp];
    cur = c.reset_cur ? 0 : (s.x ^ (s.y & 1));
    if (c.first_of_warp) return true;
    const double e = (double)c.E[sp] * (1.0 / ERR_FIX_SCALE);
    return (s.y & 1) && (!(s.y & 2) || e > c.thr);   // an unchecked iteration never ends the loop (error = max)
}

struct MedArgs { float *u[2][2]; float *tmp[2]; Geo g; };   // u[set][component]

template <int KS>
__global__ __launch_bounds__(256) void k_median(MedArgs A, CtlK ctl, int cur_host)
{
    constexpr int N = KS * KS, R = KS / 2;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z >> 1, c = blockIdx.z & 1;
    const int W = A.g.w, H = A.g.h, ld = A.g.ld;
    int cur;
    if (!resolve_active_k(ctl, b, cur_host, cur)) return;
    if (x >= W || y >= H) return;
    const float *S = A.u[cur][c] + (long long)b * A.g.ps;
    float v[N];
#pragma unroll
    for (int j = -R; j <= R; ++j)
#pragma unroll
        for (int i = -R; i <= R; ++i)
            v[(j + R) * KS + i + R] = S[(long long)min(max(y + j, 0), H - 1) * ld + min(max(x + i, 0), W - 1)];
    // selection of the N/2-th smallest by partial exchange sort (exact median, same value as a full sort)
#pragma unroll
    for (int i = 0; i <= N / 2; ++i)
#pragma unroll
        for (int j = i + 1; j < N; ++j) {
            const float lo = fminf(v[i], v[j]), hi = fmaxf(v[i], v[j]);
            v[i] = lo; v[j] = hi;
        }
    A.tmp[c][(long long)b * A.g.ps + (long long)y * ld + x] = v[N / 2];
}

__global__ __launch_bounds__(256) void k_median_copy(MedArgs A, CtlK ctl, int cur_host)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z >> 1, c = blockIdx.z & 1;
    int cur;
    if (!resolve_active_k(ctl, b, cur_host, cur)) return;
    if (x >= A.g.w || y >= A.g.h) return;
    const long long o = (long long)b * A.g.ps + (long long)y * A.g.ld + x;
    A.u[cur][c][o] = A.tmp[c][o];
}

// ------------------------------------------------------------------ resize (pyramid / flow upsample)
struct ResizeArgs {
    const float *src[3][2];
    float *dst[3];
    float post[3];
    int nsets, nplanes;
    Geo gs, gd;
    double scale_x, scale_y;  // CPU_REF: 1/inv_scale (double).  CUDA_COMPAT: (float)(1/f) stored as double
};

// (the per-pixel resize arithmetic lives in resize_dev.h: the Farneback kernels sample through it as well)
// PX consecutive destination pixels x ROWS rows per thread, blockIdx.z = plane + nplanes * pair.  With one row per thread the kernel
// was issue-stalled on the double-precision column coordinates (r02p: SQ WAIT_INST 0.70); amortised over 8 rows, one pixel per
// lane (coalesced dword rows, source taps of a wave within two or three cache lines) is the fastest shape.
#define RSZ_ROWS 8
template <int SEM, int PX, int ROWS>
__global__ __launch_bounds__(256) void k_resize(ResizeArgs A, CtlK ctl, int cur_host)
{
    const int dx = (blockIdx.x * 64 + (threadIdx.x & 63)) * PX;
    const int dy0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * ROWS;
    const int pl = blockIdx.z % A.nplanes, b = blockIdx.z / A.nplanes;
    if (dx >= A.gd.w || dy0 >= A.gd.h) return;
    const int cur = A.nsets == 2 ? resolve_cur_k(ctl, b, cur_host) : 0;
    const float *S = A.src[pl][cur] + (long long)b * A.gs.ps;
    const int sw = A.gs.w, sh = A.gs.h, ld = A.gs.ld;
    const float ps = A.post[pl];
    RszX X[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) X[j] = resize_xside<SEM>(min(dx + j, A.gd.w - 1), sw, A.scale_x);
    float *D = A.dst[pl] + (long long)b * A.gd.ps + dx;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int dy = dy0 + r;
        if (dy >= A.gd.h) break;
        const RszX Y = resize_yside<SEM>(dy, sh, A.scale_y);
        const float *R0 = S + (long long)Y.i0 * ld, *R1 = S + (long long)Y.i1 * ld;
        float o[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            o[j] = resize_combine<SEM>(R0, R1, X[j], Y);
            if (ps != 1.0f) o[j] = o[j] * ps;  // cuda::multiply(u, 1/scaleStep)  tvl1flow.cpp:299-300
        }
        // rows are `ld` floats apart (a multiple of 64) from a 256-B aligned base: dx % 4 == 0 is 16-B aligned; pixels past w
        // fall into the row padding
        if (PX == 4) *reinterpret_cast<float4 *>(D + (long long)dy * A.gd.ld) = make_float4(o[0], o[1], o[2], o[3]);
        else D[(long long)dy * A.gd.ld] = o[0];
    }
}

// ------------------------------------------------------------------ centered gradient
// tvl1flow.cu:59-69 == optflow/src/tvl1flow.cpp:688-770 (one-sided x0.5 at borders == clamp)
__global__ __launch_bounds__(256) void k_gradient(const float *src, float *dxp, float *dyp, Geo g)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (x >= g.w || y >= g.h) return;
    const float *S = src + (long long)b * g.ps;
    const long long r = (long long)y * g.ld;
    const float l = S[r + max(x - 1, 0)], rr = S[r + min(x + 1, g.w - 1)];
    const float u = S[(long long)max(y - 1, 0) * g.ld + x], d = S[(long long)min(y + 1, g.h - 1) * g.ld + x];
    const long long o = (long long)b * g.ps + r + x;
    dxp[o] = 0.5f * (rr - l);
    dyp[o] = 0.5f * (d - u);
}

// Same differences, written together with the image as one float4 per pixel {I1, I1x, I1y, 0}: the warp
// kernel then gathers ONE 16-B element per bicubic tap instead of three dwords (the gather is bound by the
// texture-addresser rate of ~4 lanes/clk per wave-load, rocprofv3 r01a: 48 dword gathers/px = 308 us/launch).
__global__ __launch_bounds__(256) void k_gradient_pack(const float *src, float4 *pk, Geo g)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (x >= g.w || y >= g.h) return;
    const float *S = src + (long long)b * g.ps;
    const long long r = (long long)y * g.ld;
    const float c = S[r + x];
    const float l = S[r + max(x - 1, 0)], rr = S[r + min(x + 1, g.w - 1)];
    const float u = S[(long long)max(y - 1, 0) * g.ld + x], d = S[(long long)min(y + 1, g.h - 1) * g.ld + x];
    pk[(long long)b * g.ps + r + x] = make_float4(c, 0.5f * (rr - l), 0.5f * (d - u), 0.f);
}

__global__ __launch_bounds__(256) void k_pack3(const float *a, const float *b_, const float *c, float4 *pk, Geo g)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (x >= g.w || y >= g.h) return;
    const long long o = (long long)b * g.ps + (long long)y * g.ld + x;
    pk[o] = make_float4(a[o], b_[o], c[o], 0.f);
}

// ------------------------------------------------------------------ warp
struct WarpArgs {
    const float *I0;
    const float4 *pk;   // {I1, I1x, I1y, 0} per pixel
    const float *u1[2], *u2[2];
    float *I1w, *I1wx, *I1wy, *grad, *rho;
    const float *tab;  // 32x4 cubic phase table (CPU_REF)
    Geo g;
};

__device__ __forceinline__ float bicubic_coeff_cuda(float x_)
{
    // tvl1flow.cu:89-104 (Keys a = -0.5)
    const float x = fabsf(x_);
    if (x <= 1.0f) return x * x * (1.5f * x - 2.5f) + 1.0f;
    else if (x < 2.0f) return x * (x * (-0.5f * x + 2.5f) - 4.0f) + 2.0f;
    return 0.0f;
}

// TX = pixels of one wave along x (64 / TX rows per wave): 64 = one row segment per wave; 16 = a 16 x 4 patch per wave, whose
// bicubic footprint (19 x 7 taps instead of 67 x 4) keeps more of the 16 gathers per pixel in the L1.
template <int SEM, int TX = 64>
__global__ __launch_bounds__(256) void k_warp(WarpArgs A, CtlK ctl, int cur_host)
{
    __shared__ float s_tab[128];
    if (SEM == MI_SEM_CPU_REF) {
        if (threadIdx.x < 128) s_tab[threadIdx.x] = A.tab[threadIdx.x];
        __syncthreads();
    }
    constexpr int TY = 64 / TX;   // rows per wave
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    const int x = blockIdx.x * TX + (lane_ % TX);
    const int y = blockIdx.y * (4 * TY) + wave_ * TY + lane_ / TX;
    const int b = blockIdx.z;
    const int W = A.g.w, H = A.g.h, ld = A.g.ld;
    if (x >= W || y >= H) return;
    const int cur = resolve_cur_k(ctl, b, cur_host);
    const long long pb = (long long)b * A.g.ps;
    const long long o = pb + (long long)y * ld + x;
    const float u1v = A.u1[cur][o], u2v = A.u2[cur][o];
    const float4 *P = A.pk + pb;
    float v0, v1, v2;
    if (SEM == MI_SEM_CPU_REF) {
        // buildFlowMap + cv::remap(INTER_CUBIC, BORDER_CONSTANT 0):
        // optflow/src/tvl1flow.cpp:650-666,1371-1374; map quantised to 1/32 px.
        const float mx = (float)x + u1v, my = (float)y + u2v;
        const int qx = __float2int_rn(mx * 32.0f), qy = __float2int_rn(my * 32.0f);
        const int sx = min(max(qx >> 5, -32768), 32767) - 1;
        const int sy = min(max(qy >> 5, -32768), 32767) - 1;
        const float *wx = s_tab + (qx & 31) * 4, *wy = s_tab + (qy & 31) * 4;
        float w[16];
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) w[k1 * 4 + k2] = wy[k1] * wx[k2];
        if ((unsigned)sx < (unsigned)max(W - 3, 0) && (unsigned)sy < (unsigned)max(H - 3, 0)) {
            const long long base = (long long)sy * ld + sx;
            float s0, s1, s2;
            {
                const float4 *S = P + base;
                float4 a = S[0], b4 = S[1], c = S[2], d = S[3];
                s0 = a.x * w[0] + b4.x * w[1] + c.x * w[2] + d.x * w[3];
                s1 = a.y * w[0] + b4.y * w[1] + c.y * w[2] + d.y * w[3];
                s2 = a.z * w[0] + b4.z * w[1] + c.z * w[2] + d.z * w[3];
                S += ld; a = S[0]; b4 = S[1]; c = S[2]; d = S[3];
                s0 += a.x * w[4] + b4.x * w[5] + c.x * w[6] + d.x * w[7];
                s1 += a.y * w[4] + b4.y * w[5] + c.y * w[6] + d.y * w[7];
                s2 += a.z * w[4] + b4.z * w[5] + c.z * w[6] + d.z * w[7];
                S += ld; a = S[0]; b4 = S[1]; c = S[2]; d = S[3];
                s0 += a.x * w[8] + b4.x * w[9] + c.x * w[10] + d.x * w[11];
                s1 += a.y * w[8] + b4.y * w[9] + c.y * w[10] + d.y * w[11];
                s2 += a.z * w[8] + b4.z * w[9] + c.z * w[10] + d.z * w[11];
                S += ld; a = S[0]; b4 = S[1]; c = S[2]; d = S[3];
                s0 += a.x * w[12] + b4.x * w[13] + c.x * w[14] + d.x * w[15];
                s1 += a.y * w[12] + b4.y * w[13] + c.y * w[14] + d.y * w[15];
                s2 += a.z * w[12] + b4.z * w[13] + c.z * w[14] + d.z * w[15];
            }
            v0 = s0; v1 = s1; v2 = s2;
        } else if (sx >= W || sx + 4 <= 0 || sy >= H || sy + 4 <= 0) {
            v0 = v1 = v2 = 0.f;
        } else {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            for (int i = 0; i < 4; ++i) {
                const int yi = sy + i;
                if (yi < 0 || yi >= H) continue;
                for (int j = 0; j < 4; ++j) {
                    const int xj = sx + j;
                    if (xj < 0 || xj >= W) continue;
                    const float4 t = P[(long long)yi * ld + xj];
                    s0 += (t.x - 0.f) * w[i * 4 + j];
                    s1 += (t.y - 0.f) * w[i * 4 + j];
                    s2 += (t.z - 0.f) * w[i * 4 + j];
                }
            }
            v0 = s0; v1 = s1; v2 = s2;
        }
    } else {
        // tvl1flow.cu:106-149: normalised bicubic, point-sampled clamp-addressed reads
        const float wxp = (float)x + u1v, wyp = (float)y + u2v;
        const int xmin = (int)ceilf(wxp - 2.0f), xmax = (int)floorf(wxp + 2.0f);
        const int ymin = (int)ceilf(wyp - 2.0f), ymax = (int)floorf(wyp + 2.0f);
        float sum = 0.f, sumx = 0.f, sumy = 0.f, wsum = 0.f;
        for (int cy = ymin; cy <= ymax; ++cy)
            for (int cx = xmin; cx <= xmax; ++cx) {
                const float wgt = bicubic_coeff_cuda(wxp - (float)cx) * bicubic_coeff_cuda(wyp - (float)cy);
                const float4 t = P[(long long)min(max(cy, 0), H - 1) * ld + min(max(cx, 0), W - 1)];
                sum += wgt * t.x;
                sumx += wgt * t.y;
                sumy += wgt * t.z;
                wsum += wgt;
            }
        const float coeff = 1.0f / wsum;
        v0 = sum * coeff; v1 = sumx * coeff; v2 = sumy * coeff;
    }
    if (A.I1w) A.I1w[o] = v0;
    A.I1wx[o] = v1;
    A.I1wy[o] = v2;
    // calcGradRho  optflow/src/tvl1flow.cpp:918-944 == tvl1flow.cu:151-163
    const float Ix2 = v1 * v1, Iy2 = v2 * v2;
    A.grad[o] = Ix2 + Iy2;
    A.rho[o] = (v0 - v1 * u1v - v2 * u2v - A.I0[o]);
}

// ------------------------------------------------------------------ fused iteration
// Per-pixel math.  EXACT: the CPU reference's operations (optflow/src/tvl1flow.cpp:989-1041
// estimateV, :857-899 divergence, :1096-1112 estimateU, :1140-1181 dual update with hypot in
// double) in reference order.  !EXACT: same formulas with fmaf + v_rcp/v_sqrt approximations.
template <bool EXACT, bool GAMMA = false>
__device__ __forceinline__ void px_update_u(float ix, float iy, float g, float rc, float u1, float u2,
                                            float div1, float div2, float l_t, float theta,
                                            float &u1n, float &u2n, float &err, float gamma = 0.f, float u3 = 0.f,
                                            float div3 = 0.f, float *u3n = nullptr, bool err_u3 = false)
{
    float rho, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    if (EXACT) rho = rc + (ix * u1 + iy * u2);
    else rho = rc + fmaf(ix, u1, iy * u2);
    if (GAMMA) rho = rho + gamma * u3;   // optflow tvl1flow.cpp:1011
    const float ltg = l_t * g;
    if (rho < -ltg) { d1 = l_t * ix; d2 = l_t * iy; if (GAMMA) d3 = l_t * gamma; }
    else if (rho > ltg) { d1 = -l_t * ix; d2 = -l_t * iy; if (GAMMA) d3 = -l_t * gamma; }
    else if (g > FLT_EPSILON) {
        const float fi = EXACT ? (-rho / g) : (-rho * __builtin_amdgcn_rcpf(g));
        d1 = fi * ix; d2 = fi * iy; if (GAMMA) d3 = fi * gamma;
    }
    if (EXACT) {
        const float v1 = u1 + d1, v2 = u2 + d2;
        u1n = v1 + theta * div1;
        u2n = v2 + theta * div2;
    } else {
        u1n = fmaf(theta, div1, u1 + d1);
        u2n = fmaf(theta, div2, u2 + d2);
    }
    const float e1 = u1n - u1, e2 = u2n - u2;
    err = EXACT ? (e1 * e1 + e2 * e2) : fmaf(e1, e1, e2 * e2);
    if (GAMMA) {
        const float v3 = u3 + d3;
        *u3n = v3 + theta * div3;
        const float e3 = *u3n - u3;
        if (err_u3) err = err + e3 * e3;   // :1110
    }
}

template <bool EXACT>
__device__ __forceinline__ void px_update_p(float ux, float uy, float taut, float &pa, float &pb)
{
    if (EXACT) {
        const float gn = (float)sqrt((double)ux * (double)ux + (double)uy * (double)uy);
        const float ng = 1.0f + taut * gn;
        pa = (pa + taut * ux) / ng;
        pb = (pb + taut * uy) / ng;
    } else {
        const float gn = __builtin_sqrtf(fmaf(ux, ux, uy * uy));
        const float r = __builtin_amdgcn_rcpf(fmaf(taut, gn, 1.0f));
        pa = fmaf(taut, ux, pa) * r;
        pb = fmaf(taut, uy, pb) * r;
    }
}

struct IterArgs {
    IterPlanes pl;
    Geo g;
    float l_t, theta, taut;
    int rows_per_wave;
};

#define STRIP_W 252  // 63 lanes x 4 px own results; lane 63 only supplies u_new(x+1) to lane 62


struct RowIn {
    float ix[4], iy[4], g[4], rc[4], u1[4], u2[4], p11[4], p12[4], p21[4], p22[4];
    float u3[4], p31[4], p32[4];   // gamma != 0 only
};


template <bool PZ, bool GAMMA = false>
__device__ __forceinline__ void load_row(RowIn &r, const IterArgs &A, const float *const u[3], const float *const p[6],
                                         long long off, bool ok)
{
    if (GAMMA) {
        float4 t3 = ld4(u[2], off, ok); UNPACK4(r.u3, t3);
        if (!PZ) {
            t3 = ld4(p[4], off, ok); UNPACK4(r.p31, t3);
            t3 = ld4(p[5], off, ok); UNPACK4(r.p32, t3);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) r.p31[j] = r.p32[j] = 0.f;
        }
    }
    float4 t;
    t = ld4(A.pl.ix, off, ok); UNPACK4(r.ix, t);
    t = ld4(A.pl.iy, off, ok); UNPACK4(r.iy, t);
    t = ld4(A.pl.g, off, ok);  UNPACK4(r.g, t);
    t = ld4(A.pl.rc, off, ok); UNPACK4(r.rc, t);
    t = ld4(u[0], off, ok);    UNPACK4(r.u1, t);
    t = ld4(u[1], off, ok);    UNPACK4(r.u2, t);
    if (!PZ) {
        t = ld4(p[0], off, ok); UNPACK4(r.p11, t);
        t = ld4(p[1], off, ok); UNPACK4(r.p12, t);
        t = ld4(p[2], off, ok); UNPACK4(r.p21, t);
        t = ld4(p[3], off, ok); UNPACK4(r.p22, t);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r.p11[j] = r.p12[j] = r.p21[j] = r.p22[j] = 0.f;
    }
}

// u_new of one row.  up12/up22 = p12,p22 of the row above (unused when y == 0);
// l11/l21 = p11,p21 at x-1 of this lane's first pixel.
template <bool EXACT, bool GAMMA = false>
__device__ __forceinline__ void row_update_u(const RowIn &r, const float up12[4], const float up22[4],
                                             float l11, float l21, int xb, int y, float l_t, float theta,
                                             float u1n[4], float u2n[4], float e[4], const float up32[4] = nullptr,
                                             float l31 = 0.f, float gamma = 0.f, bool err_u3 = false, float u3n[4] = nullptr)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = xb + j;
        const float p11l = j ? r.p11[j - 1] : l11;
        const float p21l = j ? r.p21[j - 1] : l21;
        float d1, d2;
        // divergence with first-row/column cases, optflow/src/tvl1flow.cpp:857-899
        if (y > 0) {
            if (x > 0) {
                d1 = (r.p11[j] - p11l) + (r.p12[j] - up12[j]);
                d2 = (r.p21[j] - p21l) + (r.p22[j] - up22[j]);
            } else {
                d1 = r.p11[j] + r.p12[j] - up12[j];
                d2 = r.p21[j] + r.p22[j] - up22[j];
            }
        } else {
            if (x > 0) {
                d1 = r.p11[j] - p11l + r.p12[j];
                d2 = r.p21[j] - p21l + r.p22[j];
            } else {
                d1 = r.p11[j] + r.p12[j];
                d2 = r.p21[j] + r.p22[j];
            }
        }
        if (GAMMA) {
            const float p31l = j ? r.p31[j - 1] : l31;
            float d3;
            if (y > 0) d3 = x > 0 ? (r.p31[j] - p31l) + (r.p32[j] - up32[j]) : r.p31[j] + r.p32[j] - up32[j];
            else d3 = x > 0 ? r.p31[j] - p31l + r.p32[j] : r.p31[j] + r.p32[j];
            px_update_u<EXACT, true>(r.ix[j], r.iy[j], r.g[j], r.rc[j], r.u1[j], r.u2[j], d1, d2, l_t, theta,
                                     u1n[j], u2n[j], e[j], gamma, r.u3[j], d3, &u3n[j], err_u3);
        } else {
            px_update_u<EXACT>(r.ix[j], r.iy[j], r.g[j], r.rc[j], r.u1[j], r.u2[j], d1, d2, l_t, theta,
                               u1n[j], u2n[j], e[j]);
        }
    }
}

template <bool EXACT, bool PZ, bool CHECK, bool GAMMA = false>
__global__ __launch_bounds__(256) void k_iterate(IterArgs A, CtlK ctl, int cur_host)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strip = blockIdx.x, b = blockIdx.z;
    const int band = blockIdx.y * 4 + wave;
    const int W = A.g.w, H = A.g.h, ld = A.g.ld;

    int cur = cur_host;
    bool calc_err = true;
    if (CHECK) {
        // loop control of procOneScale (optflow/src/tvl1flow.cpp:1376-1390), evaluated per launch
        // (cv::cuda schedule, cudaoptflow/src/tvl1flow.cpp:357-377, when ctl.sched)
        int cur_in = 0, active = 1;
        double prev_in = 0.0;   // prevError as this iteration sees it
        if (ctl.q_prev >= 0) {
            const long long sp = (long long)b * ctl.Q + ctl.q_prev;
            const int2 s = ctl.S[sp];
            cur_in = ctl.reset_cur ? 0 : (s.x ^ (s.y & 1));
            if (!ctl.first_of_warp) {
                const double e = (double)ctl.E[sp] * (1.0 / ERR_FIX_SCALE);
                const bool checked = (s.y & 2) != 0;
                active = (s.y & 1) && (!checked || e > ctl.thr);
                if (ctl.sched) prev_in = checked ? e : ctl.P[sp] - ctl.thr;
            }
        }
        calc_err = !ctl.sched || ((ctl.n & 1) && prev_in < ctl.thr);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            ctl.S[(long long)b * ctl.Q + ctl.q] = make_int2(cur_in, active | (calc_err ? 2 : 0));
            if (ctl.sched) ctl.P[(long long)b * ctl.Q + ctl.q] = prev_in;
        }
        if (!active) return;
        cur = cur_in;
    }

    const int y0 = band * A.rows_per_wave;
    if (y0 >= H) return;
    const int y1 = min(y0 + A.rows_per_wave, H);
    const int x0 = strip * STRIP_W;
    const int xb = x0 + lane * 4;
    const bool ok = xb < W;
    const long long pb = (long long)b * A.g.ps;

    const float *uin[3] = {A.pl.u[cur][0] + pb, A.pl.u[cur][1] + pb, GAMMA ? A.pl.u[cur][2] + pb : nullptr};
    const float *pin[6] = {A.pl.p[cur][0] + pb, A.pl.p[cur][1] + pb, A.pl.p[cur][2] + pb, A.pl.p[cur][3] + pb,
                           GAMMA ? A.pl.p[cur][4] + pb : nullptr, GAMMA ? A.pl.p[cur][5] + pb : nullptr};
    float *uout[3] = {A.pl.u[cur ^ 1][0] + pb, A.pl.u[cur ^ 1][1] + pb, GAMMA ? A.pl.u[cur ^ 1][2] + pb : nullptr};
    float *pout[6] = {A.pl.p[cur ^ 1][0] + pb, A.pl.p[cur ^ 1][1] + pb, A.pl.p[cur ^ 1][2] + pb, A.pl.p[cur ^ 1][3] + pb,
                      GAMMA ? A.pl.p[cur ^ 1][4] + pb : nullptr, GAMMA ? A.pl.p[cur ^ 1][5] + pb : nullptr};
    const float gamma = A.pl.gamma;
    const bool err_u3 = A.pl.err_u3 != 0;
    IterArgs B = A;
    B.pl.ix += pb; B.pl.iy += pb; B.pl.g += pb; B.pl.rc += pb;

    // p12/p22 of row y0-1
    float up12[4] = {0, 0, 0, 0}, up22[4] = {0, 0, 0, 0}, up32[4] = {0, 0, 0, 0};
    if (!PZ && y0 > 0) {
        const long long off = (long long)(y0 - 1) * ld + xb;
        float4 t = ld4(pin[1], off, ok); UNPACK4(up12, t);
        t = ld4(pin[3], off, ok); UNPACK4(up22, t);
        if (GAMMA) { t = ld4(pin[5], off, ok); UNPACK4(up32, t); }
    }

    RowIn r;
    float u1c[4], u2c[4], ec[4], u3c[4] = {0, 0, 0, 0};
    {
        const long long off = (long long)y0 * ld + xb;
        load_row<PZ, GAMMA>(r, B, uin, pin, off, ok);
        float l11 = lane_prev(r.p11[3]), l21 = lane_prev(r.p21[3]), l31 = GAMMA ? lane_prev(r.p31[3]) : 0.f;
        if (!PZ && lane == 0 && x0 > 0) {
            l11 = pin[0][(long long)y0 * ld + x0 - 1];
            l21 = pin[2][(long long)y0 * ld + x0 - 1];
            if (GAMMA) l31 = pin[4][(long long)y0 * ld + x0 - 1];
        }
        row_update_u<EXACT, GAMMA>(r, up12, up22, l11, l21, xb, y0, A.l_t, A.theta, u1c, u2c, ec, up32, l31, gamma, err_u3, u3c);
    }
    float errsum = 0.f;
    const bool own = lane < 63;

    for (int y = y0; y < y1; ++y) {
        // current row: r (inputs), u1c/u2c (new u), ec (error terms)
        float p11c[4], p12c[4], p21c[4], p22c[4], p31c[4], p32c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { p11c[j] = r.p11[j]; p12c[j] = r.p12[j]; p21c[j] = r.p21[j]; p22c[j] = r.p22[j];
                                      p31c[j] = GAMMA ? r.p31[j] : 0.f; p32c[j] = GAMMA ? r.p32[j] : 0.f; }
        float u1d[4], u2d[4], ed[4], u3d[4] = {0, 0, 0, 0};
        const bool has_next = (y + 1 < H);
        if (has_next) {
            const long long off = (long long)(y + 1) * ld + xb;
            // p12,p22 of the current row become the "row above" of the next row
            float c12[4], c22[4], c32[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { c12[j] = r.p12[j]; c22[j] = r.p22[j]; c32[j] = GAMMA ? r.p32[j] : 0.f; }
            load_row<PZ, GAMMA>(r, B, uin, pin, off, ok);
            float l11 = lane_prev(r.p11[3]), l21 = lane_prev(r.p21[3]), l31 = GAMMA ? lane_prev(r.p31[3]) : 0.f;
            if (!PZ && lane == 0 && x0 > 0) {
                l11 = pin[0][(long long)(y + 1) * ld + x0 - 1];
                l21 = pin[2][(long long)(y + 1) * ld + x0 - 1];
                if (GAMMA) l31 = pin[4][(long long)(y + 1) * ld + x0 - 1];
            }
            row_update_u<EXACT, GAMMA>(r, c12, c22, l11, l21, xb, y + 1, A.l_t, A.theta, u1d, u2d, ed, c32, l31, gamma, err_u3, u3d);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { u1d[j] = u1c[j]; u2d[j] = u2c[j]; u3d[j] = u3c[j]; ed[j] = 0.f; }
        }
        // forward differences (optflow/src/tvl1flow.cpp:775-840: 0 at the last row/col)
        const float r1 = lane_next(u1c[0]), r2 = lane_next(u2c[0]), r3 = GAMMA ? lane_next(u3c[0]) : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = xb + j;
            const float n1 = j < 3 ? u1c[j + 1] : r1;
            const float n2 = j < 3 ? u2c[j + 1] : r2;
            if (GAMMA) {
                const float n3 = j < 3 ? u3c[j + 1] : r3;
                const float u3x = (x + 1 < W) ? n3 - u3c[j] : 0.f;
                const float u3y = has_next ? u3d[j] - u3c[j] : 0.f;
                px_update_p<EXACT>(u3x, u3y, A.taut, p31c[j], p32c[j]);
            }
            const float u1x = (x + 1 < W) ? n1 - u1c[j] : 0.f;
            const float u2x = (x + 1 < W) ? n2 - u2c[j] : 0.f;
            const float u1y = has_next ? u1d[j] - u1c[j] : 0.f;
            const float u2y = has_next ? u2d[j] - u2c[j] : 0.f;
            px_update_p<EXACT>(u1x, u1y, A.taut, p11c[j], p12c[j]);
            px_update_p<EXACT>(u2x, u2y, A.taut, p21c[j], p22c[j]);
            if (CHECK && own && x < W) errsum += ec[j];
        }
        const long long off = (long long)y * ld + xb;
        const bool st = ok && own;
        st4(uout[0], off, st, u1c);
        st4(uout[1], off, st, u2c);
        st4(pout[0], off, st, p11c);
        st4(pout[1], off, st, p12c);
        st4(pout[2], off, st, p21c);
        st4(pout[3], off, st, p22c);
        if (GAMMA) {
            st4(uout[2], off, st, u3c);
            st4(pout[4], off, st, p31c);
            st4(pout[5], off, st, p32c);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { u1c[j] = u1d[j]; u2c[j] = u2d[j]; u3c[j] = u3d[j]; ec[j] = ed[j]; }
    }

    if (CHECK && calc_err) {
        double s = (double)errsum;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            const unsigned long long f = (unsigned long long)(s * ERR_FIX_SCALE + 0.5);
            atomicAdd(&ctl.E[(long long)b * ctl.Q + ctl.q], f);
        }
    }
}

// ------------------------------------------------------------------ host launchers
static inline dim3 grid2d(const Geo &g, int z) { return dim3(div_up(g.w, 64), div_up(g.h, 4), z); }

int convert(const PtrTab *tab, int type, float *I0, float *I1, const Geo &g, hipStream_t s)
{
    hipLaunchKernelGGL(k_convert, grid2d(g, g.batch), dim3(256), 0, s, tab, type, I0, I1, g);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

// Four planes of n floats each (n a multiple of 4, 16-B aligned bases) set to zero in ONE launch: the dual variable at the start of a
// scale on the convergence-checked path.  (Four hipMemsetAsync calls cost four launches; hipMemset2DAsync runs a fill kernel that
// takes 64 us for 4 x 8 MB on MI355X -- r10b.)
__global__ __launch_bounds__(256) void k_zero4(float4 *a, float4 *b, float4 *c, float4 *d, long long n4)
{
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        a[i] = z; b[i] = z; c[i] = z; d[i] = z;
    }
}
int zero_planes4(float *const p[4], size_t n, hipStream_t s)
{
    if ((n & 3) || (((uintptr_t)p[0] | (uintptr_t)p[1] | (uintptr_t)p[2] | (uintptr_t)p[3]) & 15)) {
        for (int j = 0; j < 4; ++j) MI_HIP_TRY(hipMemsetAsync(p[j], 0, sizeof(float) * n, s));
        return MI_OK;
    }
    const long long n4 = (long long)(n / 4);
    const int blocks = (int)std::min<long long>((n4 + 255) / 256, 2048);
    hipLaunchKernelGGL(k_zero4, dim3(std::max(blocks, 1)), dim3(256), 0, s, (float4 *)p[0], (float4 *)p[1], (float4 *)p[2], (float4 *)p[3], n4);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int unpack_flow(const PtrTab *tab, float *u1, float *u2, const Geo &g, hipStream_t s)
{
    hipLaunchKernelGGL(k_unpack_flow, grid2d(g, g.batch), dim3(256), 0, s, tab, u1, u2, g);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int pack_flow(const PtrTab *tab, const float *u1[2], const float *u2[2], const Geo &g, const Ctl *ctl,
              int cur_host, hipStream_t s)
{
    hipLaunchKernelGGL(k_pack_flow, grid2d(g, g.batch), dim3(256), 0, s, tab, u1[0], u1[1], u2[0], u2[1], g,
                       make_ctlk(ctl), cur_host);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int resize(int semantics, int nplanes, const float *const src[3][2], int src_sets, float *const dst[3],
           const Geo &gs, const Geo &gd, double inv_scale_x, double inv_scale_y, const float post_scale[3],
           const Ctl *ctl, int cur_host, hipStream_t s)
{
    ResizeArgs A;
    memset(&A, 0, sizeof(A));
    for (int i = 0; i < 3; ++i) {
        A.src[i][0] = i < nplanes ? src[i][0] : nullptr;
        A.src[i][1] = i < nplanes ? src[i][src_sets == 2 ? 1 : 0] : nullptr;
        A.dst[i] = i < nplanes ? dst[i] : nullptr;
        A.post[i] = i < nplanes ? post_scale[i] : 1.f;
    }
    A.nsets = src_sets;
    A.nplanes = nplanes;
    A.gs = gs;
    A.gd = gd;
    const CtlK ck = make_ctlk(ctl);
#define MI_RSZ(SEMV, PX, ROWS) hipLaunchKernelGGL((k_resize<SEMV, PX, ROWS>), dim3(div_up(gd.w, 64 * PX), div_up(gd.h, 4 * ROWS), nplanes * gd.batch), dim3(256), 0, s, A, ck, cur_host)
    // r02w at 1080p x 16 (total of the resize launches of a calc): 4 px x 4 rows 4.75 | 4 x 8 5.06 | 1 x 8 3.72 | 1 x 16 3.67 ms
    if (semantics == MI_SEM_CPU_REF) {
        A.scale_x = 1.0 / inv_scale_x;
        A.scale_y = 1.0 / inv_scale_y;
        MI_RSZ(MI_SEM_CPU_REF, 1, RSZ_ROWS);
    } else {
        A.scale_x = (double)(float)(1.0 / inv_scale_x);  // cudawarping/src/resize.cpp:107
        A.scale_y = (double)(float)(1.0 / inv_scale_y);
        MI_RSZ(MI_SEM_CUDA_COMPAT, 1, RSZ_ROWS);
    }
#undef MI_RSZ
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int median_flow(int ksize, float *const u1[2], float *const u2[2], float *tmp1, float *tmp2, const Geo &g, const Ctl *ctl, int cur_host,
                hipStream_t s)
{
    MedArgs A;
    A.u[0][0] = u1[0]; A.u[1][0] = u1[1]; A.u[0][1] = u2[0]; A.u[1][1] = u2[1];
    A.tmp[0] = tmp1; A.tmp[1] = tmp2; A.g = g;
    const CtlK ck = make_ctlk(ctl);
    const dim3 grid(div_up(g.w, 64), div_up(g.h, 4), g.batch * 2);
    if (ksize == 5) hipLaunchKernelGGL(k_median<5>, grid, dim3(256), 0, s, A, ck, cur_host);
    else if (ksize == 3) hipLaunchKernelGGL(k_median<3>, grid, dim3(256), 0, s, A, ck, cur_host);
    else { set_error("medianFiltering must be 1 (off), 3 or 5 for CV_32F (cv::medianBlur)"); return MI_ERR_BAD_ARG; }
    hipLaunchKernelGGL(k_median_copy, grid, dim3(256), 0, s, A, ck, cur_host);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int gradient(const float *src, float *dx, float *dy, const Geo &g, hipStream_t s)
{
    hipLaunchKernelGGL(k_gradient, grid2d(g, g.batch), dim3(256), 0, s, src, dx, dy, g);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int gradient_pack(const float *src, float *pk, const Geo &g, hipStream_t s)
{
    hipLaunchKernelGGL(k_gradient_pack, grid2d(g, g.batch), dim3(256), 0, s, src, reinterpret_cast<float4 *>(pk), g);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int pack3(const float *a, const float *b, const float *c, float *pk, const Geo &g, hipStream_t s)
{
    hipLaunchKernelGGL(k_pack3, grid2d(g, g.batch), dim3(256), 0, s, a, b, c, reinterpret_cast<float4 *>(pk), g);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int warp(int semantics, const float *I0, const float *pk, const float *u1[2], const float *u2[2], float *I1w,
         float *I1wx, float *I1wy, float *grad, float *rho, const float *cubic_tab_dev, const Geo &g, const Ctl *ctl,
         int cur_host, hipStream_t s)
{
    // Packed-plane gather: one float4 {I1, I1x, I1y, 0} per bicubic tap.  Used by the stage-level entry point when the caller
    // supplies its own derivative planes, and by calc() under MIFLOW_WARP=pk (the round-1 kernel, kept for A/B runs); calc()
    // otherwise runs warp_fused (tvl1_warp_kernels.hip), which needs half the gathered bytes.
    WarpArgs A;
    A.I0 = I0; A.pk = reinterpret_cast<const float4 *>(pk);
    A.u1[0] = u1[0]; A.u1[1] = u1[1]; A.u2[0] = u2[0]; A.u2[1] = u2[1];
    A.I1w = I1w; A.I1wx = I1wx; A.I1wy = I1wy; A.grad = grad; A.rho = rho;
    A.tab = cubic_tab_dev;
    A.g = g;
    const CtlK ck = make_ctlk(ctl);
    if (semantics == MI_SEM_CPU_REF) {
#ifdef MIFLOW_EXPERIMENTS
        const int tile = warp_tile();   // r01u: 32 x 2 patch per wave +2 % on the bench (64: 881, 32: 902, 16: 878 pairs/s)
        if (tile == 16)
            hipLaunchKernelGGL((k_warp<MI_SEM_CPU_REF, 16>), dim3(div_up(g.w, 16), div_up(g.h, 16), g.batch), dim3(256), 0, s, A, ck, cur_host);
        else if (tile == 64)
            hipLaunchKernelGGL((k_warp<MI_SEM_CPU_REF, 64>), grid2d(g, g.batch), dim3(256), 0, s, A, ck, cur_host);
        else
#endif
            hipLaunchKernelGGL((k_warp<MI_SEM_CPU_REF, 32>), dim3(div_up(g.w, 32), div_up(g.h, 8), g.batch), dim3(256), 0, s, A, ck, cur_host);
    } else
        hipLaunchKernelGGL(k_warp<MI_SEM_CUDA_COMPAT>, grid2d(g, g.batch), dim3(256), 0, s, A, ck, cur_host);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

static int pick_rows_per_wave(const Geo &g)
{
    // enough waves to fill 256 CUs x >=8 waves while keeping the 1-row halo overhead small
    const int strips = div_up(g.w, STRIP_W);
    int R = 32;
    while (R > 8 && (long long)strips * div_up(g.h, R) * g.batch < 4096) R >>= 1;
    return R;
}

template <bool EXACT, bool PZ>
static void launch_iter(const IterArgs &A, const dim3 grid, const Ctl *ctl, int cur_host, hipStream_t s)
{
    const CtlK ck = make_ctlk(ctl);
    const bool check = ctl && ctl->S;
    if (A.pl.gamma != 0.f) {   // 3-channel model (u3, p31, p32): tvl1flow.cu:233-236,266-274,336-344
        if (check) hipLaunchKernelGGL((k_iterate<EXACT, PZ, true, true>), grid, dim3(256), 0, s, A, ck, cur_host);
        else hipLaunchKernelGGL((k_iterate<EXACT, PZ, false, true>), grid, dim3(256), 0, s, A, ck, cur_host);
        return;
    }
    if (check)
        hipLaunchKernelGGL((k_iterate<EXACT, PZ, true>), grid, dim3(256), 0, s, A, ck, cur_host);
    else
        hipLaunchKernelGGL((k_iterate<EXACT, PZ, false>), grid, dim3(256), 0, s, A, ck, cur_host);
}

int iterate(bool exact, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut, bool p_zero,
            const Ctl *ctl, int cur_host, hipStream_t s)
{
    IterArgs A;
    A.pl = pl;
    A.g = g;
    A.l_t = l_t; A.theta = theta; A.taut = taut;
    A.rows_per_wave = pick_rows_per_wave(g);
    const dim3 grid(div_up(g.w, STRIP_W), div_up(div_up(g.h, A.rows_per_wave), 4), g.batch);
    if (exact) { if (p_zero) launch_iter<true, true>(A, grid, ctl, cur_host, s); else launch_iter<true, false>(A, grid, ctl, cur_host, s); }
    else       { if (p_zero) launch_iter<false, true>(A, grid, ctl, cur_host, s); else launch_iter<false, false>(A, grid, ctl, cur_host, s); }
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

void host_cubic_table(float tab[128])
{
    // interpolateCubic / initInterTab1D of cv::remap (main repo imgproc/imgwarp.cpp), A = -0.75
    const float A = -0.75f, scale = 1.f / 32;
    for (int i = 0; i < 32; ++i) {
        const float x = (float)i * scale;
        float *c = tab + i * 4;
        c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
    }
}

}  // namespace tvl1
}  // namespace mi
