// Dual TV-L1 -- warpBackward with the centred gradient of I1 derived IN the kernel (gfx950, wave64).
//
// The reference gathers three planes per bicubic tap: I1 and its centred differences I1x, I1y
// (cudaoptflow/src/cuda/tvl1flow.cu:106-164; CPU class: 3 x cv::remap, optflow/src/tvl1flow.cpp:1371-1374).
// The earlier kernel here (k_warp, tvl1_kernels.hip) packed {I1, I1x, I1y, 0} into one float4 per pixel and
// gathered 16 x 16 B per output pixel; it ran at the rate of the vector-memory data path (64 B/clk/CU), i.e. it was
// bound by the BYTES gathered.  I1x / I1y are 0.5f * (I1[x+1] - I1[x-1]) and 0.5f * (I1[y+1] - I1[y-1])
// (tvl1flow.cu:59-69 == optflow/src/tvl1flow.cpp:688-770), so every tap value of all three planes follows from a
// 6 x 6 window of I1 without its corners: four rows of six consecutive floats and two rows of four -- 128 B per
// pixel, read as 4-byte-aligned dwordx4 / dwordx2 loads (gfx950 allows them), instead of 256 B, and no packed
// plane to build per level.  The differences are the same single rounded subtraction and multiplication that
// centeredGradient performs, so the tap values -- and with the same accumulation order, the results -- are
// bit-identical to gathering stored gradient planes.
//
// Lanes whose window touches the image border take the per-tap path, where the neighbours of a tap are clamped
// exactly like centeredGradient clamps them (and, for MI_SEM_CUDA_COMPAT, the tap itself like the clamp-addressed
// texture, cudev/ptr2d/texture.hpp:228-232).
#include "tvl1_dev.h"
#include "tvl1_warp_dev.h"
#include "tvl1_warp_px.h"
#include "resize_dev.h"

namespace mi {
namespace tvl1 {

// A wave owns NP consecutive TX x TY patches of a row band.  The flow and I0 of patch i + 1 are requested BEFORE the window
// gathers of patch i are issued, so the two dependent memory phases of a pixel (flow -> addresses -> window) overlap across
// patches: with one patch per wave the kernel ran at the latency bound of 2 phases x 2048 pixels in flight per CU (r02b).
template <int SEM, int TX, int NP, bool FAST, bool UP = false>
__global__ __launch_bounds__(256) void k_warp6(Warp6Args A, CtlK ctl, int cur_host)
{
    __shared__ float s_tab[128];
    if (SEM == MI_SEM_CPU_REF) {
        if (threadIdx.x < 128) s_tab[threadIdx.x] = A.tab[threadIdx.x];
        __syncthreads();
    }
    constexpr int TY = 64 / TX;   // rows per wave: a TX x TY patch per wave keeps the gather footprint compact
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    // XCD-contiguous tile order (round 4): the 6 x 6 windows of vertically neighbouring tiles share most of their rows of I1; dealt
    // round-robin over the 8 XCDs in launch order they sat behind different L2s and every tile's halo rows came from HBM again
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (A.swz) {
        const unsigned nwg = gridDim.x * gridDim.y, orig = by * gridDim.x + bx;   // within the pair's plane (see k_iterate_tile)
        const unsigned xcd = orig & 7u, qq = nwg >> 3, rr = nwg & 7u;
        const unsigned lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
        by = lid / gridDim.x; bx = lid - by * gridDim.x;
    }
    const int x0 = (int)bx * (TX * NP) + (lane_ % TX);
    const int y = (int)by * (4 * TY) + wave_ * TY + lane_ / TX;
    const int b = (int)bz;
    const int W = A.g.w, H = A.g.h, ld = A.g.ld;
    if (x0 >= W || y >= H) return;
    if (!warp_gate_k(ctl, b)) return;
    const int cur = resolve_cur_k(ctl, b, cur_host);
    const long long pb = (long long)b * A.g.ps;
    const long long orow = pb + (long long)y * ld;
    const float *U1 = A.u1[cur] + orow, *U2 = A.u2[cur] + orow, *I0r = A.I0 + orow;
    const float *P = A.I1 + pb;
    // UP: the flow of pixel (xx, y) = the coarser scale's flow zoomed with k_resize's arithmetic x post (the same bits as the plane a
    // resize launch would have written); it is stored for the iteration pass
    const float *C1 = nullptr, *C2 = nullptr;
    RszX Y;
    if (UP) {
        C1 = A.up.u1c + (long long)b * A.up.cps; C2 = A.up.u2c + (long long)b * A.up.cps;
        Y = resize_yside<SEM>(y, A.up.ch, A.up.scy);
    }
    const auto flow_at = [&](int xx, float &a, float &c) {
        if (!UP) { a = U1[xx]; c = U2[xx]; return; }
        const RszX X = resize_xside<SEM>(xx, A.up.cw, A.up.scx);
        a = resize_combine<SEM>(C1 + (long long)Y.i0 * A.up.cld, C1 + (long long)Y.i1 * A.up.cld, X, Y);
        c = resize_combine<SEM>(C2 + (long long)Y.i0 * A.up.cld, C2 + (long long)Y.i1 * A.up.cld, X, Y);
        if (A.up.post != 1.0f) { a = a * A.up.post; c = c * A.up.post; }
        A.up.u1o[orow + xx] = a; A.up.u2o[orow + xx] = c;
    };
    float u1n, u2n, i0n = I0r[x0];
    flow_at(x0, u1n, u2n);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int x = x0 + i * TX;
        if (x >= W) break;
        const float u1v = u1n, u2v = u2n, i0 = i0n;
        if (i + 1 < NP) {
            const int xn = min(x + TX, W - 1);   // clamped: the value of a patch past the right edge is never used
            if (!UP || x + TX < W) flow_at(xn, u1n, u2n);
            i0n = I0r[xn];
        }
        warp_px<SEM, FAST>(A, s_tab, P, x, y, orow + x, u1v, u2v, i0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// LDS-staged variant.  k_warp6 gathers 6 x (16 + 8) B per pixel through the vector-memory address path -- 144 B of requests for
// 4 B of new data, every request at 4-byte alignment: the kernel runs at the rate of that path (SQ counters: 32 % of the wave
// cycles issue-stalled on it, profiles/r02p), and it competes there with the iteration kernel of the other lane.  The flow is
// smooth, so the windows of a 64 x 16 tile of pixels cover a region only a few pixels larger than the tile: the workgroup
// finds that region from its own flows (min / max of the window origins), stages it with aligned, coalesced dwordx4 loads
// (about 2 floats per pixel), and every interior window is read from LDS.  Pixels whose window touches the image border, or
// tiles whose flow is so wild that the region exceeds the buffer, take the global path of k_warp6 (same arithmetic).
// Tap values, weights and the accumulation order are those of warp_px: bit-identical planes.
#ifdef MIFLOW_EXPERIMENTS   // measured slower under the two-lane overlap (r02z3): experiments build only (ship what is used)
constexpr int WL_TW = 64, WL_TH = 16, WL_PPT = 4;   // tile; rows per thread (wave w owns rows 4w .. 4w+3, lane = column)
constexpr int WL_RW = 96, WL_RH = 40;               // staged region capacity, floats x rows (15 KB)

template <int SEM, bool FAST>
__global__ __launch_bounds__(256) void k_warp_lds(Warp6Args A, CtlK ctl, int cur_host)
{
    __shared__ __attribute__((aligned(16))) float s_reg[WL_RH * WL_RW];
    __shared__ float s_tab[128];
    __shared__ int s_mm[4];   // min sx, max sx, min sy, max sy over the tile's interior windows
    if (SEM == MI_SEM_CPU_REF && threadIdx.x < 128) s_tab[threadIdx.x] = A.tab[threadIdx.x];
    if (threadIdx.x == 0) { s_mm[0] = 0x7fffffff; s_mm[1] = -0x7fffffff; s_mm[2] = 0x7fffffff; s_mm[3] = -0x7fffffff; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = A.g.w, H = A.g.h, ld = A.g.ld;
    const int x = blockIdx.x * WL_TW + lane;
    const int yb = blockIdx.y * WL_TH + wave * WL_PPT;
    const int b = blockIdx.z;
    if (!warp_gate_k(ctl, b)) return;   // uniform over the workgroup
    const int cur = resolve_cur_k(ctl, b, cur_host);
    const long long pb = (long long)b * A.g.ps;
    const float *P = A.I1 + pb;
    const int xc = min(x, W - 1);
    float u1v[WL_PPT], u2v[WL_PPT], i0[WL_PPT];
#pragma unroll
    for (int k = 0; k < WL_PPT; ++k) {
        const long long o = pb + (long long)min(yb + k, H - 1) * ld + xc;   // clamped: values of pixels outside the image are never used
        u1v[k] = A.u1[cur][o]; u2v[k] = A.u2[cur][o]; i0[k] = A.I0[o];
    }
    __syncthreads();   // s_tab, s_mm initialised
    // window origins of this thread's pixels; extent of the tile's interior windows
    int mn_x = 0x7fffffff, mx_x = -0x7fffffff, mn_y = 0x7fffffff, mx_y = -0x7fffffff;
#pragma unroll
    for (int k = 0; k < WL_PPT; ++k) {
        int sx, sy;
        float wxv[4], wyv[4];
        warp_coords<SEM>(s_tab, x, yb + k, u1v[k], u2v[k], sx, sy, wxv, wyv);
        const bool interior = (unsigned)(sx - 1) < (unsigned)max(W - 5, 0) && (unsigned)(sy - 1) < (unsigned)max(H - 5, 0);
        if (interior && x < W && yb + k < H) { mn_x = min(mn_x, sx); mx_x = max(mx_x, sx); mn_y = min(mn_y, sy); mx_y = max(mx_y, sy); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn_x = min(mn_x, __shfl_xor(mn_x, o)); mx_x = max(mx_x, __shfl_xor(mx_x, o));
        mn_y = min(mn_y, __shfl_xor(mn_y, o)); mx_y = max(mx_y, __shfl_xor(mx_y, o));
    }
    if (lane == 0) { atomicMin(&s_mm[0], mn_x); atomicMax(&s_mm[1], mx_x); atomicMin(&s_mm[2], mn_y); atomicMax(&s_mm[3], mx_y); }
    __syncthreads();
    // region of I1 all interior windows of the tile lie in: columns rx0 .. (first column rounded down to a 16-byte boundary),
    // rows ry0 .. ; inside the image by the definition of `interior`
    const int rx0 = (s_mm[0] - 1) & ~3, ry0 = s_mm[2] - 1;
    const int rw = s_mm[1] + 4 - rx0 + 1, rh = s_mm[3] + 4 - ry0 + 1;
    const bool staged = s_mm[1] >= s_mm[0] && rw <= WL_RW && rh <= WL_RH;   // wave-uniform (LDS broadcast)
    if (staged) {
        const int rw4 = (rw + 3) >> 2;   // float4 per row; rx0 + 4 * rw4 <= ld because ld is a multiple of 64 and rx0 + rw <= W
        for (int idx = threadIdx.x; idx < rh * (WL_RW / 4); idx += 256) {
            const int r = idx / (WL_RW / 4), c4 = idx - r * (WL_RW / 4);
            if (c4 < rw4) {
                const float4 v = *reinterpret_cast<const float4 *>(P + (long long)(ry0 + r) * ld + rx0 + 4 * c4);
                *reinterpret_cast<float4 *>(&s_reg[r * WL_RW + 4 * c4]) = v;
            }
        }
    }
    __syncthreads();
    if (x >= W) return;
#pragma unroll 1
    for (int k = 0; k < WL_PPT; ++k) {
        const int y = yb + k;
        if (y >= H) break;
        const long long o = pb + (long long)y * ld + x;
        int sx, sy;
        float wxv[4], wyv[4];
        warp_coords<SEM>(s_tab, x, y, u1v[k], u2v[k], sx, sy, wxv, wyv);
        const bool interior = (unsigned)(sx - 1) < (unsigned)max(W - 5, 0) && (unsigned)(sy - 1) < (unsigned)max(H - 5, 0);
        if (!(staged && interior)) {
            warp_px<SEM, FAST>(A, s_tab, P, x, y, o, u1v[k], u2v[k], i0[k]);
            continue;
        }
        float R[6][6];
        const float *q = &s_reg[(sy - 1 - ry0) * WL_RW + (sx - 1 - rx0)];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c)
                if (!((r == 0 || r == 5) && (c == 0 || c == 5))) R[r][c] = q[r * WL_RW + c];
        float v0, v1, v2;
        window_sums<SEM, FAST>(R, wxv, wyv, v0, v1, v2);
        if (A.I1w) A.I1w[o] = v0;
        A.I1wx[o] = v1;
        A.I1wy[o] = v2;
        if (A.grad) A.grad[o] = v1 * v1 + v2 * v2;
        A.rho[o] = (v0 - v1 * u1v[k] - v2 * u2v[k] - i0[k]);
    }
}

#endif   // MIFLOW_EXPERIMENTS (k_warp_lds)

#ifdef MIFLOW_EXPERIMENTS
bool warp_zoom_ok() { return tuning().warp_zoom != 0 && tuning().warp_lds == 0 && warp_tile() == 32 && tuning().warp_np == 2; }
#else
bool warp_zoom_ok() { return false; }   // the zooming first warp (k_warp6<.., UP>) lost its A/B (r08k): experiments build only
#endif

int warp_fused(int semantics, bool fast, int lds, const float *I0, const float *I1, const float *u1[2], const float *u2[2], float *I1w, float *I1wx,
               float *I1wy, float *grad, float *rho, const float *cubic_tab_dev, const Geo &g, const Ctl *ctl, int cur_host,
               hipStream_t s, const WarpZoom *zoom)
{
    Warp6Args A;
    memset(&A.up, 0, sizeof(A.up));
    const bool up = zoom != nullptr;
    if (up) {
        MI_REQUIRE(warp_zoom_ok() && lds <= 0 && !ctl, MI_ERR_BAD_ARG, "the zooming warp needs the default patch shape and a host-known buffer set");
        A.up.u1c = zoom->u1c; A.up.u2c = zoom->u2c; A.up.u1o = zoom->u1o; A.up.u2o = zoom->u2o;
        A.up.cw = zoom->gc.w; A.up.ch = zoom->gc.h; A.up.cld = zoom->gc.ld; A.up.cps = zoom->gc.ps; A.up.post = zoom->post;
        if (semantics == MI_SEM_CPU_REF) { A.up.scx = 1.0 / zoom->inv_scale_x; A.up.scy = 1.0 / zoom->inv_scale_y; }
        else { A.up.scx = (double)(float)(1.0 / zoom->inv_scale_x); A.up.scy = (double)(float)(1.0 / zoom->inv_scale_y); }   // as tvl1::resize
    }
    A.I0 = I0; A.I1 = I1; A.swz = tuning().warp_swz != 0 ? 1 : 0;
    A.u1[0] = u1[0]; A.u1[1] = u1[1]; A.u2[0] = u2[0]; A.u2[1] = u2[1];
    A.I1w = I1w; A.I1wx = I1wx; A.I1wy = I1wy; A.grad = grad; A.rho = rho;
    A.tab = cubic_tab_dev;
    A.g = g;
    const CtlK ck = make_ctlk(ctl);
    const bool cpu = semantics == MI_SEM_CPU_REF;
    if (lds < 0 ? tuning().warp_lds != 0 : lds != 0) {
#ifndef MIFLOW_EXPERIMENTS
        set_error("the LDS-staged warp (k_warp_lds) exists in the experiments build only (-DMIFLOW_EXPERIMENTS)");
        return MI_ERR_NOT_IMPL;
#else
        const dim3 grid(div_up(g.w, WL_TW), div_up(g.h, WL_TH), g.batch);
        if (cpu && fast) hipLaunchKernelGGL((k_warp_lds<MI_SEM_CPU_REF, true>), grid, dim3(256), 0, s, A, ck, cur_host);
        else if (cpu) hipLaunchKernelGGL((k_warp_lds<MI_SEM_CPU_REF, false>), grid, dim3(256), 0, s, A, ck, cur_host);
        else if (fast) hipLaunchKernelGGL((k_warp_lds<MI_SEM_CUDA_COMPAT, true>), grid, dim3(256), 0, s, A, ck, cur_host);
        else hipLaunchKernelGGL((k_warp_lds<MI_SEM_CUDA_COMPAT, false>), grid, dim3(256), 0, s, A, ck, cur_host);
        MI_HIP_TRY(hipGetLastError());
        return MI_OK;
#endif
    }
#ifndef MIFLOW_EXPERIMENTS
    // the shipped shape: 32 x 2 patches, two per wave (r01u / r02e sweeps); the other patch shapes and the zooming form are tuning variants
    MI_REQUIRE(!up, MI_ERR_NOT_IMPL, "the zooming warp exists in the experiments build only");
    {
        const dim3 grid(div_up(g.w, 32 * 2), div_up(g.h, 4 * 2), g.batch);
        if (cpu && fast) hipLaunchKernelGGL((k_warp6<MI_SEM_CPU_REF, 32, 2, true>), grid, dim3(256), 0, s, A, ck, cur_host);
        else if (cpu) hipLaunchKernelGGL((k_warp6<MI_SEM_CPU_REF, 32, 2, false>), grid, dim3(256), 0, s, A, ck, cur_host);
        else if (fast) hipLaunchKernelGGL((k_warp6<MI_SEM_CUDA_COMPAT, 32, 2, true>), grid, dim3(256), 0, s, A, ck, cur_host);
        else hipLaunchKernelGGL((k_warp6<MI_SEM_CUDA_COMPAT, 32, 2, false>), grid, dim3(256), 0, s, A, ck, cur_host);
        MI_HIP_TRY(hipGetLastError());
        return MI_OK;
    }
#else
    const int tile = warp_tile();
    const int np = tuning().warp_np;   // patches per wave (MIFLOW_WARP_NP = 1 | 2 | 4, default 2)
    const dim3 grid(div_up(g.w, tile * np), div_up(g.h, 4 * (64 / tile)), g.batch);
#define LAUNCH_W6T(SEM, TX, NP)                                                                                          \
    do {                                                                                                                 \
        if (up && TX == 32 && NP == 2) {   /* the zooming first warp of a scale: the default patch shape only */            \
            if (fast) hipLaunchKernelGGL((k_warp6<SEM, 32, 2, true, true>), grid, dim3(256), 0, s, A, ck, cur_host);     \
            else hipLaunchKernelGGL((k_warp6<SEM, 32, 2, false, true>), grid, dim3(256), 0, s, A, ck, cur_host);         \
        } else if (fast) hipLaunchKernelGGL((k_warp6<SEM, TX, NP, true>), grid, dim3(256), 0, s, A, ck, cur_host);       \
        else hipLaunchKernelGGL((k_warp6<SEM, TX, NP, false>), grid, dim3(256), 0, s, A, ck, cur_host);                  \
    } while (0)
#define LAUNCH_W6N(SEM, NP)                                                                                              \
    do {                                                                                                                 \
        if (tile == 16) LAUNCH_W6T(SEM, 16, NP); else if (tile == 32) LAUNCH_W6T(SEM, 32, NP); else LAUNCH_W6T(SEM, 64, NP); \
    } while (0)
#define LAUNCH_W6(SEM)                                                                                                   \
    do {                                                                                                                 \
        if (np == 1) LAUNCH_W6N(SEM, 1); else if (np == 4) LAUNCH_W6N(SEM, 4); else LAUNCH_W6N(SEM, 2);                    \
    } while (0)
    if (cpu) LAUNCH_W6(MI_SEM_CPU_REF);
    else LAUNCH_W6(MI_SEM_CUDA_COMPAT);
#undef LAUNCH_W6T
#undef LAUNCH_W6N
#undef LAUNCH_W6
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
#endif
}

}  // namespace tvl1
}  // namespace mi
