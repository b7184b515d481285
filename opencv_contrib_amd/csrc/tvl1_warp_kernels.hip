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
#include "resize_dev.h"

namespace mi {
namespace tvl1 {

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ float bicubic_coeff_cuda6(float x_)
{
    // tvl1flow.cu:89-104 (Keys a = -0.5)
    const float x = fabsf(x_);
    if (x <= 1.0f) return x * x * (1.5f * x - 2.5f) + 1.0f;
    else if (x < 2.0f) return x * (x * (-0.5f * x + 2.5f) - 4.0f) + 2.0f;
    return 0.0f;
}

// {I1, I1x, I1y} at pixel (cx, cy) of the image, neighbours clamped as centeredGradient does
__device__ __forceinline__ void fetch3(const float *P, int W, int H, int ld, int cx, int cy, float &v, float &vx, float &vy)
{
    const float *row = P + (long long)cy * ld;
    v = row[cx];
    vx = 0.5f * (row[min(cx + 1, W - 1)] - row[max(cx - 1, 0)]);
    vy = 0.5f * (P[(long long)min(cy + 1, H - 1) * ld + cx] - P[(long long)max(cy - 1, 0) * ld + cx]);
}

// The three bicubic sums of an interior window R (rows sy-1 .. sy+4, columns sx-1 .. sx+4; the corners are unused):
// I1, I1x = 0.5 (I1[x+1] - I1[x-1]) and I1y = 0.5 (I1[y+1] - I1[y-1]) interpolated at the 4 x 4 taps.
// FAST = false: the reference's arithmetic, tap by tap (bit-exact against the oracle): CPU_REF = cv::remap's bicubic interior,
//   sum += S[0]*w[0] + S[1]*w[1] + S[2]*w[2] + S[3]*w[3] per row with w = wy[j]*wx[i]; CUDA_COMPAT = tvl1flow.cu:118-148, one
//   running sum per plane and the weights normalised by their sum.
// FAST = true (fast-math calcs: the default under MI_SEM_CUDA_COMPAT, opt-in MIFLOW_WARP_FAST=1 under MI_SEM_CPU_REF): the same three sums in SEPARABLE form -- six row sums of I1 with the x weights shared
//   by I1 and I1y, four row sums with the differenced x weights for I1x, then the y weights: 68 instead of 173 VALU operations per
//   pixel, +4.7 % pairs/s at 1080p (r02z5: 1 223 vs 1 167).  It differs from the tap-by-tap sums by rounding only (~3e-5 on images
//   in 0..255), but rho_c = I1w - ... - I0 cancels to a small number, so that rounding perturbs the flow -- and under the CPU
//   class's semantics every perturbation of the flow is amplified by the 1/32-px quantisation of the next warp's map (a pixel
//   whose map coordinate crosses a bin boundary samples I1 1/32 px away).  Measured mean EPE against the oracle at N = 10 on
//   240x320 .. 388x584 pairs: 2.0e-3 .. 3.4e-3 px (6e-3 with gamma = 1) against 0.7e-3 .. 1.7e-3 with the tap-by-tap sums --
//   inside the reference's own CUDA-vs-CPU acceptance by two orders of magnitude, but not inside this repo's stated 5e-3 bound
//   everywhere, hence not the default there.  cv::cuda's semantics sample the unquantised map: 1.2e-5 px with either form.  (Keeping only I1 tap by tap: 108 operations, +1 %, no better than the default.)
template <int SEM, bool FAST>
__device__ __forceinline__ void window_sums(const float (&R)[6][6], const float (&wxv)[4], const float (&wyv)[4], float &v0, float &v1,
                                            float &v2)
{
    if (FAST) {
        float a[6], bx[4];
#pragma unroll
        for (int r = 0; r < 6; ++r)
            a[r] = fmaf(wxv[3], R[r][4], fmaf(wxv[2], R[r][3], fmaf(wxv[1], R[r][2], wxv[0] * R[r][1])));
        const float d2 = wxv[0] - wxv[2], d3 = wxv[1] - wxv[3];
#pragma unroll
        for (int r = 1; r < 5; ++r)
            bx[r - 1] = fmaf(wxv[3], R[r][5], fmaf(wxv[2], R[r][4], fmaf(d3, R[r][3], fmaf(d2, R[r][2], fmaf(-wxv[1], R[r][1], -wxv[0] * R[r][0])))));
        float s0 = fmaf(wyv[3], a[4], fmaf(wyv[2], a[3], fmaf(wyv[1], a[2], wyv[0] * a[1])));
        float s1 = fmaf(wyv[3], bx[3], fmaf(wyv[2], bx[2], fmaf(wyv[1], bx[1], wyv[0] * bx[0])));
        float s2 = fmaf(wyv[3], a[5] - a[3], fmaf(wyv[2], a[4] - a[2], fmaf(wyv[1], a[3] - a[1], wyv[0] * (a[2] - a[0]))));
        if (SEM == MI_SEM_CUDA_COMPAT) {
            const float coeff = __builtin_amdgcn_rcpf(((wxv[0] + wxv[1]) + (wxv[2] + wxv[3])) * ((wyv[0] + wyv[1]) + (wyv[2] + wyv[3])));
            s0 *= coeff; s1 *= coeff; s2 *= coeff;
        }
        v0 = s0; v1 = 0.5f * s1; v2 = 0.5f * s2;
    } else if (SEM == MI_SEM_CPU_REF) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t0[4], t1[4], t2[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w[i] = wyv[j] * wxv[i];
                t0[i] = R[j + 1][i + 1];
                // the centred differences WITHOUT their factor 0.5: scaling by a power of two commutes with every rounding below
                // (products and sums; nothing here comes near the subnormal range), so 0.5 * (sum of d w) == sum of (0.5 d) w bit
                // for bit, and the 32 multiplies by 0.5 of a pixel become the two at the end
                t1[i] = R[j + 1][i + 2] - R[j + 1][i];
                t2[i] = R[j + 2][i + 1] - R[j][i + 1];
            }
            const float r0 = t0[0] * w[0] + t0[1] * w[1] + t0[2] * w[2] + t0[3] * w[3];
            const float r1 = t1[0] * w[0] + t1[1] * w[1] + t1[2] * w[2] + t1[3] * w[3];
            const float r2 = t2[0] * w[0] + t2[1] * w[1] + t2[2] * w[2] + t2[3] * w[3];
            if (j == 0) { s0 = r0; s1 = r1; s2 = r2; } else { s0 += r0; s1 += r1; s2 += r2; }
        }
        v0 = s0; v1 = 0.5f * s1; v2 = 0.5f * s2;
    } else {
        float sum = 0.f, sumx = 0.f, sumy = 0.f, wsum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float wgt = wxv[i] * wyv[j];
                sum += wgt * R[j + 1][i + 1];
                sumx += wgt * (R[j + 1][i + 2] - R[j + 1][i]);   // the 0.5 of the centred difference: hoisted, see above
                sumy += wgt * (R[j + 2][i + 1] - R[j][i + 1]);
                wsum += wgt;
            }
        const float coeff = 1.0f / wsum;
        v0 = sum * coeff; v1 = (0.5f * sumx) * coeff; v2 = (0.5f * sumy) * coeff;
    }
}

// fetch3 through a buffer descriptor of the pair's plane: one 32-bit byte offset per load instead of a 64-bit address pair
__device__ __forceinline__ void fetch3b(__amdgpu_buffer_rsrc_t rs, int W, int H, int ld, int cx, int cy, float &v, float &vx, float &vy)
{
    const unsigned ro = (unsigned)cy * (unsigned)ld;
    const auto L = [&](unsigned e) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 4u * e, 0, 0)); };
    v = L(ro + cx);
    vx = 0.5f * (L(ro + min(cx + 1, W - 1)) - L(ro + max(cx - 1, 0)));
    vy = 0.5f * (L((unsigned)min(cy + 1, H - 1) * (unsigned)ld + cx) - L((unsigned)max(cy - 1, 0) * (unsigned)ld + cx));
}

// Window origin (first tap column / row of the 4 x 4 window) and the eight tap weights of output pixel (x, y) with flow (u1v, u2v).
// CPU_REF: buildFlowMap + cv::remap(INTER_CUBIC): optflow/src/tvl1flow.cpp:650-666,1371-1374 -- the map quantised to 1/32 px, the
// weights from the 32-phase table (a = -0.75).  CUDA_COMPAT: tvl1flow.cu:106-149 -- the reference visits cx = ceil(wx - 2) ..
// floor(wx + 2): the four taps floor(wx) - 1 .. floor(wx) + 2 plus, when a bound lands on an integer, taps at distance >= 2 whose
// weight is exactly 0 and which therefore add +-0 to every sum -- the fixed 4-tap window gives the same bits.
template <int SEM>
__device__ __forceinline__ void warp_coords(const float *s_tab, int x, int y, float u1v, float u2v, int &sx, int &sy, float (&wxv)[4],
                                            float (&wyv)[4])
{
    if (SEM == MI_SEM_CPU_REF) {
        const float mx = (float)x + u1v, my = (float)y + u2v;
        const int qx = __float2int_rn(mx * 32.0f), qy = __float2int_rn(my * 32.0f);
        sx = min(max(qx >> 5, -32768), 32767) - 1;
        sy = min(max(qy >> 5, -32768), 32767) - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { wxv[k] = s_tab[(qx & 31) * 4 + k]; wyv[k] = s_tab[(qy & 31) * 4 + k]; }
    } else {
        const float wxp = (float)x + u1v, wyp = (float)y + u2v;
        sx = (int)fminf(fmaxf(floorf(wxp), -1.0e9f), 1.0e9f) - 1;
        sy = (int)fminf(fmaxf(floorf(wyp), -1.0e9f), 1.0e9f) - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wxv[k] = bicubic_coeff_cuda6(wxp - (float)(sx + k));
            wyv[k] = bicubic_coeff_cuda6(wyp - (float)(sy + k));
        }
    }
}

// One output pixel (x, y) of pair plane P: u1v, u2v = the flow at the pixel, i0 = I0 there; writes the five planes at o.
template <int SEM, bool FAST>
__device__ __forceinline__ void warp_px(const Warp6Args &A, const float *s_tab, const float *P, int x, int y, long long o, float u1v,
                                        float u2v, float i0)
{
    const int W = A.g.w, H = A.g.h, ld = A.g.ld;
    int sx, sy;          // first tap column / row of the 4 x 4 window
    float wxv[4], wyv[4];
    warp_coords<SEM>(s_tab, x, y, u1v, u2v, sx, sy, wxv, wyv);

    float v0, v1, v2;
    const bool interior = (unsigned)(sx - 1) < (unsigned)max(W - 5, 0) && (unsigned)(sy - 1) < (unsigned)max(H - 5, 0);
    if (interior) {
        // rows sy-1 .. sy+4; R[r][c] = I1(sy - 1 + r, sx - 1 + c); rows 0 and 5 are needed at columns 1..4 only
        float R[6][6];
        const float *q = P + (long long)(sy - 1) * ld + (sx - 1);
        {
            const f4u a = *reinterpret_cast<const f4u *>(q + 1);
            R[0][1] = a.x; R[0][2] = a.y; R[0][3] = a.z; R[0][4] = a.w;
        }
#pragma unroll
        for (int r = 1; r < 5; ++r) {
            const float *qr = q + (long long)r * ld;
            const f4u a = *reinterpret_cast<const f4u *>(qr);
            const f2u c = *reinterpret_cast<const f2u *>(qr + 4);
            R[r][0] = a.x; R[r][1] = a.y; R[r][2] = a.z; R[r][3] = a.w; R[r][4] = c.x; R[r][5] = c.y;
        }
        {
            const f4u a = *reinterpret_cast<const f4u *>(q + (long long)5 * ld + 1);
            R[5][1] = a.x; R[5][2] = a.y; R[5][3] = a.z; R[5][4] = a.w;
        }
        window_sums<SEM, FAST>(R, wxv, wyv, v0, v1, v2);
    } else if (SEM == MI_SEM_CPU_REF) {
        // Windows touching the border.  The values a tap row needs are I1(sy + j - 1 .. sy + j + 1, sx - 1 .. sx + 4) with CLAMPED
        // rows and columns -- for a tap inside the image that is exactly how centeredGradient clamps its neighbours, and taps
        // outside the image are not used.  One tap row per trip of a rolled loop: its 14 loads are independent (one memory round
        // trip per row; the earlier tap-by-tap loops waited for every tap's five loads in turn, ~30 us at the end of EVERY
        // launch), and the rolled loop keeps the kernel at the interior path's register count.  Buffer loads: a wave-uniform
        // descriptor of the pair's plane + one 32-bit byte offset per load.
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P), 0, (unsigned)H * (unsigned)ld * 4u, 0x00020000);
        const auto L = [&](unsigned e) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 4u * e, 0, 0)); };
        unsigned cxs[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) cxs[c] = (unsigned)min(max(sx - 1 + c, 0), W - 1);
        // the 4 x 4 window of taps is inside the image (only the ring of derivative neighbours is not): cv::remap's interior
        // formula, sum += S[0]*w[0] + S[1]*w[1] + S[2]*w[2] + S[3]*w[3] per row, w = wy[j]*wx[i].  Otherwise its border path: one
        // tap at a time, taps outside the image contribute the border value 0 (skipped: s + (0 - 0) * w == s); a window entirely
        // outside the image has no valid tap and yields 0.
        const bool win4 = (unsigned)sx < (unsigned)max(W - 3, 0) && (unsigned)sy < (unsigned)max(H - 3, 0);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int yj = sy + j;
            const unsigned ra = (unsigned)min(max(yj - 1, 0), H - 1) * (unsigned)ld, rb = (unsigned)min(max(yj, 0), H - 1) * (unsigned)ld,
                           rc = (unsigned)min(max(yj + 1, 0), H - 1) * (unsigned)ld;
            float Ra[4], Rb[6], Rc[4];
#pragma unroll
            for (int c = 0; c < 6; ++c) Rb[c] = L(rb + cxs[c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) { Ra[c] = L(ra + cxs[c + 1]); Rc[c] = L(rc + cxs[c + 1]); }
            // wy[j] with a rolled j: select instead of a dynamically indexed register array
            const float wyj = j == 0 ? wyv[0] : j == 1 ? wyv[1] : j == 2 ? wyv[2] : wyv[3];
            float t0[4], t1[4], t2[4], w[4];
            const bool okj = (unsigned)yj < (unsigned)H;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w[i] = wyj * wxv[i];
                t0[i] = Rb[i + 1];
                t1[i] = 0.5f * (Rb[i + 2] - Rb[i]);
                t2[i] = 0.5f * (Rc[i] - Ra[i]);
            }
            const float r0 = t0[0] * w[0] + t0[1] * w[1] + t0[2] * w[2] + t0[3] * w[3];
            const float r1 = t1[0] * w[0] + t1[1] * w[1] + t1[2] * w[2] + t1[3] * w[3];
            const float r2 = t2[0] * w[0] + t2[1] * w[1] + t2[2] * w[2] + t2[3] * w[3];
            if (j == 0) { s0 = r0; s1 = r1; s2 = r2; } else { s0 += r0; s1 += r1; s2 += r2; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = okj && (unsigned)(sx + i) < (unsigned)W;
                const float n0 = b0 + (t0[i] - 0.f) * w[i], n1 = b1 + (t1[i] - 0.f) * w[i], n2 = b2 + (t2[i] - 0.f) * w[i];
                b0 = ok ? n0 : b0; b1 = ok ? n1 : b1; b2 = ok ? n2 : b2;
            }
        }
        v0 = win4 ? s0 : b0; v1 = win4 ? s1 : b1; v2 = win4 ? s2 : b2;
    } else {
        // Border windows of cv::cuda's semantics (clamp-addressed taps).  The reference visits cx = ceil(wx - 2) .. floor(wx + 2):
        // the four taps floor(wx) - 1 .. floor(wx) + 2 plus, when wx is an integer, one more on each side whose weight is exactly 0
        // -- terms +-0 * (finite clamped data), which change no sum: the fixed 4 x 4 window of the interior path gives the same
        // bits.  The four taps of a row are unrolled -- their 20 loads are independent, one memory round trip per row; the
        // earlier tap-by-tap loop waited for every tap's loads in turn (~30 us at the end of every launch).
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P), 0, (unsigned)H * (unsigned)ld * 4u, 0x00020000);
        float sum = 0.f, sumx = 0.f, sumy = 0.f, wsum = 0.f;
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int cy = sy + j;
            const float wyj = j == 0 ? wyv[0] : j == 1 ? wyv[1] : j == 2 ? wyv[2] : wyv[3];
            float t0[4], t1[4], t2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fetch3b(rs, W, H, ld, min(max(sx + i, 0), W - 1), min(max(cy, 0), H - 1), t0[i], t1[i], t2[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float wgt = wxv[i] * wyj;
                sum += wgt * t0[i];
                sumx += wgt * t1[i];
                sumy += wgt * t2[i];
                wsum += wgt;
            }
        }
        const float coeff = 1.0f / wsum;
        v0 = sum * coeff; v1 = sumx * coeff; v2 = sumy * coeff;
    }
    if (A.I1w) A.I1w[o] = v0;
    A.I1wx[o] = v1;
    A.I1wy[o] = v2;
    // calcGradRho  optflow/src/tvl1flow.cpp:918-944 == tvl1flow.cu:151-163
    const float Ix2 = v1 * v1, Iy2 = v2 * v2;
    if (A.grad) A.grad[o] = Ix2 + Iy2;   // null: the consumer (k_iterate_tbr NG) forms |grad|^2 from the two planes above itself
    A.rho[o] = (v0 - v1 * u1v - v2 * u2v - i0);
}

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

bool warp_zoom_ok() { return tuning().warp_zoom != 0 && tuning().warp_lds == 0 && warp_tile() == 32 && tuning().warp_np == 2; }

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
        const dim3 grid(div_up(g.w, WL_TW), div_up(g.h, WL_TH), g.batch);
        if (cpu && fast) hipLaunchKernelGGL((k_warp_lds<MI_SEM_CPU_REF, true>), grid, dim3(256), 0, s, A, ck, cur_host);
        else if (cpu) hipLaunchKernelGGL((k_warp_lds<MI_SEM_CPU_REF, false>), grid, dim3(256), 0, s, A, ck, cur_host);
        else if (fast) hipLaunchKernelGGL((k_warp_lds<MI_SEM_CUDA_COMPAT, true>), grid, dim3(256), 0, s, A, ck, cur_host);
        else hipLaunchKernelGGL((k_warp_lds<MI_SEM_CUDA_COMPAT, false>), grid, dim3(256), 0, s, A, ck, cur_host);
        MI_HIP_TRY(hipGetLastError());
        return MI_OK;
    }
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
}

}  // namespace tvl1
}  // namespace mi
