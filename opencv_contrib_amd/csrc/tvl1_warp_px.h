// Per-pixel device routines of the fused-gradient warp (warpBackward with the centred gradient of I1 derived from a 6 x 6 window):
// shared by the warp kernels (tvl1_warp_kernels.hip) and by the producer waves of the fused pass kernel (tvl1_tbr_kernels.hip, FW),
// so that both run the SAME operations in the same order -- bit-identical planes.  See tvl1_warp_kernels.hip for the derivation.
#pragma once
#include "tvl1_dev.h"
#include "tvl1_warp_dev.h"

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

// Window of an INTERIOR pixel: rows sy-1 .. sy+4; R[r][c] = I1(sy - 1 + r, sx - 1 + c); rows 0 and 5 are needed at columns 1..4 only
// (the corners stay unset and unused).  Six requests: four dwordx4 + dwordx2 rows and two dwordx4 rows, 4-byte aligned.
__device__ __forceinline__ bool window_interior(int sx, int sy, int W, int H)
{
    return (unsigned)(sx - 1) < (unsigned)max(W - 5, 0) && (unsigned)(sy - 1) < (unsigned)max(H - 5, 0);
}
__device__ __forceinline__ void window_gather(float (&R)[6][6], const float *P, int ld, int sx, int sy)
{
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
}

// The three bicubic sums of a pixel whose window touches the image border (per-tap clamping as centeredGradient / the clamp-addressed
// texture do it); the loads happen inside.
template <int SEM>
__device__ __forceinline__ void window_border(const float *P, int W, int H, int ld, int sx, int sy, const float (&wxv)[4], const float (&wyv)[4],
                                              float &v0, float &v1, float &v2)
{
    if (SEM == MI_SEM_CPU_REF) {
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
    if (window_interior(sx, sy, W, H)) {
        float R[6][6];
        window_gather(R, P, ld, sx, sy);
        window_sums<SEM, FAST>(R, wxv, wyv, v0, v1, v2);
    } else {
        window_border<SEM>(P, W, H, ld, sx, sy, wxv, wyv, v0, v1, v2);
    }
    if (A.I1w) A.I1w[o] = v0;
    A.I1wx[o] = v1;
    A.I1wy[o] = v2;
    // calcGradRho  optflow/src/tvl1flow.cpp:918-944 == tvl1flow.cu:151-163
    const float Ix2 = v1 * v1, Iy2 = v2 * v2;
    if (A.grad) A.grad[o] = Ix2 + Iy2;   // null: the consumer (k_iterate_tbr NG) forms |grad|^2 from the two planes above itself
    A.rho[o] = (v0 - v1 * u1v - v2 * u2v - i0);
}

}  // namespace tvl1
}  // namespace mi
