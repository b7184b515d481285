// Dual TV-L1 -- temporally blocked fused iteration for gfx950 (wave64, 512 VGPR/lane budget).
//
// The reference runs 2 kernels per inner iteration (estimateU + estimateDualVariables,
// modules/cudaoptflow/src/cuda/tvl1flow.cu:209-363), i.e. 88 B of HBM traffic per pixel per
// iteration; the fused v1 kernel (tvl1_kernels.hip) needs 64 B.  Here T iterations are applied
// in ONE pass over HBM (64 B per pixel per T iterations):
//
//   * a wave owns a column strip of 64*PPL pixels and streams DOWN the rows of its band;
//   * iteration level t (1..T) is a pipeline stage that lags level t-1 by one row
//     (dependency cone of one iteration is +-1 px: tvl1flow.cu:187-207 reads p at x-1/y-1,
//     :322-327 reads u at x+1/y+1).  Each stage keeps one row of {u_t, p_(t-1), static planes}
//     in VGPRs (10*PPL registers), so all T levels of a row band live in the register file;
//   * x-neighbours come from the adjacent lane through DPP wave shifts (no LDS, no barriers);
//   * the strip loses ceil(T/PPL)*PPL px of validity on each side and T rows at band top and
//     bottom (recomputed by the neighbouring strip/band).
//
// Arithmetic: fast-math form of the CPU reference formulas (optflow/src/tvl1flow.cpp:989-1041,
// 857-899, 1096-1112, 1140-1181): thresholding written as fi = clamp(-rho/g, +-l_t), reciprocals
// via v_rcp_f32, |grad u| via v_sqrt_f32, fma contraction.  Parity of this path is tested
// against the exact-math v1 kernel and the oracle with a stated tolerance.
#include "tvl1_tb_dev.h"
#include <cfloat>
#include <vector>
#include <cstdlib>
#include <cstdio>

namespace mi {
namespace tvl1 {


__global__ void k_dbg_lane_shift(int *out)
{
    const int l = threadIdx.x;
    out[l] = __float_as_int(dpp_from_prev(__int_as_float(l + 100)));
    out[64 + l] = __float_as_int(dpp_from_next(__int_as_float(l + 100)));
}


template <int PPL>
__device__ __forceinline__ void ldv(float dst[PPL], const float *p, long long off, bool ok)
{
    if (PPL == 1) {
        dst[0] = ok ? p[off] : 0.f;
    } else if (PPL == 2) {
        float2 v = make_float2(0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float2 *>(p + off);
        dst[0] = v.x; dst[1] = v.y;
    } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4 *>(p + off);
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
}
template <int PPL>
__device__ __forceinline__ void stv(float *p, long long off, const float v[PPL])
{
    if (PPL == 1) p[off] = v[0];
    else if (PPL == 2) *reinterpret_cast<float2 *>(p + off) = make_float2(v[0], v[1]);
    else *reinterpret_cast<float4 *>(p + off) = make_float4(v[0], v[1], v[2], v[3]);
}

// One pipeline stage (iteration level t).  `in` = level t-1 row a (u, p), `st` = static row a,
// `S` = what this stage holds (u_t(a-1), p_(t-1)(a-1)); writes the new held state (row a) to `N`
// and replaces `in` by level t row a-1.  a, H are wave-uniform (SGPRs).
template <int PPL, bool EDGE>
__device__ __forceinline__ void stage(Dyn<PPL> &in, const Stat<PPL> &st, const Dyn<PPL> &S, Dyn<PPL> &N, int a, int H,
                                      bool has_left, bool has_right, bool x_is_zero, const bool right_ok[PPL],
                                      float l_t, float theta, float taut)
{
    // EDGE = false: interior step of an interior strip -- no image border can be touched by any stage, straight-line code.
    // EDGE = true: has_left / has_right / a are wave-uniform, the border fix-ups are scalar branches (the empty asm keeps
    // the compiler from if-converting them into per-pixel selects).
    // ---- u_t(a)
    float dx1[PPL], dx2[PPL];  // backward x-differences of p11, p21
    dx1[0] = in.p11[0] - dpp_from_prev(in.p11[PPL - 1]);
    dx2[0] = in.p21[0] - dpp_from_prev(in.p21[PPL - 1]);
#pragma unroll
    for (int j = 1; j < PPL; ++j) { dx1[j] = in.p11[j] - in.p11[j - 1]; dx2[j] = in.p21[j] - in.p21[j - 1]; }
    if (EDGE && has_left) {  // first column: no p(x-1) term (optflow tvl1flow.cpp:893-894)
        asm volatile("" ::: "memory");
        if (x_is_zero) { dx1[0] = in.p11[0]; dx2[0] = in.p21[0]; }
    }
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const float div1 = dx1[j] + (in.p12[j] - S.p12[j]);
        const float div2 = dx2[j] + (in.p22[j] - S.p22[j]);
        const float rho = fmaf(st.ix[j], in.u1[j], fmaf(st.iy[j], in.u2[j], st.rc[j]));
        const float fi = __builtin_amdgcn_fmed3f(-rho * st.rg[j], -l_t, l_t);  // TH: clamp(-rho/grad, +-l_t)
        N.u1[j] = fmaf(theta, div1, fmaf(fi, st.ix[j], in.u1[j]));
        N.u2[j] = fmaf(theta, div2, fmaf(fi, st.iy[j], in.u2[j]));
        N.p11[j] = in.p11[j]; N.p12[j] = in.p12[j]; N.p21[j] = in.p21[j]; N.p22[j] = in.p22[j];
    }
    if (EDGE && a == H) {  // below the last row: forward y-difference is 0 (optflow tvl1flow.cpp:826-831)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < PPL; ++j) { N.u1[j] = S.u1[j]; N.u2[j] = S.u2[j]; }
    }
    // ---- p_t(a-1)
    const float r1 = dpp_from_next(S.u1[0]);
    const float r2 = dpp_from_next(S.u2[0]);
    float u1xv[PPL], u2xv[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const float n1 = (j < PPL - 1) ? S.u1[j + 1 < PPL ? j + 1 : j] : r1;
        const float n2 = (j < PPL - 1) ? S.u2[j + 1 < PPL ? j + 1 : j] : r2;
        u1xv[j] = n1 - S.u1[j];
        u2xv[j] = n2 - S.u2[j];
    }
    if (EDGE && has_right) {  // last column: forward x-difference is 0 (optflow tvl1flow.cpp:833-838)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < PPL; ++j) { u1xv[j] = right_ok[j] ? u1xv[j] : 0.f; u2xv[j] = right_ok[j] ? u2xv[j] : 0.f; }
    }
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const float u1x = u1xv[j];
        const float u2x = u2xv[j];
        const float u1y = N.u1[j] - S.u1[j];
        const float u2y = N.u2[j] - S.u2[j];
        const float g1 = __builtin_amdgcn_sqrtf(fmaf(u1x, u1x, u1y * u1y));
        const float g2 = __builtin_amdgcn_sqrtf(fmaf(u2x, u2x, u2y * u2y));
        const float q1 = __builtin_amdgcn_rcpf(fmaf(taut, g1, 1.0f));
        const float q2 = __builtin_amdgcn_rcpf(fmaf(taut, g2, 1.0f));
        in.p11[j] = fmaf(taut, u1x, S.p11[j]) * q1;
        in.p12[j] = fmaf(taut, u1y, S.p12[j]) * q1;
        in.p21[j] = fmaf(taut, u2x, S.p21[j]) * q2;
        in.p22[j] = fmaf(taut, u2y, S.p22[j]) * q2;
        in.u1[j] = S.u1[j]; in.u2[j] = S.u2[j];
    }
    if (EDGE && a <= 0) {  // the emitted row a-1 lies above the image: p(y-1) terms vanish at y == 0 (:889-890)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < PPL; ++j) in.p12[j] = in.p22[j] = 0.f;
    }
}

template <int PPL, bool PZ>
__device__ __forceinline__ void load_input_row(Dyn<PPL> &r, Stat<PPL> &st, const TbArgs &A, const float *const u[2],
                                               const float *const p[4], int row, int H, unsigned xc)
{
    const long long ro = (long long)min(max(row, 0), H - 1) * A.g.ld;   // wave-uniform
    ldu<PPL>(st.ix, A.pl.ix + ro, xc);
    ldu<PPL>(st.iy, A.pl.iy + ro, xc);
    ldu<PPL>(st.rg, A.pl.g + ro, xc);   // RAW |grad|^2 until the row is consumed (finish_static)
    ldu<PPL>(st.rc, A.pl.rc + ro, xc);
    ldu<PPL>(r.u1, u[0] + ro, xc);
    ldu<PPL>(r.u2, u[1] + ro, xc);
    if (!PZ) {
        ldu<PPL>(r.p11, p[0] + ro, xc);
        ldu<PPL>(r.p12, p[1] + ro, xc);
        ldu<PPL>(r.p21, p[2] + ro, xc);
        ldu<PPL>(r.p22, p[3] + ro, xc);
    } else {
#pragma unroll
        for (int j = 0; j < PPL; ++j) r.p11[j] = r.p12[j] = r.p21[j] = r.p22[j] = 0.f;
    }
}
template <int PPL, bool PZ>
__device__ __forceinline__ void mask_input_row(Dyn<PPL> &r, Stat<PPL> &st, bool ok)
{
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        st.ix[j] = ok ? st.ix[j] : 0.f; st.iy[j] = ok ? st.iy[j] : 0.f;
        st.rg[j] = ok ? st.rg[j] : 0.f; st.rc[j] = ok ? st.rc[j] : 0.f;
        r.u1[j] = ok ? r.u1[j] : 0.f; r.u2[j] = ok ? r.u2[j] : 0.f;
        if (!PZ) {
            r.p11[j] = ok ? r.p11[j] : 0.f; r.p12[j] = ok ? r.p12[j] : 0.f;
            r.p21[j] = ok ? r.p21[j] : 0.f; r.p22[j] = ok ? r.p22[j] : 0.f;
        }
    }
}

// One step of the whole pipeline: row `arow` of level 0 enters, row arow-T of level T leaves in `io`.  The static row of
// stage t+1 is fetched from the LDS ring before stage t computes (ds_read latency hidden behind one stage of VALU work).
template <int T, int PPL, int K, bool EDGE>
__device__ __forceinline__ void pipeline_step(Dyn<PPL> &io, const Stat<PPL> &st0, const Dyn<PPL> (&S)[T], Dyn<PPL> (&N)[T],
                                              float *ring, int slot0, int lane, int arow, int H, bool has_left,
                                              bool has_right, bool x_is_zero, const bool right_ok[PPL], float l_t,
                                              float theta, float taut)
{
    lds_put<PPL>(ring + slot0 * (256 * PPL), lane, st0);
    Stat<PPL> st = st0, nx;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) {
            int sl = slot0 - (t + 1);
            if (sl < 0) sl += K;
            lds_get<PPL>(ring + sl * (256 * PPL), lane, nx);
        }
        stage<PPL, EDGE>(io, st, S[t], N[t], arow - t, H, has_left, has_right, x_is_zero, right_ok, l_t, theta, taut);
        st = nx;
    }
}

// PF = prefetch depth in pipeline steps (rows): the kernel is bound by HBM bytes in flight (r01i PMC: 63 % of wave time in
// s_waitcnt, ~4 TB/s of real traffic), and a prefetched row costs only 10*PPL VGPRs.
template <int T, int PPL, bool PZ, int WPS, int PF>
__global__ __launch_bounds__(256, WPS) void k_iterate_tb(TbArgs A)
{
    constexpr int M = (T + PPL - 1) / PPL * PPL;    // validity margin per side (px)
    constexpr int STRIDE = 64 * PPL - 2 * M;        // owned columns per strip
    constexpr int K = T + 1;                        // ring slots (rows of static planes in flight)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Block -> (strip, band group, pair) mapping.  swz 0: natural order (the dispatcher deals workgroup `id` to XCD id % 8, so
    // neighbouring strips sit on different XCDs).  swz 1: bijective XCD remap of the whole grid (cdna_hip_programming.md T1).
    // swz 2: remap of the strips inside one grid row only (gridDim.x padded to a multiple of 8 by the launcher).
    int strip = blockIdx.x, bgrp = blockIdx.y, b = blockIdx.z;
    if (A.swz == 1) {
        const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
        const unsigned orig = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned xcd = orig & 7u, qq = nwg >> 3, rr = nwg & 7u;
        const unsigned lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
        strip = lid % gridDim.x; bgrp = (lid / gridDim.x) % gridDim.y; b = lid / (gridDim.x * gridDim.y);
    } else if (A.swz == 2) {
        strip = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        if (strip >= A.nstrips) return;
    }
    strip = __builtin_amdgcn_readfirstlane(strip); bgrp = __builtin_amdgcn_readfirstlane(bgrp); b = __builtin_amdgcn_readfirstlane(b);
    const int band = bgrp * 4 + wave;
    const int W = A.g.w, H = A.g.h, ld = A.g.ld;
    const int y0 = band * A.rows_per_band;
    float *ring = lds + wave * (K * 256 * PPL);
    if (y0 >= H) return;
    const int y1 = min(y0 + A.rows_per_band, H);
    const int own_lo = strip * STRIDE, own_hi = min(own_lo + STRIDE, W);
    const int xl = own_lo - M + lane * PPL;         // first pixel of this lane (may be < 0 or >= W)
    const bool xok = xl >= 0 && xl < W;
    const unsigned xc = 4u * (unsigned)min(max(xl, 0), ld - PPL);   // clamped column of the unconditional loads, in bytes (ld >= W, multiple of 64)
    const bool x_is_zero = (xl == 0);
    bool right_ok[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) right_ok[j] = (xl + j + 1 < W);
    const bool st_ok = xl >= own_lo && xl < own_hi;  // owned group (groups never straddle own_lo)
    const bool has_left = (strip == 0);                              // wave-uniform: strip contains x == 0
    const bool has_right = (own_lo - M + 64 * PPL >= W);             // wave-uniform: strip reaches x == W-1
    const bool edge_strip = has_left || has_right;

    const long long pb = (long long)b * A.g.ps;
    const int cur = A.cur;
    const float *uin[2] = {A.pl.u[cur][0] + pb, A.pl.u[cur][1] + pb};
    const float *pin[4] = {A.pl.p[cur][0] + pb, A.pl.p[cur][1] + pb, A.pl.p[cur][2] + pb, A.pl.p[cur][3] + pb};
    float *uout[2] = {A.pl.u[cur ^ 1][0] + pb, A.pl.u[cur ^ 1][1] + pb};
    float *pout[4] = {A.pl.p[cur ^ 1][0] + pb, A.pl.p[cur ^ 1][1] + pb, A.pl.p[cur ^ 1][2] + pb, A.pl.p[cur ^ 1][3] + pb};
    TbArgs B = A;
    B.pl.ix += pb; B.pl.iy += pb; B.pl.g += pb; B.pl.rc += pb;

    // zero state: rows above the first streamed row do not exist for this band
    Dyn<PPL> SA[T], SB[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int j = 0; j < PPL; ++j)
            SA[t].u1[j] = SA[t].u2[j] = SA[t].p11[j] = SA[t].p12[j] = SA[t].p21[j] = SA[t].p22[j] = 0.f;
    {
        Stat<PPL> z;
#pragma unroll
        for (int j = 0; j < PPL; ++j) z.ix[j] = z.iy[j] = z.rg[j] = z.rc[j] = 0.f;
        for (int k = 0; k < K; ++k) lds_put<PPL>(ring + k * (256 * PPL), lane, z);
    }

    const int ystart = y0 - T;
    const int nsteps = (y1 - y0) + 2 * T;
    Dyn<PPL> nxt[PF];
    Stat<PPL> nst[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) load_input_row<PPL, PZ>(nxt[k], nst[k], B, uin, pin, ystart + k, H, xc);
    int slot0 = 0;

    auto emit = [&](const Dyn<PPL> &r, int orow) {
        if (orow >= y0 && orow < y1 && st_ok) {
            const long long ro = (long long)orow * ld;   // wave-uniform; owned lanes have xc == 4 * xl
            stu<PPL>(uout[0] + ro, xc, r.u1);
            stu<PPL>(uout[1] + ro, xc, r.u2);
            stu<PPL>(pout[0] + ro, xc, r.p11);
            stu<PPL>(pout[1] + ro, xc, r.p12);
            stu<PPL>(pout[2] + ro, xc, r.p21);
            stu<PPL>(pout[3] + ro, xc, r.p22);
        }
    };

    constexpr int U = PF < 2 ? 2 : PF;   // unroll: even (state ping-pongs SA <-> SB) and a multiple of PF (static prefetch slots)
    for (int s = 0; s < nsteps; s += U) {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            constexpr int dummy = 0; (void)dummy;
            const int slot = k % PF;
            Dyn<PPL> io = nxt[slot];
            Stat<PPL> st0 = nst[slot];
            const int r0 = ystart + s + k;
            // wave-uniform: can any stage of this step touch an image border (row r0-t, t < T, or the x borders)?
            const bool edge = edge_strip || r0 < T || r0 >= H;
            if (edge) mask_input_row<PPL, PZ>(io, st0, xok && r0 >= 0 && r0 < H);
            finish_static<PPL>(st0);
            load_input_row<PPL, PZ>(nxt[slot], nst[slot], B, uin, pin, r0 + PF, H, xc);
            // even step: state SA -> SB, odd step: SB -> SA (rows past the band end are computed but never stored)
            if (edge) {
                if ((k & 1) == 0)
                    pipeline_step<T, PPL, K, true>(io, st0, SA, SB, ring, slot0, lane, r0, H, has_left, has_right, x_is_zero, right_ok, A.l_t, A.theta, A.taut);
                else
                    pipeline_step<T, PPL, K, true>(io, st0, SB, SA, ring, slot0, lane, r0, H, has_left, has_right, x_is_zero, right_ok, A.l_t, A.theta, A.taut);
            } else {
                if ((k & 1) == 0)
                    pipeline_step<T, PPL, K, false>(io, st0, SA, SB, ring, slot0, lane, r0, H, has_left, has_right, x_is_zero, right_ok, A.l_t, A.theta, A.taut);
                else
                    pipeline_step<T, PPL, K, false>(io, st0, SB, SA, ring, slot0, lane, r0, H, has_left, has_right, x_is_zero, right_ok, A.l_t, A.theta, A.taut);
            }
            emit(io, ystart + s + k - T);
            slot0 = (slot0 + 1 == K) ? 0 : slot0 + 1;
        }
    }
}

// ------------------------------------------------------------------ host side
// A variant = (T iterations per pass, PPL pixels per lane, WPS = waves/SIMD the register allocator must
// leave room for).  More waves per SIMD hide the s_waitcnt stalls (rocprofv3: 40 % of wave time at
// 2 waves/SIMD), fewer registers per wave cap T: the table is the measured trade-off.
struct TbVariant {
    int T, PPL, WPS, PF;
    void (*launch)(const TbArgs &, bool, hipStream_t);
};

template <int T, int PPL, int WPS, int PF>
static void launch_tb(const TbArgs &A0, bool pz, hipStream_t s)
{
    constexpr int M = (T + PPL - 1) / PPL * PPL;
    constexpr int STRIDE = 64 * PPL - 2 * M;
    TbArgs A = A0;
    A.nstrips = div_up(A.g.w, STRIDE);
    static int swz = -1;
    if (swz < 0) { const char *e = getenv("MIFLOW_TB_SWZ"); swz = e ? atoi(e) : 1; }
    A.swz = swz;
    const dim3 grid(swz == 2 ? div_up(A.nstrips, 8) * 8 : A.nstrips, div_up(div_up(A.g.h, A.rows_per_band), 4), A.g.batch);
    constexpr size_t lds_bytes = (size_t)4 * (T + 1) * 256 * PPL * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_iterate_tb<T, PPL, true, WPS, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute((const void *)k_iterate_tb<T, PPL, false, WPS, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    if (pz) hipLaunchKernelGGL((k_iterate_tb<T, PPL, true, WPS, PF>), grid, dim3(256), lds_bytes, s, A);
    else hipLaunchKernelGGL((k_iterate_tb<T, PPL, false, WPS, PF>), grid, dim3(256), lds_bytes, s, A);
}

#define TBV(T, PPL, WPS, PF) {T, PPL, WPS, PF, launch_tb<T, PPL, WPS, PF>}
static const TbVariant g_variants[] = {
    // ping-pong-state kernels (MIFLOW_TB_ROT=0; superseded by tvl1_tbr_kernels.hip, kept as an independent cross-check).
    // r01p sweep, G px-iter/s at 1080p x 16: T8 (2 px/lane) 346 | T10 (1 px/lane, 3 waves/SIMD, 2 rows prefetched) 311 |
    // T6 (1,4,2) 274 | T5 (2,2,2) 255 | T4 (2,1,1) 227 | T3 163 | T2 111
    TBV(1, 2, 1, 1), TBV(2, 1, 8, 1), TBV(3, 2, 4, 1), TBV(4, 2, 1, 1), TBV(5, 2, 2, 2), TBV(6, 1, 4, 2), TBV(8, 2, 1, 1), TBV(10, 1, 3, 2),
};

static const TbVariant *pick_variant(int T)
{
    static int want_ppl = -1, want_wps = -1, want_pf = 1;
    static bool parsed = false;
    if (!parsed) {
        parsed = true;
        if (const char *e = getenv("MIFLOW_TB_VARIANT")) (void)sscanf(e, "%d,%d,%d", &want_ppl, &want_wps, &want_pf);
    }
    const TbVariant *def = nullptr;
    for (const TbVariant &v : g_variants) {
        if (v.T != T) continue;
        if (!def) def = &v;
        if (v.PPL == want_ppl && v.WPS == want_wps && v.PF == want_pf) return &v;
    }
    return def;
}

int tb_max_block() { return 10; }

// Decompose n iterations into supported time blocks minimising the modelled cost.  cost[T] = measured
// ps per pixel-iteration of k_iterate_tb<T> at 1080p x 16 pairs (tools/sweep_tb.py, profiles/r01s, rotating-slot kernels):
// deeper blocks save HBM passes but cost registers (occupancy) and halo recomputation.
int tb_plan(int n, int cap, int *blocks, int max_blocks)
{
    static const int sup[] = {1, 2, 3, 4, 5, 6, 8, 10};
    static const double cost[11] = {0, 15.6, 9.4, 6.6, 4.65, 3.9, 3.7, 0, 2.83, 0, 2.54};
    if (n <= 0) return 0;
    if (getenv("MIFLOW_TB_FORCE")) {   // tuning sweeps: greedy blocks of exactly `cap` (then the largest that fit)
        int k = 0;
        for (int left = n; left > 0 && k < max_blocks;) {
            int t = 1;
            for (int c : sup) if (c <= left && c <= cap) t = c;
            blocks[k++] = t;
            left -= t;
        }
        return k;
    }
    std::vector<double> best(n + 1, 1e300);
    std::vector<int> pick(n + 1, 1);
    best[0] = 0;
    for (int i = 1; i <= n; ++i)
        for (int t : sup) {
            if (t > i || t > cap) continue;
            const double c = best[i - t] + t * cost[t];
            if (c < best[i]) { best[i] = c; pick[i] = t; }
        }
    int k = 0;
    for (int i = n; i > 0 && k < max_blocks; i -= pick[i]) blocks[k++] = pick[i];
    return k;
}

// T fused iterations, set cur -> cur^1.  Returns MI_ERR_BAD_ARG for unsupported T.
int iterate_tb(int T, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut, bool p_zero,
               int cur, int rows_per_band, hipStream_t s)
{
    // kernel family: 1 = rotating-slot formulation (tvl1_tbr_kernels.hip) where it has an entry for T, 0 = ping-pong state
    static int rot = -1, want_ppl = -1, want_wps = -1, want_pf = -1;
    if (rot < 0) {
        const char *e = getenv("MIFLOW_TB_ROT");
        rot = e ? atoi(e) : 1;
        if (const char *v = getenv("MIFLOW_TB_VARIANT")) (void)sscanf(v, "%d,%d,%d", &want_ppl, &want_wps, &want_pf);
    }
    int ppl = 0, wps_v = 0, pf = 0, ring_slots = T + 1, rot_P = 0;
    TbLaunch launch = nullptr;
    if (rot) {
        launch = tbr_pick(T, want_ppl, want_wps, want_pf, &ppl, &wps_v, &pf);
        ring_slots = T > 2 ? T - 1 : 1;
        if (launch) rot_P = T + 1 + pf;
    }
    if (!launch) {
        const TbVariant *v = pick_variant(T);
        if (!v) { set_error("unsupported time block %d", T); return MI_ERR_BAD_ARG; }
        launch = v->launch; ppl = v->PPL; wps_v = v->WPS; pf = v->PF; ring_slots = T + 1; rot_P = 0;
    }
    TbArgs A;
    A.pl = pl; A.g = g; A.l_t = l_t; A.theta = theta; A.taut = taut; A.cur = cur; A.swz = 0; A.nstrips = 0;
    if (rows_per_band <= 0) {
        // Band height: every wave streams rows_per_band + 2T rows.  Pick the band count that minimises
        //   rounds x (rows + 2T),  rounds = ceil(waves / resident-wave capacity),
        // so the grid fills the 1024 SIMDs in whole rounds (no half-empty tail round) while the 2T-row
        // band overlap stays small.  Capacity: waves/SIMD allowed by the variant's VGPR count
        // (-Rpass-analysis=kernel-resource-usage, gfx950), its LDS ring and the 8-wave hardware limit.
        static const int wps_of_T_ppl2[11] = {8, 7, 5, 4, 3, 3, 2, 2, 2, 2, 2};
        int wps = wps_v > 1 ? wps_v : (ppl == 2 ? wps_of_T_ppl2[T] : 4);
        const int lds_blocks = (160 * 1024) / (ring_slots * 4 * 256 * ppl * 4);
        if (wps > lds_blocks) wps = lds_blocks;
        if (const char *e = getenv("MIFLOW_TB_WPS")) wps = atoi(e) > 0 ? atoi(e) : wps;
        const long long cap = 1024LL * wps;
        const int M = (T + ppl - 1) / ppl * ppl;
        const long long per_band = (long long)div_up(g.w, 64 * ppl - 2 * M) * g.batch;
        long long best_cost = -1;
        int best_nb = 1;
        for (int nb = 1; nb <= g.h; ++nb) {
            const int R = div_up(g.h, nb);
            if (R < 8 && nb > 1) break;
            const long long rounds = (per_band * nb + cap - 1) / cap;
            long long steps = R + 2 * T;
            if (rot_P > 0) steps = (steps + rot_P - 1) / rot_P * rot_P;   // rotating kernels run whole blocks of P steps
            const long long cost = rounds * steps;
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_nb = nb; }
        }
        rows_per_band = div_up(g.h, best_nb);
        if (const char *e = getenv("MIFLOW_TB_ROWS")) rows_per_band = atoi(e) > 0 ? atoi(e) : rows_per_band;
    }
    A.rows_per_band = rows_per_band;
    if (getenv("MIFLOW_TB_VERBOSE")) {
        static int shown = 0;
        if (shown++ < 40)
            fprintf(stderr, "[tb] T=%d rot=%d ppl=%d wps=%d pf=%d %dx%d batch=%d rows_per_band=%d\n", T, rot_P > 0, ppl, wps_v, pf, g.w, g.h,
                    g.batch, rows_per_band);
    }
    launch(A, p_zero, s);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int dbg_lane_shift(int *out_dev, hipStream_t s)
{
    hipLaunchKernelGGL(k_dbg_lane_shift, dim3(1), dim3(64), 0, s, out_dev);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace tvl1
}  // namespace mi
