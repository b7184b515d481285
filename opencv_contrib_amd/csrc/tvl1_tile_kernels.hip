// Dual TV-L1 -- fused iterations on a 2-D REGISTER TILE (gfx950, wave64): the small pyramid levels.
//
// The streaming pipeline of tvl1_tbr_kernels.hip walks a row band from top to bottom with its T iteration levels as
// pipeline stages: one dependent chain of T stages per row step, (rows + 2T) steps per band.  On a large level that chain
// is hidden by the other waves of the SIMD; on a level of a few hundred waves (small frames, the levels of a single pair)
// nothing hides it and a launch costs its serial depth -- 130 us and more whatever the pixel count (r02z4: 320 x 240 x 16
// pairs 7 850 pairs/s; a single 1080p pair per calc() 486 calcs/s).
//
// Here a workgroup of NW waves holds a whole tile of (NW * RW) rows x 64 columns in registers -- every lane a column, every
// wave RW consecutive rows, {u1, u2, p11, p12, p21, p22} and the four static planes of each row -- and runs the iterations
// in place, all rows of a phase at once:
//
//   U phase:  u_t(a)  = f(u_(t-1)(a), p_(t-1)(a), p_(t-1)(a-1))     for every row a of the tile   (x-1 neighbour by DPP)
//   P phase:  p_t(a)  = g(p_(t-1)(a), u_t(a), u_t(a+1))              for every row a of the tile   (x+1 neighbour by DPP)
//
// The RW rows of a wave are independent inside a phase (RW-fold instruction-level parallelism instead of one chain); the
// row above a wave's first row and the row below its last row belong to the neighbouring waves and cross through LDS
// (two 64-float rows per wave and phase, two workgroup barriers per iteration).  The serial depth of a launch is
// nit x (2 phases + 2 barriers), independent of the image height.
//
// The tile carries a margin of M = 10 rows / columns per side (dependency cone +-1 px per iteration), so up to 10 iterations
// run per launch; the columns are cut exactly like the strips of k_iterate_tbr (strip 0 starts at x = 0 in lane 0).
// Arithmetic and border cuts are those of stage_r (tvl1_tbr_kernels.hip), operation by operation, so the results are
// BIT-IDENTICAL to the streaming kernel's (tests/test_tvl1_gpu.py::test_iterate_tile_equals_streaming_kernel): which of the
// two kernels a level runs on never changes a flow.
#include "tvl1_tb_dev.h"
#include <cstdio>
#include <cstring>

namespace mi {
namespace tvl1 {

struct TileArgs {
    IterPlanes pl;
    Geo g;
    float l_t, theta, taut;
    int cur;   // input set
    int nit;   // iterations of this launch, 1..10 (SPEC: the most this launch may run; the device picks the count)
    int swz;   // XCD-contiguous tile order (MIFLOW_TILE_SWZ, default 1)
    // SPEC only: slot protocol of the speculative steps (tvl1_tb_dev.h spec_settle), e0 = index of the first error sum of this block
    CtlK ctl;
    SpecK sk;
    int e0;
};

constexpr int TILE_M = 10;   // validity margin per side = the most iterations one launch may run

// SPEC: one speculative step of the convergence-checked path (k_iterate_tbr MODE 1 on register tiles): the block length comes from
// the device-side settle logic, and every iteration's error sum(du1^2 + du2^2) over the tile's OWNED pixels is added -- as 2^-24
// fixed-point integers, so the totals do not depend on the tiling -- to the per-iteration slots the next launch reads.
// M = margin of the tile per side = the most iterations the launch can run.  The speculative steps of a single pair know from the
// handle's previous calc how many a warp will need (SpecK::h_in, and host-side Lane::fb_hist): a block of at most 4 iterations on
// M = 4 tiles owns 56 x 56 of its 64 x 64 pixels instead of 44 x 44 -- 1.6 x fewer workgroups, loads and lane-iterations per pass.
// Which margin a block ran on never shows: the arithmetic per pixel is the same and the error sums are exact integers.
// GAM (round 6): gamma != 0 -- the illumination channel (u3, p31, p32; tvl1flow.cu:209-288,313-348) on the register tile, the operations of
// stage_r<.., GAM> in its order: bit-identical to the streaming kernel with the channel.  Like that kernel it forms |grad|^2 from I1wx, I1wy
// itself (the warp's own expression: two separately rounded products and their sum) -- with gamma != 0 the warp never stores the plane.
template <int RW, int NW, bool PZ, bool SPEC, int M = TILE_M, bool GAM = false>
__global__ __launch_bounds__(NW * 64) void k_iterate_tile(TileArgs A)
{
    constexpr int LW = 64;
    constexpr int STRIDE = LW - 2 * M;        // owned columns of the strips >= 1 (strip 0 owns LW - M)
    constexpr int BR = NW * RW - 2 * M;       // owned rows of a tile
    static_assert(BR > 0, "tile too small for its margin");
    // [wave][0,1: u1,u2 of the wave's first row (after the U phase)  2,3: p12,p22 of its last row][lane]
    // GAM: 4: u3 of the first row, 5: p32 of the last row
    __shared__ float xch[NW][GAM ? 6 : 4][64];
    // SPEC: per-iteration error sums of the workgroup; one global add per workgroup and iteration at the end (one per WAVE and
    // iteration -- 17 600 waves at 1080p -- made the launch 10 x slower: the adds of a slot serialise in L2)
    __shared__ unsigned long long s_err[M];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.z;
    const long long pb = (long long)b * A.g.ps;
    int cur = A.cur, nit = A.nit;
    bool record = false;
    if (SPEC) {
        // every thread takes the same decision from the same device data: the whole workgroup leaves or stays
        // (before the tile order is worked out: most launches of a batch's plan end right here, and the index arithmetic below showed
        // in their cost; the writer is tile (0, 0) in either order)
        if (!spec_settle(A.ctl, A.sk, A.nit, b, blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0, cur, nit, record)) return;
    }
    // XCD-contiguous tile order (round 4): workgroups are dealt round-robin over the 8 XCDs in launch order, and a tile loads its
    // margins -- its neighbours' pixels -- itself: with the neighbours behind seven other L2s every margin came from HBM a second
    // time (x 1.3 .. 2.1 of the planes per pass).  One XCD takes a contiguous eighth of a pair's (band, strip) tiles.
    int strip = blockIdx.x, band = blockIdx.y;
    if (A.swz == 1 || (A.swz == 2 && (!SPEC || gridDim.z <= 2))) {   // MIFLOW_TILE_SWZ = 2: not for the speculative steps of a batch
        // within the pair's own (band, strip) plane: the tiles of one pair stay spread over all XCDs (pairs of a batch stop at
        // different launches; a pair per XCD left the other XCDs idle: r10 class defaults 1080p x 4 390 -> 374 pairs/s).  The residue
        // class of the in-plane index is one XCD whatever the plane's offset in launch order.
        const unsigned nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned xcd = orig & 7u, qq = nwg >> 3, rr = nwg & 7u;
        const unsigned lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
        band = (int)(lid / gridDim.x); strip = (int)(lid - (unsigned)band * gridDim.x);
    }
    const int W = A.g.w, H = A.g.h, ld = A.g.ld;
    const int y0 = band * BR, y1 = min(y0 + BR, H);
    const int own_lo = strip == 0 ? 0 : (LW - M) + (strip - 1) * STRIDE;
    const int own_hi = min(strip == 0 ? LW - M : own_lo + STRIDE, W);
    const int xl = (strip == 0 ? 0 : own_lo - M) + lane;   // this lane's column, >= 0
    const bool right_ok = xl + 1 < W;
    const bool st_ok = xl >= own_lo && xl < own_hi;
    const unsigned xc = 4u * (unsigned)min(xl, ld - 1);    // clamped column of the unconditional loads, bytes
    const int ys = y0 - M + wave * RW;                     // image row of this wave's first register row

    const float *const uin[2] = {A.pl.u[cur][0] + pb, A.pl.u[cur][1] + pb};
    const float *const pin[4] = {A.pl.p[cur][0] + pb, A.pl.p[cur][1] + pb, A.pl.p[cur][2] + pb, A.pl.p[cur][3] + pb};
    const float *const stp[4] = {A.pl.ix + pb, A.pl.iy + pb, GAM ? nullptr : A.pl.g + pb, A.pl.rc + pb};
    const float *const u3in = GAM ? A.pl.u[cur][2] + pb : nullptr;
    const float *const p3in[2] = {GAM ? A.pl.p[cur][4] + pb : nullptr, GAM ? A.pl.p[cur][5] + pb : nullptr};
    const float gamma = A.pl.gamma, eu3 = A.pl.err_u3 ? 1.f : 0.f;

    float u1[RW], u2[RW], p11[RW], p12[RW], p21[RW], p22[RW];
    float ix[RW], iy[RW], rg[RW], rc[RW];
    float u3[RW], p31[RW], p32[RW];   // GAM only
    // rows above / below the image are loaded from clamped addresses (finite data); the four border cuts below isolate the
    // valid region from them exactly as in stage_r
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const long long ro = (long long)min(max(ys + r, 0), H - 1) * ld;   // wave-uniform
        const auto ldv = [&](const float *plane) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(plane + ro) + xc); };
        ix[r] = ldv(stp[0]); iy[r] = ldv(stp[1]); rc[r] = ldv(stp[3]);
        if constexpr (GAM) { const float ix2 = ix[r] * ix[r], iy2 = iy[r] * iy[r]; rg[r] = ix2 + iy2; }   // (step_r NG: the warp's expression)
        else rg[r] = ldv(stp[2]);
        u1[r] = ldv(uin[0]); u2[r] = ldv(uin[1]);
        if (!PZ) { p11[r] = ldv(pin[0]); p12[r] = ldv(pin[1]); p21[r] = ldv(pin[2]); p22[r] = ldv(pin[3]); }
        else p11[r] = p12[r] = p21[r] = p22[r] = 0.f;
        if constexpr (GAM) {
            u3[r] = ldv(u3in);
            if (!PZ) { p31[r] = ldv(p3in[0]); p32[r] = ldv(p3in[1]); }
            else p31[r] = p32[r] = 0.f;
        } else u3[r] = p31[r] = p32[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) rg[r] = __builtin_amdgcn_rcpf(fmaxf(rg[r], 1e-30f));   // finish_static
    xch[wave][2][lane] = p12[RW - 1];
    xch[wave][3][lane] = p22[RW - 1];
    if constexpr (GAM) xch[wave][5][lane] = p32[RW - 1];
    if (SPEC && threadIdx.x < M) s_err[threadIdx.x] = 0ull;
    __syncthreads();

    const float l_t = A.l_t, theta = A.theta, taut = A.taut;
    for (int t = 0; t < nit; ++t) {
        unsigned long long acc = 0;
        // ---- U phase: u_t(a) for every row (stage_r, first half; optflow/src/tvl1flow.cpp:989-1041, 1096-1112)
        // the row above the tile's first row does not exist: any finite value (that row is margin, or cut by negm1 at a = 0)
        const float pa12 = wave > 0 ? xch[wave - 1][2][lane] : 0.f;
        const float pa22 = wave > 0 ? xch[wave - 1][3][lane] : 0.f;
        float pa32 = 0.f;
        if constexpr (GAM) pa32 = wave > 0 ? xch[wave - 1][5][lane] : 0.f;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int a = ys + r;
            const float negm1 = __uint_as_float((a == 0) ? 0u : 0xbf800000u);   // -(a != 0): the p(y-1) term of row 0 is cut
            const float dx1 = p11[r] - dpp_from_prev(p11[r]);
            const float dx2 = p21[r] - dpp_from_prev(p21[r]);
            const float ab12 = r == 0 ? pa12 : p12[r > 0 ? r - 1 : 0];
            const float ab22 = r == 0 ? pa22 : p22[r > 0 ? r - 1 : 0];
            const float div1 = dx1 + fmaf(negm1, ab12, p12[r]);
            const float div2 = dx2 + fmaf(negm1, ab22, p22[r]);
            float rho0 = rc[r];
            if constexpr (GAM) rho0 = fmaf(gamma, u3[r], rho0);
            const float rho = fmaf(ix[r], u1[r], fmaf(iy[r], u2[r], rho0));
            const float fi = __builtin_amdgcn_fmed3f(-rho * rg[r], -l_t, l_t);
            const float nu1 = fmaf(theta, div1, fmaf(fi, ix[r], u1[r]));
            const float nu2 = fmaf(theta, div2, fmaf(fi, iy[r], u2[r]));
            float e3sq = 0.f;
            if constexpr (GAM) {   // stage_r<.., GAM>, first half
                const float dx3 = p31[r] - dpp_from_prev(p31[r]);
                const float ab32 = r == 0 ? pa32 : p32[r > 0 ? r - 1 : 0];
                const float div3 = dx3 + fmaf(negm1, ab32, p32[r]);
                const float nu3 = fmaf(theta, div3, fmaf(fi, gamma, u3[r]));
                if (SPEC) { const float e3 = nu3 - u3[r]; e3sq = e3 * e3 * eu3; }
                u3[r] = nu3;
            }
            if (SPEC) {   // stage_r<ERR>: the pixel's term rounded to 2^-24 px^2; es doubles as the row-ownership mask
                const float es = __uint_as_float((a >= y0 && a < y1) ? 0x4b800000u : 0u);
                const float e1 = nu1 - u1[r], e2 = nu2 - u2[r];
                const float et = GAM ? fmaf(e1, e1, e2 * e2) + e3sq : fmaf(e1, e1, e2 * e2);
                acc += (unsigned long long)__float2uint_rn(et * es);
            }
            u1[r] = nu1;
            u2[r] = nu2;
        }
        if (SPEC && record) {   // exact integer reduction over the wave's owned lanes, one device-scope add per wave and iteration
            unsigned long long sacc = st_ok ? acc : 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
            if (lane == 0) atomicAdd(&s_err[t], sacc);
        }
        xch[wave][0][lane] = u1[0];
        xch[wave][1][lane] = u2[0];
        if constexpr (GAM) xch[wave][4][lane] = u3[0];
        __syncthreads();
        // ---- P phase: p_t(a) for every row (stage_r, second half; :1140-1181).  Below the tile's last row: the row itself
        // (zero y-difference; margin)
        const float nb1 = wave + 1 < NW ? xch[wave + 1][0][lane] : u1[RW - 1];
        const float nb2 = wave + 1 < NW ? xch[wave + 1][1][lane] : u2[RW - 1];
        float nb3 = 0.f;
        if constexpr (GAM) nb3 = wave + 1 < NW ? xch[wave + 1][4][lane] : u3[RW - 1];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int a = ys + r;
            const bool last = (a + 1 == H);   // forward y-difference of row H-1 is cut (:826-831)
            const float m2 = __uint_as_float(last ? 0u : 0x3f800000u);
            const float taum2 = __uint_as_float(last ? 0u : __float_as_uint(taut));
            const float r1 = dpp_from_next(u1[r]);
            const float r2 = dpp_from_next(u2[r]);
            const float bl1 = r + 1 < RW ? u1[r + 1 < RW ? r + 1 : r] : nb1;
            const float bl2 = r + 1 < RW ? u2[r + 1 < RW ? r + 1 : r] : nb2;
            const float u1x = right_ok ? r1 - u1[r] : 0.f;
            const float u2x = right_ok ? r2 - u2[r] : 0.f;
            const float d1 = bl1 - u1[r];
            const float d2 = bl2 - u2[r];
            const float g1 = __builtin_amdgcn_sqrtf(fmaf(d1 * d1, m2, u1x * u1x));
            const float g2 = __builtin_amdgcn_sqrtf(fmaf(d2 * d2, m2, u2x * u2x));
            const float q1 = __builtin_amdgcn_rcpf(fmaf(taut, g1, 1.0f));
            const float q2 = __builtin_amdgcn_rcpf(fmaf(taut, g2, 1.0f));
            p11[r] = fmaf(taut, u1x, p11[r]) * q1;
            p12[r] = fmaf(taum2, d1, p12[r]) * q1;
            p21[r] = fmaf(taut, u2x, p21[r]) * q2;
            p22[r] = fmaf(taum2, d2, p22[r]) * q2;
            if constexpr (GAM) {   // stage_r<.., GAM>, second half
                const float r3 = dpp_from_next(u3[r]);
                const float bl3 = r + 1 < RW ? u3[r + 1 < RW ? r + 1 : r] : nb3;
                const float u3x = right_ok ? r3 - u3[r] : 0.f;
                const float d3 = bl3 - u3[r];
                const float g3 = __builtin_amdgcn_sqrtf(fmaf(d3 * d3, m2, u3x * u3x));
                const float q3 = __builtin_amdgcn_rcpf(fmaf(taut, g3, 1.0f));
                p31[r] = fmaf(taut, u3x, p31[r]) * q3;
                p32[r] = fmaf(taum2, d3, p32[r]) * q3;
            }
        }
        xch[wave][2][lane] = p12[RW - 1];
        xch[wave][3][lane] = p22[RW - 1];
        if constexpr (GAM) xch[wave][5][lane] = p32[RW - 1];
        __syncthreads();
    }

    // (the last barrier of the loop has made every wave's s_err adds visible)
    if (SPEC && record && threadIdx.x < nit) atomicAdd(&A.ctl.E[(long long)b * A.ctl.Q + A.e0 + threadIdx.x], s_err[threadIdx.x]);
    if (st_ok) {
        float *const uout[2] = {A.pl.u[cur ^ 1][0] + pb, A.pl.u[cur ^ 1][1] + pb};
        float *const pout[4] = {A.pl.p[cur ^ 1][0] + pb, A.pl.p[cur ^ 1][1] + pb, A.pl.p[cur ^ 1][2] + pb, A.pl.p[cur ^ 1][3] + pb};
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int a = ys + r;
            if (a >= y0 && a < y1) {   // wave-uniform
                const long long o = (long long)a * ld + xl;
                uout[0][o] = u1[r]; uout[1][o] = u2[r];
                pout[0][o] = p11[r]; pout[1][o] = p12[r]; pout[2][o] = p21[r]; pout[3][o] = p22[r];
                if constexpr (GAM) {
                    (A.pl.u[cur ^ 1][2] + pb)[o] = u3[r];
                    (A.pl.p[cur ^ 1][4] + pb)[o] = p31[r];
                    (A.pl.p[cur ^ 1][5] + pb)[o] = p32[r];
                }
            }
        }
    }
}

template <int RW, int NW, bool SPEC, int M = TILE_M, bool GAM = false>
static int launch_tile(const TileArgs &A, bool pz, hipStream_t s)
{
    constexpr int LW = 64, STRIDE = LW - 2 * M, BR = NW * RW - 2 * M;
    const int nstrips = A.g.w <= LW - M ? 1 : 1 + div_up(A.g.w - (LW - M), STRIDE);
    const dim3 grid(nstrips, div_up(A.g.h, BR), A.g.batch);
    if (pz && !SPEC) hipLaunchKernelGGL((k_iterate_tile<RW, NW, true, false, M, GAM>), grid, dim3(NW * 64), 0, s, A);
    else hipLaunchKernelGGL((k_iterate_tile<RW, NW, false, SPEC, M, GAM>), grid, dim3(NW * 64), 0, s, A);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

typedef int (*TileLaunchFn)(const TileArgs &, bool, hipStream_t);
struct TileEntry {
    int RW, NW;
    TileLaunchFn launch, spec, spec4, spec7;   // spec4 / spec7: blocks of at most 4 / 7 iterations on tiles of that margin
    TileLaunchFn launch_gam, spec_gam;         // gamma != 0 (margin 10 only); nullptr for the shapes of the experiments build
};
#define TILE(RW, NW) {RW, NW, launch_tile<RW, NW, false>, launch_tile<RW, NW, true>, launch_tile<RW, NW, true, 4>, launch_tile<RW, NW, true, 7>, nullptr, nullptr}
#define TILEG(RW, NW) {RW, NW, launch_tile<RW, NW, false>, launch_tile<RW, NW, true>, launch_tile<RW, NW, true, 4>, launch_tile<RW, NW, true, 7>, \
                       launch_tile<RW, NW, false, TILE_M, true>, launch_tile<RW, NW, true, TILE_M, true>}
// [0] = the large-grid shape (64-row tiles of 16 waves), [1] = the small-grid shape (48-row tiles of 8 waves x 6 rows): what a release
// build runs (auto_variant).  The other shapes of the r02 / r10 sweeps exist in the experiments build only.
static const TileEntry g_tile[] = {TILEG(4, 16), TILEG(6, 8),
#ifdef MIFLOW_EXPERIMENTS
                                   TILE(6, 16), TILE(8, 16), TILE(8, 8), TILE(3, 16),
#endif
};
constexpr int kTileVariants = (int)(sizeof(g_tile) / sizeof(g_tile[0]));

int tile_variants() { return kTileVariants; }
// MIFLOW_TILE_VARIANT < 0 (default): by the size of the grid -- where 64-row tiles of 16 waves give the device fewer than four
// workgroups per CU (the levels of a single pair, the coarse levels of a small batch) 48-row tiles of 8 waves x 6 rows run the same
// iterations faster (r10c: one pair per calc 640 x 480 552 -> 606 calcs/s, 1080p 362 -> 375)
// spec: the speculative steps of the convergence-checked path take the small-grid shape up to twice the grid size (round 6, r19k: their
// blocks are mostly shorter than the margin, and the 48-row tiles' shorter dependent chain per iteration pays for longer -- one 1080p pair
// per calc() with the class defaults 423 -> 439 calcs/s -- while fixed-work launches lose 2-4 % there)
static int auto_variant(const Geo &g, bool spec = false)
{
    const int v = tuning().tile_variant;
    if (v >= 0) return v < kTileVariants ? v : 0;
    constexpr int M = TILE_M, LW = 64, STRIDE = LW - 2 * M;
    const long long nstrips = g.w <= LW - M ? 1 : 1 + div_up(g.w - (LW - M), STRIDE);
    const long long wgs = nstrips * div_up(g.h, 64 - 2 * M) * g.batch;
    return wgs < (long long)tuning().tile_small_wgs * (spec ? 2 : 1) ? 1 : 0;
}
int tile_max_block() { return TILE_M; }
int tile_rows_for(const Geo &g)
{
    const TileEntry &e = g_tile[auto_variant(g, true)];   // (asked by the cost model of the speculative steps)
    return e.RW * e.NW;
}
int tile_owned_rows()
{
    int v = tuning().tile_variant;
    if (v < 0 || v >= kTileVariants) v = 0;   // automatic: the large-grid variant
    return g_tile[v].RW * g_tile[v].NW - 2 * TILE_M;
}

// Levels of at most this many pixels x pairs run on the register-tile kernel (0: never).  MIFLOW_TILE_MAXPX.
bool tile_eligible(const Geo &g)
{
    const long long lim = tuning().tile_maxpx;
    return lim > 0 && (long long)g.w * g.h * g.batch <= lim;
}

// nit (1..10) fused iterations, set cur -> cur^1, on register tiles.  variant < 0: the table default (MIFLOW_TILE_VARIANT).
int iterate_tile(int variant, int nit, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut, bool p_zero, int cur,
                 hipStream_t s)
{
    if (nit < 1 || nit > TILE_M) { set_error("register-tile kernel: %d iterations per launch (1..%d)", nit, TILE_M); return MI_ERR_BAD_ARG; }
    if (variant < 0) variant = auto_variant(g);
    if (variant < 0 || variant >= kTileVariants) { set_error("register-tile kernel: no variant %d", variant); return MI_ERR_BAD_ARG; }
    TileArgs A;
    memset(&A, 0, sizeof(A));
    A.pl = pl; A.g = g; A.l_t = l_t; A.theta = theta; A.taut = taut; A.cur = cur; A.nit = nit; A.swz = tuning().tile_swz;
    if (tuning().tb_verbose) {
        static int shown = 0;
        if (shown++ < 40)
            fprintf(stderr, "[tile] rw=%d nw=%d nit=%d %dx%d batch=%d\n", g_tile[variant].RW, g_tile[variant].NW, nit, g.w, g.h, g.batch);
    }
    if (pl.gamma != 0.f) {
        if (!g_tile[variant].launch_gam) variant = auto_variant(g) < 2 ? auto_variant(g) : 0;   // the channel exists on the two release shapes
        MI_REQUIRE(pl.u[0][2] && pl.u[1][2] && pl.p[0][4] && pl.p[0][5] && pl.p[1][4] && pl.p[1][5], MI_ERR_BAD_ARG, "gamma != 0 needs the u3 / p31 / p32 planes");
        return g_tile[variant].launch_gam(A, p_zero, s);
    }
    MI_REQUIRE(pl.g, MI_ERR_BAD_ARG, "the register-tile kernel needs the |grad|^2 plane");
    return g_tile[variant].launch(A, p_zero, s);
}

// One speculative step on register tiles (see iterate_tb_spec): T = the most iterations the launch may run (1..10).
int iterate_tile_spec(int T, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut, const Ctl &ctl, const SpecK &sk, int e0,
                      hipStream_t s)
{
    if (T < 1 || T > TILE_M) { set_error("register-tile kernel: block of %d iterations (1..%d)", T, TILE_M); return MI_ERR_BAD_ARG; }
    const int variant = auto_variant(g, true);
    TileArgs A;
    memset(&A, 0, sizeof(A));
    A.pl = pl; A.g = g; A.l_t = l_t; A.theta = theta; A.taut = taut; A.cur = 0; A.nit = T; A.swz = tuning().tile_swz;
    A.ctl = make_ctlk(&ctl);
    A.sk = sk;
    A.e0 = e0;
    const TileEntry &e = g_tile[variant < 2 ? variant : 0];
    if (pl.gamma != 0.f) return e.spec_gam(A, false, s);   // (margin 10 whatever the block length)
    return (T <= 4 ? g_tile[variant].spec4 : T <= 7 ? g_tile[variant].spec7 : g_tile[variant].spec)(A, false, s);
}

}  // namespace tvl1
}  // namespace mi
