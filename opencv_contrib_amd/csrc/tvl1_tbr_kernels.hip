// Dual TV-L1 -- temporally blocked fused iteration, ROTATING-SLOT formulation (gfx950, wave64).
//
// T iteration levels = T pipeline stages, one image row apart (dependency cone +-1 px per iteration), all state in VGPRs,
// static planes on a per-wave LDS ring, x-neighbours by DPP.  The stage is written as an IN-PLACE update so that no
// state is ever copied and no border handling needs a branch:
//
//   stage t sees   A = (u_(t-1), p_(t-1)) of row a      (its input)        B = (u_t, p_(t-1)) of row a-1   (its held state)
//   and leaves     A = (u_t,     p_(t-1)) of row a      (its NEW state)    B = (u_t, p_t)     of row a-1   (its output)
//
// i.e. only A.u and B.p are overwritten, and the two register sets swap roles.  A stage's output set is the next stage's
// input set, so after one pipeline step every held state has moved by one register set: with P = T + 1 + PF sets
// (T held states + the row in flight + PF prefetched rows) and the row loop unrolled P times, set indices are compile-time
// constants and the assignment returns to itself at the loop back-edge -- no v_mov, no ping-pong copy of the state
// (6 registers per stage and pixel), prefetched rows land directly in the set that consumes them.
//
// Borders without branches (profiles/r01p: the five wave-uniform border branches per stage cost 16-25 %):
//   * left:   strip 0 starts at x = 0 in lane 0, so the zero fill of `wave_shr:1 bound_ctrl` IS the missing p(x-1) term
//             (optflow/src/tvl1flow.cpp:893-894);
//   * right:  forward x-difference forced to 0 by a per-lane select (:833-838);
//   * top:    the p(y-1) term of row 0 is cut at the consumer, div = dx + fma(-m1, p12(a-1), p12(a)), m1 = (a != 0) (:889-890);
//   * bottom: the forward y-difference of row H-1 is multiplied by m2 = (a != H) (:826-831).
// Rows above / below the image and columns right of it are loaded from clamped addresses (finite data) and are isolated
// from the valid region by exactly these four cuts, so loads are unconditional and nothing is masked.
//
// Arithmetic: fast-math form (v_rcp / v_sqrt / fma; TH written as a clamp); parity against the oracle and the exact
// kernel is tested with a stated tolerance (tests/test_tvl1_gpu.py).
#include "tvl1_tb_dev.h"
#include "tvl1_warp_px.h"
#include <utility>
#include <cstdlib>
#include <cstdio>
#include <vector>

namespace mi {
namespace tvl1 {

template <int PPL, bool GAM = false>
struct Slot {
    Dyn<PPL> d;
    Stat<PPL> s;
};
template <int PPL>
struct Slot<PPL, true> {   // gamma != 0: the illumination channel's third component (GAM kernels)
    Dyn<PPL> d;
    Stat<PPL> s;
    Dyn3<PPL> g;
};
// the third component of a register set, or nullptr in a kernel without the channel (the stage never dereferences it there)
template <int PPL> __device__ __forceinline__ Dyn3<PPL> *g3(Slot<PPL, true> &x) { return &x.g; }
template <int PPL> __device__ __forceinline__ Dyn3<PPL> *g3(Slot<PPL, false> &) { return nullptr; }

// In-place stage.  negm1 = -(a != 0), m2 = (a != H), taum2 = taut * m2 are wave-uniform scalars.
// ERR: also accumulate this iteration level's convergence error sum(du1^2 + du2^2) (optflow/src/tvl1flow.cpp:1096-1112 ==
// cuda tvl1flow.cu:276-283) as INTEGERS: every pixel's term is rounded to 2^-24 px^2 and added into a 64-bit per-lane
// accumulator, so the sum does not depend on how rows are split into bands or pairs into batches and lanes -- a pair stops
// after the same iteration whatever batch it is computed in.  es = 2^24 for rows inside the wave's band, 0 for its halo rows
// (the scale doubles as the row mask); ownership of the column is applied once, when the accumulators are reduced.
// JW (joined waves, see k_iterate_tbr): the lane without a source lane takes what the neighbouring wave handed over -- (xl1, xl2) =
// p11, p21 of the column left of lane 0, (xr1, xr2) = u1, u2 of the column right of lane 63 -- instead of the zero fill.
// MK = false (JW = 4, interior blocks): no stage of the block sees row 0 or row H, so the three border scalars are the constants
// -1, 1 and taut and the masked forms reduce to the same operations without them -- fma(-1, b, a) and a - b, fma(x, 1, y) and
// x + y round once, identically: the planes are bit-identical to the masked form's.
// GAM (round 6): the illumination channel of gamma != 0 rides along -- A3 / B3 are the u3, p31, p32 of the same two register sets, G.gamma
// the weight, G.eu3 = 1 where the convergence error includes (du3)^2 (CPU class, optflow/src/tvl1flow.cpp:1110) and 0 under cv::cuda's
// rule (tvl1flow.cu:276-283), (xl3, xr3) the hand-over values of the joined form.  The threshold test keeps |grad|^2 = I1wx^2 + I1wy^2
// (both references: no gamma^2 term), so fi is shared by the three components.
struct GamK { float gamma, eu3, xl3, xr3; };
template <int PPL, bool ERR, int JW = 0, bool MK = true, bool GAM = false>
__device__ __forceinline__ void stage_r(Dyn<PPL> &A, Dyn<PPL> &B, const Stat<PPL> &st, const bool right_ok[PPL], float negm1,
                                        float m2, float taum2, float l_t, float theta, float taut, unsigned long long &acc, float es,
                                        float xl1 = 0.f, float xl2 = 0.f, float xr1 = 0.f, float xr2 = 0.f, Dyn3<PPL> *A3 = nullptr,
                                        Dyn3<PPL> *B3 = nullptr, const GamK G = GamK{0.f, 0.f, 0.f, 0.f})
{
    float dx1[PPL], dx2[PPL], dx3[PPL];
    dx1[0] = A.p11[0] - (JW ? dpp_from_prev_fill(A.p11[PPL - 1], xl1) : dpp_from_prev(A.p11[PPL - 1]));
    dx2[0] = A.p21[0] - (JW ? dpp_from_prev_fill(A.p21[PPL - 1], xl2) : dpp_from_prev(A.p21[PPL - 1]));
    if constexpr (GAM) dx3[0] = A3->p31[0] - (JW ? dpp_from_prev_fill(A3->p31[PPL - 1], G.xl3) : dpp_from_prev(A3->p31[PPL - 1]));
#pragma unroll
    for (int j = 1; j < PPL; ++j) {
        dx1[j] = A.p11[j] - A.p11[j - 1]; dx2[j] = A.p21[j] - A.p21[j - 1];
        if constexpr (GAM) dx3[j] = A3->p31[j] - A3->p31[j - 1];
    }
    const float r1 = JW ? dpp_from_next_fill(B.u1[0], xr1) : dpp_from_next(B.u1[0]);
    const float r2 = JW ? dpp_from_next_fill(B.u2[0], xr2) : dpp_from_next(B.u2[0]);
    float r3 = 0.f;
    if constexpr (GAM) r3 = JW ? dpp_from_next_fill(B3->u3[0], G.xr3) : dpp_from_next(B3->u3[0]);
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        // ---- u_t(a)   (optflow/src/tvl1flow.cpp:989-1041, 1096-1112; TH written as clamp(-rho/grad, +-l_t))
        const float div1 = dx1[j] + (MK ? fmaf(negm1, B.p12[j], A.p12[j]) : A.p12[j] - B.p12[j]);
        const float div2 = dx2[j] + (MK ? fmaf(negm1, B.p22[j], A.p22[j]) : A.p22[j] - B.p22[j]);
        float rho0 = st.rc[j];
        if constexpr (GAM) rho0 = fmaf(G.gamma, A3->u3[j], rho0);   // optflow tvl1flow.cpp:1011 == tvl1flow.cu:233
        const float rho = fmaf(st.ix[j], A.u1[j], fmaf(st.iy[j], A.u2[j], rho0));
        const float fi = __builtin_amdgcn_fmed3f(-rho * st.rg[j], -l_t, l_t);
        const float nu1 = fmaf(theta, div1, fmaf(fi, st.ix[j], A.u1[j]));
        const float nu2 = fmaf(theta, div2, fmaf(fi, st.iy[j], A.u2[j]));
        // ---- p_t(a-1)  (:1140-1181)
        const float n1 = (j + 1 < PPL) ? B.u1[j + 1 < PPL ? j + 1 : j] : r1;
        const float n2 = (j + 1 < PPL) ? B.u2[j + 1 < PPL ? j + 1 : j] : r2;
        // (r19b: the cut as a full-rate multiply by a 0 / 1 lane mask instead of the half-rate v_cndmask changes nothing: 1 414 against 1 416 pairs/s)
        const float u1x = right_ok[j] ? n1 - B.u1[j] : 0.f;
        const float u2x = right_ok[j] ? n2 - B.u2[j] : 0.f;
        const float d1 = nu1 - B.u1[j];
        const float d2 = nu2 - B.u2[j];
        const float g1 = __builtin_amdgcn_sqrtf(MK ? fmaf(d1 * d1, m2, u1x * u1x) : d1 * d1 + u1x * u1x);
        const float g2 = __builtin_amdgcn_sqrtf(MK ? fmaf(d2 * d2, m2, u2x * u2x) : d2 * d2 + u2x * u2x);
        const float q1 = __builtin_amdgcn_rcpf(fmaf(taut, g1, 1.0f));
        const float q2 = __builtin_amdgcn_rcpf(fmaf(taut, g2, 1.0f));
        B.p11[j] = fmaf(taut, u1x, B.p11[j]) * q1;
        B.p12[j] = fmaf(MK ? taum2 : taut, d1, B.p12[j]) * q1;
        B.p21[j] = fmaf(taut, u2x, B.p21[j]) * q2;
        B.p22[j] = fmaf(MK ? taum2 : taut, d2, B.p22[j]) * q2;
        float e3sq = 0.f;
        if constexpr (GAM) {
            // the same update for (u3, p31, p32): d3 = fi * gamma (:1020-1033 == tvl1flow.cu:243-262), the dual step :1172-1178 == :341-346
            const float div3 = dx3[j] + (MK ? fmaf(negm1, B3->p32[j], A3->p32[j]) : A3->p32[j] - B3->p32[j]);
            const float nu3 = fmaf(theta, div3, fmaf(fi, G.gamma, A3->u3[j]));
            const float n3 = (j + 1 < PPL) ? B3->u3[j + 1 < PPL ? j + 1 : j] : r3;
            const float u3x = right_ok[j] ? n3 - B3->u3[j] : 0.f;
            const float d3 = nu3 - B3->u3[j];
            const float g3 = __builtin_amdgcn_sqrtf(MK ? fmaf(d3 * d3, m2, u3x * u3x) : d3 * d3 + u3x * u3x);
            const float q3 = __builtin_amdgcn_rcpf(fmaf(taut, g3, 1.0f));
            B3->p31[j] = fmaf(taut, u3x, B3->p31[j]) * q3;
            B3->p32[j] = fmaf(MK ? taum2 : taut, d3, B3->p32[j]) * q3;
            if (ERR) { const float e3 = nu3 - A3->u3[j]; e3sq = e3 * e3 * G.eu3; }
            A3->u3[j] = nu3;
        }
        if (ERR) {
            const float e1 = nu1 - A.u1[j], e2 = nu2 - A.u2[j];
            const float et = GAM ? fmaf(e1, e1, e2 * e2) + e3sq : fmaf(e1, e1, e2 * e2);
            acc += (unsigned long long)__float2uint_rn(et * es);   // v_cvt_u32_f32 saturates: a term >= 256 px^2 only under-counts
        }
        A.u1[j] = nu1;
        A.u2[j] = nu2;
        // tie the accumulator update into the stage's dependency chain: left free, the scheduler defers the 130 low-priority
        // accumulations of the unrolled block to its end and keeps their inputs alive (240 spilled registers measured)
        if (ERR) {
            unsigned lo = (unsigned)acc, hi = (unsigned)(acc >> 32);
            asm volatile("" : "+v"(lo), "+v"(hi), "+v"(A.u1[j]));
            acc = ((unsigned long long)hi << 32) | lo;
        }
    }
}

// The same stage in EXACT math (MODE 2): separately rounded binary32 operations in the CPU reference's order, IEEE division,
// the dual update's norm as a double-precision square root -- the expressions of k_iterate<EXACT> (tvl1_kernels.hip:
// px_update_u / px_update_p / row_update_u; optflow/src/tvl1flow.cpp:857-899, 989-1041, 1096-1112, 1140-1181), so a block of T
// fused iterations is BIT-IDENTICAL to T one-iteration launches.  The border cuts are the reference's case distinctions (they
// pick the association order of the divergence), evaluated as selects: top = (a == 0) and bottom = (a == H) are wave-uniform,
// x0 marks the lane holding column 0.  st.rg is the RAW |grad|^2 here (finish_static is skipped).  PPL = 1 only.
__device__ __forceinline__ void stage_r_exact(Dyn<1> &A, Dyn<1> &B, const Stat<1> &st, bool right_ok, bool x0, bool top, bool bottom,
                                              float l_t, float theta, float taut)
{
    const float p11l = dpp_from_prev(A.p11[0]), p21l = dpp_from_prev(A.p21[0]);
    const float r1 = dpp_from_next(B.u1[0]), r2 = dpp_from_next(B.u2[0]);
    const float p11 = A.p11[0], p12 = A.p12[0], p21 = A.p21[0], p22 = A.p22[0];
    const float up12 = B.p12[0], up22 = B.p22[0];
    // divergence, optflow/src/tvl1flow.cpp:857-899
    float d1, d2;
    {
        const float a1 = x0 ? p11 : p11 - p11l, a2 = x0 ? p21 : p21 - p21l;       // x > 0: (p11 - p11l) first in both row cases
        const float in1 = (p11 - p11l) + (p12 - up12), in2 = (p21 - p21l) + (p22 - up22);   // y > 0, x > 0
        const float le1 = p11 + p12 - up12, le2 = p21 + p22 - up22;               // y > 0, x == 0
        const float tp1 = a1 + p12, tp2 = a2 + p22;                                // y == 0: p11 - p11l + p12 / p11 + p12
        d1 = top ? tp1 : (x0 ? le1 : in1);
        d2 = top ? tp2 : (x0 ? le2 : in2);
    }
    // u_t(a): estimateV + estimateU, :989-1041, 1096-1112
    const float ix = st.ix[0], iy = st.iy[0], g = st.rg[0], rc = st.rc[0], u1 = A.u1[0], u2 = A.u2[0];
    const float rho = rc + (ix * u1 + iy * u2);
    const float ltg = l_t * g;
    float e1 = 0.f, e2 = 0.f;
    if (rho < -ltg) { e1 = l_t * ix; e2 = l_t * iy; }
    else if (rho > ltg) { e1 = -l_t * ix; e2 = -l_t * iy; }
    else if (g > 1.1920928955078125e-7f) { const float fi = -rho / g; e1 = fi * ix; e2 = fi * iy; }
    const float v1 = u1 + e1, v2 = u2 + e2;
    const float nu1 = v1 + theta * d1, nu2 = v2 + theta * d2;
    // p_t(a-1): forward differences of u_t with the right / bottom cuts (:826-838), dual update (:1140-1181)
    const float u1x = right_ok ? r1 - B.u1[0] : 0.f, u2x = right_ok ? r2 - B.u2[0] : 0.f;
    const float u1y = bottom ? 0.f : nu1 - B.u1[0], u2y = bottom ? 0.f : nu2 - B.u2[0];
    {
        const float gn = (float)sqrt((double)u1x * (double)u1x + (double)u1y * (double)u1y);
        const float ng = 1.0f + taut * gn;
        B.p11[0] = (B.p11[0] + taut * u1x) / ng;
        B.p12[0] = (B.p12[0] + taut * u1y) / ng;
    }
    {
        const float gn = (float)sqrt((double)u2x * (double)u2x + (double)u2y * (double)u2y);
        const float ng = 1.0f + taut * gn;
        B.p21[0] = (B.p21[0] + taut * u2x) / ng;
        B.p22[0] = (B.p22[0] + taut * u2y) / ng;
    }
    A.u1[0] = nu1;
    A.u2[0] = nu2;
}

// Row accesses as `global_* v, voffset, s[base]`: the row base (plane + row * ld) is wave-uniform and the lane offset is made
// opaque per row (empty asm), otherwise LICM re-associates base + lane offset into per-plane 64-bit VGPR addresses.
template <int PPL>
__device__ __forceinline__ void ldr(float dst[PPL], const float *rowp, unsigned xb)
{
    const char *q = reinterpret_cast<const char *>(rowp) + xb;
    if (PPL == 1) {
        dst[0] = *reinterpret_cast<const float *>(q);
    } else if (PPL == 2) {
        const float2 v = *reinterpret_cast<const float2 *>(q);
        dst[0] = v.x; dst[1] = v.y;
    } else {
        const float4 v = *reinterpret_cast<const float4 *>(q);
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
}
template <int PPL>
__device__ __forceinline__ void str(float *rowp, unsigned xb, const float v[PPL])
{
    char *q = reinterpret_cast<char *>(rowp) + xb;
    if (PPL == 1) *reinterpret_cast<float *>(q) = v[0];
    else if (PPL == 2) *reinterpret_cast<float2 *>(q) = make_float2(v[0], v[1]);
    else *reinterpret_cast<float4 *>(q) = make_float4(v[0], v[1], v[2], v[3]);
}

// NG (round 4): there is no |grad|^2 plane -- the warp did not store it and finish_static_ng forms it from the row's I1wx, I1wy when the
// row is consumed, with the warp's own expression (two separately rounded products and their sum: the same bits).  8 B per pixel and
// warp less through HBM; the step is power-limited and a byte costs ~13 f32 operations (profiles/r08/README.md).
// P16 (opt-in, MIFLOW_TB_P16=1; CHANGES RESULTS): between the passes of a scale the dual variable travels as signed 16-bit fixed point
// (|p| <= 1 by construction of the dual update; v_cvt_pknorm_i16_f32, step 2^-15 ~ 3e-5): {p11, p12} in plane 0, {p21, p22} in plane
// 2, 16 B per pixel and pass boundary less through HBM.  The raw dwords wait in the p11 / p21 registers of the set and are
// unpacked when the row enters the pipeline (unpack_p16).
template <int PPL, bool PZ, bool NG = false, bool P16 = false, int FW = 0, bool GAM = false>
__device__ __forceinline__ void load_row_r(Slot<PPL, GAM> &x, const TbArgs &A, const float *const u[3], const float *const p[6], int row,
                                           int H, unsigned xc)
{
    const long long ro = (long long)min(max(row, 0), H - 1) * A.g.ld;   // wave-uniform
    asm volatile("" : "+v"(xc));
    if (!FW) {   // FW: the row's statics come from the workgroup's producer wave through the LDS inbox (fw_inbox_get)
        ldr<PPL>(x.s.ix, A.pl.ix + ro, xc);
        ldr<PPL>(x.s.iy, A.pl.iy + ro, xc);
        if (!NG) ldr<PPL>(x.s.rg, A.pl.g + ro, xc);   // RAW |grad|^2 until the row is consumed (finish_static)
        ldr<PPL>(x.s.rc, A.pl.rc + ro, xc);
    }
    ldr<PPL>(x.d.u1, u[0] + ro, xc);
    ldr<PPL>(x.d.u2, u[1] + ro, xc);
    if (!PZ && P16) {
        ldr<PPL>(x.d.p11, p[0] + ro, xc);
        ldr<PPL>(x.d.p21, p[2] + ro, xc);
    } else if (!PZ) {
        ldr<PPL>(x.d.p11, p[0] + ro, xc);
        ldr<PPL>(x.d.p12, p[1] + ro, xc);
        ldr<PPL>(x.d.p21, p[2] + ro, xc);
        ldr<PPL>(x.d.p22, p[3] + ro, xc);
    } else {
#pragma unroll
        for (int j = 0; j < PPL; ++j) x.d.p11[j] = x.d.p12[j] = x.d.p21[j] = x.d.p22[j] = 0.f;
    }
    if constexpr (GAM) {
        ldr<PPL>(x.g.u3, u[2] + ro, xc);
        if (!PZ) {
            ldr<PPL>(x.g.p31, p[4] + ro, xc);
            ldr<PPL>(x.g.p32, p[5] + ro, xc);
        } else {
#pragma unroll
            for (int j = 0; j < PPL; ++j) x.g.p31[j] = x.g.p32[j] = 0.f;
        }
    }
}

// Wave-constant context of the row loop (all scalars after inlining).
template <int PPL>
struct CtxR {
    TbArgs B;                 // plane pointers already offset to the pair
    const float *uin[3], *pin[6];     // [2], [4], [5]: the illumination channel (GAM kernels)
    float *uout[3], *pout[6];
    float *ring;
    float *inbox;   // FW: this wave's inbox (FW_RING slots of {I1wx | I1wy | rho_c} x 64 columns), written by its producer wave
    int lane, H, ld, y0, y1, ystart, nsteps;
    unsigned xc;
    bool st_ok, x0;   // x0: this lane holds column 0 (MODE 2)
    bool right_ok[PPL];
    float l_t, theta, taut;
    float gamma, eu3;   // GAM: the channel's weight; 1 / 0 = the error sum includes (du3)^2 (CPU class) or not (cv::cuda)
    int nit;        // active stages (MODE 1: the length of the speculative block or of the replay; otherwise T)
};

// ---- joined waves (JW): LDS hand-over between the four waves of a workgroup that sit on four ADJACENT 64-column segments of one
// band.  A consumer wave owns an AREA of 2 buffers (step parity) x T slots (stage) x XS bytes:
//     +0  p11, p21 of the column left of its lane 0   (written by lane 63 of the wave to its left)   } input of stage t of
//     +8  u1, u2   of the column right of its lane 63 (written by lane 0 of the wave to its right)   } the step with that parity
//     +16 tag: low half = step + 1 of the left writer, high half = step + 1 of the right writer's TARGET step
// Dependences (tvl1_tbr_kernels.hip header): stage t of step n needs p_(t-1)(row a) of the left neighbour -- its stage t-1 of the
// SAME step (stage 0: the row it loaded) -- and u_t(row a-1) of the right neighbour -- its stage t of step n-1.  Every dependence
// points to a lexicographically smaller (step, stage), so waiting cannot cycle; the waves of a workgroup are co-resident, so
// waiting cannot starve.  In steady state a wave trails its left neighbour by one to T stages.  Two buffers suffice: the writer of
// a slot depends (through the opposite dependence) on the reader having passed the slot's previous use.  Data is written before
// its tag and read after it; LDS executes a wave's DS operations in order.
constexpr int XS = 32;   // bytes per slot: {p11, p21 | u1, u2 | tag | pad}
__host__ __device__ constexpr int xarea_bytes(int T) { return 2 * T * XS; }
__host__ __device__ constexpr int xdump_bytes(int T) { return 2 * T * XS + 64 * 8; }   // where the 63 non-publishing lanes write
// pointers into LDS keep their address space (a generic pointer makes every access a flat_load / flat_store)
#define MI_LDS __attribute__((address_space(3)))
typedef MI_LDS char *lds_ptr;
struct Xchg {
    // byte addresses in LDS, all for the CURRENT step's parity: own = the slots this wave consumes (wave-uniform, kept in a VGPR);
    // pub_r / pub_l = where a lane's hand-over to the right / left neighbour goes: the neighbour's area for the ONE lane that holds the
    // boundary column (lane 63 / lane 0), a per-lane dump address for all others -- so a publish is an ordinary full-wave store
    unsigned own, pub_r, pub_l;
    unsigned mul;               // 1 if a left neighbour feeds this wave | 0x10000 if a right one does: expected tag = (n + 1) * mul
    unsigned tag; float l1, l2, r1, r2;   // slot read ahead for the next stage
    int budget;                 // re-reads this wave may still spend waiting (all stages of the launch together)
    bool on_r, on_l;            // JW == 2: this LANE publishes to the right / left neighbour (lane 63 / lane 0 of a wave that has one)
};
__device__ __forceinline__ void xread(unsigned slot, unsigned &tag, float &l1, float &l2, float &r1, float &r2)
{
    const lds_ptr q = (lds_ptr)(unsigned long long)slot;
    tag = *reinterpret_cast<volatile MI_LDS unsigned *>(q + 16);          // the tag first: data is written before its tag
    const unsigned long long L = *reinterpret_cast<volatile MI_LDS unsigned long long *>(q);
    const unsigned long long R = *reinterpret_cast<volatile MI_LDS unsigned long long *>(q + 8);
    l1 = __uint_as_float((unsigned)L); l2 = __uint_as_float((unsigned)(L >> 32));
    r1 = __uint_as_float((unsigned)R); r2 = __uint_as_float((unsigned)(R >> 32));
}
// two dwords, then the 16-bit tag
__device__ __forceinline__ void xwrite(unsigned slot, int data_off, int tag_off, float a, float b, unsigned tag)
{
    const lds_ptr q = (lds_ptr)(unsigned long long)slot;
    *reinterpret_cast<volatile MI_LDS float *>(q + data_off) = a;
    *reinterpret_cast<volatile MI_LDS float *>(q + data_off + 4) = b;
    *reinterpret_cast<volatile MI_LDS unsigned short *>(q + tag_off) = (unsigned short)tag;
}
// JW == 2 (barrier form): the same hand-over without tags.  The four waves of a workgroup pass one s_barrier per stage, so that
// everything the left neighbour's stage t-1 published in this step (stage 0: the row that entered) is in LDS before stage t of any
// wave reads it; the right neighbour's u_t is that of the previous step (the other buffer) and long since there.  A slot is 16
// bytes {p11, p21, u1, u2}: one ds_read_b128 per stage, two exec-masked ds_write_b64 (only lane 63 / lane 0 publish: no dump area,
// 38.2 KB of LDS per workgroup = four workgroups per CU like the independent-wave kernel).  s_waitcnt lgkmcnt(0) + s_barrier, NOT
// __syncthreads: the latter also waits for the row prefetches and stores in flight (vmcnt).
constexpr int XS2 = 16;
// GAM: a slot is 32 bytes {p11, p21, u1, u2 | p31, u3, -, -}: the third component's two hand-over values behind the sixteen bytes of the
// two-channel form (one more ds_read_b64 per stage, one more exec-masked ds_write_b32 per publish)
__host__ __device__ constexpr int xs2_bytes(bool gam) { return gam ? 32 : XS2; }
// waves of a joined group: JW = 1, 2: four (a 256-column strip); JW = 3: eight (512 columns: 492 owned instead of 2 x 236), barrier form
__host__ __device__ constexpr int jw_waves(int JW) { return JW == 3 ? 8 : 4; }
// Stages per barrier interval.  With a constant SKEW of XK stages between neighbouring waves (wave w runs XK * w stages behind wave 0:
// it passes w barriers before its first step) a barrier every XK stages is enough: in global stage index g = step * T + stage, wave w
// executes g in interval floor(g / XK) + w; its left input comes from wave w-1's g - 1, executed in interval floor((g - 1) / XK) + w - 1
// -- always an earlier one --, its right input from wave w+1's g - T, executed in interval floor((g - T) / XK) + w + 1 -- earlier iff
// T >= 2 XK.  T = 10: XK = 5, two barriers per step instead of ten; T = 5: XK = 2, five per two steps.  A stage starts an interval when
// its global index is a multiple of XK; the step index enters through the compile-time phase k of the unrolled block (the block's
// first step n0 is a multiple of P, and P * T is a multiple of XK: checked in the kernel).
__host__ __device__ constexpr int xk_stages(int T) { return T >= 10 && T % 5 == 0 ? 5 : T >= 4 ? 2 : 1; }
__host__ __device__ constexpr int xarea2_bytes(int T, bool gam = false) { return 2 * T * xs2_bytes(gam); }
__device__ __forceinline__ void xbarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void xread2(unsigned slot, float &l1, float &l2, float &r1, float &r2)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = *reinterpret_cast<volatile MI_LDS f4 *>((lds_ptr)(unsigned long long)slot);
    l1 = v.x; l2 = v.y; r1 = v.z; r2 = v.w;
}
__device__ __forceinline__ void xwrite2(bool on, unsigned addr, float a, float b)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    if (on) {
        f2 v; v.x = a; v.y = b;
        *reinterpret_cast<volatile MI_LDS f2 *>((lds_ptr)(unsigned long long)addr) = v;
    }
}
// GAM: the third component's pair {p31 from the left | u3 from the right} at +16 of the slot
__device__ __forceinline__ void xread1g(unsigned slot, float &l3, float &r3)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = *reinterpret_cast<volatile MI_LDS f2 *>((lds_ptr)(unsigned long long)(slot + 16));
    l3 = v.x; r3 = v.y;
}
__device__ __forceinline__ void xwrite1(bool on, unsigned addr, float a)
{
    if (on) *reinterpret_cast<volatile MI_LDS float *>((lds_ptr)(unsigned long long)addr) = a;
}
// JW == 4: the publish as an ORDINARY full-wave store -- the one lane that holds the boundary column writes to the neighbour's slot,
// the other 63 to a dump area behind the hand-over areas (addresses prepared per lane once): no exec mask, no branch, so a whole
// pipeline step stays ONE basic block and the scheduler may overlap the tail of a stage with the head of the next.
__device__ __forceinline__ void xwrite2f(unsigned addr, float a, float b)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v; v.x = a; v.y = b;
    *reinterpret_cast<volatile MI_LDS f2 *>((lds_ptr)(unsigned long long)addr) = v;
}
__host__ __device__ constexpr int xdump4_bytes(int T) { return 64 * 8 + 2 * xarea2_bytes(T) + 64; }   // per workgroup, shared by its waves
__host__ __device__ constexpr bool jw_fast(int JW) { return JW == 4; }
__device__ int g_jw_fault;   // sticky: a bounded wait on a neighbouring wave ran out (never expected; results are then invalid)

// ---- fused warp (FW; round 5): the warp INSIDE the pass kernel -- cudaoptflow/src/cuda/tvl1flow.cu:89-164 feeding :187-348 (CPU class:
// optflow/src/tvl1flow.cpp:904-969 feeding :989-1181) without the HBM round trip of I1wx, I1wy, rho_c (12 B/px written by the warp
// launch + 12 B/px read back by the pass, and the warp launch's own 8 B/px of flow).  A workgroup is the four joined CONSUMER waves of
// the barrier form (JW = 2) plus four PRODUCER waves, one per consumer: producer j walks the rows of the band two rows ahead of the
// pipeline and computes, for the 64 columns of consumer j, what k_warp6 computes for a pixel -- the same routines (tvl1_warp_px.h), the
// same operations in the same order: the planes the consumer sees are bit-identical to the stored ones -- and writes the three values
// into consumer j's LDS INBOX.  Eight waves of 128 VGPRs = two workgroups per CU: every SIMD runs two consumers and two producers.
//
// Synchronisation rides on the barriers the joined waves pass anyway (two per step at T = 10): all eight waves execute the same
// sequence of s_barrier.  With global barrier numbers counted from the extra one that opens the pipeline: producer step n = barriers
// 2n, 2n+1, consumer w's step n = barriers 2n+w, 2n+w+1 (the skew of the hand-over scheme).  The producer finishes row n+2 in its step
// n, before barrier 2n+2; consumer w reads row n+1 after its barrier 2n+w >= 2n (written before barrier 2n) -- one step ahead of the use,
// into the register set of that row; a slot of the four-row ring is overwritten (row n+2 over row n-2) after barrier 2n, and its last
// reader (consumer 3 at its step n-3) finished before barrier 2n-2.  Rows above / below the image and columns right of it take the
// clamped pixel's values -- what the clamped loads of the unfused kernel deliver; they are cut off from the image by the same masks.
//
// The producer's own latency is hidden by software pipelining: the window gathers of row m+2 and the flow / I0 loads of row m+4 are
// issued when row m is finished, and consumed two whole pipeline steps later.
constexpr int FW_RING = 4;         // inbox slots (rows) per consumer wave: two rows of lead + the two steps of consumer skew
constexpr int FW_SLOT = 3 * 64;    // floats per slot
__host__ __device__ constexpr int fw_lds_floats(int NW) { return NW * FW_RING * FW_SLOT + 128; }   // inboxes + the cubic phase table
template <int PPL>
__device__ __forceinline__ void fw_inbox_get(const float *slot, int lane, Stat<PPL> &s)
{
    const float *q = slot + lane;   // (one pixel per lane: the kernel asserts it)
    s.ix[0] = q[0]; s.iy[0] = q[64]; s.rc[0] = q[128];
}
template <int FW> struct FwSem { static constexpr int sem = FW == 2 ? MI_SEM_CUDA_COMPAT : MI_SEM_CPU_REF; static constexpr bool fast = FW == 2; };

#ifndef FW_STAGE
#define FW_STAGE 0   // 0 (default): the producers gather their windows (fw_produce); 1: they read them from a private LDS band of I1 (fw_produce_staged) -- bit-identical, measured no faster (r14g, r14h: 1 093 - 1 180 against 1 205 pairs/s): the fused form is bound by VALU issue, not by the gathers
#endif
#ifndef FW_X
#define FW_X 0   // experiments build only: 1 = the producers write zeros (no loads, no arithmetic), 2 = arithmetic without loads -- wrong results
#endif
// A producer wave keeps FOUR rows in flight, all in registers with compile-time indices (the row loop is unrolled four times, so nothing
// is ever copied: a copy of a register a load is still filling would make the wave wait for that load at once): the flow and I0 of rows
// m .. m+3 in U[m & 3], the windows of rows m, m+1 in R[m & 1].  When row m is finished its window registers take the gathers of row
// m+2 (whose flow arrived two steps ago) and its flow registers the loads of row m+4.
struct FwU { float u1, u2, i0; };

template <int FW, int NW>
__device__ __forceinline__ void fw_produce(const TbArgs &A, float *inbox, const float *tab, int lane, int xw, int ystart, int nsteps_total,
                                           long long pb, int cur)
{
    constexpr int SEM = FwSem<FW>::sem;
    constexpr bool FAST = FwSem<FW>::fast;
    const int W = A.g.w, H = A.g.h, ld = A.g.ld;
    const int x = min(xw + lane, W - 1);
    const float *U1 = A.pl.u[cur][0] + pb + x, *U2 = A.pl.u[cur][1] + pb + x, *I0 = A.fI0 + pb + x, *P = A.fI1 + pb;
    FwU U[4];
    float R[2][6][6];
    const auto load_u = [&](int m, FwU &u) {
        const long long ro = (long long)min(max(ystart + m, 0), H - 1) * ld;   // wave-uniform
#if FW_X == 0
        u.u1 = U1[ro]; u.u2 = U2[ro]; u.i0 = I0[ro];
#else
        u.u1 = u.u2 = 0.25f; u.i0 = (float)ro;
#endif
    };
    const auto gather = [&](int m, const FwU &u, float (&Rw)[6][6]) {   // window of row m, origin from its flow
        int sx, sy;
        float wx[4], wy[4];
        warp_coords<SEM>(tab, x, min(max(ystart + m, 0), H - 1), u.u1, u.u2, sx, sy, wx, wy);
#if FW_X == 0
        // UNCONDITIONAL: a lane whose window touches the border (it takes window_border when the row is finished) gathers the nearest
        // interior window instead and ignores it.  With the gathers under a branch the compiler cannot count the loads in flight and
        // waits for nearly all of them at the head of every row (measured: the kernel then runs as slowly as warp + pass in sequence).
        window_gather(Rw, P, ld, min(max(sx, 1), W - 5), min(max(sy, 1), H - 5));
#else
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Rw[r][c] = u.u1 + (float)(r * 6 + c);
#endif
    };
    // row m: sums, rho_c, hand-over (window origin and weights are a pure function of the pixel and its flow: recomputed here instead of
    // being held for two steps -- twenty registers for twenty operations)
    const auto sums = [&](int m, const FwU &u, const float (&Rw)[6][6]) {
        float v0, v1, v2;
#if FW_X == 1
        v0 = v1 = v2 = 0.f;
#else
        int sx, sy;
        float wx[4], wy[4];
        warp_coords<SEM>(tab, x, min(max(ystart + m, 0), H - 1), u.u1, u.u2, sx, sy, wx, wy);
#if FW_X == 2
        window_sums<SEM, FAST>(Rw, wx, wy, v0, v1, v2);
#else
        if (window_interior(sx, sy, W, H)) window_sums<SEM, FAST>(Rw, wx, wy, v0, v1, v2);
        else window_border<SEM>(P, W, H, ld, sx, sy, wx, wy, v0, v1, v2);
#endif
#endif
        float *q = inbox + (m & (FW_RING - 1)) * FW_SLOT + lane;
        q[0] = v1; q[64] = v2;
        q[128] = (v0 - v1 * u.u1 - v2 * u.u2 - u.i0);   // calcGradRho, optflow/src/tvl1flow.cpp:918-944 == tvl1flow.cu:151-163 (warp_px)
    };
#if FW_X == 1
#define FW_ROW(m, k) do { sums((m), U[(k) & 3], R[(k) & 1]); } while (0)
#else
#define FW_ROW(m, k) do { sums((m), U[(k) & 3], R[(k) & 1]); gather((m) + 2, U[((k) + 2) & 3], R[(k) & 1]); load_u((m) + 4, U[(k) & 3]); } while (0)
#endif
    load_u(0, U[0]); load_u(1, U[1]); load_u(2, U[2]); load_u(3, U[3]);
    gather(0, U[0], R[0]); gather(1, U[1], R[1]);
    FW_ROW(0, 0);
    FW_ROW(1, 1);
    xbarrier();   // opens the pipeline: rows 0 and 1 are in the inbox
    // producer step n finishes row n + 2; four steps per trip (row n + 2 + k lives in U[(k + 2) & 3], R[k & 1]).  The trip is
    // UNCONDITIONAL -- every path through the loop issues the same loads in the same order, so the compiler knows how many are in flight
    // behind the one it waits for; the consumers' step count is a multiple of 13, the one to three steps left over run after the loop
    int n = 0;
#pragma unroll 1
    for (; n + 4 <= nsteps_total; n += 4) {
        xbarrier(); FW_ROW(n + 2, 2); xbarrier();
        xbarrier(); FW_ROW(n + 3, 3); xbarrier();
        xbarrier(); FW_ROW(n + 4, 0); xbarrier();
        xbarrier(); FW_ROW(n + 5, 1); xbarrier();
    }
    if (n < nsteps_total) { xbarrier(); FW_ROW(n + 2, 2); xbarrier(); }
    if (n + 1 < nsteps_total) { xbarrier(); FW_ROW(n + 3, 3); xbarrier(); }
    if (n + 2 < nsteps_total) { xbarrier(); FW_ROW(n + 4, 0); xbarrier(); }
#undef FW_ROW
    for (int i = 0; i < NW - 1; ++i) xbarrier();   // the consumers' skew
}

// ---- the producer with its OWN staged band of I1 (round 5, second form).  The gathering producer above is bound by its two windows in
// flight (Little's law: 20 gather instructions per wave against ~2 us of latency under load): measured 1 205 pairs/s against 1 527 for the
// same arithmetic without loads.  Here a producer wave keeps a private LDS ring of FWS_NR rows x FWS_CW columns of I1 around the place its
// 64 columns look at: the columns of the wave shifted by the band's mean flow (ox, oy: measured once, on the band's middle row) +- FWS_D1,
// the rows yu + oy - FWS_D2 - 2 .. yu + oy + FWS_D2 + 3 of output row yu.  One new row enters per step, streamed four steps ahead with two
// coalesced dword loads per lane (two registers per row in flight instead of thirty-two), and a pixel's window is thirty-two ds_read_b32
// with immediate offsets.  I1 is read from HBM once per strip (+ the column margins) instead of six times through L2.  A lane whose window
// leaves the staged band (flow further than FWS_D from the band's mean, e.g. across a motion boundary) gathers it from global memory as
// before, synchronously: same values, same sums -- bit-identical whichever path a pixel takes.
constexpr int FWS_D1 = 5, FWS_D2 = 8;
constexpr int FWS_CW = 64 + 2 * FWS_D1 + 6;    // 80 columns (with 88 the workgroup needs exactly 80 KB of LDS and only one fits a CU)
constexpr int FWS_NR = 2 * FWS_D2 + 6;         // 22 rows: the six window rows of every flow within +- FWS_D2 of the band's mean
__host__ __device__ constexpr int fws_lds_floats(int NW) { return NW * FWS_NR * FWS_CW; }

template <int FW, int NW>
__device__ __forceinline__ void fw_produce_staged(const TbArgs &A, float *inbox, const float *tab, float *ring, int lane, int xw, int ystart,
                                                  int nsteps_total, int y_mid, long long pb, int cur)
{
    constexpr int SEM = FwSem<FW>::sem;
    constexpr bool FAST = FwSem<FW>::fast;
    const int W = A.g.w, H = A.g.h, ld = A.g.ld;
    const int x = min(xw + lane, W - 1);
    const float *U1 = A.pl.u[cur][0] + pb + x, *U2 = A.pl.u[cur][1] + pb + x, *I0 = A.fI0 + pb + x, *P = A.fI1 + pb;
    // the band's mean flow, rounded: wave-uniform offsets of the staged band
    int ox, oy;
    {
        const long long ro = (long long)min(max(y_mid, 0), H - 1) * ld;
        float a = U1[ro], b = U2[ro];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        ox = __builtin_amdgcn_readfirstlane(__float2int_rn(fminf(fmaxf(a * (1.0f / 64.0f), -30000.f), 30000.f)));
        oy = __builtin_amdgcn_readfirstlane(__float2int_rn(fminf(fmaxf(b * (1.0f / 64.0f), -30000.f), 30000.f)));
    }
    const int cb = xw + ox - FWS_D1 - 3;                                     // image column of ring column 0
    // the two columns this lane streams per row: cb + lane and (lanes < FWS_CW - 64) cb + 64 + lane, clamped into the image
    const int ca = min(max(cb + lane, 0), W - 1), cbb = min(max(cb + 64 + lane, 0), W - 1);
    const bool second = lane < FWS_CW - 64;
    FwU U[4];
    float ST[4][2];
    const auto load_u = [&](int m, FwU &u) {
        const long long ro = (long long)min(max(ystart + m, 0), H - 1) * ld;   // wave-uniform
        u.u1 = U1[ro]; u.u2 = U2[ro]; u.i0 = I0[ro];
    };
    // row of I1 entering the band at output row index m: rtop(m) = ystart + m + oy + FWS_D2 + 3; ring slot of image row r = r mod FWS_NR
    // kept incrementally (wave-uniform): slot_top = slot of rtop(m)
    const auto load_row = [&](int r, float (&st)[2]) {
        const float *q = P + (long long)min(max(r, 0), H - 1) * ld;
        st[0] = q[ca];
        st[1] = q[cbb];   // (lanes >= 24 load a duplicate they never store: one instruction for the wave either way)
    };
    const auto store_row = [&](int slot, const float (&st)[2]) {
        float *q = ring + slot * FWS_CW + lane;
        q[0] = st[0];
        if (second) q[64] = st[1];
    };
    int rlow = ystart + oy - FWS_D2 - 2;                   // image row of the band's first row at m = 0
    int slot_low = ((rlow % FWS_NR) + FWS_NR) % FWS_NR;    // its ring slot
    // prologue: the 22 rows of m = 0 (rlow .. rlow + 21), four at a time through ST
    for (int r0 = 0; r0 < FWS_NR; r0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (r0 + k < FWS_NR) load_row(rlow + r0 + k, ST[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (r0 + k < FWS_NR) { int sl = slot_low + r0 + k; sl -= sl >= FWS_NR ? FWS_NR : 0; store_row(sl, ST[k]); }
    }
    // rows entering at m = 1 .. 4 in flight (the row entering at step m is rlow + m + FWS_NR - 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) load_row(rlow + FWS_NR + k, ST[k]);   // ST[k]: enters at m = k + 1
    load_u(0, U[0]); load_u(1, U[1]); load_u(2, U[2]); load_u(3, U[3]);

    // one output row: m = its index, u = its flow / I0, slot_lo = ring slot of image row rlow_m = ystart + m + oy - FWS_D2 - 2
    const auto row = [&](int m, const FwU &u, int rlow_m, int slot_lo) {
        const int yu = ystart + m;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, rc = 0.f;
        if (yu >= 0 && yu < H) {   // wave-uniform; rows outside the image hand over zeros: they are cut off from the image by the masks
            int sx, sy;
            float wx[4], wy[4];
            warp_coords<SEM>(tab, x, yu, u.u1, u.u2, sx, sy, wx, wy);
            const int dc = sx - 1 - cb, dr = sy - 1 - rlow_m;
            const bool inter = window_interior(sx, sy, W, H);
            const bool staged = inter && (unsigned)dc <= (unsigned)(FWS_CW - 6) && (unsigned)dr <= (unsigned)(FWS_NR - 6);
            if (staged) {
                float Rw[6][6];
                unsigned t = (unsigned)(slot_lo + dr);
                t = min(t, t - (unsigned)FWS_NR);   // wrap: t < NR stays (t - NR is huge as unsigned), otherwise t - NR
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const float *q = ring + t * FWS_CW + dc;
#pragma unroll
                    for (int c = 0; c < 6; ++c)
                        if (!((r == 0 || r == 5) && (c == 0 || c == 5))) Rw[r][c] = q[c];
                    ++t;
                    t = min(t, t - (unsigned)FWS_NR);
                }
                window_sums<SEM, FAST>(Rw, wx, wy, v0, v1, v2);
            } else if (inter) {
                float Rw[6][6];
                window_gather(Rw, P, ld, sx, sy);
                window_sums<SEM, FAST>(Rw, wx, wy, v0, v1, v2);
            } else {
                window_border<SEM>(P, W, H, ld, sx, sy, wx, wy, v0, v1, v2);
            }
            rc = (v0 - v1 * u.u1 - v2 * u.u2 - u.i0);   // calcGradRho, optflow/src/tvl1flow.cpp:918-944 == tvl1flow.cu:151-163 (warp_px)
        }
        float *q = inbox + (m & (FW_RING - 1)) * FW_SLOT + lane;
        q[0] = v1; q[64] = v2; q[128] = rc;
    };
    // step of output row m (k = m & 3): [m >= 1: the row that enters the band replaces the one that left] -> row -> next requests
    // (unconditional: every path through the row loop issues the same loads in the same order -- see fw_produce)
#define FWS_STEP(m, k)                                                                               \
    do {                                                                                             \
        store_row(slot_low, ST[((k) + 3) & 3]);   /* the slot of the row that just left the band */  \
        asm volatile("" ::: "memory");                                                               \
        load_row(rlow + FWS_NR + 4, ST[((k) + 3) & 3]);   /* enters four steps from now */           \
        ++rlow;                                                                                      \
        slot_low = slot_low + 1 == FWS_NR ? 0 : slot_low + 1;                                        \
        row((m), U[(k) & 3], rlow, slot_low);                                                        \
        load_u((m) + 4, U[(k) & 3]);                                                                 \
    } while (0)
    row(0, U[0], rlow, slot_low);
    load_u(4, U[0]);
    FWS_STEP(1, 1);
    xbarrier();   // opens the pipeline: rows 0 and 1 are in the inbox
    int n = 0;
#pragma unroll 1
    for (; n + 4 <= nsteps_total; n += 4) {
        xbarrier(); FWS_STEP(n + 2, 2); xbarrier();
        xbarrier(); FWS_STEP(n + 3, 3); xbarrier();
        xbarrier(); FWS_STEP(n + 4, 0); xbarrier();
        xbarrier(); FWS_STEP(n + 5, 1); xbarrier();
    }
    if (n < nsteps_total) { xbarrier(); FWS_STEP(n + 2, 2); xbarrier(); }
    if (n + 1 < nsteps_total) { xbarrier(); FWS_STEP(n + 3, 3); xbarrier(); }
    if (n + 2 < nsteps_total) { xbarrier(); FWS_STEP(n + 4, 0); xbarrier(); }
#undef FWS_STEP
    for (int i = 0; i < NW - 1; ++i) xbarrier();   // the consumers' skew
}

// Pipeline step with phase k (= step index mod P): every register-set index below is a compile-time constant.
// No early exit inside the unrolled block (an exit per step keeps every register set alive across P merge points): the last
// block may run up to P-1 steps past the band end; those rows are clamped loads whose results are never stored.
template <int T, int PPL, bool PZ, int PF, int MODE, int JW, bool MK, bool NG, bool P16, int FW, bool GAM, int k>
__device__ __forceinline__ void step_r(const CtxR<PPL> &c, Slot<PPL, GAM> (&X)[T + 1 + PF], int n0, int &slot0, unsigned long long (&acc)[T], Xchg &x)
{
    constexpr bool JF = jw_fast(JW);
    constexpr int XSg = xs2_bytes(GAM);          // bytes per hand-over slot and parity (barrier forms)
    constexpr int XAg = xarea2_bytes(T, GAM);    // bytes per wave's hand-over area
    constexpr int P = T + 1 + PF;
    constexpr int K = T > 2 ? T - 1 : 1;
    const int n = n0 + k;
    const int r0 = c.ystart + n;
    if (P16 && !PZ) {   // the entering row's p arrives packed: two snorm16 per dword
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int a = __float_as_int(X[k].d.p11[j]), b = __float_as_int(X[k].d.p21[j]);
            const float sc = 1.0f / 32767.0f;
            X[k].d.p11[j] = (float)(short)(a & 0xffff) * sc; X[k].d.p12[j] = (float)(a >> 16) * sc;
            X[k].d.p21[j] = (float)(short)(b & 0xffff) * sc; X[k].d.p22[j] = (float)(b >> 16) * sc;
        }
    }
    if (NG) {
#pragma unroll
        for (int j = 0; j < PPL; ++j) { const float ix2 = X[k].s.ix[j] * X[k].s.ix[j], iy2 = X[k].s.iy[j] * X[k].s.iy[j]; X[k].s.rg[j] = ix2 + iy2; }
    }
    if (MODE != 2) finish_static<PPL>(X[k].s);
    unsigned xexpect = 0, vtag_r = 0, vtag_l = 0;
    if (JW >= 2) {
        // the two hand-over base addresses live in VGPRs (a ds_* address operand is a VGPR: kept scalar they cost one v_mov per access)
        asm volatile("" : "+v"(x.own), "+v"(x.pub_l));
        if (JF) asm volatile("" : "+v"(x.pub_r));
        // the row entering the pipeline: its p11, p21 of lane 63 are the right neighbour's stage-0 input of this step
        if (JF) xwrite2f(x.pub_r, X[k].d.p11[0], X[k].d.p21[0]);
        else xwrite2(x.on_r, x.own + XAg, X[k].d.p11[0], X[k].d.p21[0]);
        if constexpr (GAM) xwrite1(x.on_r, x.own + XAg + 16, X[k].g.p31[0]);
    } else if (JW) {
        vtag_r = (unsigned)(n + 1); vtag_l = (unsigned)(n + 2);
        asm volatile("" : "+v"(vtag_r), "+v"(vtag_l), "+v"(x.own));   // one VGPR copy per step, not one v_mov per store
        // the row entering the pipeline: its p11, p21 of lane 63 are the right neighbour's stage-0 input of this step
        xwrite(x.pub_r, 0, 16, X[k].d.p11[0], X[k].d.p21[0], vtag_r);
        xexpect = ((unsigned)(n + 1) & 0xffffu) * x.mul;
    }
#ifndef TBR_X_NOLOAD   // timing experiments only (wrong results): no row loads after the prologue
    load_row_r<PPL, PZ, NG, P16, FW, GAM>(X[(k + PF) % P], c.B, c.uin, c.pin, r0 + PF, c.H, c.xc);
#else
    X[(k + PF) % P] = X[k];
#endif
    // static rows: stage 0 reads the registers of the entering row, stage 1 those of the previous row (still in its
    // register set), stages t >= 2 the LDS ring (row n - t), fetched one stage early.  The entering row is written to the ring
    // after the last read of the step has been issued: it replaces row n - (T-1), so K = T - 1 slots suffice.
    Stat<PPL> nx;
#pragma unroll
    for (int j = 0; j < PPL; ++j) nx.ix[j] = nx.iy[j] = nx.rg[j] = nx.rc[j] = 0.f;
    if (T < 3) lds_put<PPL>(c.ring + slot0 * (256 * PPL), c.lane, X[k].s);   // (unused ring: keeps the code uniform)
#pragma unroll
    for (int t = 0; t < T; ++t) {
        Stat<PPL> st;
        if (t == 0) st = X[k].s;
        else if (t == 1) st = X[(k - 1 + P) % P].s;
        else st = nx;
#ifdef TBR_X_NOLDS
        if (t >= 2) st = X[k].s;
#else
        if (t + 1 < T && t + 1 >= 2) {
            int sl = slot0 - (t + 1);
            if (sl < 0) sl += K;
            lds_get<PPL>(c.ring + sl * (256 * PPL), c.lane, nx);
            if (t + 1 == T - 1) lds_put<PPL>(c.ring + slot0 * (256 * PPL), c.lane, X[k].s);
        }
#endif
        const int a = r0 - t;   // row of the stage's input
        // wave-uniform masks, selected as INTEGERS so that they stay on the scalar unit (a float select is lowered to v_cndmask)
        const float negm1 = MK ? __uint_as_float((a == 0) ? 0u : 0xbf800000u) : -1.f;
        const float m2 = MK ? __uint_as_float((a == c.H) ? 0u : 0x3f800000u) : 1.f;
        const float taum2 = MK ? __uint_as_float((a == c.H) ? 0u : __float_as_uint(c.taut)) : c.taut;
        if (MODE == 1) {
            // speculative step: nit <= T active stages, each summing its error over the rows of this wave's band only (halo rows
            // belong to a neighbour).  A skipped stage writes nothing, which IS the identity of the rotating scheme: its output
            // set still holds the unmodified input row of the previous step.
            const float es = __uint_as_float((a >= c.y0 && a < c.y1) ? 0x4b800000u : 0u);   // 2^24 or 0, kept on the scalar unit
            if (JW >= 2) {
                // joined waves, barrier form: every wave of the workgroup passes the barrier of every stage (nit is a property of the
                // pair, i.e. of the whole workgroup); a skipped stage hands over what it holds -- its unmodified input
                if ((k * T + t) % xk_stages(T) == 0) xbarrier();
                float l1, l2, r1, r2;
                GamK G{c.gamma, c.eu3, 0.f, 0.f};
                xread2(x.own + t * 2 * XSg, l1, l2, r1, r2);
                if constexpr (GAM) xread1g(x.own + t * 2 * XSg, G.xl3, G.xr3);
                Dyn<PPL> &SA = X[(k - t + P) % P].d, &SB = X[(k - t - 1 + 2 * P) % P].d;
                Dyn3<PPL> *const SA3 = g3(X[(k - t + P) % P]), *const SB3 = g3(X[(k - t - 1 + 2 * P) % P]);
                if (t < c.nit) stage_r<PPL, true, 1, true, GAM>(SA, SB, st, c.right_ok, negm1, m2, taum2, c.l_t, c.theta, c.taut, acc[t], es, l1, l2, r1, r2, SA3, SB3, G);
                if (t + 1 < T) xwrite2(x.on_r, x.own + XAg + (t + 1) * 2 * XSg, SB.p11[0], SB.p21[0]);
                xwrite2(x.on_l, x.pub_l + t * 2 * XSg + 8, SA.u1[0], SA.u2[0]);
                if constexpr (GAM) {
                    if (t + 1 < T) xwrite1(x.on_r, x.own + XAg + (t + 1) * 2 * XSg + 16, SB3->p31[0]);
                    xwrite1(x.on_l, x.pub_l + t * 2 * XSg + 20, SA3->u3[0]);
                }
            } else if (t < c.nit)
                stage_r<PPL, true, 0, true, GAM>(X[(k - t + P) % P].d, X[(k - t - 1 + 2 * P) % P].d, st, c.right_ok, negm1, m2, taum2, c.l_t, c.theta,
                                                 c.taut, acc[t], es, 0.f, 0.f, 0.f, 0.f, g3(X[(k - t + P) % P]), g3(X[(k - t - 1 + 2 * P) % P]),
                                                 GamK{c.gamma, c.eu3, 0.f, 0.f});
        } else if (MODE == 2) {
            if constexpr (PPL == 1)
                stage_r_exact(X[(k - t + P) % P].d, X[(k - t - 1 + 2 * P) % P].d, st, c.right_ok[0], c.x0, a == 0, a == c.H, c.l_t, c.theta, c.taut);
        } else if (JF) {
            // hand-over values: read ONE STAGE EARLY wherever the interval rule allows it.  Stage g's left input was published in an
            // earlier interval than stage g - 1 runs in, always; its right input (the neighbour's stage g - T) only if g is not the
            // first stage of an interval (T = 2 XK: floor((g - T) / XK) + w + 1 < floor((g - 1) / XK) + w  <=>  g mod XK != 0) --
            // stages that open an interval read after their barrier, as before.
            constexpr int XKs = xk_stages(T);
            if ((k * T + t) % XKs == 0) xbarrier();
            if (t == 0 || (k * T + t) % XKs == 0) xread2(x.own + t * 2 * XS2, x.l1, x.l2, x.r1, x.r2);
            const float l1 = x.l1, l2 = x.l2, r1 = x.r1, r2 = x.r2;
            if (t + 1 < T && (k * T + t + 1) % XKs != 0) xread2(x.own + (t + 1) * 2 * XS2, x.l1, x.l2, x.r1, x.r2);
            unsigned long long dummy = 0;
            Dyn<PPL> &SA = X[(k - t + P) % P].d, &SB = X[(k - t - 1 + 2 * P) % P].d;
            stage_r<PPL, false, 1, MK>(SA, SB, st, c.right_ok, negm1, m2, taum2, c.l_t, c.theta, c.taut, dummy, 0.f, l1, l2, r1, r2);
            if (t + 1 < T) xwrite2f(x.pub_r + (t + 1) * 2 * XS2, SB.p11[0], SB.p21[0]);
            xwrite2f(x.pub_l + t * 2 * XS2 + 8, SA.u1[0], SA.u2[0]);
        } else if (JW >= 2) {
            if ((k * T + t) % xk_stages(T) == 0) xbarrier();
            // FW: the statics of the row that enters at the NEXT step, one step ahead of their use (the producer wave finished that row
            // before the barrier just passed -- see fw_produce); they land in the register set that row's u, p already wait in
            if constexpr (FW != 0) if (t == 0) fw_inbox_get<PPL>(c.inbox + ((n + 1) & (FW_RING - 1)) * FW_SLOT, c.lane, X[(k + 1) % P].s);
            float l1, l2, r1, r2;
            GamK G{c.gamma, 0.f, 0.f, 0.f};
            xread2(x.own + t * 2 * XSg, l1, l2, r1, r2);
            if constexpr (GAM) xread1g(x.own + t * 2 * XSg, G.xl3, G.xr3);
            unsigned long long dummy = 0;
            Dyn<PPL> &SA = X[(k - t + P) % P].d, &SB = X[(k - t - 1 + 2 * P) % P].d;
            Dyn3<PPL> *const SA3 = g3(X[(k - t + P) % P]), *const SB3 = g3(X[(k - t - 1 + 2 * P) % P]);
            stage_r<PPL, false, 1, true, GAM>(SA, SB, st, c.right_ok, negm1, m2, taum2, c.l_t, c.theta, c.taut, dummy, 0.f, l1, l2, r1, r2, SA3, SB3, G);
            // p_t(row a-1) of lane 63 -> stage t+1 of the right neighbour (the next area), this step; u_t(row a) of lane 0 -> stage t
            // of the left neighbour, NEXT step (pub_l points at its slots of the other parity)
            if (t + 1 < T) xwrite2(x.on_r, x.own + XAg + (t + 1) * 2 * XSg, SB.p11[0], SB.p21[0]);
            xwrite2(x.on_l, x.pub_l + t * 2 * XSg + 8, SA.u1[0], SA.u2[0]);
            if constexpr (GAM) {
                if (t + 1 < T) xwrite1(x.on_r, x.own + XAg + (t + 1) * 2 * XSg + 16, SB3->p31[0]);
                xwrite1(x.on_l, x.pub_l + t * 2 * XSg + 20, SA3->u3[0]);
            }
        } else if (JW) {
            // consume the slot read ahead for this stage (re-read until both writers have delivered), read ahead for the next one
            // (stage 0 of the next step lives in the other buffer), compute, hand the new boundary values over
            unsigned tag = x.tag;
            float l1 = x.l1, l2 = x.l2, r1 = x.r1, r2 = x.r2;
#pragma nounroll
            while (__builtin_amdgcn_readfirstlane(tag) != xexpect) {
                __builtin_amdgcn_s_sleep(1);
                xread(x.own + t * XS, tag, l1, l2, r1, r2);
                // every wait is bounded (a wave that never arrives must not hang the device): once the budget of the whole launch
                // is spent, whatever is there is taken and the launch reports the fault at its end
                if (--x.budget < 0) xexpect = __builtin_amdgcn_readfirstlane(tag);
            }
            // (x.own of the other buffer: own + d with d = -+T * XS, applied to all three addresses at the end of the step)
            if (t + 1 < T) xread(x.own + (t + 1) * XS, x.tag, x.l1, x.l2, x.r1, x.r2);
            else xread(x.own + ((n & 1) ? -T * XS : T * XS), x.tag, x.l1, x.l2, x.r1, x.r2);
            unsigned long long dummy = 0;
            Dyn<PPL> &SA = X[(k - t + P) % P].d, &SB = X[(k - t - 1 + 2 * P) % P].d;
            stage_r<PPL, false, 1>(SA, SB, st, c.right_ok, negm1, m2, taum2, c.l_t, c.theta, c.taut, dummy, 0.f, l1, l2, r1, r2);
            // p_t(row a-1) of lane 63 -> stage t+1 of the right neighbour, this step; u_t(row a) of lane 0 -> stage t of the left
            // neighbour, NEXT step (pub_l already points into its other buffer)
            if (t + 1 < T) xwrite(x.pub_r + (t + 1) * XS, 0, 16, SB.p11[0], SB.p21[0], vtag_r);
            xwrite(x.pub_l + t * XS, 8, 18, SA.u1[0], SA.u2[0], vtag_l);
        } else {
            unsigned long long dummy = 0;
            stage_r<PPL, false, 0, true, GAM>(X[(k - t + P) % P].d, X[(k - t - 1 + 2 * P) % P].d, st, c.right_ok, negm1, m2, taum2, c.l_t, c.theta,
                                              c.taut, dummy, 0.f, 0.f, 0.f, 0.f, 0.f, g3(X[(k - t + P) % P]), g3(X[(k - t - 1 + 2 * P) % P]),
                                              GamK{c.gamma, 0.f, 0.f, 0.f});
        }
    }
    if (JW >= 2) {   // the slots of the other parity for the next step (the two parities of a stage are adjacent: one address bit)
        x.own ^= XSg; x.pub_l ^= XSg;
        if (JF) x.pub_r ^= XSg;
    } else if (JW) {   // the other buffer for the next step
        const int d = (n & 1) ? -T * XS : T * XS;
        x.own += d; x.pub_r += d; x.pub_l -= d;   // pub_l addresses the left neighbour's buffer of the NEXT step: opposite phase
    }
    {   // level-T row r0 - T leaves the pipeline
        const Dyn<PPL> &r = X[(k - T + 2 * P) % P].d;
        const int orow = r0 - T;
#ifdef TBR_X_NOSTORE
        if (orow == c.y1 - 1 && c.st_ok) {
#else
        if (orow >= c.y0 && orow < c.y1 && c.st_ok) {
#endif
            const long long ro = (long long)orow * c.ld;   // wave-uniform; owned lanes have xc == 4 * xl
            unsigned xb = c.xc;
            asm volatile("" : "+v"(xb));
            str<PPL>(c.uout[0] + ro, xb, r.u1);
            str<PPL>(c.uout[1] + ro, xb, r.u2);
            if (P16 && !c.B.skip_p_out) {
                float pa[PPL], pb2[PPL];
#pragma unroll
                for (int j = 0; j < PPL; ++j) {
                    typedef short s2 __attribute__((ext_vector_type(2)));
                    const s2 v1 = __builtin_amdgcn_cvt_pknorm_i16(r.p11[j], r.p12[j]), v2 = __builtin_amdgcn_cvt_pknorm_i16(r.p21[j], r.p22[j]);
                    pa[j] = __int_as_float(((int)(unsigned short)v1.x) | ((int)v1.y << 16));
                    pb2[j] = __int_as_float(((int)(unsigned short)v2.x) | ((int)v2.y << 16));
                }
                str<PPL>(c.pout[0] + ro, xb, pa);
                str<PPL>(c.pout[2] + ro, xb, pb2);
            } else if (!c.B.skip_p_out) {   // the last pass of a scale: nobody reads its p (the next scale starts from p = 0)
                str<PPL>(c.pout[0] + ro, xb, r.p11);
                str<PPL>(c.pout[1] + ro, xb, r.p12);
                str<PPL>(c.pout[2] + ro, xb, r.p21);
                str<PPL>(c.pout[3] + ro, xb, r.p22);
            }
            if constexpr (GAM) {
                const Dyn3<PPL> &r3 = X[(k - T + 2 * P) % P].g;
                str<PPL>(c.uout[2] + ro, xb, r3.u3);
                if (!c.B.skip_p_out) {
                    str<PPL>(c.pout[4] + ro, xb, r3.p31);
                    str<PPL>(c.pout[5] + ro, xb, r3.p32);
                }
            }
        }
    }
    slot0 = (slot0 + 1 == K) ? 0 : slot0 + 1;
}
template <int T, int PPL, bool PZ, int PF, int MODE, int JW, bool MK, bool NG, bool P16, int FW, bool GAM, int... Ks>
__device__ __forceinline__ void steps_r(const CtxR<PPL> &c, Slot<PPL, GAM> (&X)[T + 1 + PF], int n0, int &slot0, unsigned long long (&acc)[T],
                                        Xchg &x, std::integer_sequence<int, Ks...>)
{
    (step_r<T, PPL, PZ, PF, MODE, JW, MK, NG, P16, FW, GAM, Ks>(c, X, n0, slot0, acc, x), ...);
}

// MODE 0: T iterations, fixed work.
// MODE 1: one SPECULATIVE STEP of the convergence-checked path (epsilon > 0).  The reference tests the error after every
//   iteration (CPU class, optflow/src/tvl1flow.cpp:1376-1390) or on cv::cuda's sparse schedule (cudaoptflow/src/tvl1flow.cpp:357-377)
//   and the flow depends on where the loop stops, so a block of iterations cannot simply be fused.  Instead a launch runs a block
//   of nit <= T iterations from set `base` into set base^1 while recording the nit per-iteration error sums; the NEXT launch
//   applies the stopping rule to those sums: the block stands (and the next block starts from its output), or the loop would have
//   stopped after k < nit iterations and this launch REPLAYS exactly k iterations from the block's input, which is still intact.
//   nit is chosen on the device from the error history (the previous warp's count, then the decay of the sums).  The last
//   launch of a warp only settles.  Control travels through the per-launch slots of Ctl / SpecK, so the calc stays stream-ordered
//   and bit-reproducible (integer error sums, decisions from device data only).
//
// JW = 1 (joined waves; T = 10, 1 px per lane, MODE 0): the four waves of a workgroup are not four bands of one 64-column strip but
//   four ADJACENT 64-column segments of ONE band, i.e. a 256-column strip whose margin of T columns exists only at its two outer
//   edges (236 of 256 lanes own a column instead of 44 of 64: 33 instead of 44 waves per 1080p row band).  At the three inner
//   seams the neighbouring waves hand each other the two values a stage needs from across the seam through LDS (Xchg above).  The
//   arithmetic of an owned pixel is the same operations on the same values as in the independent-wave form: bit-identical planes.
// GAM (round 6): gamma != 0 -- the illumination channel (u3, p31, p32) rides through the same pipeline: nine instead of six dynamic
//   registers per set, 32-byte hand-over slots, three more loads and stores per row; always without a |grad|^2 plane (NG).
template <int T, int PPL, bool PZ, int WPS, int PF, int MODE, int JW = 0, bool NG = false, bool P16 = false, int FW = 0, bool GAM = false>
__global__ __launch_bounds__((JW == 3 || FW) ? 512 : 256, WPS) void k_iterate_tbr(TbArgs A)
{
    static_assert(!GAM || (NG && !P16 && !FW && (JW == 0 || JW == 2) && MODE != 2), "illumination channel: independent or barrier-joined waves, fast math, no |grad|^2 plane");
    static_assert(!FW || (JW == 2 && MODE == 0 && NG && !P16 && PPL == 1 && T == 10), "fused warp: the default joined-wave fixed-work kernel only");
    static_assert(!JW || (PPL == 1 && T > 2 && (MODE == 0 || (MODE == 1 && JW >= 2 && JW != 4))), "joined waves: 1 px per lane; the speculative steps in the barrier form only");
    static_assert(JW < 2 || (((T + 1 + PF) * T) % xk_stages(T) == 0 && T >= 2 * xk_stages(T)), "barrier intervals must tile the unrolled block and leave the right neighbour a full interval");
    constexpr int M = (T + PPL - 1) / PPL * PPL;   // validity margin per side (px)
    constexpr int NW = jw_waves(JW);               // waves of a workgroup
    constexpr int LW = (JW ? 64 * NW : 64) * PPL;  // pixels a strip covers: a wave, or the joined waves of a workgroup
    constexpr int STRIDE = LW - 2 * M;             // owned columns of the strips >= 1 (strip 0 owns LW - M)
    constexpr int P = T + 1 + PF;                  // register sets
    constexpr int K = T > 2 ? T - 1 : 1;           // LDS ring slots: the row of step n is read by stages 2..T-1 at steps n+2..n+T-1
    // (r19b: a static s_setprio 1 here changes nothing -- 1 412 against 1 416 pairs/s --, s_setprio 2 on the warp kernel costs 2.5 %)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    CtxR<PPL> c;
    c.lane = threadIdx.x & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // FW: waves NW .. 2 NW - 1 are the producers of consumers 0 .. NW - 1 and share their geometry
    const bool producer = FW && wave_id >= NW;
    const int wave = producer ? wave_id - NW : wave_id;
    // block -> (strip, band group, pair); with swz the linear workgroup id is remapped so that each XCD (id % 8) owns a
    // contiguous run of strips: neighbouring strips share their halo columns through ONE L2
    int strip = blockIdx.x, bgrp = blockIdx.y, b = blockIdx.z;
    if (A.swz & 1) {
        const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
        const unsigned orig = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned xcd = orig & 7u, qq = nwg >> 3, rr = nwg & 7u;
        const unsigned lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
        strip = lid % gridDim.x; bgrp = (lid / gridDim.x) % gridDim.y; b = lid / (gridDim.x * gridDim.y);
    }
    strip = __builtin_amdgcn_readfirstlane(strip); bgrp = __builtin_amdgcn_readfirstlane(bgrp); b = __builtin_amdgcn_readfirstlane(b);
    const int band = JW ? bgrp : bgrp * 4 + wave;   // (independent waves: NW = 4 bands per workgroup)
    const int W = A.g.w;
    c.H = A.g.h; c.ld = A.g.ld;
    c.y0 = band * A.rows_per_band;
    c.ring = lds + wave * (K * 256 * PPL);
    if (c.y0 >= c.H) return;   // JW: the same for all four waves
    c.y1 = min(c.y0 + A.rows_per_band, c.H);
    // strip 0 starts at the image border in lane 0; strip s >= 1 starts M px left of what it owns
    const int own_lo = strip == 0 ? 0 : (LW - M) + (strip - 1) * STRIDE;
    const int own_hi = min(strip == 0 ? LW - M : own_lo + STRIDE, W);
    const int xw = (strip == 0 ? 0 : own_lo - M) + (JW ? wave * 64 : 0);   // first pixel of this wave
    const int xl = xw + c.lane * PPL;   // first pixel of this lane, >= 0
    Xchg x;
    x.own = x.pub_r = x.pub_l = 0; x.mul = 0; x.tag = 0; x.l1 = x.l2 = x.r1 = x.r2 = 0.f; x.budget = 1 << 20; x.on_r = x.on_l = false;
    if (JW >= 2) {
        // behind the NW rings: 4 areas of T stages x 2 parities (of the step) x 16 bytes, all zero at the start: a wave without a left
        // / right neighbour keeps reading zeros there -- the fill of the independent-wave form -- and parity 0 holds the all-zero u of
        // "step -1".  Slot of (stage t, parity q) = area + t * 32 + q * 16: the step's parity is ONE address bit (areas are 32-byte aligned)
        constexpr int XA = xarea2_bytes(T, GAM);
        const unsigned xb = (unsigned)(unsigned long long)(lds_ptr)(lds + NW * (K * 256 * PPL));
        const bool has_left = wave > 0, has_right = wave < NW - 1 && xw + 64 < W;
        x.own = xb + wave * XA;                         // parity 0; the right neighbour's area is own + XA (lane 63 of a wave that has one)
        x.pub_l = xb + (wave - 1) * XA + xs2_bytes(GAM);   // the left neighbour's slots of step 1 (parity 1); lane 0 of a wave with a left neighbour only
        x.on_r = has_right && c.lane == 63;
        x.on_l = has_left && c.lane == 0;
        if (FW && threadIdx.x < 128) (lds + NW * (K * 256 * PPL) + NW * XA / 4 + NW * FW_RING * FW_SLOT)[threadIdx.x] = A.ftab[threadIdx.x];
        if (jw_fast(JW)) {
            // per-lane publish addresses: the neighbour's slots for the one lane that holds the boundary column, the workgroup's dump
            // area (never read; the parity bit and the stage offsets apply to it alike) for everybody else
            const unsigned dump = xb + NW * XA + 32 + c.lane * 8;
            x.pub_r = x.on_r ? x.own + XA : dump;
            x.pub_l = x.on_l ? x.pub_l : dump;
        }
        for (int i = c.lane; i < XA / 4; i += 64) reinterpret_cast<volatile MI_LDS unsigned *>((lds_ptr)(unsigned long long)x.own)[i] = 0u;
        __syncthreads();
        if (xw >= W) return;
    } else if (JW) {
        // behind the four rings: 4 areas, then 4 dumps.  A wave whose first column lies beyond the image does nothing (and nobody
        // waits for it): its left neighbour's last columns are cut by right_ok exactly as at the right edge of a strip.
        constexpr int XA = xarea_bytes(T), XD = xdump_bytes(T);
        const unsigned xb = (unsigned)(unsigned long long)(lds_ptr)(lds + 4 * (K * 256 * PPL));
        const bool has_left = wave > 0, has_right = wave < 3 && xw + 64 < W;
        const unsigned dump = xb + 4 * XA + wave * XD + c.lane * 8;
        x.own = xb + wave * XA;
        x.pub_r = (has_right && c.lane == 63) ? xb + (wave + 1) * XA : dump;
        x.pub_l = ((has_left && c.lane == 0) ? xb + (wave - 1) * XA : dump) + T * XS;   // the left neighbour's buffer of step 1 (the dump addresses toggle alike)
        x.mul = (has_left ? 1u : 0u) | (has_right ? 0x10000u : 0u);
        // zero data, tags 0; buffer 0 of step 0 expects "step -1" of the right neighbour: all-zero u (the initial held state)
        for (int i = c.lane; i < XA / 4; i += 64) {
            const int w4 = i % (XS / 4), buf = i / (T * XS / 4);
            reinterpret_cast<volatile MI_LDS unsigned *>((lds_ptr)(unsigned long long)x.own)[i] = (w4 == 4 && buf == 0 && has_right) ? 0x10000u : 0u;
        }
        __syncthreads();
        if (xw >= W) return;
    }
#pragma unroll
    for (int j = 0; j < PPL; ++j) c.right_ok[j] = (xl + j + 1 < W);
    c.st_ok = xl >= own_lo && xl < own_hi;
    c.x0 = xl == 0;
    c.xc = 4u * (unsigned)min(xl, c.ld - PPL);   // clamped column of the unconditional loads, bytes

    const long long pb = (long long)b * A.g.ps;
    int cur = A.cur;
    c.ystart = c.y0 - T;
    c.nsteps = (c.y1 - c.y0) + 2 * T;
    c.inbox = nullptr;
    if constexpr (FW != 0) {
        float *fw0 = lds + NW * (K * 256 * PPL) + NW * xarea2_bytes(T) / 4;
        c.inbox = fw0 + wave * (FW_RING * FW_SLOT);
        if (producer) {
#if FW_STAGE
            fw_produce_staged<FW, NW>(A, c.inbox, fw0 + NW * FW_RING * FW_SLOT, fw0 + fw_lds_floats(NW) + wave * (FWS_NR * FWS_CW), c.lane, xw, c.ystart,
                                      (c.nsteps + P - 1) / P * P, (c.y0 + c.y1) >> 1, pb, cur);
#else
            fw_produce<FW, NW>(A, c.inbox, fw0 + NW * FW_RING * FW_SLOT, c.lane, xw, c.ystart, (c.nsteps + P - 1) / P * P, pb, cur);
#endif
            return;
        }
    }
    c.nit = T;
    bool record = false;
    if (MODE == 1) {
        // settle the previous launch's speculative block and pick this launch's work (tvl1_tb_dev.h); one writer per pair b: the
        // REMAPPED block indices
        int nit = 0;
        if (!spec_settle(A.ctl, A.sk, T, b, strip == 0 && bgrp == 0 && threadIdx.x == 0, cur, nit, record)) return;
        c.nit = __builtin_amdgcn_readfirstlane(nit);
    }
    c.uin[0] = A.pl.u[cur][0] + pb; c.uin[1] = A.pl.u[cur][1] + pb;
    c.uout[0] = A.pl.u[cur ^ 1][0] + pb; c.uout[1] = A.pl.u[cur ^ 1][1] + pb;
#pragma unroll
    for (int i = 0; i < 4; ++i) { c.pin[i] = A.pl.p[cur][i] + pb; c.pout[i] = A.pl.p[cur ^ 1][i] + pb; }
    c.uin[2] = nullptr; c.uout[2] = nullptr; c.pin[4] = c.pin[5] = nullptr; c.pout[4] = c.pout[5] = nullptr;
    c.gamma = 0.f; c.eu3 = 0.f;
    if constexpr (GAM) {
        c.uin[2] = A.pl.u[cur][2] + pb; c.uout[2] = A.pl.u[cur ^ 1][2] + pb;
#pragma unroll
        for (int i = 4; i < 6; ++i) { c.pin[i] = A.pl.p[cur][i] + pb; c.pout[i] = A.pl.p[cur ^ 1][i] + pb; }
        c.gamma = A.pl.gamma; c.eu3 = A.pl.err_u3 ? 1.f : 0.f;
    }
    c.B = A;
    if (!FW) { c.B.pl.ix += pb; c.B.pl.iy += pb; if (!NG) c.B.pl.g += pb; c.B.pl.rc += pb; }
    c.l_t = A.l_t; c.theta = A.theta; c.taut = A.taut;

    Slot<PPL, GAM> X[P];
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            X[i].d.u1[j] = X[i].d.u2[j] = X[i].d.p11[j] = X[i].d.p12[j] = X[i].d.p21[j] = X[i].d.p22[j] = 0.f;
            X[i].s.ix[j] = X[i].s.iy[j] = X[i].s.rg[j] = X[i].s.rc[j] = 0.f;
            if constexpr (GAM) X[i].g.u3[j] = X[i].g.p31[j] = X[i].g.p32[j] = 0.f;
        }
    for (int k = 0; k < K; ++k) lds_put<PPL>(c.ring + k * (256 * PPL), c.lane, X[0].s);

#pragma unroll
    for (int k = 0; k < PF; ++k) load_row_r<PPL, PZ, NG, P16, FW, GAM>(X[k], c.B, c.uin, c.pin, c.ystart + k, c.H, c.xc);
    if constexpr (FW != 0) {   // the barrier that opens the pipeline (fw_produce): rows 0 and 1 are in the inbox; row 0 enters at step 0
        xbarrier();
        fw_inbox_get<PPL>(c.inbox, c.lane, X[0].s);
    }
    int slot0 = 0;   // ring slot of the row entering at this step (= step mod K)
    unsigned long long acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0;
    if (JW >= 2 && xk_stages(T) > 1)   // the skew: wave w starts w barrier intervals after wave 0 ...
        for (int i = 0; i < wave; ++i) xbarrier();
    for (int n0 = 0; n0 < c.nsteps; n0 += P) {
        if (jw_fast(JW)) {
            // rows a = r0 - t seen by the stages of this block of P steps: [ystart + n0 - (T - 1), ystart + n0 + P - 1].  If neither row 0
            // nor row H is among them the block runs the form without border masks (wave-uniform, and the same for the four waves of
            // the workgroup: they share the band)
            const int ra = c.ystart + n0 - (T - 1), rb = c.ystart + n0 + P - 1;
            const bool plain = (ra > 0 || rb < 0) && (ra > c.H || rb < c.H);
            if (plain) { steps_r<T, PPL, PZ, PF, MODE, JW, false, NG, P16, FW, GAM>(c, X, n0, slot0, acc, x, std::make_integer_sequence<int, P>{}); continue; }
        }
        steps_r<T, PPL, PZ, PF, MODE, JW, true, NG, P16, FW, GAM>(c, X, n0, slot0, acc, x, std::make_integer_sequence<int, P>{});
    }
    if (JW >= 2 && xk_stages(T) > 1)   // ... and keeps the others company for as many at the end (every live wave passes the same number)
        for (int i = wave; i < NW - 1; ++i) xbarrier();
    if (JW == 1 && x.budget < 0 && c.lane == 0) g_jw_fault = 1;
    if (MODE == 1 && record) {
        // integer error sums: exact wave reduction of the owned lanes, one device-scope add per wave and level
#pragma unroll
        for (int t = 0; t < T; ++t) {
            unsigned long long sacc = c.st_ok ? acc[t] : 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
            if (c.lane == 0 && t < c.nit) atomicAdd(&A.ctl.E[(long long)b * A.ctl.Q + A.e0 + t], sacc);
        }
    }
}

template <int T, int PPL, int WPS, int PF, int MODE, int JW = 0, bool NG = false, bool P16 = false, int FW = 0, bool GAM = false>
static int launch_tbr(const TbArgs &A0, bool pz, hipStream_t s)
{
    constexpr int M = (T + PPL - 1) / PPL * PPL;
    constexpr int NW = jw_waves(JW);
    constexpr int LW = (JW ? 64 * NW : 64) * PPL;
    constexpr int STRIDE = LW - 2 * M;
    TbArgs A = A0;
    A.nstrips = A.g.w <= LW - M ? 1 : 1 + div_up(A.g.w - (LW - M), STRIDE);
    A.swz = tuning().tb_swz == 1 ? 1 : 0;
    // JW: a workgroup is one band of a 256-column strip; otherwise four consecutive bands of a 64-column strip
    const dim3 grid(A.nstrips, JW ? div_up(A.g.h, A.rows_per_band) : div_up(div_up(A.g.h, A.rows_per_band), 4), A.g.batch);
    constexpr size_t lds_bytes = (size_t)NW * (T > 2 ? T - 1 : 1) * 256 * PPL * sizeof(float) +
                                 (JW >= 2 ? NW * xarea2_bytes(T, GAM) + (jw_fast(JW) ? xdump4_bytes(T) : 0) : JW ? 4 * (xarea_bytes(T) + xdump_bytes(T)) : 0) +
                                 (FW ? (fw_lds_floats(NW) + (FW_STAGE ? fws_lds_floats(NW) : 0)) * sizeof(float) : 0);
    constexpr int NTHREADS = 64 * NW * (FW ? 2 : 1);
    // once per instantiation (thread-safe function-local static), result checked on every launch
    static const hipError_t attr_rc = [] {
        hipError_t e = hipFuncSetAttribute((const void *)k_iterate_tbr<T, PPL, true, WPS, PF, MODE, JW, NG, P16, FW, GAM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)k_iterate_tbr<T, PPL, false, WPS, PF, MODE, JW, NG, P16, FW, GAM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        return e;
    }();
    MI_HIP_TRY(attr_rc);
    if (tuning().tb_verbose) {
        static const int nb = [] {
            int n = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k_iterate_tbr<T, PPL, false, WPS, PF, MODE, JW, NG, P16, FW, GAM>, NTHREADS, lds_bytes);
            fprintf(stderr, "[tbr] T=%d ppl=%d wps=%d pf=%d jw=%d lds=%zu B/block -> %d resident blocks/CU\n", T, PPL, WPS, PF, JW, lds_bytes, n);
            return n;
        }();
        (void)nb;
    }
    if (pz) hipLaunchKernelGGL((k_iterate_tbr<T, PPL, true, WPS, PF, MODE, JW, NG, P16, FW, GAM>), grid, dim3(NTHREADS), lds_bytes, s, A);
    else hipLaunchKernelGGL((k_iterate_tbr<T, PPL, false, WPS, PF, MODE, JW, NG, P16, FW, GAM>), grid, dim3(NTHREADS), lds_bytes, s, A);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

// WPS = occupancy the register allocator is held to (launch bound); PLAN = waves/SIMD the band planner assumes.  r01s: the
// T10 kernel is resident 4 waves/SIMD but fastest when the grid is cut for 3 (394 vs 340 G px-iter/s).
typedef int (*TbLaunchFn)(const TbArgs &, bool, hipStream_t);
struct TbrEntry {
    int T, PPL, WPS, PF, PLAN;
    TbLaunchFn launch, spec;
    int JW;   // 1 / 2: joined waves (a workgroup = one band of a 256-column strip), hand-over by tags / by one barrier per stage
    bool GAM; // the kernel carries the illumination channel (gamma != 0)
};
#define TBR(T, PPL, WPS, PF, PLAN) {T, PPL, WPS, PF, PLAN, launch_tbr<T, PPL, WPS, PF, 0>, nullptr, 0}
// Ship what is used: the release library instantiates the kernels a release build can reach -- the barrier form of
// the joined waves (JW = 2), the independent-wave blocks of the other lengths, the exact-math blocks, the illumination channel.  The
// forms that lost their A/B (tags, eight joined waves, branch-free publishes, the fused warp, 16-bit dual storage, alternative register
// shapes, MIFLOW_TB_JW = 0) exist under -DMIFLOW_EXPERIMENTS only (libmiflow_exp.so), where the digest tests of their bit-identity run.
static const TbrEntry g_tbr_jw[] = {
#ifdef MIFLOW_EXPERIMENTS
                                     // joined waves with tags: the hand-over registers cost 14 VGPRs (3 waves/SIMD), rings + hand-over areas 40.7 KB of LDS = 3 workgroups per CU
                                     {10, 1, 3, 2, 3, launch_tbr<10, 1, 3, 2, 0, 1>, nullptr, 1},
                                     // eight joined waves (MIFLOW_TB_JW=3): 512-column strips, two workgroups of eight waves per CU
                                     {10, 1, 4, 2, 3, launch_tbr<10, 1, 4, 2, 0, 3>, nullptr, 3},
                                     // barrier form without exec-masked publishes and without border masks in interior blocks, hand-over
                                     // values read a stage early (MIFLOW_TB_JW=4)
                                     {10, 1, 4, 2, 3, launch_tbr<10, 1, 4, 2, 0, 4>, nullptr, 4},
#endif
                                     // barrier form (MIFLOW_TB_JW=2, the default): no read-ahead registers, no dump area: four waves/SIMD and four workgroups/CU
                                     {10, 1, 4, 2, 3, launch_tbr<10, 1, 4, 2, 0, 2>, nullptr, 2}};
// the default kernel without a |grad|^2 plane (tb_nograd_entry)
static const TbrEntry g_tbr_ng = {10, 1, 4, 2, 3, launch_tbr<10, 1, 4, 2, 0, 2, true>, nullptr, 2};
// ... and with the warp inside (FW: four producer waves beside the four joined consumers; two workgroups of eight waves per CU, i.e. two
// CONSUMER waves per SIMD is what the band planner fills): [0] the CPU class's arithmetic, tap-by-tap sums; [1] cv::cuda's, separable sums
#ifdef MIFLOW_EXPERIMENTS
static const TbrEntry g_tbr_fw[] = {{10, 1, 4, 2, 2, launch_tbr<10, 1, 4, 2, 0, 2, true, false, 1>, nullptr, 2},
                                    {10, 1, 4, 2, 2, launch_tbr<10, 1, 4, 2, 0, 2, true, false, 2>, nullptr, 2}};
#endif
// gamma != 0 (round 6): the blocked kernel with the illumination channel -- nine dynamic registers per set (T = 10:
// three waves/SIMD), always without a |grad|^2 plane.  Blocks of 10 and 5 as joined waves, 2 and 1 as independent waves (any iteration
// count decomposes greedily: tb_plan_gam)
// (T = 10 with ONE prefetched row: 163 VGPRs = three waves/SIMD without scratch; with two it is 168 + 12 spilled dwords)
static const TbrEntry g_tbr_gam[] = {{10, 1, 3, 1, 3, launch_tbr<10, 1, 3, 1, 0, 2, true, false, 0, true>, nullptr, 2, true},
                                     {5, 1, 4, 2, 3, launch_tbr<5, 1, 4, 2, 0, 2, true, false, 0, true>, nullptr, 2, true},
                                     {2, 1, 6, 2, 6, launch_tbr<2, 1, 6, 2, 0, 0, true, false, 0, true>, nullptr, 0, true},
                                     {1, 1, 8, 2, 8, launch_tbr<1, 1, 8, 2, 0, 0, true, false, 0, true>, nullptr, 0, true}};
// ... and its speculative steps (convergence-checked path)
static const TbrEntry g_spec_gam[] = {{10, 1, 2, 2, 2, nullptr, launch_tbr<10, 1, 2, 2, 1, 2, true, false, 0, true>, 2, true},
                                      {5, 1, 3, 2, 2, nullptr, launch_tbr<5, 1, 3, 2, 1, 2, true, false, 0, true>, 2, true}};
#ifdef MIFLOW_EXPERIMENTS
static const TbrEntry g_tbr_ng16 = {10, 1, 4, 2, 3, launch_tbr<10, 1, 4, 2, 0, 2, true, true>, nullptr, 2};   // + p as snorm16 between passes (changes results)
#endif
static const TbrEntry g_tbr[] = {
    // first entry of each T = default (r01s sweep, G px-iter/s at 1080p x 16: T10 394 | T8 353 | T6 271 | T5 256 | T4 215 | T3 152 | T2 106 | T1 64)
    TBR(10, 1, 4, 2, 3), TBR(8, 2, 2, 2, 2), TBR(6, 1, 5, 2, 3), TBR(5, 2, 3, 2, 3), TBR(4, 2, 3, 2, 3), TBR(3, 1, 7, 2, 6), TBR(2, 1, 8, 2, 8),
    TBR(1, 1, 8, 2, 8),
#ifdef MIFLOW_EXPERIMENTS
    // alternatives (tuning sweeps, MIFLOW_TB_VARIANT)
    TBR(10, 2, 2, 2, 2), TBR(8, 1, 4, 2, 2), TBR(6, 2, 3, 2, 3),
#endif
};
// The speculative steps (MODE 1: T accumulator registers more, hence one wave/SIMD less than MODE 0 at T = 10).
#define TBRS(T, PPL, WPS, PF, PLAN) {T, PPL, WPS, PF, PLAN, nullptr, launch_tbr<T, PPL, WPS, PF, 1>, 0}
#define TBRSJ(T, PPL, WPS, PF, PLAN) {T, PPL, WPS, PF, PLAN, nullptr, launch_tbr<T, PPL, WPS, PF, 1, 2>, 2}
// PLAN = 2: the bands are cut for two waves per SIMD -- fewer, taller bands (less halo) than the fixed-work kernels use; the other
// lane's kernels fill the rest of the device (r02m at 1080p x 16, class defaults: 517 -> 545 pairs/s; fixed work loses 7 %).
// (a release build runs the speculative steps of a level either on the register tiles or, without a |grad|^2 plane, on g_spec_jw_ng:
// the two tables below are reachable through experiment switches only)
#ifdef MIFLOW_EXPERIMENTS
static const TbrEntry g_spec[] = {TBRS(10, 1, 3, 2, 2), TBRS(5, 1, 4, 2, 2)};
static const TbrEntry g_spec_jw[] = {TBRSJ(10, 1, 3, 2, 2), TBRSJ(5, 1, 4, 2, 2)};   // MIFLOW_TB_JW >= 2 and MIFLOW_TB_JW_SPEC=1
#endif
// ... and the same two without a |grad|^2 plane (round 4: the convergence-checked path re-reads the statics once per block)
static const TbrEntry g_spec_jw_ng[] = {{10, 1, 3, 2, 2, nullptr, launch_tbr<10, 1, 3, 2, 1, 2, true>, 2}, {5, 1, 4, 2, 2, nullptr, launch_tbr<5, 1, 4, 2, 1, 2, true>, 2}};
bool tb_spec_nograd_ok(const Geo &g)
{
    if (!tuning().tb_nograd || tuning().tb_jw < 2 || !tuning().tb_jw_spec) return false;
    return !(tile_eligible(g) && tuning().tile_spec != 0);   // the register-tile kernel reads the stored plane
}

// Exact-math blocks (MODE 2; 1 px per lane).  The stage costs about four times the fast one (three IEEE divisions, two double
// square roots), so short blocks already move the kernel from the HBM bound of the one-iteration kernel to the issue bound.
#define TBRX(T, WPS, PLAN) {T, 1, WPS, 2, PLAN, launch_tbr<T, 1, WPS, 2, 2>, nullptr, 0}
// r02y at 1080p x 16, N = 10: blocks of 2 | 3 | 5 | 10 = 345 | 423 | 532 | 451 pairs/s (one launch per iteration: 228)
static const TbrEntry g_exact[] = {TBRX(5, 4, 3), TBRX(4, 4, 3), TBRX(3, 5, 4), TBRX(2, 6, 4), TBRX(1, 8, 4)};

// First entry of time block T, or the entry matching MIFLOW_TB_VARIANT=ppl,wps,pf.  Returns nullptr if T has none.
static const TbrEntry *tbr_pick(int T)
{
    const Tuning &tn = tuning();
    if (tn.tb_jw && tn.tb_ppl < 0)
        for (const TbrEntry &e : g_tbr_jw)
            if (e.T == T && e.JW == tn.tb_jw) return &e;
    const TbrEntry *def = nullptr;
    for (const TbrEntry &e : g_tbr) {
        if (e.T != T) continue;
        if (!def) def = &e;
        if (e.PPL == tn.tb_ppl && e.WPS == tn.tb_wps && e.PF == tn.tb_pf) { def = &e; break; }
    }
    return def;
}

int tb_max_block() { return 10; }

// Decompose n iterations into supported time blocks minimising the modelled cost.  cost[T] = measured
// ps per pixel-iteration of k_iterate_tbr<T> at 1080p x 16 pairs (tools/sweep_tb.py, profiles/r01s):
// deeper blocks save HBM passes but cost registers (occupancy) and halo recomputation.
int tb_plan(int n, int cap, int *blocks, int max_blocks)
{
    static const int sup[] = {1, 2, 3, 4, 5, 6, 8, 10};
    static const double cost[11] = {0, 15.6, 9.4, 6.6, 4.65, 3.9, 3.7, 0, 2.83, 0, 2.54};
    if (n <= 0) return 0;
    if (tuning().tb_force) {   // tuning sweeps: greedy blocks of exactly `cap` (then the largest that fit)
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

// gamma != 0: greedy blocks of the four lengths the channel's kernels exist in (g_tbr_gam)
int tb_plan_gam(int n, int cap, int *blocks, int max_blocks)
{
    int k = 0;
    for (int left = n; left > 0 && k < max_blocks;) {
        int t = 1;
        for (const TbrEntry &e : g_tbr_gam) if (e.T <= left && e.T <= cap && e.T > t) t = e.T;
        blocks[k++] = t;
        left -= t;
    }
    return k;
}

int tb_plan_level(const Geo &g, int n, int cap, int *blocks, int max_blocks, bool gam)
{
    if (gam && (!tile_eligible(g) || tuning().tb_force)) return tb_plan_gam(n, cap, blocks, max_blocks);   // the channel's streaming kernels: blocks of 10 / 5 / 2 / 1
    if (!tile_eligible(g) || tuning().tb_force) return tb_plan(n, cap, blocks, max_blocks);
    // register-tile kernel: any block length up to its margin costs one launch; fewest launches win
    int k = 0;
    const int top = cap < tile_max_block() ? cap : tile_max_block();
    for (int left = n; left > 0 && k < max_blocks;) {
        const int t = left < top ? left : top;
        blocks[k++] = t;
        left -= t;
    }
    return k;
}

// Band height: every wave streams rows_per_band + 2T rows, in whole blocks of P = T + 1 + PF steps.  Pick the band count
// that minimises rounds x steps, rounds = ceil(waves / resident-wave capacity), so that the grid fills the SIMDs of the
// device in whole rounds (no half-empty tail round) while the 2T-row band overlap stays small.
static int plan_band_rows(const TbrEntry &e, const Geo &g)
{
    const Tuning &tn = tuning();
    if (tn.tb_rows > 0) return tn.tb_rows;
    const int T = e.T, ppl = e.PPL, P = T + 1 + e.PF;
    int wps = e.PLAN;
    const int ring_slots = T > 2 ? T - 1 : 1;
    const int nw = e.JW ? jw_waves(e.JW) : 4;
    int lds_blocks = (160 * 1024) / (ring_slots * nw * 256 * ppl * 4 + (e.JW >= 2 ? nw * xarea2_bytes(T, e.GAM) + (jw_fast(e.JW) ? xdump4_bytes(T) : 0) : e.JW ? 4 * (xarea_bytes(T) + xdump_bytes(T)) : 0));
    lds_blocks = lds_blocks * nw / 4;   // in units of four-wave workgroups (= waves per SIMD)
    if (wps > lds_blocks) wps = lds_blocks;
    if (tn.tb_plan_wps > 0) wps = tn.tb_plan_wps;
    const long long cap = (long long)device_simds() * wps;
    const int M = (T + ppl - 1) / ppl * ppl;
    const int LW = (e.JW ? 64 * nw : 64) * ppl;
    const long long strips = g.w <= LW - M ? 1 : 1 + div_up(g.w - (LW - M), LW - 2 * M);
    const long long per_band = strips * g.batch * (e.JW ? nw : 1);   // waves per band row
    long long best_cost = -1;
    int best_nb = 1;
    for (int nb = 1; nb <= g.h; ++nb) {
        const int R = div_up(g.h, nb);
        if (R < 8 && nb > 1) break;
        // a workgroup is four consecutive bands of a strip and the LDS rings admit four workgroups per CU: with fewer than four
        // bands every workgroup has only nb live waves (the others exit at once, their ring stays allocated), i.e. at most nb
        // waves per SIMD (r02z4: 2 bands at 32 pairs per lane ran 1.3 x slower than 4)
        const long long cap_nb = !e.JW && nb < 4 && nb < wps ? (long long)device_simds() * nb : cap;
        const long long rounds = (per_band * nb + cap_nb - 1) / cap_nb;
        const long long steps = (long long)div_up(R + 2 * T, P) * P;
        const long long cost = rounds * steps;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_nb = nb; }
    }
    return div_up(g.h, best_nb);
}

// What iterate_tb(T, ...) would launch for level g (introspection for the bench's accounting): kernel 0 = streaming (rows = band
// height), 1 = register tile (rows = owned rows of a tile).
int tb_query_plan(int T, const Geo &g, int *kernel, int *rows)
{
    if (tile_eligible(g) && T <= tile_max_block() && !tuning().tb_force) {
        *kernel = 1;
        *rows = tile_owned_rows();
        return MI_OK;
    }
    const TbrEntry *e = tbr_pick(T);
    if (!e) { set_error("unsupported time block %d", T); return MI_ERR_BAD_ARG; }
    *kernel = 0;
    *rows = plan_band_rows(*e, g);
    return MI_OK;
}

// T fused iterations, set cur -> cur^1.  Returns MI_ERR_BAD_ARG for unsupported T.
// The launch iterate_tb(T) would make can run without a |grad|^2 plane (pl.g == nullptr): the default joined-wave T = 10 kernel on a
// level that does not take the register-tile kernel.  The caller (lane_calc) asks BEFORE the warp, which then does not store the plane.
bool tb_nograd_ok(int T, const Geo &g)
{
    if (!tuning().tb_nograd || T != 10 || tuning().tb_jw != 2 || tuning().tb_ppl >= 0) return false;
    if (tile_eligible(g) && T <= tile_max_block() && !tuning().tb_force) return false;
    return true;
}

// A warp whose iterations are ONE pass of the default T = 10 kernel can run with the warp inside that pass (k_iterate_tbr FW): no warp
// launch, no I1wx / I1wy / rho_c planes.  Which warp arithmetic: the two defaults (CPU class + tap-by-tap sums, cv::cuda + separable sums).
bool tb_fused_ok(int T, const Geo &g, int semantics, bool fast_warp)
{
#ifndef MIFLOW_EXPERIMENTS
    (void)T; (void)g; (void)semantics; (void)fast_warp;
    return false;   // measured slower (profiles/r14): the fused form is built into the experiments library only
#else
    if (!tuning().tb_fw || !tb_nograd_ok(T, g) || g.w < 6 || g.h < 6) return false;   // (the producers' unconditional window gather needs an interior)
    return (semantics == MI_SEM_CPU_REF && !fast_warp) || (semantics == MI_SEM_CUDA_COMPAT && fast_warp);
#endif
}

int iterate_tb_fused(int semantics, const float *I0, const float *I1, const float *cubic_tab_dev, int T, const IterPlanes &pl, const Geo &g,
                     float l_t, float theta, float taut, bool p_zero, int cur, hipStream_t s, bool skip_p_out)
{
#ifndef MIFLOW_EXPERIMENTS
    (void)semantics; (void)I0; (void)I1; (void)cubic_tab_dev; (void)T; (void)pl; (void)g; (void)l_t; (void)theta; (void)taut; (void)p_zero; (void)cur; (void)s; (void)skip_p_out;
    set_error("the warp fused into the pass kernel exists in the experiments build only (-DMIFLOW_EXPERIMENTS)");
    return MI_ERR_NOT_IMPL;
#else
    MI_REQUIRE(T == 10 && tb_nograd_ok(T, g), MI_ERR_BAD_ARG, "fused warp: one pass of the default T = 10 kernel only");
    const TbrEntry &e = g_tbr_fw[semantics == MI_SEM_CPU_REF ? 0 : 1];
    TbArgs A;
    memset(&A, 0, sizeof(A));
    A.pl = pl; A.pl.ix = A.pl.iy = A.pl.g = A.pl.rc = nullptr;
    A.g = g; A.l_t = l_t; A.theta = theta; A.taut = taut; A.cur = cur;
    A.skip_p_out = skip_p_out ? 1 : 0;
    A.fI0 = I0; A.fI1 = I1; A.ftab = cubic_tab_dev;
    A.rows_per_band = plan_band_rows(e, g);
    if (tuning().tb_verbose) {
        static int shown = 0;
        if (shown++ < 40) fprintf(stderr, "[tb] fused warp T=%d %dx%d batch=%d rows_per_band=%d\n", T, g.w, g.h, g.batch, A.rows_per_band);
    }
    return e.launch(A, p_zero, s);
#endif
}

int iterate_tb(int T, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut, bool p_zero,
               int cur, int rows_per_band, hipStream_t s, bool skip_p_out, bool independent_waves)
{
    if (pl.gamma != 0.f && rows_per_band == 0 && tile_eligible(g) && T <= tile_max_block() && !tuning().tb_force)
        return iterate_tile(-1, T, pl, g, l_t, theta, taut, p_zero, cur, s);   // small levels: the register tile with the channel (bit-identical)
    if (pl.gamma != 0.f) {   // the illumination channel: its own kernels (no |grad|^2 plane read, whether or not the warp stored one)
        const TbrEntry *e = nullptr;
        for (const TbrEntry &c : g_tbr_gam) if (c.T == T) e = &c;
        if (!e) { set_error("gamma != 0: unsupported time block %d", T); return MI_ERR_BAD_ARG; }
        MI_REQUIRE(pl.u[0][2] && pl.u[1][2] && pl.p[0][4] && pl.p[0][5] && pl.p[1][4] && pl.p[1][5], MI_ERR_BAD_ARG, "gamma != 0 needs the u3 / p31 / p32 planes");
        TbArgs A;
        memset(&A, 0, sizeof(A));
        A.pl = pl; A.g = g; A.l_t = l_t; A.theta = theta; A.taut = taut; A.cur = cur;
        A.skip_p_out = skip_p_out ? 1 : 0;
        A.rows_per_band = rows_per_band > 0 ? rows_per_band : plan_band_rows(*e, g);
        return e->launch(A, p_zero, s);
    }
    if (rows_per_band == 0 && tile_eligible(g) && T <= tile_max_block() && !tuning().tb_force) {
        MI_REQUIRE(pl.g, MI_ERR_BAD_ARG, "the register-tile kernel needs the |grad|^2 plane");
        return iterate_tile(-1, T, pl, g, l_t, theta, taut, p_zero, cur, s);
    }
    const TbrEntry *e = tbr_pick(T);
    if (independent_waves) {   // test hook: the first independent-wave entry of this block length
        e = nullptr;
        for (const TbrEntry &c : g_tbr) if (c.T == T && !e) e = &c;
    }
    if (!e) { set_error("unsupported time block %d", T); return MI_ERR_BAD_ARG; }
    if (!pl.g) {
        MI_REQUIRE(tb_nograd_ok(T, g), MI_ERR_BAD_ARG, "no |grad|^2 plane, but the kernel of this launch needs one");
#ifdef MIFLOW_EXPERIMENTS
        e = tuning().tb_p16 ? &g_tbr_ng16 : &g_tbr_ng;
#else
        e = &g_tbr_ng;
#endif
    }
    TbArgs A;
    A.pl = pl; A.g = g; A.l_t = l_t; A.theta = theta; A.taut = taut; A.cur = cur; A.swz = 0; A.nstrips = 0;
    A.skip_p_out = skip_p_out ? 1 : 0;
    A.rows_per_band = rows_per_band > 0 ? rows_per_band : plan_band_rows(*e, g);
    if (tuning().tb_verbose) {
        static int shown = 0;
        if (shown++ < 40)
            fprintf(stderr, "[tb] T=%d ppl=%d wps=%d pf=%d %dx%d batch=%d rows_per_band=%d\n", T, e->PPL, e->WPS, e->PF, g.w, g.h, g.batch,
                    A.rows_per_band);
    }
    return e->launch(A, p_zero, s);
}

// T fused EXACT iterations, set cur -> cur^1 (bit-identical to T launches of the one-iteration exact kernel).  T in 1..5.
int tb_exact_max_block() { return 5; }
int iterate_tb_exact(int T, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut, bool p_zero, int cur, hipStream_t s)
{
    const TbrEntry *e = nullptr;
    for (const TbrEntry &c : g_exact) if (c.T == T) e = &c;
    if (!e) { set_error("no exact-math kernel for time block %d", T); return MI_ERR_BAD_ARG; }
    TbArgs A;
    memset(&A, 0, sizeof(A));
    A.pl = pl; A.g = g; A.l_t = l_t; A.theta = theta; A.taut = taut; A.cur = cur;
    A.rows_per_band = plan_band_rows(*e, g);
    return e->launch(A, p_zero, s);
}

// Kernel block sizes of a warp's speculative steps (MODE 1 instantiations: 10 and 5).  A pass of the T = 5 kernel costs half
// of a T = 10 pass on the small pyramid levels and two thirds on a large one (profiles/r02k), and most warps settle within a few
// iterations: only the first warp of a LARGE level (px x pairs >= 12M) starts with the long kernel; after twelve short blocks
// the plan continues with long ones to bound the launch count.  It covers n + 30 iterations so that blocks cut short by the
// device's estimate cannot make the iteration limit unreachable.
int tb_spec_plan(int n, int warp_index, bool large_level, int *blocks, int max_blocks)
{
    int k = 0, sum = 0;
    if (n <= 0) return 0;
    const int want = n + (n > 10 ? 30 : n > 1 ? 10 : 0);
    while (sum < want && k < max_blocks) {
        const int t = n <= 5 ? 5 : (large_level && warp_index == 0) ? 10 : k < 12 ? 5 : 10;
        blocks[k++] = t;
        sum += t;
    }
    return k;
}

// One speculative step with kernel block size T (see k_iterate_tbr MODE 1); ctl carries the slot protocol, sk the step's
// constants, e0 the index of the first per-iteration error sum this launch may write.
int iterate_tb_spec(int T, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut, bool p_zero, const Ctl &ctl,
                    const SpecK &sk, int e0, hipStream_t s)
{
    // small levels: the same step on register tiles (serial depth of a launch = its iterations, not the image height); integer
    // error sums and identical per-pixel arithmetic => the same decisions and the same flows as the streaming kernel
    if (tile_eligible(g) && T <= tile_max_block() && !p_zero && tuning().tile_spec != 0)
        return iterate_tile_spec(T, pl, g, l_t, theta, taut, ctl, sk, e0, s);
    const TbrEntry *e = nullptr;
    if (pl.gamma != 0.f) {
        for (const TbrEntry &c : g_spec_gam) if (c.T == T) e = &c;
    } else if (!pl.g) {
        MI_REQUIRE(tb_spec_nograd_ok(g), MI_ERR_BAD_ARG, "no |grad|^2 plane, but the speculative kernel of this launch needs one");
        for (const TbrEntry &c : g_spec_jw_ng) if (c.T == T) e = &c;
    }
#ifdef MIFLOW_EXPERIMENTS
    else if (tuning().tb_jw >= 2 && tuning().tb_jw_spec) { for (const TbrEntry &c : g_spec_jw) if (c.T == T) e = &c; }
    else { for (const TbrEntry &c : g_spec) if (c.T == T) e = &c; }
#else
    else for (const TbrEntry &c : g_spec_jw_ng) if (c.T == T) e = &c;   // (the same kernel forms |grad|^2 itself and ignores the stored plane)
#endif
    if (!e) { set_error("no speculative kernel for time block %d", T); return MI_ERR_BAD_ARG; }
    TbArgs A;
    A.pl = pl; A.g = g; A.l_t = l_t; A.theta = theta; A.taut = taut; A.cur = 0; A.swz = 0; A.nstrips = 0; A.skip_p_out = 0;
    A.rows_per_band = plan_band_rows(*e, g);
    A.ctl = make_ctlk(&ctl);
    A.e0 = e0;
    A.sk = sk;
    return e->spec(A, p_zero, s);
}

// sticky fault flag of the joined-wave kernels (0 = no wait ever ran out of its budget); reading clears nothing
int tb_jw_fault(int *fault_host)
{
    // the flag is a per-DEVICE symbol: a multi-device caller (mi_tvl1_multi) must see a fault raised on any of its GPUs
    int cur = 0, n = 0, any = 0;
    MI_HIP_TRY(hipGetDevice(&cur));
    MI_HIP_TRY(hipGetDeviceCount(&n));
    for (int d = 0; d < n; ++d) {
        int f = 0;
        if (hipSetDevice(d) != hipSuccess) continue;
        const hipError_t e = hipMemcpyFromSymbol(&f, HIP_SYMBOL(g_jw_fault), sizeof(int), 0, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { (void)hipSetDevice(cur); MI_HIP_TRY(e); }
        any |= f;
    }
    MI_HIP_TRY(hipSetDevice(cur));
    *fault_host = any;
    return MI_OK;
}

// self-test of the DPP wave-shift semantics the kernels rely on (tests/test_tvl1_gpu.py::test_dpp_wave_shift_semantics)
__global__ void k_dbg_lane_shift(int *out)
{
    const int l = threadIdx.x;
    out[l] = __float_as_int(dpp_from_prev(__int_as_float(l + 100)));
    out[64 + l] = __float_as_int(dpp_from_next(__int_as_float(l + 100)));
}
int dbg_lane_shift(int *out_dev, hipStream_t s)
{
    hipLaunchKernelGGL(k_dbg_lane_shift, dim3(1), dim3(64), 0, s, out_dev);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace tvl1
}  // namespace mi
