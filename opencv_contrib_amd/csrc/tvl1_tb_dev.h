// Device helpers of the temporally blocked TV-L1 kernels (tvl1_tbr_kernels.hip).
#pragma once
#include "tvl1_dev.h"
#include "tvl1_warp_dev.h"

namespace mi {
namespace tvl1 {

// lane n <- lane n-1 (wave_shr:1) / lane n <- lane n+1 (wave_shl:1); semantics verified on HW
// by tests/test_tvl1_gpu.py::test_dpp_wave_shift_semantics through miflow_selftest_lane_shift.
__device__ __forceinline__ float dpp_from_prev(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_next(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

// The same with the lane that has no source lane (lane 0 / lane 63) taking `fill` instead of 0: the value the neighbouring wave of
// a joined group handed over through LDS (k_iterate_tbr JW).  bound_ctrl off = "keep the old value of the destination".
__device__ __forceinline__ float dpp_from_prev_fill(float v, float fill)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_next_fill(float v, float fill)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// Dynamic state of one pipeline stage: u_t(a) and p_(t-1)(a) of the row it holds.
template <int PPL>
struct Dyn {
    float u1[PPL], u2[PPL], p11[PPL], p12[PPL], p21[PPL], p22[PPL];
};
// gamma != 0 (illumination channel; cudaoptflow/src/cuda/tvl1flow.cu:209-288 u3 terms, :313-348 p31 / p32; CPU class
// optflow/src/tvl1flow.cpp:1011-1038, 1105-1110, 1172-1178): the third component of a stage's state.  Kernels without the channel never
// touch it and the registers do not exist.
template <int PPL>
struct Dyn3 {
    float u3[PPL], p31[PPL], p32[PPL];
};
template <int PPL>
struct Stat {  // static planes of one row: I1wx, I1wy, 1/grad, rho_c
    float ix[PPL], iy[PPL], rg[PPL], rc[PPL];
};

struct TbArgs {
    IterPlanes pl;
    Geo g;
    float l_t, theta, taut;
    int rows_per_band;
    int cur;  // input set
    int swz, nstrips;
    int skip_p_out;   // MODE 0: the launch does not store p (last pass of a scale)
    CtlK ctl;   // speculative convergence path only (MODE 1): slot protocol of Ctl, e0 = first error-sum index of this launch's block
    int e0;
    SpecK sk;
    // fused warp (k_iterate_tbr FW): the frames of the level and the CPU class's cubic phase table -- the producer waves of a workgroup
    // compute a row's I1wx, I1wy, rho_c from them (and from the flow the pass reads anyway); pl.ix / pl.iy / pl.rc are not touched
    const float *fI0, *fI1, *ftab;
};

#define ERR_FIX_SCALE 16777216.0 /* 2^24 fixed point for the deterministic error sum */

// One SPECULATIVE STEP of the convergence-checked path (epsilon > 0), shared by the streaming kernel (k_iterate_tbr MODE 1) and the
// register-tile kernel (k_iterate_tile SPEC): (1) settle the block the previous launch of this warp ran speculatively -- every
// workgroup, redundantly, from the same device data => the same decision; (2) pick this launch's work: a replay of exactly the
// iterations the reference would have run, the next speculative block (record = true: its per-iteration error sums are
// recorded), or nothing (returns false).  T = the most iterations this launch's kernel can run; `writer` = the one thread per
// pair that publishes the decision in the launch's control slot.  cur = the buffer set this launch reads.
__device__ __forceinline__ bool spec_settle(const CtlK &ck, const SpecK &sk, int T, int b, bool writer, int &cur, int &nit_out, bool &record)
{
    // (1) settle the block the previous launch of this warp ran speculatively (every workgroup, redundantly, from the same
    //     device data => the same decision); (2) pick this launch's work: replay, the next speculative block, or nothing.
    int base = 0, done = 0, n = 0, replay = 0, accepted = 0, pbase = 0, done_before = 0;
    double prev = 0.0;                    // cv::cuda's prevError
    float e_last = 0.f, e_before = 0.f;   // error / threshold of the last two accepted iterations (0: unknown)
    if (ck.q_prev >= 0) {
        const long long sp = (long long)b * ck.Q + ck.q_prev;
        const int2 sl = ck.S[sp];
        pbase = sl.x ^ (sl.y & MI_SLOT_FLIP);
        base = pbase;
        if (!ck.first_of_warp) {
            const int4 px = sk.X[sp];
            prev = ck.P[sp];
            n = px.x;
            e_last = __int_as_float(px.z);
            if (sl.y & MI_SLOT_DONE) {
                done = 1; done_before = 1;
            } else if (px.y > 0) {
                const int pn = px.y;
                int kk = 0, conv = 0;
                for (int t = 0; t < pn; ++t) {
                    const int na = n + t;
                    const bool calc = !ck.sched || ((na & 1) && prev < ck.thr);
                    const double e = (double)ck.E[(long long)b * ck.Q + sk.e0_prev + t] * (1.0 / ERR_FIX_SCALE);
                    e_before = e_last;
                    e_last = (float)(e / ck.thr);
                    if (calc) {
                        prev = e;
                        if (!(e > ck.thr)) { kk = t + 1; conv = 1; break; }
                    } else {
                        prev -= ck.thr;
                    }
                }
                if (conv && kk + sk.slack < pn) {   // the loop would have stopped inside the block: redo exactly kk iterations from its input
                    replay = kk; accepted = kk; n += kk; done = 1;
                } else {                 // the block stands
                    base = pbase ^ 1; accepted = pn; n += pn;
                    done = conv || n >= sk.iters;
                }
            }
        }
    }
    if (ck.reset_cur) base = pbase = 0;
    int nit = 0;
    if (replay) {
        nit = replay;
    } else if (!done && !sk.final_launch) {
        // block length: an estimate of the iterations still needed.  ANY value in [lo, hi] gives the same results; a good one
        // avoids both a replay (too long) and extra passes (too short).  A pass costs nearly the same whatever its length
        // (it is bound by its 64 B/px), so what counts is the number of passes.
        int pred = T;
        const int hist = sk.h_in ? sk.h_in[b] : 0;   // what this warp of this slot needed in the handle's previous calc
        if (!ck.sched && hist > n) {
            pred = hist - n;
        } else if (!ck.sched) {
            if (ck.first_of_warp) {
                if (sk.q_hist >= 0) pred = (sk.X[(long long)b * ck.Q + sk.q_hist].x * sk.hist_num) / sk.hist_den;
            } else if (e_before > e_last && e_last > 1.f) {
                pred = (int)ceilf(__logf(e_last) / __logf(e_before / e_last));   // geometric decay of the error sum
            } else if (e_last > 0.f) {
                pred = 2;
            }
        }
        const int hi = min(T, sk.iters - n), lo = max(1, sk.iters - n - sk.t_after);
        nit = max(lo, min(hi, pred));
        record = true;
    }
    if (writer) {
        const long long sq = (long long)b * ck.Q + ck.q;
        ck.S[sq] = make_int2(replay ? pbase : base, (replay ? MI_SLOT_FLIP : 0) | (done ? MI_SLOT_DONE : 0) | (accepted << 8));
        ck.P[sq] = prev;
        sk.X[sq] = make_int4(n, record ? nit : 0, __float_as_int(e_last), 0);
        if (sk.h_out) sk.h_out[b] = n;   // the warp's last launch leaves the final count
        if (sk.fb_flag) {   // {decision word, iterations accepted so far} per pair; the word is the release
            sk.fb_flag[2 * b + 1] = n;
            __hip_atomic_store(sk.fb_flag + 2 * b, (sk.fb_seq << 2) | (done_before << 1) | done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (nit == 0) return false;
    cur = replay ? pbase : base;
    nit_out = nit;
    return true;
}

// per-wave LDS ring of static rows: slot layout [plane 0..3][64*PPL floats]
template <int PPL>
__device__ __forceinline__ void lds_put(float *slot, int lane, const Stat<PPL> &s)
{
    float *q = slot + lane * PPL;
    if (PPL == 1) {
        q[0] = s.ix[0]; q[64] = s.iy[0]; q[128] = s.rg[0]; q[192] = s.rc[0];
    } else if (PPL == 2) {
        *reinterpret_cast<float2 *>(q) = make_float2(s.ix[0], s.ix[1]);
        *reinterpret_cast<float2 *>(q + 128) = make_float2(s.iy[0], s.iy[1]);
        *reinterpret_cast<float2 *>(q + 256) = make_float2(s.rg[0], s.rg[1]);
        *reinterpret_cast<float2 *>(q + 384) = make_float2(s.rc[0], s.rc[1]);
    } else {
        *reinterpret_cast<float4 *>(q) = make_float4(s.ix[0], s.ix[1], s.ix[2], s.ix[3]);
        *reinterpret_cast<float4 *>(q + 256) = make_float4(s.iy[0], s.iy[1], s.iy[2], s.iy[3]);
        *reinterpret_cast<float4 *>(q + 512) = make_float4(s.rg[0], s.rg[1], s.rg[2], s.rg[3]);
        *reinterpret_cast<float4 *>(q + 768) = make_float4(s.rc[0], s.rc[1], s.rc[2], s.rc[3]);
    }
}
template <int PPL>
__device__ __forceinline__ void lds_get(const float *slot, int lane, Stat<PPL> &s)
{
    const float *q = slot + lane * PPL;
    if (PPL == 1) {
        s.ix[0] = q[0]; s.iy[0] = q[64]; s.rg[0] = q[128]; s.rc[0] = q[192];
    } else if (PPL == 2) {
        float2 a = *reinterpret_cast<const float2 *>(q), b = *reinterpret_cast<const float2 *>(q + 128);
        float2 c = *reinterpret_cast<const float2 *>(q + 256), d = *reinterpret_cast<const float2 *>(q + 384);
        s.ix[0] = a.x; s.ix[1] = a.y; s.iy[0] = b.x; s.iy[1] = b.y;
        s.rg[0] = c.x; s.rg[1] = c.y; s.rc[0] = d.x; s.rc[1] = d.y;
    } else {
        float4 a = *reinterpret_cast<const float4 *>(q), b = *reinterpret_cast<const float4 *>(q + 256);
        float4 c = *reinterpret_cast<const float4 *>(q + 512), d = *reinterpret_cast<const float4 *>(q + 768);
        s.ix[0] = a.x; s.ix[1] = a.y; s.ix[2] = a.z; s.ix[3] = a.w;
        s.iy[0] = b.x; s.iy[1] = b.y; s.iy[2] = b.z; s.iy[3] = b.w;
        s.rg[0] = c.x; s.rg[1] = c.y; s.rg[2] = c.z; s.rg[3] = c.w;
        s.rc[0] = d.x; s.rc[1] = d.y; s.rc[2] = d.z; s.rc[3] = d.w;
    }
}

// 1/grad; grad == 0 -> huge, so that clamp() yields -+l_t*sign(rho) like the reference's first two branches
template <int PPL>
__device__ __forceinline__ void finish_static(Stat<PPL> &st)
{
#pragma unroll
    for (int j = 0; j < PPL; ++j) st.rg[j] = __builtin_amdgcn_rcpf(fmaxf(st.rg[j], 1e-30f));
}

}  // namespace tvl1
}  // namespace mi
