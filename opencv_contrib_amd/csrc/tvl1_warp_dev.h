// Device-side loop-control view and the argument block of the fused-gradient warp kernel; shared by tvl1_kernels.hip and
// tvl1_warp_kernels.hip.  Not part of the C-ABI.
#pragma once
#include "tvl1_dev.h"

namespace mi {
namespace tvl1 {

struct CtlK {  // by-value copy for kernels (Ctl may be absent)
    int2 *S;         // per slot {cur_in, flags}: flags bit 0 = the launch was active, bit 1 = it summed the error
    unsigned long long *E;
    int Q, q, q_prev, first_of_warp, reset_cur;
    double thr;
    double *P;
    int sched, n, need_done;
};
static inline CtlK make_ctlk(const Ctl *c)
{
    CtlK k;
    memset(&k, 0, sizeof(k));
    k.q_prev = -1;
    if (c) { k.S = c->S; k.E = c->E; k.Q = c->Q; k.q = c->q; k.q_prev = c->q_prev;
             k.first_of_warp = c->first_of_warp; k.reset_cur = c->reset_cur; k.thr = c->thr;
             k.P = c->P; k.sched = c->sched; k.n = c->n; k.need_done = c->need_done; }
    return k;
}
// Ctl::need_done: false = this pair's previous warp has not stopped yet, the launch came too early for it
__device__ __forceinline__ bool warp_gate_k(const CtlK &c, int b)
{
    if (!c.S || !c.need_done || c.q_prev < 0) return true;
    return (c.S[(long long)b * c.Q + c.q_prev].y & MI_SLOT_DONE) != 0;
}
__device__ __forceinline__ int resolve_cur_k(const CtlK &c, int b, int cur_host)
{
    if (!c.S) return cur_host;
    if (c.q_prev < 0) return 0;
    const int2 s = c.S[(long long)b * c.Q + c.q_prev];
    return s.x ^ (s.y & 1);
}

// The first warp of a scale may ZOOM the coarser scale's flow itself (round 4): its pixel's flow is sampled from the coarse planes with
// k_resize's arithmetic (resize_dev.h), multiplied by 1 / scaleStep (tvl1flow.cpp:291-300), stored into the scale's own u planes (the
// iteration pass reads them) and used at once -- the separate resize launch and its 8 B/px write + 8 B/px re-read are gone.
struct WarpUp {
    const float *u1c, *u2c;   // coarse planes (null: no zoom, the warp reads u1 / u2 of the scale)
    float *u1o, *u2o;         // the scale's flow planes (set 0) the zoomed flow is written to
    int cw, ch, cld;          // coarse geometry
    long long cps;            // pair stride of the coarse planes
    double scx, scy;          // scale factors as tvl1::resize forms them for the semantics
    float post;
};
struct Warp6Args {
    const float *I0, *I1;
    const float *u1[2], *u2[2];
    float *I1w, *I1wx, *I1wy, *grad, *rho;
    const float *tab;  // 32x4 cubic phase table (CPU_REF)
    Geo g;
    WarpUp up;
    int swz;           // XCD-contiguous tile order (MIFLOW_WARP_SWZ, default 1)
};

// pixels of a wave along x in the warp kernels (a 32 x 2 patch per wave measured best, profiles/r01u)
static inline int warp_tile() { return tuning().warp_tile; }

}  // namespace tvl1
}  // namespace mi
