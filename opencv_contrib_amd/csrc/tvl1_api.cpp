// Dual TV-L1: handle, scratch arena, pyramid orchestration and the C-ABI entry points.
// Host-side twin of OpticalFlowDual_TVL1_Impl::calcImpl / procOneScale
// (modules/cudaoptflow/src/tvl1flow.cpp:185-382; CPU modules/optflow/src/tvl1flow.cpp:402-533,
// 1313-1408) -- but fully stream-ordered: no host read-back inside the iteration loop.
#include "tvl1_dev.h"
#include "mi_selftest.h"
#include <cfloat>
#include <algorithm>
#include <cmath>
#include <vector>
#include <chrono>
#include <sched.h>
#include <time.h>

using namespace mi;
using namespace mi::tvl1;

namespace {

constexpr double kFeedbackWaitLimitUs = 30e6;   // polled host feedback: a decision word that has not arrived after 30 s never will

struct LevelBuf {
    Geo g;
    float *I0 = nullptr, *I1 = nullptr;
    float *u[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};  // [set][u1,u2,u3]
};

struct SlotInfo { int scale, warp; };

}  // namespace

// Everything one sub-batch needs: its own arena, pointer table, control slots and profiling events, so that two lanes can
// run concurrently on two internal streams (the warp kernel is bound by the vector-memory path, the blocked iteration
// kernel by VALU issue: they overlap, profiles/r02*).
struct Lane {
    // capacity the arena was built for
    int capW = 0, capH = 0, capB = 0, capScales = 0;
    double capStep = 0;
    bool capGamma = false, capMedian = false, capPack = false;
    float *arena = nullptr;
    size_t arena_floats = 0, arena_bytes = 0;
    std::vector<LevelBuf> L;
    // full-resolution-capacity scratch planes (re-laid-out densely per level)
    float *scr[6] = {};    // scr[0..1]: median-filter temporaries; I1wx, I1wy, grad, rho_c
    float *pack = nullptr; // MIFLOW_WARP=pk only: float4 {I1, I1x, I1y, 0} per pixel of the current level
    float *pbuf[2][6] = {};   // [set][p11,p12,p21,p22,p31,p32]
    PtrTab *tab_dev = nullptr;
    int tab_cap = 0;
    std::vector<PtrTab> tab_host;   // source of the asynchronous upload: must outlive the call
    // device loop control
    int2 *S = nullptr;
    unsigned long long *E = nullptr;
    double *Pd = nullptr;   // per slot prevError (cv::cuda check schedule)
    int4 *X = nullptr;      // per slot state of the speculative steps (SpecK::X)
    long long Q = 0;
    int ctlB = 0;
    // iteration counts per (scale, warp, pair slot) of the previous calc, two parities (SpecK::h_in / h_out)
    int *H = nullptr;
    size_t H_cap = 0;             // ints per parity
    unsigned long long H_sig = 0; // geometry / batch / loop shape the counts belong to (0: none)
    int H_par = 0;                // parity the NEXT calc writes
    std::vector<SlotInfo> slots;
    int batch = 0;          // pairs of the last calc
    // host feedback (mi_tvl1_params.host_feedback): pinned landing area of the control slots read back between launches
    int2 *fb_host = nullptr;
    int fb_cap = 0;
    hipEvent_t fb_ev = nullptr;
    int *fb_flag = nullptr;   // polled form (SpecK::fb_flag): {decision word, count} per pair, pinned
    int fb_seq = 0;
    std::vector<int> fb_hist;        // polled form: most iterations a (scale, warp) of the previous calc needed over its pairs (0: unknown)
    unsigned long long fb_hist_sig = 0;
    long long fb_waits = 0, fb_skipped = 0;   // of the last calc: host waits, launches not enqueued
    // profiling (mi_tvl1_set_profiling)
    std::vector<hipEvent_t> ev_pool;
    struct Region { int e0, e1; long long launches; double bytes; int kind; int level; };   // kind 0: iteration launches, 1: warp launch; level = pyramid scale
    std::vector<Region> regions;
    // internal stream of a concurrent lane + its completion event
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t idle_ev = nullptr;        // recorded behind the lane's last calc on the stream it ran on
    bool idle_valid = false;             // ... and usable (not recorded under stream capture, no error)
    bool used = false;                   // a calc has been enqueued on the current arena
};

struct mi_tvl1 {
    mi_tvl1_params P;
    int device = 0;
    float *cubic_tab = nullptr;
    static const int kMaxLanes = 4;
    Lane lane[kMaxLanes];
    hipEvent_t fork = nullptr;
    int last_nscales = 0, last_batch = 0, last_lanes = 1;
    int last_first[kMaxLanes + 1] = {};   // pairs [last_first[i], last_first[i + 1]) ran on lane i
    bool last_check = false;
    bool profiling = false;
};

static int round_half_even(double v) { return (int)std::lrint(v); }

void mi_tvl1_default_params(mi_tvl1_params *p)
{
    if (!p) return;
    // cv::cuda::OpticalFlowDual_TVL1::create defaults, cudaoptflow.hpp:375-385
    p->tau = 0.25; p->lambda = 0.15; p->theta = 0.3; p->epsilon = 0.01; p->scale_step = 0.8; p->gamma = 0.0;
    p->nscales = 5; p->warps = 5; p->iterations = 300; p->use_initial_flow = 0;
    p->inner_iterations = 1; p->median_filtering = 1;
    // SEMANTICS OF A DEFAULT-CONSTRUCTED OBJECT (read this before dropping the class in).  The acceptance reference of this
    // path is the CPU class cv::optflow::DualTVL1OpticalFlow ("EPE vs CPU ref", BASELINE.json), so the default arithmetic is
    // the CPU class's: cv::remap(INTER_CUBIC, a = -0.75, 1/32-px phases, constant-0 border) warp, cv::resize pyramid
    // (half-pixel centres), convergence test after every iteration with a float threshold.  cv::cuda's own kernels differ from
    // that -- normalised a = -0.5 bicubic with clamp addressing, cuda::resize without the half-pixel shift, a sparse check
    // schedule -- by 0.07-0.14 px mean EPE on the synthetic pairs used here (mostly at image borders the flow leaves);
    // semantics = MI_SEM_CUDA_COMPAT selects exactly that arithmetic (pinned bit for bit on the reference's OpenCL twins of the
    // CUDA kernels, tests/test_ref_pin.py) for callers validated against cv::cuda.  Math: fast device math (v_rcp / v_sqrt /
    // fma, iterations fused per HBM pass) by default, which the reference's own test tolerates (CUDA vs CPU |1 - CCORR| <= 4e-3,
    // test_optflow.cpp:465; here <= 1e-4 and mean EPE <= 5e-3 px against the oracle); exact_math = 1 performs the separately
    // rounded IEEE operations of the reference in its order.
    p->semantics = MI_SEM_CPU_REF; p->exact_math = 0; p->time_block = 0; p->lanes = 0; p->stop_slack = 0; p->host_feedback = 0;
}

// upper bound of control slots per pair (scales x warps x iterations) a convergence-checked calc may enqueue
static const long long kMaxSlots = 4000000;

static int validate_params(const mi_tvl1_params *p)
{
    MI_REQUIRE(p, MI_ERR_BAD_ARG, "null params");
    MI_REQUIRE(p->nscales > 0 && p->nscales <= 32, MI_ERR_BAD_ARG, "nscales must be in [1,32] (CV_Assert nscales_ > 0)");
    MI_REQUIRE(p->warps >= 0 && p->warps <= 64, MI_ERR_BAD_ARG, "warps must be in [0,64]");
    MI_REQUIRE(p->iterations >= 0 && p->inner_iterations >= 0, MI_ERR_BAD_ARG, "negative iteration count");
    MI_REQUIRE((long long)p->iterations * p->inner_iterations <= kMaxSlots, MI_ERR_BAD_ARG,
               "iterations x inner_iterations must not exceed %lld", kMaxSlots);
    MI_REQUIRE(p->scale_step > 0 && p->scale_step < 1, MI_ERR_BAD_ARG, "scale_step must be in (0,1)");
    MI_REQUIRE(p->theta != 0, MI_ERR_BAD_ARG, "theta must be non-zero");
    MI_REQUIRE(p->semantics == MI_SEM_CPU_REF || p->semantics == MI_SEM_CUDA_COMPAT, MI_ERR_BAD_ARG, "bad semantics");
    MI_REQUIRE(p->median_filtering <= 1 || p->median_filtering == 3 || p->median_filtering == 5, MI_ERR_BAD_ARG,
               "medianFiltering must be 1 (off), 3 or 5 (cv::medianBlur on CV_32F)");
    MI_REQUIRE(p->lanes >= 0 && p->lanes <= 4, MI_ERR_BAD_ARG, "lanes must be 0 (automatic) or 1..4");
    MI_REQUIRE(p->stop_slack >= 0 && p->stop_slack <= 8, MI_ERR_BAD_ARG, "stop_slack must be in 0..8");
    MI_REQUIRE(p->host_feedback >= -1 && p->host_feedback <= 1, MI_ERR_BAD_ARG, "host_feedback must be -1, 0 or 1");
    return MI_OK;
}

int mi_tvl1_create(const mi_tvl1_params *p, mi_tvl1 **out)
{
    MI_REQUIRE(out, MI_ERR_BAD_ARG, "null out");
    *out = nullptr;
    mi_tvl1_params d;
    if (!p) { mi_tvl1_default_params(&d); p = &d; }
    int rc = validate_params(p);
    if (rc) return rc;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device available: the miflow product path has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    mi_tvl1 *h = new mi_tvl1();
    h->P = *p;
    float tab[128];
    host_cubic_table(tab);
    auto upload = [&]() -> int {
        MI_HIP_TRY(hipGetDevice(&h->device));
        MI_HIP_TRY(hipMalloc((void **)&h->cubic_tab, sizeof(tab)));
        MI_HIP_TRY(hipMemcpy(h->cubic_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
        return MI_OK;
    };
    if (const int rc = upload()) { mi_tvl1_destroy(h); return rc; }
    *out = h;
    return MI_OK;
}

int mi_tvl1_set_params(mi_tvl1 *h, const mi_tvl1_params *p)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    int rc = validate_params(p);
    if (rc) return rc;
    h->P = *p;
    return MI_OK;
}

int mi_tvl1_get_params(const mi_tvl1 *h, mi_tvl1_params *p)
{
    MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument");
    *p = h->P;
    return MI_OK;
}

static void free_arena(Lane &ln)
{
    // the block goes back to the cache with the event the lane recorded behind its last calc (while that calc's stream was alive):
    // whoever takes the block next waits for that event only -- destroying one handle no longer stalls the other handles' streams
    if (ln.arena) {
        big_free(ln.arena, ln.arena_bytes, ln.idle_valid ? ln.idle_ev : nullptr, ln.used);
        if (ln.idle_valid) ln.idle_ev = nullptr;   // ownership went with the block
    }
    ln.idle_valid = false;
    ln.used = false;
    ln.arena = nullptr;
    ln.L.clear();
}

int mi_tvl1_query_plan(int width, int height, int pairs_per_lane, int iterations_per_launch, int *kernel, int *rows_per_band)
{
    MI_REQUIRE(kernel && rows_per_band, MI_ERR_BAD_ARG, "null output");
    MI_REQUIRE(width > 0 && height > 0 && pairs_per_lane > 0 && iterations_per_launch > 0, MI_ERR_BAD_ARG, "bad plan query");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("no HIP device"); return MI_ERR_NO_DEVICE; }
    Geo g;
    g.w = width; g.h = height; g.ld = (width + 63) / 64 * 64; g.ps = (long long)g.ld * height; g.batch = pairs_per_lane;
    return tb_query_plan(iterations_per_launch, g, kernel, rows_per_band);
}

int mi_tvl1_set_profiling(mi_tvl1 *h, int enable)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    h->profiling = enable != 0;
    return MI_OK;
}

int mi_tvl1_get_profile(mi_tvl1 *h, double *ms_total, long long *launches, double *algo_bytes)
{
    return mi_tvl1_get_profile_kind(h, 0, ms_total, launches, algo_bytes);
}

int mi_tvl1_get_profile_kind(mi_tvl1 *h, int kind, double *ms_total, long long *launches, double *algo_bytes)
{
    return mi_tvl1_get_profile_level(h, kind, -1, ms_total, launches, algo_bytes);
}

int mi_tvl1_get_profile_level(mi_tvl1 *h, int kind, int level, double *ms_total, long long *launches, double *algo_bytes)
{
    MI_REQUIRE(h && ms_total && launches && algo_bytes, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(kind == 0 || kind == 1, MI_ERR_BAD_ARG, "kind must be 0 (iteration launches) or 1 (warp launches)");
    *ms_total = 0; *launches = 0; *algo_bytes = 0;
    for (int li = 0; li < h->last_lanes; ++li) {
        Lane &ln = h->lane[li];
        for (const auto &r : ln.regions) {
            if (r.kind != kind || (level >= 0 && r.level != level)) continue;
            MI_HIP_TRY(hipEventSynchronize(ln.ev_pool[r.e1]));
            float ms = 0.f;
            MI_HIP_TRY(hipEventElapsedTime(&ms, ln.ev_pool[r.e0], ln.ev_pool[r.e1]));
            *ms_total += ms; *launches += r.launches; *algo_bytes += r.bytes;
        }
    }
    return MI_OK;
}

void mi_tvl1_destroy(mi_tvl1 *h)
{
    if (!h) return;
    for (Lane &ln : h->lane) {
        for (hipEvent_t e : ln.ev_pool) (void)hipEventDestroy(e);
        free_arena(ln);
        if (ln.idle_ev) { (void)hipEventDestroy(ln.idle_ev); ln.idle_ev = nullptr; }
        if (ln.tab_dev) (void)hipFree(ln.tab_dev);
        if (ln.S) (void)hipFree(ln.S);
        if (ln.E) (void)hipFree(ln.E);
        if (ln.Pd) (void)hipFree(ln.Pd);
        if (ln.X) (void)hipFree(ln.X);
        if (ln.H) (void)hipFree(ln.H);
        if (ln.done) (void)hipEventDestroy(ln.done);
        if (ln.fb_ev) (void)hipEventDestroy(ln.fb_ev);
        if (ln.fb_host) (void)hipHostFree(ln.fb_host);
        if (ln.fb_flag) (void)hipHostFree(ln.fb_flag);
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
    }
    if (h->fork) (void)hipEventDestroy(h->fork);
    if (h->cubic_tab) (void)hipFree(h->cubic_tab);
    delete h;
}

// Level sizes: dsize = saturate_cast<int>(ssize * scaleStep) (cudawarping/src/resize.cpp:78),
// stop below 16 px (cudaoptflow/src/tvl1flow.cpp:243-247).
static int plan_levels(const mi_tvl1_params &P, int W, int H, int B, std::vector<Geo> &geo)
{
    geo.clear();
    int w = W, h = H;
    for (int s = 0; s < P.nscales; ++s) {
        if (s > 0) {
            w = round_half_even(w * P.scale_step);
            h = round_half_even(h * P.scale_step);
            if (w < 1 || h < 1) break;
        }
        Geo g;
        g.w = w; g.h = h; g.ld = align_up(w, 64); g.ps = (long long)g.ld * h; g.batch = B;
        geo.push_back(g);
        if (s > 0 && (w < 16 || h < 16)) break;  // level s is built but not used
    }
    return (int)geo.size();
}

static int ensure_arena(const mi_tvl1_params &P, Lane &ln, int W, int H, int B)
{
    const bool gam = P.gamma != 0.0;
    const bool med = P.median_filtering > 1;
    const bool pk = tuning().warp_legacy != 0;
    if (ln.arena && ln.capW == W && ln.capH == H && ln.capB >= B && ln.capScales == P.nscales && ln.capStep == P.scale_step &&
        ln.capGamma == gam && ln.capMedian == med && ln.capPack == pk)
        return MI_OK;
    free_arena(ln);
    std::vector<Geo> geo;
    const int nl = plan_levels(P, W, H, B, geo);
    size_t total = 0;
    auto take = [&](size_t nfloats) { size_t o = total; total += (nfloats + 63) / 64 * 64; return o; };
    std::vector<size_t> offI0(nl), offI1(nl), offU(nl * 6);
    for (int l = 0; l < nl; ++l) {
        const size_t n = (size_t)geo[l].ps * B;
        offI0[l] = take(n); offI1[l] = take(n);
        for (int k = 0; k < 6; ++k) offU[l * 6 + k] = (k % 3 == 2 && !gam) ? 0 : take(n);   // u3 planes only when gamma != 0
    }
    const size_t nfull = (size_t)geo[0].ps * B;
    size_t offScr[6], offP[12];
    for (int k = 0; k < 6; ++k) offScr[k] = (k < 2 && !med) ? 0 : take(nfull);   // scr[0..1]: median-filter temporaries
    const size_t offPack = pk ? take(nfull * 4) : 0;
    for (int k = 0; k < 12; ++k) offP[k] = (k % 6 >= 4 && !gam) ? 0 : take(nfull);
    {
        void *blk = nullptr;
        const int brc = big_alloc(&blk, total * sizeof(float), &ln.arena_bytes);
        if (brc) return brc;
        ln.arena = (float *)blk;
    }
    ln.arena_floats = total;
    ln.L.resize(nl);
    for (int l = 0; l < nl; ++l) {
        ln.L[l].g = geo[l];
        ln.L[l].I0 = ln.arena + offI0[l];
        ln.L[l].I1 = ln.arena + offI1[l];
        for (int k = 0; k < 6; ++k) ln.L[l].u[k / 3][k % 3] = (k % 3 == 2 && !gam) ? nullptr : ln.arena + offU[l * 6 + k];
    }
    for (int k = 0; k < 6; ++k) ln.scr[k] = ln.arena + offScr[k];
    ln.pack = pk ? ln.arena + offPack : nullptr;
    for (int k = 0; k < 12; ++k) ln.pbuf[k / 6][k % 6] = (k % 6 >= 4 && !gam) ? nullptr : ln.arena + offP[k];
    ln.capGamma = gam; ln.capMedian = med; ln.capPack = pk;
    ln.capW = W; ln.capH = H; ln.capB = B; ln.capScales = P.nscales; ln.capStep = P.scale_step;
    return MI_OK;
}

static int check_pair(const mi_mat *I0, const mi_mat *I1, const mi_mat *flow, const mi_mat *I0ref)
{
    MI_REQUIRE(I0 && I1 && flow, MI_ERR_BAD_ARG, "null matrix");
    MI_REQUIRE(I0->data && I1->data && flow->data, MI_ERR_BAD_ARG, "null data pointer");
    // CV_Assert( I0.type() == CV_8UC1 || I0.type() == CV_32FC1 )  tvl1flow.cpp:187
    MI_REQUIRE(I0->type == MI_8UC1 || I0->type == MI_32FC1, MI_ERR_BAD_TYPE, "I0 must be CV_8UC1 or CV_32FC1");
    MI_REQUIRE(I0->rows == I1->rows && I0->cols == I1->cols, MI_ERR_BAD_SIZE, "I0.size() != I1.size()");  // :188
    MI_REQUIRE(I0->type == I1->type, MI_ERR_BAD_TYPE, "I0.type() != I1.type()");                          // :189
    MI_REQUIRE(flow->type == MI_32FC2, MI_ERR_BAD_TYPE, "flow must be CV_32FC2");
    MI_REQUIRE(flow->rows == I0->rows && flow->cols == I0->cols, MI_ERR_BAD_SIZE, "flow.size() != I0.size()");  // :190
    MI_REQUIRE(I0->rows >= 3 && I0->cols >= 3, MI_ERR_BAD_SIZE, "image must be at least 3x3");
    // the dense float planes are addressed with 32-bit byte offsets from a per-pair base (buffer loads of the warp's border windows)
    MI_REQUIRE((long long)((I0->cols + 63) / 64 * 64) * I0->rows * 4 < (1LL << 32), MI_ERR_BAD_SIZE,
               "image too large: a float plane of it must stay below 4 GiB");
    const size_t es = I0->type == MI_8UC1 ? 1 : 4;
    MI_REQUIRE(I0->step >= (size_t)I0->cols * es && I1->step >= (size_t)I1->cols * es, MI_ERR_BAD_ARG, "step < cols*elemSize");
    MI_REQUIRE(flow->step >= (size_t)flow->cols * 8, MI_ERR_BAD_ARG, "flow step < cols*8");
    if (es == 4) MI_REQUIRE(I0->step % 4 == 0 && I1->step % 4 == 0 && ((uintptr_t)I0->data % 4) == 0 && ((uintptr_t)I1->data % 4) == 0,
                            MI_ERR_BAD_ARG, "float images must be 4-byte aligned");
    MI_REQUIRE(flow->step % 8 == 0 && ((uintptr_t)flow->data % 8) == 0, MI_ERR_BAD_ARG, "flow must be 8-byte aligned");
    MI_REQUIRE(I0->rows == I0ref->rows && I0->cols == I0ref->cols && I0->type == I0ref->type, MI_ERR_BAD_SIZE,
               "all pairs of a batch must share size and type");
    return MI_OK;
}

// The whole coarse-to-fine computation of n pairs on one stream (OpticalFlowDual_TVL1_Impl::calcImpl + procOneScale).
static int lane_calc(mi_tvl1 *h, Lane &ln, int n, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows, hipStream_t st, int *ns_out)
{
    // behind everything this call enqueues for the lane (also on an error return: part of the calc may be in flight): the event the
    // arena cache waits for before it hands the lane's block to somebody else
    struct Mark {
        Lane &l; hipStream_t s;
        ~Mark()
        {
            if (!l.arena) return;
            l.used = true; l.idle_valid = false;
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
            if (!l.idle_ev && hipEventCreateWithFlags(&l.idle_ev, hipEventDisableTiming) != hipSuccess) { l.idle_ev = nullptr; (void)hipGetLastError(); return; }
            if (hipEventRecord(l.idle_ev, s) == hipSuccess) l.idle_valid = true; else (void)hipGetLastError();
        }
    } mark{ln, st};
    const mi_tvl1_params &P = h->P;
    const int W = I0s[0].cols, H = I0s[0].rows, B = n;
    int rc = ensure_arena(P, ln, W, H, B);
    if (rc) return rc;
    ln.batch = B;
    const int nl_built = (int)ln.L.size();
    // number of usable scales (tvl1flow.cpp:243-247)
    int ns = nl_built;
    if (ns > 1 && (ln.L[ns - 1].g.w < 16 || ln.L[ns - 1].g.h < 16)) ns -= 1;
    *ns_out = ns;
    for (int l = 0; l < nl_built; ++l) ln.L[l].g.batch = B;

    // external pointer table
    if (ln.tab_cap < B) {
        if (ln.tab_dev) (void)hipFree(ln.tab_dev);
        ln.tab_dev = nullptr; ln.tab_cap = 0;
        MI_HIP_TRY(hipMalloc((void **)&ln.tab_dev, sizeof(PtrTab) * B));
        ln.tab_cap = B;
    }
    {
        std::vector<PtrTab> &tab = ln.tab_host;
        tab.resize(B);
        for (int i = 0; i < B; ++i) {
            tab[i].a = I0s[i].data; tab[i].b = I1s[i].data; tab[i].out = flows[i].data;
            tab[i].step_a = (long long)I0s[i].step; tab[i].step_b = (long long)I1s[i].step;
            tab[i].step_out = (long long)flows[i].step;
        }
        // pageable source: a few KB, staged by the runtime before the call returns; kept in the lane anyway
        MI_HIP_TRY(hipMemcpyAsync(ln.tab_dev, tab.data(), sizeof(PtrTab) * B, hipMemcpyHostToDevice, st));
    }

    const int iters_per_warp = P.iterations * P.inner_iterations;
    const bool check = P.epsilon > 0.0 && iters_per_warp > 0;
    // convergence-checked path in fast math: speculative blocks (k_iterate_tbr MODE 1 / 2) instead of one launch per iteration
    // (gamma != 0 -- round 6: the illumination channel runs the same speculative steps on its own kernels, tvl1_tbr_kernels.hip GAM)
    const bool spec = check && !P.exact_math && P.time_block != 1 && P.median_filtering <= 1 && tuning().spec != 0;
    // control slots per pair: one per launch (S, P, X) and one error sum per iteration (E); both index spaces fit max(.,.)
    long long Q = (long long)ns * P.warps * iters_per_warp;
    std::vector<int> spec_plan[4];   // kernel block sizes of the speculative steps: first warp of a large level / of a small one / later warps / levels on the register-tile kernel
    const double kLargeLevel = 12e6;   // px x pairs from which the T = 10 kernel pays for a long first block
    if (spec) {
        long long e_max = 0, l_max = 0;
        for (int k = 0; k < 4; ++k) {
            spec_plan[k].resize(iters_per_warp + 40);
            if (k == 3) {
                // register-tile kernel: a launch costs its iterations, not its block size, and a converged warp pays ~3 us per
                // remaining (empty) launch -- the fewest launches: blocks of the kernel's margin
                int kk = 0, sum = 0;
                const int want = iters_per_warp + (iters_per_warp > 10 ? 30 : iters_per_warp > 1 ? 10 : 0);
                // one or two pairs (host feedback: launches behind the stop are never enqueued): shorter blocks on tiles of a
                // smaller margin own more of their 64 columns x rows (MIFLOW_TILE_FB_BLOCK)
                const int tb = (B <= 2 && tuning().tile_fb_block > 0) ? std::min(tuning().tile_fb_block, tile_max_block()) : tile_max_block();
                while (sum < want && kk < (int)spec_plan[k].size()) { spec_plan[k][kk++] = tb; sum += tb; }
                spec_plan[k].resize(kk);
            } else
            spec_plan[k].resize(tb_spec_plan(iters_per_warp, k == 2 ? 1 : 0, k == 0, spec_plan[k].data(), (int)spec_plan[k].size()));
            long long t = 0;
            for (int v : spec_plan[k]) t += v;
            e_max = std::max(e_max, t);
            l_max = std::max(l_max, (long long)spec_plan[k].size() + 1);
        }
        // the cost-model plan of a one-or-two-pair calc (below) rounds a warp's total UP to a multiple of its block length, and the
        // settling launch of a warp indexes one more block: a block of slack per warp (the launch loop checks the bound as well)
        e_max += 2 * (long long)std::max(tile_max_block(), tb_max_block());
        Q = std::max((long long)ns * P.warps * e_max, (long long)ns * P.warps * l_max);
    }
    if (check) {
        MI_REQUIRE(Q <= kMaxSlots, MI_ERR_BAD_ARG, "scales x warps x iterations = %lld control slots exceed the limit of %lld", Q, kMaxSlots);
        if (ln.Q < Q || ln.ctlB < B) {
            if (ln.S) (void)hipFree(ln.S);
            if (ln.E) (void)hipFree(ln.E);
            if (ln.Pd) (void)hipFree(ln.Pd);
            if (ln.X) (void)hipFree(ln.X);
            ln.S = nullptr; ln.E = nullptr; ln.Pd = nullptr; ln.X = nullptr; ln.Q = 0; ln.ctlB = 0;
            MI_HIP_TRY(hipMalloc((void **)&ln.S, sizeof(int2) * (size_t)Q * B));
            MI_HIP_TRY(hipMalloc((void **)&ln.E, sizeof(unsigned long long) * (size_t)Q * B));
            MI_HIP_TRY(hipMalloc((void **)&ln.Pd, sizeof(double) * (size_t)Q * B));
            MI_HIP_TRY(hipMalloc((void **)&ln.X, sizeof(int4) * (size_t)Q * B));
            ln.Q = Q; ln.ctlB = B;
        }
        MI_HIP_TRY(hipMemsetAsync(ln.E, 0, sizeof(unsigned long long) * (size_t)ln.Q * B, st));
        MI_HIP_TRY(hipMemsetAsync(ln.S, 0, sizeof(int2) * (size_t)ln.Q * B, st));
    }
    // history of the previous calc of this lane (speculative steps only)
    const size_t h_n = (size_t)ns * P.warps * B;
    bool hist = spec && tuning().tb_hist != 0;
    if (hist) {
        unsigned long long sig = 1469598103934665603ull;
        for (long long v : {(long long)W, (long long)H, (long long)B, (long long)ns, (long long)P.warps, (long long)iters_per_warp, (long long)I0s[0].type}) sig = (sig ^ (unsigned long long)v) * 1099511628211ull;
        if (ln.H_cap < h_n) {
            if (ln.H) (void)hipFree(ln.H);
            ln.H = nullptr; ln.H_cap = 0; ln.H_sig = 0;
            MI_HIP_TRY(hipMalloc((void **)&ln.H, sizeof(int) * 2 * h_n));
            ln.H_cap = h_n;
        }
        if (ln.H_sig != sig) {   // counts of another geometry say nothing: start from none (0 = no estimate)
            MI_HIP_TRY(hipMemsetAsync(ln.H, 0, sizeof(int) * 2 * ln.H_cap, st));
            ln.H_sig = sig;
        }
        if (ln.fb_hist_sig != sig || ln.fb_hist.size() != (size_t)ns * P.warps) {
            ln.fb_hist.assign((size_t)ns * P.warps, 0);
            ln.fb_hist_sig = sig;
        }
        ln.H_par ^= 1;
    }
    ln.slots.clear();
    ln.regions.clear();
    size_t ev_used = 0;
    auto next_event = [&](int *idx) -> int {
        if (ev_used == ln.ev_pool.size()) {
            hipEvent_t e;
            MI_HIP_TRY(hipEventCreate(&e));
            ln.ev_pool.push_back(e);
        }
        *idx = (int)ev_used++;
        return MI_OK;
    };

    const int sem = P.semantics;
    const bool legacy_warp = tuning().warp_legacy != 0;
    // level 0: convertTo(CV_32F, 8U ? 1 : 255)  tvl1flow.cpp:200-201
    rc = convert(ln.tab_dev, I0s[0].type, ln.L[0].I0, ln.L[0].I1, ln.L[0].g, st);
    if (rc) return rc;
    if (P.use_initial_flow) {
        // CPU behaviour (optflow/src/tvl1flow.cpp:435-439); the CUDA calc() never splits the
        // caller's flow (latent reference bug).
        rc = unpack_flow(ln.tab_dev, ln.L[0].u[0][0], ln.L[0].u[0][1], ln.L[0].g, st);
        if (rc) return rc;
    }
    const float one3[3] = {1.f, 1.f, 1.f};
    // create the scales (tvl1flow.cpp:238-266)
    for (int s = 1; s < nl_built; ++s) {
        const float *src[3][2] = {{ln.L[s - 1].I0, nullptr}, {ln.L[s - 1].I1, nullptr}, {nullptr, nullptr}};
        float *dst[3] = {ln.L[s].I0, ln.L[s].I1, nullptr};
        rc = resize(sem, 2, src, 1, dst, ln.L[s - 1].g, ln.L[s].g, P.scale_step, P.scale_step, one3, nullptr, 0, st);
        if (rc) return rc;
        if (s >= ns) break;
        if (P.use_initial_flow) {
            const float *us[3][2] = {{ln.L[s - 1].u[0][0], nullptr}, {ln.L[s - 1].u[0][1], nullptr}, {nullptr, nullptr}};
            float *ud[3] = {ln.L[s].u[0][0], ln.L[s].u[0][1], nullptr};
            const float sc = (float)P.scale_step;
            const float post[3] = {sc, sc, 1.f};
            rc = resize(sem, 2, us, 1, ud, ln.L[s - 1].g, ln.L[s].g, P.scale_step, P.scale_step, post, nullptr, 0, st);
            if (rc) return rc;
        }
    }
    if (!P.use_initial_flow) {
        const Geo &g = ln.L[ns - 1].g;
        MI_HIP_TRY(hipMemsetAsync(ln.L[ns - 1].u[0][0], 0, sizeof(float) * (size_t)g.ps * B, st));
        MI_HIP_TRY(hipMemsetAsync(ln.L[ns - 1].u[0][1], 0, sizeof(float) * (size_t)g.ps * B, st));
    }
    const bool gam = P.gamma != 0.0;
    if (gam) {   // u3 starts at 0 on the coarsest scale (tvl1flow.cpp:273-275; optflow tvl1flow.cpp:498-500)
        const Geo &g = ln.L[ns - 1].g;
        MI_HIP_TRY(hipMemsetAsync(ln.L[ns - 1].u[0][2], 0, sizeof(float) * (size_t)g.ps * B, st));
    }

    const float l_t = (float)(P.lambda * P.theta);
    const float taut = (float)(P.tau / P.theta);
    const float theta = (float)P.theta;

    int q = 0, q_last = -1;   // device-control slot counters
    int e_next = 0;           // next per-iteration error-sum index (speculative path)
    int q_settle_prev = -1, q_settle_scale = -1;   // settling launches of the previous warp / of the coarser scale's first warp
    // host feedback: only where the calc is one lane on the caller's stream (a wait inside lane k would hold up the enqueue of lane k+1)
    // ... and never while the stream is being captured into a graph: an event synchronise on a capturing stream fails and
    // invalidates the capture (hipGraph users get the fully stream-ordered enqueue, host_feedback = -1 behaviour)
    hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap_st) == hipSuccess && cap_st != hipStreamCaptureStatusNone;
    const bool fb = spec && h->last_lanes == 1 && P.host_feedback >= 0 && (P.host_feedback == 1 || B <= 2) && !capturing;
    const bool fb_poll = fb && tuning().fb_poll != 0;
    int fb_prev_warp = 2, fb_prev_scale = 2;   // launch index at which the previous warp / the coarser scale's first warp was found stopped
    ln.fb_waits = ln.fb_skipped = 0;
    if (fb) {
        if (ln.fb_cap < B) {
            if (ln.fb_host) (void)hipHostFree(ln.fb_host);
            if (ln.fb_flag) (void)hipHostFree(ln.fb_flag);
            ln.fb_host = nullptr; ln.fb_flag = nullptr; ln.fb_cap = 0;
            MI_HIP_TRY(hipHostMalloc((void **)&ln.fb_host, 2 * sizeof(int2) * (size_t)B, hipHostMallocDefault));
            MI_HIP_TRY(hipHostMalloc((void **)&ln.fb_flag, 2 * sizeof(int) * (size_t)B, hipHostMallocCoherent | hipHostMallocMapped));
            memset(ln.fb_flag, 0, 2 * sizeof(int) * (size_t)B);
            ln.fb_cap = B;
        }
        if (!ln.fb_ev) MI_HIP_TRY(hipEventCreateWithFlags(&ln.fb_ev, hipEventDisableTiming));
    }
    int cur = 0;              // host-known buffer set (fixed-work mode)
    Ctl ctl;
    memset(&ctl, 0, sizeof(ctl));
    ctl.S = ln.S; ctl.E = ln.E; ctl.Q = (int)ln.Q;
    ctl.P = ln.Pd; ctl.sched = (P.semantics == MI_SEM_CUDA_COMPAT) ? 1 : 0;   // cv::cuda's check schedule vs the CPU class's every-iteration check

    WarpZoom zoom;
    memset(&zoom, 0, sizeof(zoom));
    bool have_zoom = false;
    const bool zoom_path = !check && !P.exact_math && P.time_block != 1 && P.gamma == 0.0 && !legacy_warp && warp_zoom_ok() && tuning().x_skip == 0;
    for (int s = ns - 1; s >= 0; --s) {
        LevelBuf &Lv = ln.L[s];
        Geo g = Lv.g;
        // dense per-level re-layout of the full-resolution scratch planes
        float *I1wx = ln.scr[2], *I1wy = ln.scr[3], *grad = ln.scr[4], *rho = ln.scr[5];
        if (legacy_warp) {   // pair stride of the packed plane is g.ps float4 = 4*g.ps floats: same element index as the planes
            rc = gradient_pack(Lv.I1, ln.pack, g, st);
            if (rc) return rc;
        }
        const float *u1v[2] = {Lv.u[0][0], Lv.u[1][0]}, *u2v[2] = {Lv.u[0][1], Lv.u[1][1]};
        IterPlanes pl;
        pl.ix = I1wx; pl.iy = I1wy; pl.g = grad; pl.rc = rho;
        for (int k = 0; k < 2; ++k) { for (int j = 0; j < 3; ++j) pl.u[k][j] = Lv.u[k][j]; for (int j = 0; j < 6; ++j) pl.p[k][j] = ln.pbuf[k][j]; }
        pl.gamma = (float)P.gamma;
        pl.err_u3 = sem == MI_SEM_CPU_REF ? 1 : 0;   // optflow tvl1flow.cpp:1110 vs cuda tvl1flow.cu:276-283
        cur = 0;
        bool first_of_scale = true;
        // scaledEpsilon: float in the CPU class (optflow tvl1flow.cpp:1315), double in cv::cuda (:310)
        const double se = P.epsilon * P.epsilon * (double)(g.w * g.h);
        ctl.thr = sem == MI_SEM_CPU_REF ? (double)(float)se : se;

        bool pre_warped = false;   // this warp's kernel was enqueued ahead, behind the previous warp's last launch (host feedback)
        for (int wp = 0; wp < P.warps; ++wp) {
            Ctl wc = ctl;
            wc.q_prev = q_last;
            // at the first warp of a scale u lives in set 0 (host-known)
            const bool dev_cur = check && !first_of_scale;
            int w0 = -1, w1 = -1;
            if (h->profiling) {
                rc = next_event(&w0); if (rc) return rc;
                MI_HIP_TRY(hipEventRecord(ln.ev_pool[w0], st));
            }
            // Round 4 byte cuts of the fixed-work blocked path: (1) where every pass of this warp is the default T = 10 kernel, the
            // warp does not store |grad|^2 and the pass forms it from I1wx, I1wy (8 B/px per warp less through HBM, the same bits);
            // (2) the last pass of a scale does not store p (the next scale starts from p = 0; 16 B/px per scale).
            const bool blocked_w = !check && !P.exact_math && P.time_block != 1;
            std::vector<int> plan_w;
            bool nograd = false;
            if (blocked_w && !legacy_warp) {
                const int mfw = P.median_filtering > 1 ? P.median_filtering : 0;
                const int per = mfw ? P.inner_iterations : iters_per_warp;
                plan_w.resize(per + 1);
                plan_w.resize(tb_plan_level(g, per, P.time_block > 0 ? P.time_block : tb_max_block(), plan_w.data(), per, gam));
                nograd = !plan_w.empty();
                for (int v : plan_w) nograd = nograd && (gam || tb_nograd_ok(v, g));   // the illumination channel's kernels never read the plane
            }
            if (spec && !legacy_warp && P.median_filtering <= 1 && (gam || tb_spec_nograd_ok(g))) nograd = true;   // speculative steps: every block is a tbr launch
            float *grad_w = nograd ? nullptr : grad;
            pl.g = grad_w;
            // Round 5: a warp whose iterations are ONE pass of the default kernel runs inside that pass (producer waves, k_iterate_tbr
            // FW) -- no warp launch, no static planes through HBM
            const bool fast_w0 = !P.exact_math && (tuning().warp_fast > 0 || (tuning().warp_fast < 0 && sem == MI_SEM_CUDA_COMPAT));
            const bool fuse_w = blocked_w && !gam && !legacy_warp && nograd && plan_w.size() == 1 && P.median_filtering <= 1 && tuning().x_skip == 0 &&
                                tuning().warp_lds == 0 && !(wp == 0 && have_zoom) && tb_fused_ok(plan_w[0], g, sem, fast_w0);
            const bool warp_is_fused = !legacy_warp && tuning().x_skip == 0;
            const bool fast_w = !P.exact_math && (tuning().warp_fast > 0 || (tuning().warp_fast < 0 && sem == MI_SEM_CUDA_COMPAT));
            if (pre_warped) { rc = MI_OK; pre_warped = false; }
            else if (fuse_w) rc = MI_OK;
            else if (tuning().x_skip == 1 && wp > 0) rc = MI_OK;   // timing experiment: what a step costs without the warps' work and bytes
            else if (legacy_warp)
                rc = warp(sem, Lv.I0, ln.pack, u1v, u2v, nullptr, I1wx, I1wy, grad, rho, h->cubic_tab, g, dev_cur ? &wc : nullptr, cur, st);
            else
                rc = warp_fused(sem, !P.exact_math && (tuning().warp_fast > 0 || (tuning().warp_fast < 0 && sem == MI_SEM_CUDA_COMPAT)), -1, Lv.I0, Lv.I1, u1v, u2v, nullptr, I1wx, I1wy, grad_w, rho, h->cubic_tab, g, dev_cur ? &wc : nullptr, cur, st,
                                (wp == 0 && have_zoom) ? &zoom : nullptr);
            if (rc) return rc;
            if (w0 >= 0 && !fuse_w) {
                rc = next_event(&w1); if (rc) return rc;
                MI_HIP_TRY(hipEventRecord(ln.ev_pool[w1], st));
                ln.regions.push_back({w0, w1, 1, 44.0 * g.w * g.h * B, 1, s});   // 44 B/px per warp
            }
            int e0 = -1, e1 = -1;
            if (h->profiling && iters_per_warp > 0) {
                rc = next_event(&e0); if (rc) return rc;
                MI_HIP_TRY(hipEventRecord(ln.ev_pool[e0], st));
            }
            const bool blocked = !check && !P.exact_math && P.time_block != 1;
            const bool exact_blocked = !check && P.exact_math && P.time_block != 1 && !gam && P.median_filtering <= 1 && tuning().exact_tb != 0;
            long long nlaunch = 0;
            const int mf = P.median_filtering > 1 ? P.median_filtering : 0;
            float *const mu1[2] = {Lv.u[0][0], Lv.u[1][0]}, *const mu2[2] = {Lv.u[0][1], Lv.u[1][1]};
            if (blocked) {
                // T iterations per HBM pass (tvl1_tbr_kernels.hip), decomposition by measured cost; the optional median
                // filter sits between outer iterations, so blocks never span more than inner_iterations
                const int per = mf ? P.inner_iterations : iters_per_warp, nouter = mf ? P.iterations : 1;
                std::vector<int> plan(per + 1);
                const int nb = tb_plan_level(g, per, P.time_block > 0 ? P.time_block : tb_max_block(), plan.data(), per, gam);
                for (int no = 0; no < nouter; ++no) {
                    if (mf && (rc = median_flow(mf, mu1, mu2, ln.scr[0], ln.scr[1], g, nullptr, cur, st))) return rc;
                    for (int k = 0; k < nb; ++k) {
                        ++nlaunch;
                        if (tuning().x_skip == 2) { first_of_scale = false; continue; }   // timing experiment: the warps alone
                        const bool last_pass = tuning().tb_skip_p && wp == P.warps - 1 && no == nouter - 1 && k == nb - 1 && !mf;
                        if (fuse_w) rc = iterate_tb_fused(sem, Lv.I0, Lv.I1, h->cubic_tab, plan[k], pl, g, l_t, theta, taut, first_of_scale, cur, st, last_pass);
                        else rc = iterate_tb(plan[k], pl, g, l_t, theta, taut, first_of_scale, cur, 0, st, last_pass);
                        if (rc) return rc;
                        cur ^= 1;
                        first_of_scale = false;
                    }
                }
            } else if (exact_blocked) {
                // exact math, fixed work: blocks of up to 5 fused iterations (k_iterate_tbr MODE 2), bit-identical to the
                // one-iteration launches below
                for (int left = iters_per_warp; left > 0;) {
                    const int cap = P.time_block > 0 ? std::min(P.time_block, tb_exact_max_block()) : tb_exact_max_block();
                    const int T = std::min(left, cap);
                    ++nlaunch;
                    rc = iterate_tb_exact(T, pl, g, l_t, theta, taut, first_of_scale, cur, st);
                    if (rc) return rc;
                    cur ^= 1;
                    left -= T;
                    first_of_scale = false;
                }
            } else if (spec) {
                // Speculative steps (k_iterate_tbr MODE 1): a launch runs a block of iterations recording their error sums; the next
                // launch applies the reference's stopping rule to them and either builds on the block or replays the exact
                // count from its input.  One settling launch ends the warp.  After convergence the remaining launches end at once.
                const bool on_tiles = tile_eligible(g) && tuning().tile_spec != 0;
                std::vector<int> plan = spec_plan[on_tiles ? 3 : wp > 0 ? 2 : ((double)g.w * g.h * B >= kLargeLevel ? 0 : 1)];
                // the previous calc's count for this warp, where the host has seen it (polled host feedback): a first block of at most
                // 4 / 7 iterations runs on tiles of that margin (k_iterate_tile M), and the first poll goes where that many iterations
                // end.  A count that turns out different costs one more block or one more poll, as any estimate does.
                int hprev = 0;
                if (fb_poll && hist) {
                    int &hc = ln.fb_hist[(size_t)s * P.warps + wp];
                    hprev = hc;
                    hc = 0;   // known again once this warp's stop has been seen
                }
                if (on_tiles && hprev >= 1 && !plan.empty() && tuning().tile_fb_model != 0) {
                    // Block length = tile margin for the whole warp, from a cost model fitted to the traces of profiles/r10:
                    // a pass costs ~7 us (MIFLOW_TILE_FB_MODEL) of launch and hand-over plus (tile lanes) x (10.8 ps of loads and stores + 2.4 ps per
                    // iteration); tile lanes = pixels x 64 / (64 - 2M) x TR / (TR - 2M) with TR = 48 or 64 rows (tile_rows_for).
                    // Few iterations or a small level: one block of the margin that just holds them; many iterations on a
                    // level that fills the device: more, shorter blocks whose tiles own more of their pixels.
                    const double px = (double)g.w * g.h * B, TR = (double)tile_rows_for(g);
                    int best = 10;
                    double best_cost = 1e30;
                    for (int bl : {4, 7, 10}) {
                        const double passes = (double)((hprev + bl - 1) / bl);
                        const double lanes = px * 64.0 / (64.0 - 2.0 * bl) * TR / (TR - 2.0 * bl);
                        const double cost = passes * (double)tuning().tile_fb_model + lanes * (passes * 10.8e-6 + (double)hprev * 2.4e-6);
                        if (cost < best_cost) { best_cost = cost; best = bl; }
                    }
                    int total = 0;
                    for (int v : plan) total += v;
                    plan.assign((size_t)((total + best - 1) / best), best);
                } else if (on_tiles && hprev >= 1 && hprev <= 7 && !plan.empty()) plan[0] = hprev <= 4 ? 4 : 7;
                int t_after = 0;
                for (int v : plan) t_after += v;
                SpecK sk;
                memset(&sk, 0, sizeof(sk));
                sk.X = ln.X; sk.iters = iters_per_warp;
                // the first block's length: a fraction of what an earlier warp needed (measured on textured pairs: the second
                // warp of a scale needs about half of the first, later warps slightly fewer than their predecessor, the first
                // warp of a scale about 0.7 of the first warp one scale coarser)
                sk.q_hist = wp > 0 ? q_settle_prev : q_settle_scale;
                sk.hist_num = wp == 0 ? 7 : wp == 1 ? 9 : 4;
                sk.hist_den = wp == 0 ? 10 : wp == 1 ? 20 : 5;
                sk.slack = P.stop_slack;
                if (hist) {
                    const size_t o = ((size_t)s * P.warps + wp) * B;
                    sk.h_in = ln.H + (size_t)(ln.H_par ^ 1) * ln.H_cap + o;
                    sk.h_out = ln.H + (size_t)ln.H_par * ln.H_cap + o;
                }
                if (first_of_scale) {   // a replay of the scale's first block must see p = 0 in the input set as well
                    rc = zero_planes4(ln.pbuf[0], (size_t)g.ps * B, st);
                    if (rc) return rc;
                    if (gam) {
                        float *const p3[4] = {ln.pbuf[0][4], ln.pbuf[0][5], ln.pbuf[0][4], ln.pbuf[0][5]};
                        rc = zero_planes4(p3, (size_t)g.ps * B, st);
                        if (rc) return rc;
                    }
                }
                int e_prev = 0;
                // Host feedback (mi_tvl1_params.host_feedback): a call of one or two pairs reads the pairs' control slots back
                // between launches and stops enqueuing for this warp once every pair's slot says DONE -- the state is then settled
                // (a replay, if one was due, ran inside the launch that wrote the flag) and the slot is the one the following kernels
                // look at.  First read-back where the previous warp of this scale stopped (the first warp: where the coarser scale's
                // first warp did), then after every second launch.  The host waits like the reference's class does at each of its
                // checks (cudaoptflow/src/tvl1flow.cpp:362-368); the flows do not depend on any of it.
                int fb_next = fb ? std::max(1, wp > 0 ? fb_prev_warp : fb_prev_scale) : -1;
                if (fb && hprev > 0) {   // the launch behind the block in which iteration hprev falls
                    int kq = 0, sum = 0;
                    while (kq < (int)plan.size() && (sum += plan[kq]) < hprev) ++kq;
                    fb_next = std::max(1, std::min(kq + 1, (int)plan.size() - 1));
                }
                int fb_done_at = (int)plan.size();
                for (size_t k = 0; k <= plan.size(); ++k) {
                    const bool last = k == plan.size();
                    const int T = last ? plan.back() : plan[k];
                    if (!last) t_after -= T;
                    Ctl a = ctl;
                    a.q = q; a.q_prev = q_last; a.first_of_warp = (k == 0); a.reset_cur = (k == 0 && first_of_scale); a.n = 0;
                    sk.e0_prev = e_prev; sk.final_launch = last ? 1 : 0; sk.t_after = t_after;
                    int fb_seq_k = 0;
                    if (fb_poll) {
                        fb_seq_k = ln.fb_seq = (ln.fb_seq + 1) & 0x0fffffff;
                        sk.fb_flag = ln.fb_flag; sk.fb_seq = fb_seq_k;
                    }
                    MI_REQUIRE((long long)e_next + T <= (long long)ln.Q && q < (int)ln.Q, MI_ERR_BAD_ARG,
                               "speculative steps: error-sum slot %d + %d or launch slot %d beyond the %lld slots sized for this calc", e_next, T, q, (long long)ln.Q);
                    rc = iterate_tb_spec(T, pl, g, l_t, theta, taut, false, a, sk, e_next, st);
                    if (rc) return rc;
                    ln.slots.push_back({s, wp});
                    q_last = q++;
                    ++nlaunch;
                    e_prev = e_next;
                    if (!last) e_next += T;
                    if (fb && !last && (int)k == fb_next) {
                        bool all = true, all_before = true;
                        int n_most = 0;
                        // The next warp's kernel goes in BEHIND this launch before the host knows whether the warp has stopped: it
                        // runs only for pairs whose slot says so (Ctl::need_done) and is enqueued again, unconditionally, where one
                        // did not.  With the estimate right the device never waits for the host between two warps of a scale.
                        bool ahead = false;
                        if (fb_poll && tuning().fb_ahead != 0 && warp_is_fused && !h->profiling && wp + 1 < P.warps) {
                            Ctl nc = ctl;
                            nc.q_prev = q_last; nc.need_done = 1;
                            rc = warp_fused(sem, fast_w, -1, Lv.I0, Lv.I1, u1v, u2v, nullptr, I1wx, I1wy, grad_w, rho, h->cubic_tab, g, &nc, cur, st, nullptr);
                            if (rc) return rc;
                            ahead = true;
                        }
                        if (fb_poll) {
                            // the launch just enqueued publishes its decision when it starts: wait for that word, not for the launch
                            ++ln.fb_waits;
                            for (int b = 0; b < B; ++b) {
                                int v = 0;
                                // Bare spin (pause) for the first ~20 us -- the flag normally lands within a few microseconds of the launch
                                // starting -- then the core is handed back between looks (sched_yield; a sleep of 50 us once 2 ms have
                                // passed: earlier work queued on the caller's stream, or many handles waiting in many threads), so a
                                // waiting calc() does not hold a core.  The stream is queried on the slow path only; a wall-clock bound ends
                                // a wait no launch will ever answer.
                                const auto t_wait0 = std::chrono::steady_clock::now();
                                for (long long spin = 0;; ++spin) {
                                    v = __atomic_load_n(ln.fb_flag + 2 * b, __ATOMIC_ACQUIRE);
                                    if ((v >> 2) == fb_seq_k) break;
                                    if ((spin & 63) != 63) {
#if defined(__x86_64__) || defined(__i386__)
                                        __builtin_ia32_pause();
#endif
                                        continue;
                                    }
                                    const double waited_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_wait0).count();
                                    if (waited_us < 20.0) continue;
                                    // a failed launch never writes: the stream is idle (or in error) and the word is not there
                                    const hipError_t qe = hipStreamQuery(st);
                                    if (qe != hipErrorNotReady) {
                                        v = __atomic_load_n(ln.fb_flag + 2 * b, __ATOMIC_ACQUIRE);
                                        if ((v >> 2) == fb_seq_k) break;
                                        MI_HIP_TRY(qe);
                                        MI_REQUIRE(false, MI_ERR_HIP, "host feedback: launch finished without publishing its decision");
                                    }
                                    MI_REQUIRE(waited_us < kFeedbackWaitLimitUs, MI_ERR_HIP, "host feedback: no decision from the device within %.0f s", kFeedbackWaitLimitUs * 1e-6);
                                    if (waited_us < 2000.0) sched_yield();
                                    else { struct timespec ts_ = {0, 50000}; nanosleep(&ts_, nullptr); }
                                }
                                all_before = all_before && (v & 2);
                                all = all && (v & 1);
                                n_most = std::max(n_most, ln.fb_flag[2 * b + 1]);
                            }
                        } else {
                            // the slots of this launch and of the one before it (k >= 1), per pair
                            MI_HIP_TRY(hipMemcpy2DAsync(ln.fb_host, 2 * sizeof(int2), ln.S + (q_last - 1), sizeof(int2) * (size_t)ln.Q, 2 * sizeof(int2),
                                                        (size_t)B, hipMemcpyDeviceToHost, st));
                            MI_HIP_TRY(hipEventRecord(ln.fb_ev, st));
                            MI_HIP_TRY(hipEventSynchronize(ln.fb_ev));
                            ++ln.fb_waits;
                            for (int b = 0; b < B; ++b) {
                                all_before = all_before && (ln.fb_host[2 * b].y & MI_SLOT_DONE);
                                all = all && (ln.fb_host[2 * b + 1].y & MI_SLOT_DONE);
                            }
                        }
                        if (all) {
                            pre_warped = ahead;
                            if (fb_poll && hist) ln.fb_hist[(size_t)s * P.warps + wp] = n_most;
                            ln.fb_skipped += (long long)plan.size() - (long long)k;
                            fb_done_at = all_before ? (int)k - 1 : (int)k;   // where the next warp's first read-back goes
                            break;
                        }
                        fb_next = (int)k + 2;
                    }
                }
                if (fb) {
                    fb_prev_warp = fb_done_at;
                    if (wp == 0) fb_prev_scale = fb_done_at;
                }
                q_settle_prev = q_last;
                if (wp == 0) q_settle_scale = q_last;
                first_of_scale = false;
            } else for (int it = 0; it < iters_per_warp; ++it) {
                ++nlaunch;
                if (mf && it % P.inner_iterations == 0) {   // cv::medianBlur before each outer iteration (optflow tvl1flow.cpp:1381-1384)
                    Ctl mc = ctl;
                    mc.q_prev = q_last; mc.first_of_warp = (it == 0); mc.reset_cur = first_of_scale;
                    if ((rc = median_flow(mf, mu1, mu2, ln.scr[0], ln.scr[1], g, check ? &mc : nullptr, cur, st))) return rc;
                }
                if (check) {
                    Ctl ic = ctl;
                    ic.q = q; ic.q_prev = q_last;
                    ic.first_of_warp = (it == 0);
                    ic.reset_cur = first_of_scale;
                    ic.n = it;
                    rc = iterate(P.exact_math != 0, pl, g, l_t, theta, taut, first_of_scale, &ic, 0, st);
                    ln.slots.push_back({s, wp});
                    q_last = q++;
                } else {
                    rc = iterate(P.exact_math != 0, pl, g, l_t, theta, taut, first_of_scale, nullptr, cur, st);
                    cur ^= 1;
                }
                if (rc) return rc;
                first_of_scale = false;
            }
            if (e0 >= 0) {
                rc = next_event(&e1); if (rc) return rc;
                MI_HIP_TRY(hipEventRecord(ln.ev_pool[e1], st));
                ln.regions.push_back({e0, e1, nlaunch, 64.0 * g.w * g.h * B * iters_per_warp, 0, s});
            }
        }
        Ctl ec = ctl;
        ec.q_prev = q_last;
        const bool dev_cur = check && !first_of_scale;
        if (s == 0) {
            rc = pack_flow(ln.tab_dev, u1v, u2v, g, dev_cur ? &ec : nullptr, cur, st);
            if (rc) return rc;
            break;
        }
        // zoom the flow to the next finer scale and rescale it (tvl1flow.cpp:291-300)
        const Geo &gf = ln.L[s - 1].g;
        have_zoom = false;
        if (zoom_path && !dev_cur) {
            // fixed-work blocked path: the finer scale's FIRST WARP samples this scale's flow itself (k_warp6 UP) -- no resize launch,
            // no 8 B/px round trip of the zoomed flow
            zoom.u1c = Lv.u[cur][0]; zoom.u2c = Lv.u[cur][1];
            zoom.u1o = ln.L[s - 1].u[0][0]; zoom.u2o = ln.L[s - 1].u[0][1];
            zoom.gc = g;
            zoom.inv_scale_x = (double)gf.w / g.w; zoom.inv_scale_y = (double)gf.h / g.h;
            zoom.post = (float)(1.0 / P.scale_step);
            have_zoom = true;
            continue;
        }
        // u3 is zoomed too but NOT rescaled (tvl1flow.cpp:293-300; optflow tvl1flow.cpp:524-528)
        const float *us[3][2] = {{Lv.u[0][0], Lv.u[1][0]}, {Lv.u[0][1], Lv.u[1][1]}, {gam ? Lv.u[0][2] : nullptr, gam ? Lv.u[1][2] : nullptr}};
        float *ud[3] = {ln.L[s - 1].u[0][0], ln.L[s - 1].u[0][1], gam ? ln.L[s - 1].u[0][2] : nullptr};
        const float inv = (float)(1.0 / P.scale_step);
        const float post[3] = {inv, inv, 1.f};
        rc = resize(sem, gam ? 3 : 2, us, 2, ud, g, gf, (double)gf.w / g.w, (double)gf.h / g.h, post,
                    dev_cur ? &ec : nullptr, cur, st);
        if (rc) return rc;
    }
    return MI_OK;
}

int mi_tvl1_calc_batch(mi_tvl1 *h, int n, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows, void *stream_)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    MI_REQUIRE(n > 0 && I0s && I1s && flows, MI_ERR_BAD_ARG, "empty batch");
    hipStream_t st = (hipStream_t)stream_;
    for (int i = 0; i < n; ++i) {
        int rc = check_pair(&I0s[i], &I1s[i], &flows[i], &I0s[0]);
        if (rc) return rc;
    }
    // Lanes: a batch of >= 4 pairs is split into contiguous sub-batches that run concurrently, forked from and joined to the
    // caller's stream with events (stream-ordered for the caller exactly like the single-stream form).  The pairs are
    // independent, so the result is bit-identical to running them in one lane.
    int lanes = h->P.lanes > 0 ? h->P.lanes : (tuning().lanes > 0 ? tuning().lanes : (n >= 4 ? 2 : 1));
    if (lanes > mi_tvl1::kMaxLanes) lanes = mi_tvl1::kMaxLanes;
    if (lanes > n) lanes = n;
    int ns = 0;
    h->last_check = h->P.epsilon > 0.0 && h->P.iterations * h->P.inner_iterations > 0;
    h->last_batch = n; h->last_lanes = lanes;
    for (int i = 0; i <= lanes; ++i) h->last_first[i] = (int)((long long)n * i / lanes);   // contiguous, sizes differ by at most one
    if (lanes == 1) {
        int rc = lane_calc(h, h->lane[0], n, I0s, I1s, flows, st, &ns);
        if (rc) return rc;
    } else {
        // Lane 0 runs on the caller's own stream, every further lane on ONE internal stream each: a handle adds lanes - 1 streams
        // to the process.  (HIP multiplexes streams onto a few hardware queues -- 4 by default, GPU_MAX_HW_QUEUES; two lanes that
        // land on one queue run back to back: measured 400 instead of 520 pairs/s when enough streams of other handles were alive.)
        // every stream and event the fork / join needs exists BEFORE anything is enqueued, so no failure between the fork and the
        // join can leave an internal lane un-joined
        if (!h->fork) MI_HIP_TRY(hipEventCreateWithFlags(&h->fork, hipEventDisableTiming));
        for (int li = 1; li < lanes; ++li) {
            Lane &ln = h->lane[li];
            if (!ln.stream) MI_HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
            if (!ln.done) MI_HIP_TRY(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
        }
        MI_HIP_TRY(hipEventRecord(h->fork, st));
        int rc_first = MI_OK;
        auto note = [&](hipError_t e) {
            if (e != hipSuccess && !rc_first) { set_error("HIP error in the lane fork / join: %s", hipGetErrorString(e)); rc_first = MI_ERR_HIP; }
            return e == hipSuccess;
        };
        for (int li = lanes - 1; li >= 0; --li) {   // the internal lanes first: they start while lane 0 is still being enqueued
            Lane &ln = h->lane[li];
            const int off = h->last_first[li], cnt = h->last_first[li + 1] - off;
            hipStream_t ls = li > 0 ? ln.stream : st;
            // a lane that could not be forked behind the caller's stream must not run at all (it would read inputs too early)
            const bool forked = li == 0 || note(hipStreamWaitEvent(ln.stream, h->fork, 0));
            if (forked && !rc_first) {
                const int rc = lane_calc(h, ln, cnt, I0s + off, I1s + off, flows + off, ls, &ns);
                if (rc && !rc_first) rc_first = rc;
            }
        }
        // always join, also after an error: the caller's stream must not run ahead of work already enqueued.  If the join itself
        // fails, fall back to blocking the host until the lane has drained.
        for (int li = 1; li < lanes; ++li) {
            Lane &ln = h->lane[li];
            if (!(note(hipEventRecord(ln.done, ln.stream)) && note(hipStreamWaitEvent(st, ln.done, 0)))) (void)hipStreamSynchronize(ln.stream);
        }
        if (rc_first) return rc_first;
    }
    h->last_nscales = ns;
    // the reference shrinks nscales_ for good when a level falls below 16 px (tvl1flow.cpp:243-247: `nscales_ = s; break;`)
    if (ns < h->P.nscales) h->P.nscales = ns;
    return MI_OK;
}

int mi_tvl1_calc(mi_tvl1 *h, const mi_mat *I0, const mi_mat *I1, mi_mat *flow, void *stream)
{
    return mi_tvl1_calc_batch(h, 1, I0, I1, flow, stream);
}

int mi_tvl1_last_iterations(mi_tvl1 *h, int pair, int *nscales_used, int *iters, int cap, void *stream)
{
    MI_REQUIRE(h && iters && nscales_used, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(pair >= 0 && pair < h->last_batch, MI_ERR_BAD_ARG, "pair index out of range");
    const int ns = h->last_nscales, nw = h->P.warps;
    MI_REQUIRE(cap >= ns * nw, MI_ERR_BAD_ARG, "iters capacity too small");
    *nscales_used = ns;
    for (int i = 0; i < ns * nw; ++i) iters[i] = 0;
    if (!h->last_check) {
        for (int i = 0; i < ns * nw; ++i) iters[i] = h->P.iterations * h->P.inner_iterations;
        return MI_OK;
    }
    int li = 0;
    while (li + 1 < h->last_lanes && pair >= h->last_first[li + 1]) ++li;
    Lane &ln = h->lane[li];
    const int lp = pair - h->last_first[li];
    const int nq = (int)ln.slots.size();
    std::vector<int2> S(nq);
    MI_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    MI_HIP_TRY(hipMemcpy(S.data(), ln.S + (size_t)lp * ln.Q, sizeof(int2) * nq, hipMemcpyDeviceToHost));
    for (int i = 0; i < nq; ++i) {
        // one-iteration launches: bit 0 = the iteration was executed; speculative launches: bits 8..15 = iterations kept
        const int k = (S[i].y >> 8) & 0xff;
        iters[ln.slots[i].scale * nw + ln.slots[i].warp] += k ? k : (S[i].y & 1);
    }
    return MI_OK;
}

int miflow_selftest_tvl1_slots(mi_tvl1 *h, int pair, int *out_host, int cap_launches, void *stream)
{
    MI_REQUIRE(h && out_host, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(pair >= 0 && pair < h->last_batch, MI_ERR_BAD_ARG, "pair out of range");
    if (!h->last_check) return 0;
    int li = 0;
    while (li + 1 < h->last_lanes && pair >= h->last_first[li + 1]) ++li;
    Lane &ln = h->lane[li];
    const int lp = pair - h->last_first[li];
    const int nq = (int)ln.slots.size();
    MI_REQUIRE(nq <= cap_launches, MI_ERR_BAD_ARG, "capacity too small");
    std::vector<int2> S(nq);
    std::vector<int4> X(nq);
    MI_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    MI_HIP_TRY(hipMemcpy(S.data(), ln.S + (size_t)lp * ln.Q, sizeof(int2) * nq, hipMemcpyDeviceToHost));
    if (ln.X) MI_HIP_TRY(hipMemcpy(X.data(), ln.X + (size_t)lp * ln.Q, sizeof(int4) * nq, hipMemcpyDeviceToHost));
    for (int i = 0; i < nq; ++i) {
        int *o = out_host + 8 * i;
        o[0] = ln.slots[i].scale; o[1] = ln.slots[i].warp; o[2] = S[i].x; o[3] = S[i].y;
        o[4] = X[i].x; o[5] = X[i].y; o[6] = X[i].z; o[7] = X[i].w;
    }
    return nq;
}
