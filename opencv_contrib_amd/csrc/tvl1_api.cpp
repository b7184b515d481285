// Dual TV-L1: handle, scratch arena, pyramid orchestration and the C-ABI entry points.
// Host-side twin of OpticalFlowDual_TVL1_Impl::calcImpl / procOneScale
// (modules/cudaoptflow/src/tvl1flow.cpp:185-382; CPU modules/optflow/src/tvl1flow.cpp:402-533,
// 1313-1408) -- but fully stream-ordered: no host read-back inside the iteration loop.
#include "tvl1_dev.h"
#include <cfloat>
#include <cmath>
#include <vector>

using namespace mi;
using namespace mi::tvl1;

namespace {

struct LevelBuf {
    Geo g;
    float *I0 = nullptr, *I1 = nullptr;
    float *u[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};  // [set][u1,u2,u3]
};

struct SlotInfo { int scale, warp; };

}  // namespace

struct mi_tvl1 {
    mi_tvl1_params P;
    int device = 0;
    // capacity the arena was built for
    int capW = 0, capH = 0, capB = 0, capScales = 0;
    double capStep = 0;
    float *arena = nullptr;
    size_t arena_floats = 0;
    std::vector<LevelBuf> L;
    // full-resolution-capacity scratch planes (re-laid-out densely per level)
    float *scr[6] = {};    // scr[0..1] unused (kept for layout), I1wx, I1wy, grad, rho_c
    float *pack = nullptr; // float4 {I1, I1x, I1y, 0} per pixel of the current level
    float *pbuf[2][6] = {};   // [set][p11,p12,p21,p22,p31,p32]
    bool capGamma = false, capMedian = false;
    float *cubic_tab = nullptr;
    PtrTab *tab_dev = nullptr;
    int tab_cap = 0;
    std::vector<PtrTab> tab_host;   // source of the asynchronous upload: must outlive the call
    // device loop control
    int2 *S = nullptr;
    unsigned long long *E = nullptr;
    double *Pd = nullptr;   // per slot prevError (cv::cuda check schedule)
    int Q = 0, ctlB = 0;
    std::vector<SlotInfo> slots;
    int last_nscales = 0, last_batch = 0;
    bool last_check = false;
    // profiling (mi_tvl1_set_profiling)
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    struct Region { int e0, e1; long long launches; double bytes; };
    std::vector<Region> regions;
};

static int round_half_even(double v) { return (int)std::lrint(v); }

void mi_tvl1_default_params(mi_tvl1_params *p)
{
    if (!p) return;
    // cv::cuda::OpticalFlowDual_TVL1::create defaults, cudaoptflow.hpp:375-385
    p->tau = 0.25; p->lambda = 0.15; p->theta = 0.3; p->epsilon = 0.01; p->scale_step = 0.8; p->gamma = 0.0;
    p->nscales = 5; p->warps = 5; p->iterations = 300; p->use_initial_flow = 0;
    p->inner_iterations = 1; p->median_filtering = 1;
    p->semantics = MI_SEM_CPU_REF; p->exact_math = 1; p->time_block = 0;
}

static int validate_params(const mi_tvl1_params *p)
{
    MI_REQUIRE(p, MI_ERR_BAD_ARG, "null params");
    MI_REQUIRE(p->nscales > 0 && p->nscales <= 32, MI_ERR_BAD_ARG, "nscales must be in [1,32] (CV_Assert nscales_ > 0)");
    MI_REQUIRE(p->warps >= 0 && p->warps <= 64, MI_ERR_BAD_ARG, "warps must be in [0,64]");
    MI_REQUIRE(p->iterations >= 0 && p->inner_iterations >= 0, MI_ERR_BAD_ARG, "negative iteration count");
    MI_REQUIRE(p->scale_step > 0 && p->scale_step < 1, MI_ERR_BAD_ARG, "scale_step must be in (0,1)");
    MI_REQUIRE(p->theta != 0, MI_ERR_BAD_ARG, "theta must be non-zero");
    MI_REQUIRE(p->semantics == MI_SEM_CPU_REF || p->semantics == MI_SEM_CUDA_COMPAT, MI_ERR_BAD_ARG, "bad semantics");
    MI_REQUIRE(p->median_filtering <= 1 || p->median_filtering == 3 || p->median_filtering == 5, MI_ERR_BAD_ARG,
               "medianFiltering must be 1 (off), 3 or 5 (cv::medianBlur on CV_32F)");
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

static void free_arena(mi_tvl1 *h)
{
    if (h->arena) (void)hipFree(h->arena);
    h->arena = nullptr;
    h->L.clear();
}

int mi_tvl1_set_profiling(mi_tvl1 *h, int enable)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    h->profiling = enable != 0;
    return MI_OK;
}

int mi_tvl1_get_profile(mi_tvl1 *h, double *ms_total, long long *launches, double *algo_bytes)
{
    MI_REQUIRE(h && ms_total && launches && algo_bytes, MI_ERR_BAD_ARG, "null argument");
    *ms_total = 0; *launches = 0; *algo_bytes = 0;
    for (const auto &r : h->regions) {
        MI_HIP_TRY(hipEventSynchronize(h->ev_pool[r.e1]));
        float ms = 0.f;
        MI_HIP_TRY(hipEventElapsedTime(&ms, h->ev_pool[r.e0], h->ev_pool[r.e1]));
        *ms_total += ms; *launches += r.launches; *algo_bytes += r.bytes;
    }
    return MI_OK;
}

void mi_tvl1_destroy(mi_tvl1 *h)
{
    if (!h) return;
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    free_arena(h);
    if (h->cubic_tab) (void)hipFree(h->cubic_tab);
    if (h->tab_dev) (void)hipFree(h->tab_dev);
    if (h->S) (void)hipFree(h->S);
    if (h->E) (void)hipFree(h->E);
    if (h->Pd) (void)hipFree(h->Pd);
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

static int ensure_arena(mi_tvl1 *h, int W, int H, int B)
{
    const bool gam = h->P.gamma != 0.0;
    if (h->arena && h->capW == W && h->capH == H && h->capB >= B && h->capScales == h->P.nscales &&
        h->capStep == h->P.scale_step && h->capGamma == gam && h->capMedian == (h->P.median_filtering > 1))
        return MI_OK;
    const bool med = h->P.median_filtering > 1;
    free_arena(h);
    std::vector<Geo> geo;
    const int nl = plan_levels(h->P, W, H, B, geo);
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
    const size_t offPack = take(nfull * 4);
    for (int k = 0; k < 12; ++k) offP[k] = (k % 6 >= 4 && !gam) ? 0 : take(nfull);
    MI_HIP_TRY(hipMalloc((void **)&h->arena, total * sizeof(float)));
    h->arena_floats = total;
    h->L.resize(nl);
    for (int l = 0; l < nl; ++l) {
        h->L[l].g = geo[l];
        h->L[l].I0 = h->arena + offI0[l];
        h->L[l].I1 = h->arena + offI1[l];
        for (int k = 0; k < 6; ++k) h->L[l].u[k / 3][k % 3] = (k % 3 == 2 && !gam) ? nullptr : h->arena + offU[l * 6 + k];
    }
    for (int k = 0; k < 6; ++k) h->scr[k] = h->arena + offScr[k];
    h->pack = h->arena + offPack;
    for (int k = 0; k < 12; ++k) h->pbuf[k / 6][k % 6] = (k % 6 >= 4 && !gam) ? nullptr : h->arena + offP[k];
    h->capGamma = gam; h->capMedian = med;
    h->capW = W; h->capH = H; h->capB = B; h->capScales = h->P.nscales; h->capStep = h->P.scale_step;
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

int mi_tvl1_calc_batch(mi_tvl1 *h, int n, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows, void *stream_)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    MI_REQUIRE(n > 0 && I0s && I1s && flows, MI_ERR_BAD_ARG, "empty batch");
    hipStream_t st = (hipStream_t)stream_;
    const mi_tvl1_params &P = h->P;
    for (int i = 0; i < n; ++i) {
        int rc = check_pair(&I0s[i], &I1s[i], &flows[i], &I0s[0]);
        if (rc) return rc;
    }
    const int W = I0s[0].cols, H = I0s[0].rows, B = n;
    int rc = ensure_arena(h, W, H, B);
    if (rc) return rc;
    const int nl_built = (int)h->L.size();
    // number of usable scales (tvl1flow.cpp:243-247)
    int ns = nl_built;
    if (ns > 1 && (h->L[ns - 1].g.w < 16 || h->L[ns - 1].g.h < 16)) ns -= 1;
    for (int l = 0; l < nl_built; ++l) h->L[l].g.batch = B;

    // external pointer table
    if (h->tab_cap < B) {
        if (h->tab_dev) (void)hipFree(h->tab_dev);
        MI_HIP_TRY(hipMalloc((void **)&h->tab_dev, sizeof(PtrTab) * B));
        h->tab_cap = B;
    }
    {
        std::vector<PtrTab> &tab = h->tab_host;
        tab.resize(B);
        for (int i = 0; i < B; ++i) {
            tab[i].a = I0s[i].data; tab[i].b = I1s[i].data; tab[i].out = flows[i].data;
            tab[i].step_a = (long long)I0s[i].step; tab[i].step_b = (long long)I1s[i].step;
            tab[i].step_out = (long long)flows[i].step;
        }
        // pageable source: a few KB, staged by the runtime before the call returns; kept in the handle anyway
        MI_HIP_TRY(hipMemcpyAsync(h->tab_dev, tab.data(), sizeof(PtrTab) * B, hipMemcpyHostToDevice, st));
    }

    const int iters_per_warp = P.iterations * P.inner_iterations;
    const bool check = P.epsilon > 0.0 && iters_per_warp > 0;
    const int Q = ns * P.warps * iters_per_warp;
    if (check) {
        if (h->Q < Q || h->ctlB < B) {
            if (h->S) (void)hipFree(h->S);
            if (h->E) (void)hipFree(h->E);
            if (h->Pd) (void)hipFree(h->Pd);
            h->S = nullptr; h->E = nullptr; h->Pd = nullptr;
            MI_HIP_TRY(hipMalloc((void **)&h->S, sizeof(int2) * (size_t)Q * B));
            MI_HIP_TRY(hipMalloc((void **)&h->E, sizeof(unsigned long long) * (size_t)Q * B));
            MI_HIP_TRY(hipMalloc((void **)&h->Pd, sizeof(double) * (size_t)Q * B));
            h->Q = Q; h->ctlB = B;
        }
        MI_HIP_TRY(hipMemsetAsync(h->E, 0, sizeof(unsigned long long) * (size_t)h->Q * B, st));
        MI_HIP_TRY(hipMemsetAsync(h->S, 0, sizeof(int2) * (size_t)h->Q * B, st));
    }
    h->slots.clear();
    h->regions.clear();
    size_t ev_used = 0;
    auto next_event = [&](int *idx) -> int {
        if (ev_used == h->ev_pool.size()) {
            hipEvent_t e;
            MI_HIP_TRY(hipEventCreate(&e));
            h->ev_pool.push_back(e);
        }
        *idx = (int)ev_used++;
        return MI_OK;
    };
    h->last_nscales = ns; h->last_batch = B; h->last_check = check;

    const int sem = P.semantics;
    // level 0: convertTo(CV_32F, 8U ? 1 : 255)  tvl1flow.cpp:200-201
    rc = convert(h->tab_dev, I0s[0].type, h->L[0].I0, h->L[0].I1, h->L[0].g, st);
    if (rc) return rc;
    if (P.use_initial_flow) {
        // CPU behaviour (optflow/src/tvl1flow.cpp:435-439); the CUDA calc() never splits the
        // caller's flow (latent reference bug).
        rc = unpack_flow(h->tab_dev, h->L[0].u[0][0], h->L[0].u[0][1], h->L[0].g, st);
        if (rc) return rc;
    }
    const float one3[3] = {1.f, 1.f, 1.f};
    // create the scales (tvl1flow.cpp:238-266)
    for (int s = 1; s < nl_built; ++s) {
        const float *src[3][2] = {{h->L[s - 1].I0, nullptr}, {h->L[s - 1].I1, nullptr}, {nullptr, nullptr}};
        float *dst[3] = {h->L[s].I0, h->L[s].I1, nullptr};
        rc = resize(sem, 2, src, 1, dst, h->L[s - 1].g, h->L[s].g, P.scale_step, P.scale_step, one3, nullptr, 0, st);
        if (rc) return rc;
        if (s >= ns) break;
        if (P.use_initial_flow) {
            const float *us[3][2] = {{h->L[s - 1].u[0][0], nullptr}, {h->L[s - 1].u[0][1], nullptr}, {nullptr, nullptr}};
            float *ud[3] = {h->L[s].u[0][0], h->L[s].u[0][1], nullptr};
            const float sc = (float)P.scale_step;
            const float post[3] = {sc, sc, 1.f};
            rc = resize(sem, 2, us, 1, ud, h->L[s - 1].g, h->L[s].g, P.scale_step, P.scale_step, post, nullptr, 0, st);
            if (rc) return rc;
        }
    }
    if (!P.use_initial_flow) {
        const Geo &g = h->L[ns - 1].g;
        MI_HIP_TRY(hipMemsetAsync(h->L[ns - 1].u[0][0], 0, sizeof(float) * (size_t)g.ps * B, st));
        MI_HIP_TRY(hipMemsetAsync(h->L[ns - 1].u[0][1], 0, sizeof(float) * (size_t)g.ps * B, st));
    }
    const bool gam = P.gamma != 0.0;
    if (gam) {   // u3 starts at 0 on the coarsest scale (tvl1flow.cpp:273-275; optflow tvl1flow.cpp:498-500)
        const Geo &g = h->L[ns - 1].g;
        MI_HIP_TRY(hipMemsetAsync(h->L[ns - 1].u[0][2], 0, sizeof(float) * (size_t)g.ps * B, st));
    }

    const float l_t = (float)(P.lambda * P.theta);
    const float taut = (float)(P.tau / P.theta);
    const float theta = (float)P.theta;

    int q = 0, q_last = -1;   // device-control slot counters
    int cur = 0;              // host-known buffer set (fixed-work mode)
    Ctl ctl;
    memset(&ctl, 0, sizeof(ctl));
    ctl.S = h->S; ctl.E = h->E; ctl.Q = h->Q;
    ctl.P = h->Pd; ctl.sched = (P.semantics == MI_SEM_CUDA_COMPAT) ? 1 : 0;   // cv::cuda's check schedule vs the CPU class's every-iteration check

    for (int s = ns - 1; s >= 0; --s) {
        LevelBuf &Lv = h->L[s];
        Geo g = Lv.g;
        // dense per-level re-layout of the full-resolution scratch planes
        float *I1wx = h->scr[2], *I1wy = h->scr[3], *grad = h->scr[4], *rho = h->scr[5];
        // pair stride of the packed plane is g.ps float4 = 4*g.ps floats: same element index as the planes
        rc = gradient_pack(Lv.I1, h->pack, g, st);
        if (rc) return rc;
        const float *u1v[2] = {Lv.u[0][0], Lv.u[1][0]}, *u2v[2] = {Lv.u[0][1], Lv.u[1][1]};
        IterPlanes pl;
        pl.ix = I1wx; pl.iy = I1wy; pl.g = grad; pl.rc = rho;
        for (int k = 0; k < 2; ++k) { for (int j = 0; j < 3; ++j) pl.u[k][j] = Lv.u[k][j]; for (int j = 0; j < 6; ++j) pl.p[k][j] = h->pbuf[k][j]; }
        pl.gamma = (float)P.gamma;
        pl.err_u3 = sem == MI_SEM_CPU_REF ? 1 : 0;   // optflow tvl1flow.cpp:1110 vs cuda tvl1flow.cu:276-283
        cur = 0;
        bool first_of_scale = true;
        // scaledEpsilon: float in the CPU class (optflow tvl1flow.cpp:1315), double in cv::cuda (:310)
        const double se = P.epsilon * P.epsilon * (double)(g.w * g.h);
        ctl.thr = sem == MI_SEM_CPU_REF ? (double)(float)se : se;

        for (int wp = 0; wp < P.warps; ++wp) {
            Ctl wc = ctl;
            wc.q_prev = q_last;
            // at the first warp of a scale u lives in set 0 (host-known)
            const bool dev_cur = check && !first_of_scale;
            rc = warp(sem, Lv.I0, h->pack, u1v, u2v, nullptr, I1wx, I1wy, grad, rho, h->cubic_tab, g,
                      dev_cur ? &wc : nullptr, cur, st);
            if (rc) return rc;
            int e0 = -1, e1 = -1;
            if (h->profiling && iters_per_warp > 0) {
                rc = next_event(&e0); if (rc) return rc;
                MI_HIP_TRY(hipEventRecord(h->ev_pool[e0], st));
            }
            const bool blocked = !check && !P.exact_math && P.time_block != 1 && !gam;
            long long nlaunch = 0;
            const int mf = P.median_filtering > 1 ? P.median_filtering : 0;
            float *const mu1[2] = {Lv.u[0][0], Lv.u[1][0]}, *const mu2[2] = {Lv.u[0][1], Lv.u[1][1]};
            if (blocked) {
                // T iterations per HBM pass (tvl1_tb_kernels.hip), decomposition by measured cost; the optional median
                // filter sits between outer iterations, so blocks never span more than inner_iterations
                const int per = mf ? P.inner_iterations : iters_per_warp, nouter = mf ? P.iterations : 1;
                std::vector<int> plan(per + 1);
                const int nb = tb_plan(per, P.time_block > 0 ? P.time_block : tb_max_block(), plan.data(), per);
                for (int no = 0; no < nouter; ++no) {
                    if (mf && (rc = median_flow(mf, mu1, mu2, h->scr[0], h->scr[1], g, nullptr, cur, st))) return rc;
                    for (int k = 0; k < nb; ++k) {
                        ++nlaunch;
                        rc = iterate_tb(plan[k], pl, g, l_t, theta, taut, first_of_scale, cur, 0, st);
                        if (rc) return rc;
                        cur ^= 1;
                        first_of_scale = false;
                    }
                }
            } else for (int it = 0; it < iters_per_warp; ++it) {
                ++nlaunch;
                if (mf && it % P.inner_iterations == 0) {   // cv::medianBlur before each outer iteration (optflow tvl1flow.cpp:1381-1384)
                    Ctl mc = ctl;
                    mc.q_prev = q_last; mc.first_of_warp = (it == 0); mc.reset_cur = first_of_scale;
                    if ((rc = median_flow(mf, mu1, mu2, h->scr[0], h->scr[1], g, check ? &mc : nullptr, cur, st))) return rc;
                }
                if (check) {
                    Ctl ic = ctl;
                    ic.q = q; ic.q_prev = q_last;
                    ic.first_of_warp = (it == 0);
                    ic.reset_cur = first_of_scale;
                    ic.n = it;
                    rc = iterate(P.exact_math != 0, pl, g, l_t, theta, taut, first_of_scale, &ic, 0, st);
                    h->slots.push_back({s, wp});
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
                MI_HIP_TRY(hipEventRecord(h->ev_pool[e1], st));
                h->regions.push_back({e0, e1, nlaunch, 64.0 * g.w * g.h * B * iters_per_warp});
            }
        }
        Ctl ec = ctl;
        ec.q_prev = q_last;
        const bool dev_cur = check && !first_of_scale;
        if (s == 0) {
            rc = pack_flow(h->tab_dev, u1v, u2v, g, dev_cur ? &ec : nullptr, cur, st);
            if (rc) return rc;
            break;
        }
        // zoom the flow to the next finer scale and rescale it (tvl1flow.cpp:291-300)
        const Geo &gf = h->L[s - 1].g;
        // u3 is zoomed too but NOT rescaled (tvl1flow.cpp:293-300; optflow tvl1flow.cpp:524-528)
        const float *us[3][2] = {{Lv.u[0][0], Lv.u[1][0]}, {Lv.u[0][1], Lv.u[1][1]}, {gam ? Lv.u[0][2] : nullptr, gam ? Lv.u[1][2] : nullptr}};
        float *ud[3] = {h->L[s - 1].u[0][0], h->L[s - 1].u[0][1], gam ? h->L[s - 1].u[0][2] : nullptr};
        const float inv = (float)(1.0 / P.scale_step);
        const float post[3] = {inv, inv, 1.f};
        rc = resize(sem, gam ? 3 : 2, us, 2, ud, g, gf, (double)gf.w / g.w, (double)gf.h / g.h, post,
                    dev_cur ? &ec : nullptr, cur, st);
        if (rc) return rc;
    }
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
    const int nq = (int)h->slots.size();
    std::vector<int2> S(nq);
    MI_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    MI_HIP_TRY(hipMemcpy(S.data(), h->S + (size_t)pair * h->Q, sizeof(int2) * nq, hipMemcpyDeviceToHost));
    for (int i = 0; i < nq; ++i)
        if (S[i].y & 1) iters[h->slots[i].scale * nw + h->slots[i].warp] += 1;
    return MI_OK;
}
