// cv::cuda::FarnebackOpticalFlow: handle, level loop and C-ABI entry points.  Host-side twin of
// FarnebackOpticalFlowImpl (modules/cudaoptflow/src/farneback.cpp:96-492), ordered on ONE stream (the
// reference fans out over 5 streams and blocks the host once per level, :319-324,366,456).
#include "farneback_dev.h"
#include "fb_groups.h"
#include "fb_plan.h"
#include "mi_selftest.h"
#include "tvl1_dev.h"   // tvl1::resize == cuda::resize(INTER_LINEAR) on dense f32 planes
#include <cfloat>
#include <cmath>
#include <vector>

using namespace mi;
using namespace mi::fb;

namespace mi { namespace fb {

// cv::getGaussianKernel(n, sigma, CV_32F) (main repo imgproc): fixed tables for n <= 7 with sigma <= 0,
// else exp(-x^2/(2 sigma^2)) normalised to sum 1
void gaussian_kernel(int n, double sigma, float *k)
{
    static const float t1[] = {1.f}, t3[] = {0.25f, 0.5f, 0.25f}, t5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
                       t7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    if (sigma <= 0 && n <= 7 && (n & 1)) {
        const float *t = n == 1 ? t1 : n == 3 ? t3 : n == 5 ? t5 : t7;
        for (int i = 0; i < n; ++i) k[i] = t[i];
        return;
    }
    const double sx = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2 = -0.5 / (sx * sx);
    std::vector<double> w(n);
    double sum = 0;
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; w[i] = std::exp(scale2 * x * x); sum += w[i]; }
    for (int i = 0; i < n; ++i) k[i] = (float)(w[i] / sum);
}

// prepareGaussian, farneback.cpp:209-260: Gaussian-weighted moments and the 4 used entries of inv(G)
int prepare_gaussian(int n, double sigma, PolyC *C)
{
    if (sigma < FLT_EPSILON) sigma = n * 0.3;   // :270-271
    float gb[17], *g = gb + 8;
    double s = 0.;
    for (int x = -n; x <= n; x++) { g[x] = (float)std::exp(-x * x / (2 * sigma * sigma)); s += g[x]; }
    s = 1. / s;
    memset(C, 0, sizeof(*C));
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        if (x >= 0) { C->g[x] = g[x]; C->xg[x] = (float)(x * g[x]); C->xxg[x] = (float)(x * x * g[x]); }
    }
    double G[6][6] = {{0}};
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0][0] += g[y] * g[x];
            G[1][1] += g[y] * g[x] * x * x;
            G[3][3] += g[y] * g[x] * x * x * x * x;
            G[5][5] += g[y] * g[x] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    // inv(G) by Cholesky (Mat::inv(DECOMP_CHOLESKY), :254)
    double L[6][6] = {{0}}, inv[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double a = G[i][j];
            for (int k = 0; k < j; ++k) a -= L[i][k] * L[j][k];
            if (i == j) { if (a <= 0) { set_error("moment matrix is not positive definite"); return MI_ERR_BAD_ARG; } L[i][i] = std::sqrt(a); }
            else L[i][j] = a / L[j][j];
        }
    for (int c = 0; c < 6; ++c) {
        double yv[6], xv[6];
        for (int i = 0; i < 6; ++i) { double a = i == c ? 1.0 : 0.0; for (int k = 0; k < i; ++k) a -= L[i][k] * yv[k]; yv[i] = a / L[i][i]; }
        for (int i = 5; i >= 0; --i) { double a = yv[i]; for (int k = i + 1; k < 6; ++k) a -= L[k][i] * xv[k]; xv[i] = a / L[i][i]; }
        for (int i = 0; i < 6; ++i) inv[i][c] = xv[i];
    }
    C->ig11 = (float)inv[1][1]; C->ig03 = (float)inv[0][3]; C->ig33 = (float)inv[3][3]; C->ig55 = (float)inv[5][5];
    return MI_OK;
}

}}  // namespace mi::fb

struct mi_farneback {
    mi_farneback_params P;
    float *arena = nullptr;     // capB per-pair blocks of `bs` floats: every plane of pair b lives at (plane of pair 0) + b * bs
    int capW = 0, capH = 0, capB = 0;
    long long bs = 0;
    // full-size-capacity planes
    float *frames[2] = {}, *blurred = nullptr, *lvl[2] = {}, *R[2] = {}, *M = nullptr, *bufM = nullptr;
    float *flow[3][2] = {};   // [slot][x,y]: slot 0 = level 0 (and the initial flow), slots 1,2 alternate over the coarse levels
    std::vector<float *> pyr[2];
    std::vector<Plane> pyrg;
    float *pyr_base = nullptr;  // fast-pyramid levels of pair 0 (inside the pair block)
    // internal stream + events of the two-chain pair groups (enqueue_level).  ONE calc in flight per handle: a handle used from two host
    // threads, or on two streams without a synchronisation in between, races on the arena AND on this stream's ordering (c_api.h).
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

static int cv_round(double v) { return (int)std::lrint(v); }

extern "C" {

void mi_farneback_default_params(mi_farneback_params *p)
{
    if (!p) return;
    // cv::cuda::FarnebackOpticalFlow::create defaults, cudaoptflow.hpp:285-293
    p->num_levels = 5; p->pyr_scale = 0.5; p->fast_pyramids = 0; p->win_size = 13; p->num_iters = 10;
    p->poly_n = 5; p->poly_sigma = 1.1; p->flags = 0;
}

static int validate(const mi_farneback_params *p)
{
    MI_REQUIRE(p, MI_ERR_BAD_ARG, "null params");
    MI_REQUIRE(p->poly_n == 5 || p->poly_n == 7, MI_ERR_BAD_ARG, "polyN must be 5 or 7");                                   // farneback.cpp:316
    MI_REQUIRE(!p->fast_pyramids || std::abs(p->pyr_scale - 0.5) < 1e-6, MI_ERR_BAD_ARG, "fastPyramids requires pyrScale == 0.5");  // :317
    MI_REQUIRE(p->num_levels >= 0 && p->num_levels <= 32, MI_ERR_BAD_ARG, "numLevels out of range");
    MI_REQUIRE(p->pyr_scale > 0 && p->pyr_scale < 1, MI_ERR_BAD_ARG, "pyrScale must be in (0,1)");
    MI_REQUIRE(p->win_size >= 1 && (p->win_size & 1) && p->win_size / 2 <= MI_FB_MAX_KSIZE_HALF, MI_ERR_BAD_ARG, "winSize must be odd and <= 201");
    MI_REQUIRE(p->num_iters >= 0, MI_ERR_BAD_ARG, "numIters must be >= 0");
    return MI_OK;
}

int mi_farneback_create(const mi_farneback_params *p, mi_farneback **out)
{
    MI_REQUIRE(out, MI_ERR_BAD_ARG, "null out");
    *out = nullptr;
    mi_farneback_params d;
    if (!p) { mi_farneback_default_params(&d); p = &d; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device available: the miflow product path has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    mi_farneback *h = new mi_farneback();
    h->P = *p;   // validated at calc(), like the reference (CV_Assert inside calcImpl)
    *out = h;
    return MI_OK;
}

int mi_farneback_set_params(mi_farneback *h, const mi_farneback_params *p)
{
    MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument");
    h->P = *p;
    return MI_OK;
}

int mi_farneback_get_params(const mi_farneback *h, mi_farneback_params *p)
{
    MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument");
    *p = h->P;
    return MI_OK;
}

void mi_farneback_destroy(mi_farneback *h)
{
    if (!h) return;
    if (h->arena) (void)hipFree(h->arena);
    if (h->aux) { (void)hipStreamSynchronize(h->aux); (void)hipStreamDestroy(h->aux); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    delete h;
}

static int ensure(mi_farneback *h, int W, int H, int B)
{
    if (h->arena && h->capW == W && h->capH == H && h->capB >= B) return MI_OK;
    if (h->arena) (void)hipFree(h->arena);
    h->arena = nullptr;
    const Plane g = plane_of(W, H);
    const size_t n = (size_t)g.ld * H;
    // per pair: frames 2, blurred 2, lvl 2, R 2x5, M 5, bufM 5, flows 6 = 32 planes, + the fast-pyramid levels of both frames
    // (sum over the half-size levels < 2/3 of a plane per frame: 2 planes reserved).  The two frames' planes of a kind are ADJACENT
    // (frame 1 = frame 0 + one plane; R[1] = R[0] + five): the pyramid kernels run both frames in one launch
    // (round 4's all-levels-at-once expansion planes and the internal-stream pyramid they served measured slower, r08i, and are gone)
    const size_t per_pair = n * 34;
    MI_HIP_TRY(hipMalloc((void **)&h->arena, sizeof(float) * per_pair * (size_t)B));
    float *p = h->arena;
    auto take = [&](size_t k) { float *q = p; p += n * k; return q; };
    h->frames[0] = take(1); h->frames[1] = take(1); h->blurred = take(2); h->lvl[0] = take(1); h->lvl[1] = take(1);
    h->R[0] = take(5); h->R[1] = take(5); h->M = take(5); h->bufM = take(5);
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 2; ++b) h->flow[a][b] = take(1);
    h->pyr_base = take(2);
    h->bs = (long long)per_pair;
    h->capW = W; h->capH = H; h->capB = B;
    return MI_OK;
}

static void geo_tvl1(const Plane &p, mi::tvl1::Geo &g) { g.w = p.w; g.h = p.h; g.ld = p.ld; g.ps = p.batch > 1 ? p.bs : (long long)p.ld * p.h; g.batch = p.batch; }

// cuda::resize(src -> dst, Size(dw,dh), INTER_LINEAR) followed by convertTo(*alpha) on up to 2 planes
static int resize2(const float *s0, const float *s1, const Plane &gs, float *d0, float *d1, const Plane &gd, float alpha, hipStream_t st)
{
    if (gs.w == gd.w && gs.h == gd.h && alpha == 1.f) {   // dsize == src.size(): copy (cudawarping/src/resize.cpp:89-93)
        const size_t bytes = sizeof(float) * (size_t)gs.ld * gs.h, pitch = sizeof(float) * (size_t)(gs.batch > 1 ? gs.bs : (long long)gs.ld * gs.h);
        MI_HIP_TRY(hipMemcpy2DAsync(d0, pitch, s0, pitch, bytes, (size_t)gs.batch, hipMemcpyDeviceToDevice, st));
        if (s1) MI_HIP_TRY(hipMemcpy2DAsync(d1, pitch, s1, pitch, bytes, (size_t)gs.batch, hipMemcpyDeviceToDevice, st));
        return MI_OK;
    }
    mi::tvl1::Geo a, b;
    geo_tvl1(gs, a); geo_tvl1(gd, b);
    const float *src[3][2] = {{s0, nullptr}, {s1, nullptr}, {nullptr, nullptr}};
    float *dst[3] = {d0, d1, nullptr};
    const float post[3] = {alpha, alpha, 1.f};
    return mi::tvl1::resize(MI_SEM_CUDA_COMPAT, s1 ? 2 : 1, src, 1, dst, a, b, (double)gd.w / gs.w, (double)gd.h / gs.h, post, nullptr, 0, st);
}

static int check_pair(const mi_mat *I0, const mi_mat *I1, const mi_mat *flow, const mi_mat *ref)
{
    MI_REQUIRE(I0 && I1 && flow && I0->data && I1->data && flow->data, MI_ERR_BAD_ARG, "null matrix");
    // CV_Assert(frame0.channels() == 1 && frame1.channels() == 1)  farneback.cpp:173; depths supported here: 8U, 32F
    MI_REQUIRE((I0->type == MI_8UC1 || I0->type == MI_32FC1) && I1->type == I0->type, MI_ERR_BAD_TYPE, "frames must be CV_8UC1 or CV_32FC1, same type");
    MI_REQUIRE(I0->rows == I1->rows && I0->cols == I1->cols, MI_ERR_BAD_SIZE, "frame0.size() != frame1.size()");   // :174
    MI_REQUIRE(flow->type == MI_32FC2 && flow->rows == I0->rows && flow->cols == I0->cols, MI_ERR_BAD_SIZE, "flow must be CV_32FC2 of the frame size");  // :181-182
    MI_REQUIRE(flow->step % 8 == 0 && ((uintptr_t)flow->data % 8) == 0, MI_ERR_BAD_ARG, "flow must be 8-byte aligned");
    if (I0->type == MI_32FC1) MI_REQUIRE(I0->step % 4 == 0 && I1->step % 4 == 0, MI_ERR_BAD_ARG, "float frames must be 4-byte aligned");
    MI_REQUIRE(I0->rows == ref->rows && I0->cols == ref->cols && I0->type == ref->type, MI_ERR_BAD_SIZE, "all pairs of a batch must share size and type");
    return MI_OK;
}

// ---- one mi_farneback_calc_batch call: what the level loop needs besides the plan, and the state it carries from level to level
struct FbCall {
    mi_farneback *h;
    const FbPlan *plan;
    int B;
    long long bs, pn;          // floats between consecutive pairs' blocks / in one full-resolution plane (= the distance between the two frames' planes of a kind)
    Plane g0;                  // full-resolution geometry of the batch
    const mi_mat *I0s, *I1s;
    mi_mat *flows;
    hipStream_t st;            // the caller's stream
    bool gauss, direct;        // OPTFLOW_FARNEBACK_GAUSSIAN; the pre-blur reads the caller's matrices itself
    PolyC C;
    Taps wk;                   // the Gaussian window of the iterations (gauss)
    float *fx0, *fy0;          // level-0 flow planes (and the caller's initial flow)
    float *prevx = nullptr, *prevy = nullptr;   // the coarser level's flow
    Plane gprev;
    bool merged = false;       // the last iteration wrote the caller's flow matrix itself
};

// blur + resize + polynomial expansion of BOTH frames for level k (the reference's loop over the two frames, farneback.cpp:434-454,
// each stage once for both: 3 launches instead of 6), for pairs b0 .. b0 + nb - 1 of the batch: a pair group of the level runs its own
// stage, so that the expansions it is about to iterate on are still in the last-level cache
static int pyramid_stage(const FbCall &c, int k, hipStream_t sx, int b0, int nb)
{
    mi_farneback *h = c.h;
    const mi_farneback_params &P = h->P;
    const FbLevel &L = c.plan->lv[k];
    const long long po = (long long)b0 * c.bs, pn = c.pn;
    Plane g = plane_of(L.w, L.h, c.bs, nb), gf = c.g0;
    gf.batch = nb;
    float *const R0 = h->R[0] + po;
    const long long fsR = 5 * pn;   // R[1] = R[0] + five full-size planes
    if (P.fast_pyramids)
        return poly_exp(h->pyr[0][k] + po, R0, g, P.poly_n, c.C, sx, 2, h->pyr[1][k] - h->pyr[0][k], fsR);
    const int smoothSize = L.smooth;
    std::vector<float> gk(smoothSize);
    gaussian_kernel(smoothSize, L.sigma, gk.data());
    Taps K;
    memset(&K, 0, sizeof(K));
    for (int i = 0; i <= smoothSize / 2; ++i) K.k[i] = gk[smoothSize / 2 + i];
    int r;
    if (c.direct) {
        for (int c0 = 0; c0 < nb; c0 += kFmtPairs) {
            FmtTab T;
            memset(&T, 0, sizeof(T));
            Plane gc = gf;
            gc.batch = std::min(kFmtPairs, nb - c0);
            for (int j = 0; j < gc.batch; ++j) {
                T.a[j] = c.I0s[b0 + c0 + j].data; T.sa[j] = (long long)c.I0s[b0 + c0 + j].step;
                T.b[j] = c.I1s[b0 + c0 + j].data; T.sb[j] = (long long)c.I1s[b0 + c0 + j].step;
            }
            if ((r = gaussian_blur_tab(T, c.I0s[0].type, h->blurred + po + (long long)c0 * c.bs, gc, smoothSize / 2, K, sx, pn))) return r;
        }
    } else if ((r = gaussian_blur(h->frames[0] + po, h->blurred + po, gf, smoothSize / 2, K, MI_BORDER_REFLECT101, sx, 2, pn))) return r;
    const float *src = h->blurred + po;
    if (!(g.w == gf.w && g.h == gf.h)) {
        // the level image = cuda::resize of the blurred frame (:447-448).  Few pairs: sampled inside the expansion kernel (same
        // arithmetic, same bits, one launch less); many pairs: through the level planes (the expansion reads each sample 11 times)
        if (c.plan->fuse_small) return poly_exp(h->blurred + po, R0, g, P.poly_n, c.C, sx, 2, pn, fsR, &gf);
        if ((r = resize2(h->blurred + po, h->blurred + po + pn, gf, h->lvl[0] + po, h->lvl[1] + po, g, 1.f, sx))) return r;
        src = h->lvl[0] + po;
    }
    return poly_exp(src, R0, g, P.poly_n, c.C, sx, 2, pn, fsR);
}

// Level k of the plan: the flow planes' start, then -- for the whole batch or pair group by pair group (fb_groups.h) -- the frames' side,
// the first matrix update and the iterations (farneback.cpp:396-471; updateFlow_boxFilter / _gaussianBlur :278-312 fused into `iterate`).
static int enqueue_level(FbCall &c, int k)
{
    mi_farneback *h = c.h;
    const mi_farneback_params &P = h->P;
    const FbLevel &L = c.plan->lv[k];
    const int B = c.B;
    const long long bs = c.bs, pn = c.pn;
    hipStream_t const caller_st = c.st;
    const Plane g = plane_of(L.w, L.h, bs, B);
    float *curx, *cury;
    if (k > 0) { curx = h->flow[1 + (k & 1)][0]; cury = h->flow[1 + (k & 1)][1]; }
    else { curx = c.fx0; cury = c.fy0; }
    int rc;
    if (L.init_resize) {   // :398-404
        if ((rc = resize2(c.fx0, c.fy0, c.g0, curx, cury, g, (float)L.scale, caller_st))) return rc;
    } else if (L.clear_flow) {
        const size_t bytes = sizeof(float) * (size_t)g.ld * g.h;
        // the two planes are neighbours in the arena: one fill over both and the gap between them while that gap is small (a single
        // pair: one launch less); a batch would clear B full-size planes for two coarse ones (r16d: 57 us of a 32-pair calc)
        if (cury == curx + pn && bytes <= sizeof(float) * (size_t)pn && sizeof(float) * (size_t)pn * B <= (8u << 20)) {
            MI_HIP_TRY(hipMemset2DAsync(curx, sizeof(float) * (size_t)bs, 0, sizeof(float) * (size_t)pn + bytes, (size_t)B, caller_st));
        } else {
            MI_HIP_TRY(hipMemset2DAsync(curx, sizeof(float) * (size_t)bs, 0, bytes, (size_t)B, caller_st));
            MI_HIP_TRY(hipMemset2DAsync(cury, sizeof(float) * (size_t)bs, 0, bytes, (size_t)B, caller_st));
        }
    }   // (L.zero_flow: no fill -- the first matrix update takes the flow as zero and the iterations write every pixel of both planes;
        //  tests/test_farneback.py poisons the arena to hold every iterate variant to that)
    const float *R0 = h->R[0], *R1 = h->R[0] + 5 * pn;
    // Pair GROUPS (round 5; the plan: fb_groups.h).  An iteration of a level streams 22 planes per pair (M in and out, both expansions,
    // the flow); with the whole batch per launch a 640 x 480 level of 32 pairs is 0.9 GB per iteration -- every iteration comes from
    // HBM.  Group by group instead: the matrix update and ALL iterations of a few pairs whose planes fit the 256 MB last-level cache,
    // then the next group.  The same launches on the same data in the same order per pair: bit-identical.  Groups are independent
    // chains of launches: two of them run side by side, the second on the handle's internal stream, each half the size -- the
    // launches of a chain wait for each other's last workgroups (r16g), and the other chain fills those tails.
    const int G = L.groups.pairs;
    const bool two = L.groups.two;
    if (!L.staged) {   // once for the batch, as the reference orders them
        if (L.zoom && (rc = resize2(c.prevx, c.prevy, c.gprev, curx, cury, g, (float)(1. / P.pyr_scale), caller_st))) return rc;
        if ((rc = pyramid_stage(c, k, caller_st, 0, B))) return rc;
    }
    if (two) {
        if (!h->aux) MI_HIP_TRY(hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));
        if (!h->ev_fork) MI_HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        if (!h->ev_join) MI_HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        MI_HIP_TRY(hipEventRecord(h->ev_fork, caller_st));
        MI_HIP_TRY(hipStreamWaitEvent(h->aux, h->ev_fork, 0));
    }
    // an error return from inside the group loop still orders the caller's stream behind whatever the internal one was given
    struct ChainJoin {
        hipStream_t st, aux; hipEvent_t ev; bool armed;
        ~ChainJoin() { if (armed) { (void)hipEventRecord(ev, aux); (void)hipStreamWaitEvent(st, ev, 0); } }
    } chain_join = {caller_st, h->aux, h->ev_join, two};
    int gi = 0;
    for (int b0 = 0; b0 < B; b0 += G, ++gi) {
        hipStream_t st = (two && (gi & 1)) ? h->aux : caller_st;   // this group's chain
        const long long goff = (long long)b0 * bs;
        Plane gg = g;
        gg.batch = std::min(G, B - b0);
        float *M = h->M + goff, *bufM = h->bufM + goff;
        float *gx = curx + goff, *gy = cury + goff;
        const float *gR0 = R0 + goff, *gR1 = R1 + goff;
        Plane gp = c.gprev;
        gp.batch = gg.batch;
        if (L.staged) {
            if (L.zoom && (rc = resize2(c.prevx + goff, c.prevy + goff, gp, gx, gy, gg, (float)(1. / P.pyr_scale), st))) return rc;
            if ((rc = pyramid_stage(c, k, st, b0, gg.batch))) return rc;
        }
        if (L.zoom_fused) {
            if ((rc = update_matrices_resized(c.prevx + goff, c.prevy + goff, gp, (float)(1. / P.pyr_scale), gx, gy, gR0, gR1, M, gg, st))) return rc;
        } else if ((rc = update_matrices(L.zero_flow ? nullptr : gx, L.zero_flow ? nullptr : gy, gR0, gR1, M, gg, st))) return rc;   // :458
        for (int i = 0; i < P.num_iters; i++) {   // :465-471 -> :278-312, fused
            if (L.pair_it && i + 1 < P.num_iters) {
                const bool last2 = k == 0 && i + 1 == P.num_iters - 1 && B == 1;
                bool dm2 = false;
                if ((rc = iterate2(M, gR0, gR1, gx, gy, bufM, gg, P.win_size, c.gauss ? &c.wk : nullptr, i + 1 < P.num_iters - 1, st,
                                   last2 ? c.flows[0].data : nullptr, last2 ? (long long)c.flows[0].step : 0, &dm2))) return rc;
                c.merged = c.merged || dm2;
                std::swap(M, bufM);
                ++i;
                continue;
            }
            const bool last = k == 0 && i == P.num_iters - 1 && B == 1 && c.plan->fuse_small;
            bool dm = false;
            if ((rc = iterate(M, gR0, gR1, gx, gy, bufM, gg, P.win_size, c.gauss ? &c.wk : nullptr, i < P.num_iters - 1, st,
                              last ? c.flows[0].data : nullptr, last ? (long long)c.flows[0].step : 0, &dm))) return rc;
            c.merged = c.merged || dm;
            std::swap(M, bufM);
        }
    }
    if (two) {
        chain_join.armed = false;
        MI_HIP_TRY(hipEventRecord(h->ev_join, h->aux));
        MI_HIP_TRY(hipStreamWaitEvent(caller_st, h->ev_join, 0));
    }
    c.prevx = curx; c.prevy = cury; c.gprev = g;
    return MI_OK;
}

// n independent pairs per launch sequence: blockIdx.z = pair in every kernel of the level loop (a 640 x 480 pair alone is about
// 70 launches of 5-50 us, i.e. launch-latency bound; batched, the same 70 launches carry n pairs).
int mi_farneback_calc_batch(mi_farneback *h, int nb, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows, void *stream)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    MI_REQUIRE(nb > 0 && I0s && I1s && flows, MI_ERR_BAD_ARG, "empty batch");
    hipStream_t st = (hipStream_t)stream;
    const mi_farneback_params &P = h->P;
    int rc = validate(&P);
    if (rc) return rc;
    for (int i = 0; i < nb; ++i) if ((rc = check_pair(&I0s[i], &I1s[i], &flows[i], &I0s[0]))) return rc;
    const int W = I0s[0].cols, H = I0s[0].rows, B = nb;
    MI_REQUIRE(W >= 2 && H >= 2, MI_ERR_BAD_SIZE, "frames too small");
    if ((rc = ensure(h, W, H, B))) return rc;
    const long long bs = h->bs;
    const Plane g0 = plane_of(W, H, bs, B);
    const Plane g0s = plane_of(W, H);   // single-pair view for the per-pair conversion kernels
    const bool use_init = (P.flags & MI_OPTFLOW_USE_INITIAL_FLOW) != 0;
    const bool gauss = (P.flags & MI_OPTFLOW_FARNEBACK_GAUSSIAN) != 0;

    // flowx0/flowy0 (level-0 flow planes; also the caller's initial flow)
    float *fx0 = h->flow[0][0], *fy0 = h->flow[0][1];
    // convertTo(CV_32F), farneback.cpp:342-345; the caller's initial flow split into planes
    auto convert_frames = [&]() -> int {
        if (B > 1) {   // one launch per 64 pairs (blockIdx.z = pair) instead of one per pair
            for (int b0 = 0; b0 < B; b0 += kFmtPairs) {
                const int nb = std::min(kFmtPairs, B - b0);
                FmtTab T;
                memset(&T, 0, sizeof(T));
                for (int j = 0; j < nb; ++j) {
                    T.a[j] = I0s[b0 + j].data; T.sa[j] = (long long)I0s[b0 + j].step;
                    T.b[j] = I1s[b0 + j].data; T.sb[j] = (long long)I1s[b0 + j].step;
                }
                const int r = convert_batch(T, nb, I0s[0].type, h->frames[0] + b0 * bs, h->frames[1] + b0 * bs, bs, g0s, st);
                if (r) return r;
            }
            return MI_OK;
        }
        return convert(I0s[0].data, (long long)I0s[0].step, I1s[0].data, (long long)I1s[0].step, I0s[0].type, h->frames[0], h->frames[1], g0s, st);
    };
    if (P.fast_pyramids && (rc = convert_frames())) return rc;   // the fast pyramid is built from the converted frames, below
    for (int b = 0; b < B && use_init; ++b)
        if ((rc = split_flow(flows[b].data, (long long)flows[b].step, fx0 + b * bs, fy0 + b * bs, g0s, st))) return rc;

    // ---- the plan (fb_plan.h: pure arithmetic over the call's shape and the tuning knobs, unit-tested without a device)
    FbKnobs knobs;
    knobs.fuse = tuning().fb_fuse; knobs.pair = tuning().fb_pair; knobs.group_mb = tuning().fb_group_mb;
    knobs.chains = tuning().fb_group_streams == 2 ? 2 : 1;
    {   // groups alternate over two streams only while the caller's stream is NOT being captured (a captured graph stays one chain);
        // asked once per call
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (knobs.chains == 2 && (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)) { (void)hipGetLastError(); knobs.chains = 1; }
    }
    knobs.simds = device_simds();
    knobs.iterate2_ok = iterate2_supported(P.win_size);
    const FbShape shape = {W, H, B, P.num_levels, P.pyr_scale, P.fast_pyramids != 0, P.num_iters, use_init};
    const FbPlan plan = fb_make_plan(shape, knobs);
    const int levels = plan.levels;
    for (int k = 0; k <= levels; ++k)
        MI_REQUIRE(plan.lv[k].smooth / 2 <= MI_FB_MAX_KSIZE_HALF, MI_ERR_BAD_ARG, "pyramid smoothing kernel too large (MAX_KSIZE_HALF)");

    FbCall c;
    c.h = h; c.plan = &plan; c.B = B; c.bs = bs; c.g0 = g0; c.pn = (long long)g0.ld * H;
    c.I0s = I0s; c.I1s = I1s; c.flows = flows; c.st = st; c.gauss = gauss;
    c.fx0 = fx0; c.fy0 = fy0; c.gprev = g0;
    if (P.fast_pyramids) {   // :346-359
        std::vector<Plane> pg(levels + 1);
        for (int i = 0; i <= levels; ++i) pg[i] = plane_of(plan.lv[i].w, plan.lv[i].h, bs, B);
        h->pyrg = pg;
        for (int f = 0; f < 2; ++f) {
            h->pyr[f].assign(levels + 1, nullptr);
            h->pyr[f][0] = h->frames[f];
        }
        float *p = h->pyr_base;
        for (int i = 1; i <= levels; ++i)
            for (int f = 0; f < 2; ++f) {
                h->pyr[f][i] = p; p += (size_t)pg[i].ld * pg[i].h;
                if ((rc = pyr_down(h->pyr[f][i - 1], pg[i - 1], h->pyr[f][i], pg[i], st))) return rc;
            }
    }
    if ((rc = prepare_gaussian(P.poly_n, P.poly_sigma, &c.C))) return rc;   // setPolynomialExpansionConsts :361
    memset(&c.wk, 0, sizeof(c.wk));
    if (gauss) {   // :460-464
        std::vector<float> k(P.win_size);
        gaussian_kernel(P.win_size, (double)(P.win_size / 2 * 0.3f), k.data());
        for (int i = 0; i <= P.win_size / 2; ++i) c.wk.k[i] = k[P.win_size / 2 + i];
    }
    // The frames as f32 planes -- unless every level's pre-blur can read the caller's matrices itself (`direct`, round 5: the tiled blur
    // has an instantiation for every level's kernel size): then the conversion pass and its planes are not needed at all.
    c.direct = !P.fast_pyramids && tuning().fb_direct != 0;
    for (int k = 0; k <= levels && c.direct; ++k) c.direct = gaussian_blur_tab_ok(g0, plan.lv[k].smooth / 2);
    if (!P.fast_pyramids && !c.direct && (rc = convert_frames())) return rc;

    for (int k = levels; k >= 0; k--)
        if ((rc = enqueue_level(c, k))) return rc;
    const bool merged = c.merged;
    if (merged) return MI_OK;
    if (B > 1) {
        for (int b0 = 0; b0 < B; b0 += kFmtPairs) {
            const int nb = std::min(kFmtPairs, B - b0);
            FmtTab T;
            memset(&T, 0, sizeof(T));
            for (int j = 0; j < nb; ++j) { T.a[j] = flows[b0 + j].data; T.sa[j] = (long long)flows[b0 + j].step; }
            if ((rc = merge_flow_batch(fx0 + b0 * bs, fy0 + b0 * bs, T, nb, bs, g0s, st))) return rc;   // cuda::merge :197-198
        }
        return MI_OK;
    }
    for (int b = 0; b < B; ++b)
        if ((rc = merge_flow(fx0 + b * bs, fy0 + b * bs, flows[b].data, (long long)flows[b].step, g0s, st))) return rc;   // cuda::merge :197-198
    return MI_OK;
}

int mi_farneback_calc(mi_farneback *h, const mi_mat *I0, const mi_mat *I1, mi_mat *flow, void *stream)
{
    return mi_farneback_calc_batch(h, 1, I0, I1, flow, stream);
}

// test hook (mi_selftest.h): NaNs over the whole arena
int miflow_selftest_farneback_poison(mi_farneback *h, void *stream)
{
    MI_REQUIRE(h && h->arena, MI_ERR_BAD_ARG, "no arena yet: run a calc first");
    MI_HIP_TRY(hipMemsetAsync(h->arena, 0xff, sizeof(float) * (size_t)h->bs * (size_t)h->capB, (hipStream_t)stream));
    return MI_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ stage-level entry points
namespace {
struct Stage {
    std::vector<float *> bufs;
    ~Stage() { for (float *p : bufs) (void)hipFree(p); }
    float *alloc(size_t n)
    {
        float *p = nullptr;
        if (hipMalloc((void **)&p, sizeof(float) * n) != hipSuccess) return nullptr;
        bufs.push_back(p);
        return p;
    }
};
int check_f32(const mi_mat *m, const char *name)
{
    MI_REQUIRE(m && m->data, MI_ERR_BAD_ARG, "%s: null matrix", name);
    MI_REQUIRE(m->type == MI_32FC1, MI_ERR_BAD_TYPE, "%s: must be CV_32FC1", name);
    MI_REQUIRE(m->rows > 0 && m->cols > 0, MI_ERR_BAD_SIZE, "%s: empty", name);
    MI_REQUIRE(m->step >= (size_t)m->cols * 4 && m->step % 4 == 0, MI_ERR_BAD_ARG, "%s: bad step", name);
    return MI_OK;
}
// rows x cols matrix (rows may be 5*h) -> dense plane(s) with pitch ld
int stage_in(Stage &S, const mi_mat *m, int ld, float **out, hipStream_t st)
{
    float *p = S.alloc((size_t)ld * m->rows);
    MI_REQUIRE(p, MI_ERR_OOM, "stage allocation failed");
    MI_HIP_TRY(hipMemcpy2DAsync(p, (size_t)ld * 4, m->data, m->step, (size_t)m->cols * 4, (size_t)m->rows, hipMemcpyDeviceToDevice, st));
    *out = p;
    return MI_OK;
}
int stage_out(const float *p, int ld, mi_mat *m, hipStream_t st)
{
    MI_HIP_TRY(hipMemcpy2DAsync(m->data, m->step, p, (size_t)ld * 4, (size_t)m->cols * 4, (size_t)m->rows, hipMemcpyDeviceToDevice, st));
    return MI_OK;
}
#define TRY(x) do { int _rc = (x); if (_rc) return _rc; } while (0)
int make_win_taps(int ksize, Taps *K)
{
    MI_REQUIRE(ksize >= 1 && (ksize & 1) && ksize / 2 <= MI_FB_MAX_KSIZE_HALF, MI_ERR_BAD_ARG, "ksize must be odd and <= 201");
    std::vector<float> k(ksize);
    gaussian_kernel(ksize, (double)(ksize / 2 * 0.3f), k.data());
    memset(K, 0, sizeof(*K));
    for (int i = 0; i <= ksize / 2; ++i) K->k[i] = k[ksize / 2 + i];
    return MI_OK;
}
}  // namespace

extern "C" {

int mi_farneback_poly_exp(const mi_mat *src, mi_mat *dst5, int poly_n, double poly_sigma, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    TRY(check_f32(src, "src")); TRY(check_f32(dst5, "dst"));
    MI_REQUIRE(dst5->rows == 5 * src->rows && dst5->cols == src->cols, MI_ERR_BAD_SIZE, "dst must be 5*rows x cols");
    MI_REQUIRE(poly_n == 5 || poly_n == 7, MI_ERR_BAD_ARG, "polyN must be 5 or 7");
    const Plane g = plane_of(src->cols, src->rows);
    Stage S;
    float *in = nullptr, *out = S.alloc((size_t)g.ld * g.h * 5);
    MI_REQUIRE(out, MI_ERR_OOM, "stage allocation failed");
    TRY(stage_in(S, src, g.ld, &in, st));
    PolyC C;
    TRY(prepare_gaussian(poly_n, poly_sigma, &C));
    TRY(poly_exp(in, out, g, poly_n, C, st));
    TRY(stage_out(out, g.ld, dst5, st));
    MI_HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_farneback_update_matrices(const mi_mat *flowx, const mi_mat *flowy, const mi_mat *R0, const mi_mat *R1, mi_mat *M5, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    TRY(check_f32(flowx, "flowx")); TRY(check_f32(flowy, "flowy")); TRY(check_f32(R0, "R0")); TRY(check_f32(R1, "R1")); TRY(check_f32(M5, "M"));
    const int w = flowx->cols, hh = flowx->rows;
    MI_REQUIRE(flowy->rows == hh && flowy->cols == w && R0->rows == 5 * hh && R1->rows == 5 * hh && M5->rows == 5 * hh &&
               R0->cols == w && R1->cols == w && M5->cols == w, MI_ERR_BAD_SIZE, "size mismatch");
    const Plane g = plane_of(w, hh);
    Stage S;
    float *fx, *fy, *r0, *r1, *m = S.alloc((size_t)g.ld * hh * 5);
    MI_REQUIRE(m, MI_ERR_OOM, "stage allocation failed");
    TRY(stage_in(S, flowx, g.ld, &fx, st)); TRY(stage_in(S, flowy, g.ld, &fy, st));
    TRY(stage_in(S, R0, g.ld, &r0, st)); TRY(stage_in(S, R1, g.ld, &r1, st));
    TRY(update_matrices(fx, fy, r0, r1, m, g, st));
    TRY(stage_out(m, g.ld, M5, st));
    MI_HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_farneback_blur5(const mi_mat *M5, mi_mat *dst5, int ksize, int gaussian, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    TRY(check_f32(M5, "M")); TRY(check_f32(dst5, "dst"));
    MI_REQUIRE(M5->rows % 5 == 0 && dst5->rows == M5->rows && dst5->cols == M5->cols, MI_ERR_BAD_SIZE, "size mismatch");
    const Plane g = plane_of(M5->cols, M5->rows / 5);
    Taps K;
    TRY(make_win_taps(ksize, &K));
    Stage S;
    float *m, *out = S.alloc((size_t)g.ld * g.h * 5);
    MI_REQUIRE(out, MI_ERR_OOM, "stage allocation failed");
    TRY(stage_in(S, M5, g.ld, &m, st));
    TRY(blur5(m, out, g, ksize, gaussian ? &K : nullptr, st));
    TRY(stage_out(out, g.ld, dst5, st));
    MI_HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_farneback_update_flow(const mi_mat *M5, mi_mat *flowx, mi_mat *flowy, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    TRY(check_f32(M5, "M")); TRY(check_f32(flowx, "flowx")); TRY(check_f32(flowy, "flowy"));
    MI_REQUIRE(M5->rows == 5 * flowx->rows && M5->cols == flowx->cols && flowy->rows == flowx->rows && flowy->cols == flowx->cols,
               MI_ERR_BAD_SIZE, "size mismatch");
    const Plane g = plane_of(flowx->cols, flowx->rows);
    Stage S;
    float *m, *fx = S.alloc((size_t)g.ld * g.h), *fy = S.alloc((size_t)g.ld * g.h);
    MI_REQUIRE(fx && fy, MI_ERR_OOM, "stage allocation failed");
    TRY(stage_in(S, M5, g.ld, &m, st));
    TRY(update_flow(m, fx, fy, g, st));
    TRY(stage_out(fx, g.ld, flowx, st)); TRY(stage_out(fy, g.ld, flowy, st));
    MI_HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_farneback_iterate(const mi_mat *M5, const mi_mat *R0, const mi_mat *R1, mi_mat *flowx, mi_mat *flowy, mi_mat *M5out,
                         int ksize, int gaussian, int update, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    TRY(check_f32(M5, "M")); TRY(check_f32(R0, "R0")); TRY(check_f32(R1, "R1")); TRY(check_f32(flowx, "flowx")); TRY(check_f32(flowy, "flowy"));
    TRY(check_f32(M5out, "Mout"));
    const int w = flowx->cols, hh = flowx->rows;
    MI_REQUIRE(flowy->rows == hh && flowy->cols == w && M5->rows == 5 * hh && R0->rows == 5 * hh && R1->rows == 5 * hh && M5out->rows == 5 * hh &&
               M5->cols == w && R0->cols == w && R1->cols == w && M5out->cols == w, MI_ERR_BAD_SIZE, "size mismatch");
    MI_REQUIRE(M5->data != M5out->data, MI_ERR_BAD_ARG, "Mout must not alias M");
    const Plane g = plane_of(w, hh);
    Taps K;
    TRY(make_win_taps(ksize, &K));
    Stage S;
    float *m, *r0, *r1, *fx = S.alloc((size_t)g.ld * hh), *fy = S.alloc((size_t)g.ld * hh), *mo = S.alloc((size_t)g.ld * hh * 5);
    MI_REQUIRE(fx && fy && mo, MI_ERR_OOM, "stage allocation failed");
    TRY(stage_in(S, M5, g.ld, &m, st)); TRY(stage_in(S, R0, g.ld, &r0, st)); TRY(stage_in(S, R1, g.ld, &r1, st));
    MI_HIP_TRY(hipMemsetAsync(mo, 0, sizeof(float) * (size_t)g.ld * hh * 5, st));
    TRY(iterate(m, r0, r1, fx, fy, mo, g, ksize, gaussian ? &K : nullptr, update != 0, st));
    TRY(stage_out(fx, g.ld, flowx, st)); TRY(stage_out(fy, g.ld, flowy, st)); TRY(stage_out(mo, g.ld, M5out, st));
    MI_HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_farneback_gaussian_blur(const mi_mat *src, mi_mat *dst, int ksize, double sigma, int border, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    TRY(check_f32(src, "src")); TRY(check_f32(dst, "dst"));
    MI_REQUIRE(dst->rows == src->rows && dst->cols == src->cols, MI_ERR_BAD_SIZE, "size mismatch");
    MI_REQUIRE(ksize >= 1 && (ksize & 1) && ksize / 2 <= MI_FB_MAX_KSIZE_HALF, MI_ERR_BAD_ARG, "ksize must be odd and <= 201");
    const Plane g = plane_of(src->cols, src->rows);
    std::vector<float> k(ksize);
    gaussian_kernel(ksize, sigma, k.data());
    Taps K;
    memset(&K, 0, sizeof(K));
    for (int i = 0; i <= ksize / 2; ++i) K.k[i] = k[ksize / 2 + i];
    Stage S;
    float *in, *out = S.alloc((size_t)g.ld * g.h);
    MI_REQUIRE(out, MI_ERR_OOM, "stage allocation failed");
    TRY(stage_in(S, src, g.ld, &in, st));
    TRY(gaussian_blur(in, out, g, ksize / 2, K, border, st));
    TRY(stage_out(out, g.ld, dst, st));
    MI_HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_pyr_down(const mi_mat *src, mi_mat *dst, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    TRY(check_f32(src, "src")); TRY(check_f32(dst, "dst"));
    MI_REQUIRE(dst->rows == (src->rows + 1) / 2 && dst->cols == (src->cols + 1) / 2, MI_ERR_BAD_SIZE, "dst must be ((rows+1)/2, (cols+1)/2)");
    const Plane gs = plane_of(src->cols, src->rows), gd = plane_of(dst->cols, dst->rows);
    Stage S;
    float *in, *out = S.alloc((size_t)gd.ld * gd.h);
    MI_REQUIRE(out, MI_ERR_OOM, "stage allocation failed");
    TRY(stage_in(S, src, gs.ld, &in, st));
    TRY(pyr_down(in, gs, out, gd, st));
    TRY(stage_out(out, gd.ld, dst, st));
    MI_HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

}  // extern "C"
