// Stage-level C-ABI entry points of the TV-L1 path (the reference's internal device-layer
// boundary: tvl1flow::centeredGradient / warpBackward / estimateU / estimateDualVariables,
// modules/cudaoptflow/src/tvl1flow.cpp:58-76, and cuda::resize).  Caller planes may be
// arbitrarily pitched; they are staged through dense 256-B-aligned scratch planes.
#include <algorithm>
#include "tvl1_dev.h"
#include "mi_selftest.h"
#include <vector>

using namespace mi;
using namespace mi::tvl1;

namespace {

struct Stage {
    std::vector<float *> bufs;
    ~Stage() { for (float *p : bufs) (void)hipFree(p); }
    float *alloc(const Geo &g)
    {
        float *p = nullptr;
        if (hipMalloc((void **)&p, sizeof(float) * (size_t)g.ps) != hipSuccess) return nullptr;
        bufs.push_back(p);
        return p;
    }
};

Geo geo_of(int w, int h)
{
    Geo g;
    g.w = w; g.h = h; g.ld = align_up(w, 64); g.ps = (long long)g.ld * h; g.batch = 1;
    return g;
}

int check_f32(const mi_mat *m, const char *name)
{
    MI_REQUIRE(m && m->data, MI_ERR_BAD_ARG, "%s: null matrix", name);
    MI_REQUIRE(m->type == MI_32FC1, MI_ERR_BAD_TYPE, "%s: must be CV_32FC1", name);
    MI_REQUIRE(m->rows > 0 && m->cols > 0, MI_ERR_BAD_SIZE, "%s: empty", name);
    MI_REQUIRE(m->step >= (size_t)m->cols * 4 && m->step % 4 == 0, MI_ERR_BAD_ARG, "%s: bad step", name);
    return MI_OK;
}

int stage_in(Stage &S, const mi_mat *m, const Geo &g, float **out, hipStream_t st)
{
    float *p = S.alloc(g);
    MI_REQUIRE(p, MI_ERR_OOM, "stage allocation failed");
    MI_HIP_TRY(hipMemcpy2DAsync(p, (size_t)g.ld * 4, m->data, m->step, (size_t)g.w * 4, (size_t)g.h, hipMemcpyDeviceToDevice, st));
    *out = p;
    return MI_OK;
}

int stage_out(const float *p, const Geo &g, mi_mat *m, hipStream_t st)
{
    MI_HIP_TRY(hipMemcpy2DAsync(m->data, m->step, p, (size_t)g.ld * 4, (size_t)g.w * 4, (size_t)g.h, hipMemcpyDeviceToDevice, st));
    return MI_OK;
}

#define TRY(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

}  // namespace

extern "C" {

int mi_tvl1_centered_gradient(const mi_mat *src, mi_mat *dx, mi_mat *dy, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    TRY(check_f32(src, "src")); TRY(check_f32(dx, "dx")); TRY(check_f32(dy, "dy"));
    MI_REQUIRE(dx->rows == src->rows && dx->cols == src->cols && dy->rows == src->rows && dy->cols == src->cols,
               MI_ERR_BAD_SIZE, "dx/dy size != src size");
    const Geo g = geo_of(src->cols, src->rows);
    Stage S;
    float *s, *ox = S.alloc(g), *oy = S.alloc(g);
    MI_REQUIRE(ox && oy, MI_ERR_OOM, "stage allocation failed");
    TRY(stage_in(S, src, g, &s, st));
    TRY(gradient(s, ox, oy, g, st));
    TRY(stage_out(ox, g, dx, st)); TRY(stage_out(oy, g, dy, st));
    MI_HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_tvl1_warp_backward(int semantics, const mi_mat *I0, const mi_mat *I1, const mi_mat *I1x, const mi_mat *I1y,
                          const mi_mat *u1, const mi_mat *u2, mi_mat *I1w, mi_mat *I1wx, mi_mat *I1wy, mi_mat *grad,
                          mi_mat *rho)
{
    hipStream_t st = nullptr;
    // test hook: semantics | MI_WARP_STAGE_FAST runs the fast-math form of the fused-gradient kernel (separable bicubic sums)
    const bool fast = (semantics & MI_WARP_STAGE_FAST) != 0;
    const int lds = (semantics & MI_WARP_STAGE_LDS) ? 1 : (semantics & MI_WARP_STAGE_GATHER) ? 0 : -1;
    semantics &= ~(MI_WARP_STAGE_FAST | MI_WARP_STAGE_LDS | MI_WARP_STAGE_GATHER);
    MI_REQUIRE(semantics == MI_SEM_CPU_REF || semantics == MI_SEM_CUDA_COMPAT, MI_ERR_BAD_ARG, "bad semantics");
    // I1x == I1y == NULL: the derivative planes are the centred differences of I1, formed inside the warp kernel (the
    // kernel calc() runs, tvl1_warp_kernels.hip); otherwise the caller's planes are gathered (any planes, k_warp)
    const bool fused = !I1x && !I1y;
    MI_REQUIRE(fused || (I1x && I1y), MI_ERR_BAD_ARG, "I1x and I1y must be both given or both NULL");
    MI_REQUIRE(I0 && I1, MI_ERR_BAD_ARG, "null matrix");
    const mi_mat *ins[6] = {I0, I1, fused ? I1 : I1x, fused ? I1 : I1y, u1, u2};
    mi_mat *outs[5] = {I1w, I1wx, I1wy, grad, rho};
    for (int i = 0; i < 6; ++i) { TRY(check_f32(ins[i], "input")); MI_REQUIRE(ins[i]->rows == I0->rows && ins[i]->cols == I0->cols, MI_ERR_BAD_SIZE, "input size mismatch"); }
    for (int i = 0; i < 5; ++i) { TRY(check_f32(outs[i], "output")); MI_REQUIRE(outs[i]->rows == I0->rows && outs[i]->cols == I0->cols, MI_ERR_BAD_SIZE, "output size mismatch"); }
    const Geo g = geo_of(I0->cols, I0->rows);
    Stage S;
    float *in[6], *out[5];
    for (int i = 0; i < 6; ++i) TRY(stage_in(S, ins[i], g, &in[i], st));
    for (int i = 0; i < 5; ++i) { out[i] = S.alloc(g); MI_REQUIRE(out[i], MI_ERR_OOM, "stage allocation failed"); }
    float tabh[128], *tabd = nullptr;
    host_cubic_table(tabh);
    MI_HIP_TRY(hipMalloc((void **)&tabd, sizeof(tabh)));
    S.bufs.push_back(tabd);
    MI_HIP_TRY(hipMemcpyAsync(tabd, tabh, sizeof(tabh), hipMemcpyHostToDevice, st));
    const float *u1v[2] = {in[4], in[4]}, *u2v[2] = {in[5], in[5]};
    if (fused) {
        TRY(warp_fused(semantics, fast, lds, in[0], in[1], u1v, u2v, out[0], out[1], out[2], out[3], out[4], tabd, g, nullptr, 0, st));
    } else {
        float *pk = nullptr;   // {I1, I1x, I1y, 0} per pixel, the layout the gather kernel reads
        MI_HIP_TRY(hipMalloc((void **)&pk, sizeof(float) * 4 * (size_t)g.ps));
        S.bufs.push_back(pk);
        TRY(pack3(in[1], in[2], in[3], pk, g, st));
        TRY(warp(semantics, in[0], pk, u1v, u2v, out[0], out[1], out[2], out[3], out[4], tabd, g, nullptr, 0, st));
    }
    for (int i = 0; i < 5; ++i) TRY(stage_out(out[i], g, outs[i], st));
    MI_HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_tvl1_iterate(int exact_math, int time_block, int niter, const mi_mat *I1wx, const mi_mat *I1wy, const mi_mat *grad,
                    const mi_mat *rho_c, const mi_mat *u_in, const mi_mat *p_in, mi_mat *u_out, mi_mat *p_out, float l_t,
                    float theta, float taut, double *err_host, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    MI_REQUIRE(niter >= 1, MI_ERR_BAD_ARG, "niter must be >= 1");
    // test hook: time_block = 100 + T runs the INDEPENDENT-wave kernel of block length T (every wave its own 64-column strip) where
    // T alone runs the kernel of record (T = 10: four joined waves per 256-column strip) -- the two must agree bit for bit
    const bool indep = !exact_math && time_block >= 100;
    if (indep) time_block -= 100;
    const bool tiled = !exact_math && time_block < 0;   // test hook: register-tile kernel, variant -time_block - 1
    MI_REQUIRE(!tiled || -time_block - 1 < tile_variants(), MI_ERR_BAD_ARG, "no such register-tile variant");
    const bool blocked = (!exact_math && time_block > 0) || tiled;
    MI_REQUIRE(!(blocked && err_host), MI_ERR_BAD_ARG, "per-iteration error sums are not available from the blocked kernel");
    MI_REQUIRE(u_in && p_in && u_out && p_out, MI_ERR_BAD_ARG, "null plane array");
    const mi_mat *stat[4] = {I1wx, I1wy, grad, rho_c};
    for (int i = 0; i < 4; ++i) { TRY(check_f32(stat[i], "static plane")); MI_REQUIRE(stat[i]->rows == I1wx->rows && stat[i]->cols == I1wx->cols, MI_ERR_BAD_SIZE, "size mismatch"); }
    for (int i = 0; i < 2; ++i) { TRY(check_f32(&u_in[i], "u_in")); TRY(check_f32(&u_out[i], "u_out")); }
    for (int i = 0; i < 4; ++i) { TRY(check_f32(&p_in[i], "p_in")); TRY(check_f32(&p_out[i], "p_out")); }
    const Geo g = geo_of(I1wx->cols, I1wx->rows);
    Stage S;
    float *sp[4];
    for (int i = 0; i < 4; ++i) TRY(stage_in(S, stat[i], g, &sp[i], st));
    IterPlanes pl;
    memset(&pl, 0, sizeof(pl));
    pl.ix = sp[0]; pl.iy = sp[1]; pl.g = sp[2]; pl.rc = sp[3];
    for (int i = 0; i < 2; ++i) { TRY(stage_in(S, &u_in[i], g, &pl.u[0][i], st)); pl.u[1][i] = S.alloc(g); MI_REQUIRE(pl.u[1][i], MI_ERR_OOM, "oom"); }
    for (int i = 0; i < 4; ++i) { TRY(stage_in(S, &p_in[i], g, &pl.p[0][i], st)); pl.p[1][i] = S.alloc(g); MI_REQUIRE(pl.p[1][i], MI_ERR_OOM, "oom"); }
    Ctl ctl;
    memset(&ctl, 0, sizeof(ctl));
    if (err_host) {
        MI_HIP_TRY(hipMalloc((void **)&ctl.S, sizeof(int2) * niter));
        S.bufs.push_back((float *)ctl.S);
        MI_HIP_TRY(hipMalloc((void **)&ctl.E, sizeof(unsigned long long) * niter));
        S.bufs.push_back((float *)ctl.E);
        MI_HIP_TRY(hipMemsetAsync(ctl.E, 0, sizeof(unsigned long long) * niter, st));
        ctl.Q = niter;
        ctl.thr = -1.0;  // always active
    }
    int cur = 0;
    for (int it = 0; it < niter;) {
        if (tiled) {
            const int T = std::min(niter - it, tile_max_block());
            TRY(iterate_tile(-time_block - 1, T, pl, g, l_t, theta, taut, false, cur, st));
            it += T;
            cur ^= 1;
            continue;
        }
        if (blocked) {
            const int T = tb_pick_block(niter - it, time_block);
            // rows_per_band = -1: always the streaming kernel (the tile kernel is compared against it)
            TRY(iterate_tb(T, pl, g, l_t, theta, taut, false, cur, -1, st, false, indep));
            it += T;
            cur ^= 1;
            continue;
        }
        ++it;
        if (err_host) {
            const int it0 = it - 1;
            Ctl c = ctl;
            c.q = it0; c.q_prev = it0 - 1; c.first_of_warp = (it0 == 0); c.reset_cur = (it0 == 0);
            TRY(iterate(exact_math != 0, pl, g, l_t, theta, taut, false, &c, 0, st));
        } else {
            TRY(iterate(exact_math != 0, pl, g, l_t, theta, taut, false, nullptr, cur, st));
        }
        cur ^= 1;
    }
    for (int i = 0; i < 2; ++i) TRY(stage_out(pl.u[cur][i], g, &u_out[i], st));
    for (int i = 0; i < 4; ++i) TRY(stage_out(pl.p[cur][i], g, &p_out[i], st));
    MI_HIP_TRY(hipStreamSynchronize(st));
    if (err_host) {
        std::vector<unsigned long long> e(niter);
        MI_HIP_TRY(hipMemcpy(e.data(), ctl.E, sizeof(unsigned long long) * niter, hipMemcpyDeviceToHost));
        for (int i = 0; i < niter; ++i) err_host[i] = (double)e[i] / 16777216.0;
    }
    return MI_OK;
}

int mi_resize_linear(int semantics, const mi_mat *src, mi_mat *dst, double fx, double fy, int explicit_dsize, float post_scale,
                     void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    MI_REQUIRE(semantics == MI_SEM_CPU_REF || semantics == MI_SEM_CUDA_COMPAT, MI_ERR_BAD_ARG, "bad semantics");
    TRY(check_f32(src, "src")); TRY(check_f32(dst, "dst"));
    const Geo gs = geo_of(src->cols, src->rows), gd = geo_of(dst->cols, dst->rows);
    double isx = fx, isy = fy;
    if (explicit_dsize) { isx = (double)gd.w / gs.w; isy = (double)gd.h / gs.h; }
    MI_REQUIRE(isx > 0 && isy > 0, MI_ERR_BAD_ARG, "fx, fy must be > 0 when dsize is not explicit");
    Stage S;
    float *s, *d = S.alloc(gd);
    MI_REQUIRE(d, MI_ERR_OOM, "stage allocation failed");
    TRY(stage_in(S, src, gs, &s, st));
    const float *srcs[3][2] = {{s, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    float *dsts[3] = {d, nullptr, nullptr};
    const float post[3] = {post_scale, 1.f, 1.f};
    if (semantics == MI_SEM_CUDA_COMPAT && gd.w == gs.w && gd.h == gs.h) {
        // dsize == src.size(): plain copy (cudawarping/src/resize.cpp:89-93)
        MI_HIP_TRY(hipMemcpyAsync(d, s, sizeof(float) * (size_t)gs.ps, hipMemcpyDeviceToDevice, st));
    } else {
        TRY(resize(semantics, 1, srcs, 1, dsts, gs, gd, isx, isy, post, nullptr, 0, st));
    }
    TRY(stage_out(d, gd, dst, st));
    MI_HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

int miflow_selftest_lane_shift(int *out_host)
{
    MI_REQUIRE(out_host, MI_ERR_BAD_ARG, "null out");
    int *d = nullptr;
    MI_HIP_TRY(hipMalloc((void **)&d, 128 * sizeof(int)));
    int rc = dbg_lane_shift(d, nullptr);
    if (rc == MI_OK) {
        hipError_t e = hipMemcpy(out_host, d, 128 * sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error("hipMemcpy failed: %s", hipGetErrorString(e)); rc = MI_ERR_HIP; }
    }
    (void)hipFree(d);
    return rc;
}

int miflow_selftest_jw_fault(int *fault)
{
    MI_REQUIRE(fault, MI_ERR_BAD_ARG, "null out");
    return tb_jw_fault(fault);
}

}  // extern "C"
