// cv::cuda::DensePyrLKOpticalFlow for gfx950 (second part).  Replaces cudaoptflow/src/pyrlk.cpp:238-299 and
// cudaoptflow/src/cuda/pyrlk.cu:709-847 behind the C-ABI; texture reads are defined as in oracle/pyrlk_ref.c (texel centre /
// one common sub-pixel offset per window, clamp addressing), arithmetic separately rounded in the same order => bit-identical
// to that restatement.
//
// Construction: 16 x 16 pixels per workgroup, the integer patches {I, dI/dx, dI/dy} of the tile + window halo in LDS (like the
// reference); the J window of a Newton step is sampled on the integer lattice with ONE bilinear weight pair, row by row: the
// horizontally interpolated row is kept in registers and reused by the next window row, so a step reads (w+1)^2 texels instead
// of 4 w^2 (the reference pays 4 texture-filtered fetches' worth per sample in hardware; there is no texture unit here).
#include "farneback_dev.h"
#include <cfloat>

struct mi_densepyrlk {
    mi_densepyrlk_params P;
    float *buf = nullptr;
    size_t buf_floats = 0;
};

namespace mi {
namespace lk {

struct Args {
    const float *I, *J;      // level images, dense rows of ld_img floats
    int rows, cols, ld_img;
    float *u, *v;            // output (full-size buffers, ld_uv floats per row)
    const float *pu, *pv;    // previous (coarser) level's flow, indexed [y/2][x/2]
    int ld_uv;
    int wx, wy, hx, hy, iters;
};

__device__ __forceinline__ float tx(const float *im, int rows, int cols, int ld, int y, int x)
{
    return im[(size_t)min(max(y, 0), rows - 1) * ld + min(max(x, 0), cols - 1)];
}

// WXT > 0: window width known at compile time (the horizontally interpolated row lives in registers); 0: generic
template <int WXT>
__global__ __launch_bounds__(256) void k_dense(Args A)
{
    extern __shared__ int smem[];
    const int pw = 16 + 2 * A.hx, ph = 16 + 2 * A.hy;
    int *Ip = smem, *dxp = Ip + pw * ph, *dyp = dxp + pw * ph;
    const int xb = blockIdx.x * 16, yb = blockIdx.y * 16;
    const int rows = A.rows, cols = A.cols, ld = A.ld_img;
    for (int e = threadIdx.x; e < pw * ph; e += 256) {
        const int i = e / pw, j = e - i * pw;
        const int yy = yb - A.hy + i, xx = xb - A.hx + j;
#define TI(a, b) tx(A.I, rows, cols, ld, (a), (b))
        const float a00 = TI(yy - 1, xx - 1), a01 = TI(yy - 1, xx), a02 = TI(yy - 1, xx + 1);
        const float a10 = TI(yy, xx - 1), a11 = TI(yy, xx), a12 = TI(yy, xx + 1);
        const float a20 = TI(yy + 1, xx - 1), a21 = TI(yy + 1, xx), a22 = TI(yy + 1, xx + 1);
#undef TI
        Ip[e] = (int)a11;
        // Scharr (pyrlk.cu:735-739), left to right
        dxp[e] = (int)(3 * a02 + 10 * a12 + 3 * a22 - (3 * a00 + 10 * a10 + 3 * a20));
        dyp[e] = (int)(3 * a20 + 10 * a21 + 3 * a22 - (3 * a00 + 10 * a01 + 3 * a02));
    }
    __syncthreads();
    const int txi = threadIdx.x & 15, tyi = threadIdx.x >> 4;
    const int x = xb + txi, y = yb + tyi;
    if (x >= cols || y >= rows) return;
    const int wx = WXT > 0 ? WXT : A.wx, wy = A.wy;
    unsigned A11i = 0, A12i = 0, A22i = 0;
    for (int i = 0; i < wy; ++i)
        for (int j = 0; j < wx; ++j) {
            const int e = (tyi + i) * pw + txi + j;
            const int gx = dxp[e], gy = dyp[e];
            A11i += (unsigned)(gx * gx); A12i += (unsigned)(gx * gy); A22i += (unsigned)(gy * gy);
        }
    float A11 = (float)(int)A11i, A12 = (float)(int)A12i, A22 = (float)(int)A22i;
    float D = A11 * A22 - A12 * A12;
    if (D < FLT_EPSILON) return;   // pyrlk.cu:777-782: (u, v) stay unwritten
    D = 1.f / D;
    A11 *= D; A12 *= D; A22 *= D;
    float nx = x + A.pu[(size_t)(y / 2) * A.ld_uv + x / 2] * 2.0f;
    float ny = y + A.pv[(size_t)(y / 2) * A.ld_uv + x / 2] * 2.0f;
    for (int k = 0; k < A.iters; ++k) {
        if (nx < 0 || nx >= cols || ny < 0 || ny >= rows) return;   // pyrlk.cu:796-802
        const float bx = nx - A.hx, by = ny - A.hy;
        const float x0f = floorf(bx), y0f = floorf(by);
        const float fx = bx - x0f, fy = by - y0f, gxw = 1.0f - fx, gyw = 1.0f - fy;
        const int x0 = (int)x0f, y0 = (int)y0f;
        unsigned b1 = 0, b2 = 0;
        if (WXT > 0) {
            float hprev[WXT > 0 ? WXT : 1];
            {
                float t = tx(A.J, rows, cols, ld, y0, x0);
#pragma unroll
                for (int j = 0; j < WXT; ++j) {
                    const float t1 = tx(A.J, rows, cols, ld, y0, x0 + j + 1);
                    hprev[j] = t * gxw + t1 * fx;
                    t = t1;
                }
            }
            for (int i = 0; i < wy; ++i) {
                float t = tx(A.J, rows, cols, ld, y0 + i + 1, x0);
                const int eb = (tyi + i) * pw + txi;
#pragma unroll
                for (int j = 0; j < WXT; ++j) {
                    const float t1 = tx(A.J, rows, cols, ld, y0 + i + 1, x0 + j + 1);
                    const float hcur = t * gxw + t1 * fx;
                    t = t1;
                    const int Jv = (int)(hprev[j] * gyw + hcur * fy);
                    hprev[j] = hcur;
                    const int diff = (Jv - Ip[eb + j]) * 32;
                    b1 += (unsigned)(diff * dxp[eb + j]);
                    b2 += (unsigned)(diff * dyp[eb + j]);
                }
            }
        } else {
            for (int i = 0; i < wy; ++i)
                for (int j = 0; j < wx; ++j) {
                    const float t00 = tx(A.J, rows, cols, ld, y0 + i, x0 + j), t01 = tx(A.J, rows, cols, ld, y0 + i, x0 + j + 1);
                    const float t10 = tx(A.J, rows, cols, ld, y0 + i + 1, x0 + j), t11 = tx(A.J, rows, cols, ld, y0 + i + 1, x0 + j + 1);
                    const float top = t00 * gxw + t01 * fx, bot = t10 * gxw + t11 * fx;
                    const int Jv = (int)(top * gyw + bot * fy);
                    const int e = (tyi + i) * pw + txi + j;
                    const int diff = (Jv - Ip[e]) * 32;
                    b1 += (unsigned)(diff * dxp[e]);
                    b2 += (unsigned)(diff * dyp[e]);
                }
        }
        const float fb1 = (float)(int)b1, fb2 = (float)(int)b2;
        const float ddx = A12 * fb2 - A22 * fb1, ddy = A12 * fb1 - A11 * fb2;
        nx += ddx; ny += ddy;
        if (fabsf(ddx) < 0.01f && fabsf(ddy) < 0.01f) break;
    }
    A.u[(size_t)y * A.ld_uv + x] = nx - x;
    A.v[(size_t)y * A.ld_uv + x] = ny - y;
}

}  // namespace lk
}  // namespace mi

using namespace mi;

extern "C" {

void mi_densepyrlk_default_params(mi_densepyrlk_params *p)
{
    if (!p) return;
    p->win_width = 13; p->win_height = 13; p->max_level = 3; p->iters = 30; p->use_initial_flow = 0;   // cudaoptflow.hpp DensePyrLKOpticalFlow::create
}

int mi_densepyrlk_create(const mi_densepyrlk_params *p, mi_densepyrlk **out)
{
    MI_REQUIRE(out, MI_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device available: the miflow product path has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    mi_densepyrlk *h = new mi_densepyrlk();
    if (p) h->P = *p; else mi_densepyrlk_default_params(&h->P);
    *out = h;
    return MI_OK;
}

int mi_densepyrlk_set_params(mi_densepyrlk *h, const mi_densepyrlk_params *p)
{
    MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument");
    h->P = *p;
    return MI_OK;
}

int mi_densepyrlk_get_params(const mi_densepyrlk *h, mi_densepyrlk_params *p)
{
    MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument");
    *p = h->P;
    return MI_OK;
}

void mi_densepyrlk_destroy(mi_densepyrlk *h)
{
    if (!h) return;
    if (h->buf) (void)hipFree(h->buf);
    delete h;
}

int mi_densepyrlk_calc(mi_densepyrlk *h, const mi_mat *prev, const mi_mat *next, mi_mat *flow, void *stream)
{
    MI_REQUIRE(h && prev && next && flow && prev->data && next->data && flow->data, MI_ERR_BAD_ARG, "null argument");
    const mi_densepyrlk_params &P = h->P;
    MI_REQUIRE(prev->type == MI_8UC1, MI_ERR_BAD_TYPE, "prevImg.type() == CV_8UC1");                                  // pyrlk.cpp:240
    MI_REQUIRE(next->type == prev->type && next->rows == prev->rows && next->cols == prev->cols, MI_ERR_BAD_SIZE,
               "prevImg.size() == nextImg.size() && prevImg.type() == nextImg.type()");
    MI_REQUIRE(P.max_level >= 0, MI_ERR_BAD_ARG, "maxLevel >= 0");
    MI_REQUIRE(P.win_width > 2 && P.win_height > 2, MI_ERR_BAD_ARG, "winSize.width > 2 && winSize.height > 2");
    MI_REQUIRE(P.win_width <= 31 && P.win_height <= 31 && P.max_level <= 16, MI_ERR_BAD_ARG, "winSize <= 31, maxLevel <= 16");
    MI_REQUIRE(flow->type == MI_32FC2 && flow->rows == prev->rows && flow->cols == prev->cols, MI_ERR_BAD_SIZE,
               "flow must be CV_32FC2 of the image size");
    hipStream_t st = (hipStream_t)stream;
    const int rows = prev->rows, cols = prev->cols, nl = P.max_level + 1;
    fb::Plane g[17];
    size_t total = 0;
    g[0] = fb::plane_of(cols, rows);
    for (int l = 1; l < nl; ++l) g[l] = fb::plane_of((g[l - 1].w + 1) / 2, (g[l - 1].h + 1) / 2);
    for (int l = 0; l < nl; ++l) total += 2 * (size_t)g[l].ld * g[l].h;
    const size_t uvn = (size_t)g[0].ld * g[0].h;
    total += 4 * uvn;
    if (h->buf_floats < total) {
        if (h->buf) { (void)hipFree(h->buf); h->buf = nullptr; h->buf_floats = 0; }
        MI_HIP_TRY(hipMalloc(&h->buf, total * sizeof(float)));
        h->buf_floats = total;
    }
    float *Pp[17], *Np[17], *cur = h->buf;
    for (int l = 0; l < nl; ++l) { Pp[l] = cur; cur += (size_t)g[l].ld * g[l].h; Np[l] = cur; cur += (size_t)g[l].ld * g[l].h; }
    float *U[2] = {cur, cur + uvn}, *V[2] = {cur + 2 * uvn, cur + 3 * uvn};
    int rc;
    // prevImg.convertTo(prevPyr_[0], CV_32F) / pyrDown, pyrlk.cpp:252-259
    if ((rc = fb::convert(prev->data, (long long)prev->step, next->data, (long long)next->step, MI_8UC1, Pp[0], Np[0], g[0], st))) return rc;
    for (int l = 1; l < nl; ++l) {
        if ((rc = fb::pyr_down(Pp[l - 1], g[l - 1], Pp[l], g[l], st))) return rc;
        if ((rc = fb::pyr_down(Np[l - 1], g[l - 1], Np[l], g[l], st))) return rc;
    }
    MI_HIP_TRY(hipMemsetAsync(U[0], 0, 4 * uvn * sizeof(float), st));   // uPyr / vPyr .setTo(0), pyrlk.cpp:272-275
    lk::Args A;
    A.wx = P.win_width; A.wy = P.win_height; A.hx = (P.win_width - 1) / 2; A.hy = (P.win_height - 1) / 2; A.iters = P.iters;
    A.ld_uv = g[0].ld;
    const size_t lds = 3 * (size_t)(16 + 2 * A.hx) * (16 + 2 * A.hy) * sizeof(int);
    int idx = 0;
    for (int l = P.max_level; l >= 0; --l) {   // pyrlk.cpp:284-295
        const int idx2 = (idx + 1) & 1;
        A.I = Pp[l]; A.J = Np[l]; A.rows = g[l].h; A.cols = g[l].w; A.ld_img = g[l].ld;
        A.u = U[idx]; A.v = V[idx]; A.pu = U[idx2]; A.pv = V[idx2];
        const dim3 grid(div_up(A.cols, 16), div_up(A.rows, 16));
        if (A.wx == 13) hipLaunchKernelGGL((lk::k_dense<13>), grid, dim3(256), lds, st, A);
        else if (A.wx == 21) hipLaunchKernelGGL((lk::k_dense<21>), grid, dim3(256), lds, st, A);
        else hipLaunchKernelGGL((lk::k_dense<0>), grid, dim3(256), lds, st, A);
        if (l > 0) idx = idx2;
    }
    MI_HIP_TRY(hipGetLastError());
    return fb::merge_flow(U[idx], V[idx], flow->data, (long long)flow->step, g[0], st);   // cuda::merge, pyrlk.cpp:390-391
}

}  // extern "C"
