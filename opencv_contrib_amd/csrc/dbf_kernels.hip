// cv::cuda::DisparityBilateralFilter for gfx950 (first part): joint bilateral refinement of a disparity map at
// its discontinuities, red/black passes.  Replaces cudastereo/src/cuda/disparity_bilateral_filter.cu:76-199 behind the C-ABI.
//
// Differences in construction (not in arithmetic): a pass is OUT of place (read the map of the previous pass, write the next),
// which makes the result independent of thread timing (the reference's in-place pass lets a thread read same-colour pixels
// that another thread of the same launch may or may not have rewritten); a lane owns a horizontal pixel PAIR (the one of the
// pass colour is refined, the other copied), both tables sit in LDS, and the five candidate costs are accumulated in the
// reference's order (window rows top to bottom, columns left to right, separately rounded binary32 multiply and add).
#include "mi_common.h"
#include <cmath>
#include <vector>

struct mi_disp_bilateral {
    mi_disp_bilateral_params P;
    float *tab = nullptr;        // device: 255 colour weights, then (radius+1)^2 space weights
    int tab_radius = -1;
    float tab_sigma = -1.f;
    void *tmp = nullptr;         // ping-pong map
    size_t tmp_bytes = 0;
};

namespace mi {
namespace dbf {

struct Args {
    const unsigned char *src; size_t sstep;
    unsigned char *dst; size_t dstep;
    const unsigned char *img; size_t istep;
    const float *tab;
    int rows, cols, radius, t;
    short edge_disc, max_disc;
};

template <typename T, int CN>
__global__ __launch_bounds__(256) void k_pass(Args A)
{
    extern __shared__ float s_tab[];   // [255 colour][(r+1)^2 space]
    const int half = A.radius, ntab = 255 + (half + 1) * (half + 1);
    for (int i = threadIdx.x; i < ntab; i += 256) s_tab[i] = A.tab[i];
    __syncthreads();
    const float *ctab = s_tab, *stab = s_tab + 255;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int xp = (blockIdx.x * 64 + (threadIdx.x & 63)) * 2;   // pixel pair xp, xp + 1
    if (y >= A.rows || xp >= A.cols) return;
    const T *srow = reinterpret_cast<const T *>(A.src + (size_t)y * A.sstep);
    T *drow = reinterpret_cast<T *>(A.dst + (size_t)y * A.dstep);
    const int x = xp + ((y + A.t) & 1);    // the pixel of this pass's colour (disparity_bilateral_filter.cu:84)
    const int xo = xp + 1 - ((y + A.t) & 1);
    if (xo < A.cols) drow[xo] = srow[xo];
    if (x >= A.cols) return;
    int out = srow[x];
    if (y > 0 && y < A.rows - 1 && x > 0 && x < A.cols - 1) {
        const T *up = reinterpret_cast<const T *>(A.src + (size_t)(y - 1) * A.sstep);
        const T *dn = reinterpret_cast<const T *>(A.src + (size_t)(y + 1) * A.sstep);
        const int dp0 = srow[x], dp1 = up[x], dp2 = srow[x - 1], dp3 = dn[x], dp4 = srow[x + 1];
        const int e = A.edge_disc;
        if (abs(dp1 - dp0) >= e || abs(dp2 - dp0) >= e || abs(dp3 - dp0) >= e || abs(dp4 - dp0) >= e) {
            const int ymin = max(0, y - half), xmin = max(0, x - half);
            const int ymax = min(A.rows - 1, y + half), xmax = min(A.cols - 1, x + half);
            float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f;
            const unsigned char *ic = A.img + (size_t)y * A.istep + (size_t)CN * x;
            const int i0 = ic[0], i1 = CN == 3 ? ic[1] : 0, i2 = CN == 3 ? ic[2] : 0;
            const int md = A.max_disc;
            for (int yi = ymin; yi <= ymax; ++yi) {
                const T *dy = reinterpret_cast<const T *>(A.src + (size_t)yi * A.sstep);
                const unsigned char *iy = A.img + (size_t)yi * A.istep;
                const float *srowtab = stab + abs(y - yi) * (half + 1);
                for (int xi = xmin; xi <= xmax; ++xi) {
                    const unsigned char *in = iy + (size_t)CN * xi;
                    int d = abs((int)in[0] - i0);
                    if (CN == 3) d = max(max(d, abs((int)in[1] - i1)), abs((int)in[2] - i2));
                    const float weight = ctab[d] * srowtab[abs(x - xi)];
                    const int dr = dy[xi];
                    c0 += (float)min(md, abs(dr - dp0)) * weight;
                    c1 += (float)min(md, abs(dr - dp1)) * weight;
                    c2 += (float)min(md, abs(dr - dp2)) * weight;
                    c3 += (float)min(md, abs(dr - dp3)) * weight;
                    c4 += (float)min(md, abs(dr - dp4)) * weight;
                }
            }
            // first strict minimum in candidate order (disparity_bilateral_filter.cu:121-148)
            float m = 3.402823466e+38f;
            int best = dp0;
            if (c0 < m) { m = c0; best = dp0; }
            if (c1 < m) { m = c1; best = dp1; }
            if (c2 < m) { m = c2; best = dp2; }
            if (c3 < m) { m = c3; best = dp3; }
            if (c4 < m) { m = c4; best = dp4; }
            out = best;
        }
    }
    drow[x] = (T)out;
}

static int build_tables(mi_disp_bilateral *h)
{
    const mi_disp_bilateral_params &P = h->P;
    if (h->tab && h->tab_radius == P.radius && h->tab_sigma == P.sigma_range) return MI_OK;
    const int half = P.radius, n = 255 + (half + 1) * (half + 1);
    std::vector<float> t(n);
    // calc_color_weighted_table / calc_space_weighted_filter, disparity_bilateral_filter.cpp:96-123
    const float den = 2 * P.sigma_range * P.sigma_range;
    for (int i = 0; i < 255; ++i) t[i] = static_cast<float>(std::exp(-double(i * i) / den));
    const float dist_space = P.radius + 1.0f;
    for (int y = 0; y <= half; ++y)
        for (int x = 0; x <= half; ++x) t[255 + y * (half + 1) + x] = std::exp(-std::sqrt(float(y * y) + float(x * x)) / dist_space);
    if (h->tab) { (void)hipFree(h->tab); h->tab = nullptr; }
    MI_HIP_TRY(hipMalloc(&h->tab, sizeof(float) * n));
    MI_HIP_TRY(hipMemcpy(h->tab, t.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    h->tab_radius = P.radius;
    h->tab_sigma = P.sigma_range;
    return MI_OK;
}

template <typename T>
static void launch(int cn, const Args &A, dim3 grid, size_t lds, hipStream_t st)
{
    if (cn == 1) hipLaunchKernelGGL((k_pass<T, 1>), grid, dim3(256), lds, st, A);
    else hipLaunchKernelGGL((k_pass<T, 3>), grid, dim3(256), lds, st, A);
}

}  // namespace dbf
}  // namespace mi

using namespace mi;

extern "C" {

void mi_disp_bilateral_default_params(mi_disp_bilateral_params *p)
{
    if (!p) return;
    p->ndisp = 64; p->radius = 3; p->iters = 1;
    p->edge_threshold = 0.1f; p->max_disc_threshold = 0.2f; p->sigma_range = 10.0f;
}

int mi_disp_bilateral_create(const mi_disp_bilateral_params *p, mi_disp_bilateral **out)
{
    MI_REQUIRE(out, MI_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device available: the miflow product path has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    mi_disp_bilateral *h = new mi_disp_bilateral();
    if (p) h->P = *p; else mi_disp_bilateral_default_params(&h->P);
    *out = h;
    return MI_OK;
}

int mi_disp_bilateral_set_params(mi_disp_bilateral *h, const mi_disp_bilateral_params *p)
{
    MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument");
    h->P = *p;   // validated at apply(), like the reference's setters (disparity_bilateral_filter.cpp:176)
    return MI_OK;
}

int mi_disp_bilateral_get_params(const mi_disp_bilateral *h, mi_disp_bilateral_params *p)
{
    MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument");
    *p = h->P;
    return MI_OK;
}

void mi_disp_bilateral_destroy(mi_disp_bilateral *h)
{
    if (!h) return;
    if (h->tab) (void)hipFree(h->tab);
    if (h->tmp) (void)hipFree(h->tmp);
    delete h;
}

int mi_disp_bilateral_apply(mi_disp_bilateral *h, const mi_mat *disp, const mi_mat *img, mi_mat *dst, void *stream)
{
    MI_REQUIRE(h && disp && img && dst && disp->data && img->data && dst->data, MI_ERR_BAD_ARG, "null argument");
    const mi_disp_bilateral_params &P = h->P;
    MI_REQUIRE(0 < P.ndisp && 0 < P.radius && 0 < P.iters, MI_ERR_BAD_ARG, "0 < ndisp && 0 < radius && 0 < iters");   // .cpp:176
    MI_REQUIRE(P.radius <= 64, MI_ERR_BAD_ARG, "radius <= 64 (LDS table)");
    MI_REQUIRE(disp->type == MI_8UC1 || disp->type == MI_16SC1, MI_ERR_BAD_TYPE, "disp.type() == CV_8U || disp.type() == CV_16S");
    MI_REQUIRE(img->type == MI_8UC1 || img->type == MI_8UC3, MI_ERR_BAD_TYPE, "img.type() == CV_8UC1 || img.type() == CV_8UC3");
    MI_REQUIRE(disp->rows > 0 && disp->cols > 0 && img->rows == disp->rows && img->cols == disp->cols, MI_ERR_BAD_SIZE,
               "disp.size() == img.size()");
    MI_REQUIRE(dst->type == disp->type && dst->rows == disp->rows && dst->cols == disp->cols, MI_ERR_BAD_SIZE,
               "dst must have the type and size of disp");
    hipStream_t st = (hipStream_t)stream;
    int rc = dbf::build_tables(h);
    if (rc) return rc;
    const int es = disp->type == MI_8UC1 ? 1 : 2, cn = img->type == MI_8UC1 ? 1 : 3;
    const size_t tstep = (size_t)align_up(disp->cols * es, 256), need = tstep * disp->rows;
    if (h->tmp_bytes < need) {
        if (h->tmp) { (void)hipFree(h->tmp); h->tmp = nullptr; h->tmp_bytes = 0; }
        MI_HIP_TRY(hipMalloc(&h->tmp, need));
        h->tmp_bytes = need;
    }
    if (dst->data != disp->data)   // disp.copyTo(dst), .cpp:155-156
        MI_HIP_TRY(hipMemcpy2DAsync(dst->data, dst->step, disp->data, disp->step, (size_t)disp->cols * es, disp->rows, hipMemcpyDeviceToDevice, st));
    dbf::Args A;
    A.img = (const unsigned char *)img->data; A.istep = img->step;
    A.tab = h->tab;
    A.rows = disp->rows; A.cols = disp->cols; A.radius = P.radius;
    // .cpp:146-147
    short edge_disc = short(P.ndisp * P.edge_threshold + 0.5);
    A.edge_disc = edge_disc < 1 ? short(1) : edge_disc;
    A.max_disc = short(P.ndisp * P.max_disc_threshold + 0.5);
    const dim3 grid(div_up(div_up(A.cols, 2), 64), div_up(A.rows, 4));
    const size_t lds = sizeof(float) * (255 + (P.radius + 1) * (P.radius + 1));
    for (int i = 0; i < P.iters; ++i)
        for (int t = 0; t < 2; ++t) {   // cu:164-170; pass 0: dst -> tmp, pass 1: tmp -> dst
            A.t = t;
            if (t == 0) { A.src = (const unsigned char *)dst->data; A.sstep = dst->step; A.dst = (unsigned char *)h->tmp; A.dstep = tstep; }
            else { A.src = (const unsigned char *)h->tmp; A.sstep = tstep; A.dst = (unsigned char *)dst->data; A.dstep = dst->step; }
            if (es == 1) dbf::launch<unsigned char>(cn, A, grid, lds, st);
            else dbf::launch<short>(cn, A, grid, lds, st);
        }
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // extern "C"
