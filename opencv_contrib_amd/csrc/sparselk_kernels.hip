// cv::cuda::SparsePyrLKOpticalFlow (CV_8UC1 frames): handle, 8-bit pyramids and the per-point kernel.  Host-side twin of
// PyrLKOpticalFlowBase::sparse / buildImagePyramid (modules/cudaoptflow/src/pyrlk.cpp:134-231); the device logic lives in
// sparselk_dev.h (phases shared with a host build that tests hold bit for bit to oracle/pyrlk_ref.c).  One wave per point and level:
// the window (<= 32 x 32) is spread over the 64 lanes, patch and derivatives stay in LDS across the Newton steps.
#include "mi_common.h"
#include "sparselk_dev.h"
#include <vector>

struct mi_sparsepyrlk {
    mi_sparsepyrlk_params P;
    unsigned char *buf = nullptr;       // levels 1 .. max_level of both frames, dense rows
    size_t buf_bytes = 0;
};

namespace mi {
namespace slk {

__global__ __launch_bounds__(256) void k_pyr_down_u8(Image S, unsigned char *dst, int dw, int dh)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < dw && y < dh) dst[(size_t)y * dw + x] = pyr_down_pixel(S, y, x);
}

// nextPts = (useInitialFlow ? nextPts : prevPts) * (1 / 2^maxLevel / 2) (cuda::multiply with a double scale, pyrlk.cpp:167-169); status = 1
__global__ __launch_bounds__(256) void k_prepare(const float *prev, float *next, unsigned char *status, int n, double sc, int use_next)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *src = use_next ? next : prev;
    const float x = (float)(src[2 * i] * sc), y = (float)(src[2 * i + 1] * sc);
    next[2 * i] = x; next[2 * i + 1] = y;
    status[i] = 1;
}

__global__ __launch_bounds__(T) void k_sparse(Image I, Image J, const float *prev, float *next, unsigned char *status, float *err, int level,
                                              int wx, int wy, int iters)
{
    __shared__ Shared sm;
    const int i = blockIdx.x;
    point_block(I, J, prev[2 * i], prev[2 * i + 1], next + 2 * i, level, wx, wy, iters, status + i, err ? err + i : nullptr, sm);
}

}  // namespace slk
}  // namespace mi

using namespace mi;

extern "C" {

void mi_sparsepyrlk_default_params(mi_sparsepyrlk_params *p)
{
    if (!p) return;
    p->win_width = 21; p->win_height = 21; p->max_level = 3; p->iters = 30; p->use_initial_flow = 0;   // cudaoptflow.hpp:218-222
}

static int validate(const mi_sparsepyrlk_params *p)
{
    MI_REQUIRE(p, MI_ERR_BAD_ARG, "null params");
    MI_REQUIRE(p->max_level >= 0 && p->max_level < 16, MI_ERR_BAD_ARG, "maxLevel >= 0");                      // pyrlk.cpp:156
    MI_REQUIRE(p->win_width > 2 && p->win_height > 2, MI_ERR_BAD_ARG, "winSize.width > 2 && winSize.height > 2");   // :157
    MI_REQUIRE(p->win_width * p->win_height <= slk::MAX_K * slk::T, MI_ERR_NOT_IMPL, "windows of more than 1024 pixels are not built");
    MI_REQUIRE(p->iters >= 0, MI_ERR_BAD_ARG, "iters >= 0");
    return MI_OK;
}

int mi_sparsepyrlk_create(const mi_sparsepyrlk_params *p, mi_sparsepyrlk **out)
{
    MI_REQUIRE(out, MI_ERR_BAD_ARG, "null out");
    *out = nullptr;
    mi_sparsepyrlk_params d;
    if (!p) { mi_sparsepyrlk_default_params(&d); p = &d; }
    if (int rc = validate(p)) return rc;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device available: the miflow product path has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    *out = new mi_sparsepyrlk();
    (*out)->P = *p;
    return MI_OK;
}

int mi_sparsepyrlk_set_params(mi_sparsepyrlk *h, const mi_sparsepyrlk_params *p)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    if (int rc = validate(p)) return rc;
    h->P = *p;
    return MI_OK;
}

int mi_sparsepyrlk_get_params(const mi_sparsepyrlk *h, mi_sparsepyrlk_params *p)
{
    MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument");
    *p = h->P;
    return MI_OK;
}

void mi_sparsepyrlk_destroy(mi_sparsepyrlk *h)
{
    if (!h) return;
    if (h->buf) (void)hipFree(h->buf);
    delete h;
}

int mi_sparsepyrlk_calc(mi_sparsepyrlk *h, const mi_mat *prev_img, const mi_mat *next_img, const mi_mat *prev_pts, mi_mat *next_pts,
                        mi_mat *status, mi_mat *err, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    MI_REQUIRE(h && prev_img && next_img && prev_pts && next_pts && status, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(prev_img->data && next_img->data && prev_img->type == MI_8UC1 && next_img->type == MI_8UC1, MI_ERR_BAD_TYPE,
               "frames must be CV_8UC1 (the other depths / channel counts of pyrlk.cpp:190-198 are not built)");
    MI_REQUIRE(prev_img->rows == next_img->rows && prev_img->cols == next_img->cols && prev_img->rows > 0 && prev_img->cols > 0,
               MI_ERR_BAD_SIZE, "prevImg.size() == nextImg.size()");                                              // :218
    const int n = prev_pts->cols;
    MI_REQUIRE(prev_pts->data && prev_pts->rows == 1 && prev_pts->type == MI_32FC2 && n > 0, MI_ERR_BAD_ARG,
               "prevPts must be a non-empty 1 x N CV_32FC2 row");                                                  // :155
    MI_REQUIRE(next_pts->data && next_pts->rows == 1 && next_pts->cols == n && next_pts->type == MI_32FC2, MI_ERR_BAD_SIZE,
               "nextPts must be 1 x N CV_32FC2");                                                                  // :159-161
    MI_REQUIRE(status->data && status->rows == 1 && status->cols == n && status->type == MI_8UC1, MI_ERR_BAD_SIZE, "status must be 1 x N CV_8UC1");
    if (err) MI_REQUIRE(err->data && err->rows == 1 && err->cols == n && err->type == MI_32FC1, MI_ERR_BAD_SIZE, "err must be 1 x N CV_32FC1");
    const mi_sparsepyrlk_params &P = h->P;
    // pyramid geometry (cuda::pyrDown: (w + 1) / 2) and scratch
    std::vector<int> pw(P.max_level + 1), ph(P.max_level + 1);
    std::vector<size_t> off(P.max_level + 1, 0);
    pw[0] = prev_img->cols; ph[0] = prev_img->rows;
    size_t per_frame = 0;
    for (int l = 1; l <= P.max_level; ++l) {
        pw[l] = (pw[l - 1] + 1) / 2; ph[l] = (ph[l - 1] + 1) / 2;
        off[l] = per_frame;
        per_frame += ((size_t)pw[l] * ph[l] + 255) / 256 * 256;
    }
    if (h->buf_bytes < 2 * per_frame) {
        if (h->buf) { (void)hipFree(h->buf); h->buf = nullptr; h->buf_bytes = 0; }
        if (per_frame) MI_HIP_TRY(hipMalloc((void **)&h->buf, 2 * per_frame));
        h->buf_bytes = 2 * per_frame;
    }
    std::vector<slk::Image> I(P.max_level + 1), J(P.max_level + 1);
    I[0] = slk::Image{(const unsigned char *)prev_img->data, (long long)prev_img->step, ph[0], pw[0]};
    J[0] = slk::Image{(const unsigned char *)next_img->data, (long long)next_img->step, ph[0], pw[0]};
    for (int l = 1; l <= P.max_level; ++l) {
        unsigned char *dp = h->buf + off[l], *dn = h->buf + per_frame + off[l];
        const dim3 grid(div_up(pw[l], 64), div_up(ph[l], 4));
        hipLaunchKernelGGL(slk::k_pyr_down_u8, grid, dim3(256), 0, st, I[l - 1], dp, pw[l], ph[l]);
        hipLaunchKernelGGL(slk::k_pyr_down_u8, grid, dim3(256), 0, st, J[l - 1], dn, pw[l], ph[l]);
        I[l] = slk::Image{dp, (long long)pw[l], ph[l], pw[l]};
        J[l] = slk::Image{dn, (long long)pw[l], ph[l], pw[l]};
    }
    hipLaunchKernelGGL(slk::k_prepare, dim3(div_up(n, 256)), dim3(256), 0, st, (const float *)prev_pts->data, (float *)next_pts->data,
                       (unsigned char *)status->data, n, 1.0 / (1 << P.max_level) / 2.0, P.use_initial_flow != 0);
    for (int l = P.max_level; l >= 0; --l)
        hipLaunchKernelGGL(slk::k_sparse, dim3(n), dim3(slk::T), 0, st, I[l], J[l], (const float *)prev_pts->data, (float *)next_pts->data,
                           (unsigned char *)status->data, l == 0 && err ? (float *)err->data : nullptr, l, P.win_width, P.win_height, P.iters);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // extern "C"
