// Integer-descriptor brute-force matcher: kernels and validation (see bfint_dev.h for the design and the reference pointers).
#include "bf_dispatch.h"
#include "bfint_dev.h"

namespace mi {
namespace bfint {

__global__ __launch_bounds__(T) void k_knn(Desc Q, Desc Tr, Mask M, int norm, int k, int image, int first, Lists L)
{
    __shared__ Shared sm;
    knn_block(Q, Tr, M, blockIdx.x * T, norm, k, image, first, L, sm);
}

__global__ __launch_bounds__(T) void k_radius(Desc Q, Desc Tr, Mask M, int norm, float max_dist, int cols, int image, int first, Lists L, int *n_matches)
{
    __shared__ Shared sm;
    radius_block(Q, Tr, M, blockIdx.x * T, norm, max_dist, cols, image, first, L, n_matches, sm);
}

// the reference's (depth, norm) table, brute_force_matcher.cpp:336-356
static int check(int norm, const mi_mat *query, const mi_mat *trains, const mi_mat *masks, int n_trains)
{
    MI_REQUIRE(query && trains && query->data && n_trains > 0, MI_ERR_BAD_ARG, "null argument");
    const int ty = query->type;
    const bool ok = (norm == NORM_L1 && (ty == MI_8UC1 || ty == MI_16UC1 || ty == MI_16SC1 || ty == MI_32SC1)) ||
                    (norm == NORM_HAMMING && (ty == MI_8UC1 || ty == MI_16UC1 || ty == MI_32SC1));
    MI_REQUIRE(ok, MI_ERR_BAD_TYPE, "unsupported combination of query.depth() and norm");
    MI_REQUIRE(query->rows > 0 && query->cols > 0, MI_ERR_BAD_SIZE, "empty query");
    MI_REQUIRE(query->cols <= MAX_D, MI_ERR_BAD_SIZE, "integer descriptors of up to 128 elements; longer ones are not built");
    for (int m = 0; m < n_trains; ++m) {
        MI_REQUIRE(trains[m].data && trains[m].type == ty, MI_ERR_BAD_TYPE, "train.type() == query.type()");
        MI_REQUIRE(trains[m].rows > 0 && trains[m].cols == query->cols, MI_ERR_BAD_SIZE, "query.cols == train.cols, non-empty");
        if (masks && masks[m].data)
            MI_REQUIRE(masks[m].type == MI_8UC1 && masks[m].rows == query->rows && masks[m].cols == trains[m].rows, MI_ERR_BAD_SIZE,
                       "mask must be CV_8UC1, query.rows x train.rows");
    }
    return MI_OK;
}

static Desc desc_of(const mi_mat &m) { return Desc{m.data, (long long)m.step, m.rows, m.cols, m.type}; }      // single channel: type == depth
static Mask mask_of(const mi_mat *masks, int m) { return masks && masks[m].data ? Mask{(const unsigned char *)masks[m].data, (long long)masks[m].step} : Mask{nullptr, 0}; }

int knn(int norm, const mi_mat *query, const mi_mat *trains, const mi_mat *masks, int n_trains, int k, int *idx, size_t istep, int *img,
        size_t mstep, float *dist, size_t dstep, hipStream_t st)
{
    if (int rc = check(norm, query, trains, masks, n_trains)) return rc;
    MI_REQUIRE(k >= 1 && k <= MAX_K, MI_ERR_NOT_IMPL, "integer descriptors: 1 <= k <= 16");
    MI_REQUIRE(n_trains == 1 || img, MI_ERR_BAD_ARG, "a collection needs img_idx");
    const Lists L = {idx, (long long)istep, img, (long long)mstep, dist, (long long)dstep};
    const int blocks = div_up(query->rows, T);
    for (int m = 0; m < n_trains; ++m)
        hipLaunchKernelGGL(k_knn, dim3(blocks), dim3(T), 0, st, desc_of(*query), desc_of(trains[m]), mask_of(masks, m), norm, k, m, m == 0, L);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int radius(int norm, const mi_mat *query, const mi_mat *trains, const mi_mat *masks, int n_trains, float max_distance, int cols, int *idx,
           size_t istep, int *img, size_t mstep, float *dist, size_t dstep, int *n_matches, hipStream_t st)
{
    if (int rc = check(norm, query, trains, masks, n_trains)) return rc;
    MI_REQUIRE(n_trains == 1 || img, MI_ERR_BAD_ARG, "a collection needs img_idx");
    const Lists L = {idx, (long long)istep, img, (long long)mstep, dist, (long long)dstep};
    const int blocks = div_up(query->rows, T);
    for (int m = 0; m < n_trains; ++m)
        hipLaunchKernelGGL(k_radius, dim3(blocks), dim3(T), 0, st, desc_of(*query), desc_of(trains[m]), mask_of(masks, m), norm, max_distance, cols,
                           m, m == 0, L, n_matches);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace bfint
}  // namespace mi
