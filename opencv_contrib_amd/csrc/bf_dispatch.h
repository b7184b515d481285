// Internal: the integer-descriptor matcher (bfint_kernels.hip) behind the mi_bf_* entry points of bfmatch_kernels.hip.
#pragma once
#include "mi_common.h"

namespace mi {
namespace bfint {

// outputs: n_q x k (knn) / n_q x cols (radius) matrices given as base pointers and per-query element strides; img may be null
int knn(int norm, const mi_mat *query, const mi_mat *trains, const mi_mat *masks, int n_trains, int k, int *idx, size_t istep, int *img,
        size_t mstep, float *dist, size_t dstep, hipStream_t st);
int radius(int norm, const mi_mat *query, const mi_mat *trains, const mi_mat *masks, int n_trains, float max_distance, int cols, int *idx,
           size_t istep, int *img, size_t mstep, float *dist, size_t dstep, int *n_matches, hipStream_t st);

}  // namespace bfint
}  // namespace mi
