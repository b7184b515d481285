// Brute-force L2 matcher for float descriptors (SURF output), gfx950.  Replaces cudafeatures2d/src/cuda/bf_match.cu:92-183 and
// bf_knnmatch.cu (k = 2) behind the C-ABI.
//
// Layout: a lane owns one QUERY; the 64 queries of a wave sit transposed in LDS (conflict-free lane-contiguous reads) and the
// train set streams through LDS in tiles of 32 descriptors read as broadcasts, four train descriptors at a time = four
// independent fma chains per lane.  The train range is cut into contiguous splits over blockIdx.y so that a few thousand
// queries still fill 256 CUs; a second kernel merges the per-split (best, second best) in split order, which reproduces the
// strict-< scan over ascending train indices.  Distances follow the reference's summation order exactly (k ascending,
// sum = fma(d, d, sum), then sqrtf), so the HIP result is bit-identical to oracle/bfmatch_ref.c.
#include "mi_common.h"
#include <cfloat>
#include <utility>

struct mi_bfmatcher {
    int norm = MI_NORM_L2;
    float *part = nullptr;      // [split][query] {d1, d2} + indices
    size_t part_bytes = 0;
};

namespace mi {
namespace bf {

constexpr int TT = 32;   // train descriptors per LDS tile

struct Args {
    const float *q; size_t qstep;      // bytes
    const float *t; size_t tstep;
    const unsigned char *mask; size_t mstep;
    int nq, nt, d;
    int rows_per_split;
    float4 *part;                      // [split * nq + query] = {d1, as_float(i1), d2, as_float(i2)}
};

// One wave per workgroup: 64 queries (lane = query) transposed in LDS (s_q[k][lane]: lane-contiguous, conflict-free), the train
// tile next to it; runtime loops with a 4 x 4 (train x k) body, so no large register arrays (a lane-resident query of 64-128
// floats made the compiler hoist every LDS read of the unrolled chain and spill).
template <int D>
__global__ __launch_bounds__(64) void k_match(Args A)
{
    __shared__ __attribute__((aligned(16))) float s_q[D * 64];
    __shared__ __attribute__((aligned(16))) float s_t[TT * D];
    const int lane = threadIdx.x;
    const int qi = blockIdx.x * 64 + lane;
    const int split = blockIdx.y;
    const int t0 = split * A.rows_per_split, t1 = min(t0 + A.rows_per_split, A.nt);
    for (int e = lane; e < D * 64; e += 64) {   // e = ql * D + k would stride the global reads; read rows, write transposed
        const int ql = e / D, k = e % D;
        const int qrow = min(blockIdx.x * 64 + ql, A.nq - 1);
        // zero padding like loadQueryToSmem, bf_match.cu:81-90
        s_q[k * 64 + ql] = k < A.d ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(A.q) + (size_t)qrow * A.qstep)[k] : 0.f;
    }
    float b1 = FLT_MAX, b2 = FLT_MAX;
    int i1 = -1, i2 = -1;
#pragma unroll 1
    for (int tb = t0; tb < t1; tb += TT) {
        __syncthreads();
        for (int e = lane; e < TT * D; e += 64) {
            const int r = e / D, k = e % D;
            const int tr = min(tb + r, A.nt - 1);
            s_t[e] = k < A.d ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(A.t) + (size_t)tr * A.tstep)[k] : 0.f;
        }
        __syncthreads();
        const int nrow = min(TT, t1 - tb);
#pragma unroll 1
        for (int r = 0; r < nrow; r += 4) {
            // four train descriptors at a time = four independent chains sum = fma(d, d, sum), k ascending (bf_match.cu:100-121)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            const float *p0 = s_t + r * D;
#pragma unroll 2
            for (int k = 0; k < D; k += 4) {
                const float q0 = s_q[k * 64 + lane], q1 = s_q[(k + 1) * 64 + lane], q2 = s_q[(k + 2) * 64 + lane], q3 = s_q[(k + 3) * 64 + lane];
                const float4 a = *reinterpret_cast<const float4 *>(p0 + k), b = *reinterpret_cast<const float4 *>(p0 + D + k);
                const float4 c = *reinterpret_cast<const float4 *>(p0 + 2 * D + k), d = *reinterpret_cast<const float4 *>(p0 + 3 * D + k);
                float e;
                e = q0 - a.x; s0 = fmaf(e, e, s0); e = q1 - a.y; s0 = fmaf(e, e, s0); e = q2 - a.z; s0 = fmaf(e, e, s0); e = q3 - a.w; s0 = fmaf(e, e, s0);
                e = q0 - b.x; s1 = fmaf(e, e, s1); e = q1 - b.y; s1 = fmaf(e, e, s1); e = q2 - b.z; s1 = fmaf(e, e, s1); e = q3 - b.w; s1 = fmaf(e, e, s1);
                e = q0 - c.x; s2 = fmaf(e, e, s2); e = q1 - c.y; s2 = fmaf(e, e, s2); e = q2 - c.z; s2 = fmaf(e, e, s2); e = q3 - c.w; s2 = fmaf(e, e, s2);
                e = q0 - d.x; s3 = fmaf(e, e, s3); e = q1 - d.y; s3 = fmaf(e, e, s3); e = q2 - d.z; s3 = fmaf(e, e, s3); e = q3 - d.w; s3 = fmaf(e, e, s3);
            }
            const float dv[4] = {sqrtf(s0), sqrtf(s1), sqrtf(s2), sqrtf(s3)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ti = tb + r + j;
                bool ok = (r + j < nrow) && qi < A.nq;
                if (ok && A.mask) ok = A.mask[(size_t)qi * A.mstep + ti] != 0;
                // bf_knnmatch.cu: if (d < best1) { best2 = best1; best1 = d } else if (d < best2) best2 = d   (train order)
                if (ok && dv[j] < b1) { b2 = b1; i2 = i1; b1 = dv[j]; i1 = ti; }
                else if (ok && dv[j] < b2) { b2 = dv[j]; i2 = ti; }
            }
        }
    }
    if (qi < A.nq) A.part[(size_t)split * A.nq + qi] = make_float4(b1, __int_as_float(i1), b2, __int_as_float(i2));
}

// merges the splits in ascending train order; KNN = 0: writes (idx, dist), KNN = 1: writes pairs
template <int KNN>
__global__ __launch_bounds__(256) void k_merge(const float4 *part, int nq, int nsplit, int *idx, float *dist)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    float b1 = FLT_MAX, b2 = FLT_MAX;
    int i1 = -1, i2 = -1;
    for (int s = 0; s < nsplit; ++s) {
        const float4 p = part[(size_t)s * nq + qi];
        const float c[2] = {p.x, p.z};
        const int ci[2] = {__float_as_int(p.y), __float_as_int(p.w)};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (ci[j] < 0) continue;
            if (c[j] < b1) { b2 = b1; i2 = i1; b1 = c[j]; i1 = ci[j]; }
            else if (c[j] < b2) { b2 = c[j]; i2 = ci[j]; }
        }
    }
    if (KNN) { idx[2 * qi] = i1; idx[2 * qi + 1] = i2; dist[2 * qi] = b1; dist[2 * qi + 1] = b2; }
    else { idx[qi] = i1; dist[qi] = b1; }
}

static int run(mi_bfmatcher *h, const mi_mat *query, const mi_mat *train, const mi_mat *mask, mi_mat *train_idx, mi_mat *distance,
               int knn, hipStream_t st)
{
    MI_REQUIRE(h && query && train && train_idx && distance && query->data && train->data && train_idx->data && distance->data,
               MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(query->type == MI_32FC1 && train->type == MI_32FC1, MI_ERR_BAD_TYPE, "descriptors must be CV_32FC1");
    MI_REQUIRE(query->rows > 0 && train->rows > 0 && query->cols == train->cols, MI_ERR_BAD_SIZE, "query.cols == train.cols, non-empty");
    MI_REQUIRE(query->cols <= 128, MI_ERR_BAD_SIZE, "descriptor length <= 128 (SURF: 64 / 128); longer descriptors are not built");
    const int nq = query->rows, nt = train->rows;
    MI_REQUIRE(train_idx->type == (knn ? MI_32SC2 : MI_32SC1) && distance->type == (knn ? MI_32FC2 : MI_32FC1), MI_ERR_BAD_TYPE,
               "train_idx CV_32SC1 / distance CV_32FC1 (k = 2: CV_32SC2 / CV_32FC2)");
    MI_REQUIRE(train_idx->rows == 1 && distance->rows == 1 && train_idx->cols == nq && distance->cols == nq, MI_ERR_BAD_SIZE,
               "outputs must be 1 x query.rows");
    if (mask) MI_REQUIRE(mask->data && mask->type == MI_8UC1 && mask->rows == nq && mask->cols == nt, MI_ERR_BAD_SIZE,
                         "mask must be CV_8UC1, query.rows x train.rows");
    // enough workgroups for 256 CUs: splits of the train range (multiples of the LDS tile)
    const int qblocks = div_up(nq, 64);
    int nsplit = min(max(1, 2048 / qblocks), div_up(nt, TT));
    const int rows_per_split = align_up(div_up(nt, nsplit), TT);
    nsplit = div_up(nt, rows_per_split);
    const size_t need = sizeof(float4) * (size_t)nsplit * nq;
    if (h->part_bytes < need) {
        if (h->part) { (void)hipFree(h->part); h->part = nullptr; h->part_bytes = 0; }
        MI_HIP_TRY(hipMalloc(&h->part, need));
        h->part_bytes = need;
    }
    Args A;
    A.q = (const float *)query->data; A.qstep = query->step;
    A.t = (const float *)train->data; A.tstep = train->step;
    A.mask = mask ? (const unsigned char *)mask->data : nullptr; A.mstep = mask ? mask->step : 0;
    A.nq = nq; A.nt = nt; A.d = query->cols; A.rows_per_split = rows_per_split;
    A.part = reinterpret_cast<float4 *>(h->part);
    const dim3 grid(qblocks, nsplit);
    if (query->cols <= 64) hipLaunchKernelGGL((k_match<64>), grid, dim3(64), 0, st, A);      // matchDispatcher, bf_match.cu:563-570
    else hipLaunchKernelGGL((k_match<128>), grid, dim3(64), 0, st, A);
    const int mblocks = div_up(nq, 256);
    if (knn) hipLaunchKernelGGL((k_merge<1>), dim3(mblocks), dim3(256), 0, st, A.part, nq, nsplit, (int *)train_idx->data, (float *)distance->data);
    else hipLaunchKernelGGL((k_merge<0>), dim3(mblocks), dim3(256), 0, st, A.part, nq, nsplit, (int *)train_idx->data, (float *)distance->data);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace bf
}  // namespace mi

using namespace mi;

extern "C" {

int mi_bf_create(int norm_type, mi_bfmatcher **out)
{
    MI_REQUIRE(out, MI_ERR_BAD_ARG, "null out");
    *out = nullptr;
    MI_REQUIRE(norm_type == MI_NORM_L2, MI_ERR_BAD_ARG, "only NORM_L2 (float descriptors) is built");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device available: the miflow product path has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    *out = new mi_bfmatcher();
    return MI_OK;
}

void mi_bf_destroy(mi_bfmatcher *h)
{
    if (!h) return;
    if (h->part) (void)hipFree(h->part);
    delete h;
}

int mi_bf_match(mi_bfmatcher *h, const mi_mat *query, const mi_mat *train, const mi_mat *mask, mi_mat *train_idx, mi_mat *distance, void *stream)
{
    return bf::run(h, query, train, mask, train_idx, distance, 0, (hipStream_t)stream);
}

int mi_bf_knn_match2(mi_bfmatcher *h, const mi_mat *query, const mi_mat *train, const mi_mat *mask, mi_mat *train_idx, mi_mat *distance,
                     void *stream)
{
    return bf::run(h, query, train, mask, train_idx, distance, 1, (hipStream_t)stream);
}

}  // extern "C"
