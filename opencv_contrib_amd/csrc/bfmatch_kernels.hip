// Brute-force matcher for float descriptors (SURF output), NORM_L1 / NORM_L2, gfx950.  Replaces cudafeatures2d/src/cuda/bf_match.cu,
// bf_knnmatch.cu and bf_radius_match.cu behind the C-ABI (match, knnMatch for any k, radiusMatch; single train set or a collection).
//
// Layout: a lane owns one QUERY; the 64 queries of a workgroup sit transposed in LDS (conflict-free lane-contiguous reads) and the
// train set streams through LDS in tiles read as broadcasts, four train descriptors at a time = four independent accumulation
// chains per lane.  W waves of a workgroup share the queries and the tile and take TT / W rows of every tile each.  The train
// range is cut into contiguous splits over blockIdx.y so that a few thousand queries still fill 256 CUs.  Every lane keeps its K
// best candidates as a list sorted by (distance, index) -- the order the reference's strict-< scan over ascending train indices
// produces -- so waves, splits and images can be merged in any order; a second kernel merges the per-split lists.  k > K runs
// further passes that only admit candidates after the last entry found so far (no distance matrix).  Distances follow the
// reference's summation order exactly (k ascending; L2: sum = fma(d, d, sum) then sqrtf; L1: sum += |d|), so the HIP result is
// bit-identical to oracle/bfmatch_ref.c.
#include "mi_common.h"
#include "bf_dispatch.h"
#include <cfloat>
#include <climits>
#include <cstdlib>
#include <vector>

struct mi_bfmatcher {
    int norm = MI_NORM_L2;
    void *part = nullptr;       // knn: [segment][query][K] {distance, index}; radius: [segment][query] count -> offset
    size_t part_bytes = 0;
};

namespace mi {
namespace bf {

struct Args {
    const float *q; size_t qstep;      // bytes
    const float *t; size_t tstep;
    const unsigned char *mask; size_t mstep;
    int nq, nt, d;
    int rows_per_split;
    int t_base;                        // index of this train set's row 0 in the concatenated collection
    int seg0;                          // first segment (split) of this train set in `part` / `cnt`
    float2 *part;                      // knn: [(seg0 + split) * nq + query][K] = {distance, as_float(collection index)}
    // knn continuation pass (k > K): only candidates after (lo_dist, lo_idx) in (distance, index) order; per-query element strides
    const float *lo_dist; const int *lo_idx; size_t lo_dstep, lo_istep;
    // radius
    int *cnt;                          // [(seg0 + split) * nq + query]: hits of the segment (count pass) / first output slot (write pass)
    float max_dist;
    int write, cols;
    int *r_idx, *r_img; float *r_dist; size_t r_istep, r_mstep, r_dstep;   // element strides per query
};

// K best (distance, index) pairs of a lane, ascending; empty slots are (FLT_MAX, INT_MAX).  push() keeps the K smallest pairs in
// lexicographic order, which is what "if (d < best1) {...} else if (d < best2) {...}" (bf_knnmatch.cu:353-371) yields when the
// candidates arrive in ascending index order, and does not depend on the arrival order.
template <int K>
struct KBest {
    float d[K];
    int i[K];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int j = 0; j < K; ++j) { d[j] = FLT_MAX; i[j] = INT_MAX; }
    }
    __device__ __forceinline__ void push(float dv, int ti, bool ok)
    {
        ok = ok && dv < FLT_MAX && (dv < d[K - 1] || (dv == d[K - 1] && ti < i[K - 1]));
        if (!__any(ok)) return;          // wave-uniform: after the first tiles almost no candidate enters the list
        if (ok) { d[K - 1] = dv; i[K - 1] = ti; }
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            const bool sw = d[j] < d[j - 1] || (d[j] == d[j - 1] && i[j] < i[j - 1]);
            const float dl = sw ? d[j] : d[j - 1], dh = sw ? d[j - 1] : d[j];
            const int il = sw ? i[j] : i[j - 1], ih = sw ? i[j - 1] : i[j];
            d[j - 1] = dl; d[j] = dh; i[j - 1] = il; i[j] = ih;
        }
    }
};

// 64 queries of the workgroup, transposed (s_q[k][query]); zero padding like loadQueryToSmem, bf_match.cu:81-90
template <int D, int W>
__device__ __forceinline__ void stage_queries(const Args &A, float *s_q)
{
    for (int e = threadIdx.x; e < D * 64; e += 64 * W) {   // e = ql * D + k: global reads run along the rows
        const int ql = e / D, k = e % D;
        const int qrow = min((int)blockIdx.x * 64 + ql, A.nq - 1);
        s_q[k * 64 + ql] = k < A.d ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(A.q) + (size_t)qrow * A.qstep)[k] : 0.f;
    }
}

template <int D, int TT, int W>
__device__ __forceinline__ void stage_tile(const Args &A, int tb, float *s_t)
{
    for (int e = threadIdx.x; e < TT * D; e += 64 * W) {
        const int r = e / D, k = e % D;
        const int tr = min(tb + r, A.nt - 1);
        s_t[e] = k < A.d ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(A.t) + (size_t)tr * A.tstep)[k] : 0.f;
    }
}

template <int NORM>
__device__ __forceinline__ void acc(float &s, float q, float t)
{
    const float e = q - t;
    if (NORM == MI_NORM_L2) s = fmaf(e, e, s);      // L2Dist::reduceIter, one fma
    else s = s + fabsf(e);                          // L1Dist::reduceIter
}

// distances of this lane's query to the four train descriptors at p0: four independent k-ascending chains (bf_match.cu:100-121)
template <int D, int NORM>
__device__ __forceinline__ void dist4(const float *s_q, const float *p0, int lane, float (&dv)[4])
{
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 2
    for (int k = 0; k < D; k += 4) {
        const float q0 = s_q[k * 64 + lane], q1 = s_q[(k + 1) * 64 + lane], q2 = s_q[(k + 2) * 64 + lane], q3 = s_q[(k + 3) * 64 + lane];
        const float4 a = *reinterpret_cast<const float4 *>(p0 + k), b = *reinterpret_cast<const float4 *>(p0 + D + k);
        const float4 c = *reinterpret_cast<const float4 *>(p0 + 2 * D + k), d = *reinterpret_cast<const float4 *>(p0 + 3 * D + k);
        acc<NORM>(s0, q0, a.x); acc<NORM>(s0, q1, a.y); acc<NORM>(s0, q2, a.z); acc<NORM>(s0, q3, a.w);
        acc<NORM>(s1, q0, b.x); acc<NORM>(s1, q1, b.y); acc<NORM>(s1, q2, b.z); acc<NORM>(s1, q3, b.w);
        acc<NORM>(s2, q0, c.x); acc<NORM>(s2, q1, c.y); acc<NORM>(s2, q2, c.z); acc<NORM>(s2, q3, c.w);
        acc<NORM>(s3, q0, d.x); acc<NORM>(s3, q1, d.y); acc<NORM>(s3, q2, d.z); acc<NORM>(s3, q3, d.w);
    }
    if (NORM == MI_NORM_L2) { dv[0] = sqrtf(s0); dv[1] = sqrtf(s1); dv[2] = sqrtf(s2); dv[3] = sqrtf(s3); }
    else { dv[0] = s0; dv[1] = s1; dv[2] = s2; dv[3] = s3; }
}

// W waves per workgroup, 64 queries (lane = query), runtime loops with a 4 x 4 (train x k) body, so no large register arrays (a
// lane-resident query of 64-128 floats made the compiler hoist every LDS read of the unrolled chain and spill).
template <int D, int TT, int W, int NORM, int K>
__global__ __launch_bounds__(64 * W) void k_knn(Args A)
{
    static_assert(TT % (4 * W) == 0 && D % 4 == 0 && D >= 2 * W * K, "tile / scratch shape");
    __shared__ __attribute__((aligned(16))) float s_q[D * 64];
    __shared__ __attribute__((aligned(16))) float s_t[TT * D];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int qi = blockIdx.x * 64 + lane;
    const int split = blockIdx.y;
    const int t0 = split * A.rows_per_split, t1 = min(t0 + A.rows_per_split, A.nt);
    stage_queries<D, W>(A, s_q);
    float lo_d = -1.f;
    int lo_i = -1;
    if (A.lo_idx && qi < A.nq) {
        lo_i = A.lo_idx[(size_t)qi * A.lo_istep];
        lo_d = lo_i < 0 ? FLT_MAX : A.lo_dist[(size_t)qi * A.lo_dstep];     // list already exhausted: admit nothing
    }
    KBest<K> best;
    best.init();
    constexpr int RW = TT / W;      // tile rows per wave
#pragma unroll 1
    for (int tb = t0; tb < t1; tb += TT) {
        __syncthreads();
        stage_tile<D, TT, W>(A, tb, s_t);
        __syncthreads();
        const int nrow = min(TT, t1 - tb);
        const int rend = min((wv + 1) * RW, nrow);
#pragma unroll 1
        for (int r = wv * RW; r < rend; r += 4) {
            float dv[4];
            dist4<D, NORM>(s_q, s_t + r * D, lane, dv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tl = tb + r + j;
                bool ok = (r + j < nrow) && qi < A.nq;
                if (ok && A.mask) ok = A.mask[(size_t)qi * A.mstep + tl] != 0;
                const int ti = A.t_base + tl;
                ok = ok && (dv[j] > lo_d || (dv[j] == lo_d && ti > lo_i));
                best.push(dv[j], ti, ok);
            }
        }
    }
    if (W > 1) {                     // fold the other waves' lists into wave 0's through the (now free) query area
        __syncthreads();
        float *sc = s_q;             // [W][K][2][64]
        if (wv > 0) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                sc[((wv * K + j) * 2 + 0) * 64 + lane] = best.d[j];
                sc[((wv * K + j) * 2 + 1) * 64 + lane] = __int_as_float(best.i[j]);
            }
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll 1
            for (int w = 1; w < W; ++w) {
#pragma unroll
                for (int j = 0; j < K; ++j)
                    best.push(sc[((w * K + j) * 2 + 0) * 64 + lane], __float_as_int(sc[((w * K + j) * 2 + 1) * 64 + lane]), true);
            }
        }
    }
    if (wv == 0 && qi < A.nq) {
        float2 *o = A.part + ((size_t)(A.seg0 + split) * A.nq + qi) * K;
#pragma unroll
        for (int j = 0; j < K; ++j) o[j] = make_float2(best.d[j], __int_as_float(best.i[j]));
    }
}

// merges the segment lists of every query; writes columns [col0, col0 + ncol) of the n_q x k outputs (element strides per query)
template <int K>
__global__ __launch_bounds__(256) void k_merge(const float2 *part, int nq, int nseg, int *idx, size_t istep, float *dist, size_t dstep,
                                               int *img, size_t mstep, int col0, int ncol)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    KBest<K> b;
    b.init();
    for (int s = 0; s < nseg; ++s) {
        const float2 *p = part + ((size_t)s * nq + qi) * K;
#pragma unroll
        for (int j = 0; j < K; ++j) b.push(p[j].x, __float_as_int(p[j].y), true);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j >= ncol) break;
        const bool none = b.i[j] == INT_MAX;
        idx[(size_t)qi * istep + col0 + j] = none ? -1 : b.i[j];
        dist[(size_t)qi * dstep + col0 + j] = b.d[j];
        if (img) img[(size_t)qi * mstep + col0 + j] = none ? -1 : -2;      // -2: collection index not yet resolved
    }
}

// Collection index -> (image, train index).  Launched for m = n_trains - 1 ... 0: an unresolved entry >= off belongs to image m.
__global__ __launch_bounds__(256) void k_assign_image(int *idx, size_t istep, int *img, size_t mstep, int nq, int ncols,
                                                      const int *nvalid, int off, int m)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nq * ncols) return;
    const int q = e / ncols, j = e % ncols;
    if (nvalid && j >= min(nvalid[q], ncols)) return;
    int *pi = idx + (size_t)q * istep + j, *pm = img + (size_t)q * mstep + j;
    if (*pm == -2 && *pi >= off) { *pm = m; *pi -= off; }
}

// radiusMatch, one wave per workgroup.  Pass 0 (write = 0) counts the hits of every (segment, query); k_radius_scan turns the counts
// into output slots; pass 1 recomputes the distances and stores the hits, so a row lists them in ascending collection order.
template <int D, int TT, int NORM>
__global__ __launch_bounds__(64) void k_radius(Args A)
{
    __shared__ __attribute__((aligned(16))) float s_q[D * 64];
    __shared__ __attribute__((aligned(16))) float s_t[TT * D];
    const int lane = threadIdx.x;
    const int qi = blockIdx.x * 64 + lane;
    const int split = blockIdx.y;
    const int t0 = split * A.rows_per_split, t1 = min(t0 + A.rows_per_split, A.nt);
    stage_queries<D, 1>(A, s_q);
    const size_t slot = (size_t)(A.seg0 + split) * A.nq + min(qi, A.nq - 1);
    const int base = A.write ? A.cnt[slot] : 0;
    int n = 0;
#pragma unroll 1
    for (int tb = t0; tb < t1; tb += TT) {
        __syncthreads();
        stage_tile<D, TT, 1>(A, tb, s_t);
        __syncthreads();
        const int nrow = min(TT, t1 - tb);
#pragma unroll 1
        for (int r = 0; r < nrow; r += 4) {
            float dv[4];
            dist4<D, NORM>(s_q, s_t + r * D, lane, dv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tl = tb + r + j;
                bool ok = (r + j < nrow) && qi < A.nq;
                if (ok && A.mask) ok = A.mask[(size_t)qi * A.mstep + tl] != 0;
                ok = ok && dv[j] < A.max_dist;                       // bf_radius_match.cu:106
                if (ok) {
                    const int pos = base + n;
                    if (A.write && pos < A.cols) {
                        A.r_idx[(size_t)qi * A.r_istep + pos] = A.t_base + tl;
                        A.r_dist[(size_t)qi * A.r_dstep + pos] = dv[j];
                        if (A.r_img) A.r_img[(size_t)qi * A.r_mstep + pos] = -2;
                    }
                    ++n;
                }
            }
        }
    }
    if (!A.write && qi < A.nq) A.cnt[slot] = n;
}

__global__ __launch_bounds__(256) void k_radius_scan(int *cnt, int nq, int nseg, int *n_matches)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    int run = 0;
    for (int s = 0; s < nseg; ++s) {
        const int c = cnt[(size_t)s * nq + qi];
        cnt[(size_t)s * nq + qi] = run;
        run += c;
    }
    n_matches[qi] = run;             // every hit is counted, like the reference's atomicInc (may exceed cols)
}

// ------------------------------------------------------------------------------------------------ host side
typedef void (*launch_t)(const Args &, dim3, hipStream_t);

template <int D, int TT, int W, int NORM, int K>
static void launch_knn(const Args &A, dim3 grid, hipStream_t st)
{
    hipLaunchKernelGGL((k_knn<D, TT, W, NORM, K>), grid, dim3(64 * W), 0, st, A);
}

template <int D, int TT, int NORM>
static void launch_radius(const Args &A, dim3 grid, hipStream_t st)
{
    hipLaunchKernelGGL((k_radius<D, TT, NORM>), grid, dim3(64), 0, st, A);
}

template <int D, int TT, int W>
static launch_t pick_knn(int norm, int K)
{
    if (norm == MI_NORM_L2) return K == 2 ? launch_knn<D, TT, W, MI_NORM_L2, 2> : launch_knn<D, TT, W, MI_NORM_L2, 8>;
    return K == 2 ? launch_knn<D, TT, W, MI_NORM_L1, 2> : launch_knn<D, TT, W, MI_NORM_L1, 8>;
}

template <int D, int TT>
static launch_t pick_radius(int norm)
{
    return norm == MI_NORM_L2 ? launch_radius<D, TT, MI_NORM_L2> : launch_radius<D, TT, MI_NORM_L1>;
}

struct Shape { launch_t fn; int tt, w; };

// matchDispatcher (bf_match.cu:563-570) by descriptor length; long descriptors trade tile height for the query area
static Shape knn_shape(int d, int norm, int K)
{
    // four waves per workgroup measured 2.8-3.5 x faster than one (profiles/r01y: 12 564^2 pairs, D = 64: 0.69 vs 1.95 ms);
    // MIFLOW_BF_W=1 keeps the single-wave shape reachable (read per call: tests and sweeps switch it inside one process)
    const char *e = getenv("MIFLOW_BF_W");
    const int w_small = e && atoi(e) == 1 ? 1 : 4;
    if (d <= 64) return w_small == 4 ? Shape{pick_knn<64, 32, 4>(norm, K), 32, 4} : Shape{pick_knn<64, 32, 1>(norm, K), 32, 1};
    if (d <= 128) return w_small == 4 ? Shape{pick_knn<128, 32, 4>(norm, K), 32, 4} : Shape{pick_knn<128, 32, 1>(norm, K), 32, 1};
    if (d <= 256) return Shape{pick_knn<256, 16, 4>(norm, K), 16, 4};
    return Shape{pick_knn<512, 8, 2>(norm, K), 8, 2};
}

static Shape radius_shape(int d, int norm)
{
    if (d <= 64) return Shape{pick_radius<64, 32>(norm), 32, 1};
    if (d <= 128) return Shape{pick_radius<128, 32>(norm), 32, 1};
    if (d <= 256) return Shape{pick_radius<256, 16>(norm), 16, 1};
    return Shape{pick_radius<512, 8>(norm), 8, 1};
}

struct Seg { int nsplit, rows_per_split, seg0, t_base; };

// validates (query, trains[], masks[]) and plans the splits: enough workgroups for 256 CUs, multiples of the LDS tile
static int plan(const mi_mat *query, const mi_mat *trains, const mi_mat *masks, int n_trains, int tt, std::vector<Seg> &segs, int *nseg)
{
    MI_REQUIRE(query && trains && query->data && n_trains > 0, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(query->type == MI_32FC1, MI_ERR_BAD_TYPE, "descriptors must be CV_32FC1");
    MI_REQUIRE(query->rows > 0 && query->cols > 0, MI_ERR_BAD_SIZE, "empty query");
    MI_REQUIRE(query->cols <= 512, MI_ERR_BAD_SIZE, "descriptor length <= 512; longer descriptors are not built");
    const int qblocks = div_up(query->rows, 64);
    long long total = 0;
    int s0 = 0;
    segs.resize(n_trains);
    for (int m = 0; m < n_trains; ++m) {
        const mi_mat &t = trains[m];
        MI_REQUIRE(t.data && t.type == MI_32FC1, MI_ERR_BAD_TYPE, "descriptors must be CV_32FC1");
        MI_REQUIRE(t.rows > 0 && t.cols == query->cols, MI_ERR_BAD_SIZE, "query.cols == train.cols, non-empty");
        if (masks && masks[m].data)
            MI_REQUIRE(masks[m].type == MI_8UC1 && masks[m].rows == query->rows && masks[m].cols == t.rows, MI_ERR_BAD_SIZE,
                       "mask must be CV_8UC1, query.rows x train.rows");
        int nsplit = min(max(1, 2048 / qblocks), div_up(t.rows, tt));
        const int rps = align_up(div_up(t.rows, nsplit), tt);
        nsplit = div_up(t.rows, rps);
        segs[m] = Seg{nsplit, rps, s0, (int)total};
        s0 += nsplit;
        total += t.rows;
        MI_REQUIRE(total < INT_MAX, MI_ERR_BAD_SIZE, "train collection too large");
    }
    *nseg = s0;
    return MI_OK;
}

static int reserve(mi_bfmatcher *h, size_t need)
{
    if (h->part_bytes >= need) return MI_OK;
    if (h->part) { (void)hipFree(h->part); h->part = nullptr; h->part_bytes = 0; }
    MI_HIP_TRY(hipMalloc(&h->part, need));
    h->part_bytes = need;
    return MI_OK;
}

static void fill_args(Args &A, const mi_mat *query, const mi_mat &t, const mi_mat *mask, const Seg &s)
{
    A.q = (const float *)query->data; A.qstep = query->step;
    A.t = (const float *)t.data; A.tstep = t.step;
    A.mask = mask && mask->data ? (const unsigned char *)mask->data : nullptr; A.mstep = A.mask ? mask->step : 0;
    A.nq = query->rows; A.nt = t.rows; A.d = query->cols;
    A.rows_per_split = s.rows_per_split; A.t_base = s.t_base; A.seg0 = s.seg0;
}

struct Out { int *idx; size_t istep; int *img; size_t mstep; float *dist; size_t dstep; };   // element strides per query

static void assign_images(const Out &o, int nq, int ncols, const int *nvalid, const std::vector<Seg> &segs, hipStream_t st)
{
    const int blocks = div_up(nq * ncols, 256);
    for (int m = (int)segs.size() - 1; m >= 0; --m)
        hipLaunchKernelGGL(k_assign_image, dim3(blocks), dim3(256), 0, st, o.idx, o.istep, o.img, o.mstep, nq, ncols, nvalid, segs[m].t_base, m);
}

static int run_knn(mi_bfmatcher *h, const mi_mat *query, const mi_mat *trains, const mi_mat *masks, int n_trains, int k, const Out &o,
                   hipStream_t st)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    MI_REQUIRE(k >= 1, MI_ERR_BAD_ARG, "k >= 1");
    if (query && query->type != MI_32FC1)          // integer descriptors: the other rows of the reference's (depth, norm) table
        return bfint::knn(h->norm, query, trains, masks, n_trains, k, o.idx, o.istep, o.img, o.mstep, o.dist, o.dstep, st);
    MI_REQUIRE(h->norm == MI_NORM_L1 || h->norm == MI_NORM_L2, MI_ERR_BAD_TYPE, "unsupported combination of query.depth() and norm");
    MI_REQUIRE(n_trains == 1 || o.img, MI_ERR_BAD_ARG, "a collection needs img_idx");
    const int K = k <= 2 ? 2 : 8;
    const Shape sh = knn_shape(query ? query->cols : 0, h->norm, K);
    std::vector<Seg> segs;
    int nseg = 0;
    if (int rc = plan(query, trains, masks, n_trains, sh.tt, segs, &nseg)) return rc;
    const int nq = query->rows;
    if (int rc = reserve(h, sizeof(float2) * (size_t)nseg * nq * K)) return rc;
    const int qblocks = div_up(nq, 64), mblocks = div_up(nq, 256);
    for (int col0 = 0; col0 < k; col0 += K) {
        for (int m = 0; m < n_trains; ++m) {
            Args A = {};
            fill_args(A, query, trains[m], masks ? &masks[m] : nullptr, segs[m]);
            A.part = reinterpret_cast<float2 *>(h->part);
            if (col0) { A.lo_idx = o.idx + col0 - 1; A.lo_istep = o.istep; A.lo_dist = o.dist + col0 - 1; A.lo_dstep = o.dstep; }
            sh.fn(A, dim3(qblocks, segs[m].nsplit), st);
        }
        const int ncol = min(K, k - col0);
        if (K == 2) hipLaunchKernelGGL((k_merge<2>), dim3(mblocks), dim3(256), 0, st, reinterpret_cast<const float2 *>(h->part), nq, nseg,
                                       o.idx, o.istep, o.dist, o.dstep, o.img, o.mstep, col0, ncol);
        else hipLaunchKernelGGL((k_merge<8>), dim3(mblocks), dim3(256), 0, st, reinterpret_cast<const float2 *>(h->part), nq, nseg,
                                o.idx, o.istep, o.dist, o.dstep, o.img, o.mstep, col0, ncol);
    }
    if (o.img) assign_images(o, nq, k, nullptr, segs, st);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

static bool is_rows(const mi_mat *m, int type, int rows, int cols)
{
    return m && m->data && m->type == type && m->rows == rows && m->cols == cols;
}

}  // namespace bf
}  // namespace mi

using namespace mi;

extern "C" {

int mi_bf_create(int norm_type, mi_bfmatcher **out)
{
    MI_REQUIRE(out, MI_ERR_BAD_ARG, "null out");
    *out = nullptr;
    MI_REQUIRE(norm_type == MI_NORM_L2 || norm_type == MI_NORM_L1 || norm_type == MI_NORM_HAMMING, MI_ERR_BAD_ARG,
               "norm == NORM_L1 || norm == NORM_L2 || norm == NORM_HAMMING");      // BFMatcher_Impl constructor
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device available: the miflow product path has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    *out = new mi_bfmatcher();
    (*out)->norm = norm_type;
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
    MI_REQUIRE(query && train, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(train_idx && distance && train_idx->data && distance->data, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(train_idx->type == MI_32SC1 && distance->type == MI_32FC1, MI_ERR_BAD_TYPE, "train_idx CV_32SC1 / distance CV_32FC1");
    MI_REQUIRE(bf::is_rows(train_idx, MI_32SC1, 1, query->rows) && bf::is_rows(distance, MI_32FC1, 1, query->rows), MI_ERR_BAD_SIZE,
               "outputs must be 1 x query.rows");
    const bf::Out o = {(int *)train_idx->data, 1, nullptr, 0, (float *)distance->data, 1};
    return bf::run_knn(h, query, train, mask, 1, 1, o, (hipStream_t)stream);
}

int mi_bf_knn_match2(mi_bfmatcher *h, const mi_mat *query, const mi_mat *train, const mi_mat *mask, mi_mat *train_idx, mi_mat *distance,
                     void *stream)
{
    MI_REQUIRE(query && train, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(train_idx && distance && train_idx->data && distance->data, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(train_idx->type == MI_32SC2 && distance->type == MI_32FC2, MI_ERR_BAD_TYPE, "k = 2: train_idx CV_32SC2 / distance CV_32FC2");
    MI_REQUIRE(bf::is_rows(train_idx, MI_32SC2, 1, query->rows) && bf::is_rows(distance, MI_32FC2, 1, query->rows), MI_ERR_BAD_SIZE,
               "outputs must be 1 x query.rows");
    const bf::Out o = {(int *)train_idx->data, 2, nullptr, 0, (float *)distance->data, 2};
    return bf::run_knn(h, query, train, mask, 1, 2, o, (hipStream_t)stream);
}

int mi_bf_knn_match(mi_bfmatcher *h, const mi_mat *query, const mi_mat *trains, const mi_mat *masks, int n_trains, int k, mi_mat *train_idx,
                    mi_mat *img_idx, mi_mat *distance, void *stream)
{
    MI_REQUIRE(query && trains && train_idx && distance, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(k >= 1, MI_ERR_BAD_ARG, "k >= 1");
    MI_REQUIRE(bf::is_rows(train_idx, MI_32SC1, query->rows, k) && bf::is_rows(distance, MI_32FC1, query->rows, k) &&
                   (!img_idx || bf::is_rows(img_idx, MI_32SC1, query->rows, k)),
               MI_ERR_BAD_SIZE, "train_idx / img_idx CV_32SC1 and distance CV_32FC1 must be query.rows x k");
    const bf::Out o = {(int *)train_idx->data, train_idx->step / 4, img_idx ? (int *)img_idx->data : nullptr, img_idx ? img_idx->step / 4 : 0,
                       (float *)distance->data, distance->step / 4};
    return bf::run_knn(h, query, trains, masks, n_trains, k, o, (hipStream_t)stream);
}

int mi_bf_radius_match(mi_bfmatcher *h, const mi_mat *query, const mi_mat *trains, const mi_mat *masks, int n_trains, float max_distance,
                       mi_mat *train_idx, mi_mat *img_idx, mi_mat *distance, mi_mat *n_matches, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    MI_REQUIRE(h && query && trains && train_idx && distance && n_matches, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(n_trains == 1 || img_idx, MI_ERR_BAD_ARG, "a collection needs img_idx");
    const int cols = train_idx->cols;
    MI_REQUIRE(cols > 0 && bf::is_rows(train_idx, MI_32SC1, query->rows, cols) && bf::is_rows(distance, MI_32FC1, query->rows, cols) &&
                   (!img_idx || bf::is_rows(img_idx, MI_32SC1, query->rows, cols)) && bf::is_rows(n_matches, MI_32SC1, 1, query->rows),
               MI_ERR_BAD_SIZE, "train_idx / img_idx CV_32SC1, distance CV_32FC1: query.rows x cols; n_matches CV_32SC1 1 x query.rows");
    if (query->type != MI_32FC1)
        return bfint::radius(h->norm, query, trains, masks, n_trains, max_distance, cols, (int *)train_idx->data, train_idx->step / 4,
                             img_idx ? (int *)img_idx->data : nullptr, img_idx ? img_idx->step / 4 : 0, (float *)distance->data,
                             distance->step / 4, (int *)n_matches->data, st);
    MI_REQUIRE(h->norm == MI_NORM_L1 || h->norm == MI_NORM_L2, MI_ERR_BAD_TYPE, "unsupported combination of query.depth() and norm");
    const bf::Shape sh = bf::radius_shape(query->cols, h->norm);
    std::vector<bf::Seg> segs;
    int nseg = 0;
    if (int rc = bf::plan(query, trains, masks, n_trains, sh.tt, segs, &nseg)) return rc;
    const int nq = query->rows;
    if (int rc = bf::reserve(h, sizeof(int) * (size_t)nseg * nq)) return rc;
    const bf::Out o = {(int *)train_idx->data, train_idx->step / 4, img_idx ? (int *)img_idx->data : nullptr, img_idx ? img_idx->step / 4 : 0,
                       (float *)distance->data, distance->step / 4};
    const int qblocks = div_up(nq, 64);
    for (int pass = 0; pass < 2; ++pass) {
        for (int m = 0; m < n_trains; ++m) {
            bf::Args A = {};
            bf::fill_args(A, query, trains[m], masks ? &masks[m] : nullptr, segs[m]);
            A.cnt = reinterpret_cast<int *>(h->part);
            A.max_dist = max_distance; A.write = pass; A.cols = cols;
            A.r_idx = o.idx; A.r_istep = o.istep; A.r_img = o.img; A.r_mstep = o.mstep; A.r_dist = o.dist; A.r_dstep = o.dstep;
            sh.fn(A, dim3(qblocks, segs[m].nsplit), st);
        }
        if (pass == 0)
            hipLaunchKernelGGL(bf::k_radius_scan, dim3(div_up(nq, 256)), dim3(256), 0, st, reinterpret_cast<int *>(h->part), nq, nseg,
                               (int *)n_matches->data);
    }
    if (o.img) bf::assign_images(o, nq, cols, (const int *)n_matches->data, segs, st);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // extern "C"
