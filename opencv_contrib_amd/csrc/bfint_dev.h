// Brute-force matcher for INTEGER descriptors: NORM_L1 on CV_8U / 16U / 16S / 32S, NORM_HAMMING on CV_8U / 16U / 32S -- the rest of
// the reference's (depth, norm) table (cudafeatures2d/src/brute_force_matcher.cpp:336-356; kernels cuda/bf_match.cu, bf_knnmatch.cu,
// bf_radius_match.cu with L1Dist<int types> = __sad and HammingDist = __popc of the un-vendored vec_distance.hpp).
// One wave per 64 queries (lane = query); queries transposed and the train tile widened to int32 in LDS; a lane walks the tile's
// rows in train order and keeps its k best in LDS columns, so the strict-< scan of the reference is literally what runs.  Images of a
// collection are separate launches that continue the lists held in the output matrices.  Phase-structured like surfcpu_dev.h: the
// host build (tests/cpp/bfint_emul.cpp) is held bit for bit to oracle/bfmatch_ref.c.  Integer arithmetic: results are exact.
#pragma once
#include <cfloat>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MI_HD __host__ __device__
#else
#define MI_HD
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define MI_FOR_TID(T) for (int tid = (int)threadIdx.x, mi_once_ = 1; mi_once_; mi_once_ = 0)
#define MI_BARRIER() __syncthreads()
#else
#define MI_FOR_TID(T) for (int tid = 0; tid < (T); ++tid)
#define MI_BARRIER() ((void)0)
#endif

namespace mi {
namespace bfint {

constexpr int T = 64;          // queries per workgroup
constexpr int TT = 32;         // train rows per tile
constexpr int MAX_D = 128;     // descriptor elements (ORB 32, BRISK / FREAK 64, AKAZE 61)
constexpr int MAX_K = 16;
constexpr int NORM_L1 = 2, NORM_HAMMING = 6;

struct Desc { const void *p; long long step; int rows, cols, depth; };      // depth: 0 = 8U, 2 = 16U, 3 = 16S, 4 = 32S (cv depths)
struct Mask { const unsigned char *p; long long step; };                    // p == nullptr: none
struct Lists { int *idx; long long istep; int *img; long long mstep; float *dist; long long dstep; };   // n_q x k, element strides; img may be null

struct Shared {
    int q[MAX_D][T];
    int t[TT][MAX_D];
    float ld[MAX_K][T];
    int li[MAX_K][T], lm[MAX_K][T];
};

MI_HD inline int element(const Desc &D, int r, int c)
{
    const unsigned char *row = (const unsigned char *)D.p + (long long)r * D.step;
    switch (D.depth) {
    case 0: return row[c];
    case 2: return ((const unsigned short *)row)[c];
    case 3: return ((const short *)row)[c];
    default: return ((const int *)row)[c];
    }
}

MI_HD inline int popcount32(unsigned v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}

MI_HD inline float distance(const Shared &sm, int lane, int r, int d, int norm)
{
    unsigned sum = 0;
    if (norm == NORM_HAMMING) for (int k = 0; k < d; ++k) sum += (unsigned)popcount32((unsigned)(sm.q[k][lane] ^ sm.t[r][k]));
    else for (int k = 0; k < d; ++k) { const long long v = (long long)sm.q[k][lane] - sm.t[r][k]; sum += (unsigned)(v < 0 ? -v : v); }
    return (float)sum;
}

MI_HD inline void stage_queries(const Desc &Q, int q0, int d, Shared &sm)
{
    MI_FOR_TID(T) {
        for (int e = tid; e < d * T; e += T) {
            const int ql = e / d, k = e % d;
            const int row = q0 + ql < Q.rows ? q0 + ql : Q.rows - 1;
            sm.q[k][ql] = element(Q, row, k);
        }
    }
}

MI_HD inline void stage_tile(const Desc &Tr, int tb, int d, Shared &sm)
{
    MI_FOR_TID(T) {
        for (int e = tid; e < TT * d; e += T) {
            const int r = e / d, k = e % d;
            const int row = tb + r < Tr.rows ? tb + r : Tr.rows - 1;
            sm.t[r][k] = element(Tr, row, k);
        }
    }
}

// One image: continues the k-lists of queries [q0, q0 + 64) (first != 0: starts them empty).  List entries are ordered by
// (distance, image, train index) because candidates arrive in that order and only a strictly smaller distance moves ahead.
MI_HD inline void knn_block(const Desc &Q, const Desc &Tr, const Mask &M, int q0, int norm, int k, int image, int first, const Lists &L, Shared &sm)
{
    const int d = Q.cols;
    stage_queries(Q, q0, d, sm);
    MI_FOR_TID(T) {
        const int qi = q0 + tid;
        for (int j = 0; j < k; ++j) {
            const bool have = !first && qi < Q.rows && L.idx[(long long)qi * L.istep + j] >= 0;
            sm.ld[j][tid] = have ? L.dist[(long long)qi * L.dstep + j] : FLT_MAX;
            sm.li[j][tid] = have ? L.idx[(long long)qi * L.istep + j] : -1;
            sm.lm[j][tid] = have ? (L.img ? L.img[(long long)qi * L.mstep + j] : 0) : -1;
        }
    }
    MI_BARRIER();
    for (int tb = 0; tb < Tr.rows; tb += TT) {
        stage_tile(Tr, tb, d, sm);
        MI_BARRIER();
        MI_FOR_TID(T) {
            const int qi = q0 + tid;
            const int nrow = Tr.rows - tb < TT ? Tr.rows - tb : TT;
            if (qi < Q.rows) {
                for (int r = 0; r < nrow; ++r) {
                    const int ti = tb + r;
                    if (M.p && !M.p[(long long)qi * M.step + ti]) continue;
                    const float dv = distance(sm, tid, r, d, norm);
                    int pos = k;
                    while (pos > 0 && dv < sm.ld[pos - 1][tid]) --pos;
                    if (pos == k) continue;
                    for (int j = k - 1; j > pos; --j) { sm.ld[j][tid] = sm.ld[j - 1][tid]; sm.li[j][tid] = sm.li[j - 1][tid]; sm.lm[j][tid] = sm.lm[j - 1][tid]; }
                    sm.ld[pos][tid] = dv; sm.li[pos][tid] = ti; sm.lm[pos][tid] = image;
                }
            }
        }
        MI_BARRIER();
    }
    MI_FOR_TID(T) {
        const int qi = q0 + tid;
        if (qi < Q.rows)
            for (int j = 0; j < k; ++j) {
                L.idx[(long long)qi * L.istep + j] = sm.li[j][tid];
                L.dist[(long long)qi * L.dstep + j] = sm.ld[j][tid];
                if (L.img) L.img[(long long)qi * L.mstep + j] = sm.lm[j][tid];
            }
    }
}

// radiusMatch for one image: hits in train order continue at n_matches[q] (first != 0: from 0); every hit is counted, the first
// `cols` of a row are stored (bf_radius_match.cu:104-116).
MI_HD inline void radius_block(const Desc &Q, const Desc &Tr, const Mask &M, int q0, int norm, float max_dist, int cols, int image, int first,
                               const Lists &L, int *n_matches, Shared &sm)
{
    const int d = Q.cols;
    stage_queries(Q, q0, d, sm);
    MI_FOR_TID(T) { sm.li[0][tid] = (!first && q0 + tid < Q.rows) ? n_matches[q0 + tid] : 0; }
    MI_BARRIER();
    for (int tb = 0; tb < Tr.rows; tb += TT) {
        stage_tile(Tr, tb, d, sm);
        MI_BARRIER();
        MI_FOR_TID(T) {
            const int qi = q0 + tid;
            const int nrow = Tr.rows - tb < TT ? Tr.rows - tb : TT;
            if (qi < Q.rows) {
                int n = sm.li[0][tid];
                for (int r = 0; r < nrow; ++r) {
                    const int ti = tb + r;
                    if (M.p && !M.p[(long long)qi * M.step + ti]) continue;
                    const float dv = distance(sm, tid, r, d, norm);
                    if (!(dv < max_dist)) continue;
                    if (n < cols) {
                        L.idx[(long long)qi * L.istep + n] = ti;
                        L.dist[(long long)qi * L.dstep + n] = dv;
                        if (L.img) L.img[(long long)qi * L.mstep + n] = image;
                    }
                    ++n;
                }
                sm.li[0][tid] = n;
            }
        }
        MI_BARRIER();
    }
    MI_FOR_TID(T) { if (q0 + tid < Q.rows) n_matches[q0 + tid] = sm.li[0][tid]; }
}

}  // namespace bfint
}  // namespace mi
