/* CPU restatement of the brute-force descriptor matcher (float descriptors, L1 / L2) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Follows modules/cudafeatures2d/src/cuda/bf_match.cu:92-136 (loopUnrolledCached: for every (query, train) pair the distance is
 * accumulated over k ascending, descriptors zero-padded; L1Dist / L2Dist of the main repository's
 * opencv2/core/cuda/vec_distance.hpp -- un-vendored, restated: L1 reduceIter: sum += fabs(a - b); L2: reg = a - b;
 * sum += reg * reg (one fma under nvcc's default contraction), result sqrtf(sum)), the strict-< best update in ascending train
 * order (:127-133), cuda/bf_knnmatch.cu's two-best update for k = 2 (:61-102, 353-371) and, for any other k, its distance
 * matrix + k rounds of minimum extraction (:918-1110: masked pairs become FLT_MAX, a round that finds no value < FLT_MAX writes
 * nothing, so short lists end in trainIdx -1), cuda/bf_radius_match.cu:58-118 (mask && dist < maxDistance; count every hit,
 * store the first `cols`) and src/brute_force_matcher.cpp:296-1070 for the collection forms (images scanned in order).
 * Exact ties go to the lowest (image, train) index -- cv::BFMatcher's CPU rule; the reference's cross-thread reductions /
 * atomicInc make the tie order and the stored subset of an overflowing radius list scheduling dependent, and this is one of
 * its possible outcomes.
 * Pinned on the reference's own known-answer test: tests/test_bfmatch.py restates the generator of
 * cudafeatures2d/test/test_features2d.cpp:274-330 and asserts the expectations of its Match / KnnMatch_2 / KnnMatch_3 /
 * RadiusMatch cases (single image and collection) on this oracle.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>

enum { ORC_NORM_L1 = 2, ORC_NORM_L2 = 4 };   /* cv::NormTypes */

static float bf_distance(const float *qr, const float *tr, int d, int norm)
{
    float sum = 0.f;
    if (norm == ORC_NORM_L1) {
        for (int k = 0; k < d; ++k) sum += fabsf(qr[k] - tr[k]);
        return sum;
    }
    for (int k = 0; k < d; ++k) {
        const float reg = qr[k] - tr[k];
        sum = fmaf(reg, reg, sum);
    }
    return sqrtf(sum);
}

/* k nearest train descriptors of every query over a collection of n_img train sets (n_img = 1: plain match / knnMatch).
 * trains[m]: nts[m] x d dense rows; masks NULL, or masks[m] NULL / nq x nts[m] bytes.
 * idx / img / dist: nq x k; missing entries: idx = img = -1, dist = FLT_MAX.  img may be NULL. */
int orc_bf_knn(const float *query, int nq, const float *const *trains, const int *nts, const unsigned char *const *masks, int n_img,
               int d, int norm, int k, int *idx, int *img, float *dist)
{
    if (nq <= 0 || n_img <= 0 || d <= 0 || k <= 0 || (norm != ORC_NORM_L1 && norm != ORC_NORM_L2)) return -1;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < nq; ++q) {
        int *bi = idx + (size_t)q * k;
        float *bd = dist + (size_t)q * k;
        int *bm = img ? img + (size_t)q * k : NULL;
        for (int j = 0; j < k; ++j) { bi[j] = -1; bd[j] = FLT_MAX; if (bm) bm[j] = -1; }
        const float *qr = query + (size_t)q * d;
        for (int m = 0; m < n_img; ++m) {
            const unsigned char *mk = masks ? masks[m] : NULL;
            for (int t = 0; t < nts[m]; ++t) {
                if (mk && !mk[(size_t)q * nts[m] + t]) continue;
                const float dv = bf_distance(qr, trains[m] + (size_t)t * d, d, norm);
                /* strict < against the current list, scanned in ascending (image, train) order: "if (d < best1) {...} else if
                 * (d < best2) {...}" generalised to k entries */
                int pos = k;
                while (pos > 0 && dv < bd[pos - 1]) --pos;
                if (pos == k) continue;
                for (int j = k - 1; j > pos; --j) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; if (bm) bm[j] = bm[j - 1]; }
                bd[pos] = dv; bi[pos] = t; if (bm) bm[pos] = m;
            }
        }
    }
    return 0;
}

/* query nq x d, train nt x d (dense rows); mask NULL or nq x nt bytes.  idx/dist: nq x 2 ({best, second}). */
int orc_bf_knn2(const float *query, int nq, const float *train, int nt, int d, const unsigned char *mask, int *idx, float *dist)
{
    if (nt <= 0 || d > 128) return -1;
    return orc_bf_knn(query, nq, &train, &nt, mask ? &mask : NULL, 1, d, ORC_NORM_L2, 2, idx, NULL, dist);
}

/* All train descriptors closer than max_dist, in ascending (image, train) order; n[q] counts every hit, the first `cols` are stored
 * (idx / img / dist: nq x cols, entries past min(n[q], cols) are left untouched). */
int orc_bf_radius(const float *query, int nq, const float *const *trains, const int *nts, const unsigned char *const *masks, int n_img,
                  int d, int norm, float max_dist, int cols, int *idx, int *img, float *dist, int *n)
{
    if (nq <= 0 || n_img <= 0 || d <= 0 || cols <= 0 || (norm != ORC_NORM_L1 && norm != ORC_NORM_L2)) return -1;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < nq; ++q) {
        const float *qr = query + (size_t)q * d;
        int cnt = 0;
        for (int m = 0; m < n_img; ++m) {
            const unsigned char *mk = masks ? masks[m] : NULL;
            for (int t = 0; t < nts[m]; ++t) {
                if (mk && !mk[(size_t)q * nts[m] + t]) continue;
                const float dv = bf_distance(qr, trains[m] + (size_t)t * d, d, norm);
                if (!(dv < max_dist)) continue;
                if (cnt < cols) {
                    idx[(size_t)q * cols + cnt] = t;
                    dist[(size_t)q * cols + cnt] = dv;
                    if (img) img[(size_t)q * cols + cnt] = m;
                }
                ++cnt;
            }
        }
        n[q] = cnt;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------------------
 * Integer descriptors (the reference's other (depth, norm) pairs, brute_force_matcher.cpp:336-356: NORM_L1 on CV_8U / 16U / 16S / 32S,
 * NORM_HAMMING on CV_8U / 16U / 32S).  Elements arrive widened to int32 (zero extension for the unsigned depths, which leaves the
 * bits Hamming counts unchanged); L1Dist<int types> accumulates |a - b| in an integer (__sad), HammingDist adds __popc(a ^ b), both
 * are returned as float (vec_distance.hpp of the main repository, restated).  Same list order and radius rule as above. */
enum { ORC_NORM_HAMMING = 6 };

static float bf_distance_int(const int *qr, const int *tr, int d, int norm)
{
    unsigned sum = 0;
    if (norm == ORC_NORM_HAMMING) for (int k = 0; k < d; ++k) sum += (unsigned)__builtin_popcount((unsigned)(qr[k] ^ tr[k]));
    else for (int k = 0; k < d; ++k) { const long long v = (long long)qr[k] - tr[k]; sum += (unsigned)(v < 0 ? -v : v); }
    return (float)sum;
}

int orc_bf_knn_int(const int *query, int nq, const int *const *trains, const int *nts, const unsigned char *const *masks, int n_img,
                   int d, int norm, int k, int *idx, int *img, float *dist)
{
    if (nq <= 0 || n_img <= 0 || d <= 0 || k <= 0 || (norm != ORC_NORM_L1 && norm != ORC_NORM_HAMMING)) return -1;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < nq; ++q) {
        int *bi = idx + (size_t)q * k, *bm = img + (size_t)q * k;
        float *bd = dist + (size_t)q * k;
        for (int j = 0; j < k; ++j) { bi[j] = -1; bd[j] = FLT_MAX; bm[j] = -1; }
        for (int m = 0; m < n_img; ++m) {
            const unsigned char *mk = masks ? masks[m] : NULL;
            for (int t = 0; t < nts[m]; ++t) {
                if (mk && !mk[(size_t)q * nts[m] + t]) continue;
                const float dv = bf_distance_int(query + (size_t)q * d, trains[m] + (size_t)t * d, d, norm);
                int pos = k;
                while (pos > 0 && dv < bd[pos - 1]) --pos;
                if (pos == k) continue;
                for (int j = k - 1; j > pos; --j) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; bm[j] = bm[j - 1]; }
                bd[pos] = dv; bi[pos] = t; bm[pos] = m;
            }
        }
    }
    return 0;
}

int orc_bf_radius_int(const int *query, int nq, const int *const *trains, const int *nts, const unsigned char *const *masks, int n_img,
                      int d, int norm, float max_dist, int cols, int *idx, int *img, float *dist, int *n)
{
    if (nq <= 0 || n_img <= 0 || d <= 0 || cols <= 0 || (norm != ORC_NORM_L1 && norm != ORC_NORM_HAMMING)) return -1;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < nq; ++q) {
        int cnt = 0;
        for (int m = 0; m < n_img; ++m) {
            const unsigned char *mk = masks ? masks[m] : NULL;
            for (int t = 0; t < nts[m]; ++t) {
                if (mk && !mk[(size_t)q * nts[m] + t]) continue;
                const float dv = bf_distance_int(query + (size_t)q * d, trains[m] + (size_t)t * d, d, norm);
                if (!(dv < max_dist)) continue;
                if (cnt < cols) { idx[(size_t)q * cols + cnt] = t; dist[(size_t)q * cols + cnt] = dv; img[(size_t)q * cols + cnt] = m; }
                ++cnt;
            }
        }
        n[q] = cnt;
    }
    return 0;
}
