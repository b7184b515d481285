/* CPU restatement of the brute-force L2 descriptor matcher -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Follows modules/cudafeatures2d/src/cuda/bf_match.cu:92-136 (loopUnrolledCached: for every (query, train) pair the distance is
 * accumulated over k ascending, descriptors zero-padded to 64 / 128 elements; L2Dist of the main repository's
 * opencv2/core/cuda/vec_distance.hpp -- un-vendored, restated: reduceIter: reg = a - b; sum += reg * reg (one fma under nvcc's
 * default contraction); result sqrtf(sum)), the strict-< best update in ascending train order (:127-133) and
 * cuda/bf_knnmatch.cu's two-best update for k = 2.  Exact ties go to the lowest train index (cv::BFMatcher's CPU rule; the
 * reference's cross-thread reduction prefers the lowest index modulo 16 first).  parity unpinned (no fixture in the reference's
 * tests: test_features2d.cpp generates random descriptors at run time).
 */
#include <float.h>
#include <math.h>
#include <stddef.h>

/* query nq x d, train nt x d (dense rows); mask NULL or nq x nt bytes.  idx/dist: nq x 2 ({best, second}). */
int orc_bf_knn2(const float *query, int nq, const float *train, int nt, int d, const unsigned char *mask, int *idx, float *dist)
{
    if (nq <= 0 || nt <= 0 || d <= 0 || d > 128) return -1;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < nq; ++q) {
        float b1 = FLT_MAX, b2 = FLT_MAX;
        int i1 = -1, i2 = -1;
        const float *qr = query + (size_t)q * d;
        for (int t = 0; t < nt; ++t) {
            if (mask && !mask[(size_t)q * nt + t]) continue;
            const float *tr = train + (size_t)t * d;
            float sum = 0.f;
            for (int k = 0; k < d; ++k) {
                const float reg = qr[k] - tr[k];
                sum = fmaf(reg, reg, sum);
            }
            const float dv = sqrtf(sum);
            if (dv < b1) { b2 = b1; i2 = i1; b1 = dv; i1 = t; }
            else if (dv < b2) { b2 = dv; i2 = t; }
        }
        idx[2 * q] = i1; idx[2 * q + 1] = i2;
        dist[2 * q] = b1; dist[2 * q + 1] = b2;
    }
    return 0;
}
