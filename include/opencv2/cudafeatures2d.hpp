// cv::cuda::DescriptorMatcher (brute force, NORM_L2, float descriptors) over libmiflow (first part).
// Interface subset of modules/cudafeatures2d/include/opencv2/cudafeatures2d.hpp: createBFMatcher, match, knnMatch (k = 2);
// the matching itself is mi_bf_match / mi_bf_knn_match2 (include/miflow/c_api.h).
#ifndef OPENCV_CUDAFEATURES2D_MIFLOW_HPP
#define OPENCV_CUDAFEATURES2D_MIFLOW_HPP

#include <vector>
#include "opencv2/core/cuda.hpp"

namespace cv {

#ifndef MIFLOW_HAVE_DMATCH
#define MIFLOW_HAVE_DMATCH
/** opencv2/core/types.hpp DMatch */
struct DMatch {
    DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.402823466e+38f) {}
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
    int queryIdx, trainIdx, imgIdx;
    float distance;
    bool operator<(const DMatch &m) const { return distance < m.distance; }
};
#endif

namespace cuda {

class CV_EXPORTS_W DescriptorMatcher : public cv::Algorithm {
public:
    static Ptr<DescriptorMatcher> createBFMatcher(int normType = 4 /* cv::NORM_L2 */);
    virtual bool isMaskSupported() const = 0;
    /** queries without an allowed candidate are skipped, like the reference's matchConvert (brute_force_matcher.cpp) */
    virtual void match(InputArray queryDescriptors, InputArray trainDescriptors, std::vector<DMatch> &matches,
                       InputArray mask = GpuMat()) = 0;
    virtual void knnMatch(InputArray queryDescriptors, InputArray trainDescriptors, std::vector<std::vector<DMatch> > &matches, int k,
                          InputArray mask = GpuMat(), bool compactResult = false) = 0;
};

namespace miflow_detail {
class BFMatcherImpl final : public DescriptorMatcher {
public:
    explicit BFMatcherImpl(int normType) { miCheck(mi_bf_create(normType, &h_)); }
    ~BFMatcherImpl() override { mi_bf_destroy(h_); }
    BFMatcherImpl(const BFMatcherImpl &) = delete;
    BFMatcherImpl &operator=(const BFMatcherImpl &) = delete;
    bool isMaskSupported() const override { return true; }
    void match(InputArray query, InputArray train, std::vector<DMatch> &matches, InputArray mask) override
    {
        matches.clear();
        if (query.empty() || train.empty()) return;
        const int nq = query.rows;
        GpuMat idx(1, nq, CV_32SC1), dist(1, nq, CV_32FC1);
        mi_mat q = miMat(query), t = miMat(train), m = miMat(mask), i = miMat(idx), d = miMat(dist);
        miCheck(mi_bf_match(h_, &q, &t, mask.empty() ? nullptr : &m, &i, &d, nullptr));
        std::vector<int> hi(nq);
        std::vector<float> hd(nq);
        idx.download(hi.data(), sizeof(int) * nq);
        dist.download(hd.data(), sizeof(float) * nq);
        for (int k = 0; k < nq; ++k)
            if (hi[k] >= 0) matches.push_back(DMatch(k, hi[k], hd[k]));
    }
    void knnMatch(InputArray query, InputArray train, std::vector<std::vector<DMatch> > &matches, int k, InputArray mask,
                  bool compactResult) override
    {
        CV_Assert(k == 2);   // the ratio-test form; other k are not built
        matches.clear();
        if (query.empty() || train.empty()) return;
        const int nq = query.rows;
        GpuMat idx(1, nq, CV_MAKETYPE(CV_32S, 2)), dist(1, nq, CV_32FC2);
        mi_mat q = miMat(query), t = miMat(train), m = miMat(mask), i = miMat(idx), d = miMat(dist);
        miCheck(mi_bf_knn_match2(h_, &q, &t, mask.empty() ? nullptr : &m, &i, &d, nullptr));
        std::vector<int> hi(2 * nq);
        std::vector<float> hd(2 * nq);
        idx.download(hi.data(), sizeof(int) * 2 * nq);
        dist.download(hd.data(), sizeof(float) * 2 * nq);
        for (int r = 0; r < nq; ++r) {
            std::vector<DMatch> row;
            for (int j = 0; j < 2; ++j)
                if (hi[2 * r + j] >= 0) row.push_back(DMatch(r, hi[2 * r + j], hd[2 * r + j]));
            if (!compactResult || !row.empty()) matches.push_back(row);
        }
    }

private:
    mi_bfmatcher *h_ = nullptr;
};
}  // namespace miflow_detail

inline Ptr<DescriptorMatcher> DescriptorMatcher::createBFMatcher(int normType) { return makePtr<miflow_detail::BFMatcherImpl>(normType); }

}  // namespace cuda
}  // namespace cv
#endif
