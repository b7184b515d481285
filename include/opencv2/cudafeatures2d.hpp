// cv::cuda::DescriptorMatcher (brute force, NORM_L1 / NORM_L2, float descriptors) over libmiflow.
// Interface of modules/cudafeatures2d/include/opencv2/cudafeatures2d.hpp:75-372: createBFMatcher, the train collection,
// match / knnMatch / radiusMatch, their *Async forms (the reference's packed device matrices, brute_force_matcher.cpp:314-980)
// and *Convert unpackers (:440-495, 727-812, 982-1063).  The matching itself is mi_bf_knn_match / mi_bf_radius_match
// (include/miflow/c_api.h), which write straight into the rows of the packed matrix.
#ifndef OPENCV_CUDAFEATURES2D_MIFLOW_HPP
#define OPENCV_CUDAFEATURES2D_MIFLOW_HPP

#include <algorithm>
#include <vector>
#include "opencv2/core/cuda.hpp"

namespace cv {

#ifndef MIFLOW_HAVE_DMATCH
#define MIFLOW_HAVE_DMATCH
/** opencv2/core/types.hpp DMatch */
struct DMatch {
    DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.402823466e+38f) {}
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
    DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
    int queryIdx, trainIdx, imgIdx;
    float distance;
    bool operator<(const DMatch &m) const { return distance < m.distance; }
};
#endif
#ifndef MIFLOW_HAVE_NORM_TYPES
#define MIFLOW_HAVE_NORM_TYPES
enum NormTypes { NORM_INF = 1, NORM_L1 = 2, NORM_L2 = 4, NORM_HAMMING = 6 };   // opencv2/core/base.hpp
#endif

namespace cuda {

class CV_EXPORTS_W DescriptorMatcher : public cv::Algorithm {
public:
    static Ptr<DescriptorMatcher> createBFMatcher(int normType = 4 /* cv::NORM_L2 */);
    virtual bool isMaskSupported() const = 0;
    virtual void add(const std::vector<GpuMat> &descriptors) = 0;
    virtual const std::vector<GpuMat> &getTrainDescriptors() const = 0;
    virtual void train() = 0;

    /** queries without an allowed candidate are skipped, like the reference's matchConvert */
    virtual void match(InputArray queryDescriptors, InputArray trainDescriptors, std::vector<DMatch> &matches,
                       InputArray mask = GpuMat()) = 0;
    virtual void match(InputArray queryDescriptors, std::vector<DMatch> &matches,
                       const std::vector<GpuMat> &masks = std::vector<GpuMat>()) = 0;
    virtual void matchAsync(InputArray queryDescriptors, InputArray trainDescriptors, OutputArray matches, InputArray mask = GpuMat(),
                            Stream &stream = Stream::Null()) = 0;
    virtual void matchAsync(InputArray queryDescriptors, OutputArray matches, const std::vector<GpuMat> &masks = std::vector<GpuMat>(),
                            Stream &stream = Stream::Null()) = 0;
    virtual void matchConvert(InputArray gpu_matches, std::vector<DMatch> &matches) = 0;

    virtual void knnMatch(InputArray queryDescriptors, InputArray trainDescriptors, std::vector<std::vector<DMatch> > &matches, int k,
                          InputArray mask = GpuMat(), bool compactResult = false) = 0;
    virtual void knnMatch(InputArray queryDescriptors, std::vector<std::vector<DMatch> > &matches, int k,
                          const std::vector<GpuMat> &masks = std::vector<GpuMat>(), bool compactResult = false) = 0;
    virtual void knnMatchAsync(InputArray queryDescriptors, InputArray trainDescriptors, OutputArray matches, int k,
                               InputArray mask = GpuMat(), Stream &stream = Stream::Null()) = 0;
    virtual void knnMatchAsync(InputArray queryDescriptors, OutputArray matches, int k,
                               const std::vector<GpuMat> &masks = std::vector<GpuMat>(), Stream &stream = Stream::Null()) = 0;
    virtual void knnMatchConvert(InputArray gpu_matches, std::vector<std::vector<DMatch> > &matches, bool compactResult = false) = 0;

    virtual void radiusMatch(InputArray queryDescriptors, InputArray trainDescriptors, std::vector<std::vector<DMatch> > &matches,
                             float maxDistance, InputArray mask = GpuMat(), bool compactResult = false) = 0;
    virtual void radiusMatch(InputArray queryDescriptors, std::vector<std::vector<DMatch> > &matches, float maxDistance,
                             const std::vector<GpuMat> &masks = std::vector<GpuMat>(), bool compactResult = false) = 0;
    virtual void radiusMatchAsync(InputArray queryDescriptors, InputArray trainDescriptors, OutputArray matches, float maxDistance,
                                  InputArray mask = GpuMat(), Stream &stream = Stream::Null()) = 0;
    virtual void radiusMatchAsync(InputArray queryDescriptors, OutputArray matches, float maxDistance,
                                  const std::vector<GpuMat> &masks = std::vector<GpuMat>(), Stream &stream = Stream::Null()) = 0;
    virtual void radiusMatchConvert(InputArray gpu_matches, std::vector<std::vector<DMatch> > &matches, bool compactResult = false) = 0;
};

namespace miflow_detail {

// rows [r0, r0 + n) of a packed CV_32S matrix seen as an n x c matrix of `type`
inline mi_mat packedRows(GpuMat &m, int r0, int n, int c, int type)
{
    mi_mat r;
    r.data = m.ptr<uchar>(r0); r.step = m.step; r.rows = n; r.cols = c; r.type = type;
    return r;
}
// one packed row of nq entries with cn channels, seen as the nq x cn matrix the C-ABI fills
inline mi_mat packedRow(GpuMat &m, int row, int nq, int cn, int type)
{
    mi_mat r;
    r.data = m.ptr<uchar>(row); r.step = (size_t)cn * 4; r.rows = nq; r.cols = cn; r.type = type;
    return r;
}

class BFMatcherImpl final : public DescriptorMatcher {
public:
    explicit BFMatcherImpl(int normType) { miCheck(mi_bf_create(normType, &h_)); }
    ~BFMatcherImpl() override { mi_bf_destroy(h_); }
    BFMatcherImpl(const BFMatcherImpl &) = delete;
    BFMatcherImpl &operator=(const BFMatcherImpl &) = delete;

    bool isMaskSupported() const override { return true; }
    void add(const std::vector<GpuMat> &descriptors) override { coll_.insert(coll_.end(), descriptors.begin(), descriptors.end()); }
    const std::vector<GpuMat> &getTrainDescriptors() const override { return coll_; }
    void clear() override { coll_.clear(); }
    bool empty() const override { return coll_.empty(); }
    void train() override {}

    // ------------------------------------------------------------------ match
    void match(InputArray query, InputArray train, std::vector<DMatch> &matches, InputArray mask) override
    {
        GpuMat g;
        matchAsync(query, train, g, mask, Stream::Null());
        matchConvert(g, matches);
    }
    void match(InputArray query, std::vector<DMatch> &matches, const std::vector<GpuMat> &masks) override
    {
        GpuMat g;
        matchAsync(query, g, masks, Stream::Null());
        matchConvert(g, matches);
    }
    void matchAsync(InputArray query, InputArray train, OutputArray matches, InputArray mask, Stream &stream) override
    {
        if (query.empty() || train.empty()) { matches.release(); return; }
        const int nq = query.rows;
        matches.create(2, nq, CV_32SC1);                                   // {trainIdx; distance}, brute_force_matcher.cpp:367-372
        mi_mat q = miMat(query), t = miMat(train), m = miMat(mask);
        mi_mat i = packedRow(matches, 0, nq, 1, CV_32SC1), d = packedRow(matches, 1, nq, 1, CV_32FC1);
        miCheck(mi_bf_knn_match(h_, &q, &t, mask.empty() ? nullptr : &m, 1, 1, &i, nullptr, &d, stream.hipStream()));
    }
    void matchAsync(InputArray query, OutputArray matches, const std::vector<GpuMat> &masks, Stream &stream) override
    {
        if (query.empty() || coll_.empty()) { matches.release(); return; }
        const int nq = query.rows;
        matches.create(3, nq, CV_32SC1);                                   // {trainIdx; imgIdx; distance}, :429-435
        std::vector<mi_mat> ts, ms;
        collection(masks, ts, ms);
        mi_mat q = miMat(query);
        mi_mat i = packedRow(matches, 0, nq, 1, CV_32SC1), g = packedRow(matches, 1, nq, 1, CV_32SC1), d = packedRow(matches, 2, nq, 1, CV_32FC1);
        miCheck(mi_bf_knn_match(h_, &q, ts.data(), ms.empty() ? nullptr : ms.data(), (int)ts.size(), 1, &i, &g, &d, stream.hipStream()));
    }
    void matchConvert(InputArray gpu_matches, std::vector<DMatch> &matches) override
    {
        matches.clear();
        if (gpu_matches.empty()) return;
        CV_Assert(gpu_matches.type() == CV_32SC1 && (gpu_matches.rows == 2 || gpu_matches.rows == 3));
        const int nq = gpu_matches.cols, nr = gpu_matches.rows;
        std::vector<int> h((size_t)nr * nq);
        gpu_matches.download(h.data(), sizeof(int) * nq);
        const int *idx = h.data(), *img = nr == 3 ? h.data() + nq : nullptr;
        const float *dist = reinterpret_cast<const float *>(h.data() + (size_t)(nr - 1) * nq);
        for (int q = 0; q < nq; ++q)
            if (idx[q] != -1) matches.push_back(DMatch(q, idx[q], img ? img[q] : 0, dist[q]));
    }

    // ------------------------------------------------------------------ knnMatch
    void knnMatch(InputArray query, InputArray train, std::vector<std::vector<DMatch> > &matches, int k, InputArray mask,
                  bool compactResult) override
    {
        GpuMat g;
        knnMatchAsync(query, train, g, k, mask, Stream::Null());
        knnMatchConvert(g, matches, compactResult);
    }
    void knnMatch(InputArray query, std::vector<std::vector<DMatch> > &matches, int k, const std::vector<GpuMat> &masks,
                  bool compactResult) override
    {
        if (k == 2) {
            GpuMat g;
            knnMatchAsync(query, g, k, masks, Stream::Null());
            knnMatchConvert(g, matches, compactResult);
            return;
        }
        // the reference merges per-image lists on the host (:527-571); the C-ABI searches the whole collection in one pass
        matches.clear();
        if (query.empty() || coll_.empty()) return;
        const int nq = query.rows;
        GpuMat g(3 * nq, k, CV_32SC1);
        std::vector<mi_mat> ts, ms;
        collection(masks, ts, ms);
        mi_mat q = miMat(query);
        mi_mat i = packedRows(g, 0, nq, k, CV_32SC1), im = packedRows(g, nq, nq, k, CV_32SC1), d = packedRows(g, 2 * nq, nq, k, CV_32FC1);
        miCheck(mi_bf_knn_match(h_, &q, ts.data(), ms.empty() ? nullptr : ms.data(), (int)ts.size(), k, &i, &im, &d, nullptr));
        std::vector<int> h((size_t)3 * nq * k);
        g.download(h.data(), sizeof(int) * k);
        unpackLists(h.data(), h.data() + (size_t)nq * k, reinterpret_cast<const float *>(h.data() + (size_t)2 * nq * k), nq, k, k, nullptr,
                    compactResult, false, matches);
    }
    void knnMatchAsync(InputArray query, InputArray train, OutputArray matches, int k, InputArray mask, Stream &stream) override
    {
        if (query.empty() || train.empty()) { matches.release(); return; }
        const int nq = query.rows;
        mi_mat q = miMat(query), t = miMat(train), m = miMat(mask), i, d;
        if (k == 2) {
            matches.create(2, nq, CV_MAKETYPE(CV_32S, 2));                 // :619-626
            i = packedRow(matches, 0, nq, 2, CV_32SC1); d = packedRow(matches, 1, nq, 2, CV_32FC1);
        } else {
            matches.create(2 * nq, k, CV_32SC1);                           // :630-636 (no allDist buffer is needed here)
            i = packedRows(matches, 0, nq, k, CV_32SC1); d = packedRows(matches, nq, nq, k, CV_32FC1);
        }
        miCheck(mi_bf_knn_match(h_, &q, &t, mask.empty() ? nullptr : &m, 1, k, &i, nullptr, &d, stream.hipStream()));
    }
    void knnMatchAsync(InputArray query, OutputArray matches, int k, const std::vector<GpuMat> &masks, Stream &stream) override
    {
        if (k != 2) CV_Error(Error::StsNotImplemented, "only k=2 mode is supported for now");   // :664-667
        if (query.empty() || coll_.empty()) { matches.release(); return; }
        const int nq = query.rows;
        matches.create(3, nq, CV_MAKETYPE(CV_32S, 2));                     // {trainIdx; imgIdx; distance} pairs
        std::vector<mi_mat> ts, ms;
        collection(masks, ts, ms);
        mi_mat q = miMat(query);
        mi_mat i = packedRow(matches, 0, nq, 2, CV_32SC1), g = packedRow(matches, 1, nq, 2, CV_32SC1), d = packedRow(matches, 2, nq, 2, CV_32FC1);
        miCheck(mi_bf_knn_match(h_, &q, ts.data(), ms.empty() ? nullptr : ms.data(), (int)ts.size(), 2, &i, &g, &d, stream.hipStream()));
    }
    void knnMatchConvert(InputArray gpu_matches, std::vector<std::vector<DMatch> > &matches, bool compactResult) override
    {
        matches.clear();
        if (gpu_matches.empty()) return;
        const bool pairs = gpu_matches.type() == CV_MAKETYPE(CV_32S, 2);
        CV_Assert((pairs && (gpu_matches.rows == 2 || gpu_matches.rows == 3)) || gpu_matches.type() == CV_32SC1);
        const int cn = pairs ? 2 : 1;
        std::vector<int> h((size_t)gpu_matches.rows * gpu_matches.cols * cn);
        gpu_matches.download(h.data(), sizeof(int) * gpu_matches.cols * cn);
        if (pairs) {
            const int nq = gpu_matches.cols, nr = gpu_matches.rows;
            unpackLists(h.data(), nr == 3 ? h.data() + (size_t)2 * nq : nullptr, reinterpret_cast<const float *>(h.data() + (size_t)(nr - 1) * 2 * nq),
                        nq, 2, 2, nullptr, compactResult, false, matches);
        } else {
            const int nq = gpu_matches.rows / 2, k = gpu_matches.cols;
            unpackLists(h.data(), nullptr, reinterpret_cast<const float *>(h.data() + (size_t)nq * k), nq, k, k, nullptr, compactResult, false, matches);
        }
    }

    // ------------------------------------------------------------------ radiusMatch
    void radiusMatch(InputArray query, InputArray train, std::vector<std::vector<DMatch> > &matches, float maxDistance, InputArray mask,
                     bool compactResult) override
    {
        GpuMat g;
        radiusMatchAsync(query, train, g, maxDistance, mask, Stream::Null());
        radiusMatchConvert(g, matches, compactResult);
    }
    void radiusMatch(InputArray query, std::vector<std::vector<DMatch> > &matches, float maxDistance, const std::vector<GpuMat> &masks,
                     bool compactResult) override
    {
        GpuMat g;
        radiusMatchAsync(query, g, maxDistance, masks, Stream::Null());
        radiusMatchConvert(g, matches, compactResult);
    }
    void radiusMatchAsync(InputArray query, InputArray train, OutputArray matches, float maxDistance, InputArray mask, Stream &stream) override
    {
        if (query.empty() || train.empty()) { matches.release(); return; }
        const int nq = query.rows, cols = std::max(train.rows / 100, nq);
        matches.create(2 * nq + 1, cols, CV_32SC1);                        // :897-904
        mi_mat q = miMat(query), t = miMat(train), m = miMat(mask);
        mi_mat i = packedRows(matches, 0, nq, cols, CV_32SC1), d = packedRows(matches, nq, nq, cols, CV_32FC1);
        mi_mat n = packedRows(matches, 2 * nq, 1, nq, CV_32SC1);
        miCheck(mi_bf_radius_match(h_, &q, &t, mask.empty() ? nullptr : &m, 1, maxDistance, &i, nullptr, &d, &n, stream.hipStream()));
    }
    void radiusMatchAsync(InputArray query, OutputArray matches, float maxDistance, const std::vector<GpuMat> &masks, Stream &stream) override
    {
        if (query.empty() || coll_.empty()) { matches.release(); return; }
        const int nq = query.rows;
        matches.create(3 * nq + 1, nq, CV_32FC1);                          // :969-975 (typed CV_32FC1 there too)
        std::vector<mi_mat> ts, ms;
        collection(masks, ts, ms);
        mi_mat q = miMat(query);
        mi_mat i = packedRows(matches, 0, nq, nq, CV_32SC1), g = packedRows(matches, nq, nq, nq, CV_32SC1);
        mi_mat d = packedRows(matches, 2 * nq, nq, nq, CV_32FC1), n = packedRows(matches, 3 * nq, 1, nq, CV_32SC1);
        miCheck(mi_bf_radius_match(h_, &q, ts.data(), ms.empty() ? nullptr : ms.data(), (int)ts.size(), maxDistance, &i, &g, &d, &n,
                                   stream.hipStream()));
    }
    void radiusMatchConvert(InputArray gpu_matches, std::vector<std::vector<DMatch> > &matches, bool compactResult) override
    {
        matches.clear();
        if (gpu_matches.empty()) return;
        CV_Assert(gpu_matches.type() == CV_32SC1 || gpu_matches.type() == CV_32FC1);
        const bool coll = gpu_matches.type() == CV_32FC1;                  // :1007-1024
        const int cols = gpu_matches.cols, nq = (gpu_matches.rows - 1) / (coll ? 3 : 2);
        std::vector<int> h((size_t)gpu_matches.rows * cols);
        gpu_matches.download(h.data(), sizeof(int) * cols);
        const size_t blk = (size_t)nq * cols;
        unpackLists(h.data(), coll ? h.data() + blk : nullptr, reinterpret_cast<const float *>(h.data() + (coll ? 2 : 1) * blk), nq, cols, cols,
                    h.data() + (coll ? 3 : 2) * blk, compactResult, true, matches);
    }

private:
    // makeGpuCollection (brute_force_matcher.cpp:143-185)
    void collection(const std::vector<GpuMat> &masks, std::vector<mi_mat> &ts, std::vector<mi_mat> &ms) const
    {
        CV_Assert(masks.empty() || masks.size() == coll_.size());
        for (size_t j = 0; j < coll_.size(); ++j) ts.push_back(miMat(coll_[j]));
        for (size_t j = 0; j < masks.size(); ++j) {
            mi_mat m = miMat(masks[j]);
            if (masks[j].empty()) { m.data = nullptr; m.step = 0; m.rows = m.cols = 0; }
            ms.push_back(m);
        }
    }
    // idx / img / dist: nq rows of `stride` entries; n: entries valid per row (NULL: every entry whose trainIdx != -1)
    static void unpackLists(const int *idx, const int *img, const float *dist, int nq, int k, int stride, const int *n, bool compactResult,
                            bool sortRows, std::vector<std::vector<DMatch> > &matches)
    {
        for (int q = 0; q < nq; ++q) {
            std::vector<DMatch> row;
            const int cnt = n ? std::min(n[q], k) : k;
            for (int j = 0; j < cnt; ++j) {
                const size_t e = (size_t)q * stride + j;
                if (n || idx[e] != -1) row.push_back(DMatch(q, idx[e], img ? img[e] : 0, dist[e]));
            }
            if (sortRows) std::stable_sort(row.begin(), row.end());
            if (!compactResult || !row.empty()) matches.push_back(row);
        }
    }

    mi_bfmatcher *h_ = nullptr;
    std::vector<GpuMat> coll_;
};
}  // namespace miflow_detail

inline Ptr<DescriptorMatcher> DescriptorMatcher::createBFMatcher(int normType) { return makePtr<miflow_detail::BFMatcherImpl>(normType); }

}  // namespace cuda
}  // namespace cv
#endif
