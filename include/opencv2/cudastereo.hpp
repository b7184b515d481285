// Drop-in for modules/cudastereo/include/opencv2/cudastereo.hpp (cuda::StereoBM + createStereoBM).
#ifndef MIFLOW_OPENCV_CUDASTEREO_HPP
#define MIFLOW_OPENCV_CUDASTEREO_HPP

#include "opencv2/core/cuda.hpp"

namespace cv {

#ifndef MIFLOW_WITH_OPENCV
/** the part of cv::StereoMatcher / cv::StereoBM (main repo calib3d.hpp) that cuda::StereoBM inherits */
class StereoMatcher : public Algorithm {
public:
    virtual int getMinDisparity() const = 0;       virtual void setMinDisparity(int) = 0;
    virtual int getNumDisparities() const = 0;     virtual void setNumDisparities(int) = 0;
    virtual int getBlockSize() const = 0;          virtual void setBlockSize(int) = 0;
    virtual int getSpeckleWindowSize() const = 0;  virtual void setSpeckleWindowSize(int) = 0;
    virtual int getSpeckleRange() const = 0;       virtual void setSpeckleRange(int) = 0;
    virtual int getDisp12MaxDiff() const = 0;      virtual void setDisp12MaxDiff(int) = 0;
};
class StereoBM : public StereoMatcher {
public:
    enum { PREFILTER_NORMALIZED_RESPONSE = 0, PREFILTER_XSOBEL = 1 };
    virtual int getPreFilterType() const = 0;      virtual void setPreFilterType(int) = 0;
    virtual int getPreFilterSize() const = 0;      virtual void setPreFilterSize(int) = 0;
    virtual int getPreFilterCap() const = 0;       virtual void setPreFilterCap(int) = 0;
    virtual int getTextureThreshold() const = 0;   virtual void setTextureThreshold(int) = 0;
    virtual int getUniquenessRatio() const = 0;    virtual void setUniquenessRatio(int) = 0;
    virtual int getSmallerBlockSize() const = 0;   virtual void setSmallerBlockSize(int) = 0;
    virtual Rect getROI1() const = 0;              virtual void setROI1(Rect) = 0;
    virtual Rect getROI2() const = 0;              virtual void setROI2(Rect) = 0;
};
#endif

namespace cuda {

/** cudastereo.hpp:72-84 */
class StereoBM : public cv::StereoBM {
public:
    virtual void compute(InputArray left, InputArray right, OutputArray disparity) = 0;
    virtual void compute(InputArray left, InputArray right, OutputArray disparity, Stream &stream) = 0;
};

namespace miflow_detail {
/** twin of StereoBMImpl, cudastereo/src/stereobm.cpp:67-132 (no-op setters and constant getters included) */
class StereoBMImpl final : public cuda::StereoBM {
public:
    StereoBMImpl(int numDisparities, int blockSize)
    {
        mi_stereobm_default_params(&p_);
        p_.num_disparities = numDisparities; p_.block_size = blockSize;
        miCheck(mi_stereobm_create(&p_, &h_));
        ok_ = p_;
    }
    ~StereoBMImpl() override { mi_stereobm_destroy(h_); }
    StereoBMImpl(const StereoBMImpl &) = delete;
    StereoBMImpl &operator=(const StereoBMImpl &) = delete;
    void compute(InputArray left, InputArray right, OutputArray disparity) override { compute(left, right, disparity, Stream::Null()); }
    void compute(InputArray left, InputArray right, OutputArray disparity, Stream &stream) override
    {
        disparity.create(left.size(), CV_8UC1);   // stereobm.cpp:154
        mi_mat l = miMat(left), r = miMat(right), d = miMat(disparity);
        miCheck(mi_stereobm_compute(h_, &l, &r, &d, stream.hipStream()));
    }
    // n pairs through this object (miflow extension; reached through cv::cuda::miflow::computeBatch)
    void computeBatch(const std::vector<GpuMat> &lefts, const std::vector<GpuMat> &rights, std::vector<GpuMat> &disps, Stream &stream)
    {
        CV_Assert(!lefts.empty() && lefts.size() == rights.size());
        disps.resize(lefts.size());
        std::vector<mi_mat> l(lefts.size()), r(lefts.size()), d(lefts.size());
        for (size_t i = 0; i < lefts.size(); ++i) {
            disps[i].create(lefts[i].size(), CV_8UC1);
            l[i] = miMat(lefts[i]); r[i] = miMat(rights[i]); d[i] = miMat(disps[i]);
        }
        miCheck(mi_stereobm_compute_batch(h_, (int)l.size(), l.data(), r.data(), d.data(), stream.hipStream()));
    }
    int getMinDisparity() const override { return 0; }           void setMinDisparity(int) override {}
    int getNumDisparities() const override { return p_.num_disparities; }
    void setNumDisparities(int v) override { p_.num_disparities = v; push(); }
    int getBlockSize() const override { return p_.block_size; }   void setBlockSize(int v) override { p_.block_size = v; push(); }
    int getSpeckleWindowSize() const override { return 0; }      void setSpeckleWindowSize(int) override {}
    int getSpeckleRange() const override { return 0; }           void setSpeckleRange(int) override {}
    int getDisp12MaxDiff() const override { return 0; }          void setDisp12MaxDiff(int) override {}
    int getPreFilterType() const override { return p_.prefilter_type; }  void setPreFilterType(int v) override { p_.prefilter_type = v; push(); }
    int getPreFilterSize() const override { return p_.prefilter_size; }  void setPreFilterSize(int v) override { p_.prefilter_size = v; push(); }
    int getPreFilterCap() const override { return p_.prefilter_cap; }    void setPreFilterCap(int v) override { p_.prefilter_cap = v; push(); }
    int getTextureThreshold() const override { return (int)p_.texture_threshold; }
    void setTextureThreshold(int v) override { p_.texture_threshold = (float)v; push(); }
    int getUniquenessRatio() const override { return p_.uniqueness_ratio; }  void setUniquenessRatio(int v) override { p_.uniqueness_ratio = v; push(); }
    int getSmallerBlockSize() const override { return 0; }       void setSmallerBlockSize(int) override {}
    Rect getROI1() const override { return Rect(); }             void setROI1(Rect) override {}
    Rect getROI2() const override { return Rect(); }             void setROI2(Rect) override {}
private:
    void push() { const int rc = mi_stereobm_set_params(h_, &p_); if (rc) p_ = ok_; miCheck(rc); ok_ = p_; }   // a rejected value is rolled back
    mi_stereobm_params p_;
    mi_stereobm_params ok_{};   // last parameters the library accepted (set at the end of the constructor)
    mi_stereobm *h_ = nullptr;
};
}  // namespace miflow_detail

/** cudastereo.hpp:90 */
inline Ptr<cuda::StereoBM> createStereoBM(int numDisparities = 64, int blockSize = 19)
{
    return makePtr<miflow_detail::StereoBMImpl>(numDisparities, blockSize);
}

namespace miflow {
/** n stereo pairs through one StereoBM object, back to back on the stream (miflow extension). */
inline void computeBatch(const Ptr<cuda::StereoBM> &bm, const std::vector<GpuMat> &lefts, const std::vector<GpuMat> &rights,
                         std::vector<GpuMat> &disps, Stream &stream = Stream::Null())
{
    auto *impl = dynamic_cast<miflow_detail::StereoBMImpl *>(bm.get());
    CV_Assert(impl);
    impl->computeBatch(lefts, rights, disps, stream);
}
}  // namespace miflow

/** cudastereo.hpp: class StereoSGM (cv::StereoSGBM interface subset used by the CUDA class, cudastereo/src/stereosgm.cpp:20-80) */
class CV_EXPORTS_W StereoSGM : public cv::StereoMatcher {
public:
    enum { MODE_SGBM = 0, MODE_HH = 1, MODE_SGBM_3WAY = 2, MODE_HH4 = 3 };
    virtual void compute(InputArray left, InputArray right, OutputArray disparity) = 0;
    virtual void compute(InputArray left, InputArray right, OutputArray disparity, Stream &stream) = 0;
    virtual int getPreFilterCap() const = 0;      virtual void setPreFilterCap(int) = 0;
    virtual int getUniquenessRatio() const = 0;   virtual void setUniquenessRatio(int) = 0;
    virtual int getP1() const = 0;                virtual void setP1(int) = 0;
    virtual int getP2() const = 0;                virtual void setP2(int) = 0;
    virtual int getMode() const = 0;              virtual void setMode(int) = 0;
};

namespace miflow_detail {
/** twin of StereoSGMImpl, cudastereo/src/stereosgm.cpp:20-146 */
class StereoSGMImpl final : public cuda::StereoSGM {
public:
    StereoSGMImpl(int minDisparity, int numDisparities, int P1, int P2, int uniquenessRatio, int mode)
    {
        mi_stereosgm_default_params(&p_);
        p_.min_disparity = minDisparity; p_.num_disparities = numDisparities; p_.P1 = P1; p_.P2 = P2;
        p_.uniqueness_ratio = uniquenessRatio; p_.mode = mode;
        miCheck(mi_stereosgm_create(&p_, &h_));
        ok_ = p_;
    }
    ~StereoSGMImpl() override { mi_stereosgm_destroy(h_); }
    StereoSGMImpl(const StereoSGMImpl &) = delete;
    StereoSGMImpl &operator=(const StereoSGMImpl &) = delete;
    void compute(InputArray left, InputArray right, OutputArray disparity) override { compute(left, right, disparity, Stream::Null()); }
    void compute(InputArray left, InputArray right, OutputArray disparity, Stream &stream) override
    {
        disparity.create(left.size(), CV_MAKETYPE(3 /* CV_16S */, 1));   // stereosgm.cpp:110
        mi_mat l = miMat(left), r = miMat(right), d = miMat(disparity);
        miCheck(mi_stereosgm_compute(h_, &l, &r, &d, stream.hipStream()));
    }
    int getBlockSize() const override { return -1; }             void setBlockSize(int) override {}
    int getDisp12MaxDiff() const override { return 1; }          void setDisp12MaxDiff(int) override {}
    int getMinDisparity() const override { return p_.min_disparity; }       void setMinDisparity(int v) override { p_.min_disparity = v; push(); }
    int getNumDisparities() const override { return p_.num_disparities; }   void setNumDisparities(int v) override { p_.num_disparities = v; push(); }
    int getSpeckleWindowSize() const override { return 0; }      void setSpeckleWindowSize(int) override {}
    int getSpeckleRange() const override { return 0; }           void setSpeckleRange(int) override {}
    int getP1() const override { return p_.P1; }                 void setP1(int v) override { p_.P1 = v; push(); }
    int getP2() const override { return p_.P2; }                 void setP2(int v) override { p_.P2 = v; push(); }
    int getUniquenessRatio() const override { return p_.uniqueness_ratio; } void setUniquenessRatio(int v) override { p_.uniqueness_ratio = v; push(); }
    int getMode() const override { return p_.mode; }             void setMode(int v) override { p_.mode = v; push(); }
    int getPreFilterCap() const override { return -1; }          void setPreFilterCap(int) override {}
private:
    void push() { const int rc = mi_stereosgm_set_params(h_, &p_); if (rc) p_ = ok_; miCheck(rc); ok_ = p_; }   // a rejected value is rolled back
    mi_stereosgm_params p_;
    mi_stereosgm_params ok_{};   // last parameters the library accepted (set at the end of the constructor)
    mi_stereosgm *h_ = nullptr;
};
}  // namespace miflow_detail

/** cudastereo.hpp: createStereoSGM */
inline Ptr<cuda::StereoSGM> createStereoSGM(int minDisparity = 0, int numDisparities = 128, int P1 = 10, int P2 = 120, int uniquenessRatio = 5,
                                            int mode = cuda::StereoSGM::MODE_HH4)
{
    return makePtr<miflow_detail::StereoSGMImpl>(minDisparity, numDisparities, P1, P2, uniquenessRatio, mode);
}

/** cudastereo.hpp:298-330 */
class CV_EXPORTS_W DisparityBilateralFilter : public cv::Algorithm {
public:
    virtual void apply(InputArray disparity, InputArray image, OutputArray dst, Stream &stream = Stream::Null()) = 0;
    virtual int getNumDisparities() const = 0;        virtual void setNumDisparities(int numDisparities) = 0;
    virtual int getRadius() const = 0;                virtual void setRadius(int radius) = 0;
    virtual int getNumIters() const = 0;              virtual void setNumIters(int iters) = 0;
    virtual double getEdgeThreshold() const = 0;      virtual void setEdgeThreshold(double edge_threshold) = 0;
    virtual double getMaxDiscThreshold() const = 0;   virtual void setMaxDiscThreshold(double max_disc_threshold) = 0;
    virtual double getSigmaRange() const = 0;         virtual void setSigmaRange(double sigma_range) = 0;
};

namespace miflow_detail {
/** twin of DispBilateralFilterImpl, cudastereo/src/disparity_bilateral_filter.cpp:58-190 */
class DispBilateralFilterImpl final : public cuda::DisparityBilateralFilter {
public:
    DispBilateralFilterImpl(int ndisp, int radius, int iters)
    {
        mi_disp_bilateral_default_params(&p_);
        p_.ndisp = ndisp; p_.radius = radius; p_.iters = iters;
        miCheck(mi_disp_bilateral_create(&p_, &h_));
        ok_ = p_;
    }
    ~DispBilateralFilterImpl() override { mi_disp_bilateral_destroy(h_); }
    DispBilateralFilterImpl(const DispBilateralFilterImpl &) = delete;
    DispBilateralFilterImpl &operator=(const DispBilateralFilterImpl &) = delete;
    void apply(InputArray disparity, InputArray image, OutputArray dst, Stream &stream) override
    {
        if (dst.data != disparity.data) dst.create(disparity.size(), disparity.type());   // disparity_bilateral_filter.cpp:152
        mi_mat d = miMat(disparity), i = miMat(image), o = miMat(dst);
        miCheck(mi_disp_bilateral_apply(h_, &d, &i, &o, stream.hipStream()));
    }
    int getNumDisparities() const override { return p_.ndisp; }                  void setNumDisparities(int v) override { p_.ndisp = v; push(); }
    int getRadius() const override { return p_.radius; }                         void setRadius(int v) override { p_.radius = v; push(); }
    int getNumIters() const override { return p_.iters; }                        void setNumIters(int v) override { p_.iters = v; push(); }
    double getEdgeThreshold() const override { return p_.edge_threshold; }       void setEdgeThreshold(double v) override { p_.edge_threshold = (float)v; push(); }
    double getMaxDiscThreshold() const override { return p_.max_disc_threshold; } void setMaxDiscThreshold(double v) override { p_.max_disc_threshold = (float)v; push(); }
    double getSigmaRange() const override { return p_.sigma_range; }             void setSigmaRange(double v) override { p_.sigma_range = (float)v; push(); }
private:
    void push() { const int rc = mi_disp_bilateral_set_params(h_, &p_); if (rc) p_ = ok_; miCheck(rc); ok_ = p_; }   // a rejected value is rolled back
    mi_disp_bilateral_params p_;
    mi_disp_bilateral_params ok_{};   // last parameters the library accepted (set at the end of the constructor)
    mi_disp_bilateral *h_ = nullptr;
};
}  // namespace miflow_detail

/** cudastereo.hpp:338-339 */
inline Ptr<cuda::DisparityBilateralFilter> createDisparityBilateralFilter(int ndisp = 64, int radius = 3, int iters = 1)
{
    return makePtr<miflow_detail::DispBilateralFilterImpl>(ndisp, radius, iters);
}

}}  // namespace cv::cuda
#endif
