/* oracle/refshim/cvsurf: cv::Feature2D as far as xfeatures2d::SURF derives from it (features2d.hpp of the main repo, not under /root/reference) */
#ifndef MIFLOW_CVSURF_FEATURES2D_HPP
#define MIFLOW_CVSURF_FEATURES2D_HPP
#include "core.hpp"
namespace cv {
class Feature2D : public virtual Algorithm {
public:
    virtual ~Feature2D() {}
    virtual void detectAndCompute(InputArray image, InputArray mask, std::vector<KeyPoint> &keypoints, OutputArray descriptors,
                                  bool useProvidedKeypoints = false) = 0;
    virtual int descriptorSize() const { return 0; }
    virtual int descriptorType() const { return CV_32F; }
    virtual int defaultNorm() const { return NORM_L2; }
    virtual String getDefaultName() const override { return "Feature2D"; }
};
}  // namespace cv
#endif
