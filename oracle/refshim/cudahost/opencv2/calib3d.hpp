/* oracle/refshim/cudahost: of opencv2/calib3d.hpp the three abstract bases modules/cudastereo/include/opencv2/cudastereo.hpp derives from
 * (cv::StereoMatcher, cv::StereoBM, cv::StereoSGBM: the main repo's calib3d.hpp; property lists as cudastereo/src/stereobm.cpp overrides
 * them).  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDAHOST_CALIB3D_HPP
#define ORACLE_CUDAHOST_CALIB3D_HPP
#include "opencv2/core/cuda.hpp"
namespace cv {
class StereoMatcher : public Algorithm {
public:
    virtual void compute(InputArray left, InputArray right, OutputArray disparity) = 0;
    virtual int getMinDisparity() const = 0;
    virtual void setMinDisparity(int minDisparity) = 0;
    virtual int getNumDisparities() const = 0;
    virtual void setNumDisparities(int numDisparities) = 0;
    virtual int getBlockSize() const = 0;
    virtual void setBlockSize(int blockSize) = 0;
    virtual int getSpeckleWindowSize() const = 0;
    virtual void setSpeckleWindowSize(int speckleWindowSize) = 0;
    virtual int getSpeckleRange() const = 0;
    virtual void setSpeckleRange(int speckleRange) = 0;
    virtual int getDisp12MaxDiff() const = 0;
    virtual void setDisp12MaxDiff(int disp12MaxDiff) = 0;
};
class StereoBM : public StereoMatcher {
public:
    enum { PREFILTER_NORMALIZED_RESPONSE = 0, PREFILTER_XSOBEL = 1 };
    virtual int getPreFilterType() const = 0;
    virtual void setPreFilterType(int preFilterType) = 0;
    virtual int getPreFilterSize() const = 0;
    virtual void setPreFilterSize(int preFilterSize) = 0;
    virtual int getPreFilterCap() const = 0;
    virtual void setPreFilterCap(int preFilterCap) = 0;
    virtual int getTextureThreshold() const = 0;
    virtual void setTextureThreshold(int textureThreshold) = 0;
    virtual int getUniquenessRatio() const = 0;
    virtual void setUniquenessRatio(int uniquenessRatio) = 0;
    virtual int getSmallerBlockSize() const = 0;
    virtual void setSmallerBlockSize(int blockSize) = 0;
    virtual Rect getROI1() const = 0;
    virtual void setROI1(Rect roi1) = 0;
    virtual Rect getROI2() const = 0;
    virtual void setROI2(Rect roi2) = 0;
};
class StereoSGBM : public StereoMatcher {
public:
    enum { MODE_SGBM = 0, MODE_HH = 1, MODE_SGBM_3WAY = 2, MODE_HH4 = 3 };
};
}
#endif
