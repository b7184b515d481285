/* oracle/refshim/cvstub: the declaration of cv::optflow::DualTVL1OpticalFlow as modules/optflow/include/opencv2/optflow.hpp:218-300
 * states it (interface only: pure virtual accessors, create(), createOptFlow_DualTVL1) -- the rest of that header needs the
 * main repo.  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CVSTUB_OPTFLOW_HPP
#define ORACLE_CVSTUB_OPTFLOW_HPP
#include "opencv2/core.hpp"
namespace cv { namespace optflow {
class CV_EXPORTS_W DualTVL1OpticalFlow : public DenseOpticalFlow
{
public:
    CV_WRAP virtual double getTau() const = 0;               CV_WRAP virtual void setTau(double val) = 0;
    CV_WRAP virtual double getLambda() const = 0;            CV_WRAP virtual void setLambda(double val) = 0;
    CV_WRAP virtual double getTheta() const = 0;             CV_WRAP virtual void setTheta(double val) = 0;
    CV_WRAP virtual double getGamma() const = 0;             CV_WRAP virtual void setGamma(double val) = 0;
    CV_WRAP virtual int getScalesNumber() const = 0;         CV_WRAP virtual void setScalesNumber(int val) = 0;
    CV_WRAP virtual int getWarpingsNumber() const = 0;       CV_WRAP virtual void setWarpingsNumber(int val) = 0;
    CV_WRAP virtual double getEpsilon() const = 0;           CV_WRAP virtual void setEpsilon(double val) = 0;
    CV_WRAP virtual int getInnerIterations() const = 0;      CV_WRAP virtual void setInnerIterations(int val) = 0;
    CV_WRAP virtual int getOuterIterations() const = 0;      CV_WRAP virtual void setOuterIterations(int val) = 0;
    CV_WRAP virtual bool getUseInitialFlow() const = 0;      CV_WRAP virtual void setUseInitialFlow(bool val) = 0;
    CV_WRAP virtual double getScaleStep() const = 0;         CV_WRAP virtual void setScaleStep(double val) = 0;
    CV_WRAP virtual int getMedianFiltering() const = 0;      CV_WRAP virtual void setMedianFiltering(int val) = 0;
    CV_WRAP static Ptr<DualTVL1OpticalFlow> create(double tau = 0.25, double lambda = 0.15, double theta = 0.3, int nscales = 5,
                                                   int warps = 5, double epsilon = 0.01, int innnerIterations = 30,
                                                   int outerIterations = 10, double scaleStep = 0.8, double gamma = 0.0,
                                                   int medianFiltering = 5, bool useInitialFlow = false);
};
CV_EXPORTS_W Ptr<DualTVL1OpticalFlow> createOptFlow_DualTVL1();
}}
#endif
