// cv::superres GPU optical-flow adapters over libmiflow -- the in-tree CALLER of the hot path.
//
// Same interface names as the reference (superres/include/opencv2/superres/optical_flow.hpp:56-141): DenseOpticalFlowExt,
// FarnebackOpticalFlow, DualTVL1OpticalFlow, createOptFlow_Farneback_CUDA(), createOptFlow_DualTVL1_CUDA().  The classes
// follow superres/src/optical_flow.cpp:436-505 (GpuOpticalFlow), :665-750 (Farneback_CUDA) and :757-845 (DualTVL1_CUDA):
// convert both frames to CV_8UC1, push the cached parameters into the cv::cuda class, calc, split the flow into two planes.
// Shim difference: frames and flows are cv::cuda::GpuMat (this repository's stand-in core has no cv::Mat / InputArray).
#ifndef OPENCV_SUPERRES_OPTICAL_FLOW_MIFLOW_HPP
#define OPENCV_SUPERRES_OPTICAL_FLOW_MIFLOW_HPP

#include "opencv2/core/cuda.hpp"
#include "opencv2/cudaoptflow.hpp"

namespace cv {
namespace superres {

class CV_EXPORTS DenseOpticalFlowExt : public cv::Algorithm {
public:
    // flow2 == nullptr: flow1 receives the merged CV_32FC2 field (the reference's `!_flow2.needed()` branch)
    virtual void calc(const cuda::GpuMat &frame0, const cuda::GpuMat &frame1, cuda::GpuMat &flow1, cuda::GpuMat *flow2 = nullptr) = 0;
    virtual void collectGarbage() = 0;
};

class CV_EXPORTS FarnebackOpticalFlow : public virtual DenseOpticalFlowExt {
public:
    virtual double getPyrScale() const = 0;
    virtual void setPyrScale(double val) = 0;
    virtual int getLevelsNumber() const = 0;
    virtual void setLevelsNumber(int val) = 0;
    virtual int getWindowSize() const = 0;
    virtual void setWindowSize(int val) = 0;
    virtual int getIterations() const = 0;
    virtual void setIterations(int val) = 0;
    virtual int getPolyN() const = 0;
    virtual void setPolyN(int val) = 0;
    virtual double getPolySigma() const = 0;
    virtual void setPolySigma(double val) = 0;
    virtual int getFlags() const = 0;
    virtual void setFlags(int val) = 0;
};

class CV_EXPORTS DualTVL1OpticalFlow : public virtual DenseOpticalFlowExt {
public:
    virtual double getTau() const = 0;
    virtual void setTau(double val) = 0;
    virtual double getLambda() const = 0;
    virtual void setLambda(double val) = 0;
    virtual double getTheta() const = 0;
    virtual void setTheta(double val) = 0;
    virtual int getScalesNumber() const = 0;
    virtual void setScalesNumber(int val) = 0;
    virtual int getWarpingsNumber() const = 0;
    virtual void setWarpingsNumber(int val) = 0;
    virtual double getEpsilon() const = 0;
    virtual void setEpsilon(double val) = 0;
    virtual int getIterations() const = 0;
    virtual void setIterations(int val) = 0;
    virtual bool getUseInitialFlow() const = 0;
    virtual void setUseInitialFlow(bool val) = 0;
};

namespace detail {

// convertToType(frame, CV_8UC1, ...) of input_array_utility.cpp:291-314, one kernel
inline cuda::GpuMat toGray8(const cuda::GpuMat &src, cuda::GpuMat &buf)
{
    if (src.type() == CV_8UC1) return src;
    buf.create(src.size(), CV_8UC1);
    mi_mat s = cuda::miMat(src), d = cuda::miMat(buf);
    cuda::miCheck(mi_superres_to_gray8(&s, &d, nullptr));
    return buf;
}

class GpuOpticalFlow : public virtual DenseOpticalFlowExt {
public:
    void calc(const cuda::GpuMat &frame0, const cuda::GpuMat &frame1, cuda::GpuMat &flow1, cuda::GpuMat *flow2) CV_OVERRIDE
    {
        CV_Assert(frame1.type() == frame0.type());   // optical_flow.cpp:466-467
        CV_Assert(frame1.size() == frame0.size());
        const cuda::GpuMat input0 = toGray8(frame0, buf_[0]), input1 = toGray8(frame1, buf_[1]);
        impl(input0, input1, flow_);
        if (!flow2) { flow1 = flow_; return; }
        flow1.create(flow_.size(), CV_32FC1);
        flow2->create(flow_.size(), CV_32FC1);
        mi_mat f = cuda::miMat(flow_), u = cuda::miMat(flow1), v = cuda::miMat(*flow2);
        cuda::miCheck(mi_split_flow(&f, &u, &v, nullptr));   // cuda::split, optical_flow.cpp:737-741
        cuda::miCheck(mi_stream_synchronize(nullptr));
    }
    void collectGarbage() CV_OVERRIDE
    {
        buf_[0].release(); buf_[1].release(); flow_.release();
    }

protected:
    virtual void impl(const cuda::GpuMat &input0, const cuda::GpuMat &input1, cuda::GpuMat &flow) = 0;

private:
    cuda::GpuMat buf_[2], flow_;
};

class Farneback_CUDA : public GpuOpticalFlow, public FarnebackOpticalFlow {
public:
    Farneback_CUDA() : alg_(cuda::FarnebackOpticalFlow::create())
    {
        pyrScale_ = alg_->getPyrScale(); numLevels_ = alg_->getNumLevels(); winSize_ = alg_->getWinSize();
        numIters_ = alg_->getNumIters(); polyN_ = alg_->getPolyN(); polySigma_ = alg_->getPolySigma(); flags_ = alg_->getFlags();
    }
    double getPyrScale() const CV_OVERRIDE { return pyrScale_; }
    void setPyrScale(double val) CV_OVERRIDE { pyrScale_ = val; }
    int getLevelsNumber() const CV_OVERRIDE { return numLevels_; }
    void setLevelsNumber(int val) CV_OVERRIDE { numLevels_ = val; }
    int getWindowSize() const CV_OVERRIDE { return winSize_; }
    void setWindowSize(int val) CV_OVERRIDE { winSize_ = val; }
    int getIterations() const CV_OVERRIDE { return numIters_; }
    void setIterations(int val) CV_OVERRIDE { numIters_ = val; }
    int getPolyN() const CV_OVERRIDE { return polyN_; }
    void setPolyN(int val) CV_OVERRIDE { polyN_ = val; }
    double getPolySigma() const CV_OVERRIDE { return polySigma_; }
    void setPolySigma(double val) CV_OVERRIDE { polySigma_ = val; }
    int getFlags() const CV_OVERRIDE { return flags_; }
    void setFlags(int val) CV_OVERRIDE { flags_ = val; }
    void collectGarbage() CV_OVERRIDE
    {
        alg_ = cuda::FarnebackOpticalFlow::create();   // optical_flow.cpp:744-748
        GpuOpticalFlow::collectGarbage();
    }

protected:
    void impl(const cuda::GpuMat &input0, const cuda::GpuMat &input1, cuda::GpuMat &flow) CV_OVERRIDE
    {
        alg_->setPyrScale(pyrScale_); alg_->setNumLevels(numLevels_); alg_->setWinSize(winSize_); alg_->setNumIters(numIters_);
        alg_->setPolyN(polyN_); alg_->setPolySigma(polySigma_); alg_->setFlags(flags_);
        alg_->calc(input0, input1, flow);
    }

private:
    double pyrScale_; int numLevels_, winSize_, numIters_, polyN_; double polySigma_; int flags_;
    Ptr<cuda::FarnebackOpticalFlow> alg_;
};

class DualTVL1_CUDA : public GpuOpticalFlow, public DualTVL1OpticalFlow {
public:
    DualTVL1_CUDA() : alg_(cuda::OpticalFlowDual_TVL1::create())
    {
        tau_ = alg_->getTau(); lambda_ = alg_->getLambda(); theta_ = alg_->getTheta(); nscales_ = alg_->getNumScales();
        warps_ = alg_->getNumWarps(); epsilon_ = alg_->getEpsilon(); iterations_ = alg_->getNumIterations();
        useInitialFlow_ = alg_->getUseInitialFlow();
    }
    double getTau() const CV_OVERRIDE { return tau_; }
    void setTau(double val) CV_OVERRIDE { tau_ = val; }
    double getLambda() const CV_OVERRIDE { return lambda_; }
    void setLambda(double val) CV_OVERRIDE { lambda_ = val; }
    double getTheta() const CV_OVERRIDE { return theta_; }
    void setTheta(double val) CV_OVERRIDE { theta_ = val; }
    int getScalesNumber() const CV_OVERRIDE { return nscales_; }
    void setScalesNumber(int val) CV_OVERRIDE { nscales_ = val; }
    int getWarpingsNumber() const CV_OVERRIDE { return warps_; }
    void setWarpingsNumber(int val) CV_OVERRIDE { warps_ = val; }
    double getEpsilon() const CV_OVERRIDE { return epsilon_; }
    void setEpsilon(double val) CV_OVERRIDE { epsilon_ = val; }
    int getIterations() const CV_OVERRIDE { return iterations_; }
    void setIterations(int val) CV_OVERRIDE { iterations_ = val; }
    bool getUseInitialFlow() const CV_OVERRIDE { return useInitialFlow_; }
    void setUseInitialFlow(bool val) CV_OVERRIDE { useInitialFlow_ = val; }
    void collectGarbage() CV_OVERRIDE
    {
        alg_ = cuda::OpticalFlowDual_TVL1::create();   // optical_flow.cpp:841-845
        GpuOpticalFlow::collectGarbage();
    }

protected:
    void impl(const cuda::GpuMat &input0, const cuda::GpuMat &input1, cuda::GpuMat &flow) CV_OVERRIDE
    {
        alg_->setTau(tau_); alg_->setLambda(lambda_); alg_->setTheta(theta_); alg_->setNumScales(nscales_);
        alg_->setNumWarps(warps_); alg_->setEpsilon(epsilon_); alg_->setNumIterations(iterations_);
        alg_->setUseInitialFlow(useInitialFlow_);
        alg_->calc(input0, input1, flow);
    }

private:
    double tau_, lambda_, theta_; int nscales_, warps_; double epsilon_; int iterations_; bool useInitialFlow_;
    Ptr<cuda::OpticalFlowDual_TVL1> alg_;
};

}  // namespace detail

inline Ptr<FarnebackOpticalFlow> createOptFlow_Farneback_CUDA() { return makePtr<detail::Farneback_CUDA>(); }
inline Ptr<DualTVL1OpticalFlow> createOptFlow_DualTVL1_CUDA() { return makePtr<detail::DualTVL1_CUDA>(); }

}  // namespace superres
}  // namespace cv

#endif
