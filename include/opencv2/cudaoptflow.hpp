// Drop-in for modules/cudaoptflow/include/opencv2/cudaoptflow.hpp (the classes on the hot path):
// same class names, factory signatures, defaults and getters/setters; implementation = the miflow
// C-ABI (include/miflow/c_api.h, libmiflow.so) running hand-written gfx950 HIP kernels.
#ifndef MIFLOW_OPENCV_CUDAOPTFLOW_HPP
#define MIFLOW_OPENCV_CUDAOPTFLOW_HPP

#include "opencv2/core/cuda.hpp"

namespace cv { namespace cuda {

/** cudaoptflow.hpp:70-81 */
class DenseOpticalFlow : public Algorithm {
public:
    virtual void calc(InputArray I0, InputArray I1, InputOutputArray flow, Stream &stream = Stream::Null()) = 0;
};

/** cudaoptflow.hpp:305-386; implementation twin of OpticalFlowDual_TVL1_Impl, cudaoptflow/src/tvl1flow.cpp:80-168 */
class OpticalFlowDual_TVL1 : public DenseOpticalFlow {
public:
    virtual double getTau() const = 0;             virtual void setTau(double tau) = 0;
    virtual double getLambda() const = 0;          virtual void setLambda(double lambda) = 0;
    virtual double getGamma() const = 0;           virtual void setGamma(double gamma) = 0;
    virtual double getTheta() const = 0;           virtual void setTheta(double theta) = 0;
    virtual int getNumScales() const = 0;          virtual void setNumScales(int nscales) = 0;
    virtual int getNumWarps() const = 0;           virtual void setNumWarps(int warps) = 0;
    virtual double getEpsilon() const = 0;         virtual void setEpsilon(double epsilon) = 0;
    virtual int getNumIterations() const = 0;      virtual void setNumIterations(int iterations) = 0;
    virtual double getScaleStep() const = 0;       virtual void setScaleStep(double scaleStep) = 0;
    virtual bool getUseInitialFlow() const = 0;    virtual void setUseInitialFlow(bool useInitialFlow) = 0;

    static Ptr<OpticalFlowDual_TVL1> create(double tau = 0.25, double lambda = 0.15, double theta = 0.3, int nscales = 5,
                                            int warps = 5, double epsilon = 0.01, int iterations = 300,
                                            double scaleStep = 0.8, double gamma = 0.0, bool useInitialFlow = false);
};

namespace miflow_detail {
class TVL1Impl final : public OpticalFlowDual_TVL1 {
public:
    explicit TVL1Impl(const mi_tvl1_params &p) : p_(p) { miCheck(mi_tvl1_create(&p_, &h_)); }
    ~TVL1Impl() override { mi_tvl1_destroy(h_); }
    TVL1Impl(const TVL1Impl &) = delete;
    TVL1Impl &operator=(const TVL1Impl &) = delete;
    void calc(InputArray I0, InputArray I1, InputOutputArray flow, Stream &stream) override
    {
        // BufferPool/merge of the reference (tvl1flow.cpp:175-182): the flow matrix is (re)allocated by the callee
        if (!p_.use_initial_flow) flow.create(I0.size(), CV_32FC2);
        mi_mat a = miMat(I0), b = miMat(I1), f = miMat(flow);
        miCheck(mi_tvl1_calc(h_, &a, &b, &f, stream.hipStream()));
        // the reference shrinks nscales_ for good when a pyramid level falls below 16 px (tvl1flow.cpp:243-247)
        miCheck(mi_tvl1_get_params(h_, &p_));
    }
    // Batched-frames mode (miflow extension; reached through cv::cuda::miflow::calcBatch): n independent pairs of one size and
    // type per launch sequence, blockIdx.z = pair.  flows[i] is (re)allocated like calc() does.
    void calcBatch(const std::vector<GpuMat> &I0s, const std::vector<GpuMat> &I1s, std::vector<GpuMat> &flows, Stream &stream)
    {
        CV_Assert(!I0s.empty() && I0s.size() == I1s.size());
        if (!p_.use_initial_flow) flows.resize(I0s.size());
        CV_Assert(flows.size() == I0s.size());
        std::vector<mi_mat> a(I0s.size()), b(I0s.size()), f(I0s.size());
        for (size_t i = 0; i < I0s.size(); ++i) {
            if (!p_.use_initial_flow) flows[i].create(I0s[i].size(), CV_32FC2);
            a[i] = miMat(I0s[i]); b[i] = miMat(I1s[i]); f[i] = miMat(flows[i]);
        }
        miCheck(mi_tvl1_calc_batch(h_, (int)a.size(), a.data(), b.data(), f.data(), stream.hipStream()));
        miCheck(mi_tvl1_get_params(h_, &p_));
    }
    void setExtra(int semantics, int exact_math, int time_block, int lanes)
    {
        mi_tvl1_params q = p_;
        q.semantics = semantics; q.exact_math = exact_math; q.time_block = time_block; q.lanes = lanes;
        miCheck(mi_tvl1_set_params(h_, &q));
        p_ = q;
    }
    void setStopSlack(int slack)
    {
        mi_tvl1_params q = p_;
        q.stop_slack = slack;
        miCheck(mi_tvl1_set_params(h_, &q));
        p_ = q;
    }
    void setHostFeedback(int mode)
    {
        mi_tvl1_params q = p_;
        q.host_feedback = mode;
        miCheck(mi_tvl1_set_params(h_, &q));
        p_ = q;
    }
    const mi_tvl1_params &params() const { return p_; }
    String getDefaultName() const override { return "DenseOpticalFlow.OpticalFlowDual_TVL1"; }   // tvl1flow.cpp:122
#define MIFLOW_PROP(T, Name, field) \
    T get##Name() const override { return (T)p_.field; } \
    void set##Name(T v) override { mi_tvl1_params q = p_; q.field = v; miCheck(mi_tvl1_set_params(h_, &q)); p_ = q; }
    MIFLOW_PROP(double, Tau, tau) MIFLOW_PROP(double, Lambda, lambda) MIFLOW_PROP(double, Gamma, gamma)
    MIFLOW_PROP(double, Theta, theta) MIFLOW_PROP(int, NumScales, nscales) MIFLOW_PROP(int, NumWarps, warps)
    MIFLOW_PROP(double, Epsilon, epsilon) MIFLOW_PROP(int, NumIterations, iterations) MIFLOW_PROP(double, ScaleStep, scale_step)
#undef MIFLOW_PROP
    bool getUseInitialFlow() const override { return p_.use_initial_flow != 0; }
    void setUseInitialFlow(bool v) override { mi_tvl1_params q = p_; q.use_initial_flow = v; miCheck(mi_tvl1_set_params(h_, &q)); p_ = q; }
    mi_tvl1 *handle() { return h_; }
private:
    mi_tvl1_params p_;
    mi_tvl1 *h_ = nullptr;
};
}  // namespace miflow_detail

inline Ptr<OpticalFlowDual_TVL1> OpticalFlowDual_TVL1::create(double tau, double lambda, double theta, int nscales, int warps,
                                                              double epsilon, int iterations, double scaleStep, double gamma,
                                                              bool useInitialFlow)
{
    mi_tvl1_params p;
    mi_tvl1_default_params(&p);
    p.tau = tau; p.lambda = lambda; p.theta = theta; p.nscales = nscales; p.warps = warps; p.epsilon = epsilon;
    p.iterations = iterations; p.scale_step = scaleStep; p.gamma = gamma; p.use_initial_flow = useInitialFlow;
    return makePtr<miflow_detail::TVL1Impl>(p);
}

/** miflow extensions to OpticalFlowDual_TVL1 -- everything the reference class does not have lives in this namespace, the class
 *  declaration above is the reference's.  NOTE ON NUMERICS: a default-constructed object computes with the arithmetic of the
 *  CPU class cv::optflow::DualTVL1OpticalFlow (MI_SEM_CPU_REF: the acceptance reference of this path, "EPE vs CPU ref") in fast
 *  device math, iterations fused per HBM pass; cv::cuda's own kernels differ from the CPU class by ~0.1 px mean EPE (border
 *  addressing, resize convention, check schedule) and are selected with setSemantics(alg, MI_SEM_CUDA_COMPAT). */
namespace miflow {
/** MI_SEM_CPU_REF (default) = the arithmetic of cv::optflow::DualTVL1OpticalFlow (cv::remap warp, cv::resize pyramid,
 *  convergence test after every iteration); MI_SEM_CUDA_COMPAT = cv::cuda's kernels (normalised a = -0.5 bicubic with clamp
 *  addressing, cuda::resize, sparse check schedule), bit-pinned on the reference's OpenCL twins. */
inline void setSemantics(const Ptr<OpticalFlowDual_TVL1> &alg, int semantics)
{
    auto *impl = dynamic_cast<miflow_detail::TVL1Impl *>(alg.get());
    CV_Assert(impl);
    impl->setExtra(semantics, impl->params().exact_math, impl->params().time_block, impl->params().lanes);
}
/** true: IEEE divide, f64 hypot, separately rounded operations in the reference's order (fused in blocks of up to 5 iterations when epsilon == 0)
 *  (bit-comparable with the CPU restatement); false (default): v_rcp / v_sqrt / fma, temporally blocked. */
inline void setExactMath(const Ptr<OpticalFlowDual_TVL1> &alg, bool exact)
{
    auto *impl = dynamic_cast<miflow_detail::TVL1Impl *>(alg.get());
    CV_Assert(impl);
    impl->setExtra(impl->params().semantics, exact ? 1 : 0, impl->params().time_block, impl->params().lanes);
}
/** 0 (default): the inner loop of a warp stops exactly where the reference's convergence test stops it; s > 0: it may run up
 *  to s iterations further (mi_tvl1_params.stop_slack). */
inline void setStopSlack(const Ptr<OpticalFlowDual_TVL1> &alg, int slack)
{
    auto *impl = dynamic_cast<miflow_detail::TVL1Impl *>(alg.get());
    CV_Assert(impl);
    impl->setStopSlack(slack);
}
/** mi_tvl1_params.host_feedback: 0 (default) = a convergence-checked calc() of one or two pairs waits for the device about once per
 *  warp -- as the reference's class does at every convergence check (cudaoptflow/src/tvl1flow.cpp:362-368) -- and stops enqueuing
 *  launches for a warp that has converged; -1 = never wait inside calc(); 1 = for every single-lane call.  Same flows either way. */
inline void setHostFeedback(const Ptr<OpticalFlowDual_TVL1> &alg, int mode)
{
    auto *impl = dynamic_cast<miflow_detail::TVL1Impl *>(alg.get());
    CV_Assert(impl);
    impl->setHostFeedback(mode);
}
/** n independent image pairs of identical size and type in one pass (the batched-frames mode of BASELINE configs[4]). */
inline void calcBatch(const Ptr<OpticalFlowDual_TVL1> &alg, const std::vector<GpuMat> &I0s, const std::vector<GpuMat> &I1s,
                      std::vector<GpuMat> &flows, Stream &stream = Stream::Null())
{
    auto *impl = dynamic_cast<miflow_detail::TVL1Impl *>(alg.get());
    CV_Assert(impl);
    impl->calcBatch(I0s, I1s, flows, stream);
}
}  // namespace miflow

/** cudaoptflow.hpp:258-294; implementation twin of FarnebackOpticalFlowImpl, cudaoptflow/src/farneback.cpp:96-165 */
class FarnebackOpticalFlow : public DenseOpticalFlow {
public:
    virtual int getNumLevels() const = 0;        virtual void setNumLevels(int numLevels) = 0;
    virtual double getPyrScale() const = 0;      virtual void setPyrScale(double pyrScale) = 0;
    virtual bool getFastPyramids() const = 0;    virtual void setFastPyramids(bool fastPyramids) = 0;
    virtual int getWinSize() const = 0;          virtual void setWinSize(int winSize) = 0;
    virtual int getNumIters() const = 0;         virtual void setNumIters(int numIters) = 0;
    virtual int getPolyN() const = 0;            virtual void setPolyN(int polyN) = 0;
    virtual double getPolySigma() const = 0;     virtual void setPolySigma(double polySigma) = 0;
    virtual int getFlags() const = 0;            virtual void setFlags(int flags) = 0;

    static Ptr<FarnebackOpticalFlow> create(int numLevels = 5, double pyrScale = 0.5, bool fastPyramids = false, int winSize = 13,
                                            int numIters = 10, int polyN = 5, double polySigma = 1.1, int flags = 0);
};

namespace miflow_detail {
class FarnebackImpl final : public FarnebackOpticalFlow {
public:
    explicit FarnebackImpl(const mi_farneback_params &p) : p_(p) { miCheck(mi_farneback_create(&p_, &h_)); }
    ~FarnebackImpl() override { mi_farneback_destroy(h_); }
    FarnebackImpl(const FarnebackImpl &) = delete;
    FarnebackImpl &operator=(const FarnebackImpl &) = delete;
    void calc(InputArray I0, InputArray I1, InputOutputArray flow, Stream &stream) override
    {
        if (!(p_.flags & MI_OPTFLOW_USE_INITIAL_FLOW)) flow.create(I0.size(), CV_32FC2);   // farneback.cpp:189-198 (create + merge)
        mi_mat a = miMat(I0), b = miMat(I1), f = miMat(flow);
        miCheck(mi_farneback_calc(h_, &a, &b, &f, stream.hipStream()));
    }
    // Batched-frames mode (miflow extension; reached through cv::cuda::miflow::calcBatch)
    void calcBatch(const std::vector<GpuMat> &I0s, const std::vector<GpuMat> &I1s, std::vector<GpuMat> &flows, Stream &stream)
    {
        CV_Assert(!I0s.empty() && I0s.size() == I1s.size());
        const bool init = (p_.flags & MI_OPTFLOW_USE_INITIAL_FLOW) != 0;
        if (!init) flows.resize(I0s.size());
        CV_Assert(flows.size() == I0s.size());
        std::vector<mi_mat> a(I0s.size()), b(I0s.size()), f(I0s.size());
        for (size_t i = 0; i < I0s.size(); ++i) {
            if (!init) flows[i].create(I0s[i].size(), CV_32FC2);
            a[i] = miMat(I0s[i]); b[i] = miMat(I1s[i]); f[i] = miMat(flows[i]);
        }
        miCheck(mi_farneback_calc_batch(h_, (int)a.size(), a.data(), b.data(), f.data(), stream.hipStream()));
    }
    String getDefaultName() const override { return "DenseOpticalFlow.FarnebackOpticalFlow"; }   // farneback.cpp:132
#define MIFLOW_PROP(T, Name, field) \
    T get##Name() const override { return (T)p_.field; } \
    void set##Name(T v) override { mi_farneback_params q = p_; q.field = v; miCheck(mi_farneback_set_params(h_, &q)); p_ = q; }
    MIFLOW_PROP(int, NumLevels, num_levels) MIFLOW_PROP(double, PyrScale, pyr_scale) MIFLOW_PROP(int, WinSize, win_size)
    MIFLOW_PROP(int, NumIters, num_iters) MIFLOW_PROP(int, PolyN, poly_n) MIFLOW_PROP(double, PolySigma, poly_sigma)
    MIFLOW_PROP(int, Flags, flags)
#undef MIFLOW_PROP
    bool getFastPyramids() const override { return p_.fast_pyramids != 0; }
    void setFastPyramids(bool v) override { mi_farneback_params q = p_; q.fast_pyramids = v; miCheck(mi_farneback_set_params(h_, &q)); p_ = q; }
private:
    mi_farneback_params p_;
    mi_farneback *h_ = nullptr;
};
}  // namespace miflow_detail

inline Ptr<FarnebackOpticalFlow> FarnebackOpticalFlow::create(int numLevels, double pyrScale, bool fastPyramids, int winSize,
                                                              int numIters, int polyN, double polySigma, int flags)
{
    mi_farneback_params p;
    mi_farneback_default_params(&p);
    p.num_levels = numLevels; p.pyr_scale = pyrScale; p.fast_pyramids = fastPyramids; p.win_size = winSize;
    p.num_iters = numIters; p.poly_n = polyN; p.poly_sigma = polySigma; p.flags = flags;
    return makePtr<miflow_detail::FarnebackImpl>(p);
}

namespace miflow {
/** n independent pairs in one pass through a FarnebackOpticalFlow object (blockIdx.z = pair in every kernel of the level loop). */
inline void calcBatch(const Ptr<FarnebackOpticalFlow> &alg, const std::vector<GpuMat> &I0s, const std::vector<GpuMat> &I1s,
                      std::vector<GpuMat> &flows, Stream &stream = Stream::Null())
{
    auto *impl = dynamic_cast<miflow_detail::FarnebackImpl *>(alg.get());
    CV_Assert(impl);
    impl->calcBatch(I0s, I1s, flows, stream);
}
}  // namespace miflow

/** cudaoptflow.hpp: class DensePyrLKOpticalFlow; implementation twin of DensePyrLKOpticalFlowImpl, cudaoptflow/src/pyrlk.cpp:354-406 */
class DensePyrLKOpticalFlow : public DenseOpticalFlow {
public:
    virtual Size getWinSize() const = 0;          virtual void setWinSize(Size winSize) = 0;
    virtual int getMaxLevel() const = 0;          virtual void setMaxLevel(int maxLevel) = 0;
    virtual int getNumIters() const = 0;          virtual void setNumIters(int iters) = 0;
    virtual bool getUseInitialFlow() const = 0;   virtual void setUseInitialFlow(bool useInitialFlow) = 0;
    static Ptr<DensePyrLKOpticalFlow> create(Size winSize = Size(13, 13), int maxLevel = 3, int iters = 30, bool useInitialFlow = false);
};

namespace miflow_detail {
class DensePyrLKImpl final : public DensePyrLKOpticalFlow {
public:
    explicit DensePyrLKImpl(const mi_densepyrlk_params &p) : p_(p) { miCheck(mi_densepyrlk_create(&p_, &h_)); }
    ~DensePyrLKImpl() override { mi_densepyrlk_destroy(h_); }
    DensePyrLKImpl(const DensePyrLKImpl &) = delete;
    DensePyrLKImpl &operator=(const DensePyrLKImpl &) = delete;
    void calc(InputArray prevImg, InputArray nextImg, InputOutputArray flow, Stream &stream) override
    {
        flow.create(prevImg.size(), CV_32FC2);   // cuda::merge into _flow, pyrlk.cpp:390-391
        mi_mat a = miMat(prevImg), b = miMat(nextImg), f = miMat(flow);
        miCheck(mi_densepyrlk_calc(h_, &a, &b, &f, stream.hipStream()));
    }
    String getDefaultName() const override { return "DenseOpticalFlow.DensePyrLKOpticalFlow"; }   // pyrlk.cpp:394
    Size getWinSize() const override { return Size(p_.win_width, p_.win_height); }
    void setWinSize(Size v) override { p_.win_width = v.width; p_.win_height = v.height; push(); }
    int getMaxLevel() const override { return p_.max_level; }           void setMaxLevel(int v) override { p_.max_level = v; push(); }
    int getNumIters() const override { return p_.iters; }               void setNumIters(int v) override { p_.iters = v; push(); }
    bool getUseInitialFlow() const override { return p_.use_initial_flow != 0; }
    void setUseInitialFlow(bool v) override { p_.use_initial_flow = v; push(); }
private:
    void push() { const int rc = mi_densepyrlk_set_params(h_, &p_); if (rc) p_ = ok_; miCheck(rc); ok_ = p_; }   // a rejected value is rolled back
    mi_densepyrlk_params p_;
    mi_densepyrlk_params ok_ = p_;   // last parameters the library accepted
    mi_densepyrlk *h_ = nullptr;
};
}  // namespace miflow_detail

inline Ptr<DensePyrLKOpticalFlow> DensePyrLKOpticalFlow::create(Size winSize, int maxLevel, int iters, bool useInitialFlow)
{
    mi_densepyrlk_params p;
    mi_densepyrlk_default_params(&p);
    p.win_width = winSize.width; p.win_height = winSize.height; p.max_level = maxLevel; p.iters = iters; p.use_initial_flow = useInitialFlow;
    return makePtr<miflow_detail::DensePyrLKImpl>(p);
}

/** cudaoptflow.hpp:85-104 SparseOpticalFlow, :203-223 SparsePyrLKOpticalFlow; implementation twin of SparsePyrLKOpticalFlowImpl,
 *  cudaoptflow/src/pyrlk.cpp:308-352.  CV_8UC1 frames.  `err` defaults to cuda::noArray() (a shared empty GpuMat): not computed. */
inline GpuMat &noArray() { static thread_local GpuMat none; none.release(); return none; }

class SparseOpticalFlow : public Algorithm {
public:
    virtual void calc(InputArray prevImg, InputArray nextImg, InputArray prevPts, InputOutputArray nextPts, OutputArray status,
                      OutputArray err = noArray(), Stream &stream = Stream::Null()) = 0;
};

class SparsePyrLKOpticalFlow : public SparseOpticalFlow {
public:
    virtual Size getWinSize() const = 0;          virtual void setWinSize(Size winSize) = 0;
    virtual int getMaxLevel() const = 0;          virtual void setMaxLevel(int maxLevel) = 0;
    virtual int getNumIters() const = 0;          virtual void setNumIters(int iters) = 0;
    virtual bool getUseInitialFlow() const = 0;   virtual void setUseInitialFlow(bool useInitialFlow) = 0;
    static Ptr<SparsePyrLKOpticalFlow> create(Size winSize = Size(21, 21), int maxLevel = 3, int iters = 30, bool useInitialFlow = false);
};

namespace miflow_detail {
class SparsePyrLKImpl final : public SparsePyrLKOpticalFlow {
public:
    explicit SparsePyrLKImpl(const mi_sparsepyrlk_params &p) : p_(p) { miCheck(mi_sparsepyrlk_create(&p_, &h_)); }
    ~SparsePyrLKImpl() override { mi_sparsepyrlk_destroy(h_); }
    SparsePyrLKImpl(const SparsePyrLKImpl &) = delete;
    SparsePyrLKImpl &operator=(const SparsePyrLKImpl &) = delete;
    void calc(InputArray prevImg, InputArray nextImg, InputArray prevPts, InputOutputArray nextPts, OutputArray status, OutputArray err,
              Stream &stream) override
    {
        const bool wantErr = &err != &noArray();
        if (prevPts.empty()) {                                    // pyrlk.cpp:209-215
            nextPts.release(); status.release();
            if (wantErr) err.release();
            return;
        }
        if (p_.use_initial_flow) CV_Assert(nextPts.size() == prevPts.size() && nextPts.type() == prevPts.type());   // :159-160
        else nextPts.create(1, prevPts.cols, prevPts.type());
        status.create(1, prevPts.cols, CV_8UC1);
        if (wantErr) err.create(1, prevPts.cols, CV_32FC1);
        mi_mat a = miMat(prevImg), b = miMat(nextImg), pp = miMat(prevPts), np = miMat(nextPts), st = miMat(status), er = miMat(err);
        miCheck(mi_sparsepyrlk_calc(h_, &a, &b, &pp, &np, &st, wantErr ? &er : nullptr, stream.hipStream()));
    }
    String getDefaultName() const override { return "SparseOpticalFlow.SparsePyrLKOpticalFlow"; }   // pyrlk.cpp:351
    Size getWinSize() const override { return Size(p_.win_width, p_.win_height); }
    void setWinSize(Size v) override { p_.win_width = v.width; p_.win_height = v.height; push(); }
    int getMaxLevel() const override { return p_.max_level; }           void setMaxLevel(int v) override { p_.max_level = v; push(); }
    int getNumIters() const override { return p_.iters; }               void setNumIters(int v) override { p_.iters = v; push(); }
    bool getUseInitialFlow() const override { return p_.use_initial_flow != 0; }
    void setUseInitialFlow(bool v) override { p_.use_initial_flow = v; push(); }
private:
    void push() { const int rc = mi_sparsepyrlk_set_params(h_, &p_); if (rc) p_ = ok_; miCheck(rc); ok_ = p_; }   // a rejected value is rolled back
    mi_sparsepyrlk_params p_;
    mi_sparsepyrlk_params ok_ = p_;   // last parameters the library accepted
    mi_sparsepyrlk *h_ = nullptr;
};
}  // namespace miflow_detail

inline Ptr<SparsePyrLKOpticalFlow> SparsePyrLKOpticalFlow::create(Size winSize, int maxLevel, int iters, bool useInitialFlow)
{
    mi_sparsepyrlk_params p;
    mi_sparsepyrlk_default_params(&p);
    p.win_width = winSize.width; p.win_height = winSize.height; p.max_level = maxLevel; p.iters = iters; p.use_initial_flow = useInitialFlow;
    return makePtr<miflow_detail::SparsePyrLKImpl>(p);
}

}}  // namespace cv::cuda
#endif
