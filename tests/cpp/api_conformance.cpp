// Compile-time check that the drop-in headers keep the reference's public signatures (member-function pointer types, factory
// defaults, public data members).  Every block cites the reference declaration it transcribes.  Nothing here runs.
#include <type_traits>
#include <vector>
#include "opencv2/cudaoptflow.hpp"
#include "opencv2/cudastereo.hpp"
#include "opencv2/xfeatures2d/cuda.hpp"
#include "opencv2/cudafeatures2d.hpp"
#include "opencv2/superres/optical_flow.hpp"

using namespace cv;
using cuda::GpuMat;
using cuda::Stream;
typedef std::vector<DMatch> Matches;
typedef std::vector<std::vector<DMatch> > MatchLists;
typedef std::vector<GpuMat> GpuMats;

#define SAME(expr, ...) static_assert(std::is_same<decltype(expr), __VA_ARGS__>::value, #expr)

// ---- cudaoptflow.hpp:80 DenseOpticalFlow::calc(InputArray I0, InputArray I1, InputOutputArray flow, Stream& = Stream::Null())
SAME(&cuda::DenseOpticalFlow::calc, void (cuda::DenseOpticalFlow::*)(cuda::InputArray, cuda::InputArray, cuda::InputOutputArray, Stream &));

// ---- cudaoptflow.hpp:305-386 OpticalFlowDual_TVL1: 10 getter / setter pairs + create with the class defaults
#define PROP(C, T, name) SAME(&C::get##name, T (C::*)() const); SAME(&C::set##name, void (C::*)(T))
PROP(cuda::OpticalFlowDual_TVL1, double, Tau);
PROP(cuda::OpticalFlowDual_TVL1, double, Lambda);
PROP(cuda::OpticalFlowDual_TVL1, double, Gamma);
PROP(cuda::OpticalFlowDual_TVL1, double, Theta);
PROP(cuda::OpticalFlowDual_TVL1, int, NumScales);
PROP(cuda::OpticalFlowDual_TVL1, int, NumWarps);
PROP(cuda::OpticalFlowDual_TVL1, double, Epsilon);
PROP(cuda::OpticalFlowDual_TVL1, int, NumIterations);
PROP(cuda::OpticalFlowDual_TVL1, double, ScaleStep);
PROP(cuda::OpticalFlowDual_TVL1, bool, UseInitialFlow);
SAME(&cuda::OpticalFlowDual_TVL1::create, Ptr<cuda::OpticalFlowDual_TVL1> (*)(double, double, double, int, int, double, int, double, double, bool));

// ---- cudaoptflow.hpp:258-294 FarnebackOpticalFlow: 8 properties + create
PROP(cuda::FarnebackOpticalFlow, int, NumLevels);
PROP(cuda::FarnebackOpticalFlow, double, PyrScale);
PROP(cuda::FarnebackOpticalFlow, bool, FastPyramids);
PROP(cuda::FarnebackOpticalFlow, int, WinSize);
PROP(cuda::FarnebackOpticalFlow, int, NumIters);
PROP(cuda::FarnebackOpticalFlow, int, PolyN);
PROP(cuda::FarnebackOpticalFlow, double, PolySigma);
PROP(cuda::FarnebackOpticalFlow, int, Flags);
SAME(&cuda::FarnebackOpticalFlow::create, Ptr<cuda::FarnebackOpticalFlow> (*)(int, double, bool, int, int, int, double, int));

// ---- cudaoptflow.hpp:207-225 DensePyrLKOpticalFlow
PROP(cuda::DensePyrLKOpticalFlow, Size, WinSize);
PROP(cuda::DensePyrLKOpticalFlow, int, MaxLevel);
PROP(cuda::DensePyrLKOpticalFlow, int, NumIters);
PROP(cuda::DensePyrLKOpticalFlow, bool, UseInitialFlow);
SAME(&cuda::DensePyrLKOpticalFlow::create, Ptr<cuda::DensePyrLKOpticalFlow> (*)(Size, int, int, bool));

// ---- cudaoptflow.hpp:85-104 SparseOpticalFlow::calc, :203-223 SparsePyrLKOpticalFlow
SAME(&cuda::SparseOpticalFlow::calc, void (cuda::SparseOpticalFlow::*)(cuda::InputArray, cuda::InputArray, cuda::InputArray, cuda::InputOutputArray,
                                                                        cuda::OutputArray, cuda::OutputArray, Stream &));
PROP(cuda::SparsePyrLKOpticalFlow, Size, WinSize);
PROP(cuda::SparsePyrLKOpticalFlow, int, MaxLevel);
PROP(cuda::SparsePyrLKOpticalFlow, int, NumIters);
PROP(cuda::SparsePyrLKOpticalFlow, bool, UseInitialFlow);
SAME(&cuda::SparsePyrLKOpticalFlow::create, Ptr<cuda::SparsePyrLKOpticalFlow> (*)(Size, int, int, bool));

// ---- cudastereo.hpp:72-90 StereoBM + createStereoBM, :246-286 StereoSGM + createStereoSGM, :298-341 DisparityBilateralFilter
SAME(static_cast<void (cuda::StereoBM::*)(cuda::InputArray, cuda::InputArray, cuda::OutputArray, Stream &)>(&cuda::StereoBM::compute),
     void (cuda::StereoBM::*)(cuda::InputArray, cuda::InputArray, cuda::OutputArray, Stream &));
SAME(&cuda::createStereoBM, Ptr<cuda::StereoBM> (*)(int, int));
SAME(static_cast<void (cuda::StereoSGM::*)(cuda::InputArray, cuda::InputArray, cuda::OutputArray, Stream &)>(&cuda::StereoSGM::compute),
     void (cuda::StereoSGM::*)(cuda::InputArray, cuda::InputArray, cuda::OutputArray, Stream &));
SAME(&cuda::createStereoSGM, Ptr<cuda::StereoSGM> (*)(int, int, int, int, int, int));
static_assert(cuda::StereoSGM::MODE_HH == 1 && cuda::StereoSGM::MODE_HH4 == 3, "cv::StereoSGBM mode values");
SAME(&cuda::DisparityBilateralFilter::apply, void (cuda::DisparityBilateralFilter::*)(cuda::InputArray, cuda::InputArray, cuda::OutputArray, Stream &));
PROP(cuda::DisparityBilateralFilter, int, NumDisparities);
PROP(cuda::DisparityBilateralFilter, int, Radius);
PROP(cuda::DisparityBilateralFilter, int, NumIters);
PROP(cuda::DisparityBilateralFilter, double, EdgeThreshold);
PROP(cuda::DisparityBilateralFilter, double, MaxDiscThreshold);
PROP(cuda::DisparityBilateralFilter, double, SigmaRange);
SAME(&cuda::createDisparityBilateralFilter, Ptr<cuda::DisparityBilateralFilter> (*)(int, int, int));

// ---- xfeatures2d/cuda.hpp:86-196 SURF_CUDA: constructors, create, 5 operator() overloads, detect*, up/download, public fields
static_assert(std::is_default_constructible<cuda::SURF_CUDA>::value && std::is_constructible<cuda::SURF_CUDA, double>::value &&
                  std::is_constructible<cuda::SURF_CUDA, double, int, int, bool, float, bool>::value, "SURF_CUDA constructors");
SAME(&cuda::SURF_CUDA::create, Ptr<cuda::SURF_CUDA> (*)(double, int, int, bool, float, bool));
SAME(&cuda::SURF_CUDA::descriptorSize, int (cuda::SURF_CUDA::*)() const);
SAME(&cuda::SURF_CUDA::defaultNorm, int (cuda::SURF_CUDA::*)() const);
SAME(&cuda::SURF_CUDA::uploadKeypoints, void (cuda::SURF_CUDA::*)(const std::vector<KeyPoint> &, GpuMat &));
SAME(&cuda::SURF_CUDA::downloadKeypoints, void (cuda::SURF_CUDA::*)(const GpuMat &, std::vector<KeyPoint> &));
SAME(&cuda::SURF_CUDA::downloadDescriptors, void (cuda::SURF_CUDA::*)(const GpuMat &, std::vector<float> &));
SAME(&cuda::SURF_CUDA::detect, void (cuda::SURF_CUDA::*)(const GpuMat &, const GpuMat &, GpuMat &));
SAME(&cuda::SURF_CUDA::detectWithDescriptors, void (cuda::SURF_CUDA::*)(const GpuMat &, const GpuMat &, GpuMat &, GpuMat &, bool));
SAME(&cuda::SURF_CUDA::releaseMemory, void (cuda::SURF_CUDA::*)());
#define SURF_CALL(...) static_cast<void (cuda::SURF_CUDA::*)(__VA_ARGS__)>(&cuda::SURF_CUDA::operator())
static const auto surf_op1 = SURF_CALL(const GpuMat &, const GpuMat &, GpuMat &);
static const auto surf_op2 = SURF_CALL(const GpuMat &, const GpuMat &, GpuMat &, GpuMat &, bool);
static const auto surf_op3 = SURF_CALL(const GpuMat &, const GpuMat &, std::vector<KeyPoint> &);
static const auto surf_op4 = SURF_CALL(const GpuMat &, const GpuMat &, std::vector<KeyPoint> &, GpuMat &, bool);
static const auto surf_op5 = SURF_CALL(const GpuMat &, const GpuMat &, std::vector<KeyPoint> &, std::vector<float> &, bool);
SAME(cuda::SURF_CUDA::hessianThreshold, double);
SAME(cuda::SURF_CUDA::nOctaves, int);
SAME(cuda::SURF_CUDA::nOctaveLayers, int);
SAME(cuda::SURF_CUDA::extended, bool);
SAME(cuda::SURF_CUDA::upright, bool);
SAME(cuda::SURF_CUDA::keypointsRatio, float);
static_assert(cuda::SURF_CUDA::X_ROW == 0 && cuda::SURF_CUDA::HESSIAN_ROW == 6 && cuda::SURF_CUDA::ROWS_COUNT == 7, "keypoint matrix rows, cuda.hpp:89-99");

// ---- cudafeatures2d.hpp:75-372 DescriptorMatcher
typedef cuda::DescriptorMatcher DM;
SAME(&DM::createBFMatcher, Ptr<DM> (*)(int));
SAME(&DM::isMaskSupported, bool (DM::*)() const);
SAME(&DM::add, void (DM::*)(const GpuMats &));
SAME(&DM::getTrainDescriptors, const GpuMats &(DM::*)() const);
SAME(&DM::train, void (DM::*)());
#define DM_FN(name, ...) static_cast<void (DM::*)(__VA_ARGS__)>(&DM::name)
static const auto dm_match1 = DM_FN(match, cuda::InputArray, cuda::InputArray, Matches &, cuda::InputArray);
static const auto dm_match2 = DM_FN(match, cuda::InputArray, Matches &, const GpuMats &);
static const auto dm_matchA1 = DM_FN(matchAsync, cuda::InputArray, cuda::InputArray, cuda::OutputArray, cuda::InputArray, Stream &);
static const auto dm_matchA2 = DM_FN(matchAsync, cuda::InputArray, cuda::OutputArray, const GpuMats &, Stream &);
static const auto dm_matchC = DM_FN(matchConvert, cuda::InputArray, Matches &);
static const auto dm_knn1 = DM_FN(knnMatch, cuda::InputArray, cuda::InputArray, MatchLists &, int, cuda::InputArray, bool);
static const auto dm_knn2 = DM_FN(knnMatch, cuda::InputArray, MatchLists &, int, const GpuMats &, bool);
static const auto dm_knnA1 = DM_FN(knnMatchAsync, cuda::InputArray, cuda::InputArray, cuda::OutputArray, int, cuda::InputArray, Stream &);
static const auto dm_knnA2 = DM_FN(knnMatchAsync, cuda::InputArray, cuda::OutputArray, int, const GpuMats &, Stream &);
static const auto dm_knnC = DM_FN(knnMatchConvert, cuda::InputArray, MatchLists &, bool);
static const auto dm_rad1 = DM_FN(radiusMatch, cuda::InputArray, cuda::InputArray, MatchLists &, float, cuda::InputArray, bool);
static const auto dm_rad2 = DM_FN(radiusMatch, cuda::InputArray, MatchLists &, float, const GpuMats &, bool);
static const auto dm_radA1 = DM_FN(radiusMatchAsync, cuda::InputArray, cuda::InputArray, cuda::OutputArray, float, cuda::InputArray, Stream &);
static const auto dm_radA2 = DM_FN(radiusMatchAsync, cuda::InputArray, cuda::OutputArray, float, const GpuMats &, Stream &);
static const auto dm_radC = DM_FN(radiusMatchConvert, cuda::InputArray, MatchLists &, bool);

// ---- superres/optical_flow.hpp:57-199 (the in-tree callers of the flow classes)
SAME(&superres::createOptFlow_Farneback_CUDA, Ptr<superres::FarnebackOpticalFlow> (*)());
SAME(&superres::createOptFlow_DualTVL1_CUDA, Ptr<superres::DualTVL1OpticalFlow> (*)());

// default arguments exist (a call with the fewest arguments the reference allows must compile); never executed
void defaults_compile_only()
{
    (void)sizeof(cuda::OpticalFlowDual_TVL1::create());
    (void)sizeof(cuda::FarnebackOpticalFlow::create());
    (void)sizeof(cuda::DensePyrLKOpticalFlow::create());
    (void)sizeof(cuda::SparsePyrLKOpticalFlow::create());
    (void)sizeof(cuda::createStereoBM());
    (void)sizeof(cuda::createStereoSGM());
    (void)sizeof(cuda::createDisparityBilateralFilter());
    (void)sizeof(cuda::SURF_CUDA::create(100.0));
    (void)sizeof(DM::createBFMatcher());
    (void)surf_op1; (void)surf_op2; (void)surf_op3; (void)surf_op4; (void)surf_op5;
    (void)dm_match1; (void)dm_match2; (void)dm_matchA1; (void)dm_matchA2; (void)dm_matchC; (void)dm_knn1; (void)dm_knn2; (void)dm_knnA1;
    (void)dm_knnA2; (void)dm_knnC; (void)dm_rad1; (void)dm_rad2; (void)dm_radA1; (void)dm_radA2; (void)dm_radC;
}

int main() { return 0; }
