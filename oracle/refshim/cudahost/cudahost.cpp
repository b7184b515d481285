/*
 * oracle/refshim/cudahost/cudahost.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See opencv2/core/cuda.hpp in this directory.
 * Host glue of the cv::cuda functions the reference's host class calls (restated from the cited lines; their device parts are
 * elementwise loops or the reference kernels of libref_cu.so), and the C entry point that drives
 * cv::cuda::OpticalFlowDual_TVL1 (modules/cudaoptflow/src/tvl1flow.cpp, compiled verbatim by oracle/Makefile.ref).
 */
#include "opencv2/cudaoptflow.hpp"      // the reference's own headers (-I $(REF)/modules/cudaoptflow/include, .../cudastereo/include)
#include "opencv2/cudastereo.hpp"
#include "opencv2/xfeatures2d/cuda.hpp"
#include "opencv2/cudaarithm.hpp"
#include "opencv2/cudawarping.hpp"
#include "opencv2/video.hpp"
#include <cmath>

extern "C" void ref_cu_resize_linear_f32(const float *src, int sr, int sc, float *dst, int dr, int dc, float fy, float fx);
extern "C" void ref_cu_pyr_down_f32(const float *src, int sr, int sc, float *dst, int dr, int dc);

namespace cv { namespace cuda {

// GpuMat::convertTo(dst, rtype, alpha, stream): core/src/cuda/gpu_mat.cu convertToScale -- 8U / 32F -> 32F computes in float:
// saturate_cast<float>(alpha * src [+ beta = 0]); alpha arrives as double and is used as a float scalar
void GpuMat::convertTo(GpuMat &dst, int rtype, double alpha, Stream &) const
{
    CV_Assert((rtype & 7) == CV_32F && channels() == 1);
    GpuMat out;
    out.create(rows, cols, CV_32FC1);
    const float a = (float)alpha;
    for (int y = 0; y < rows; ++y) {
        float *d = out.ptr<float>(y);
        if (depth() == CV_8U) { const unsigned char *s = ptr<unsigned char>(y); for (int x = 0; x < cols; ++x) d[x] = a * (float)s[x]; }
        else { const float *s = ptr<float>(y); for (int x = 0; x < cols; ++x) d[x] = a * s[x]; }
    }
    dst = out;
}
// GpuMat::convertTo(dst, rtype, stream): no scale -- 8U -> 32F is the exact conversion, 32F -> 32F a copy
void GpuMat::convertTo(GpuMat &dst, int rtype, Stream &stream) const
{
    CV_Assert((rtype & 7) == CV_32F && channels() == 1);
    if (depth() == CV_32F) { if (dst.data != data) copyTo(dst, stream); return; }
    GpuMat out;
    out.create(rows, cols, CV_32FC1);
    for (int y = 0; y < rows; ++y) {
        const unsigned char *s = ptr<unsigned char>(y);
        float *d = out.ptr<float>(y);
        for (int x = 0; x < cols; ++x) d[x] = (float)s[x];
    }
    dst = out;
}
GpuMat &GpuMat::setTo(Scalar s, Stream &)
{
    CV_Assert(type() == CV_32FC1);
    for (int y = 0; y < rows; ++y) { float *r = ptr<float>(y); for (int x = 0; x < cols; ++x) r[x] = (float)s[0]; }
    return *this;
}
void GpuMat::copyTo(GpuMat &dst, Stream &) const
{
    dst.create(rows, cols, type());
    if (dst.data == data) return;
    for (int y = 0; y < rows; ++y) memcpy(dst.ptr<unsigned char>(y), ptr<unsigned char>(y), (size_t)cols * elem_size_of(type()));
}
void GpuMat::download(Mat &dst) const
{
    if (dst.empty() || dst.rows != rows || dst.cols != cols || dst.type() != type()) dst.create(rows, cols, type());   // a user buffer of the right shape is kept
    for (int y = 0; y < rows; ++y) memcpy(dst.ptr<unsigned char>(y), ptr<unsigned char>(y), (size_t)cols * elem_size_of(type()));
}
void GpuMat::download(Mat &dst, Stream &) const { download(dst); }
void GpuMat::upload(const Mat &src)
{
    create(src.rows, src.cols, src.type());
    for (int y = 0; y < rows; ++y) memcpy(ptr<unsigned char>(y), src.ptr<unsigned char>(y), (size_t)cols * elem_size_of(type()));
}
GpuMat &GpuMat::setTo(Scalar s)
{
    const int d = depth();
    CV_Assert(d == CV_32F || d == CV_32S);
    for (int y = 0; y < rows; ++y) {
        float *rf = ptr<float>(y);
        int *ri = ptr<int>(y);
        for (int x = 0; x < cols * channels(); ++x) { if (d == CV_32F) rf[x] = (float)s[0]; else ri[x] = (int)s[0]; }
    }
    return *this;
}

// cuda::multiply(src, Scalar, dst, 1, -1): cudaarithm/src/cuda/mul_scalar.cu:62-70,118-175 -- for a CV_32F source the scalar type is
// float (the funcs table row {..., mulScalarImpl<float, float, float>, ...}) and the functor is saturate_cast<float>(scale * a * val)
// with scale = 1: one float multiply by (float)scalar
void multiply(const GpuMat &src1, const Scalar &src2, GpuMat &dst, double scale, int dtype, Stream &)
{
    CV_Assert(src1.type() == CV_32FC1 && scale == 1 && dtype == -1);
    GpuMat out = dst.data == src1.data ? dst : GpuMat();
    out.create(src1.rows, src1.cols, CV_32FC1);
    const float v = (float)src2[0];
    for (int y = 0; y < src1.rows; ++y) {
        const float *s = src1.ptr<float>(y);
        float *d = out.ptr<float>(y);
        for (int x = 0; x < src1.cols; ++x) d[x] = s[x] * v;
    }
    dst = out;
}

// cuda::merge of two CV_32FC1 planes into CV_32FC2 (cudaarithm/src/cuda/split_merge.cu: interleave)
void merge(const GpuMat *src, size_t n, OutputArray dst, Stream &)
{
    CV_Assert(n == 2 && src[0].type() == CV_32FC1 && src[1].size() == src[0].size());
    dst.create(src[0].size(), CV_32FC2);
    GpuMat &d = *dst.gpuMatPtr();
    for (int y = 0; y < d.rows; ++y) {
        const float *a = src[0].ptr<float>(y), *b = src[1].ptr<float>(y);
        float *o = d.ptr<float>(y);
        for (int x = 0; x < d.cols; ++x) { o[2 * x] = a[x]; o[2 * x + 1] = b[x]; }
    }
}

// cuda::split of a CV_32FC2 matrix into two CV_32FC1 planes (cudaarithm/src/cuda/split_merge.cu: de-interleave)
void split(InputArray _src, std::vector<GpuMat> &dst, Stream &)
{
    const GpuMat src = _src.getGpuMat();
    CV_Assert(src.type() == CV_32FC2);
    dst.resize(2);
    for (int c = 0; c < 2; ++c) dst[c].create(src.rows, src.cols, CV_32FC1);
    for (int y = 0; y < src.rows; ++y) {
        const float *s = src.ptr<float>(y);
        float *a = dst[0].ptr<float>(y), *b = dst[1].ptr<float>(y);
        for (int x = 0; x < src.cols; ++x) { a[x] = s[2 * x]; b[x] = s[2 * x + 1]; }
    }
}

// cuda::integral of a CV_8UC1 image: sum(y, x) = the sum over rows < y, columns < x, CV_32SC1, (rows + 1) x (cols + 1) -- exact
// integer arithmetic whatever the device's scan order (cudaarithm/src/cuda/integral.cu in the main repo)
void integral(InputArray _src, OutputArray _sum, Stream &)
{
    const GpuMat src = _src.getGpuMat();
    CV_Assert(src.type() == CV_8UC1);
    _sum.create(Size(src.cols + 1, src.rows + 1), CV_32SC1);
    GpuMat &sum = *_sum.gpuMatPtr();
    memset(sum.ptr<unsigned char>(0), 0, sizeof(unsigned) * (src.cols + 1));
    for (int y = 0; y < src.rows; ++y) {
        const unsigned char *s = src.ptr<unsigned char>(y);
        const unsigned *up = sum.ptr<unsigned>(y);
        unsigned *o = sum.ptr<unsigned>(y + 1);
        unsigned run = 0;
        o[0] = 0;
        for (int x = 0; x < src.cols; ++x) { run += s[x]; o[x + 1] = up[x + 1] + run; }
    }
}
// cuda::min(src CV_8UC1, scalar, dst)
void min(InputArray _src1, InputArray src2, OutputArray _dst, Stream &)
{
    const GpuMat src = _src1.getGpuMat();
    CV_Assert(src.type() == CV_8UC1 && src2.isScalar());
    _dst.create(src.size(), CV_8UC1);
    GpuMat &dst = *_dst.gpuMatPtr();
    const double v = src2.scalar();
    for (int y = 0; y < src.rows; ++y) {
        const unsigned char *s = src.ptr<unsigned char>(y);
        unsigned char *d = dst.ptr<unsigned char>(y);
        for (int x = 0; x < src.cols; ++x) d[x] = (double)s[x] < v ? s[x] : (unsigned char)v;
    }
}

// cuda::calcSum: cudaarithm/src/cuda/sum.cu -- a CV_32F source is reduced in double (the reduction tree's association is the
// device's; the value only feeds `error > scaledEpsilon`, a sequential double sum differs from any tree by ~1e-16 relative)
void calcSum(InputArray src, OutputArray dst, InputArray mask, Stream &)
{
    CV_Assert(mask.empty());
    const GpuMat s = src.getGpuMat();
    CV_Assert(s.type() == CV_32FC1);
    double acc = 0;
    for (int y = 0; y < s.rows; ++y) { const float *r = s.ptr<float>(y); for (int x = 0; x < s.cols; ++x) acc += (double)r[x]; }
    dst.create(Size(1, 1), CV_64FC1);
    *dst.gpuMatPtr()->ptr<double>(0) = acc;
}

// cuda::resize: cudawarping/src/resize.cpp:55-103 (the size / scale glue) around the reference's resize_linear kernel
static int saturate_int(double v) { return (int)lrint(v); }   // saturate_cast<int>(double) = cvRound: round half to even
void resize(InputArray _src, OutputArray _dst, Size dsize, double fx, double fy, int interpolation, Stream &stream)
{
    const GpuMat src = _src.getGpuMat();
    CV_Assert(src.type() == CV_32FC1 && interpolation == INTER_LINEAR);
    CV_Assert(!(dsize == Size()) || (fx > 0 && fy > 0));
    if (dsize == Size()) {
        dsize = Size(saturate_int(src.cols * fx), saturate_int(src.rows * fy));
    } else {
        fx = static_cast<double>(dsize.width) / src.cols;
        fy = static_cast<double>(dsize.height) / src.rows;
    }
    _dst.create(dsize, src.type());
    GpuMat &dst = *_dst.gpuMatPtr();
    if (dsize == src.size()) { src.copyTo(dst, stream); return; }
    // dense copies for the C entry of libref_cu.so (the kernel itself addresses through PtrStepSz: pitch-independent)
    std::vector<float> s((size_t)src.rows * src.cols), d((size_t)dsize.height * dsize.width);
    for (int y = 0; y < src.rows; ++y) memcpy(&s[(size_t)y * src.cols], src.ptr<float>(y), sizeof(float) * src.cols);
    ref_cu_resize_linear_f32(s.data(), src.rows, src.cols, d.data(), dsize.height, dsize.width, static_cast<float>(1.0 / fy), static_cast<float>(1.0 / fx));
    for (int y = 0; y < dsize.height; ++y) memcpy(dst.ptr<float>(y), &d[(size_t)y * dsize.width], sizeof(float) * dsize.width);
}

// cuda::pyrDown: cudawarping/src/pyramids.cpp:60-88 (dst size) around the reference's pyrDown kernel
void pyrDown(InputArray _src, OutputArray _dst, Stream &)
{
    const GpuMat src = _src.getGpuMat();
    CV_Assert(src.type() == CV_32FC1);
    _dst.create(Size((src.cols + 1) / 2, (src.rows + 1) / 2), src.type());
    GpuMat &dst = *_dst.gpuMatPtr();
    std::vector<float> s((size_t)src.rows * src.cols), d((size_t)dst.rows * dst.cols);
    for (int y = 0; y < src.rows; ++y) memcpy(&s[(size_t)y * src.cols], src.ptr<float>(y), sizeof(float) * src.cols);
    ref_cu_pyr_down_f32(s.data(), src.rows, src.cols, d.data(), dst.rows, dst.cols);
    for (int y = 0; y < dst.rows; ++y) memcpy(dst.ptr<float>(y), &d[(size_t)y * dst.cols], sizeof(float) * dst.cols);
}
}}  // namespace cv::cuda

extern "C" {
/* cv::cuda::FarnebackOpticalFlow::create(...)->calc(I0, I1, flow): the reference host class (modules/cudaoptflow/src/farneback.cpp,
 * verbatim) over the reference kernels.  Frames as in ref_cuhost_tvl1_calc; flags & OPTFLOW_USE_INITIAL_FLOW reads `flow`.
 * Returns 0, or 1 if the class threw. */
int ref_cuhost_farneback_calc(int num_levels, double pyr_scale, int fast_pyramids, int win_size, int num_iters, int poly_n, double poly_sigma,
                              int flags, const void *I0, const void *I1, int type, int cols, int rows, float *flow)
{
    using namespace cv;
    try {
        Ptr<cuda::FarnebackOpticalFlow> alg = cuda::FarnebackOpticalFlow::create(num_levels, pyr_scale, fast_pyramids != 0, win_size, num_iters,
                                                                                 poly_n, poly_sigma, flags);
        const int t = type == 0 ? CV_8UC1 : CV_32FC1;
        cuda::GpuMat a(Size(cols, rows), t), b(Size(cols, rows), t), f;
        const size_t rb = (size_t)cols * elem_size_of(t);
        for (int y = 0; y < rows; ++y) {
            memcpy(a.ptr<unsigned char>(y), (const unsigned char *)I0 + y * rb, rb);
            memcpy(b.ptr<unsigned char>(y), (const unsigned char *)I1 + y * rb, rb);
        }
        if (flags & OPTFLOW_USE_INITIAL_FLOW) {
            f.create(Size(cols, rows), CV_32FC2);
            for (int y = 0; y < rows; ++y) memcpy(f.ptr<float>(y), flow + (size_t)y * cols * 2, sizeof(float) * cols * 2);
        }
        alg->calc(a, b, f, cuda::Stream::Null());
        CV_Assert(f.rows == rows && f.cols == cols && f.type() == CV_32FC2);
        for (int y = 0; y < rows; ++y) memcpy(flow + (size_t)y * cols * 2, f.ptr<float>(y), sizeof(float) * cols * 2);
        return 0;
    } catch (const std::exception &) {
        return 1;
    }
}

/* cv::cuda::OpticalFlowDual_TVL1::create(...)->calc(I0, I1, flow): the reference host class over the reference kernels.
 * I0 / I1: rows x cols, type 0 = CV_8UC1, 1 = CV_32FC1 (dense); flow: rows x cols x 2 floats; *nscales_out = nscales after the call
 * (the class shrinks it for small images).  Returns 0, or 1 if the class threw. */
int ref_cuhost_tvl1_calc(double tau, double lambda, double theta, int nscales, int warps, double epsilon, int iterations, double scale_step,
                         double gamma, int use_initial_flow, const void *I0, const void *I1, int type, int cols, int rows, float *flow,
                         int *nscales_out)
{
    using namespace cv;
    try {
        Ptr<cuda::OpticalFlowDual_TVL1> alg = cuda::OpticalFlowDual_TVL1::create(tau, lambda, theta, nscales, warps, epsilon, iterations, scale_step,
                                                                                 gamma, use_initial_flow != 0);
        const int t = type == 0 ? CV_8UC1 : CV_32FC1;
        cuda::GpuMat a(Size(cols, rows), t), b(Size(cols, rows), t), f;
        const size_t rb = (size_t)cols * elem_size_of(t);
        for (int y = 0; y < rows; ++y) {
            memcpy(a.ptr<unsigned char>(y), (const unsigned char *)I0 + y * rb, rb);
            memcpy(b.ptr<unsigned char>(y), (const unsigned char *)I1 + y * rb, rb);
        }
        if (use_initial_flow) {
            f.create(Size(cols, rows), CV_32FC2);
            for (int y = 0; y < rows; ++y) memcpy(f.ptr<float>(y), flow + (size_t)y * cols * 2, sizeof(float) * cols * 2);
        }
        alg->calc(a, b, f, cuda::Stream::Null());
        for (int y = 0; y < rows; ++y) memcpy(flow + (size_t)y * cols * 2, f.ptr<float>(y), sizeof(float) * cols * 2);
        if (nscales_out) *nscales_out = alg->getNumScales();
        return 0;
    } catch (const std::exception &) {
        return 1;
    }
}

/* cv::cuda::createStereoBM(ndisp, block)->compute(left, right, disp): the reference host class (modules/cudastereo/src/stereobm.cpp, verbatim)
 * over the reference kernels (stereobm.cu).  prefilter_type -1 = none (the constructor's preset_), 0 / 1 = cv::StereoBM::PREFILTER_*.
 * Returns 0, or 1 if the class threw (its CV_Asserts on ndisp / block size / types). */
int ref_cuhost_stereobm_compute(int ndisp, int block, int prefilter_type, int prefilter_size, int prefilter_cap, int texture_threshold,
                                int uniqueness_ratio, const unsigned char *left, const unsigned char *right, int cols, int rows,
                                unsigned char *disp)
{
    using namespace cv;
    try {
        Ptr<cuda::StereoBM> bm = cuda::createStereoBM(ndisp, block);
        if (prefilter_type >= 0) bm->setPreFilterType(prefilter_type);
        if (prefilter_size >= 0) bm->setPreFilterSize(prefilter_size);
        if (prefilter_cap >= 0) bm->setPreFilterCap(prefilter_cap);
        if (texture_threshold >= 0) bm->setTextureThreshold(texture_threshold);
        if (uniqueness_ratio >= 0) bm->setUniquenessRatio(uniqueness_ratio);
        cuda::GpuMat l(Size(cols, rows), CV_8UC1), r(Size(cols, rows), CV_8UC1), d;
        for (int y = 0; y < rows; ++y) {
            memcpy(l.ptr<unsigned char>(y), left + (size_t)y * cols, cols);
            memcpy(r.ptr<unsigned char>(y), right + (size_t)y * cols, cols);
        }
        for (int rep = 0; rep < 2; ++rep)   // twice: the second call reuses minSSD_ / leBuf_ / riBuf_ (ensureSizeIsEnough keeps them)
            bm->compute(l, r, d, cuda::Stream::Null());
        CV_Assert(d.rows == rows && d.cols == cols && d.type() == CV_8UC1);
        for (int y = 0; y < rows; ++y) memcpy(disp + (size_t)y * cols, d.ptr<unsigned char>(y), cols);
        return 0;
    } catch (const std::exception &) {
        return 1;
    }
}

/* cv::cuda::SURF_CUDA::create(...) then operator()(img, mask, keypoints[, descriptors, useProvidedKeypoints]): the reference host class
 * (modules/xfeatures2d/src/surf.cuda.cpp, verbatim: SURF_CUDA_Invoker and the upload / download helpers) over the reference kernels
 * (xfeatures2d/src/cuda/surf.cu on the fiber shim).  mask may be NULL.  kp: ROWS_COUNT (7) rows of `cap` 32-bit words, the GpuMat
 * rows as they are (X, Y, LAPLACIAN [int], OCTAVE [int], SIZE, ANGLE, HESSIAN); with use_provided the first n_in columns are the
 * caller's keypoints in that layout.  desc: cap x descriptorSize floats (NULL: detect only).  Returns the number of features, -1 if
 * the class threw, -2 if cap is too small.  Feature ORDER is that of the fibers' atomicInc calls -- as arbitrary as on a GPU. */
int ref_cuhost_surf(double hessian_threshold, int n_octaves, int n_octave_layers, int extended, float keypoints_ratio, int upright,
                    const unsigned char *img, const unsigned char *mask, int cols, int rows, int use_provided, int n_in, unsigned *kp, int cap,
                    float *desc)
{
    using namespace cv;
    try {
        Ptr<cuda::SURF_CUDA> surf = cuda::SURF_CUDA::create(hessian_threshold, n_octaves, n_octave_layers, extended != 0, keypoints_ratio, upright != 0);
        cuda::GpuMat g(Size(cols, rows), CV_8UC1), m, k, d;
        for (int y = 0; y < rows; ++y) memcpy(g.ptr<unsigned char>(y), img + (size_t)y * cols, cols);
        if (mask) {
            m.create(Size(cols, rows), CV_8UC1);
            for (int y = 0; y < rows; ++y) memcpy(m.ptr<unsigned char>(y), mask + (size_t)y * cols, cols);
        }
        if (use_provided) {
            k.create(cuda::SURF_CUDA::ROWS_COUNT, n_in, CV_32FC1);
            for (int r = 0; r < cuda::SURF_CUDA::ROWS_COUNT; ++r) memcpy(k.ptr<unsigned>(r), kp + (size_t)r * cap, sizeof(unsigned) * n_in);
        }
        if (desc) (*surf)(g, m, k, d, use_provided != 0);
        else (*surf)(g, m, k);
        const int n = k.cols;
        if (n > cap) return -2;
        for (int r = 0; r < cuda::SURF_CUDA::ROWS_COUNT && n > 0; ++r) memcpy(kp + (size_t)r * cap, k.ptr<unsigned>(r), sizeof(unsigned) * n);
        if (desc && n > 0) {
            CV_Assert(d.rows == n && d.cols == surf->descriptorSize() && d.type() == CV_32FC1);
            for (int i = 0; i < n; ++i) memcpy(desc + (size_t)i * d.cols, d.ptr<float>(i), sizeof(float) * d.cols);
        }
        return n;
    } catch (const std::exception &) {
        return -1;
    }
}

/* cv::cuda::createDisparityBilateralFilter(ndisp, radius, iters)->apply(disp, image, dst): the reference host class
 * (modules/cudastereo/src/disparity_bilateral_filter.cpp, verbatim: the colour / space weight tables, edge_disc / max_disc, the type
 * dispatch) over the reference kernel (disparity_bilateral_filter.cu).  disp_type 0 = CV_8UC1, 3 = CV_16SC1; channels 1 or 3.
 * Thresholds < 0 keep the constructor's values.  In place on `disp`.  Returns 0, or 1 if the class threw. */
int ref_cuhost_dbf_apply(int ndisp, int radius, int iters, double edge_threshold, double max_disc_threshold, double sigma_range, void *disp,
                         int disp_type, const unsigned char *img, int channels, int cols, int rows)
{
    using namespace cv;
    try {
        Ptr<cuda::DisparityBilateralFilter> f = cuda::createDisparityBilateralFilter(ndisp, radius, iters);
        if (edge_threshold >= 0) f->setEdgeThreshold(edge_threshold);
        if (max_disc_threshold >= 0) f->setMaxDiscThreshold(max_disc_threshold);
        if (sigma_range >= 0) f->setSigmaRange(sigma_range);
        const int dt = disp_type == 0 ? CV_8UC1 : CV_MAKETYPE(CV_16S, 1), it = channels == 3 ? CV_8UC3 : CV_8UC1;
        cuda::GpuMat d(Size(cols, rows), dt), im(Size(cols, rows), it), out;
        const size_t db = (size_t)cols * elem_size_of(dt), ib = (size_t)cols * elem_size_of(it);
        for (int y = 0; y < rows; ++y) {
            memcpy(d.ptr<unsigned char>(y), (const unsigned char *)disp + y * db, db);
            memcpy(im.ptr<unsigned char>(y), img + y * ib, ib);
        }
        f->apply(d, im, out, cuda::Stream::Null());
        CV_Assert(out.rows == rows && out.cols == cols && out.type() == dt);
        for (int y = 0; y < rows; ++y) memcpy((unsigned char *)disp + y * db, out.ptr<unsigned char>(y), db);
        return 0;
    } catch (const std::exception &) {
        return 1;
    }
}
}
