/*
 * oracle/refshim/cudahost -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The smallest stand-in for the main-repo headers (opencv/opencv: core.hpp, core/cuda.hpp, core/private.cuda.hpp; absent from
 * /root/reference) that lets the reference's OWN host class of cv::cuda::OpticalFlowDual_TVL1 --
 * modules/cudaoptflow/src/tvl1flow.cpp, compiled VERBATIM from where it lies, against the reference's OWN public header
 * modules/cudaoptflow/include/opencv2/cudaoptflow.hpp -- build and run on the CPU, driving the reference's OWN kernels
 * (modules/cudaoptflow/src/cuda/tvl1flow.cu and the resize_linear template of modules/cudawarping/src/cuda/resize.cu, both already
 * in oracle/_ref/libref_cu.so).  calc / calcImpl / procOneScale -- the pyramid loop, the level sizes and the 16-px rule, the
 * per-warp loop with cv::cuda's sparse convergence schedule, the flow upsampling and its 1 / scaleStep multiplies -- are then
 * REFERENCE code end to end (VERDICT r02 missing #4).  What this stub supplies: GpuMat as host memory, Stream / BufferPool as
 * no-ops, InputArray proxies, and the five cv::cuda functions the file calls (cudahost.cpp: convertTo, setTo, multiply, merge,
 * calcSum, resize -- host glue of cudaarithm / cudawarping / core restated from the cited lines; their kernels are elementwise).
 *
 * The same for cv::cuda::FarnebackOpticalFlow -- modules/cudaoptflow/src/farneback.cpp, verbatim, over the reference's farneback.cu,
 * resize.cu and pyr_down.cu kernels: calc / calcImpl (level cropping, the per-level blur + resize or the pyrDown pyramid, the flow
 * upsampling, prepareGaussian, the iteration loop) are reference code.  Added for it: Event, Stream's bool / waitEvent, Mat_<double>
 * with the Cholesky inverse of a 6 x 6 matrix (main repo core/src/matrix_decomp.cpp CholImpl, restated), getGaussianKernel (main
 * repo imgproc/src/smooth.dispatch.cpp, restated), cuda::split / pyrDown.
 */
#ifndef ORACLE_CUDAHOST_CORE_CUDA_HPP
#define ORACLE_CUDAHOST_CORE_CUDA_HPP
#include "../../../cudashim/opencv2/core/cuda/common.hpp"   // PtrStepSz / cudaStream_t exactly as the kernels of libref_cu.so were built with
#include <cfloat>
#include <cmath>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#define HAVE_CUDA 1
#define CV_EXPORTS
#define CV_EXPORTS_W
#define CV_WRAP
#define CV_OUT
#define CV_IN_OUT
#define CV_WRAP_AS(x)
#define CV_PROP
#define CV_PROP_RW
#define CV_OVERRIDE override
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr); } while (0)
#define CV_DbgAssert(expr) CV_Assert(expr)
#define CV_8U 0
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32SC4 CV_MAKETYPE(CV_32S, 4)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

namespace cv {
typedef std::string String;
enum { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2 };
enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT101 = 4 };
enum { NORM_INF = 1, NORM_L1 = 2, NORM_L2 = 4 };
enum { DECOMP_LU = 0, DECOMP_SVD = 1, DECOMP_EIG = 2, DECOMP_CHOLESKY = 3 };
inline int cvRound(double v) { return (int)lrint(v); }   // round half to even, like the SSE2 path of core/fast_math.hpp
struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
    int area() const { return width * height; }
    bool operator==(const Size &o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size &o) const { return !(*this == o); }
};
struct Point2f { float x = 0, y = 0; };
struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1; };   // core/types.hpp
class Mutex { public: void lock() {} void unlock() {} };
class AutoLock { public: explicit AutoLock(Mutex &) {} };
struct Rect { int x = 0, y = 0, width = 0, height = 0; Rect() {} Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {} };
struct Scalar {
    double val[4] = {0, 0, 0, 0};
    Scalar() {}
    Scalar(double v0) { val[0] = v0; }
    static Scalar all(double v) { Scalar s; s.val[0] = s.val[1] = s.val[2] = s.val[3] = v; return s; }
    double operator[](int i) const { return val[i]; }
};
template <typename T> using Ptr = std::shared_ptr<T>;
template <typename T, typename... A> Ptr<T> makePtr(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }
class Algorithm {
public:
    virtual ~Algorithm() {}
    virtual String getDefaultName() const { return "my_object"; }
};
inline size_t elem_size_of(int type) { const int d = type & 7, cn = (type >> 3) + 1; return (size_t)cn * (d == CV_8U ? 1 : d == CV_16S ? 2 : d == CV_64F ? 8 : 4); }

namespace cuda { class GpuMat; }
class Mat {   // a dense host matrix: what diff_sum_host (1 x 1 CV_64F), getGaussianKernel (n x 1 CV_32F) and SURF_CUDA's keypoint /
public:       // descriptor transfers (7 x n CV_32F, n x 64|128 CV_32F over a caller's buffer) need
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char *data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type, void *user) : rows(s.height), cols(s.width), step((size_t)s.width * elem_size_of(type)), data((unsigned char *)user), type_(type) {}
    explicit Mat(const cuda::GpuMat &m);   // downloads (core/src/cuda_gpu_mat... Mat::Mat(const GpuMat&))
    void create(int r, int c, int type)
    {
        rows = r; cols = c; type_ = type; step = (size_t)c * elem_size_of(type);
        buf_ = std::make_shared<std::vector<unsigned char> >((size_t)r * step + 16);
        data = buf_->data();
    }
    int type() const { return type_; }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == nullptr; }
    template <typename T> T &at(int y, int x) { return reinterpret_cast<T *>(data + (size_t)y * step)[x]; }
    template <typename T> T *ptr(int y = 0) { return reinterpret_cast<T *>(data + (size_t)y * step); }
    template <typename T> const T *ptr(int y = 0) const { return reinterpret_cast<const T *>(data + (size_t)y * step); }
private:
    int type_ = CV_8UC1;
    std::shared_ptr<std::vector<unsigned char> > buf_;
};
// Mat_<double>: the 6 x 6 normal matrix of FarnebackOpticalFlowImpl::prepareGaussian and its inverse
template <typename T> class Mat_ {
public:
    int rows = 0, cols = 0;
    std::vector<T> v;
    Mat_() {}
    Mat_(int r, int c) : rows(r), cols(c), v((size_t)r * c) {}
    void setTo(T s) { for (auto &e : v) e = s; }
    T &operator()(int y, int x) { return v[(size_t)y * cols + x]; }
    const T &operator()(int y, int x) const { return v[(size_t)y * cols + x]; }
    // Mat::inv(DECOMP_CHOLESKY) of an n x n matrix, n > 3 (core/src/lapack.cpp cv::invert: dst = I, then hal::Cholesky(src, dst) solves
    // L L^T X = I in place; core/src/matrix_decomp.cpp CholImpl: L stored with RECIPROCAL diagonal, sums accumulated in double)
    Mat_ inv(int method) const
    {
        CV_Assert(method == DECOMP_CHOLESKY && rows == cols);
        const int m = rows;
        Mat_ A = *this, B(m, m);
        B.setTo(0);
        for (int i = 0; i < m; ++i) B(i, i) = 1;
        for (int i = 0; i < m; ++i) {
            for (int j = 0; j < i; ++j) {
                double s = A(i, j);
                for (int k = 0; k < j; ++k) s -= A(i, k) * A(j, k);
                A(i, j) = s * A(j, j);
            }
            double s = A(i, i);
            for (int k = 0; k < i; ++k) { const double t = A(i, k); s -= t * t; }
            if (s < std::numeric_limits<T>::epsilon()) { B.setTo(0); return B; }   // cv::invert returns 0 and zeroes dst
            A(i, i) = 1. / std::sqrt(s);
        }
        for (int i = 0; i < m; ++i)          // L y = b
            for (int j = 0; j < m; ++j) {
                double s = B(i, j);
                for (int k = 0; k < i; ++k) s -= A(i, k) * B(k, j);
                B(i, j) = s * A(i, i);
            }
        for (int i = m - 1; i >= 0; --i)     // L^T x = y
            for (int j = 0; j < m; ++j) {
                double s = B(i, j);
                for (int k = m - 1; k > i; --k) s -= A(k, i) * B(k, j);
                B(i, j) = s * A(i, i);
            }
        return B;
    }
};
// cv::getGaussianKernel(n, sigma, CV_32F) (imgproc.hpp; main repo imgproc/src/smooth.dispatch.cpp): the fixed tables for n <= 7 with
// sigma <= 0, else exp(-x^2 / (2 sigma^2)) over x = i - (n - 1) / 2, normalised in double, rounded to float
inline Mat getGaussianKernel(int n, double sigma, int ktype)
{
    CV_Assert(ktype == CV_32F && n > 0);
    Mat k(n, 1, CV_32FC1);
    float *kf = k.ptr<float>(0);
    static const float small[4][7] = {{1.f}, {0.25f, 0.5f, 0.25f}, {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
                                      {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    if (sigma <= 0 && (n & 1) && n <= 7) {
        for (int i = 0; i < n; ++i) kf[i] = small[n >> 1][i];
        return k;
    }
    const double sx = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8, s2 = -0.5 / (sx * sx);
    std::vector<double> w(n);
    double sum = 0;
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; w[i] = std::exp(s2 * x * x); sum += w[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) kf[i] = (float)(w[i] * sum);
    return k;
}

namespace cuda {
enum FeatureSet { FEATURE_SET_COMPUTE_10 = 10, FEATURE_SET_COMPUTE_11 = 11, FEATURE_SET_COMPUTE_12 = 12, FEATURE_SET_COMPUTE_13 = 13 };
inline bool deviceSupports(FeatureSet) { return true; }
class Event;
class Stream {   // every call of this stub runs synchronously on the host: the streams only order work that is already ordered
public:
    void waitForCompletion() {}
    void waitEvent(const Event &) {}
    explicit operator bool() const { return false; }   // "the default stream": FarnebackOpticalFlowImpl::calcImpl then skips its event plumbing
    static Stream &Null() { static Stream s; return s; }
};
class Event {
public:
    void record(Stream & = Stream::Null()) {}
};
class GpuMat {   // host memory; views share the buffer (like the reference's refcounted device buffers)
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char *data = nullptr;
    GpuMat() {}
    GpuMat(Size s, int type) { create(s, type); }
    GpuMat(int r, int c, int type) { create(r, c, type); }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == nullptr; }
    void create(int r, int c, int type)   // GpuMat::create: keeps the buffer when size and type match (core/src/cuda/gpu_mat.cu)
    {
        if (data && rows == r && cols == c && type_ == type) return;
        type_ = type; rows = r; cols = c;
        step = ((size_t)c * elem_size_of(type) + 255) / 256 * 256;   // a pitched allocation, like cudaMallocPitch
        buf_ = std::make_shared<std::vector<unsigned char> >((size_t)r * step + 64);
        data = buf_->data();
    }
    void create(Size s, int type) { create(s.height, s.width, type); }
    GpuMat operator()(const Rect &r) const
    {
        CV_Assert(r.x >= 0 && r.y >= 0 && r.x + r.width <= cols && r.y + r.height <= rows);
        GpuMat m = *this;
        m.data = data + (size_t)r.y * step + (size_t)r.x * elem_size_of(type_);
        m.rows = r.height; m.cols = r.width;
        return m;
    }
    template <typename T> T *ptr(int y = 0) { return reinterpret_cast<T *>(data + (size_t)y * step); }
    template <typename T> const T *ptr(int y = 0) const { return reinterpret_cast<const T *>(data + (size_t)y * step); }
    GpuMat row(int y) const { return (*this)(Rect(0, y, cols, 1)); }
    void release() { *this = GpuMat(); }
    template <typename T> operator PtrStep<T>() const { return PtrStep<T>((T *)data, step); }
    template <typename T> operator PtrStepSz<T>() const { return PtrStepSz<T>(rows, cols, (T *)data, step); }
    // cudahost.cpp
    void convertTo(GpuMat &dst, int rtype, double alpha, Stream &stream) const;
    void convertTo(GpuMat &dst, int rtype, Stream &stream) const;
    GpuMat &setTo(Scalar s, Stream &stream);
    GpuMat &setTo(Scalar s);                      // any 4-byte-element type (CV_32F / CV_32S, any channel count): s[0] everywhere is all SURF_CUDA asks
    void download(Mat &dst, Stream &stream) const;
    void download(Mat &dst) const;
    void upload(const Mat &src);
    void copyTo(GpuMat &dst, Stream &stream) const;
private:
    int type_ = CV_8UC1;
    std::shared_ptr<std::vector<unsigned char> > buf_;
};
inline void swap(GpuMat &a, GpuMat &b) { GpuMat t = a; a = b; b = t; }
// cuda::ensureSizeIsEnough (core/src/cuda/gpu_mat.cu... core/cuda.inl.hpp): keep a buffer that is large enough and take its top-left view
inline void ensureSizeIsEnough(Size s, int type, GpuMat &m)
{
    if (!m.empty() && m.type() == type && m.rows >= s.height && m.cols >= s.width) m = m(Rect(0, 0, s.width, s.height));
    else m.create(s, type);
}
inline void ensureSizeIsEnough(int rows, int cols, int type, GpuMat &m) { ensureSizeIsEnough(Size(cols, rows), type, m); }
}  // namespace cuda
inline Mat::Mat(const cuda::GpuMat &m) { m.download(*this); }

// InputArray / OutputArray proxies over GpuMat (the only kind this translation unit passes)
class _InputArray {
public:
    _InputArray() : m_(nullptr) {}
    _InputArray(const cuda::GpuMat &m) : m_(const_cast<cuda::GpuMat *>(&m)) {}
    _InputArray(const Mat &) : m_(nullptr) {}   // cudastereo.hpp:360 casts a Mat Q; never called here
    _InputArray(const double &v) : m_(nullptr), scalar_(v), is_scalar_(true) {}   // cuda::min(mask, 1.0, dst) of surf.cuda.cpp:164
    bool isScalar() const { return is_scalar_; }
    double scalar() const { return scalar_; }
    cuda::GpuMat getGpuMat() const { return m_ ? *m_ : cuda::GpuMat(); }
    bool empty() const { return !m_ || m_->empty(); }
    cuda::GpuMat *gpuMatPtr() const { return m_; }
protected:
    cuda::GpuMat *m_;
    double scalar_ = 0;
    bool is_scalar_ = false;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(cuda::GpuMat &m) : _InputArray(m) {}
    void create(Size s, int type) const { CV_Assert(m_); m_->create(s, type); }
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
typedef const _OutputArray &InputOutputArray;
typedef InputArray InputArrayOfArrays;
typedef OutputArray OutputArrayOfArrays;
inline const _OutputArray &noArray() { static _OutputArray a; return a; }
}  // namespace cv
#endif
