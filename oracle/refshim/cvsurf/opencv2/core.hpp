/* oracle/refshim/cvsurf: a stub of the main-repo headers (opencv/opencv: core, imgproc, features2d -- not under /root/reference) that is
 * just large enough to compile the reference's CPU SURF class, modules/xfeatures2d/src/surf.cpp, VERBATIM (oracle/Makefile.ref,
 * libref_surfcpu.so).  Test infrastructure only.  The class logic (layer plan, det / trace, 3 x 3 x 3 maxima, interpolation,
 * orientation, descriptor) is the reference's own code; the main-repo FUNCTIONS it calls -- integral, resize(INTER_AREA),
 * getGaussianKernel, phase / fastAtan2, Matx33f::solve -- are not in the reference tree and come from the restatements of
 * oracle/surfcpu_ref.c (exported there as orc_cv_*), i.e. the pin is of the class, not of those five functions. */
#ifndef MIFLOW_CVSURF_CORE_HPP
#define MIFLOW_CVSURF_CORE_HPP
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CV_EXPORTS
#define CV_EXPORTS_W
#define CV_WRAP
#define CV_OUT
#define CV_IN_OUT
#define CV_OVERRIDE override
#define CV_PI 3.1415926535897932384626433832795
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr); } while (0)
#define CV_DbgAssert(expr) CV_Assert(expr)
#define CV_Error(code, msg) throw std::runtime_error(msg)
#define CV_8U 0
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAT_DEPTH(t) ((t) & 7)
#define CV_MAT_CN(t) ((((t) >> 3) & 511) + 1)
#ifndef MIN
#define MIN(a, b) ((a) > (b) ? (b) : (a))
#define MAX(a, b) ((a) < (b) ? (b) : (a))
#endif
typedef unsigned char uchar;

extern "C" {   // oracle/surfcpu_ref.c: the restated main-repo functions
float orc_cv_fast_atan2(float y, float x);
void orc_cv_gauss_kernel(int n, double sigma, float *k);
void orc_cv_integral_u8(const unsigned char *img, long long step, int rows, int cols, int *sum);
int orc_cv_resize_area_u8(const unsigned char *src, int n, unsigned char *dst, int m);
int orc_cv_round(double v);
}

namespace cv {
typedef std::string String;
enum { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2, INTER_AREA = 3 };
enum { DECOMP_LU = 0 };
enum { NORM_L2 = 4 };
enum { COLOR_BGR2GRAY = 6 };
namespace Error { enum { StsNotImplemented = -213 }; }

inline int cvRound(double v) { return orc_cv_round(v); }
inline int cvFloor(double v) { return (int)std::floor(v); }
inline int cvCeil(double v) { return (int)std::ceil(v); }
inline float fastAtan2(float y, float x) { return orc_cv_fast_atan2(y, x); }

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Point {
    int x = 0, y = 0;
    Point() {}
    Point(int x_, int y_) : x(x_), y(y_) {}
    Point(const Point2f &p) : x(cvRound(p.x)), y(cvRound(p.y)) {}   // Point_<int>(Point_<float>): saturate_cast<int> = cvRound
};
struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size &o) const { return width == o.width && height == o.height; }
};
struct Range { int start = 0, end = 0; Range() {} Range(int s, int e) : start(s), end(e) {} };
struct KeyPoint {   // core/types.hpp
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
    KeyPoint() {}
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};
struct Vec3f {
    float val[3] = {0, 0, 0};
    Vec3f() {}
    Vec3f(float a, float b, float c) { val[0] = a; val[1] = b; val[2] = c; }
    float &operator[](int i) { return val[i]; }
    const float &operator[](int i) const { return val[i]; }
};
// Matx33f::solve(b, DECOMP_LU): core/operations.hpp Matx_FastSolveOp<_Tp, 3, 3, 1> -- Cramer's rule on one reciprocal of the determinant
struct Matx33f {
    float val[9];
    Matx33f(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float a8)
    { val[0] = a0; val[1] = a1; val[2] = a2; val[3] = a3; val[4] = a4; val[5] = a5; val[6] = a6; val[7] = a7; val[8] = a8; }
    float operator()(int i, int j) const { return val[i * 3 + j]; }
    Vec3f solve(const Vec3f &b, int method) const;
};
inline float det3(const Matx33f &a)   // core/matx.hpp: determinant of a 3 x 3
{
    return a(0, 0) * (a(1, 1) * a(2, 2) - a(2, 1) * a(1, 2)) - a(0, 1) * (a(1, 0) * a(2, 2) - a(2, 0) * a(1, 2)) +
           a(0, 2) * (a(1, 0) * a(2, 1) - a(2, 0) * a(1, 1));
}
inline Vec3f Matx33f::solve(const Vec3f &b, int method) const
{
    CV_Assert(method == DECOMP_LU);
    const Matx33f &a = *this;
    float d = det3(a);
    Vec3f x;
    if (d == 0) return x;   // the reference's solve() then leaves x zero-initialised
    d = 1 / d;
    x[0] = d * (b[0] * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) - a(0, 1) * (b[1] * a(2, 2) - a(1, 2) * b[2]) + a(0, 2) * (b[1] * a(2, 1) - a(1, 1) * b[2]));
    x[1] = d * (a(0, 0) * (b[1] * a(2, 2) - a(1, 2) * b[2]) - b[0] * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) + a(0, 2) * (a(1, 0) * b[2] - b[1] * a(2, 0)));
    x[2] = d * (a(0, 0) * (a(1, 1) * b[2] - b[1] * a(2, 1)) - a(0, 1) * (a(1, 0) * b[2] - b[1] * a(2, 0)) + b[0] * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0)));
    return x;
}

inline size_t elem_size_of(int type) { const int d = CV_MAT_DEPTH(type); return (size_t)CV_MAT_CN(type) * (d == CV_8U ? 1 : d == CV_16S ? 2 : d == CV_64F ? 8 : 4); }
class _OutputArray;
class Mat {   // a dense host matrix; copies share the buffer (views of a caller's memory own nothing)
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uchar *data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void *user, size_t step_ = 0)
        : rows(r), cols(c), step(step_ ? step_ : (size_t)c * elem_size_of(type)), data((uchar *)user), type_(type) {}
    void create(int r, int c, int type)
    {
        if (data && rows == r && cols == c && type_ == type) return;
        rows = r; cols = c; type_ = type; step = (size_t)c * elem_size_of(type);
        buf_ = std::make_shared<std::vector<uchar> >((size_t)r * step + 16, (uchar)0);
        data = buf_->data();
    }
    int type() const { return type_; }
    size_t elemSize() const { return elem_size_of(type_); }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    template <typename T> T &at(int y, int x) { return reinterpret_cast<T *>(data + (size_t)y * step)[x]; }
    template <typename T> const T &at(int y, int x) const { return reinterpret_cast<const T *>(data + (size_t)y * step)[x]; }
    template <typename T> T *ptr(int y = 0) { return reinterpret_cast<T *>(data + (size_t)y * step); }
    template <typename T> const T *ptr(int y = 0) const { return reinterpret_cast<const T *>(data + (size_t)y * step); }
    uchar *ptr(int y = 0) { return data + (size_t)y * step; }
    const uchar *ptr(int y = 0) const { return data + (size_t)y * step; }
    Mat rowRange(int r0, int r1) const { Mat m = *this; m.data = data + (size_t)r0 * step; m.rows = r1 - r0; return m; }
    Mat reshape(int cn, int new_rows) const
    {
        CV_Assert(cn == 1 && CV_MAT_CN(type_) == 1 && isContinuous() && new_rows > 0 && ((size_t)rows * cols) % new_rows == 0);
        Mat m = *this;
        m.rows = new_rows; m.cols = (int)((size_t)rows * cols / new_rows); m.step = (size_t)m.cols * elemSize();
        return m;
    }
    void copyTo(const _OutputArray &dst) const;
private:
    int type_ = CV_8U;
    std::shared_ptr<std::vector<uchar> > buf_;
};
class UMat {};

class _InputArray {
public:
    enum { NONE = 0, MAT = 1 << 16, STD_VECTOR = 3 << 16 };
    _InputArray() {}
    _InputArray(const Mat &m) : m_(const_cast<Mat *>(&m)) {}
    int kind() const { return m_ ? MAT : NONE; }
    int type() const { return m_ ? m_->type() : 0; }
    bool empty() const { return !m_ || m_->empty(); }
    bool isUMat() const { return false; }
    Mat getMat() const { return m_ ? *m_ : Mat(); }
protected:
    Mat *m_ = nullptr;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(Mat &m) : _InputArray(m) {}
    bool needed() const { return m_ != nullptr; }
    void create(int r, int c, int type) const { CV_Assert(m_); m_->create(r, c, type); }
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
inline const _OutputArray &noArray() { static _OutputArray a; return a; }
inline void Mat::copyTo(const _OutputArray &dst) const
{
    dst.create(rows, cols, type_);
    Mat d = dst.getMat();
    for (int y = 0; y < rows; ++y) memmove(d.ptr(y), ptr(y), (size_t)cols * elemSize());
}

template <typename T> using Ptr = std::shared_ptr<T>;
template <typename T, typename... A> Ptr<T> makePtr(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }

// persistence: only what the inline read() / write() of surf.hpp need in order to compile (never called)
class FileNode {
public:
    FileNode operator[](const char *) const { return FileNode(); }
    bool empty() const { return true; }
};
template <typename T> inline void operator>>(const FileNode &, T &) {}
class FileStorage {
public:
    bool isOpened() const { return false; }
};
template <typename T> inline FileStorage &operator<<(FileStorage &fs, const T &) { return fs; }

class Algorithm {
public:
    virtual ~Algorithm() {}
    virtual String getDefaultName() const { return "my_object"; }
};

typedef std::recursive_mutex Mutex;
class AutoLock {
public:
    explicit AutoLock(Mutex &m) : m_(m) { m_.lock(); }
    ~AutoLock() { m_.unlock(); }
private:
    Mutex &m_;
};
template <typename T> class AutoBuffer {
public:
    explicit AutoBuffer(size_t n) : v_(n) {}
    T *data() { return v_.data(); }
    operator T *() { return v_.data(); }
private:
    std::vector<T> v_;
};
class ParallelLoopBody {
public:
    virtual ~ParallelLoopBody() {}
    virtual void operator()(const Range &range) const = 0;
};
// OpenMP stripes (the main repo's parallel_for_ runs the body on sub-ranges from a thread pool): the class guards its shared keypoint
// vector with a Mutex and sorts it afterwards (KeypointGreater), so neither the striping nor the order of execution shows
inline void parallel_for_(const Range &r, const ParallelLoopBody &body, double = -1.)
{
    const int n = r.end - r.start;
    if (n <= 0) return;
    int stripes = 1;
#ifdef _OPENMP
    stripes = std::min(n, 4 * omp_get_max_threads());
#endif
    if (stripes <= 1) { body(r); return; }
#pragma omp parallel for schedule(dynamic, 1)
    for (int k = 0; k < stripes; ++k) {
        const int a = r.start + (int)((long long)n * k / stripes), b = r.start + (int)((long long)n * (k + 1) / stripes);
        if (b > a) body(Range(a, b));
    }
}

namespace ocl { inline bool useOpenCL() { return false; } }

// ---- imgproc / core functions the class calls (main repo; restated in oracle/surfcpu_ref.c, see the header of this file)
Mat getGaussianKernel(int n, double sigma, int ktype);
void integral(InputArray src, Mat &sum, int sdepth);
void resize(const Mat &src, Mat &dst, Size dsize, double fx, double fy, int interpolation);
void phase(const Mat &x, const Mat &y, const Mat &angle, bool angleInDegrees);
void cvtColor(const Mat &src, Mat &dst, int code);
void min(const Mat &src, double v, Mat &dst);
}  // namespace cv
#endif
