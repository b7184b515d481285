/*
 * oracle/refshim/cvstub -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The smallest stand-in for the main-repo OpenCV headers (opencv/opencv, not under /root/reference) that lets the reference's
 * OWN CPU implementation of Dual TV-L1 -- modules/optflow/src/tvl1flow.cpp, compiled VERBATIM from where it lies -- build and
 * run here (oracle/Makefile.ref -> oracle/_ref/libref_cpu.so).  Everything arithmetic inside tvl1flow.cpp (centred / forward
 * gradients, divergence, thresholding, primal and dual updates with the f64 hypot, the serial float error sum, the loop
 * structure, level sizes, the 1/scaleStep multiplies) is then REFERENCE code.  What this stub supplies is container plumbing
 * (Mat, Mat_<T>, ROI views, Ptr, parallel_for_) and the main-repo functions the file calls -- cv::resize(INTER_LINEAR),
 * cv::remap(INTER_CUBIC), cv::medianBlur, convertTo, multiply, split, merge -- which forward to the restatements in
 * oracle/imgproc_ref.c (their arithmetic is not in /root/reference: restated from the library's documented behaviour, and
 * that part stays unpinned).
 */
#ifndef ORACLE_CVSTUB_CORE_HPP
#define ORACLE_CVSTUB_CORE_HPP
#include <cstddef>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_EXPORTS
#define CV_EXPORTS_W
#define CV_WRAP
#define CV_OVERRIDE override
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr); } while (0)
#define CV_DbgAssert(expr) CV_Assert(expr)
#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)

namespace cv {

typedef unsigned char uchar;
typedef std::string String;
enum { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2 };

struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
    int area() const { return width * height; }
    bool operator==(const Size &o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size &o) const { return !(*this == o); }
};
struct Rect { int x = 0, y = 0, width = 0, height = 0; Rect() {} Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {} };
struct Scalar {
    double val[4] = {0, 0, 0, 0};
    static Scalar all(double v) { Scalar s; s.val[0] = s.val[1] = s.val[2] = s.val[3] = v; return s; }
    double operator[](int i) const { return val[i]; }
};
struct Range { int start = 0, end = 0; Range() {} Range(int s, int e) : start(s), end(e) {} };

template <typename T> using Ptr = std::shared_ptr<T>;
template <typename T, typename... A> Ptr<T> makePtr(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }

class Algorithm {
public:
    virtual ~Algorithm() {}
    virtual String getDefaultName() const { return "my_object"; }
};

class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uchar *data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { return (size_t)channels() * (depth() == CV_8U ? 1 : 4); }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == nullptr || rows * cols == 0; }
    void create(int r, int c, int type)
    {
        if (data && rows == r && cols == c && type_ == type) return;
        type_ = type; rows = r; cols = c;
        step = (size_t)c * elemSize();
        buf_ = std::make_shared<std::vector<uchar> >((size_t)r * step + 64);
        data = buf_->data();
    }
    void create(Size s, int type) { create(s.height, s.width, type); }
    void release() { buf_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    template <typename T> T *ptr(int y) { return reinterpret_cast<T *>(data + (size_t)y * step); }
    template <typename T> const T *ptr(int y) const { return reinterpret_cast<const T *>(data + (size_t)y * step); }
    Mat operator()(const Rect &r) const
    {
        CV_Assert(r.x >= 0 && r.y >= 0 && r.x + r.width <= cols && r.y + r.height <= rows);
        Mat m = *this;
        m.data = data + (size_t)r.y * step + (size_t)r.x * elemSize();
        m.rows = r.height; m.cols = r.width;
        return m;
    }
    Mat &setTo(const Scalar &s);
    void convertTo(Mat &dst, int rtype, double alpha = 1.0) const;
    // InputArray / OutputArray facade (the stub's InputArray IS a Mat)
    const Mat &getMat() const { return *this; }
    Mat &getMat() { return *this; }
    bool isUMat() const { return false; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
protected:
    int type_ = CV_8UC1;
    std::shared_ptr<std::vector<uchar> > buf_;
};

template <typename T> struct DataDepth;
template <> struct DataDepth<float> { enum { type = CV_32FC1 }; };
template <> struct DataDepth<uchar> { enum { type = CV_8UC1 }; };

template <typename T>
class Mat_ : public Mat {
public:
    Mat_() { type_ = DataDepth<T>::type; }
    Mat_(const Mat &m) : Mat(m) { CV_Assert(m.empty() || m.type() == DataDepth<T>::type); type_ = DataDepth<T>::type; }
    Mat_(int r, int c) { type_ = DataDepth<T>::type; Mat::create(r, c, DataDepth<T>::type); }
    void create(int r, int c) { Mat::create(r, c, DataDepth<T>::type); }
    void create(Size s) { Mat::create(s.height, s.width, DataDepth<T>::type); }
    T *operator[](int y) { return reinterpret_cast<T *>(data + (size_t)y * step); }
    const T *operator[](int y) const { return reinterpret_cast<const T *>(data + (size_t)y * step); }
    Mat_ operator()(const Rect &r) const { return Mat_(Mat::operator()(r)); }
    T &operator()(int y, int x) { return (*this)[y][x]; }
    const T &operator()(int y, int x) const { return (*this)[y][x]; }
};

typedef const Mat &InputArray;
typedef Mat &OutputArray;
typedef Mat &InputOutputArray;

class ParallelLoopBody {
public:
    virtual ~ParallelLoopBody() {}
    virtual void operator()(const Range &range) const = 0;
};
// cv::parallel_for_: the range cut into stripes, one per thread (every body in tvl1flow.cpp is a loop over independent rows)
inline void parallel_for_(const Range &r, const ParallelLoopBody &body, double = -1.)
{
    const int n = r.end - r.start, stripes = n < 64 ? 1 : 64;
#pragma omp parallel for schedule(static)
    for (int s = 0; s < stripes; ++s) {
        const int a = r.start + (int)((long long)n * s / stripes), b = r.start + (int)((long long)n * (s + 1) / stripes);
        if (b > a) body(Range(a, b));
    }
}

// main-repo functions tvl1flow.cpp calls (cvstub.cpp)
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void remap(InputArray src, OutputArray dst, InputArray map1, InputArray map2, int interpolation);
void medianBlur(InputArray src, OutputArray dst, int ksize);
void multiply(InputArray src, const Scalar &s, OutputArray dst);
void split(const Mat &src, Mat_<float> *mv);
void merge(const Mat *mv, size_t count, OutputArray dst);

class DenseOpticalFlow : public Algorithm {   // main repo video/tracking.hpp
public:
    virtual void calc(InputArray I0, InputArray I1, InputOutputArray flow) = 0;
    virtual void collectGarbage() = 0;
};

}  // namespace cv
#endif
