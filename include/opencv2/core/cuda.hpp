// Minimal stand-in for opencv2/core.hpp + opencv2/core/cuda.hpp (main repo, NOT under the reference tree)
// so that the miflow drop-in headers compile with no OpenCV installed.  When real OpenCV headers are
// available define MIFLOW_WITH_OPENCV and include them first: this file then only checks the layout.
//
// Layout contract: cv::cuda::GpuMat = {int flags; int rows, cols; size_t step; uchar* data;
// int* refcount; uchar* datastart; const uchar* dataend; Allocator* allocator;}, pitched row-major,
// element (y,x) at data + y*step + x*elemSize(); ROI views share datastart/dataend/refcount.
#ifndef MIFLOW_OPENCV_CORE_CUDA_SHIM_HPP
#define MIFLOW_OPENCV_CORE_CUDA_SHIM_HPP

#include <cstddef>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "miflow/c_api.h"

#ifndef MIFLOW_WITH_OPENCV

#define CV_CN_SHIFT 3
#define CV_8U 0
#define CV_32S 4
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32SC4 CV_MAKETYPE(CV_32S, 4)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)
#define CV_MAT_DEPTH(t) ((t) & 7)
#define CV_MAT_CN(t) ((((t) >> CV_CN_SHIFT) & 63) + 1)
#define CV_EXPORTS
#define CV_EXPORTS_W
#define CV_WRAP
#define CV_OVERRIDE override

namespace cv {

typedef unsigned char uchar;
typedef std::string String;

namespace Error {
enum Code { StsOk = 0, StsError = -2, StsBadArg = -5, StsNoMem = -4, StsNotImplemented = -213, StsUnsupportedFormat = -210,
            StsUnmatchedSizes = -209, StsAssert = -215, GpuNotSupported = -216, GpuApiCallError = -217 };
}

class Exception : public std::exception {
public:
    Exception(int c, const String &m) : code(c), err(m), msg("OpenCV(miflow) error: " + m) {}
    const char *what() const noexcept override { return msg.c_str(); }
    int code;
    String err, msg;
};

#define CV_Error(code_, msg_) throw ::cv::Exception(code_, msg_)
#define CV_Assert(expr) do { if (!(expr)) throw ::cv::Exception(::cv::Error::StsAssert, #expr); } while (0)

struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size &o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size &o) const { return !(*this == o); }
    int area() const { return width * height; }
};
struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() {}
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};

template <typename T> using Ptr = std::shared_ptr<T>;
template <typename T, typename... A> Ptr<T> makePtr(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }

class Algorithm {
public:
    virtual ~Algorithm() {}
    virtual void clear() {}
    virtual bool empty() const { return false; }
    virtual String getDefaultName() const { return "my_object"; }
};

namespace cuda {

// maps mi_status to the exception the reference would throw (CV_Assert / CV_Error / cudaSafeCall)
inline void miCheck(int rc)
{
    if (rc == MI_OK) return;
    const String m = mi_last_error();
    switch (rc) {
    case MI_ERR_BAD_ARG: throw Exception(Error::StsBadArg, m);
    case MI_ERR_BAD_TYPE: throw Exception(Error::StsAssert, m);
    case MI_ERR_BAD_SIZE: throw Exception(Error::StsAssert, m);
    case MI_ERR_OOM: throw Exception(Error::StsNoMem, m);
    case MI_ERR_NOT_IMPL: throw Exception(Error::StsNotImplemented, m);
    case MI_ERR_NO_DEVICE: throw Exception(Error::GpuNotSupported, m);   // throw_no_cuda() analogue
    default: throw Exception(Error::GpuApiCallError, m);
    }
}

inline int getCudaEnabledDeviceCount() { return mi_device_count(); }
inline void setDevice(int d) { miCheck(mi_set_device(d)); }
inline int getDevice() { int d = 0; miCheck(mi_get_device(&d)); return d; }
namespace miflow {
/** miflow extension: return the scratch blocks that destroyed handles left in the library's cache to the driver (the counterpart of
 * BufferPool's release; call it when another allocator of the process runs out of device memory). */
inline void releaseCachedMemory() { miCheck(mi_release_cached_memory()); }
}  // namespace miflow

class Stream {
public:
    Stream() : impl_(std::make_shared<Impl>(true)) {}
    static Stream &Null() { static Stream s(nullptr); return s; }
    void waitForCompletion() { miCheck(mi_stream_synchronize(impl_->s)); }
    void *hipStream() const { return impl_->s; }   // StreamAccessor::getStream analogue
private:
    struct Impl {
        void *s = nullptr;
        bool own = false;
        explicit Impl(bool create) { if (create) { miCheck(mi_stream_create(&s)); own = true; } }
        ~Impl() { if (own && s) mi_stream_destroy(s); }
    };
    explicit Stream(std::nullptr_t) : impl_(std::make_shared<Impl>(false)) {}
    std::shared_ptr<Impl> impl_;
};
struct StreamAccessor { static void *getStream(const Stream &s) { return s.hipStream(); } };

class GpuMat {
public:
    class Allocator;
    GpuMat() {}
    GpuMat(int rows_, int cols_, int type_) { create(rows_, cols_, type_); }
    GpuMat(Size sz, int type_) { create(sz.height, sz.width, type_); }
    GpuMat(const GpuMat &m) { copyHeader(m); if (refcount) ++*refcount; }
    GpuMat(const GpuMat &m, Rect roi) { copyHeader(m); if (refcount) ++*refcount; applyRoi(roi); }
    ~GpuMat() { release(); }
    GpuMat &operator=(const GpuMat &m)
    {
        if (this != &m) { GpuMat t(m); swap(t); }
        return *this;
    }
    void swap(GpuMat &o)
    {
        GpuMat *a = this;
        char tmp[sizeof(GpuMat)];
        std::memcpy(tmp, (void *)a, sizeof(GpuMat)); std::memcpy((void *)a, (void *)&o, sizeof(GpuMat)); std::memcpy((void *)&o, tmp, sizeof(GpuMat));
    }
    void create(int rows_, int cols_, int type_)
    {
        type_ &= 0xFFF;
        if (rows == rows_ && cols == cols_ && type() == type_ && data) return;
        release();
        if (rows_ <= 0 || cols_ <= 0) return;
        flags = 0x42FF0000 | type_;
        rows = rows_; cols = cols_;
        void *p = nullptr; size_t st = 0;
        miCheck(mi_malloc_pitch(&p, &st, (size_t)cols * elemSize(), rows));   // pitched like cudaMallocPitch (App. B Q13)
        data = datastart = (uchar *)p; step = st; dataend = data + st * (size_t)(rows - 1) + (size_t)cols * elemSize();
        refcount = new int(1);
    }
    void create(Size sz, int type_) { create(sz.height, sz.width, type_); }
    void release()
    {
        if (refcount && --*refcount == 0) { mi_free(datastart); delete refcount; }
        data = datastart = nullptr; dataend = nullptr; refcount = nullptr; rows = cols = 0; step = 0;
    }
    GpuMat operator()(Rect roi) const { return GpuMat(*this, roi); }
    // shim-only host transfers (real OpenCV: upload(InputArray)/download(OutputArray) with cv::Mat)
    void upload(const void *host, size_t hstep, Stream &s = Stream::Null())
    {
        miCheck(mi_memcpy_h2d(data, step, host, hstep, (size_t)cols * elemSize(), rows, s.hipStream()));
        if (!s.hipStream()) miCheck(mi_stream_synchronize(nullptr));
    }
    void download(void *host, size_t hstep, Stream &s = Stream::Null()) const
    {
        miCheck(mi_memcpy_d2h(host, hstep, data, step, (size_t)cols * elemSize(), rows, s.hipStream()));
        if (!s.hipStream()) miCheck(mi_stream_synchronize(nullptr));
    }
    void setTo(int byteValue, Stream &s = Stream::Null()) { miCheck(mi_memset(data, step, byteValue, (size_t)cols * elemSize(), rows, s.hipStream())); }
    int type() const { return flags & 0xFFF; }
    int depth() const { return CV_MAT_DEPTH(flags); }
    int channels() const { return CV_MAT_CN(flags); }
    size_t elemSize() const { static const int sz[8] = {1, 1, 2, 2, 4, 4, 8, 2}; return (size_t)sz[depth()] * channels(); }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == nullptr; }
    template <typename T> T *ptr(int y = 0) { return (T *)(data + step * (size_t)y); }
    template <typename T> const T *ptr(int y = 0) const { return (const T *)(data + step * (size_t)y); }

    // ---- layout of cv::cuda::GpuMat (main repo core/cuda.hpp), in this order
    int flags = 0;
    int rows = 0, cols = 0;
    size_t step = 0;
    uchar *data = nullptr;
    int *refcount = nullptr;
    uchar *datastart = nullptr;
    const uchar *dataend = nullptr;
    Allocator *allocator = nullptr;

private:
    void copyHeader(const GpuMat &m)
    {
        flags = m.flags; rows = m.rows; cols = m.cols; step = m.step; data = m.data; refcount = m.refcount;
        datastart = m.datastart; dataend = m.dataend; allocator = m.allocator;
    }
    void applyRoi(Rect r)
    {
        CV_Assert(0 <= r.x && 0 <= r.width && r.x + r.width <= cols && 0 <= r.y && 0 <= r.height && r.y + r.height <= rows);
        data += step * (size_t)r.y + (size_t)r.x * elemSize();
        rows = r.height; cols = r.width;
    }
};

// In real OpenCV these are proxy classes; a GpuMat binds to all of them, which is what the hot path's callers pass.
typedef const GpuMat &InputArray;
typedef GpuMat &OutputArray;
typedef GpuMat &InputOutputArray;

inline mi_mat miMat(const GpuMat &m)
{
    mi_mat r;
    r.data = m.data; r.step = m.step; r.rows = m.rows; r.cols = m.cols; r.type = m.type();
    return r;
}

}  // namespace cuda
}  // namespace cv

#endif  // !MIFLOW_WITH_OPENCV
#endif
