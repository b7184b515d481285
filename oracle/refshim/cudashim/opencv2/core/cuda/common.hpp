/* oracle/refshim/cudashim: stand-in for the main-repo opencv2/core/cuda/common.hpp + core/cuda_types.hpp (absent from
 * /root/reference): the PtrStep / PtrStepSz device views and divUp.  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDASHIM_COMMON_HPP
#define ORACLE_CUDASHIM_COMMON_HPP
#include "cudashim.h"
namespace cv { namespace cuda {
template <typename T> struct DevPtr { typedef T elem_type; typedef int index_type; enum { elem_size = sizeof(elem_type) }; T *data; DevPtr() : data(0) {} DevPtr(T *d) : data(d) {} size_t elemSize() const { return elem_size; } operator T *() { return data; } operator const T *() const { return data; } };
template <typename T> struct PtrStep : public DevPtr<T> {
    PtrStep() : step(0) {}
    PtrStep(T *data_, size_t step_) : DevPtr<T>(data_), step(step_) {}
    size_t step;
    T *ptr(int y = 0) { return (T *)((char *)DevPtr<T>::data + y * step); }
    const T *ptr(int y = 0) const { return (const T *)((const char *)DevPtr<T>::data + y * step); }
    T &operator()(int y, int x) { return ptr(y)[x]; }
    const T &operator()(int y, int x) const { return ptr(y)[x]; }
};
template <typename T> struct PtrStepSz : public PtrStep<T> {
    PtrStepSz() : cols(0), rows(0) {}
    PtrStepSz(int rows_, int cols_, T *data_, size_t step_) : PtrStep<T>(data_, step_), cols(cols_), rows(rows_) {}
    template <typename U> explicit PtrStepSz(const PtrStepSz<U> &d) : PtrStep<T>((T *)d.data, d.step), cols(d.cols), rows(d.rows) {}
    int cols, rows;
};
typedef PtrStepSz<unsigned char> PtrStepSzb;
typedef PtrStepSz<float> PtrStepSzf;
typedef PtrStepSz<int> PtrStepSzi;
typedef PtrStep<unsigned char> PtrStepb;
typedef PtrStep<float> PtrStepf;
typedef PtrStep<int> PtrStepi;
namespace device {
static inline int divUp(int total, int grain) { return (total + grain - 1) / grain; }
}
}}
#endif
