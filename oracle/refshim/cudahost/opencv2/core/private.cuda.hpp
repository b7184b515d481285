/* oracle/refshim/cudahost: core/private.cuda.hpp stand-in (StreamAccessor, BufferPool).  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDAHOST_PRIVATE_CUDA_HPP
#define ORACLE_CUDAHOST_PRIVATE_CUDA_HPP
#include "opencv2/core/cuda.hpp"
namespace cv { namespace cuda {
struct StreamAccessor { static cudaStream_t getStream(const Stream &) { return nullptr; } };
class BufferPool {
public:
    explicit BufferPool(Stream &) {}
    GpuMat getBuffer(Size s, int type) { return GpuMat(s, type); }
};
}}
static inline void throw_no_cuda() { throw std::runtime_error("no cuda"); }
#endif
