/* oracle/refshim/cudahost: the three cudaarithm functions cudaoptflow/src/tvl1flow.cpp calls (cudahost.cpp).  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDAHOST_CUDAARITHM_HPP
#define ORACLE_CUDAHOST_CUDAARITHM_HPP
#include "opencv2/core/cuda.hpp"
namespace cv { namespace cuda {
// cudaarithm.hpp:185: multiply(src1, src2, dst, scale, dtype, stream) -- here only matrix x Scalar, scale 1, dtype -1
void multiply(const GpuMat &src1, const Scalar &src2, GpuMat &dst, double scale, int dtype, Stream &stream);
// cudaarithm.hpp:  merge(const GpuMat* src, size_t n, OutputArray dst, Stream&)
void merge(const GpuMat *src, size_t n, OutputArray dst, Stream &stream);
// cudaarithm.hpp: calcSum(src, dst, mask, stream): dst = 1 x 1 CV_64FC(cn)
void calcSum(InputArray src, OutputArray dst, InputArray mask, Stream &stream);
// cudaarithm.hpp: split(InputArray src, std::vector<GpuMat>& dst, Stream&) -- farneback.cpp:185 (OPTFLOW_USE_INITIAL_FLOW)
void split(InputArray src, std::vector<GpuMat> &dst, Stream &stream);
// cudaarithm.hpp: integral(src CV_8UC1, sum CV_32SC1 of (rows + 1) x (cols + 1)), min(src1, scalar, dst) -- SURF_CUDA_Invoker (surf.cuda.cpp:160-166)
void integral(InputArray src, OutputArray sum, Stream &stream = Stream::Null());
void min(InputArray src1, InputArray src2, OutputArray dst, Stream &stream = Stream::Null());
}}
#endif
