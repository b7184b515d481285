/* oracle/refshim/cudahost: cuda::resize (cudawarping.hpp:104-105; host glue cudawarping/src/resize.cpp:55-103 restated in cudahost.cpp,
 * kernel = the reference's resize_linear of libref_cu.so).  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDAHOST_CUDAWARPING_HPP
#define ORACLE_CUDAHOST_CUDAWARPING_HPP
#include "opencv2/core/cuda.hpp"
namespace cv { namespace cuda {
void resize(InputArray src, OutputArray dst, Size dsize, double fx, double fy, int interpolation, Stream &stream);
// cudawarping.hpp: pyrDown(src, dst, stream); host glue cudawarping/src/pyramids.cpp (dst = ((rows + 1) / 2, (cols + 1) / 2)), kernel = the
// reference's pyrDown of libref_cu.so
void pyrDown(InputArray src, OutputArray dst, Stream &stream);
}}
#endif
