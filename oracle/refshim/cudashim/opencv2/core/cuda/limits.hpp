/* oracle/refshim/cudashim: stand-in for the main-repo opencv2/core/cuda/limits.hpp (device numeric_limits).  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDASHIM_LIMITS_HPP
#define ORACLE_CUDASHIM_LIMITS_HPP
#include <cfloat>
#include <climits>
namespace cv { namespace cuda { namespace device {
template <class T> struct numeric_limits;
template <> struct numeric_limits<float> { static float min() { return FLT_MIN; } static float max() { return FLT_MAX; } static float epsilon() { return FLT_EPSILON; } };
template <> struct numeric_limits<int> { static int min() { return INT_MIN; } static int max() { return INT_MAX; } };
template <> struct numeric_limits<short> { static short min() { return SHRT_MIN; } static short max() { return SHRT_MAX; } };
}}}
#endif
