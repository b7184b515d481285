/* oracle/refshim/cudashim: stand-in for the main-repo opencv2/core/cuda/functional.hpp -- the one functor surf.cu uses.
 * TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDASHIM_FUNCTIONAL_HPP
#define ORACLE_CUDASHIM_FUNCTIONAL_HPP
namespace cv { namespace cuda { namespace device {
template <typename T> struct plus { T operator()(const T &a, const T &b) const { return a + b; } };
}}}
#endif
