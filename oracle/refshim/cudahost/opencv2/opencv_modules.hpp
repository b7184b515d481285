/* oracle/refshim/cudahost: opencv2/opencv_modules.hpp of a build with cudaarithm (xfeatures2d/src/precomp.hpp and surf.cuda.cpp gate on
 * it; cudaoptflow / cudastereo use nothing of it).  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDAHOST_OPENCV_MODULES_HPP
#define ORACLE_CUDAHOST_OPENCV_MODULES_HPP
#define HAVE_OPENCV_CUDAARITHM
#endif
