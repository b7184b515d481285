/* oracle/refshim/cudashim: opencv2/opencv_modules.hpp of a build with cudaarithm (xfeatures2d/src/cuda/surf.cu:43-45 compiles its body
 * only then; OPENCV_ENABLE_NONFREE comes from the command line of oracle/Makefile.ref).  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDASHIM_OPENCV_MODULES_HPP
#define ORACLE_CUDASHIM_OPENCV_MODULES_HPP
#define HAVE_OPENCV_CUDAARITHM
#endif
