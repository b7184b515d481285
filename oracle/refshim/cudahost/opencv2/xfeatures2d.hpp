/* oracle/refshim/cudahost: shadows modules/xfeatures2d/include/opencv2/xfeatures2d.hpp (the CPU feature classes over features2d.hpp,
 * none of which xfeatures2d/src/surf.cuda.cpp uses); the CUDA class's header opencv2/xfeatures2d/cuda.hpp is NOT shadowed -- it is the
 * reference's own.  TEST INFRASTRUCTURE. */
