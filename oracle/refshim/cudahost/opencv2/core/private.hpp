/* oracle/refshim/cudahost: nothing of opencv2/core/private.hpp is used by xfeatures2d/src/surf.cuda.cpp.  TEST INFRASTRUCTURE. */
