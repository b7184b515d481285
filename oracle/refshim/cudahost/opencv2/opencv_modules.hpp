/* oracle/refshim/cudahost: nothing of opencv2/opencv_modules.hpp is used by cudaoptflow/src/tvl1flow.cpp.  TEST INFRASTRUCTURE. */
