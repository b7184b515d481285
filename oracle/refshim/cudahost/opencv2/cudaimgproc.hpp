/* oracle/refshim/cudahost: nothing of opencv2/cudaimgproc.hpp is used by cudaoptflow/src/tvl1flow.cpp.  TEST INFRASTRUCTURE. */
