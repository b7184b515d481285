/* oracle/refshim/cudahost: nothing of opencv2/core/utility.hpp is used by cudastereo/src/stereobm.cpp.  TEST INFRASTRUCTURE. */
