/* oracle/refshim/cudahost: of opencv2/video.hpp (video/tracking.hpp) the two Farneback flags of cudaoptflow/src/farneback.cpp; nothing is
 * used by cudaoptflow/src/tvl1flow.cpp.  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDAHOST_VIDEO_HPP
#define ORACLE_CUDAHOST_VIDEO_HPP
namespace cv { enum { OPTFLOW_USE_INITIAL_FLOW = 4, OPTFLOW_LK_GET_MIN_EIGENVALS = 8, OPTFLOW_FARNEBACK_GAUSSIAN = 256 }; }
#endif
