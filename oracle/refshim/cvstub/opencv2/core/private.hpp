#include "opencv2/core.hpp"
#define CV_INSTRUMENT_REGION()
#define CV_OCL_RUN(condition, func)
