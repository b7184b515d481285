#include "opencv2/core.hpp"
