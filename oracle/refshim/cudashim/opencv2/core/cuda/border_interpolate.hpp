/* oracle/refshim/cudashim: stand-in for the main-repo opencv2/core/cuda/border_interpolate.hpp (absent from /root/reference):
 * the two index maps farneback.cu instantiates, BrdReplicate (clamp) and BrdReflect101 (mirror without repeating the edge:
 * gfedcb|abcdefgh|gfedcba), as that header defines them.  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDASHIM_BORDER_HPP
#define ORACLE_CUDASHIM_BORDER_HPP
#include "opencv2/core/cuda/common.hpp"
namespace cv { namespace cuda { namespace device {
template <typename D> struct BrdReplicate {
    typedef D result_type;
    BrdReplicate(int height, int width) : last_row(height - 1), last_col(width - 1) {}
    int idx_row_low(int y) const { return ::max(y, 0); }
    int idx_row_high(int y) const { return ::min(y, last_row); }
    int idx_row(int y) const { return idx_row_low(idx_row_high(y)); }
    int idx_col_low(int x) const { return ::max(x, 0); }
    int idx_col_high(int x) const { return ::min(x, last_col); }
    int idx_col(int x) const { return idx_col_low(idx_col_high(x)); }
    int last_row, last_col;
};
template <typename D> struct BrdReflect101 {
    typedef D result_type;
    BrdReflect101(int height, int width) : last_row(height - 1), last_col(width - 1) {}
    int idx_row_low(int y) const { return ::abs(y) % (last_row + 1); }
    int idx_row_high(int y) const { return ::abs(last_row - ::abs(last_row - y)) % (last_row + 1); }
    int idx_row(int y) const { return idx_row_low(idx_row_high(y)); }
    int idx_col_low(int x) const { return ::abs(x) % (last_col + 1); }
    int idx_col_high(int x) const { return ::abs(last_col - ::abs(last_col - x)) % (last_col + 1); }
    int idx_col(int x) const { return idx_col_low(idx_col_high(x)); }
    int last_row, last_col;
};
}}}
#endif
