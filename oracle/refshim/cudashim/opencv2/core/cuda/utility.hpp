/* oracle/refshim/cudashim: stand-in for the main-repo opencv2/core/cuda/utility.hpp -- solve3x3 (Cramer's rule as that header writes
 * it: the determinant and the three cofactor sums in T, the reciprocal of the determinant in DOUBLE, each solution component
 * rounded back to T by saturate_cast<T>).  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDASHIM_UTILITY_HPP
#define ORACLE_CUDASHIM_UTILITY_HPP
namespace cv { namespace cuda { namespace device {
template <typename T> static inline bool solve3x3(const T A[3][3], const T b[3], T x[3])
{
    const T det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                  A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    if (det != 0) {
        const double invdet = 1.0 / det;
        x[0] = (T)(invdet * (b[0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (b[1] * A[2][2] - A[1][2] * b[2]) +
                             A[0][2] * (b[1] * A[2][1] - A[1][1] * b[2])));
        x[1] = (T)(invdet * (A[0][0] * (b[1] * A[2][2] - A[1][2] * b[2]) - b[0] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                             A[0][2] * (A[1][0] * b[2] - b[1] * A[2][0])));
        x[2] = (T)(invdet * (A[0][0] * (A[1][1] * b[2] - b[1] * A[2][1]) - A[0][1] * (A[1][0] * b[2] - b[1] * A[2][0]) +
                             b[0] * (A[1][0] * A[2][1] - A[1][1] * A[2][0])));
        return true;
    }
    return false;
}
}}}
#endif
