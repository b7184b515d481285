/* oracle/refshim/cudashim: stand-in for the main-repo opencv2/core/cuda/saturate_cast.hpp -- the conversions surf.cu reaches through
 * filters.hpp: float -> uchar is PTX cvt.rni.sat.u8.f32 (round to nearest even, saturate), float -> float the identity.
 * TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDASHIM_SATURATE_CAST_HPP
#define ORACLE_CUDASHIM_SATURATE_CAST_HPP
#include "opencv2/core/cuda/common.hpp"
namespace cv { namespace cuda { namespace device {
template <typename T> static inline T saturate_cast(float v);
template <> inline float saturate_cast<float>(float v) { return v; }
template <> inline uchar saturate_cast<uchar>(float v)
{
    const float r = nearbyintf(v);
    return (uchar)(r < 0.f ? 0 : r > 255.f ? 255 : (int)r);   // NaN -> 0, like cvt.sat
}
}}}
#endif
