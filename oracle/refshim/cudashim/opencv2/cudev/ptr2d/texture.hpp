/* oracle/refshim/cudashim: stand-in for modules/cudev/include/opencv2/cudev/ptr2d/texture.hpp (which needs the CUDA runtime's
 * texture objects).  A texture read is DEFINED by the CUDA programming guide, not by reference source: point sampling reads
 * texel floor(x); linear filtering of unnormalised coordinates samples at x - 0.5 with the fraction kept in 1.8 fixed point;
 * out-of-range unnormalised coordinates clamp (wrap needs normalised coordinates); cudaReadModeNormalizedFloat maps u8 to
 * [0, 1] by / 255.  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDASHIM_TEXTURE_HPP
#define ORACLE_CUDASHIM_TEXTURE_HPP
#include "opencv2/core/cuda/common.hpp"
namespace cv { namespace cudev {
using cv::cuda::PtrStepSz;
template <class T, class R = T> struct TexturePtr {
    const T *data = nullptr; size_t step = 0; int rows = 0, cols = 0; int linear = 0, normfloat = 0;
    R fetch(int x, int y) const
    {
        x = x < 0 ? 0 : (x > cols - 1 ? cols - 1 : x);
        y = y < 0 ? 0 : (y > rows - 1 ? rows - 1 : y);
        const T v = *(const T *)((const char *)data + (size_t)y * step + (size_t)x * sizeof(T));
        return normfloat ? (R)((float)v / 255.0f) : (R)v;
    }
    R operator()(float y, float x) const
    {
        if (!linear) return fetch((int)floorf(x), (int)floorf(y));
        const float xb = x - 0.5f, yb = y - 0.5f;
        const int i = (int)floorf(xb), j = (int)floorf(yb);
        const float a = floorf((xb - (float)i) * 256.0f + 0.5f) / 256.0f, b = floorf((yb - (float)j) * 256.0f + 0.5f) / 256.0f;   // 1.8 fixed point
        return (R)((1 - a) * (1 - b) * fetch(i, j) + a * (1 - b) * fetch(i + 1, j) + (1 - a) * b * fetch(i, j + 1) + a * b * fetch(i + 1, j + 1));
    }
};
template <class T, class R = T> class Texture {
public:
    Texture(PtrStepSz<T> src, bool = false, cudaTextureFilterMode f = cudaFilterModePoint, cudaTextureAddressMode = cudaAddressModeClamp,
            cudaTextureReadMode r = cudaReadModeElementType)
    {
        p.data = src.data; p.step = src.step; p.rows = src.rows; p.cols = src.cols; p.linear = f == cudaFilterModeLinear; p.normfloat = r == cudaReadModeNormalizedFloat;
    }
    operator TexturePtr<T, R>() const { return p; }
private:
    TexturePtr<T, R> p;
};
}}
#endif
