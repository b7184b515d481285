/* oracle/refshim/cudashim: stand-in for the main-repo opencv2/core/cuda/filters.hpp -- LinearFilter and AreaFilter over a Ptr2D with
 * a single-channel elem_type, as surf.cu:770-778 instantiates them with WinReader (elem_type = uchar).  Restated (the header is not
 * under /root/reference): both accumulate in float in the order written below and RETURN saturate_cast<elem_type>(out) -- for
 * WinReader the patch sample is rounded to an 8-bit value; LinearFilter takes floor(x), floor(y) (__float2int_rd); AreaFilter
 * normalises by 1 / (min(scale_x, src.width - fsx1) * min(scale_y, src.height - fsy1)) -- which is why WinReader carries
 * width / height.  TEST INFRASTRUCTURE. */
#ifndef ORACLE_CUDASHIM_FILTERS_HPP
#define ORACLE_CUDASHIM_FILTERS_HPP
#include "opencv2/core/cuda/saturate_cast.hpp"
namespace cv { namespace cuda { namespace device {
template <typename Ptr2D> struct LinearFilter {
    typedef typename Ptr2D::elem_type elem_type;
    typedef float index_type;
    explicit LinearFilter(const Ptr2D &src_, float = 0.f, float = 0.f) : src(src_) {}
    elem_type operator()(float y, float x) const
    {
        float out = 0.f;
        const int x1 = __float2int_rd(x), y1 = __float2int_rd(y), x2 = x1 + 1, y2 = y1 + 1;
        elem_type src_reg = src(y1, x1);
        out = out + src_reg * ((x2 - x) * (y2 - y));
        src_reg = src(y1, x2);
        out = out + src_reg * ((x - x1) * (y2 - y));
        src_reg = src(y2, x1);
        out = out + src_reg * ((x2 - x) * (y - y1));
        src_reg = src(y2, x2);
        out = out + src_reg * ((x - x1) * (y - y1));
        return saturate_cast<elem_type>(out);
    }
    Ptr2D src;
};
template <typename Ptr2D> struct AreaFilter {
    typedef typename Ptr2D::elem_type elem_type;
    typedef float index_type;
    explicit AreaFilter(const Ptr2D &src_, float scale_x_, float scale_y_) : src(src_), scale_x(scale_x_), scale_y(scale_y_) {}
    elem_type operator()(float y, float x) const
    {
        const float fsx1 = x * scale_x, fsx2 = fsx1 + scale_x;
        const int sx1 = __float2int_ru(fsx1), sx2 = __float2int_rd(fsx2);
        const float fsy1 = y * scale_y, fsy2 = fsy1 + scale_y;
        const int sy1 = __float2int_ru(fsy1), sy2 = __float2int_rd(fsy2);
        const float scale = 1.f / (fminf(scale_x, src.width - fsx1) * fminf(scale_y, src.height - fsy1));
        float out = 0.f;
        for (int dy = sy1; dy < sy2; ++dy) {
            for (int dx = sx1; dx < sx2; ++dx) out = out + src(dy, dx) * scale;
            if (sx1 > fsx1) out = out + src(dy, (sx1 - 1)) * ((sx1 - fsx1) * scale);
            if (sx2 < fsx2) out = out + src(dy, sx2) * ((fsx2 - sx2) * scale);
        }
        if (sy1 > fsy1)
            for (int dx = sx1; dx < sx2; ++dx) out = out + src((sy1 - 1), dx) * ((sy1 - fsy1) * scale);
        if (sy2 < fsy2)
            for (int dx = sx1; dx < sx2; ++dx) out = out + src(sy2, dx) * ((fsy2 - sy2) * scale);
        if ((sy1 > fsy1) && (sx1 > fsx1)) out = out + src((sy1 - 1), (sx1 - 1)) * ((sy1 - fsy1) * (sx1 - fsx1) * scale);
        if ((sy1 > fsy1) && (sx2 < fsx2)) out = out + src((sy1 - 1), sx2) * ((sy1 - fsy1) * (fsx2 - sx2) * scale);
        if ((sy2 < fsy2) && (sx2 < fsx2)) out = out + src(sy2, sx2) * ((fsy2 - sy2) * (fsx2 - sx2) * scale);
        if ((sy2 < fsy2) && (sx1 > fsx1)) out = out + src(sy2, (sx1 - 1)) * ((fsy2 - sy2) * (sx1 - fsx1) * scale);
        return saturate_cast<elem_type>(out);
    }
    Ptr2D src;
    float scale_x, scale_y;
};
}}}
#endif
