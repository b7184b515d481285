/*
 * oracle/refshim/cvstub/cvstub.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See opencv2/core.hpp in this directory.
 * The main-repo functions tvl1flow.cpp calls, forwarded to oracle/imgproc_ref.c (dense planes), and the C entry point that
 * drives the reference class cv::optflow::DualTVL1OpticalFlow (compiled from /root/reference by oracle/Makefile.ref).
 */
#include "opencv2/optflow.hpp"
#include "../../imgproc_ref.h"
#include <cmath>

namespace cv {

static std::vector<float> dense(const Mat &m)
{
    CV_Assert(m.type() == CV_32FC1);
    std::vector<float> v((size_t)m.rows * m.cols);
    for (int y = 0; y < m.rows; ++y) memcpy(&v[(size_t)y * m.cols], m.ptr<float>(y), sizeof(float) * m.cols);
    return v;
}
static void undense(const std::vector<float> &v, Mat &m)
{
    for (int y = 0; y < m.rows; ++y) memcpy(m.ptr<float>(y), &v[(size_t)y * m.cols], sizeof(float) * m.cols);
}

Mat &Mat::setTo(const Scalar &s)
{
    CV_Assert(type_ == CV_32FC1);
    for (int y = 0; y < rows; ++y) { float *r = ptr<float>(y); for (int x = 0; x < cols; ++x) r[x] = (float)s[0]; }
    return *this;
}

// Mat::convertTo(dst, CV_32F, alpha): cvtScale 8u->32f / 32f->32f computes src * (float)alpha in float
void Mat::convertTo(Mat &dst, int rtype, double alpha) const
{
    CV_Assert((rtype & 7) == CV_32F && channels() == 1);
    Mat out;
    out.create(rows, cols, CV_32FC1);
    const float a = (float)alpha;
    for (int y = 0; y < rows; ++y) {
        float *d = out.ptr<float>(y);
        if (depth() == CV_8U) { const uchar *s = ptr<uchar>(y); for (int x = 0; x < cols; ++x) d[x] = (float)s[x] * a; }
        else { const float *s = ptr<float>(y); for (int x = 0; x < cols; ++x) d[x] = s[x] * a; }
    }
    static_cast<Mat &>(dst) = out;
}

// cv::resize: dsize empty -> dsize = saturate_cast<int>(ssize * f) (round half to even), inv_scale = f; else inv_scale = dsize / ssize
void resize(InputArray src, OutputArray dst, Size dsize, double fx, double fy, int interpolation)
{
    CV_Assert(interpolation == INTER_LINEAR && src.type() == CV_32FC1);
    double isx, isy;
    if (dsize.area() == 0) {
        dsize = Size(orc_scaled_dim(src.cols, fx), orc_scaled_dim(src.rows, fy));
        isx = fx; isy = fy;
    } else {
        isx = (double)dsize.width / src.cols; isy = (double)dsize.height / src.rows;
    }
    CV_Assert(dsize.width > 0 && dsize.height > 0);
    const std::vector<float> s = dense(src);
    std::vector<float> d((size_t)dsize.width * dsize.height);
    orc_resize_linear_cv(s.data(), src.cols, src.rows, d.data(), dsize.width, dsize.height, 1.0 / isx, 1.0 / isy);
    const int type = CV_32FC1;
    if (!(dst.rows == dsize.height && dst.cols == dsize.width && dst.type() == type && dst.data)) dst.create(dsize.height, dsize.width, type);
    undense(d, dst);
}

void remap(InputArray src, OutputArray dst, InputArray map1, InputArray map2, int interpolation)
{
    CV_Assert(interpolation == INTER_CUBIC && src.type() == CV_32FC1 && map1.type() == CV_32FC1 && map2.type() == CV_32FC1);
    const std::vector<float> s = dense(src), mx = dense(map1), my = dense(map2);
    std::vector<float> d((size_t)map1.rows * map1.cols);
    orc_remap_cubic_cv(s.data(), src.cols, src.rows, mx.data(), my.data(), d.data(), map1.cols, map1.rows);
    if (!(dst.rows == map1.rows && dst.cols == map1.cols && dst.data)) dst.create(map1.rows, map1.cols, CV_32FC1);
    undense(d, dst);
}

void medianBlur(InputArray src, OutputArray dst, int ksize)
{
    const std::vector<float> s = dense(src);
    std::vector<float> d(s.size());
    orc_median_blur(s.data(), d.data(), src.cols, src.rows, ksize);
    if (!(dst.rows == src.rows && dst.cols == src.cols && dst.data)) dst.create(src.rows, src.cols, CV_32FC1);
    undense(d, dst);
}

// cv::multiply(f32 matrix, Scalar): the scalar is converted to the working type float, products in float
void multiply(InputArray src, const Scalar &s, OutputArray dst)
{
    CV_Assert(src.type() == CV_32FC1 && dst.rows == src.rows && dst.cols == src.cols);
    const float k = (float)s[0];
    for (int y = 0; y < src.rows; ++y) { const float *a = src.ptr<float>(y); float *d = dst.ptr<float>(y); for (int x = 0; x < src.cols; ++x) d[x] = a[x] * k; }
}

void split(const Mat &src, Mat_<float> *mv)
{
    CV_Assert(src.type() == CV_32FC2);
    for (int c = 0; c < 2; ++c) if (mv[c].rows != src.rows || mv[c].cols != src.cols || !mv[c].data) mv[c].create(src.rows, src.cols);
    for (int y = 0; y < src.rows; ++y) {
        const float *s = src.ptr<float>(y);
        float *a = mv[0][y], *b = mv[1][y];
        for (int x = 0; x < src.cols; ++x) { a[x] = s[2 * x]; b[x] = s[2 * x + 1]; }
    }
}

void merge(const Mat *mv, size_t count, OutputArray dst)
{
    CV_Assert(count == 2 && mv[0].type() == CV_32FC1);
    if (!(dst.rows == mv[0].rows && dst.cols == mv[0].cols && dst.type() == CV_32FC2 && dst.data)) dst.create(mv[0].rows, mv[0].cols, CV_32FC2);
    for (int y = 0; y < dst.rows; ++y) {
        float *d = dst.ptr<float>(y);
        const float *a = mv[0].ptr<float>(y), *b = mv[1].ptr<float>(y);
        for (int x = 0; x < dst.cols; ++x) { d[2 * x] = a[x]; d[2 * x + 1] = b[x]; }
    }
}

}  // namespace cv

// C entry: the reference class, end to end.  type 0 = CV_8UC1, 1 = CV_32FC1; images dense; flow dense interleaved (read when
// use_initial_flow).  Returns 0, or -1 when the class throws (CV_Assert).  *nscales_out = the class's nscales after calc (it shrinks it).
extern "C" int ref_cpu_tvl1_calc(double tau, double lambda, double theta, int nscales, int warps, double epsilon, int inner_iterations,
                                 int outer_iterations, double scale_step, double gamma, int median_filtering, int use_initial_flow,
                                 const void *I0, const void *I1, int type, int w, int h, float *flow, int *nscales_out)
{
    try {
        cv::Mat a(h, w, type == 0 ? CV_8UC1 : CV_32FC1), b(h, w, type == 0 ? CV_8UC1 : CV_32FC1);
        memcpy(a.data, I0, (size_t)h * a.step);
        memcpy(b.data, I1, (size_t)h * b.step);
        cv::Mat f;
        if (use_initial_flow) { f.create(h, w, CV_32FC2); memcpy(f.data, flow, (size_t)h * f.step); }
        cv::Ptr<cv::optflow::DualTVL1OpticalFlow> alg = cv::optflow::DualTVL1OpticalFlow::create(
            tau, lambda, theta, nscales, warps, epsilon, inner_iterations, outer_iterations, scale_step, gamma, median_filtering,
            use_initial_flow != 0);
        alg->calc(a, b, f);
        if (nscales_out) *nscales_out = alg->getScalesNumber();
        if (f.rows != h || f.cols != w || f.type() != CV_32FC2) return -2;
        memcpy(flow, f.data, (size_t)h * f.step);
        return 0;
    } catch (const std::exception &) {
        return -1;
    }
}
