// oracle/refshim/cvsurf/cvsurf.cpp -- the non-inline part of the stub (see opencv2/core.hpp) and the C entry point through which the
// tests drive the reference's CPU SURF class (modules/xfeatures2d/src/surf.cpp, compiled verbatim into the same library).
// Test infrastructure only.
#include "opencv2/xfeatures2d.hpp"
#include <cstdio>

namespace cv {
Mat getGaussianKernel(int n, double sigma, int ktype)
{
    CV_Assert(ktype == CV_32F && n > 0 && n <= 64 && sigma > 0);
    Mat k(n, 1, CV_32F);
    orc_cv_gauss_kernel(n, sigma, k.ptr<float>(0));
    return k;
}
void integral(InputArray src_, Mat &sum, int sdepth)
{
    const Mat src = src_.getMat();
    CV_Assert(sdepth == CV_32S && src.type() == CV_8U);
    sum.create(src.rows + 1, src.cols + 1, CV_32S);
    orc_cv_integral_u8(src.data, (long long)src.step, src.rows, src.cols, sum.ptr<int>(0));
}
void resize(const Mat &src, Mat &dst, Size dsize, double, double, int interpolation)
{
    CV_Assert(interpolation == INTER_AREA && src.type() == CV_8U && src.rows == src.cols && dsize.width == dsize.height && src.isContinuous());
    CV_Assert(dst.rows == dsize.height && dst.cols == dsize.width && dst.isContinuous());   // the class hands a preallocated patch
    const int rc = orc_cv_resize_area_u8(src.data, src.rows, dst.data, dsize.width);
    if (rc) throw std::runtime_error("resize(INTER_AREA): enlargement is not restated (keypoint size below 7.5)");
}
void phase(const Mat &x, const Mat &y, const Mat &angle, bool angleInDegrees)
{
    CV_Assert(angleInDegrees && x.type() == CV_32F && x.rows == 1 && y.cols == x.cols && angle.cols == x.cols);
    float *a = const_cast<float *>(angle.ptr<float>(0));
    for (int i = 0; i < x.cols; ++i) a[i] = orc_cv_fast_atan2(y.ptr<float>(0)[i], x.ptr<float>(0)[i]);
}
void cvtColor(const Mat &, Mat &, int) { throw std::runtime_error("cvtColor: colour input is outside this pin (8-bit grey only)"); }
void min(const Mat &src, double v, Mat &dst)
{
    CV_Assert(src.type() == CV_8U);
    dst.create(src.rows, src.cols, CV_8U);
    for (int y = 0; y < src.rows; ++y)
        for (int x = 0; x < src.cols; ++x) dst.at<uchar>(y, x) = (uchar)std::min<double>(src.at<uchar>(y, x), v);
}
}  // namespace cv

// keypoints: n x 7 floats {x, y, size, angle, response, octave, class_id}.  use_provided: kp holds n_in keypoints, descriptors only.
// Returns the number of keypoints (<= cap), -1 on an exception (message on stderr), -2 if cap is too small.
extern "C" int ref_surfcpu_detect_and_compute(const unsigned char *img, const unsigned char *mask, int rows, int cols, double hessian_threshold,
                                              int n_octaves, int n_octave_layers, int extended, int upright, int use_provided, int n_in,
                                              float *kp, int cap, float *desc)
{
    try {
        cv::Mat I(rows, cols, CV_8U, const_cast<unsigned char *>(img)), M;
        if (mask) M = cv::Mat(rows, cols, CV_8U, const_cast<unsigned char *>(mask));
        cv::Ptr<cv::xfeatures2d::SURF> surf = cv::xfeatures2d::SURF::create(hessian_threshold, n_octaves, n_octave_layers, extended != 0, upright != 0);
        std::vector<cv::KeyPoint> kps;
        if (use_provided)
            for (int i = 0; i < n_in; ++i)
                kps.push_back(cv::KeyPoint(kp[7 * i], kp[7 * i + 1], kp[7 * i + 2], kp[7 * i + 3], kp[7 * i + 4], (int)kp[7 * i + 5], (int)kp[7 * i + 6]));
        cv::Mat D;
        if (desc) surf->detectAndCompute(cv::_InputArray(I), mask ? cv::_InputArray(M) : cv::_InputArray(), kps, cv::_OutputArray(D), use_provided != 0);
        else surf->detectAndCompute(cv::_InputArray(I), mask ? cv::_InputArray(M) : cv::_InputArray(), kps, cv::noArray(), false);
        const int n = (int)kps.size();
        if (n > cap) return -2;
        for (int i = 0; i < n; ++i) {
            kp[7 * i] = kps[i].pt.x; kp[7 * i + 1] = kps[i].pt.y; kp[7 * i + 2] = kps[i].size; kp[7 * i + 3] = kps[i].angle;
            kp[7 * i + 4] = kps[i].response; kp[7 * i + 5] = (float)kps[i].octave; kp[7 * i + 6] = (float)kps[i].class_id;
        }
        if (desc && n > 0) {
            const int dc = extended ? 128 : 64;
            CV_Assert(D.rows == n && D.cols == dc && D.type() == CV_32F);
            for (int i = 0; i < n; ++i) memcpy(desc + (size_t)i * dc, D.ptr<float>(i), sizeof(float) * dc);
        }
        return n;
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_surfcpu_detect_and_compute: %s\n", e.what());
        return -1;
    }
}
