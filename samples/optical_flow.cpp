// The calling pattern of the reference's sample (modules/cudaoptflow/samples/optical_flow.cpp: upload two frames, run each dense
// optical-flow class, download and report), written against the drop-in headers of this repository.  The frames are synthetic (no
// image codecs here): a smooth texture and the same texture shifted by (2, 1) pixels, so every algorithm should report a mean flow
// near (2, 1).
//   g++ -std=c++17 -Iinclude samples/optical_flow.cpp -Lopencv_contrib_amd -lmiflow -Wl,-rpath,$PWD/opencv_contrib_amd -o optical_flow
#include <cmath>
#include <cstdio>
#include <vector>
#include "opencv2/cudaoptflow.hpp"

using namespace cv;

static void make_frames(int rows, int cols, std::vector<uchar> &f0, std::vector<uchar> &f1)
{
    f0.resize((size_t)rows * cols);
    f1.resize((size_t)rows * cols);
    auto tex = [](double x, double y) {
        return 127.5 + 60.0 * std::sin(0.11 * x + 0.5 * std::sin(0.07 * y)) + 50.0 * std::cos(0.09 * y + 0.4 * std::sin(0.05 * x));
    };
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            f0[(size_t)y * cols + x] = (uchar)std::lround(tex(x, y));
            f1[(size_t)y * cols + x] = (uchar)std::lround(tex(x - 2.0, y - 1.0));      // content moves by (+2, +1)
        }
}

static void report(const char *name, const cuda::GpuMat &flow)
{
    std::vector<float> h((size_t)flow.rows * flow.cols * 2);
    flow.download(h.data(), (size_t)flow.cols * 8);
    double su = 0, sv = 0;
    long n = 0;
    for (int y = 20; y < flow.rows - 20; ++y)
        for (int x = 20; x < flow.cols - 20; ++x, ++n) { su += h[((size_t)y * flow.cols + x) * 2]; sv += h[((size_t)y * flow.cols + x) * 2 + 1]; }
    std::printf("%-12s mean flow (%.3f, %.3f)\n", name, su / n, sv / n);
}

int main()
{
    try {
        const int rows = 240, cols = 320;
        std::vector<uchar> f0, f1;
        make_frames(rows, cols, f0, f1);
        cuda::GpuMat d0(rows, cols, CV_8UC1), d1(rows, cols, CV_8UC1), flow;
        d0.upload(f0.data(), cols);
        d1.upload(f1.data(), cols);

        Ptr<cuda::OpticalFlowDual_TVL1> tvl1 = cuda::OpticalFlowDual_TVL1::create();
        tvl1->calc(d0, d1, flow);
        report("TV-L1", flow);

        Ptr<cuda::FarnebackOpticalFlow> fb = cuda::FarnebackOpticalFlow::create();
        fb->calc(d0, d1, flow);
        report("Farneback", flow);

        Ptr<cuda::DensePyrLKOpticalFlow> lk = cuda::DensePyrLKOpticalFlow::create(Size(13, 13), 3, 30);
        lk->calc(d0, d1, flow);
        report("DensePyrLK", flow);

        // sparse tracking of a grid of points
        std::vector<float> pts;
        for (int y = 40; y < rows - 40; y += 20) for (int x = 40; x < cols - 40; x += 20) { pts.push_back((float)x); pts.push_back((float)y); }
        const int n = (int)pts.size() / 2;
        cuda::GpuMat prevPts(1, n, CV_32FC2), nextPts, status, err;
        prevPts.upload(pts.data(), (size_t)n * 8);
        Ptr<cuda::SparsePyrLKOpticalFlow> slk = cuda::SparsePyrLKOpticalFlow::create();
        slk->calc(d0, d1, prevPts, nextPts, status, err);
        std::vector<float> np((size_t)n * 2);
        std::vector<uchar> st(n);
        nextPts.download(np.data(), (size_t)n * 8);
        status.download(st.data(), n);
        double su = 0, sv = 0;
        int ok = 0;
        for (int i = 0; i < n; ++i) if (st[i]) { su += np[2 * i] - pts[2 * i]; sv += np[2 * i + 1] - pts[2 * i + 1]; ++ok; }
        std::printf("%-12s mean flow (%.3f, %.3f) over %d / %d tracked points\n", "SparsePyrLK", su / (ok ? ok : 1), sv / (ok ? ok : 1), ok, n);
        return 0;
    } catch (const cv::Exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 3;
    }
}
