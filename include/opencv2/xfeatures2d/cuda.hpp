// Drop-in for modules/xfeatures2d/include/opencv2/xfeatures2d/cuda.hpp (cv::cuda::SURF_CUDA) over the miflow C-ABI.
// cv::KeyPoint / std::vector overloads need main-repo types (core/types.hpp); in the stand-alone shim the
// host-side keypoints are the plain struct below with the fields the reference fills (surf.cuda.cpp:319-356).
#ifndef MIFLOW_OPENCV_XFEATURES2D_CUDA_HPP
#define MIFLOW_OPENCV_XFEATURES2D_CUDA_HPP

#include <vector>
#include "opencv2/core/cuda.hpp"

namespace cv {
#ifndef MIFLOW_WITH_OPENCV
struct Point2f { float x = 0, y = 0; };
struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1; };
#ifndef MIFLOW_HAVE_NORM_TYPES
#define MIFLOW_HAVE_NORM_TYPES
enum NormTypes { NORM_INF = 1, NORM_L1 = 2, NORM_L2 = 4, NORM_HAMMING = 6 };   // opencv2/core/base.hpp
#endif
#endif

namespace cuda {

class SURF_CUDA {
public:
    enum KeypointLayout { X_ROW = 0, Y_ROW, LAPLACIAN_ROW, OCTAVE_ROW, SIZE_ROW, ANGLE_ROW, HESSIAN_ROW, ROWS_COUNT };

    //! the default constructor (surf.cuda.cpp:257-265: extended = true)
    SURF_CUDA() : hessianThreshold(100), nOctaves(4), nOctaveLayers(2), extended(true), upright(false), keypointsRatio(0.01f) { init(); }
    explicit SURF_CUDA(double _hessianThreshold, int _nOctaves = 4, int _nOctaveLayers = 2, bool _extended = false,
                       float _keypointsRatio = 0.01f, bool _upright = false)
        : hessianThreshold(_hessianThreshold), nOctaves(_nOctaves), nOctaveLayers(_nOctaveLayers), extended(_extended), upright(_upright),
          keypointsRatio(_keypointsRatio) { init(); }
    ~SURF_CUDA() { mi_surf_destroy(h_); }
    SURF_CUDA(const SURF_CUDA &) = delete;
    SURF_CUDA &operator=(const SURF_CUDA &) = delete;

    static Ptr<SURF_CUDA> create(double _hessianThreshold, int _nOctaves = 4, int _nOctaveLayers = 2, bool _extended = false,
                                 float _keypointsRatio = 0.01f, bool _upright = false)
    {
        return makePtr<SURF_CUDA>(_hessianThreshold, _nOctaves, _nOctaveLayers, _extended, _keypointsRatio, _upright);
    }

    int descriptorSize() const { return extended ? 128 : 64; }
    int defaultNorm() const { return NORM_L2; }

    void uploadKeypoints(const std::vector<KeyPoint> &keypoints, GpuMat &keypointsGPU)
    {
        // surf.cuda.cpp:287-317
        if (keypoints.empty()) { keypointsGPU.release(); return; }
        const int n = (int)keypoints.size();
        std::vector<float> host((size_t)ROWS_COUNT * n);
        for (int i = 0; i < n; ++i) {
            const KeyPoint &kp = keypoints[i];
            host[(size_t)X_ROW * n + i] = kp.pt.x; host[(size_t)Y_ROW * n + i] = kp.pt.y;
            reinterpret_cast<int *>(host.data())[(size_t)LAPLACIAN_ROW * n + i] = kp.class_id;
            reinterpret_cast<int *>(host.data())[(size_t)OCTAVE_ROW * n + i] = kp.octave;
            host[(size_t)SIZE_ROW * n + i] = kp.size; host[(size_t)ANGLE_ROW * n + i] = kp.angle; host[(size_t)HESSIAN_ROW * n + i] = kp.response;
        }
        keypointsGPU.create(ROWS_COUNT, n, CV_32FC1);
        keypointsGPU.upload(host.data(), (size_t)n * 4);
    }
    void downloadKeypoints(const GpuMat &keypointsGPU, std::vector<KeyPoint> &keypoints)
    {
        // surf.cuda.cpp:319-356
        const int n = keypointsGPU.cols;
        keypoints.resize(n);
        if (n == 0) return;
        CV_Assert(keypointsGPU.type() == CV_32FC1 && keypointsGPU.rows == ROWS_COUNT);
        std::vector<float> host((size_t)ROWS_COUNT * n);
        keypointsGPU.download(host.data(), (size_t)n * 4);
        for (int i = 0; i < n; ++i) {
            KeyPoint &kp = keypoints[i];
            kp.pt.x = host[(size_t)X_ROW * n + i]; kp.pt.y = host[(size_t)Y_ROW * n + i];
            kp.class_id = reinterpret_cast<int *>(host.data())[(size_t)LAPLACIAN_ROW * n + i];
            kp.octave = reinterpret_cast<int *>(host.data())[(size_t)OCTAVE_ROW * n + i];
            kp.size = host[(size_t)SIZE_ROW * n + i]; kp.angle = host[(size_t)ANGLE_ROW * n + i]; kp.response = host[(size_t)HESSIAN_ROW * n + i];
        }
    }
    void downloadDescriptors(const GpuMat &descriptorsGPU, std::vector<float> &descriptors)
    {
        // surf.cuda.cpp:358-367
        if (descriptorsGPU.empty()) { descriptors.clear(); return; }
        CV_Assert(descriptorsGPU.type() == CV_32F);
        descriptors.resize((size_t)descriptorsGPU.rows * descriptorsGPU.cols);
        descriptorsGPU.download(descriptors.data(), (size_t)descriptorsGPU.cols * 4);
    }

    // ensureSizeIsEnough on the ALLOCATION, not on the header: a call hands `m` back with cols / rows cut to the feature count,
    // so a test of the header would free and re-create the buffers on every frame (up to 33 MB of hipFree + hipMalloc, each a device
    // synchronisation).  The buffer a previous call created -- its pitch and extent are still in step / datastart / dataend -- is reused
    // whenever it holds rows x cols elements of the type; the header is restored to that size.
    static bool holds(const GpuMat &m, int rows, int cols, int type)
    {
        static const int dsz[8] = {1, 1, 2, 2, 4, 4, 8, 2};
        const size_t es = (size_t)dsz[type & 7] * (size_t)CV_MAT_CN(type);
        return m.data && m.data == m.datastart && m.type() == (type & 0xFFF) && m.step >= (size_t)cols * es &&
               (size_t)(m.dataend - m.datastart) >= m.step * (size_t)(rows - 1) + (size_t)cols * es;
    }
    static void ensureCapacity(GpuMat &m, int rows, int cols, int type)
    {
        if (holds(m, rows, cols, type)) { m.rows = rows; m.cols = cols; return; }
        m.release();
        m.create(rows, cols, type);
    }

    //! finds the keypoints (surf.cuda.cpp:369-378)
    void operator()(const GpuMat &img, const GpuMat &mask, GpuMat &keypoints)
    {
        push();
        int maxf = 0;
        miCheck(mi_surf_max_features(h_, img.rows, img.cols, &maxf));
        ensureCapacity(keypoints, ROWS_COUNT, maxf, CV_32FC1);   // ensureSizeIsEnough(ROWS_COUNT, maxFeatures) :179
        mi_mat i = miMat(img), m = miMat(mask), k = miMat(keypoints);
        int n = 0;
        miCheck(mi_surf_detect(h_, &i, mask.empty() ? nullptr : &m, &k, &n, nullptr));
        keypoints.cols = n;   // :209
    }
    //! finds the keypoints and computes their descriptors (surf.cuda.cpp:380-397)
    void operator()(const GpuMat &img, const GpuMat &mask, GpuMat &keypoints, GpuMat &descriptors, bool useProvidedKeypoints = false)
    {
        push();
        mi_mat i = miMat(img);
        if (!useProvidedKeypoints) {
            // one enqueue for the frame: the descriptor kernels read the feature count on the device (mi_surf_detect_and_compute),
            // no read-back of keypoints.cols between detectKeypoints and computeDescriptors (surf.cuda.cpp:205-209)
            int maxf = 0;
            miCheck(mi_surf_max_features(h_, img.rows, img.cols, &maxf));
            ensureCapacity(keypoints, ROWS_COUNT, maxf, CV_32FC1);
            // the count is not known before the enqueue: the descriptors of up to maxFeatures keypoints are written into the caller's matrix
            // where its ALLOCATION holds them (the matrix a previous frame got back, cut to its count), else into a new one that replaces
            // it.  No keypoint: the caller's matrix is left as it was, as computeDescriptors leaves it (surf.cuda.cpp:227-236).
            const bool fits = holds(descriptors, maxf, descriptorSize(), CV_32FC1);
            const int rows0 = descriptors.rows, cols0 = descriptors.cols;
            GpuMat fresh;
            if (fits) { descriptors.rows = maxf; descriptors.cols = descriptorSize(); } else fresh.create(maxf, descriptorSize(), CV_32FC1);
            GpuMat &dst = fits ? descriptors : fresh;
            mi_mat m = miMat(mask), k = miMat(keypoints), d = miMat(dst);
            int n = 0;
            miCheck(mi_surf_detect_and_compute(h_, &i, mask.empty() ? nullptr : &m, &k, &d, &n, nullptr));
            keypoints.cols = n;   // :209
            if (n > 0) { if (!fits) descriptors = fresh; descriptors.rows = n; }
            else if (fits) { descriptors.rows = rows0; descriptors.cols = cols0; }
            return;
        }
        if (!upright) { mi_mat k = miMat(keypoints); miCheck(mi_surf_compute_orientation(h_, &i, &k, keypoints.cols, nullptr)); }
        const int n = keypoints.cols;
        if (n > 0) {
            ensureCapacity(descriptors, n, descriptorSize(), CV_32FC1);   // ensureSizeIsEnough(nFeatures, descriptorSize) :232
            mi_mat k = miMat(keypoints), d = miMat(descriptors);
            miCheck(mi_surf_compute_descriptors(h_, &i, &k, n, &d, nullptr));
            miCheck(mi_stream_synchronize(nullptr));   // SURF_CUDA has no stream parameter: every wrapper ends synchronised (surf.cu:923,928)
        }   // (no keypoint: descriptors stay as they were, surf.cuda.cpp:229-235)
    }
    void detect(const GpuMat &img, const GpuMat &mask, GpuMat &keypoints) { (*this)(img, mask, keypoints); }
    void detectWithDescriptors(const GpuMat &img, const GpuMat &mask, GpuMat &keypoints, GpuMat &descriptors, bool useProvidedKeypoints = false)
    {
        (*this)(img, mask, keypoints, descriptors, useProvidedKeypoints);
    }
    void operator()(const GpuMat &img, const GpuMat &mask, std::vector<KeyPoint> &keypoints)
    {
        GpuMat k; (*this)(img, mask, k); downloadKeypoints(k, keypoints);   // surf.cuda.cpp:399-406
    }
    void operator()(const GpuMat &img, const GpuMat &mask, std::vector<KeyPoint> &keypoints, GpuMat &descriptors, bool useProvidedKeypoints = false)
    {
        GpuMat k;
        if (useProvidedKeypoints) uploadKeypoints(keypoints, k);
        (*this)(img, mask, k, descriptors, useProvidedKeypoints);
        downloadKeypoints(k, keypoints);   // :408-419
    }
    void operator()(const GpuMat &img, const GpuMat &mask, std::vector<KeyPoint> &keypoints, std::vector<float> &descriptors, bool useProvidedKeypoints = false)
    {
        GpuMat d; (*this)(img, mask, keypoints, d, useProvidedKeypoints); downloadDescriptors(d, descriptors);   // :421-432
    }
    void releaseMemory() { mi_surf_release_memory(h_); sum.release(); mask1.release(); maskSum.release(); det.release(); trace.release(); maxPosBuffer.release(); }

    // SURF parameters (public fields, cuda.hpp:182-189)
    double hessianThreshold;
    int nOctaves;
    int nOctaveLayers;
    bool extended;
    bool upright;
    float keypointsRatio;
    // public scratch of the reference class (cuda.hpp:191-195): owned by the C handle here, these stay empty
    GpuMat sum, mask1, maskSum, det, trace, maxPosBuffer;

private:
    void init() { mi_surf_params p = params(); miCheck(mi_surf_create(&p, &h_)); }
    mi_surf_params params() const
    {
        mi_surf_params p;
        p.hessian_threshold = hessianThreshold; p.n_octaves = nOctaves; p.n_octave_layers = nOctaveLayers; p.extended = extended;
        p.keypoints_ratio = keypointsRatio; p.upright = upright;
        return p;
    }
    void push() { mi_surf_params p = params(); miCheck(mi_surf_set_params(h_, &p)); }   // public fields may have been edited
    mi_surf *h_ = nullptr;
};

}}  // namespace cv::cuda
#endif
