// Internal launch API of the Farneback HIP kernels (farneback_kernels.hip).  Not part of the C-ABI.
#pragma once
#include "mi_common.h"

#define MI_FB_MAX_KSIZE_HALF 100   /* MAX_KSIZE_HALF, cudaoptflow/src/cuda/farneback.cu:56 */
enum { MI_BORDER_REPLICATE = 1, MI_BORDER_REFLECT101 = 4 };   /* cv::BorderTypes */

namespace mi {
namespace fb {

struct Plane {                                  // dense f32 plane(s), ld floats per row
    int w, h, ld;
    long long bs;   // floats between the same plane of consecutive pairs of a batch (blockIdx.z = pair); 0 for a single pair
    int batch;
};
struct Taps { float k[MI_FB_MAX_KSIZE_HALF + 1]; };   // centre + positive half of a symmetric kernel (by value)
struct PolyC { float g[8], xg[8], xxg[8]; float ig11, ig03, ig33, ig55; };

inline Plane plane_of(int w, int h, long long bs = 0, int batch = 1) { Plane p; p.w = w; p.h = h; p.ld = align_up(w, 64); p.bs = bs; p.batch = batch; return p; }

// by-value table of the caller's matrices for the batched format steps (blockIdx.z = pair)
constexpr int kFmtPairs = 64;
struct FmtTab { const void *a[kFmtPairs], *b[kFmtPairs]; long long sa[kFmtPairs], sb[kFmtPairs]; };
int convert_batch(const FmtTab &T, int n, int type, float *A, float *B, long long bs, const Plane &g, hipStream_t s);   // A, B: pair 0's planes
int merge_flow_batch(const float *fx, const float *fy, const FmtTab &T, int n, long long bs, const Plane &g, hipStream_t s);   // T.a / T.sa: the flow matrices
int convert(const void *a, long long sa, const void *b, long long sb, int type, float *A, float *B, const Plane &g, hipStream_t s);
int split_flow(const void *flow, long long sf, float *fx, float *fy, const Plane &g, hipStream_t s);
int merge_flow(const float *fx, const float *fy, void *flow, long long sf, const Plane &g, hipStream_t s);
// nf = 2: both frames of every pair in one launch (blockIdx.z = pair * 2 + frame; frame 1 lies fs floats behind frame 0 in src / dst)
int gaussian_blur(const float *src, float *dst, const Plane &g, int kh, const Taps &K, int border, hipStream_t s, int nf = 1, long long fs = 0);
// the same (REFLECT101, both frames of g.batch <= kFmtPairs pairs) reading the caller's CV_8UC1 / CV_32FC1 matrices T.a / T.b instead of converted planes
bool gaussian_blur_tab_ok(const Plane &g, int kh);
int gaussian_blur_tab(const FmtTab &T, int type, float *dst, const Plane &g, int kh, const Taps &K, hipStream_t s, long long fs);
// resize_from: src is a plane of that (larger) geometry and the expansion reads its cuda::resize to g, sampled on the fly
int poly_exp(const float *src, float *dst5, const Plane &g, int polyN, const PolyC &C, hipStream_t s, int nf = 1, long long fs_src = 0, long long fs_dst = 0,
             const Plane *resize_from = nullptr);
// flow = resize(prev) * alpha (stored) and M = updateMatrices(flow) in one launch
int update_matrices_resized(const float *prevx, const float *prevy, const Plane &gprev, float alpha, float *flowx, float *flowy, const float *R0,
                            const float *R1, float *M, const Plane &g, hipStream_t s);
int update_matrices(const float *flowx, const float *flowy, const float *R0, const float *R1, float *M, const Plane &g, hipStream_t s);
// fused blur5 (box when gauss == nullptr) + updateFlow + (update ? updateMatrices -> Mout)
// merged / merged_step: the caller's CV_32FC2 flow matrix of a SINGLE pair, written together with the planes (tiled kernels only; *did_merge says)
int iterate(const float *M, const float *R0, const float *R1, float *flowx, float *flowy, float *Mout, const Plane &g, int ksize,
            const Taps *gauss, bool update, hipStream_t s, void *merged = nullptr, long long merged_step = 0, bool *did_merge = nullptr);
// two iterations in one launch (64 x 4 tiles with a recomputed halo; M and Mout must differ: ONE buffer swap per call); bit-identical to
// iterate(update = true) followed by iterate(update)
bool iterate2_supported(int ksize);
int iterate2(const float *M, const float *R0, const float *R1, float *flowx, float *flowy, float *Mout, const Plane &g, int ksize, const Taps *gauss,
             bool update, hipStream_t s, void *merged = nullptr, long long merged_step = 0, bool *did_merge = nullptr);
int blur5(const float *M, float *dst, const Plane &g, int ksize, const Taps *gauss, hipStream_t s);
int update_flow(const float *M, float *flowx, float *flowy, const Plane &g, hipStream_t s);
int pyr_down(const float *src, const Plane &gs, float *dst, const Plane &gd, hipStream_t s);

// host math (farneback.cpp:209-276 and the main-repo cv::getGaussianKernel)
void gaussian_kernel(int n, double sigma, float *k);
int prepare_gaussian(int n, double sigma, PolyC *C);

}  // namespace fb
}  // namespace mi
