// Device side of the superres optical-flow adapters: the two data-format steps that sit between a caller's
// frames and the flow classes in cv::superres::GpuOpticalFlow::calc (superres/src/optical_flow.cpp:462-494).
#include "mi_common.h"

namespace mi {
namespace superres {

struct CvtArgs {
    const unsigned char *src;
    size_t sstep;
    unsigned char *dst;
    size_t dstep;
    int rows, cols;
    float scale;   // 255 / maxVal(source depth)
};

// saturate_cast<uchar>(float): round to nearest even, clamp (cudev saturate_cast == __float2int_rn + clamp)
__device__ __forceinline__ unsigned char sat_u8(float v)
{
    const int i = __float2int_rn(v);
    return (unsigned char)min(max(i, 0), 255);
}

// BGR -> gray exactly as cuda::cvtColor does it: integer types CV_DESCALE(b*1868 + g*9617 + r*4899, 14), float types
// 0.114 b + 0.587 g + 0.299 r (cudev color conversion constants B2Y/G2Y/R2Y, yuv_shift = 14); then the depth conversion
// saturate_cast<uchar>(scale * v) in binary32.  DEPTH: 0 = 8U, 2 = 16U, 5 = 32F; CN: 1, 3, 4.
template <int DEPTH, int CN>
__global__ __launch_bounds__(256) void k_to_gray8(CvtArgs A)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= A.cols || y >= A.rows) return;
    const unsigned char *row = A.src + (size_t)y * A.sstep;
    float v;
    if (DEPTH == 5) {
        const float *p = reinterpret_cast<const float *>(row) + (size_t)x * CN;
        v = CN == 1 ? p[0] : p[0] * 0.114f + p[1] * 0.587f + p[2] * 0.299f;
    } else if (DEPTH == 2) {
        const unsigned short *p = reinterpret_cast<const unsigned short *>(row) + (size_t)x * CN;
        v = CN == 1 ? (float)p[0] : (float)(unsigned short)((p[0] * 1868u + p[1] * 9617u + p[2] * 4899u + (1u << 13)) >> 14);
    } else {
        const unsigned char *p = row + (size_t)x * CN;
        v = CN == 1 ? (float)p[0] : (float)(unsigned char)((p[0] * 1868u + p[1] * 9617u + p[2] * 4899u + (1u << 13)) >> 14);
    }
    A.dst[(size_t)y * A.dstep + x] = (DEPTH == 0) ? (unsigned char)v : sat_u8(A.scale * v);
}

__global__ __launch_bounds__(256) void k_split2(const unsigned char *src, size_t sstep, unsigned char *u, size_t ustep, unsigned char *v,
                                                size_t vstep, int rows, int cols)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const float2 f = reinterpret_cast<const float2 *>(src + (size_t)y * sstep)[x];
    reinterpret_cast<float *>(u + (size_t)y * ustep)[x] = f.x;
    reinterpret_cast<float *>(v + (size_t)y * vstep)[x] = f.y;
}

template <int DEPTH>
static void launch_cvt(int cn, const CvtArgs &A, dim3 grid, hipStream_t st)
{
    if (cn == 1) hipLaunchKernelGGL((k_to_gray8<DEPTH, 1>), grid, dim3(256), 0, st, A);
    else if (cn == 3) hipLaunchKernelGGL((k_to_gray8<DEPTH, 3>), grid, dim3(256), 0, st, A);
    else hipLaunchKernelGGL((k_to_gray8<DEPTH, 4>), grid, dim3(256), 0, st, A);
}

}  // namespace superres
}  // namespace mi

using namespace mi;

extern "C" {

int mi_superres_to_gray8(const mi_mat *src, mi_mat *dst, void *stream)
{
    MI_REQUIRE(src && dst && src->data && dst->data, MI_ERR_BAD_ARG, "mi_superres_to_gray8: null matrix");
    const int depth = src->type & 7, cn = ((src->type >> 3) & 63) + 1;
    MI_REQUIRE(depth == 0 || depth == 2 || depth == 5, MI_ERR_BAD_TYPE, "mi_superres_to_gray8: source depth must be 8U, 16U or 32F");
    MI_REQUIRE(cn == 1 || cn == 3 || cn == 4, MI_ERR_BAD_TYPE, "mi_superres_to_gray8: 1, 3 or 4 channels");   // input_array_utility.cpp:168
    MI_REQUIRE(dst->type == MI_8UC1, MI_ERR_BAD_TYPE, "mi_superres_to_gray8: dst must be CV_8UC1");
    MI_REQUIRE(src->rows > 0 && src->cols > 0 && dst->rows == src->rows && dst->cols == src->cols, MI_ERR_BAD_SIZE,
               "mi_superres_to_gray8: size mismatch");
    int ndev = 0;
    MI_REQUIRE(hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0, MI_ERR_NO_DEVICE, "no HIP device");
    superres::CvtArgs A;
    A.src = (const unsigned char *)src->data; A.sstep = src->step;
    A.dst = (unsigned char *)dst->data; A.dstep = dst->step;
    A.rows = src->rows; A.cols = src->cols;
    // convertToDepth: scale = maxVals[CV_8U] / maxVals[sdepth] (double), applied by convertTo in binary32
    A.scale = depth == 0 ? 1.0f : depth == 2 ? (float)(255.0 / 65535.0) : 255.0f;
    const dim3 grid(div_up(A.cols, 64), div_up(A.rows, 4));
    hipStream_t st = (hipStream_t)stream;
    if (depth == 0) superres::launch_cvt<0>(cn, A, grid, st);
    else if (depth == 2) superres::launch_cvt<2>(cn, A, grid, st);
    else superres::launch_cvt<5>(cn, A, grid, st);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int mi_split_flow(const mi_mat *flow, mi_mat *u, mi_mat *v, void *stream)
{
    MI_REQUIRE(flow && u && v && flow->data && u->data && v->data, MI_ERR_BAD_ARG, "mi_split_flow: null matrix");
    MI_REQUIRE(flow->type == MI_32FC2 && u->type == MI_32FC1 && v->type == MI_32FC1, MI_ERR_BAD_TYPE,
               "mi_split_flow: flow CV_32FC2 -> u, v CV_32FC1");
    MI_REQUIRE(flow->rows > 0 && flow->cols > 0 && u->rows == flow->rows && u->cols == flow->cols && v->rows == flow->rows &&
                   v->cols == flow->cols, MI_ERR_BAD_SIZE, "mi_split_flow: size mismatch");
    int ndev = 0;
    MI_REQUIRE(hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0, MI_ERR_NO_DEVICE, "no HIP device");
    const dim3 grid(div_up(flow->cols, 64), div_up(flow->rows, 4));
    hipLaunchKernelGGL(superres::k_split2, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char *)flow->data, flow->step,
                       (unsigned char *)u->data, u->step, (unsigned char *)v->data, v->step, flow->rows, flow->cols);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // extern "C"
