/*
 * oracle/refshim/cudashim/stereobm_cu_host.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * C entry points over the reference's own StereoBM CUDA source (modules/cudastereo/src/cuda/stereobm.cu, rewritten only at its
 * launch sites by cu2host.py and run on the CPU through cudashim.h), called the way the host class does
 * (modules/cudastereo/src/stereobm.cpp:54-63, 139-191).  Dense u8 images, step = cols.
 */
#include "opencv2/core/cuda/common.hpp"
#include <vector>

namespace cv { namespace cuda { namespace device { namespace stereobm {
void stereoBM_CUDA(const PtrStepSzb &left, const PtrStepSzb &right, const PtrStepSzb &disp, int ndisp, int winsz, int uniquenessRatio,
                   const PtrStepSz<unsigned int> &minSSD_buf, cudaStream_t &stream);
void prefilter_xsobel(const PtrStepSzb &input, const PtrStepSzb &output, int prefilterCap, cudaStream_t &stream);
void prefilter_norm(const PtrStepSzb &input, const PtrStepSzb &output, int prefilterCap, int winsize, cudaStream_t &stream);
void postfilter_textureness(const PtrStepSzb &input, int winsz, float avgTexturenessThreshold, const PtrStepSzb &disp, cudaStream_t &stream);
}}}}

using namespace cv::cuda;
namespace sbm = cv::cuda::device::stereobm;

extern "C" {

int ref_cu_sbm_block_match(const unsigned char *left, const unsigned char *right, int rows, int cols, int ndisp, int winsz, int uniqueness_ratio,
                           unsigned char *disp, unsigned int *min_ssd)
{
    try {
        cudaStream_t st = nullptr;
        std::vector<unsigned int> tmp;
        if (!min_ssd) { tmp.resize((size_t)rows * cols); min_ssd = tmp.data(); }
        sbm::stereoBM_CUDA(PtrStepSzb(rows, cols, (unsigned char *)left, cols), PtrStepSzb(rows, cols, (unsigned char *)right, cols),
                           PtrStepSzb(rows, cols, disp, cols), ndisp, winsz, uniqueness_ratio,
                           PtrStepSz<unsigned int>(rows, cols, min_ssd, (size_t)cols * 4), st);
        return 0;
    } catch (const std::exception &) { return -1; }
}

int ref_cu_sbm_prefilter_xsobel(const unsigned char *src, int rows, int cols, int cap, unsigned char *dst)
{
    cudaStream_t st = nullptr;
    sbm::prefilter_xsobel(PtrStepSzb(rows, cols, (unsigned char *)src, cols), PtrStepSzb(rows, cols, dst, cols), cap, st);
    return 0;
}

int ref_cu_sbm_prefilter_norm(const unsigned char *src, int rows, int cols, int cap, int winsize, unsigned char *dst)
{
    cudaStream_t st = nullptr;
    sbm::prefilter_norm(PtrStepSzb(rows, cols, (unsigned char *)src, cols), PtrStepSzb(rows, cols, dst, cols), cap, winsize, st);
    return 0;
}

int ref_cu_sbm_textureness(const unsigned char *img, int rows, int cols, int winsz, float avg_threshold, unsigned char *disp)
{
    cudaStream_t st = nullptr;
    sbm::postfilter_textureness(PtrStepSzb(rows, cols, (unsigned char *)img, cols), winsz, avg_threshold, PtrStepSzb(rows, cols, disp, cols), st);
    return 0;
}

}  // extern "C"
