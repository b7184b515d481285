// Internal launch API of the StereoBM HIP kernels (stereobm_kernels.hip).  Not part of the C-ABI.
#pragma once
#include "mi_common.h"

namespace mi {
namespace sbm {

// SSD block matching + winner-take-all (+ uniqueness verification pass when uniqueness_ratio > 0).
// disp must be zero-filled by the caller (stereobm.cu:506); minssd (rows x mstep uint32) is required when
// uniqueness_ratio > 0, optional otherwise.
int block_match(const unsigned char *left, long long lstep, const unsigned char *right, long long rstep, unsigned char *disp,
                long long dstep, unsigned *minssd, long long mstep, int rows, int cols, int ndisp, int winsz,
                int uniqueness_ratio, int emulate_edge, hipStream_t s);
// the same for `pairs` image pairs of one size in ONE launch (blockIdx.z = pair): the batch supplies the parallelism, so the row
// bands are taller and the 2R-row start-up of a band weighs less.  tab_dev: device array of the pairs' pointers; minssd: pairs x mpair
struct BmPair { const unsigned char *left, *right; unsigned char *disp; long long lstep, rstep, dstep; };
int block_match_batch(const BmPair *tab_dev, int pairs, unsigned *minssd, long long mstep, long long mpair, int rows, int cols, int ndisp,
                      int winsz, int uniqueness_ratio, int emulate_edge, hipStream_t s);
int prefilter_xsobel(const unsigned char *src, long long sstep, unsigned char *dst, long long dstep, int rows, int cols,
                     int cap, hipStream_t s);
int prefilter_norm(const unsigned char *src, long long sstep, unsigned char *dst, long long dstep, int rows, int cols,
                   int cap, int winsize, hipStream_t s);
// S: int scratch of textureness_scratch_dims() = sld x sh elements
void textureness_scratch_dims(int rows, int cols, int *sld, int *sh);
int zero_disp_batch(const BmPair *tab_dev, int pairs, int rows, int cols, hipStream_t s);   // disp := 0 of every pair, one launch
int textureness_fused(const unsigned char *img, long long istep, unsigned char *disp, long long dstep, const BmPair *tab_dev, int pairs,
                      int rows, int cols, int winsz, float avg_threshold, hipStream_t s);   // one launch, no scratch plane; bit-identical
int textureness(const unsigned char *img, long long istep, unsigned char *disp, long long dstep, int rows, int cols,
                int winsz, float avg_threshold, int *S, hipStream_t s);
int dbg_wave_min(const unsigned *in_dev, unsigned *out_dev, hipStream_t s);
int dbg_tmax16(const unsigned *in_dev /*[16][64]*/, unsigned *out_dev /*[64]*/, hipStream_t s);

}  // namespace sbm
}  // namespace mi
