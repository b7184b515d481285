// Internal launch API of the SURF HIP kernels (surf_kernels.hip).  Not part of the C-ABI.
#pragma once
#include "mi_common.h"

namespace mi {
namespace surf {

// integral image of a CV_8UC1 image into sum ((rows+1) x sld u32); V: rows x vld scratch, BT: integral_bands(rows) x vld scratch
int integral(const unsigned char *img, long long istep, int rows, int cols, bool clamp1, unsigned *V, unsigned *BT, int vld,
             unsigned *sum, int sld, hipStream_t s);
int integral_bands(int rows);
int det_trace(const unsigned *sum, int sld, int rows, int cols, int octave, int nOctaveLayers, float *det, float *trace, int dld,
              hipStream_t s);
// bits: nOctaveLayers*layer_rows*ceil(layer_cols/64) u64; rowcnt: nOctaveLayers*layer_rows + 1 u32
int find_maxima(const float *det, const float *trace, int dld, const unsigned *mask_sum, int sld, int rows, int cols, int octave,
                int nOctaveLayers, float thr, unsigned long long *bits, unsigned *rowcnt, unsigned *segcnt, int4 *cand, int max_candidates,
                unsigned *ncand, hipStream_t s);
int nms_segments(int cols);   // row segments k_nms_flag cuts a row of `cols` samples into (segcnt: layers x rows x segments)
// all octaves of a frame in one launch per stage (surf_kernels.hip: k_*_all)
struct FusedSizes { size_t plane_floats, bits_words, seg_counts, row_counts, geo_bytes, poly_words; };
bool fused_supported(int n_octaves, int nOctaveLayers);
void fused_sizes(int rows, int cols, int dld, int n_octaves, int nOctaveLayers, FusedSizes *z);
// poly: the table of octaves >= 1 addresses the polyphase planes of the integral image (detect_fused with lds_tiles & 4 and a poly buffer)
void fused_geometry(int sld, int n_octaves, int nOctaveLayers, void *geo_host, int rows = 0, int cols = 0, bool poly = false);
bool lds_geometry_self_check();      // the compile-time tap geometry of that path equals haar_geo's
int detect_fused(const unsigned *sum, const unsigned *mask_sum, int sld, int rows, int cols, int n_octaves, int nOctaveLayers, float thr,
                 float *det, float *trace, int dld, unsigned long long *bits, unsigned *rowcnt, unsigned *segcnt, int4 *cand, int max_candidates,
                 unsigned *ncand, void *tmp, const void *geo_dev, float *kp, int kld, int max_features, unsigned *nfeat, int lds_tiles, hipStream_t s,
                 unsigned long long *sbits = nullptr, unsigned *poly = nullptr);
// lds_tiles & 3: 0 = global taps, 1 = octave 0 on LDS tiles, 2 (needs sbits: as many words as octave 0's bits) = ... and its maxima flagged
// there; lds_tiles & 4 (needs poly: FusedSizes::poly_words words, and the geometry table built with poly = true): octaves >= 1 read
// their taps from polyphase planes of the integral image
// tmp: interp_tmp_bytes(max_candidates) bytes of scratch
int interpolate(const float *det, int dld, int rows, int cols, int octave, const int4 *cand, const unsigned *ncand, int max_candidates,
                void *tmp, float *kp, int kld, int max_features, unsigned *nfeat, hipStream_t s);
size_t interp_tmp_bytes(int max_candidates);
// nfeat_dev != nullptr: count read on the device (grid sized for n_or_max); else n_or_max features
int orientation(const unsigned *sum, int sld, int rows, int cols, float *kp, int kld, const unsigned *nfeat_dev, int n_or_max,
                bool upright, const float *apt /* [3][113] x, y, w */, hipStream_t s);
// nfeat_dev != nullptr: the count is read on the device and nfeat is its upper bound (desc must have that many rows)
int descriptors(const unsigned char *img, long long istep, int rows, int cols, const float *kp, int kld, int nfeat, bool extended,
                float *desc, long long dstep_floats, const float *dw /* [400] */, hipStream_t s, const unsigned *nfeat_dev = nullptr);
int dbg_scan(const unsigned *in_dev, unsigned *out_dev, hipStream_t s);

}  // namespace surf
}  // namespace mi
