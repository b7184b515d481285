/*
 * oracle/stereobm_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Sequential CPU restatement of cv::cuda::StereoBM (the CUDA block matcher; the reference has no CPU
 * implementation of THIS algorithm -- cv::StereoBM in the main repo is SAD / CV_16S):
 *   modules/cudastereo/src/stereobm.cpp:139-191          compute(): prefilter -> BM -> textureness
 *   modules/cudastereo/src/cuda/stereobm.cu:71-128       CalcSSD / MinSSD (window sum, batch argmin)
 *   modules/cudastereo/src/cuda/stereobm.cu:232-458      stereoKernel (batches of 8, uniqueness)
 *   modules/cudastereo/src/cuda/stereobm.cu:522-599      prefilter_xsobel / prefilter_norm
 *   modules/cudastereo/src/cuda/stereobm.cu:606-711      textureness post-filter
 * Everything is integer arithmetic except the uniqueness threshold (binary32, same operations) and the
 * textureness compare (see orc_sbm_textureness).
 * PARITY UNPINNED: the reference pins this path only with golden PNGs from opencv_extra
 * (cudastereo/test/test_stereo.cpp:64-158), absent here; the restatement is anchored on the cited lines.
 */
#include "stereobm_ref.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_W 128       /* stereobm.cu:60 */
#define N_DISPARITIES 8   /* stereobm.cu:61 */

static inline int clampi(int x, int a, int b) { return x < a ? a : (x > b ? b : x); }

/* stereobm.cu:522-536 */
void orc_sbm_prefilter_xsobel(const uint8_t *src, uint8_t *dst, int rows, int cols, int cap)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const uint8_t *r0 = src + (size_t)(y > 0 ? y - 1 : 0) * cols;
        const uint8_t *r1 = src + (size_t)y * cols;
        const uint8_t *r2 = src + (size_t)(y + 1 < rows ? y + 1 : rows - 1) * cols;
        for (int x = 0; x < cols; ++x) {
            const int xl = x > 0 ? x - 1 : 0, xr = x + 1 < cols ? x + 1 : cols - 1;
            int conv = r0[xl] * (-1) + r0[xr] * (1) + r1[xl] * (-2) + r1[xr] * (2) + r2[xl] * (-1) + r2[xr] * (1);
            int v = conv < -cap ? -cap : conv;
            v = v > cap ? cap : v;
            v += cap;
            if (v > 255) v = 255;
            dst[(size_t)y * cols + x] = (uint8_t)(v & 0xFF);
        }
    }
}

/* stereobm.cu:557-599, including the `x+1` used twice in the 5-point term (:568) and the integer
 * scale computation of the host wrapper (:591-592) */
void orc_sbm_prefilter_norm(const uint8_t *src, uint8_t *dst, int rows, int cols, int cap, int winsize)
{
    int scale_g = winsize * winsize / 8, scale_s = (1024 + scale_g) / (scale_g * 2);
    scale_g *= scale_s;
    const int W2 = winsize / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const uint8_t *ru = src + (size_t)(y > 0 ? y - 1 : 0) * cols;
            const uint8_t *rc = src + (size_t)y * cols;
            const uint8_t *rd = src + (size_t)(y + 1 < rows ? y + 1 : rows - 1) * cols;
            const int xr = x + 1 < cols ? x + 1 : cols - 1;
            const int cov1 = ru[x] * 1 + rc[xr] * 1 + rc[x] * 4 + rc[xr] * 1 + rd[x] * 1;
            int cov2 = 0;
            for (int i = -W2; i < W2 + 1; ++i)
                for (int j = -W2; j < W2 + 1; ++j)
                    cov2 += src[(size_t)clampi(y + i, 0, rows - 1) * cols + clampi(x + j, 0, cols - 1)];
            int res = (cov1 * scale_g - cov2 * scale_s) >> 10;
            res = clampi(res, -cap, cap) + cap;
            dst[(size_t)y * cols + x] = (uint8_t)res;
        }
}

/* One row of window SSDs for every disparity: ssd[d][X] for X in [ndisp+R, cols-R).
 * Column sums over the 2R+1 rows, then the horizontal window of CalcSSD (stereobm.cu:71-96):
 *   col_ssd[0] + cache (columns +1..+R) + cache2 (columns +R+1..+2R, counted from X-R)
 * where cache2 is taken from thread t+R's `cache`, which is 0 when that thread's own X >= cols-R
 * (emulate_edge: the truncated right half-window of SURVEY Appendix B Q1). */
static void row_ssd(const uint8_t *L, const uint8_t *Rt, int cols, int Y, int R, int ndisp, int emulate_edge,
                    uint32_t *colsum /* [cols] */, uint32_t *ssd /* [ndisp][cols] */)
{
    for (int d = 0; d < ndisp; ++d) {
        uint32_t *out = ssd + (size_t)d * cols;
        for (int x = ndisp; x < cols; ++x) {   /* columns a window can touch: x - d >= 1 */
            uint32_t s = 0;
            for (int dy = -R; dy <= R; ++dy) {
                const int a = L[(size_t)(Y + dy) * cols + x], b = Rt[(size_t)(Y + dy) * cols + x - d];
                s += (uint32_t)((a - b) * (a - b));
            }
            colsum[x] = s;
        }
        for (int X = ndisp + R; X < cols - R; ++X) {
            uint32_t left = 0, right = 0;
            for (int i = 0; i <= R; ++i) left += colsum[X - R + i];          /* col_ssd[0] + cache */
            for (int i = R + 1; i <= 2 * R; ++i) right += colsum[X - R + i]; /* cache2 */
            if (emulate_edge) {
                const int t = (X - ndisp - R) % BLOCK_W;   /* threadIdx.x, stereobm.cu:247 */
                if (t < BLOCK_W - R && X + R >= cols - R) right = 0;
            }
            out[X] = left + right;
        }
    }
}

int orc_sbm_block_match(const uint8_t *left, const uint8_t *right, int rows, int cols, int ndisp, int winsz,
                        int uniqueness_ratio, int emulate_edge, uint8_t *disp, uint32_t *min_ssd)
{
    const int R = winsz >> 1;
    if (!(0 < ndisp && ndisp <= 256) || ndisp % 8 != 0 || winsz % 2 != 1) return -1;  /* stereobm.cpp:143-146 */
    if (R == 0 || R > 25) return -1;                                                   /* stereobm.cu:502-504 */
    memset(disp, 0, (size_t)rows * cols);                                              /* :506 */
    if (min_ssd) memset(min_ssd, 0xFF, (size_t)rows * cols * sizeof(uint32_t));        /* :507 */
    if (cols - ndisp - 2 * R <= 0 || rows - 2 * R <= 0) return -3;  /* empty launch grid (:469-470) */
    int fail = 0;
#pragma omp parallel
    {
        uint32_t *colsum = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cols);
        uint32_t *ssd = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cols * ndisp);
        if (!colsum || !ssd) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 4)
            for (int Y = R; Y < rows - R; ++Y) {
                row_ssd(left, right, cols, Y, R, ndisp, emulate_edge, colsum, ssd);
                for (int X = ndisp + R; X < cols - R; ++X) {
                    /* per-pixel state of stereoKernel (stereobm.cu:236-283) */
                    uint32_t line_ssds[2 + N_DISPARITIES];
                    uint32_t *batch = line_ssds + 2;
                    uint32_t tail0 = UINT_MAX, tail1 = UINT_MAX, tail2 = UINT_MAX;
                    unsigned char local_disp = 0, approved = 1;
                    const float thresh_scale = (float)(1.0 + uniqueness_ratio / 100.0f);  /* :273 */
                    for (int d = 0; d < ndisp; d += N_DISPARITIES) {
                        /* MinSSD (:98-128): min of the batch, LAST index among equals */
                        for (int i = 0; i < N_DISPARITIES; ++i) batch[i] = ssd[(size_t)(d + i) * cols + X];
                        uint32_t mssd = batch[0];
                        for (int i = 1; i < N_DISPARITIES; ++i) if (batch[i] < mssd) mssd = batch[i];
                        int best = 0;
                        for (int i = 0; i < N_DISPARITIES; ++i) if (mssd == batch[i]) best = i;
                        /* :308-353 / :387-431 */
                        const uint32_t last_opt = tail0;
                        const uint32_t opt = last_opt < mssd ? last_opt : mssd;
                        if (uniqueness_ratio > 0) {
                            line_ssds[0] = tail1;
                            line_ssds[1] = tail2;
                            const float thresh = thresh_scale * (float)opt;
                            int dtest = local_disp;
                            if (mssd < last_opt) {
                                approved = 1;
                                dtest = d + best;
                                if ((local_disp < dtest - 1 || local_disp > dtest + 1) && ((float)last_opt <= thresh))
                                    approved = 0;
                            }
                            if (approved) {
                                for (int ld = d - 2; ld < d + N_DISPARITIES; ++ld)
                                    if ((ld < dtest - 1 || ld > dtest + 1) && ((float)line_ssds[ld - d + 2] <= thresh)) {
                                        approved = 0;
                                        break;
                                    }
                            }
                            tail1 = batch[6];
                            tail2 = batch[7];
                        }
                        tail0 = opt;
                        if (mssd < last_opt) local_disp = (unsigned char)(d + best);
                    }
                    if (min_ssd) min_ssd[(size_t)Y * cols + X] = tail0;
                    disp[(size_t)Y * cols + X] = uniqueness_ratio > 0 ? (uint8_t)(local_disp * approved) : local_disp;
                }
            }
        }
        free(colsum);
        free(ssd);
    }
    return fail ? -5 : 0;
}

/* Textureness post-filter, stereobm.cu:606-711.  The reference reads the image through a
 * linear-filtered, normalised-float texture at INTEGER unnormalised coordinates, i.e. the hardware
 * returns the bilinear blend centred at (x-1/2, y-1/2): the mean of texels (x-1,y-1),(x,y-1),(x-1,y),
 * (x,y) divided by 255 (weights exactly 1/4; clamp addressing -- wrap is not honoured for
 * unnormalised coordinates).  DEFINITION used here: with
 *   B(x,y) = sum of those 4 texels (an integer, 4 x the blended value x 255),
 *   S(x,y) = |-B(x-1,y-1) + B(x+1,y-1) - 2B(x-1,y) + 2B(x+1,y) - B(x-1,y+1) + B(x+1,y+1)|,
 * a disparity is zeroed when  (float)(sum of S over the winsz x winsz window) * 0.25f < thr * winsz^2
 * (threshold product in binary32 as in :700).  The reference accumulates the same quantity in binary32
 * with a running sum down the rows; the two agree except when the window sum lies within float
 * rounding noise of the threshold. */
static inline int box4(const uint8_t *img, int rows, int cols, int x, int y)
{
    const int x0 = clampi(x - 1, 0, cols - 1), x1 = clampi(x, 0, cols - 1);
    const int y0 = clampi(y - 1, 0, rows - 1), y1 = clampi(y, 0, rows - 1);
    return img[(size_t)y0 * cols + x0] + img[(size_t)y0 * cols + x1] + img[(size_t)y1 * cols + x0] + img[(size_t)y1 * cols + x1];
}

static inline int tex_sobel(const uint8_t *img, int rows, int cols, int x, int y)
{
    const int c = -box4(img, rows, cols, x - 1, y - 1) + box4(img, rows, cols, x + 1, y - 1)
                  - 2 * box4(img, rows, cols, x - 1, y) + 2 * box4(img, rows, cols, x + 1, y)
                  - box4(img, rows, cols, x - 1, y + 1) + box4(img, rows, cols, x + 1, y + 1);
    return c < 0 ? -c : c;
}

void orc_sbm_textureness(const uint8_t *img, int rows, int cols, int winsz, float avg_threshold, uint8_t *disp)
{
    const int R = winsz / 2;
    const float threshold = avg_threshold * (float)(winsz * winsz);  /* stereobm.cu:700 */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            if (!disp[(size_t)y * cols + x]) continue;  /* zeroing a zero is a no-op */
            long long s = 0;
            for (int i = -R; i <= R; ++i)
                for (int j = -R; j <= R; ++j) s += tex_sobel(img, rows, cols, x + j, y + i);
            if ((float)s * 0.25f < threshold) disp[(size_t)y * cols + x] = 0;
        }
}

void orc_sbm_default_params(orc_sbm_params *p)
{
    /* createStereoBM(64, 19) cudastereo.hpp:90; StereoBMImpl ctor stereobm.cpp:129-132 */
    p->num_disparities = 64; p->block_size = 19; p->prefilter_type = -1; p->prefilter_cap = 31;
    p->prefilter_size = 9; p->texture_threshold = 3.0f; p->uniqueness_ratio = 0; p->emulate_edge = 1;
}

int orc_sbm_compute(const orc_sbm_params *p, const uint8_t *left, const uint8_t *right, int rows, int cols, uint8_t *disp)
{
    const uint8_t *le = left, *ri = right;
    uint8_t *lb = NULL, *rb = NULL;
    if (p->prefilter_type == 1 || p->prefilter_type == 0) {
        lb = (uint8_t *)malloc((size_t)rows * cols);
        rb = (uint8_t *)malloc((size_t)rows * cols);
        if (!lb || !rb) { free(lb); free(rb); return -5; }
        if (p->prefilter_type == 1) {   /* cv::StereoBM::PREFILTER_XSOBEL, stereobm.cpp:164-173 */
            orc_sbm_prefilter_xsobel(left, lb, rows, cols, p->prefilter_cap);
            orc_sbm_prefilter_xsobel(right, rb, rows, cols, p->prefilter_cap);
        } else {                        /* PREFILTER_NORMALIZED_RESPONSE, :175-185 */
            orc_sbm_prefilter_norm(left, lb, rows, cols, p->prefilter_cap, p->prefilter_size);
            orc_sbm_prefilter_norm(right, rb, rows, cols, p->prefilter_cap, p->prefilter_size);
        }
        le = lb; ri = rb;
    }
    int rc = orc_sbm_block_match(le, ri, rows, cols, p->num_disparities, p->block_size, p->uniqueness_ratio,
                                 p->emulate_edge, disp, NULL);
    if (rc == 0 && p->texture_threshold > 0)  /* stereobm.cpp:189-190 */
        orc_sbm_textureness(le, rows, cols, p->block_size, p->texture_threshold, disp);
    free(lb);
    free(rb);
    return rc;
}
