/* CPU restatement of cv::cuda::StereoSGM -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * PINNED on the reference's own CPU restatements: census_transform(), path_aggregation() and winner_takes_all_left() below
 * follow modules/cudastereo/test/test_sgm_funcs.cpp:131-151, 225-260 and 354-403, the functions the reference's unit tests
 * compare its CUDA kernels against with EXPECT_MAT_NEAR(..., 0) (the images those tests use are random matrices generated at
 * run time, not files).  The stages the reference does not test against a CPU twin are restated from the kernels:
 * right-image winner-takes-all (cuda/stereosgm.cu:1524-1541,1560-1568), 3x3 median (:1719-1922), left-right consistency check
 * (:1959-1980) and the disparity range correction (:2031-2055); pipeline and defaults from src/stereosgm.cpp:84-144.
 *
 * Two reference quirks are switchable (emulate_quirks, default on in the product, as for StereoBM):
 *   Q-a  median_kernel_3x3_16u_v2 sorts the pixels of its scalar fallback columns through a uint8_t buffer
 *        (stereosgm.cu:1871-1896): column 1 and the last column pair that does not fit the vector path get the median of the
 *        LOW BYTES of the 16-bit disparities;
 *   Q-b  check_consistency launches width/16 x height/16 blocks (stereosgm.cu:1985), so the rightmost width%16 columns and the
 *        bottom height%16 rows are never checked.
 * One definition where the reference leaves memory undefined: the 1-pixel border the median kernels never write
 * (ensureSizeIsEnough'ed, uninitialised buffers) is defined as a copy of the median's input.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int min_disparity, num_disparities, P1, P2, uniqueness_ratio, mode, emulate_quirks;
} orc_sgm_params;

#define SGM_INVALID (-1)
#define DISP_SHIFT 4

static inline int popc32(uint32_t v) { return __builtin_popcount(v); }

/* test_sgm_funcs.cpp:131-151; src: rows x cols, 8- or 16-bit (es = 1 | 2) */
void orc_sgm_census(const void *src, int es, int rows, int cols, int32_t *dst)
{
    const int hor = 9 / 2, ver = 7 / 2;
    memset(dst, 0, sizeof(int32_t) * (size_t)rows * cols);
    for (int y = ver; y < rows - ver; ++y)
        for (int x = hor; x < cols - hor; ++x) {
            int32_t value = 0;
            for (int dy = -ver; dy <= 0; ++dy)
                for (int dx = -hor; dx <= (dy == 0 ? -1 : hor); ++dx) {
                    const size_t ia = (size_t)(y + dy) * cols + (x + dx), ib = (size_t)(y - dy) * cols + (x - dx);
                    const int a = es == 1 ? ((const uint8_t *)src)[ia] : ((const uint16_t *)src)[ia];
                    const int b = es == 1 ? ((const uint8_t *)src)[ib] : ((const uint16_t *)src)[ib];
                    value = (int32_t)((uint32_t)value << 1);
                    if (a > b) value |= 1;
                }
            dst[(size_t)y * cols + x] = value;
        }
}

/* test_sgm_funcs.cpp:225-260: one path direction (dx, dy); dst: width * height * max_disparity bytes */
void orc_sgm_path(const int32_t *left, const int32_t *right, uint8_t *dst, int width, int height, int max_disparity, int min_disparity,
                  int p1, int p2, int dx, int dy)
{
    int *before = (int *)malloc(sizeof(int) * max_disparity);
    for (int i = (dy < 0 ? height - 1 : 0); 0 <= i && i < height; i += (dy < 0 ? -1 : 1))
        for (int j = (dx < 0 ? width - 1 : 0); 0 <= j && j < width; j += (dx < 0 ? -1 : 1)) {
            const int i2 = i - dy, j2 = j - dx;
            const int inside = (0 <= i2 && i2 < height && 0 <= j2 && j2 < width);
            int min_cost = 1 << 30;
            for (int k = 0; k < max_disparity; ++k) {
                before[k] = inside ? dst[(size_t)k + ((size_t)j2 + (size_t)i2 * width) * max_disparity] : 0;
                if (before[k] < min_cost) min_cost = before[k];
            }
            const uint32_t l = (uint32_t)left[(size_t)i * width + j];
            for (int k = 0; k < max_disparity; ++k) {
                /* (j - k - min_disparity >= width only with a negative minDisparity, where the reference's twin reads out of bounds: 0) */
                const uint32_t r = (k + min_disparity > j || j - k - min_disparity >= width) ? 0u : (uint32_t)right[(size_t)i * width + (j - k - min_disparity)];
                int cost = before[k] - min_cost < p2 ? before[k] - min_cost : p2;
                if (k > 0 && before[k - 1] - min_cost + p1 < cost) cost = before[k - 1] - min_cost + p1;
                if (k + 1 < max_disparity && before[k + 1] - min_cost + p1 < cost) cost = before[k + 1] - min_cost + p1;
                cost += popc32(l ^ r);
                dst[(size_t)k + ((size_t)j + (size_t)i * width) * max_disparity] = (uint8_t)cost;
            }
        }
    free(before);
}

/* path order of PathAggregation::operator(), stereosgm.cu:1352-1362 */
static const int PATH_DX[8] = {0, 0, 1, -1, 1, -1, -1, 1};
static const int PATH_DY[8] = {1, -1, 0, 0, 1, 1, -1, -1};

/* left: test_sgm_funcs.cpp:354-403 (uint16 result, -1 = 0xffff); right: stereosgm.cu:1524-1541,1560-1568 */
void orc_sgm_wta(const uint8_t *src, int16_t *left, int16_t *right, int width, int height, int disparity, int num_paths, float uniqueness,
                 int subpixel)
{
    const size_t cost_step = (size_t)disparity * width * height;
    int *sum = (int *)malloc(sizeof(int) * (size_t)width * disparity);
    for (int i = 0; i < height; ++i) {
        for (int j = 0; j < width; ++j)
            for (int k = 0; k < disparity; ++k) {
                int s = 0;
                for (int p = 0; p < num_paths; ++p) s += src[p * cost_step + ((size_t)i * width + j) * disparity + k];
                sum[(size_t)j * disparity + k] = s;
            }
        for (int j = 0; j < width; ++j) {
            const int *v = sum + (size_t)j * disparity;
            int best_cost = v[0], best_disp = 0;
            for (int k = 1; k < disparity; ++k)
                if (v[k] < best_cost) { best_cost = v[k]; best_disp = k; }   /* min_element on (cost, disp) pairs */
            int ans = best_disp;
            if (subpixel) {
                ans <<= DISP_SHIFT;
                if (0 < best_disp && best_disp < disparity - 1) {
                    const int l = v[best_disp - 1], r = v[best_disp + 1];
                    const int numer = l - r, denom = l - 2 * best_cost + r;
                    ans += ((numer << DISP_SHIFT) + denom) / (2 * denom);
                }
            }
            for (int k = 0; k < disparity; ++k)
                if (v[k] * uniqueness < best_cost && abs(k - best_disp) > 1) { ans = SGM_INVALID; break; }
            left[(size_t)i * width + j] = (int16_t)ans;
        }
        for (int p = 0; p < width; ++p) {   /* right pixel p: argmin over d of the summed cost at left pixel p + d, lowest d on ties */
            uint32_t best = 0xffffffffu;
            for (int d = 0; d < disparity && p + d < width; ++d) {
                const uint32_t packed = ((uint32_t)sum[(size_t)(p + d) * disparity + d] << 16) | (uint32_t)d;
                if (packed < best) best = packed;
            }
            right[(size_t)i * width + p] = (int16_t)(best & 0xffffu);
        }
    }
    free(sum);
}

static uint16_t median9(uint16_t *b)
{
    /* median_selection_network_9, stereosgm.cu:1699-1716 (any correct median: the network's result is the 5th smallest) */
    for (int i = 1; i < 9; ++i) {
        const uint16_t v = b[i];
        int j = i - 1;
        while (j >= 0 && b[j] > v) { b[j + 1] = b[j]; --j; }
        b[j + 1] = v;
    }
    return b[4];
}

/* 3x3 median on uint16 (the bit pattern of the int16 map, like the reference's median_filter<uint16_t>), border = copy */
void orc_sgm_median(const int16_t *src, int16_t *dst, int rows, int cols, int emulate_quirks)
{
    memcpy(dst, src, sizeof(int16_t) * (size_t)rows * cols);
    for (int y = 1; y < rows - 1; ++y)
        for (int x = 1; x < cols - 1; ++x) {
            /* scalar fallback columns of median_kernel_3x3_16u_v2 (stereosgm.cu:1871-1896) */
            const int x2 = x & ~1;
            const int scalar_col = (x2 == 0) || !(x2 >= 2 && x2 + 3 < cols);
            uint16_t b[9];
            for (int i = 0; i < 9; ++i) {
                const uint16_t v = (uint16_t)src[(size_t)(y - 1 + i / 3) * cols + (x - 1 + i % 3)];
                b[i] = (emulate_quirks && scalar_col) ? (uint16_t)(uint8_t)v : v;
            }
            dst[(size_t)y * cols + x] = (int16_t)median9(b);
        }
}

/* stereosgm.cu:1959-1980 (+ the launch geometry of :1985) and :2031-2055, in place on left_disp */
void orc_sgm_check_and_range(int16_t *left_disp, const int16_t *right_disp, const void *left_img, int es, int rows, int cols, int subpixel,
                             int min_disp, int emulate_quirks)
{
    const int cw = emulate_quirks ? (cols / 16) * 16 : cols, ch = emulate_quirks ? (rows / 16) * 16 : rows;
    for (int i = 0; i < ch; ++i)
        for (int j = 0; j < cw; ++j) {
            const int mask = es == 1 ? ((const uint8_t *)left_img)[(size_t)i * cols + j] : ((const uint16_t *)left_img)[(size_t)i * cols + j];
            const uint16_t org = (uint16_t)left_disp[(size_t)i * cols + j];
            int d = org;
            if (subpixel) d >>= DISP_SHIFT;
            const int k = j - d;
            /* the reference's `org == INVALID_DISP` branch re-writes INVALID_DISP over an already invalid pixel: no effect */
            if (mask == 0 || (k >= 0 && k < cols && abs((int)(uint16_t)right_disp[(size_t)i * cols + k] - d) > 1))
                left_disp[(size_t)i * cols + j] = (int16_t)SGM_INVALID;
        }
    const int scale = subpixel ? 16 : 1;
    const int min_scaled = min_disp * scale, inv_scaled = (min_disp - 1) * scale;
    for (size_t i = 0; i < (size_t)rows * cols; ++i) {
        uint16_t d = (uint16_t)left_disp[i];
        d = (d == (uint16_t)SGM_INVALID) ? (uint16_t)inv_scaled : (uint16_t)(d + min_scaled);
        left_disp[i] = (int16_t)d;
    }
}

/* StereoSGMImpl::compute, stereosgm.cpp:96-144.  left/right: 8- or 16-bit images; disp: int16 rows x cols. */
int orc_sgm_compute(const orc_sgm_params *p, const void *left, const void *right, int es, int rows, int cols, int16_t *disp)
{
    if (p->mode != 1 && p->mode != 3) return -1;                                            /* MODE_HH / MODE_HH4 */
    if (p->num_disparities != 64 && p->num_disparities != 128 && p->num_disparities != 256) return -2;
    const int np = p->mode == 3 ? 4 : 8, D = p->num_disparities;
    const size_t n = (size_t)rows * cols;
    int32_t *cl = (int32_t *)malloc(sizeof(int32_t) * n), *cr = (int32_t *)malloc(sizeof(int32_t) * n);
    uint8_t *agg = (uint8_t *)malloc(n * D * np);
    int16_t *lt = (int16_t *)malloc(sizeof(int16_t) * n), *rt = (int16_t *)malloc(sizeof(int16_t) * n), *rm = (int16_t *)malloc(sizeof(int16_t) * n);
    orc_sgm_census(left, es, rows, cols, cl);
    orc_sgm_census(right, es, rows, cols, cr);
#pragma omp parallel for schedule(dynamic)
    for (int q = 0; q < np; ++q)
        orc_sgm_path(cl, cr, agg + (size_t)q * n * D, cols, rows, D, p->min_disparity, p->P1, p->P2, PATH_DX[q], PATH_DY[q]);
    orc_sgm_wta(agg, lt, rt, cols, rows, D, np, (float)(100 - p->uniqueness_ratio) / 100, 1);
    orc_sgm_median(lt, disp, rows, cols, p->emulate_quirks);
    orc_sgm_median(rt, rm, rows, cols, p->emulate_quirks);
    orc_sgm_check_and_range(disp, rm, left, es, rows, cols, 1, p->min_disparity, p->emulate_quirks);
    free(cl); free(cr); free(agg); free(lt); free(rt); free(rm);
    return 0;
}
