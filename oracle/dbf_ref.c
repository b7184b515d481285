/* CPU restatement of cv::cuda::DisparityBilateralFilter -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Follows modules/cudastereo/src/disparity_bilateral_filter.cpp:96-175 (tables, thresholds, apply) and
 * modules/cudastereo/src/cuda/disparity_bilateral_filter.cu:54-199 (the per-pixel refinement and the red/black pass order).
 *
 * One deliberate definition: the reference updates the disparity IN PLACE while other threads of the same pass read it
 * (disparity_bilateral_filter.cu:98-118 reads every pixel of the (2r+1)^2 window, including pixels of the colour being
 * written by this very launch), so its result depends on thread timing.  Here -- and in the HIP kernel -- every pass reads
 * the image as it was when the pass started (pixels of the other colour cannot change during a pass anyway), which is the
 * only deterministic reading of that code.  parity unpinned: the reference's golden (stereobm/aloe-disp.png family) is in
 * opencv_extra, which is absent.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int ndisp, radius, iters;
    float edge_threshold, max_disc_threshold, sigma_range;
} orc_dbf_params;

/* calc_color_weighted_table, disparity_bilateral_filter.cpp:96-106 */
static void color_table(float *t, float sigma_range, int len)
{
    const float den = 2 * sigma_range * sigma_range;   /* float arithmetic, then promoted in the division */
    for (int i = 0; i < len; ++i) t[i] = (float)exp(-(double)(i * i) / den);
}

/* calc_space_weighted_filter, :108-123 (float overloads of sqrt / exp) */
static void space_table(float *t, int half, float dist_space)
{
    for (int y = 0; y <= half; ++y)
        for (int x = 0; x <= half; ++x) t[y * (half + 1) + x] = expf(-sqrtf((float)(y * y) + (float)(x * x)) / dist_space);
}

static inline int iabs(int v) { return v < 0 ? -v : v; }

/* disp: rows x cols, element size es (1: uchar, 2: short), dense.  img: rows x cols x cn uchar, dense.  In/out in `disp`. */
int orc_dbf_apply(const orc_dbf_params *p, void *disp, int es, const unsigned char *img, int cn, int rows, int cols)
{
    if (!(p->ndisp > 0 && p->radius > 0 && p->iters > 0)) return -1;   /* CV_Assert, .cpp:176 */
    if ((es != 1 && es != 2) || (cn != 1 && cn != 3)) return -2;
    const int r = p->radius, half = r;
    float ctab[255];
    float *stab = (float *)malloc(sizeof(float) * (half + 1) * (half + 1));
    color_table(ctab, p->sigma_range, 255);
    space_table(stab, half, r + 1.0f);
    /* .cpp:146-147 */
    short edge_disc = (short)(p->ndisp * p->edge_threshold + 0.5);
    if (edge_disc < 1) edge_disc = 1;
    const short max_disc = (short)(p->ndisp * p->max_disc_threshold + 0.5);
    const size_t n = (size_t)rows * cols;
    int *cur = (int *)malloc(sizeof(int) * n), *nxt = (int *)malloc(sizeof(int) * n);
    for (size_t i = 0; i < n; ++i) cur[i] = es == 1 ? ((unsigned char *)disp)[i] : ((short *)disp)[i];
    for (int it = 0; it < p->iters; ++it)
        for (int t = 0; t < 2; ++t) {   /* .cu:164-170: pass t updates the pixels with (x + y + t) odd ... x = 2k + ((y + t) & 1) */
            memcpy(nxt, cur, sizeof(int) * n);
#pragma omp parallel for schedule(static)
            for (int y = 1; y < rows - 1; ++y)
                for (int x = 1; x < cols - 1; ++x) {
                    if ((x & 1) != ((y + t) & 1)) continue;
                    int dp[5];
                    dp[0] = cur[(size_t)y * cols + x];
                    dp[1] = cur[(size_t)(y - 1) * cols + x];
                    dp[2] = cur[(size_t)y * cols + x - 1];
                    dp[3] = cur[(size_t)(y + 1) * cols + x];
                    dp[4] = cur[(size_t)y * cols + x + 1];
                    if (!(iabs(dp[1] - dp[0]) >= edge_disc || iabs(dp[2] - dp[0]) >= edge_disc || iabs(dp[3] - dp[0]) >= edge_disc ||
                          iabs(dp[4] - dp[0]) >= edge_disc))
                        continue;
                    const int ymin = y - r > 0 ? y - r : 0, xmin = x - r > 0 ? x - r : 0;
                    const int ymax = y + r < rows - 1 ? y + r : rows - 1, xmax = x + r < cols - 1 ? x + r : cols - 1;
                    float cost[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
                    const unsigned char *ic = img + ((size_t)y * cols + x) * cn;
                    for (int yi = ymin; yi <= ymax; ++yi)
                        for (int xi = xmin; xi <= xmax; ++xi) {
                            const unsigned char *in = img + ((size_t)yi * cols + xi) * cn;
                            int d = iabs(in[0] - ic[0]);   /* DistRgbMax, .cu:54-74 */
                            if (cn == 3) {
                                const int d1 = iabs(in[1] - ic[1]), d2 = iabs(in[2] - ic[2]);
                                if (d1 > d) d = d1;
                                if (d2 > d) d = d2;
                            }
                            const float weight = ctab[d] * stab[iabs(y - yi) * (half + 1) + iabs(x - xi)];
                            const int dr = cur[(size_t)yi * cols + xi];
                            for (int k = 0; k < 5; ++k) {
                                int a = iabs(dr - dp[k]);
                                if (a > max_disc) a = max_disc;
                                cost[k] += a * weight;
                            }
                        }
                    float minimum = 3.402823466e+38f;
                    int id = 0;
                    for (int k = 0; k < 5; ++k)
                        if (cost[k] < minimum) { minimum = cost[k]; id = k; }
                    nxt[(size_t)y * cols + x] = dp[id];
                }
            int *sw = cur; cur = nxt; nxt = sw;
        }
    for (size_t i = 0; i < n; ++i) {
        if (es == 1) ((unsigned char *)disp)[i] = (unsigned char)cur[i];
        else ((short *)disp)[i] = (short)cur[i];
    }
    free(cur); free(nxt); free(stab);
    return 0;
}
