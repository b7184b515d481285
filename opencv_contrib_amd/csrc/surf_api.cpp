// cv::cuda::SURF_CUDA: handle + C-ABI entry points.  Host-side twin of SURF_CUDA_Invoker
// (modules/xfeatures2d/src/surf.cuda.cpp:134-255): validation, integral(s), octave loop, orientation, descriptors.
// One stream, no host round trip inside the octave loop; the only synchronisation is the final read-back of the
// feature count (the reference's keypoints.cols = featureCounter, :205-209).
#include "surf_dev.h"
#include <cstdlib>
#include "mi_selftest.h"
#include <cmath>
#include <vector>

using namespace mi;

struct mi_surf {
    mi_surf_params P;
    // tables (surf.cpp:544-565 generators; see oracle/surf_ref.c for the correspondence with surf.cu:520-522,685-707)
    float *apt = nullptr, *dw = nullptr;
    // scratch sized for (rows, cols, layers)
    int capR = 0, capC = 0, capL = 0, capCand = 0;
    unsigned *sum = nullptr, *msum = nullptr, *V = nullptr, *BT = nullptr, *poly = nullptr;
    float *det = nullptr, *trace = nullptr;
    unsigned long long *bits = nullptr, *sbits = nullptr;
    unsigned *rowcnt = nullptr, *segcnt = nullptr;
    int4 *cand = nullptr;
    void *itmp = nullptr;   // per-candidate interpolation results
    unsigned *counters = nullptr;   // [0] = features, [1 + octave] = candidates of the octave (surf.cuda.cpp:158-159)
    int sld = 0, vld = 0, dld = 0;
    // all octaves per launch (surf::detect_fused): region sizes and the per-(octave, layer) filter geometry on the device
    bool fused = false;
    bool use_poly = false;
    int lds_tiles = 0;   // octave 0 of the fused det / trace launch on LDS tiles: this handle's decision (MIFLOW_SURF_LDS=0 switches it off)
    int capO = 0;
    void *geo = nullptr;
    std::vector<unsigned char> geo_host;   // source of the asynchronous upload: outlives the call
    bool geo_dirty = false;
};

static int calc_size(int octave, int layer) { return (9 + 6 * layer) << octave; }

// The Gaussian kernel behind surf.cu's literal weight tables (c_aptW :522, c_DW :685-707): cv::getGaussianKernel(n, sigma, CV_32F) as
// OpenCV 2.4 rounded it -- exp() in double rounded to float, the float terms summed in double, float * (1 / sum) in double rounded to
// float -- for the CPU class's float sigmas promoted to double.  The outer products below reproduce every literal bit for bit
// (tests/test_ref_pin_cuda.py::test_surf_weight_tables_equal_the_literals_of_surf_cu checks the same generator in the oracle against the
// parsed reference file).
static void gauss_tab(int n, float sigma_f, float *k)
{
    const double sigma = (double)sigma_f, scale2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; k[i] = (float)std::exp(scale2 * x * x); sum += k[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
}

extern "C" {

void mi_surf_default_params(mi_surf_params *p)
{
    if (!p) return;
    // SURF_CUDA::create(thr, nOctaves=4, nOctaveLayers=2, extended=false, keypointsRatio=0.01f, upright=false), cuda.hpp:117-118
    p->hessian_threshold = 100; p->n_octaves = 4; p->n_octave_layers = 2; p->extended = 0; p->keypoints_ratio = 0.01f; p->upright = 0;
}

int mi_surf_create(const mi_surf_params *p, mi_surf **out)
{
    MI_REQUIRE(out, MI_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device available: the miflow product path has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    mi_surf *h = new mi_surf();
    if (p) h->P = *p; else mi_surf_default_params(&h->P);
    // orientation samples: disc of radius 6 (c_aptX / c_aptY, surf.cu:520-521), weights c_aptW (:522) = outer product of the 13-tap
    // kernel for sigma 2.5f; descriptor weights c_DW (:685-707) = outer product of the 20-tap kernel for sigma 3.3f
    float g[13], G[20], apt[3 * 113], dw[400];
    gauss_tab(13, 2.5f, g);
    int k = 0;
    for (int i = -6; i <= 6; i++)
        for (int j = -6; j <= 6; j++)
            if (i * i + j * j <= 36) { apt[k] = (float)i; apt[113 + k] = (float)j; apt[226 + k] = g[i + 6] * g[j + 6]; ++k; }
    gauss_tab(20, 3.3f, G);
    for (int i = 0; i < 20; i++) for (int j = 0; j < 20; j++) dw[i * 20 + j] = G[i] * G[j];
    auto upload = [&]() -> int {
        MI_HIP_TRY(hipMalloc((void **)&h->apt, sizeof(apt)));
        MI_HIP_TRY(hipMalloc((void **)&h->dw, sizeof(dw)));
        MI_HIP_TRY(hipMemcpy(h->apt, apt, sizeof(apt), hipMemcpyHostToDevice));
        MI_HIP_TRY(hipMemcpy(h->dw, dw, sizeof(dw), hipMemcpyHostToDevice));
        MI_HIP_TRY(hipMalloc((void **)&h->counters, sizeof(unsigned) * 64));
        return MI_OK;
    };
    if (const int rc = upload()) { mi_surf_destroy(h); return rc; }
    *out = h;
    return MI_OK;
}

int mi_surf_set_params(mi_surf *h, const mi_surf_params *p) { MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument"); h->P = *p; return MI_OK; }
int mi_surf_get_params(const mi_surf *h, mi_surf_params *p) { MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument"); *p = h->P; return MI_OK; }

static void free_scratch(mi_surf *h)
{
    void *ps[] = {h->poly, h->sum, h->msum, h->V, h->BT, h->det, h->trace, h->bits, h->sbits, h->rowcnt, h->segcnt, h->cand, h->itmp, h->geo};
    for (void *p : ps) if (p) (void)hipFree(p);
    h->sum = h->msum = h->V = h->BT = h->poly = nullptr; h->det = h->trace = nullptr; h->bits = nullptr; h->sbits = nullptr; h->rowcnt = nullptr; h->segcnt = nullptr; h->cand = nullptr; h->itmp = nullptr; h->geo = nullptr;
    h->capR = h->capC = h->capL = h->capCand = h->capO = 0;
}

void mi_surf_release_memory(mi_surf *h) { if (h) free_scratch(h); }   // SURF_CUDA::releaseMemory, surf.cuda.cpp:434-442

void mi_surf_destroy(mi_surf *h)
{
    if (!h) return;
    free_scratch(h);
    if (h->apt) (void)hipFree(h->apt);
    if (h->dw) (void)hipFree(h->dw);
    if (h->counters) (void)hipFree(h->counters);
    delete h;
}

int mi_surf_descriptor_size(const mi_surf *h) { return h && h->P.extended ? 128 : 64; }   // surf.cuda.cpp:277-280

// limits of SURF_CUDA_Invoker's constructor, surf.cuda.cpp:137-156
static int limits(const mi_surf_params &P, int rows, int cols, int *maxFeatures, int *maxCandidates)
{
    MI_REQUIRE(P.n_octaves > 0 && P.n_octave_layers > 0, MI_ERR_BAD_ARG, "nOctaves and nOctaveLayers must be > 0");          // :141
    MI_REQUIRE(P.n_octaves <= 16 && P.n_octave_layers <= 16, MI_ERR_BAD_ARG, "nOctaves / nOctaveLayers too large");
    const int min_size = calc_size(P.n_octaves - 1, 0);
    MI_REQUIRE(rows - min_size >= 0 && cols - min_size >= 0, MI_ERR_BAD_SIZE, "image too small for nOctaves");               // :144-145
    const int lr = rows >> (P.n_octaves - 1), lc = cols >> (P.n_octaves - 1);
    const int min_margin = ((calc_size(P.n_octaves - 1, 2) >> 1) >> (P.n_octaves - 1)) + 1;
    MI_REQUIRE(lr - 2 * min_margin > 0 && lc - 2 * min_margin > 0, MI_ERR_BAD_SIZE, "image too small for nOctaves");         // :150-151
    int mf = (int)((float)(rows * cols) * P.keypoints_ratio);                                                               // :153
    if (mf > 65535) mf = 65535;
    int mc = (int)(1.5 * mf);
    if (mc > 65535) mc = 65535;
    MI_REQUIRE(mf > 0, MI_ERR_BAD_ARG, "maxFeatures <= 0 (keypointsRatio too small)");                                        // :156
    *maxFeatures = mf; *maxCandidates = mc;
    return MI_OK;
}

int mi_surf_max_features(const mi_surf *h, int rows, int cols, int *max_features)
{
    MI_REQUIRE(h && max_features, MI_ERR_BAD_ARG, "null argument");
    int mc;
    return limits(h->P, rows, cols, max_features, &mc);
}

static int ensure(mi_surf *h, int rows, int cols, int layers, int maxCand, bool need_mask, hipStream_t st)
{
    const int octaves = h->P.n_octaves;
    // the plan (fused or not, region sizes, the geometry table) is that of EXACTLY this (size, octaves, layers): a handle whose
    // nOctaveLayers was lowered re-plans instead of running on the larger plan's table
    if (!(h->capR == rows && h->capC == cols && h->capL == layers && h->capCand >= maxCand && h->capO == octaves)) {
        free_scratch(h);
        h->sld = align_up(cols + 1, 64); h->vld = align_up(cols, 64); h->dld = align_up(cols, 64);
        // one launch per stage for all octaves where the kernel arguments hold them (surf::fused_supported), else octave by octave
        // through one set of planes; the fused form keeps every octave's planes: ~4/3 of octave 0's
        h->fused = surf::fused_supported(octaves, layers) && !(MI_EXP_ENV("MIFLOW_SURF_FUSED") && atoi(MI_EXP_ENV("MIFLOW_SURF_FUSED")) == 0);
        {
            static const bool lds_ok = surf::lds_geometry_self_check();   // the LDS path's compile-time geometry against the host's
            const char *e = MI_EXP_ENV("MIFLOW_SURF_LDS");
            h->lds_tiles = (lds_ok && !(e && atoi(e) == 0)) ? 1 : 0;
            // round 5 experiment, kept as a tested opt-in (MIFLOW_SURF_NMS0=1): octave 0's maxima flagged inside the det kernel, no det /
            // trace planes for three quarters of the samples.  Bit-identical, but SLOWER on MI355X (r14j, 4K frame: k_det_nms0 180 us
            // against 103 + 65 us for the tile kernel + the flag launch -- the 18 % of overlapping samples cost more than the planes'
            // 265 MB -- and the refinement's 27 re-evaluations per candidate 143 us against 15): 736 against 826 frames/s
            const char *n0 = getenv("MIFLOW_SURF_NMS0");
            if (h->lds_tiles && n0 && *n0 && atoi(n0) != 0) h->lds_tiles = 2;
            // round 5: octaves >= 1 read their taps from polyphase planes of the integral image (consecutive lanes = consecutive words
            // instead of words 2^o apart); MIFLOW_SURF_POLY=0 keeps the strided gathers (bit-identical either way)
            const char *pp = getenv("MIFLOW_SURF_POLY");
            h->use_poly = h->fused && octaves > 1 && !(pp && *pp && atoi(pp) == 0);
            if (h->use_poly) h->lds_tiles |= 4;
        }
        surf::FusedSizes z;
        z.plane_floats = (size_t)h->dld * rows * (layers + 2);
        z.bits_words = (size_t)layers * rows * div_up(cols, 64);
        z.row_counts = (size_t)layers * rows + 1;
        z.seg_counts = (size_t)layers * rows * surf::nms_segments(cols);
        z.geo_bytes = 0; z.poly_words = 0;
        if (h->fused) surf::fused_sizes(rows, cols, h->dld, octaves, layers, &z);
        const size_t nlists = h->fused ? (size_t)octaves : 1;
        MI_HIP_TRY(hipMalloc((void **)&h->sum, sizeof(unsigned) * (size_t)h->sld * (rows + 1)));
        MI_HIP_TRY(hipMalloc((void **)&h->V, sizeof(unsigned) * (size_t)h->vld * rows));
        MI_HIP_TRY(hipMalloc((void **)&h->BT, sizeof(unsigned) * (size_t)h->vld * surf::integral_bands(rows)));
        MI_HIP_TRY(hipMalloc((void **)&h->det, sizeof(float) * z.plane_floats));
        MI_HIP_TRY(hipMalloc((void **)&h->trace, sizeof(float) * z.plane_floats));
        MI_HIP_TRY(hipMalloc((void **)&h->bits, sizeof(unsigned long long) * z.bits_words));
        MI_HIP_TRY(hipMalloc((void **)&h->sbits, sizeof(unsigned long long) * z.bits_words));
        MI_HIP_TRY(hipMalloc((void **)&h->rowcnt, sizeof(unsigned) * z.row_counts));
        MI_HIP_TRY(hipMalloc((void **)&h->segcnt, sizeof(unsigned) * z.seg_counts));
        MI_HIP_TRY(hipMalloc((void **)&h->cand, sizeof(int4) * (size_t)maxCand * nlists));
        MI_HIP_TRY(hipMalloc(&h->itmp, surf::interp_tmp_bytes(maxCand) * nlists));
        if (h->fused) {
            h->geo_host.assign(z.geo_bytes, 0);
            surf::fused_geometry(h->sld, octaves, layers, h->geo_host.data(), rows, cols, h->use_poly);
            if (h->use_poly) MI_HIP_TRY(hipMalloc((void **)&h->poly, sizeof(unsigned) * z.poly_words));
            MI_HIP_TRY(hipMalloc(&h->geo, z.geo_bytes));
            h->geo_dirty = true;
        }
        h->capR = rows; h->capC = cols; h->capL = layers; h->capCand = maxCand; h->capO = octaves;
    }
    if (need_mask && !h->msum) MI_HIP_TRY(hipMalloc((void **)&h->msum, sizeof(unsigned) * (size_t)h->sld * (rows + 1)));
    if (h->geo_dirty) {   // on the call's stream, not a blocking null-stream copy
        MI_HIP_TRY(hipMemcpyAsync(h->geo, h->geo_host.data(), h->geo_host.size(), hipMemcpyHostToDevice, st));
        h->geo_dirty = false;
    }
    return MI_OK;
}

static int check_img(const mi_mat *m, const char *name)
{
    MI_REQUIRE(m && m->data, MI_ERR_BAD_ARG, "%s: empty", name);
    MI_REQUIRE(m->type == MI_8UC1, MI_ERR_BAD_TYPE, "%s: must be CV_8UC1", name);   // surf.cuda.cpp:139
    MI_REQUIRE(m->rows > 0 && m->cols > 0 && m->step >= (size_t)m->cols, MI_ERR_BAD_ARG, "%s: bad size/step", name);
    return MI_OK;
}
static int check_kp(const mi_mat *kp, int min_cols)
{
    MI_REQUIRE(kp && kp->data && kp->type == MI_32FC1 && kp->rows == 7 && kp->cols >= min_cols && kp->step % 4 == 0 &&
               kp->step >= (size_t)kp->cols * 4, MI_ERR_BAD_ARG, "keypoints must be CV_32FC1 with ROWS_COUNT = 7 rows and enough columns");
    return MI_OK;
}

// Everything of SURF_CUDA_Invoker's detectKeypoints + findOrientation enqueued on `st`; the feature count stays on the device
// (h->counters[0]).  *max_features = the bound the keypoint matrix was checked against.
static int detect_enqueue(mi_surf *h, const mi_mat *img, const mi_mat *mask, mi_mat *keypoints, int *max_features, hipStream_t st)
{
    const mi_surf_params &P = h->P;
    int rc;
    if ((rc = check_img(img, "img"))) return rc;
    const bool use_mask = mask && mask->data;
    if (use_mask) {
        if ((rc = check_img(mask, "mask"))) return rc;
        MI_REQUIRE(mask->rows == img->rows && mask->cols == img->cols, MI_ERR_BAD_SIZE, "mask.size() != img.size()");   // :140
    }
    const int rows = img->rows, cols = img->cols;
    int maxF, maxC;
    if ((rc = limits(P, rows, cols, &maxF, &maxC))) return rc;
    if ((rc = check_kp(keypoints, maxF))) return rc;
    if ((rc = ensure(h, rows, cols, P.n_octave_layers, maxC, use_mask, st))) return rc;
    const int kld = (int)(keypoints->step / 4);
    float *kp = (float *)keypoints->data;

    MI_HIP_TRY(hipMemsetAsync(h->counters, 0, sizeof(unsigned) * 64, st));                                           // :158-159
    if ((rc = surf::integral((const unsigned char *)img->data, (long long)img->step, rows, cols, false, h->V, h->BT, h->vld, h->sum, h->sld, st))) return rc;   // :163
    if (use_mask && (rc = surf::integral((const unsigned char *)mask->data, (long long)mask->step, rows, cols, true, h->V, h->BT, h->vld, h->msum, h->sld, st)))
        return rc;                                                                                                    // :165-169
    MI_HIP_TRY(hipMemset2DAsync(kp, keypoints->step, 0, (size_t)maxF * 4, 7, st));                                   // keypoints.setTo(0) :180
    if (h->fused) {                                                                                                  // :182-204, all octaves per launch
        if ((rc = surf::detect_fused(h->sum, use_mask ? h->msum : nullptr, h->sld, rows, cols, P.n_octaves, P.n_octave_layers,
                                     (float)P.hessian_threshold, h->det, h->trace, h->dld, h->bits, h->rowcnt, h->segcnt, h->cand, maxC,
                                     h->counters + 1, h->itmp, h->geo, kp, kld, maxF, h->counters, h->lds_tiles, st, h->sbits, h->poly))) return rc;
    } else
    for (int octave = 0; octave < P.n_octaves; ++octave) {                                                           // :182-204
        if ((rc = surf::det_trace(h->sum, h->sld, rows, cols, octave, P.n_octave_layers, h->det, h->trace, h->dld, st))) return rc;
        if ((rc = surf::find_maxima(h->det, h->trace, h->dld, use_mask ? h->msum : nullptr, h->sld, rows, cols, octave, P.n_octave_layers,
                                    (float)P.hessian_threshold, h->bits, h->rowcnt, h->segcnt, h->cand, maxC, h->counters + 1 + octave, st))) return rc;
        if ((rc = surf::interpolate(h->det, h->dld, rows, cols, octave, h->cand, h->counters + 1 + octave, maxC, h->itmp, kp, kld, maxF, h->counters, st))) return rc;
    }
    if ((rc = surf::orientation(h->sum, h->sld, rows, cols, kp, kld, h->counters, maxF, P.upright != 0, h->apt, st))) return rc;   // :211-214
    *max_features = maxF;
    return MI_OK;
}

static int read_count(mi_surf *h, int maxF, int *n_features, hipStream_t st)
{
    unsigned nf = 0;
    MI_HIP_TRY(hipMemcpyAsync(&nf, h->counters, sizeof(unsigned), hipMemcpyDeviceToHost, st));                       // :205-207
    MI_HIP_TRY(hipStreamSynchronize(st));
    *n_features = (int)(nf < (unsigned)maxF ? nf : (unsigned)maxF);
    return MI_OK;
}

int mi_surf_detect(mi_surf *h, const mi_mat *img, const mi_mat *mask, mi_mat *keypoints, int *n_features, void *stream)
{
    MI_REQUIRE(h && n_features, MI_ERR_BAD_ARG, "null argument");
    hipStream_t st = (hipStream_t)stream;
    int maxF = 0;
    if (const int rc = detect_enqueue(h, img, mask, keypoints, &maxF, st)) return rc;
    return read_count(h, maxF, n_features, st);
}

// detect + describe with the feature count left on the device between the two (round 5): the reference's operator()(img, mask,
// keypoints, descriptors) reads keypoints.cols back before it launches compute_descriptors (surf.cuda.cpp:205-209, 227-236) -- a host
// round trip in the middle of every frame.  Here the descriptor kernels read the count themselves; the one synchronisation that
// returns it to the caller comes after everything is enqueued.  descriptors: at least max_features rows.
int mi_surf_detect_and_compute(mi_surf *h, const mi_mat *img, const mi_mat *mask, mi_mat *keypoints, mi_mat *descriptors, int *n_features, void *stream)
{
    MI_REQUIRE(h && n_features, MI_ERR_BAD_ARG, "null argument");
    hipStream_t st = (hipStream_t)stream;
    int maxF = 0, rc;
    if ((rc = check_img(img, "img"))) return rc;
    {
        int mc;
        if ((rc = limits(h->P, img->rows, img->cols, &maxF, &mc))) return rc;
    }
    const int dsz = h->P.extended ? 128 : 64;
    MI_REQUIRE(descriptors && descriptors->data && descriptors->type == MI_32FC1 && descriptors->rows >= maxF && descriptors->cols == dsz &&
               descriptors->step % 4 == 0 && descriptors->step >= (size_t)dsz * 4, MI_ERR_BAD_ARG,
               "descriptors must be CV_32FC1 with at least maxFeatures rows and descriptorSize() columns");
    if ((rc = detect_enqueue(h, img, mask, keypoints, &maxF, st))) return rc;
    if ((rc = surf::descriptors((const unsigned char *)img->data, (long long)img->step, img->rows, img->cols, (const float *)keypoints->data,
                                (int)(keypoints->step / 4), maxF, h->P.extended != 0, (float *)descriptors->data,
                                (long long)(descriptors->step / 4), h->dw, st, h->counters))) return rc;
    return read_count(h, maxF, n_features, st);
}

// n frames through one handle (masks may be NULL, or hold NULL-data entries).  A 4K frame fills the device and the feature
// count of every frame is read back like SURF_CUDA does (surf.cuda.cpp:205-207), so this is a loop sharing the handle's scratch.
int mi_surf_detect_batch(mi_surf *h, int n, const mi_mat *imgs, const mi_mat *masks, mi_mat *keypoints, int *n_features, void *stream)
{
    MI_REQUIRE(h && n > 0 && imgs && keypoints && n_features, MI_ERR_BAD_ARG, "empty batch");
    for (int i = 0; i < n; ++i) {
        const int rc = mi_surf_detect(h, &imgs[i], masks ? &masks[i] : nullptr, &keypoints[i], &n_features[i], stream);
        if (rc) return rc;
    }
    return MI_OK;
}

int mi_surf_compute_orientation(mi_surf *h, const mi_mat *img, mi_mat *keypoints, int n_features, void *stream)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = check_img(img, "img")) || (rc = check_kp(keypoints, n_features))) return rc;
    if (n_features <= 0) return MI_OK;
    const int rows = img->rows, cols = img->cols;
    int maxF, maxC;
    if ((rc = limits(h->P, rows, cols, &maxF, &maxC))) return rc;
    if ((rc = ensure(h, rows, cols, h->P.n_octave_layers, maxC, false, st))) return rc;
    if ((rc = surf::integral((const unsigned char *)img->data, (long long)img->step, rows, cols, false, h->V, h->BT, h->vld, h->sum, h->sld, st))) return rc;
    return surf::orientation(h->sum, h->sld, rows, cols, (float *)keypoints->data, (int)(keypoints->step / 4), nullptr, n_features,
                             h->P.upright != 0, h->apt, st);
}

int mi_surf_compute_descriptors(mi_surf *h, const mi_mat *img, const mi_mat *keypoints, int n_features, mi_mat *descriptors, void *stream)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    int rc;
    if ((rc = check_img(img, "img")) || (rc = check_kp(keypoints, n_features))) return rc;
    if (n_features <= 0) return MI_OK;
    const int dsz = h->P.extended ? 128 : 64;
    MI_REQUIRE(descriptors && descriptors->data && descriptors->type == MI_32FC1 && descriptors->rows >= n_features && descriptors->cols == dsz &&
               descriptors->step % 4 == 0, MI_ERR_BAD_ARG, "descriptors must be CV_32FC1, nFeatures x descriptorSize()");   // :232
    return surf::descriptors((const unsigned char *)img->data, (long long)img->step, img->rows, img->cols, (const float *)keypoints->data,
                             (int)(keypoints->step / 4), n_features, h->P.extended != 0, (float *)descriptors->data,
                             (long long)(descriptors->step / 4), h->dw, (hipStream_t)stream);
}

// ---- stage-level entry points (parity tests)
int mi_surf_integral(mi_surf *h, const mi_mat *img, int clamp_to_one, mi_mat *sum, void *stream)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = check_img(img, "img"))) return rc;
    MI_REQUIRE(sum && sum->data && sum->type == MI_32SC1 && sum->rows == img->rows + 1 && sum->cols == img->cols + 1 && sum->step % 4 == 0,
               MI_ERR_BAD_ARG, "sum must be CV_32SC1 (rows+1) x (cols+1)");   // cuda::integral, cudaarithm/src/cuda/integral.cu:62-83
    const int vld = align_up(img->cols, 64);
    unsigned *V = nullptr, *BT = nullptr;
    DevTmp tmp;
    MI_HIP_TRY(tmp.alloc(&V, (size_t)vld * img->rows));
    MI_HIP_TRY(tmp.alloc(&BT, (size_t)vld * surf::integral_bands(img->rows)));
    rc = surf::integral((const unsigned char *)img->data, (long long)img->step, img->rows, img->cols, clamp_to_one != 0, V, BT, vld,
                        (unsigned *)sum->data, (int)(sum->step / 4), st);
    if (rc) return rc;
    MI_HIP_TRY(hipStreamSynchronize(st));   // the temporaries are freed on return: the kernels must be done (and their faults seen)
    return MI_OK;
}

int mi_surf_det_trace(mi_surf *h, const mi_mat *sum, int octave, int n_octave_layers, mi_mat *det, mi_mat *trace, void *stream)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    MI_REQUIRE(sum && sum->data && sum->type == MI_32SC1 && sum->step % 4 == 0 && sum->rows > 1 && sum->cols > 1, MI_ERR_BAD_ARG, "bad sum");
    const int rows = sum->rows - 1, cols = sum->cols - 1, lr = rows >> octave;
    MI_REQUIRE(octave >= 0 && octave < 16 && n_octave_layers > 0 && lr > 0, MI_ERR_BAD_ARG, "bad octave / layers");
    MI_REQUIRE(det && trace && det->data && trace->data && det->type == MI_32FC1 && trace->type == MI_32FC1 &&
               det->rows == (n_octave_layers + 2) * lr && trace->rows == det->rows && det->cols == cols && trace->cols == cols &&
               det->step == trace->step && det->step % 4 == 0, MI_ERR_BAD_ARG, "det/trace must be CV_32FC1 ((layers+2)*layer_rows) x cols, same step");
    return surf::det_trace((const unsigned *)sum->data, (int)(sum->step / 4), rows, cols, octave, n_octave_layers, (float *)det->data,
                           (float *)trace->data, (int)(det->step / 4), (hipStream_t)stream);
}

int miflow_selftest_wave_scan(const unsigned *in_host, unsigned *out_host)
{
    MI_REQUIRE(in_host && out_host, MI_ERR_BAD_ARG, "null argument");
    unsigned *d = nullptr;
    DevTmp tmp;
    MI_HIP_TRY(tmp.alloc(&d, 128));
    MI_HIP_TRY(hipMemcpy(d, in_host, sizeof(unsigned) * 64, hipMemcpyHostToDevice));
    int rc = surf::dbg_scan(d, d + 64, nullptr);
    if (!rc) { MI_HIP_TRY(hipDeviceSynchronize()); MI_HIP_TRY(hipMemcpy(out_host, d + 64, sizeof(unsigned) * 64, hipMemcpyDeviceToHost)); }
    return rc;
}

}  // extern "C"
