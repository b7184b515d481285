// cv::cuda::StereoBM: handle + C-ABI entry points.  Host-side twin of StereoBMImpl
// (modules/cudastereo/src/stereobm.cpp:67-197): validation, optional prefilter of both images,
// block matching, textureness post-filter -- all stream-ordered on the caller's stream.
#include "stereobm_dev.h"
#include <vector>
#include <algorithm>
#include "mi_selftest.h"

using namespace mi;

struct mi_stereobm {
    mi_stereobm_params P;
    // scratch owned by the handle (stereobm.cpp:126: minSSD_, leBuf_, riBuf_)
    unsigned *minssd = nullptr;
    unsigned char *lebuf = nullptr, *ribuf = nullptr;
    int *tex = nullptr;   // |Sobel| plane of the textureness filter (extended domain)
    int cap_rows = 0, cap_cols = 0, cap_pairs = 0;
    long long step = 0;   // bytes per row of lebuf/ribuf; minssd uses step elements
    // batch: per-pair pointer table (device) and the host copy the asynchronous upload reads
    sbm::BmPair *tab_dev = nullptr;
    int tab_cap = 0;
    std::vector<sbm::BmPair> tab_host;
};

extern "C" {

void mi_stereobm_default_params(mi_stereobm_params *p)
{
    if (!p) return;
    // createStereoBM(64, 19) cudastereo.hpp:90; StereoBMImpl ctor stereobm.cpp:129-132
    p->num_disparities = 64; p->block_size = 19; p->prefilter_type = MI_PREFILTER_NONE; p->prefilter_cap = 31;
    p->prefilter_size = 9; p->texture_threshold = 3.0f; p->uniqueness_ratio = 0; p->emulate_cuda_edge = 1;
}

int mi_stereobm_create(const mi_stereobm_params *p, mi_stereobm **out)
{
    MI_REQUIRE(out, MI_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device available: the miflow product path has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    mi_stereobm *h = new mi_stereobm();
    if (p) h->P = *p; else mi_stereobm_default_params(&h->P);
    *out = h;
    return MI_OK;
}

int mi_stereobm_set_params(mi_stereobm *h, const mi_stereobm_params *p)
{
    MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument");
    h->P = *p;   // like the reference's setters: validated at compute() (stereobm.cpp:143-146)
    return MI_OK;
}

int mi_stereobm_get_params(const mi_stereobm *h, mi_stereobm_params *p)
{
    MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument");
    *p = h->P;
    return MI_OK;
}

void mi_stereobm_destroy(mi_stereobm *h)
{
    if (!h) return;
    if (h->minssd) (void)hipFree(h->minssd);
    if (h->lebuf) (void)hipFree(h->lebuf);
    if (h->ribuf) (void)hipFree(h->ribuf);
    if (h->tex) (void)hipFree(h->tex);
    if (h->tab_dev) (void)hipFree(h->tab_dev);
    delete h;
}

static int check_u8(const mi_mat *m, const char *name)
{
    MI_REQUIRE(m && m->data, MI_ERR_BAD_ARG, "%s: null matrix", name);
    MI_REQUIRE(m->type == MI_8UC1, MI_ERR_BAD_TYPE, "%s: must be CV_8UC1", name);   // stereobm.cpp:151
    MI_REQUIRE(m->rows > 0 && m->cols > 0, MI_ERR_BAD_SIZE, "%s: empty", name);
    MI_REQUIRE(m->step >= (size_t)m->cols, MI_ERR_BAD_ARG, "%s: step < cols", name);
    return MI_OK;
}

static int check_bm_params(int ndisp, int winsz, int rows, int cols)
{
    // CV_Assert( 0 < ndisp_ && ndisp_ <= 256 ); CV_Assert( ndisp_ % 8 == 0 ); CV_Assert( winSize_ % 2 == 1 )  :143-146
    MI_REQUIRE(0 < ndisp && ndisp <= 256, MI_ERR_BAD_ARG, "numDisparities must be in (0,256]");
    MI_REQUIRE(ndisp % 8 == 0, MI_ERR_BAD_ARG, "numDisparities must be a multiple of 8");
    MI_REQUIRE(winsz % 2 == 1, MI_ERR_BAD_ARG, "blockSize must be odd");
    MI_REQUIRE((winsz >> 1) >= 1 && (winsz >> 1) <= 25, MI_ERR_BAD_ARG, "Unsupported window size");   // stereobm.cu:503-504
    // the reference launches an empty grid here (stereobm.cu:469-470), which the CUDA runtime rejects
    MI_REQUIRE(cols - ndisp - 2 * (winsz >> 1) > 0 && rows - 2 * (winsz >> 1) > 0, MI_ERR_BAD_SIZE,
               "image too small for numDisparities + blockSize");
    return MI_OK;
}

static int ensure_scratch(mi_stereobm *h, int rows, int cols, bool need_bufs, int pairs = 1)
{
    if (h->cap_rows < rows || h->cap_cols < cols || h->cap_pairs < pairs) {
        if (h->minssd) (void)hipFree(h->minssd);
        if (h->lebuf) (void)hipFree(h->lebuf);
        if (h->ribuf) (void)hipFree(h->ribuf);
        if (h->tex) (void)hipFree(h->tex);
        h->minssd = nullptr; h->lebuf = h->ribuf = nullptr; h->tex = nullptr;
        h->cap_rows = std::max(h->cap_rows, rows); h->cap_cols = std::max(h->cap_cols, cols); h->cap_pairs = std::max(h->cap_pairs, pairs);
        h->step = align_up(h->cap_cols, 256);
    }
    const size_t per_pair = (size_t)h->step * h->cap_rows;
    if (!h->minssd) MI_HIP_TRY(hipMalloc((void **)&h->minssd, sizeof(unsigned) * per_pair * h->cap_pairs));
    if (need_bufs && !h->lebuf) {
        MI_HIP_TRY(hipMalloc((void **)&h->lebuf, per_pair * h->cap_pairs));
        MI_HIP_TRY(hipMalloc((void **)&h->ribuf, per_pair * h->cap_pairs));
    }
    return MI_OK;
}

int mi_stereobm_compute(mi_stereobm *h, const mi_mat *left, const mi_mat *right, mi_mat *disp, void *stream)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    hipStream_t st = (hipStream_t)stream;
    const mi_stereobm_params &P = h->P;
    int rc;
    if ((rc = check_u8(left, "left")) || (rc = check_u8(right, "right")) || (rc = check_u8(disp, "disparity"))) return rc;
    // CV_Assert( left.size() == right.size() && left.type() == right.type() )  stereobm.cpp:152
    MI_REQUIRE(left->rows == right->rows && left->cols == right->cols, MI_ERR_BAD_SIZE, "left.size() != right.size()");
    MI_REQUIRE(disp->rows == left->rows && disp->cols == left->cols, MI_ERR_BAD_SIZE, "disparity.size() != left.size()");
    const int rows = left->rows, cols = left->cols;
    if ((rc = check_bm_params(P.num_disparities, P.block_size, rows, cols))) return rc;
    const bool pre = P.prefilter_type == MI_PREFILTER_XSOBEL || P.prefilter_type == MI_PREFILTER_NORMALIZED_RESPONSE;
    if ((rc = ensure_scratch(h, rows, cols, pre))) return rc;

    const unsigned char *le = (const unsigned char *)left->data, *ri = (const unsigned char *)right->data;
    long long ls = (long long)left->step, rs = (long long)right->step;
    if (P.prefilter_type == MI_PREFILTER_XSOBEL) {                      // stereobm.cpp:164-173
        if ((rc = sbm::prefilter_xsobel(le, ls, h->lebuf, h->step, rows, cols, P.prefilter_cap, st))) return rc;
        if ((rc = sbm::prefilter_xsobel(ri, rs, h->ribuf, h->step, rows, cols, P.prefilter_cap, st))) return rc;
        le = h->lebuf; ri = h->ribuf; ls = rs = h->step;
    } else if (P.prefilter_type == MI_PREFILTER_NORMALIZED_RESPONSE) {  // :175-185
        if ((rc = sbm::prefilter_norm(le, ls, h->lebuf, h->step, rows, cols, P.prefilter_cap, P.prefilter_size, st))) return rc;
        if ((rc = sbm::prefilter_norm(ri, rs, h->ribuf, h->step, rows, cols, P.prefilter_cap, P.prefilter_size, st))) return rc;
        le = h->lebuf; ri = h->ribuf; ls = rs = h->step;
    }
    // stereoBM_CUDA: memset disp = 0 (stereobm.cu:506); the 0xFF fill of minSSD (:507) is not needed -- the
    // kernel writes every element it later reads
    MI_HIP_TRY(hipMemset2DAsync(disp->data, disp->step, 0, (size_t)cols, (size_t)rows, st));
    // the winners' SSDs are the uniqueness pass's input only: without that test nothing reads them and the kernel does not store them
    // (8.3 of the 11.8 MB a 1080p pair's launch wrote, profiles/r10 StereoBM traffic)
    if ((rc = sbm::block_match(le, ls, ri, rs, (unsigned char *)disp->data, (long long)disp->step, P.uniqueness_ratio > 0 ? h->minssd : nullptr, h->step, rows, cols,
                               P.num_disparities, P.block_size, P.uniqueness_ratio, P.emulate_cuda_edge, st)))
        return rc;
    if (P.texture_threshold > 0 && tuning().sbm_texfuse != 0) {         // stereobm.cpp:189-190, one launch (k_textureness_fused)
        rc = sbm::textureness_fused(le, ls, (unsigned char *)disp->data, (long long)disp->step, nullptr, 1, rows, cols, P.block_size,
                                    P.texture_threshold, st);
    } else if (P.texture_threshold > 0) {
        if (!h->tex) {
            int sld, sh;
            sbm::textureness_scratch_dims(h->cap_rows, h->cap_cols, &sld, &sh);
            MI_HIP_TRY(hipMalloc((void **)&h->tex, sizeof(int) * (size_t)sld * sh));
        }
        rc = sbm::textureness(le, ls, (unsigned char *)disp->data, (long long)disp->step, rows, cols, P.block_size,
                              P.texture_threshold, h->tex, st);
    }
    return rc;
}

// n stereo pairs of one size through one handle: prefilters and the textureness post-filter run pair by pair (small, bandwidth-bound
// kernels sharing the handle's scratch), the block matching -- where the time goes -- as ONE launch with blockIdx.z = pair.  A single
// 1080p pair needs ~16-row bands to put enough waves on the device, and every band spends 2R rows building its first window
// (47 % of the rows at block size 15); the batch supplies the waves, so its bands are up to 48 rows tall (23 %;
// block_match_impl caps the band height at 48 rows for single pairs and batches alike -- 96-row bands lost occupancy to the
// staged rows in LDS: 32 | 48 | 96 rows = 4 990 | 5 070 | 4 575 pairs/s, which also retuned the single-pair path).
int mi_stereobm_compute_batch(mi_stereobm *h, int n, const mi_mat *lefts, const mi_mat *rights, mi_mat *disps, void *stream)
{
    MI_REQUIRE(h, MI_ERR_BAD_ARG, "null handle");
    MI_REQUIRE(n > 0 && lefts && rights && disps, MI_ERR_BAD_ARG, "empty batch");
    if (n == 1) return mi_stereobm_compute(h, lefts, rights, disps, stream);
    hipStream_t st = (hipStream_t)stream;
    const mi_stereobm_params &P = h->P;
    int rc;
    for (int i = 0; i < n; ++i) {
        if ((rc = check_u8(&lefts[i], "left")) || (rc = check_u8(&rights[i], "right")) || (rc = check_u8(&disps[i], "disparity"))) return rc;
        MI_REQUIRE(lefts[i].rows == lefts[0].rows && lefts[i].cols == lefts[0].cols, MI_ERR_BAD_SIZE, "the pairs of a batch must have one size");
        MI_REQUIRE(lefts[i].rows == rights[i].rows && lefts[i].cols == rights[i].cols, MI_ERR_BAD_SIZE, "left.size() != right.size()");
        MI_REQUIRE(disps[i].rows == lefts[i].rows && disps[i].cols == lefts[i].cols, MI_ERR_BAD_SIZE, "disparity.size() != left.size()");
    }
    const int rows = lefts[0].rows, cols = lefts[0].cols;
    if ((rc = check_bm_params(P.num_disparities, P.block_size, rows, cols))) return rc;
    const bool pre = P.prefilter_type == MI_PREFILTER_XSOBEL || P.prefilter_type == MI_PREFILTER_NORMALIZED_RESPONSE;
    if ((rc = ensure_scratch(h, rows, cols, pre, n))) return rc;
    if (h->tab_cap < n) {
        if (h->tab_dev) (void)hipFree(h->tab_dev);
        h->tab_dev = nullptr; h->tab_cap = 0;
        MI_HIP_TRY(hipMalloc((void **)&h->tab_dev, sizeof(sbm::BmPair) * n));
        h->tab_cap = n;
    }
    const long long pp = h->step * h->cap_rows;   // bytes (lebuf / ribuf) = elements (minssd) per pair
    h->tab_host.resize(n);
    for (int i = 0; i < n; ++i) {
        const unsigned char *le = (const unsigned char *)lefts[i].data, *ri = (const unsigned char *)rights[i].data;
        long long ls = (long long)lefts[i].step, rs = (long long)rights[i].step;
        if (pre) {
            unsigned char *lb = h->lebuf + i * pp, *rb = h->ribuf + i * pp;
            if (P.prefilter_type == MI_PREFILTER_XSOBEL) {
                if ((rc = sbm::prefilter_xsobel(le, ls, lb, h->step, rows, cols, P.prefilter_cap, st))) return rc;
                if ((rc = sbm::prefilter_xsobel(ri, rs, rb, h->step, rows, cols, P.prefilter_cap, st))) return rc;
            } else {
                if ((rc = sbm::prefilter_norm(le, ls, lb, h->step, rows, cols, P.prefilter_cap, P.prefilter_size, st))) return rc;
                if ((rc = sbm::prefilter_norm(ri, rs, rb, h->step, rows, cols, P.prefilter_cap, P.prefilter_size, st))) return rc;
            }
            le = lb; ri = rb; ls = rs = h->step;
        }
        h->tab_host[i] = {le, ri, (unsigned char *)disps[i].data, ls, rs, (long long)disps[i].step};
    }
    MI_HIP_TRY(hipMemcpyAsync(h->tab_dev, h->tab_host.data(), sizeof(sbm::BmPair) * n, hipMemcpyHostToDevice, st));
    if ((rc = sbm::zero_disp_batch(h->tab_dev, n, rows, cols, st))) return rc;   // stereobm.cu:506, all pairs in one launch
    if ((rc = sbm::block_match_batch(h->tab_dev, n, P.uniqueness_ratio > 0 ? h->minssd : nullptr, h->step, pp, rows, cols, P.num_disparities, P.block_size, P.uniqueness_ratio,
                                     P.emulate_cuda_edge, st)))
        return rc;
    if (P.texture_threshold > 0 && tuning().sbm_texfuse != 0) {   // the post-filter of all pairs in one launch (the block matcher's table)
        if ((rc = sbm::textureness_fused(nullptr, 0, nullptr, 0, h->tab_dev, n, rows, cols, P.block_size, P.texture_threshold, st))) return rc;
    } else if (P.texture_threshold > 0) {
        if (!h->tex) {
            int sld, sh;
            sbm::textureness_scratch_dims(h->cap_rows, h->cap_cols, &sld, &sh);
            MI_HIP_TRY(hipMalloc((void **)&h->tex, sizeof(int) * (size_t)sld * sh));
        }
        for (int i = 0; i < n; ++i)
            if ((rc = sbm::textureness(h->tab_host[i].left, h->tab_host[i].lstep, (unsigned char *)disps[i].data, (long long)disps[i].step, rows,
                                       cols, P.block_size, P.texture_threshold, h->tex, st)))
                return rc;
    }
    return MI_OK;
}

// ---- stage-level entry points (the reference's device-layer functions, stereobm.cpp:54-63)
int mi_stereobm_prefilter_xsobel(const mi_mat *src, mi_mat *dst, int prefilter_cap, void *stream)
{
    int rc;
    if ((rc = check_u8(src, "input")) || (rc = check_u8(dst, "output"))) return rc;
    MI_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, MI_ERR_BAD_SIZE, "size mismatch");
    return sbm::prefilter_xsobel((const unsigned char *)src->data, (long long)src->step, (unsigned char *)dst->data,
                                 (long long)dst->step, src->rows, src->cols, prefilter_cap, (hipStream_t)stream);
}

int mi_stereobm_prefilter_norm(const mi_mat *src, mi_mat *dst, int prefilter_cap, int winsize, void *stream)
{
    int rc;
    if ((rc = check_u8(src, "input")) || (rc = check_u8(dst, "output"))) return rc;
    MI_REQUIRE(src->rows == dst->rows && src->cols == dst->cols, MI_ERR_BAD_SIZE, "size mismatch");
    return sbm::prefilter_norm((const unsigned char *)src->data, (long long)src->step, (unsigned char *)dst->data,
                               (long long)dst->step, src->rows, src->cols, prefilter_cap, winsize, (hipStream_t)stream);
}

int mi_stereobm_block_match(const mi_mat *left, const mi_mat *right, mi_mat *disp, mi_mat *min_ssd, int ndisp, int winsz,
                            int uniqueness_ratio, int emulate_cuda_edge, void *stream)
{
    int rc;
    if ((rc = check_u8(left, "left")) || (rc = check_u8(right, "right")) || (rc = check_u8(disp, "disparity"))) return rc;
    MI_REQUIRE(left->rows == right->rows && left->cols == right->cols && disp->rows == left->rows && disp->cols == left->cols,
               MI_ERR_BAD_SIZE, "size mismatch");
    MI_REQUIRE(min_ssd && min_ssd->data && min_ssd->type == MI_32SC1 && min_ssd->rows == left->rows && min_ssd->cols == left->cols &&
               min_ssd->step % 4 == 0 && min_ssd->step >= (size_t)left->cols * 4, MI_ERR_BAD_ARG, "min_ssd must be CV_32SC1 of the image size");
    if ((rc = check_bm_params(ndisp, winsz, left->rows, left->cols))) return rc;
    hipStream_t st = (hipStream_t)stream;
    MI_HIP_TRY(hipMemset2DAsync(disp->data, disp->step, 0, (size_t)left->cols, (size_t)left->rows, st));
    MI_HIP_TRY(hipMemset2DAsync(min_ssd->data, min_ssd->step, 0xFF, (size_t)left->cols * 4, (size_t)left->rows, st));  // stereobm.cu:507
    return sbm::block_match((const unsigned char *)left->data, (long long)left->step, (const unsigned char *)right->data,
                            (long long)right->step, (unsigned char *)disp->data, (long long)disp->step, (unsigned *)min_ssd->data,
                            (long long)(min_ssd->step / 4), left->rows, left->cols, ndisp, winsz, uniqueness_ratio,
                            emulate_cuda_edge, st);
}

int mi_stereobm_textureness(const mi_mat *img, mi_mat *disp, int winsz, float avg_texture_threshold, void *stream)
{
    int rc;
    if ((rc = check_u8(img, "input")) || (rc = check_u8(disp, "disparity"))) return rc;
    MI_REQUIRE(img->rows == disp->rows && img->cols == disp->cols, MI_ERR_BAD_SIZE, "size mismatch");
    MI_REQUIRE(winsz % 2 == 1 && winsz / 2 <= 25, MI_ERR_BAD_ARG, "Unsupported window size");
    if (tuning().sbm_texfuse != 0)
        return sbm::textureness_fused((const unsigned char *)img->data, (long long)img->step, (unsigned char *)disp->data, (long long)disp->step,
                                      nullptr, 1, img->rows, img->cols, winsz, avg_texture_threshold, (hipStream_t)stream);
    int sld, sh;
    sbm::textureness_scratch_dims(img->rows, img->cols, &sld, &sh);
    int *S = nullptr;
    DevTmp tmp;
    MI_HIP_TRY(tmp.alloc(&S, (size_t)sld * sh));
    rc = sbm::textureness((const unsigned char *)img->data, (long long)img->step, (unsigned char *)disp->data,
                          (long long)disp->step, img->rows, img->cols, winsz, avg_texture_threshold, S, (hipStream_t)stream);
    if (rc) return rc;
    MI_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));   // S is freed on return
    return MI_OK;
}

int miflow_selftest_tmax16(const unsigned *in_host, unsigned *out_host)
{
    MI_REQUIRE(in_host && out_host, MI_ERR_BAD_ARG, "null argument");
    unsigned *d = nullptr;
    DevTmp tmp;
    MI_HIP_TRY(tmp.alloc(&d, 1024 + 64));
    MI_HIP_TRY(hipMemcpy(d, in_host, sizeof(unsigned) * 1024, hipMemcpyHostToDevice));
    int rc = sbm::dbg_tmax16(d, d + 1024, nullptr);
    if (!rc) { MI_HIP_TRY(hipDeviceSynchronize()); MI_HIP_TRY(hipMemcpy(out_host, d + 1024, sizeof(unsigned) * 64, hipMemcpyDeviceToHost)); }
    return rc;
}

int miflow_selftest_wave_min(const unsigned *in_host, unsigned *out_host)
{
    MI_REQUIRE(in_host && out_host, MI_ERR_BAD_ARG, "null argument");
    unsigned *d = nullptr;
    DevTmp tmp;
    MI_HIP_TRY(tmp.alloc(&d, 192));
    MI_HIP_TRY(hipMemcpy(d, in_host, sizeof(unsigned) * 64, hipMemcpyHostToDevice));
    int rc = sbm::dbg_wave_min(d, d + 64, nullptr);
    if (!rc) { MI_HIP_TRY(hipDeviceSynchronize()); MI_HIP_TRY(hipMemcpy(out_host, d + 64, sizeof(unsigned) * 65, hipMemcpyDeviceToHost)); }
    return rc;
}

}  // extern "C"
