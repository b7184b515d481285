/*
 * oracle/refshim/surf_ocl_host.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Launches the reference's OWN SURF kernels -- modules/xfeatures2d/src/opencl/surf.cl, compiled verbatim with
 * -D DOUBLE_SUPPORT and without HAVE_IMAGE2D (oracle/Makefile.ref) -- on the CPU through oclrt.  Only host glue is written
 * here: launch geometry and argument marshalling of SURF_OCL::calcLayerDetAndTrace / findMaximaInLayer /
 * interpolateKeypoint / calcOrientation / computeDescriptors (modules/xfeatures2d/src/surf.ocl.cpp:347-468, :218-266) and
 * the octave loop of SURF_OCL::detectKeypoints (:152-200).  ocl::KernelArg::ReadOnlyNoSize(m) = (pointer, step in bytes,
 * offset in bytes), PtrReadWrite(m) = pointer.  Work-groups run one after the other in row-major order, so the atomic_inc
 * appends (candidates, features) come out in a deterministic order; callers compare them as sets.
 */
#include "oclrt.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int x, y, z, w; } int4_;

extern void SURF_calcLayerDetAndTrace(const unsigned *sumTex, int sum_step, int sum_offset, int img_rows, int img_cols, int c_nOctaveLayers,
                                      int c_octave, int c_layer_rows, float *det, int det_step, int det_offset, float *trace, int trace_step,
                                      int trace_offset);
extern void SURF_findMaximaInLayer(float *det, int det_step, int det_offset, float *trace, int trace_step, int trace_offset,
                                   int4_ *maxPosBuffer, volatile int *maxCounter, int counter_offset, int img_rows, int img_cols,
                                   int c_nOctaveLayers, int c_octave, int c_layer_rows, int c_layer_cols, int c_max_candidates,
                                   float c_hessianThreshold);
extern void SURF_interpolateKeypoint(const float *det, int det_step, int det_offset, const int4_ *maxPosBuffer, float *keypoints,
                                     int keypoints_step, int keypoints_offset, volatile int *featureCounter, int img_rows, int img_cols,
                                     int c_octave, int c_layer_rows, int c_max_features);
extern void SURF_calcOrientation(const unsigned *sumTex, int sum_step, int sum_offset, int img_rows, int img_cols, float *keypoints,
                                 int keypoints_step, int keypoints_offset);
extern void SURF_setUpRight(float *keypoints, int keypoints_step, int keypoints_offset, int rows, int cols);
extern void SURF_computeDescriptors64(const unsigned char *imgTex, int img_step, int img_offset, int img_rows, int img_cols,
                                      const float *keypoints, int keypoints_step, int keypoints_offset, float *descriptors,
                                      int descriptors_step, int descriptors_offset);
extern void SURF_computeDescriptors128(const unsigned char *imgTex, int img_step, int img_offset, int img_rows, int img_cols,
                                       const float *keypoints, int keypoints_step, int keypoints_offset, float *descriptors,
                                       int descriptors_step, int descriptors_offset);
extern void SURF_normalizeDescriptors64(float *descriptors, int descriptors_step, int descriptors_offset);
extern void SURF_normalizeDescriptors128(float *descriptors, int descriptors_step, int descriptors_offset);

static inline int calc_size(int octave, int layer) { return (9 + 6 * layer) << octave; }   /* surf.ocl.cpp:62-75 */
static inline int div_up(int a, int b) { return (a + b - 1) / b; }

typedef struct {
    const unsigned *sum; const unsigned char *img; int rows, cols, layers, octave, layer_rows, layer_cols;
    float *det, *trace; int4_ *maxpos; int *counters; int counter_offset, max_candidates, max_features; float thr;
    float *kp; int kp_pitch; float *desc; int dsize;
} surf_args;

static void det_body(void *p)
{
    surf_args *a = (surf_args *)p;
    SURF_calcLayerDetAndTrace(a->sum, (a->cols + 1) * 4, 0, a->rows, a->cols, a->layers, a->octave, a->layer_rows, a->det, a->cols * 4, 0,
                              a->trace, a->cols * 4, 0);
}
/* SURF_OCL::calcLayerDetAndTrace, surf.ocl.cpp:347-381.  det/trace: ((layers + 2) * (rows >> octave)) x cols planes. */
void ref_ocl_surf_det_trace(const unsigned *sum, int rows, int cols, int octave, int layers, float *det, float *trace)
{
    surf_args a;
    memset(&a, 0, sizeof(a));
    a.sum = sum; a.rows = rows; a.cols = cols; a.layers = layers; a.octave = octave; a.layer_rows = rows >> octave; a.det = det; a.trace = trace;
    const int min_size = calc_size(octave, 0);
    const int max_samples_i = 1 + ((rows - min_size) >> octave), max_samples_j = 1 + ((cols - min_size) >> octave);
    const size_t l[2] = {16, 16};
    const size_t g[2] = {(size_t)div_up(max_samples_j, 16) * 16, (size_t)div_up(max_samples_i, 16) * 16 * (layers + 2)};
    oclrt_run(2, g, l, 0, 0, det_body, &a);
}

static void max_body(void *p)
{
    surf_args *a = (surf_args *)p;
    SURF_findMaximaInLayer(a->det, a->cols * 4, 0, a->trace, a->cols * 4, 0, a->maxpos, a->counters, a->counter_offset, a->rows, a->cols,
                           a->layers, a->octave, a->layer_rows, a->layer_cols, a->max_candidates, a->thr);
}
/* SURF_OCL::findMaximaInLayer, surf.ocl.cpp:383-407.  Returns the counter (may exceed max_candidates); cand: int4 each. */
int ref_ocl_surf_find_maxima(float *det, float *trace, int rows, int cols, int octave, int layers, float thr, int max_candidates, int *cand)
{
    surf_args a;
    memset(&a, 0, sizeof(a));
    int counter = 0;
    a.rows = rows; a.cols = cols; a.layers = layers; a.octave = octave; a.layer_rows = rows >> octave; a.layer_cols = cols >> octave;
    a.det = det; a.trace = trace; a.maxpos = (int4_ *)cand; a.counters = &counter; a.counter_offset = 0; a.max_candidates = max_candidates;
    a.thr = thr;
    const int min_margin = ((calc_size(octave, 2) >> 1) >> octave) + 1;
    if (a.layer_cols - 2 * min_margin <= 0 || a.layer_rows - 2 * min_margin <= 0) return 0;
    const size_t l[2] = {16, 16};
    const size_t g[2] = {(size_t)div_up(a.layer_cols - 2 * min_margin, 14) * 16, (size_t)div_up(a.layer_rows - 2 * min_margin, 14) * layers * 16};
    oclrt_run(2, g, l, 1, 0, max_body, &a);
    return counter;
}

static void interp_body(void *p)
{
    surf_args *a = (surf_args *)p;
    SURF_interpolateKeypoint(a->det, a->cols * 4, 0, a->maxpos, a->kp, a->kp_pitch * 4, 0, a->counters, a->rows, a->cols, a->octave,
                             a->layer_rows, a->max_features);
}
/* SURF_OCL::interpolateKeypoint, surf.ocl.cpp:409-424.  kp: 7 rows x kp_pitch floats; *feature_counter is advanced. */
void ref_ocl_surf_interpolate(const float *det, int rows, int cols, int octave, const int *cand, int ncand, float *kp, int kp_pitch,
                              int max_features, int *feature_counter)
{
    surf_args a;
    memset(&a, 0, sizeof(a));
    a.rows = rows; a.cols = cols; a.octave = octave; a.layer_rows = rows >> octave; a.det = (float *)det; a.maxpos = (int4_ *)cand;
    a.kp = kp; a.kp_pitch = kp_pitch; a.counters = feature_counter; a.max_features = max_features;
    if (ncand <= 0) return;
    const size_t l[3] = {3, 3, 3};
    const size_t g[3] = {(size_t)ncand * 3, 3, 3};
    oclrt_run(3, g, l, 1, 0, interp_body, &a);
}

static void ori_body(void *p)
{
    surf_args *a = (surf_args *)p;
    SURF_calcOrientation(a->sum, (a->cols + 1) * 4, 0, a->rows, a->cols, a->kp, a->kp_pitch * 4, 0);
}
/* SURF_OCL::calcOrientation, surf.ocl.cpp:426-445 (ORI_LOCAL_SIZE = 72 work-items per feature) */
void ref_ocl_surf_orientation(const unsigned *sum, int rows, int cols, float *kp, int kp_pitch, int nfeatures)
{
    surf_args a;
    memset(&a, 0, sizeof(a));
    a.sum = sum; a.rows = rows; a.cols = cols; a.kp = kp; a.kp_pitch = kp_pitch;
    if (nfeatures <= 0) return;
    const size_t l[2] = {72, 1};
    const size_t g[2] = {(size_t)nfeatures * 72, 1};
    oclrt_run(2, g, l, 1, 0, ori_body, &a);
}

static void desc_body(void *p)
{
    surf_args *a = (surf_args *)p;
    if (a->dsize == 64)
        SURF_computeDescriptors64(a->img, a->cols, 0, a->rows, a->cols, a->kp, a->kp_pitch * 4, 0, a->desc, a->dsize * 4, 0);
    else
        SURF_computeDescriptors128(a->img, a->cols, 0, a->rows, a->cols, a->kp, a->kp_pitch * 4, 0, a->desc, a->dsize * 4, 0);
}
static void norm_body(void *p)
{
    surf_args *a = (surf_args *)p;
    if (a->dsize == 64) SURF_normalizeDescriptors64(a->desc, a->dsize * 4, 0);
    else SURF_normalizeDescriptors128(a->desc, a->dsize * 4, 0);
}
/* SURF_OCL::computeDescriptors, surf.ocl.cpp:218-266.  desc: nfeatures x dsize floats (dsize 64 or 128). */
void ref_ocl_surf_descriptors(const unsigned char *img, int rows, int cols, const float *kp, int kp_pitch, int nfeatures, int dsize,
                              float *desc)
{
    surf_args a;
    memset(&a, 0, sizeof(a));
    a.img = img; a.rows = rows; a.cols = cols; a.kp = (float *)kp; a.kp_pitch = kp_pitch; a.desc = desc; a.dsize = dsize;
    if (nfeatures <= 0) return;
    const size_t l[2] = {6, 6};
    const size_t g[2] = {(size_t)nfeatures * 6, 16 * 6};
    oclrt_run(2, g, l, 1, 0, desc_body, &a);
    const size_t ln[2] = {(size_t)dsize, 1};
    const size_t gn[2] = {(size_t)nfeatures * dsize, 1};
    oclrt_run(2, gn, ln, 1, 0, norm_body, &a);
}

/* SURF_OCL::detectKeypoints, surf.ocl.cpp:152-200 (without the orientation step): octave loop with the per-octave candidate
 * counter read back between the kernels.  maxFeatures / maxCandidates as SURF_OCL::setImage computes them (:117-118), with the
 * caller's keypoints ratio (the class hard-codes 0.01f).  Returns the feature count. */
int ref_ocl_surf_detect(const unsigned *sum, int rows, int cols, int n_octaves, int layers, float thr, float keypoints_ratio, float *kp,
                        int kp_pitch)
{
    int max_features = (int)((float)(cols * rows) * keypoints_ratio);
    if (max_features > 65535) max_features = 65535;
    if (max_features > kp_pitch) max_features = kp_pitch;
    int max_candidates = (int)(1.5 * max_features);
    if (max_candidates > 65535) max_candidates = 65535;
    float *det = (float *)calloc((size_t)rows * (layers + 2) * cols, sizeof(float));
    float *trace = (float *)calloc((size_t)rows * (layers + 2) * cols, sizeof(float));
    int *cand = (int *)calloc((size_t)max_candidates * 4 + 4, sizeof(int));
    int features = 0;
    for (int octave = 0; octave < n_octaves; ++octave) {
        if ((rows >> octave) < 1 || (cols >> octave) < 1) break;
        ref_ocl_surf_det_trace(sum, rows, cols, octave, layers, det, trace);
        int n = ref_ocl_surf_find_maxima(det, trace, rows, cols, octave, layers, thr, max_candidates, cand);
        if (n > max_candidates) n = max_candidates;
        if (n > 0) ref_ocl_surf_interpolate(det, rows, cols, octave, cand, n, kp, kp_pitch, max_features, &features);
    }
    free(det); free(trace); free(cand);
    return features < max_features ? features : max_features;
}
