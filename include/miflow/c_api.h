/*
 * miflow C-ABI -- the drop-in boundary of the MI355X-native dense-flow / stereo / SURF
 * hot path (libmiflow.so).  Plain C: pointers, sizes, PODs; no C++/torch/OpenCV types.
 *
 * Each entry point replaces an internal (C++-linkage, PtrStepSz-by-value) device-layer
 * function of the reference; the `Replaces:` line cites it (paths relative to the
 * opencv_contrib tree).  The C++ shim in include/opencv2/ binds these under the
 * reference's own cv::cuda class names; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - all image pointers are DEVICE pointers (HIP), pitched row-major, `step` in bytes
 *     (GpuMat layout: element (y,x) at data + y*step + x*elemSize); step need not equal
 *     cols*elemSize and data need only be element-aligned.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work is
 *     stream-ordered; no entry point synchronises the device unless documented.
 *   - every function returns mi_status (0 = ok, <0 = error) and never throws;
 *     mi_last_error() returns a thread-local message for the last failure.
 *   - handles are not re-entrant (they own scratch memory); distinct handles are fully
 *     independent (no global/__constant__ state) and may be used concurrently from
 *     different host threads / streams with bit-identical results.
 */
#ifndef MIFLOW_C_API_H
#define MIFLOW_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MI_API __attribute__((visibility("default")))
#else
#define MI_API
#endif

typedef enum mi_status {
    MI_OK = 0,
    MI_ERR_BAD_ARG = -1,   /* cv::Error::StsBadArg / failed CV_Assert on a parameter */
    MI_ERR_BAD_TYPE = -2,  /* unsupported matrix type (CV_Assert on type()) */
    MI_ERR_BAD_SIZE = -3,  /* size mismatch (CV_Assert on size()) */
    MI_ERR_HIP = -4,       /* cv::Error::GpuApiCallError */
    MI_ERR_OOM = -5,
    MI_ERR_NOT_IMPL = -6,
    MI_ERR_NO_DEVICE = -7
} mi_status;

/* OpenCV type codes (CV_MAKETYPE(depth, cn)), so GpuMat::type() passes straight through. */
enum { MI_8UC1 = 0, MI_32SC1 = 4, MI_32FC1 = 5, MI_32SC2 = 12 /* knn match indices */, MI_32FC2 = 13, MI_32SC4 = 28,
       /* accepted only by mi_superres_to_gray8 (OpenCV codes CV_8UC3/4, CV_16UC1/3/4, CV_32FC3/4) */
       MI_16UC1 = 2, MI_16SC1 = 3 /* disparity maps of mi_disp_bilateral_apply */, MI_8UC3 = 16, MI_16UC3 = 18, MI_32FC3 = 21, MI_8UC4 = 24, MI_16UC4 = 26, MI_32FC4 = 29 };

/* Device-side matrix view == cv::cuda::PtrStepSz<T> {data, step, cols, rows} + type.
 * Replaces: opencv2/core/cuda_types.hpp PtrStepSz (main repo); in-tree twin
 * modules/cudev/include/opencv2/cudev/ptr2d/glob.hpp:62-89. */
typedef struct mi_mat {
    void *data;
    size_t step; /* bytes */
    int rows, cols;
    int type;
} mi_mat;

MI_API const char *mi_last_error(void);
MI_API const char *mi_version(void);
MI_API int mi_device_count(void);          /* cv::cuda::getCudaEnabledDeviceCount */
MI_API int mi_set_device(int device);      /* cv::cuda::setDevice (cudaoptflow/test/test_optflow.cpp:62) */
/* Destroyed handles leave their large scratch blocks in a small process-wide cache (re-used by the next handle of a similar
 * size: create / destroy cycles do not go through the driver).  This returns the cached blocks to the driver -- the counterpart of
 * cv::cuda::BufferPool's release at shutdown. */
MI_API int mi_release_cached_memory(void);
MI_API int mi_get_device(int *device);

/* Device memory helpers for hosts without a HIP runtime binding (the C++ shim's GpuMat
 * allocator uses them).  mi_malloc_pitch rounds the step up to 256 B. */
MI_API int mi_malloc(void **dptr, size_t bytes);
MI_API int mi_malloc_pitch(void **dptr, size_t *step, size_t width_bytes, int rows);
MI_API int mi_free(void *dptr);
MI_API int mi_memcpy_h2d(void *dst, size_t dstep, const void *src, size_t sstep, size_t width_bytes, int rows, void *stream);
MI_API int mi_memcpy_d2h(void *dst, size_t dstep, const void *src, size_t sstep, size_t width_bytes, int rows, void *stream);
MI_API int mi_memset(void *dst, size_t dstep, int value, size_t width_bytes, int rows, void *stream);
MI_API int mi_stream_create(void **stream);
MI_API int mi_stream_destroy(void *stream);
MI_API int mi_stream_synchronize(void *stream);

/* ===================================================================== Dual TV-L1 ===== */

enum { MI_SEM_CPU_REF = 0,     /* arithmetic of cv::optflow::DualTVL1OpticalFlow (optflow/src/tvl1flow.cpp) */
       MI_SEM_CUDA_COMPAT = 1  /* arithmetic of cv::cuda::OpticalFlowDual_TVL1 (cudaoptflow/src/cuda/tvl1flow.cu) */ };

/* Parameters of cv::cuda::OpticalFlowDual_TVL1::create (cudaoptflow.hpp:375-385) plus the
 * three CPU-class-only knobs (optflow.hpp:283-295).  `iterations` is the cv::cuda name for
 * the outer count; the inner loop runs inner_iterations (cv::cuda equivalent: 1) times per
 * outer iteration, with a median filter of u before each outer iteration when
 * median_filtering > 1 (cv::cuda equivalent: 1 = off). */
typedef struct mi_tvl1_params {
    double tau, lambda, theta, epsilon, scale_step, gamma;
    int nscales, warps, iterations;
    int use_initial_flow;
    int inner_iterations;
    int median_filtering;
    int semantics;       /* MI_SEM_CPU_REF (default) = the arithmetic of cv::optflow::DualTVL1OpticalFlow, the acceptance reference
                          * ("EPE vs CPU ref"): cv::remap warp, cv::resize pyramid, convergence test every iteration;
                          * MI_SEM_CUDA_COMPAT = the arithmetic of cv::cuda's own kernels (normalised a = -0.5 bicubic, clamp
                          * addressing, cuda::resize, sparse check schedule), for callers validated against cv::cuda: the two
                          * differ by ~0.1 px mean EPE, mostly at image borders (see mi_tvl1_default_params) */
    int exact_math;      /* 0 (default): fast device math (v_rcp / v_sqrt / fma), held to the oracle with a stated tolerance;
                          * 1: IEEE divide + f64 hypot, separately rounded operations in the reference's order (with fixed work, epsilon = 0,
                          * still fused in blocks of up to 5 iterations per pass: bit-identical to one launch per iteration) */
    int time_block;      /* inner iterations fused per HBM pass (0 = auto, 1 = one iteration per launch) */
    int lanes;           /* concurrent sub-batches of mi_tvl1_calc_batch (the first on the caller's stream, the others on one internal
                          * stream each): 0 = automatic (2 from 4 pairs on), 1..4 */
    int stop_slack;      /* 0 (default): a warp's inner loop stops exactly where the reference's convergence test stops it.
                          * s > 0 (fast math, epsilon > 0 only): a fused block of iterations is also kept when the test first
                          * passed up to s iterations before the block's end, i.e. up to s iterations MORE than the reference
                          * may run (each of them below the convergence threshold); saves the pass that re-runs the exact count */
    int host_feedback;   /* convergence-checked fast path (epsilon > 0): the launches of a warp's inner loop are enqueued before the
                          * device knows where the loop stops (most of the ~34 per warp then end at once, ~3 us each).
                          * 0 (default) = automatic: a call of at most 2 pairs -- the reference's own calling pattern, one pair per
                          * calc() -- reads the device's "converged" flags back between launches and stops enqueuing for a warp as
                          * soon as every pair has stopped: the host waits inside calc() like the reference's own class does at each
                          * of its convergence checks (cudaoptflow/src/tvl1flow.cpp:362-368), about once per warp; larger batches stay
                          * fully stream-ordered.  -1 = never (no host wait inside calc()); 1 = for every single-lane call.  Results
                          * do not depend on it.  The wait also covers whatever the caller had queued earlier on that stream, and a
                          * host thread driving several handles is serialised by it (use -1 there).  While `stream` is being captured
                          * into a graph (hipStreamBeginCapture) the read-back is switched off whatever this field says: the calc is
                          * then enqueued fully stream-ordered and the capture stays valid (handle warm, i.e. sized by an earlier
                          * calc: allocations are not capturable).
                          * Round 4: the flags are words the deciding launch stores into pinned host memory when it starts and the
                          * host polls them (no copy, no event); a handle also remembers, on the device, how many iterations each
                          * warp of each pair slot needed in its previous calc and sizes the next calc's first blocks by that --
                          * consecutive calcs of a video stream run one pass per warp.  Neither changes a flow or a count. */
} mi_tvl1_params;

typedef struct mi_tvl1 mi_tvl1;

MI_API void mi_tvl1_default_params(mi_tvl1_params *p);
/* Replaces: cv::cuda::OpticalFlowDual_TVL1::create, cudaoptflow/src/tvl1flow.cpp:385-391 */
MI_API int mi_tvl1_create(const mi_tvl1_params *p, mi_tvl1 **out);
MI_API int mi_tvl1_set_params(mi_tvl1 *h, const mi_tvl1_params *p);
MI_API int mi_tvl1_get_params(const mi_tvl1 *h, mi_tvl1_params *p);
/* Replaces: OpticalFlowDual_TVL1_Impl::calc + calcImpl + procOneScale,
 * cudaoptflow/src/tvl1flow.cpp:170-382 (CPU twin optflow/src/tvl1flow.cpp:402-533,1313-1408).
 * I0,I1: MI_8UC1 or MI_32FC1 (floats in [0,1], scaled x255), same size/type.
 * flow: MI_32FC2, same size; read as the initial flow when use_initial_flow.
 * Stream-ordered: the convergence test runs on the device; see mi_tvl1_params.host_feedback for the one case in which calc()
 * waits for the device (a convergence-checked call of one or two pairs). */
MI_API int mi_tvl1_calc(mi_tvl1 *h, const mi_mat *I0, const mi_mat *I1, mi_mat *flow, void *stream);
/* n independent pairs of identical size/type in one pass (blockIdx.z = pair). */
MI_API int mi_tvl1_calc_batch(mi_tvl1 *h, int n, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows, void *stream);
/* Executed inner iterations per (scale, warp) of pair `pair` of the last calc
 * (synchronises `stream`).  iters: [nscales_used][warps] row-major, capacity `cap` ints. */
MI_API int mi_tvl1_last_iterations(mi_tvl1 *h, int pair, int *nscales_used, int *iters, int cap, void *stream);
/* Kernel-level timing of the dominant kernel (the fused iteration) with HIP events recorded on the
 * calc stream around each warp's run of iteration launches.  get_profile synchronises the events and
 * returns, for the last calc: total ms inside those regions, the number of iteration launches inside
 * them, and their algorithmic bytes (64 B x level pixels x batch per launch). */
MI_API int mi_tvl1_set_profiling(mi_tvl1 *h, int enable);
/* Introspection of the launch plan (for roofline accounting; no reference counterpart): what a fast-math calc launches for a
 * pyramid level of `width` x `height` with `pairs_per_lane` pairs and `iterations_per_launch` fused iterations -- *kernel 0: the
 * streaming temporally blocked kernel, *rows_per_band = its band height (every band also streams 2 x iterations halo rows);
 * *kernel 1: the register-tile kernel of the small levels, *rows_per_band = the rows a tile owns.  Needs a HIP device. */
MI_API int mi_tvl1_query_plan(int width, int height, int pairs_per_lane, int iterations_per_launch, int *kernel, int *rows_per_band);
MI_API int mi_tvl1_get_profile(mi_tvl1 *h, double *ms_total, long long *launches, double *algo_bytes);
/* The same for kind 0 (iteration launches, as above) or kind 1 (the warp launches; algorithmic bytes 44 B x level pixels x batch). */
MI_API int mi_tvl1_get_profile_kind(mi_tvl1 *h, int kind, double *ms_total, long long *launches, double *algo_bytes);
/* ... restricted to pyramid level `level` (0 = finest; -1 = all levels): where the kernel time of a calc goes, level by level. */
MI_API int mi_tvl1_get_profile_level(mi_tvl1 *h, int kind, int level, double *ms_total, long long *launches, double *algo_bytes);
MI_API void mi_tvl1_destroy(mi_tvl1 *h);

/* Batched-frames mode over the GPUs of one node (BASELINE configs[4]): independent pairs are cut into contiguous shards,
 * one per device, one host thread + handle + stream pair per device (the reference's multi-device idiom is cv::cuda::setDevice per
 * thread, cudaoptflow/test/test_optflow.cpp:62).  device_ids[0] is the ROOT device: the caller's I0 / I1 / flow matrices live
 * there; shards of the other devices travel over xGMI, double buffered in chunks, overlapping the compute -- as RCCL point-to-point
 * messages (grouped ncclSend / ncclRecv on a two-rank communicator per worker, librccl bound at run time: the scatter / gather of
 * north_star; no reduction, the pairs are independent) or, where RCCL is absent, switched off (MIFLOW_MULTI_RCCL=0), refuses the
 * pair (the same GPU listed twice) or the matrix is pitched, as 2-D peer copies.  device_ids == NULL: devices 0 .. n_devices - 1;
 * n_devices <= 0: all.  The same id may be listed more than once (several workers on one GPU).  calc_batch returns when every flow is
 * in the caller's matrices; results are bit-identical to mi_tvl1_calc_batch. */
typedef struct mi_tvl1_multi mi_tvl1_multi;
MI_API int mi_tvl1_multi_create(const mi_tvl1_params *p, int n_devices, const int *device_ids, mi_tvl1_multi **out);
MI_API int mi_tvl1_multi_device_count(const mi_tvl1_multi *m);
MI_API int mi_tvl1_multi_set_chunk(mi_tvl1_multi *m, int pairs_per_chunk);   /* pairs staged per copy / compute step (default 16) */
MI_API int mi_tvl1_multi_calc_batch(mi_tvl1_multi *m, int n, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows);
/* how the non-root workers are connected to the root: *rccl_links over RCCL, *peer_copy_links by peer copies (either may be NULL) */
MI_API int mi_tvl1_multi_transport(const mi_tvl1_multi *m, int *rccl_links, int *peer_copy_links);
/* why the most recent link of this process fell back to peer copies (the RCCL error text, "librccl.so not found", "MIFLOW_MULTI_RCCL=0");
 * "" if none did.  Valid until the calling thread's next call. */
MI_API const char *mi_tvl1_multi_transport_why(void);
MI_API void mi_tvl1_multi_destroy(mi_tvl1_multi *m);

/* Stage-level entry points (dense or pitched MI_32FC1 planes) == the reference's internal
 * device-layer boundary, exported for plane-by-plane parity tests.
 * Replaces: tvl1flow::centeredGradient  cudaoptflow/src/cuda/tvl1flow.cu:59-81 */
MI_API int mi_tvl1_centered_gradient(const mi_mat *src, mi_mat *dx, mi_mat *dy, void *stream);
/* OR-ed into `semantics` of mi_tvl1_warp_backward (test hook): the fast-math form of the kernel calc() runs when
 * exact_math = 0 -- the bicubic sums of I1, I1x, I1y formed separably; differs from the reference order by rounding only. */
#define MI_WARP_STAGE_FAST 0x100
/* likewise: the LDS-staged / the global-gather formulation of that kernel (default: the library's tuned choice) */
#define MI_WARP_STAGE_LDS 0x200
#define MI_WARP_STAGE_GATHER 0x400
/* Replaces: tvl1flow::warpBackward  tvl1flow.cu:106-179  (CPU: 3x remap + calcGradRho,
 * optflow/src/tvl1flow.cpp:1371-1376) */
MI_API int mi_tvl1_warp_backward(int semantics, const mi_mat *I0, const mi_mat *I1, const mi_mat *I1x,
                                 const mi_mat *I1y, const mi_mat *u1, const mi_mat *u2, mi_mat *I1w,
                                 mi_mat *I1wx, mi_mat *I1wy, mi_mat *grad, mi_mat *rho);
/* Replaces: tvl1flow::estimateU + estimateDualVariables  tvl1flow.cu:209-363, fused into
 * one pass.  `niter` iterations from (u,p) in -> out (separate buffers); err_host[niter]
 * (HOST pointer, may be NULL) receives the per-iteration sum of (du1^2 + du2^2); when it is
 * non-NULL the call synchronises `stream` before returning (test hook).  time_block > 0: the
 * streaming temporally blocked kernel with blocks of at most that many iterations; time_block < 0: the
 * register-tile kernel of the small pyramid levels, variant -time_block - 1 (both fast math). */
MI_API int mi_tvl1_iterate(int exact_math, int time_block, int niter, const mi_mat *I1wx, const mi_mat *I1wy,
                           const mi_mat *grad, const mi_mat *rho_c, const mi_mat *u_in /*[2]*/,
                           const mi_mat *p_in /*[4]*/, mi_mat *u_out /*[2]*/, mi_mat *p_out /*[4]*/,
                           float l_t, float theta, float taut, double *err_host, void *stream);
/* Replaces: cv::cuda::resize(INTER_LINEAR) on CV_32FC1, cudawarping/src/resize.cpp:57-108
 * (semantics CUDA_COMPAT) / cv::resize (semantics CPU_REF); inv_scale = fx given by the
 * caller when `explicit_dsize` is 0, else dst/src.  dst *= post_scale afterwards. */
MI_API int mi_resize_linear(int semantics, const mi_mat *src, mi_mat *dst, double fx, double fy,
                            int explicit_dsize, float post_scale, void *stream);

/* ======================================================================= StereoBM ===== */

/* cv::StereoBM::PREFILTER_* (main repo calib3d.hpp); -1 = no prefilter (the cv::cuda default preset_,
 * cudastereo/src/stereobm.cpp:130) */
enum { MI_PREFILTER_NONE = -1, MI_PREFILTER_NORMALIZED_RESPONSE = 0, MI_PREFILTER_XSOBEL = 1 };

/* State of StereoBMImpl (cudastereo/src/stereobm.cpp:117-126) + one non-reference knob. */
typedef struct mi_stereobm_params {
    int num_disparities;      /* ndisp_: (0,256], multiple of 8 */
    int block_size;           /* winSize_: odd, 3..51 */
    int prefilter_type;       /* preset_: MI_PREFILTER_* */
    int prefilter_cap;        /* preFilterCap_ (31) */
    int prefilter_size;       /* preFilterSize_ (9), NORMALIZED_RESPONSE window */
    float texture_threshold;  /* avergeTexThreshold_ (3); <= 0 disables the post-filter */
    int uniqueness_ratio;     /* uniquenessRatio_ (0 = off) */
    int emulate_cuda_edge;    /* 1 (default): bit-identical to the CUDA kernel, including the truncated right
                                 half-window its 128-wide block mapping produces for columns X in [cols-2R, cols-R)
                                 (stereobm.cu:77-89); 0: full window everywhere */
} mi_stereobm_params;

typedef struct mi_stereobm mi_stereobm;

MI_API void mi_stereobm_default_params(mi_stereobm_params *p);
/* Replaces: cv::cuda::createStereoBM, cudastereo/src/stereobm.cpp:194-197 */
MI_API int mi_stereobm_create(const mi_stereobm_params *p, mi_stereobm **out);
MI_API int mi_stereobm_set_params(mi_stereobm *h, const mi_stereobm_params *p);
MI_API int mi_stereobm_get_params(const mi_stereobm *h, mi_stereobm_params *p);
/* Replaces: StereoBMImpl::compute, cudastereo/src/stereobm.cpp:134-191.  left,right: MI_8UC1, same size;
 * disp: MI_8UC1 of the same size (allocated by the caller / the C++ shim's OutputArray::create). */
MI_API int mi_stereobm_compute(mi_stereobm *h, const mi_mat *left, const mi_mat *right, mi_mat *disp, void *stream);
/* n pairs of one size through one handle: the block matching of all pairs is ONE launch (blockIdx.z = pair; the batch supplies the
 * waves, so the row bands are taller and their 2R-row start-up weighs less), prefilters and the textureness post-filter pair by
 * pair; every disparity map equals mi_stereobm_compute's.  The concurrency the reference offers for this is one StereoBM object per
 * stream (cudastereo/perf/perf_stereo.cpp). */
MI_API int mi_stereobm_compute_batch(mi_stereobm *h, int n, const mi_mat *lefts, const mi_mat *rights, mi_mat *disps, void *stream);
MI_API void mi_stereobm_destroy(mi_stereobm *h);

/* Stage-level entry points == cv::cuda::device::stereobm:: functions (cudastereo/src/stereobm.cpp:54-63).
 * Replaces: prefilter_xsobel  cudastereo/src/cuda/stereobm.cu:522-551 */
MI_API int mi_stereobm_prefilter_xsobel(const mi_mat *src, mi_mat *dst, int prefilter_cap, void *stream);
/* Replaces: prefilter_norm  stereobm.cu:557-599 */
MI_API int mi_stereobm_prefilter_norm(const mi_mat *src, mi_mat *dst, int prefilter_cap, int winsize, void *stream);
/* Replaces: stereoBM_CUDA  stereobm.cu:498-511 (memsets included).  min_ssd: MI_32SC1 scratch/out. */
MI_API int mi_stereobm_block_match(const mi_mat *left, const mi_mat *right, mi_mat *disp, mi_mat *min_ssd, int ndisp,
                                   int winsz, int uniqueness_ratio, int emulate_cuda_edge, void *stream);
/* Replaces: postfilter_textureness  stereobm.cu:698-711 (exact-integer definition, see oracle/stereobm_ref.c) */
MI_API int mi_stereobm_textureness(const mi_mat *img, mi_mat *disp, int winsz, float avg_texture_threshold, void *stream);

/* ====================================================================== Farneback ===== */

/* cv::OPTFLOW_* flags (main repo video/tracking.hpp), as passed to FarnebackOpticalFlow::create */
enum { MI_OPTFLOW_USE_INITIAL_FLOW = 4, MI_OPTFLOW_FARNEBACK_GAUSSIAN = 256 };

/* Parameters of cv::cuda::FarnebackOpticalFlow::create (cudaoptflow.hpp:285-293) */
typedef struct mi_farneback_params {
    int num_levels;
    double pyr_scale;
    int fast_pyramids;
    int win_size, num_iters, poly_n;
    double poly_sigma;
    int flags;
} mi_farneback_params;

typedef struct mi_farneback mi_farneback;

MI_API void mi_farneback_default_params(mi_farneback_params *p);
/* Replaces: cv::cuda::FarnebackOpticalFlow::create, cudaoptflow/src/farneback.cpp:485-489 */
MI_API int mi_farneback_create(const mi_farneback_params *p, mi_farneback **out);
MI_API int mi_farneback_set_params(mi_farneback *h, const mi_farneback_params *p);
MI_API int mi_farneback_get_params(const mi_farneback *h, mi_farneback_params *p);
/* Replaces: FarnebackOpticalFlowImpl::calc + calcImpl, cudaoptflow/src/farneback.cpp:167-199,314-482.
 * I0,I1: MI_8UC1 or MI_32FC1 (convertTo(CV_32F), no scaling), same size/type; flow: MI_32FC2 of the frame size,
 * read as the initial flow when MI_OPTFLOW_USE_INITIAL_FLOW.  One stream, no host synchronisation. */
MI_API int mi_farneback_calc(mi_farneback *h, const mi_mat *I0, const mi_mat *I1, mi_mat *flow, void *stream);
/* n independent pairs of identical size and type in one pass (blockIdx.z = pair in every kernel of the level loop).  Ordered on
 * `stream` like calc(): a level whose planes exceed the last-level cache runs group by group of pairs, every second group on a stream the
 * handle owns, forked from and joined back into `stream` inside the call (not while `stream` is being captured: one chain then).  The
 * frames and flows must stay valid until the work enqueued on `stream` has run -- the pyramid reads the caller's matrices in place at
 * every level.  Results are the bytes of n calc()s.
 * ONE calc in flight per handle: a handle owns its scratch planes AND the internal stream / events of the pair groups, so calls on one
 * handle must be ordered -- the same `stream`, or a synchronisation between calls on different streams or from different host threads
 * (the reference's object has the same rule: its GpuMat members are per-object scratch).  Use one handle per concurrent caller. */
MI_API int mi_farneback_calc_batch(mi_farneback *h, int n, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows, void *stream);
MI_API void mi_farneback_destroy(mi_farneback *h);

/* Stage-level entry points == cv::cuda::device::optflow_farneback:: functions (cudaoptflow/src/farneback.cpp:60-92);
 * 5-plane buffers are MI_32FC1 matrices of 5*rows x cols (planes stacked vertically) like the reference's.
 * Replaces: setPolynomialExpansionConsts + polynomialExpansionGpu  farneback.cpp:262-276, cuda/farneback.cu:122-151 */
MI_API int mi_farneback_poly_exp(const mi_mat *src, mi_mat *dst5, int poly_n, double poly_sigma, void *stream);
/* Replaces: setUpdateMatricesConsts + updateMatricesGpu  cuda/farneback.cu:244-264 */
MI_API int mi_farneback_update_matrices(const mi_mat *flowx, const mi_mat *flowy, const mi_mat *R0, const mi_mat *R1,
                                        mi_mat *M5, void *stream);
/* Replaces: boxFilter5Gpu / gaussianBlur5Gpu(BORDER_REPLICATE)  cuda/farneback.cu:415-450,598-651
 * (gaussian != 0: kernel getGaussianKernel(ksize, ksize/2*0.3f), farneback.cpp:460-464) */
MI_API int mi_farneback_blur5(const mi_mat *M5, mi_mat *dst5, int ksize, int gaussian, void *stream);
/* Replaces: updateFlowGpu  cuda/farneback.cu:289-300 */
MI_API int mi_farneback_update_flow(const mi_mat *M5, mi_mat *flowx, mi_mat *flowy, void *stream);
/* One fused inner iteration (blur5 + updateFlow + optional updateMatrices): updateFlow_boxFilter /
 * updateFlow_gaussianBlur, farneback.cpp:278-312.  M5out must not alias M5. */
MI_API int mi_farneback_iterate(const mi_mat *M5, const mi_mat *R0, const mi_mat *R1, mi_mat *flowx, mi_mat *flowy,
                                mi_mat *M5out, int ksize, int gaussian, int update_matrices, void *stream);
/* Replaces: setGaussianBlurKernel + gaussianBlurGpu  cuda/farneback.cu:495-536; border: 1 REPLICATE, 4 REFLECT101 */
MI_API int mi_farneback_gaussian_blur(const mi_mat *src, mi_mat *dst, int ksize, double sigma, int border, void *stream);
/* Replaces: cv::cuda::pyrDown on CV_32FC1  cudawarping/src/pyramids.cpp:66-94 */
MI_API int mi_pyr_down(const mi_mat *src, mi_mat *dst, void *stream);

/* =========================================================================== SURF ===== */

/* Public fields of cv::cuda::SURF_CUDA (xfeatures2d/cuda.hpp:182-189) */
typedef struct mi_surf_params {
    double hessian_threshold;
    int n_octaves, n_octave_layers;
    int extended;             /* 0: 64-float descriptors, 1: 128 */
    float keypoints_ratio;    /* maxFeatures = min(int(area * ratio), 65535) */
    int upright;
} mi_surf_params;

typedef struct mi_surf mi_surf;

MI_API void mi_surf_default_params(mi_surf_params *p);
/* Replaces: cv::cuda::SURF_CUDA::create / constructors, xfeatures2d/src/surf.cuda.cpp:257-285,444-448 */
MI_API int mi_surf_create(const mi_surf_params *p, mi_surf **out);
MI_API int mi_surf_set_params(mi_surf *h, const mi_surf_params *p);
MI_API int mi_surf_get_params(const mi_surf *h, mi_surf_params *p);
MI_API int mi_surf_descriptor_size(const mi_surf *h);                 /* SURF_CUDA::descriptorSize, surf.cuda.cpp:277-280 */
/* maxFeatures of SURF_CUDA_Invoker (surf.cuda.cpp:153): the keypoint matrix must have at least this many columns */
MI_API int mi_surf_max_features(const mi_surf *h, int rows, int cols, int *max_features);
/* Replaces: SURF_CUDA::operator()(img, mask, keypoints) = SURF_CUDA_Invoker ctor + detectKeypoints,
 * surf.cuda.cpp:137-215,369-378.  img: MI_8UC1; mask: MI_8UC1 of the same size or NULL; keypoints: MI_32FC1 with
 * ROWS_COUNT = 7 rows {X, Y, LAPLACIAN(int bits), OCTAVE(int bits), SIZE, ANGLE, HESSIAN} (cuda.hpp:89-99) and
 * >= max_features columns.  Features come out in a deterministic order (octave, layer, row, column).
 * Synchronises `stream` once to return the count (the reference's keypoints.cols = featureCounter). */
MI_API int mi_surf_detect(mi_surf *h, const mi_mat *img, const mi_mat *mask, mi_mat *keypoints, int *n_features, void *stream);
/* Replaces: SURF_CUDA::operator()(img, mask, keypoints, descriptors) = detectKeypoints + computeDescriptors,
 * surf.cuda.cpp:380-397 (137-236) WITHOUT the read-back of keypoints.cols between the two (:205-209): the descriptor kernels read the
 * feature count on the device.  keypoints as mi_surf_detect; descriptors: MI_32FC1, >= max_features rows x descriptorSize(); rows
 * [0, *n_features) are written.  Synchronises `stream` once, at the end, to return the count.  Same keypoints and descriptors, bit
 * for bit, as mi_surf_detect followed by mi_surf_compute_descriptors. */
MI_API int mi_surf_detect_and_compute(mi_surf *h, const mi_mat *img, const mi_mat *mask, mi_mat *keypoints, mi_mat *descriptors,
                                      int *n_features, void *stream);
/* n frames through one handle; masks may be NULL.  keypoints[i] / n_features[i] as mi_surf_detect. */
MI_API int mi_surf_detect_batch(mi_surf *h, int n, const mi_mat *imgs, const mi_mat *masks, mi_mat *keypoints, int *n_features, void *stream);
/* Replaces: SURF_CUDA_Invoker::findOrientation for provided keypoints, surf.cuda.cpp:217-225,391-393 */
MI_API int mi_surf_compute_orientation(mi_surf *h, const mi_mat *img, mi_mat *keypoints, int n_features, void *stream);
/* Replaces: SURF_CUDA_Invoker::computeDescriptors, surf.cuda.cpp:227-236 (+ normalize_descriptors).
 * descriptors: MI_32FC1, n_features x descriptorSize(). */
MI_API int mi_surf_compute_descriptors(mi_surf *h, const mi_mat *img, const mi_mat *keypoints, int n_features, mi_mat *descriptors,
                                       void *stream);
MI_API void mi_surf_release_memory(mi_surf *h);                       /* SURF_CUDA::releaseMemory */
MI_API void mi_surf_destroy(mi_surf *h);
/* Stage level.  Replaces: cv::cuda::integral (CV_8UC1 -> CV_32SC1 (rows+1)x(cols+1)), cudaarithm/src/cuda/integral.cu:62-83;
 * clamp_to_one applies cuda::min(mask, 1.0) first (surf.cuda.cpp:167). */
MI_API int mi_surf_integral(mi_surf *h, const mi_mat *img, int clamp_to_one, mi_mat *sum, void *stream);
/* Replaces: icvCalcLayerDetAndTrace_gpu, xfeatures2d/src/cuda/surf.cu:205-222; det/trace: MI_32FC1,
 * ((n_octave_layers+2) * (rows >> octave)) x cols, same step */
MI_API int mi_surf_det_trace(mi_surf *h, const mi_mat *sum, int octave, int n_octave_layers, mi_mat *det, mi_mat *trace, void *stream);

/* ======================================================== sparse PyrLK ===== */

/* cv::cuda::SparsePyrLKOpticalFlow::create(winSize = (21, 21), maxLevel = 3, iters = 30, useInitialFlow = false), cudaoptflow.hpp:203-223,
 * for CV_8UC1 frames; calc = PyrLKOpticalFlowBase::sparse (cudaoptflow/src/pyrlk.cpp:149-231) + sparseKernel (cuda/pyrlk.cu:148-340).
 * prev_pts / next_pts: 1 x N CV_32FC2 (next_pts is read when use_initial_flow), status: 1 x N CV_8UC1 (1 = tracked), err: NULL or
 * 1 x N CV_32FC1 (mean |J - I| over the window at level 0, in 8-bit units).  Windows of up to 1024 pixels. */
typedef struct mi_sparsepyrlk_params {
    int win_width, win_height, max_level, iters, use_initial_flow;
} mi_sparsepyrlk_params;
typedef struct mi_sparsepyrlk mi_sparsepyrlk;
MI_API void mi_sparsepyrlk_default_params(mi_sparsepyrlk_params *p);
MI_API int mi_sparsepyrlk_create(const mi_sparsepyrlk_params *p, mi_sparsepyrlk **out);
MI_API int mi_sparsepyrlk_set_params(mi_sparsepyrlk *h, const mi_sparsepyrlk_params *p);
MI_API int mi_sparsepyrlk_get_params(const mi_sparsepyrlk *h, mi_sparsepyrlk_params *p);
MI_API int mi_sparsepyrlk_calc(mi_sparsepyrlk *h, const mi_mat *prev_img, const mi_mat *next_img, const mi_mat *prev_pts, mi_mat *next_pts,
                               mi_mat *status, mi_mat *err, void *stream);
MI_API void mi_sparsepyrlk_destroy(mi_sparsepyrlk *h);

/* ---- the CPU SURF class's orientation and descriptor for given keypoints (xfeatures2d::SURF_Impl, surf.cpp:568-866) ----
 * cv::cuda::SURF_CUDA samples its descriptor patch differently from cv::xfeatures2d::SURF (the reference's own test only asks for a
 * 60 % nearest-neighbour agreement, test_surf.cuda.cpp:166-169).  These two entry points run the CPU class's arithmetic -- Haar
 * responses on the integral image, fastAtan2 polynomial, sequential 60-degree window search; bilinear window, INTER_AREA patch,
 * 4 x 4 x (4 | 8) bins -- for callers that need its numbers (the reference's known-answer vectors, misc/java/test/SURF*Test.java,
 * are reproduced).  keypoints: the CV_32FC1 7 x nFeatures matrix of SURF_CUDA (cuda.hpp:89-99); sum: CV_32SC1 (rows + 1) x (cols + 1)
 * from mi_surf_integral.  Orientation writes the ANGLE row (270 when upright) and sets SIZE to -1 for keypoints the CPU class erases
 * (no sample inside the image, surf.cpp:604-611,631-638); descriptors of erased keypoints are zero rows. */
MI_API int mi_surfcpu_orientation(const mi_mat *sum, mi_mat *keypoints, int n, int upright, void *stream);
MI_API int mi_surfcpu_descriptors(const mi_mat *img, const mi_mat *keypoints, int n, int extended, int upright, mi_mat *descriptors,
                                  void *stream);

/* ================================== brute-force descriptor matcher for SURF output (first part) ===== */

/* cv::cuda::DescriptorMatcher::createBFMatcher(NORM_L1 | NORM_L2) for CV_32F descriptors of up to 512 elements (SURF: 64 / 128),
 * cudafeatures2d/src/brute_force_matcher.cpp:296-1070 + cuda/bf_match.cu:92-183,559-587, cuda/bf_knnmatch.cu, cuda/bf_radius_match.cu.
 * Distance: k-ascending chain over the descriptor elements -- L2: sum = fma(d, d, sum), d = q[k] - t[k], result sqrtf(sum) (the
 * order of loopUnrolledCached, bf_match.cu:100-121, with nvcc's default mul+add contraction); L1: sum += fabsf(q[k] - t[k]).
 * Best lists are ordered by (distance, image index, train index): the first strict minimum in scan order, exact ties to the
 * lowest index (cv::BFMatcher's CPU rule; the reference's cross-thread reductions make its tie order scheduling dependent).
 * Integer descriptors -- the other rows of the reference's (depth, norm) table, brute_force_matcher.cpp:336-356: NORM_L1 on
 * MI_8UC1 / MI_16UC1 / MI_16SC1 / MI_32SC1, NORM_HAMMING on MI_8UC1 / MI_16UC1 / MI_32SC1 -- go through the same entry points
 * (descriptors of up to 128 elements, k <= 16; exact integer distances returned as float).  Any other (depth, norm) pair fails
 * with MI_ERR_BAD_TYPE like the reference's StsUnsupportedFormat. */
typedef struct mi_bfmatcher mi_bfmatcher;
enum { MI_NORM_L1 = 2, MI_NORM_L2 = 4, MI_NORM_HAMMING = 6 };   /* cv::NORM_L1, cv::NORM_L2, cv::NORM_HAMMING */
MI_API int mi_bf_create(int norm_type, mi_bfmatcher **out);
MI_API void mi_bf_destroy(mi_bfmatcher *h);
/* matchSingle: query n_q x D, train n_t x D (MI_32FC1), mask NULL or MI_8UC1 n_q x n_t (non-zero = allowed);
 * train_idx MI_32SC1 1 x n_q (-1 = none), distance MI_32FC1 1 x n_q (FLT_MAX = none). */
MI_API int mi_bf_match(mi_bfmatcher *h, const mi_mat *query, const mi_mat *train, const mi_mat *mask, mi_mat *train_idx,
                       mi_mat *distance, void *stream);
/* knnMatch with k = 2 (bf_knnmatch.cu match2): train_idx MI_32SC2 1 x n_q, distance MI_32FC2 1 x n_q. */
MI_API int mi_bf_knn_match2(mi_bfmatcher *h, const mi_mat *query, const mi_mat *train, const mi_mat *mask, mi_mat *train_idx,
                            mi_mat *distance, void *stream);
/* knnMatchAsync for any k >= 1 over a collection of n_trains train sets (brute_force_matcher.cpp:574-725; n_trains = 1 is the
 * (query, train) form).  trains: n_trains mi_mat, MI_32FC1 n_t[m] x D; masks: NULL, or n_trains mi_mat each with data == NULL
 * (none) or MI_8UC1 n_q x n_t[m].  train_idx / img_idx MI_32SC1 n_q x k, distance MI_32FC1 n_q x k (img_idx may be NULL);
 * entries past the number of candidates: -1 / -1 / FLT_MAX.  The reference's distance matrix + k extraction rounds (k != 2) and
 * its per-image host merge (:512-571) are replaced by per-lane sorted lists; the result is the same list. */
MI_API int mi_bf_knn_match(mi_bfmatcher *h, const mi_mat *query, const mi_mat *trains, const mi_mat *masks, int n_trains, int k,
                           mi_mat *train_idx, mi_mat *img_idx, mi_mat *distance, void *stream);
/* radiusMatchAsync (brute_force_matcher.cpp:841-980, bf_radius_match.cu:58-118): every train descriptor with mask != 0 and
 * distance < max_distance.  train_idx / img_idx MI_32SC1 and distance MI_32FC1 are n_q x cols (the reference sizes cols =
 * max(n_t / 100, n_q), resp. n_q for collections; img_idx may be NULL), n_matches MI_32SC1 1 x n_q counts every hit (it may
 * exceed cols); the first min(n_matches, cols) entries of a row are the hits in ascending (image, train) order -- the reference
 * stores them in atomicInc order (scheduling dependent) and sorts on the host; entries past them are left untouched. */
MI_API int mi_bf_radius_match(mi_bfmatcher *h, const mi_mat *query, const mi_mat *trains, const mi_mat *masks, int n_trains,
                              float max_distance, mi_mat *train_idx, mi_mat *img_idx, mi_mat *distance, mi_mat *n_matches,
                              void *stream);

/* ======================================================== dense PyrLK ===== */

/* cv::cuda::DensePyrLKOpticalFlow::create(winSize = (13, 13), maxLevel = 3, iters = 30, useInitialFlow = false), cudaoptflow.hpp;
 * use_initial_flow is stored but, like the reference's dense path (pyrlk.cpp:238-299), not used. */
typedef struct mi_densepyrlk_params {
    int win_width, win_height, max_level, iters, use_initial_flow;
} mi_densepyrlk_params;
typedef struct mi_densepyrlk mi_densepyrlk;
MI_API void mi_densepyrlk_default_params(mi_densepyrlk_params *p);
MI_API int mi_densepyrlk_create(const mi_densepyrlk_params *p, mi_densepyrlk **out);
MI_API int mi_densepyrlk_set_params(mi_densepyrlk *h, const mi_densepyrlk_params *p);
MI_API int mi_densepyrlk_get_params(const mi_densepyrlk *h, mi_densepyrlk_params *p);
/* Replaces: DensePyrLKOpticalFlowImpl::calc + PyrLKOpticalFlowBase::dense + pyrlk::denseKernel, cudaoptflow/src/pyrlk.cpp:238-299,
 * 379-392, cuda/pyrlk.cu:709-847.  prev, next: MI_8UC1 of the same size; flow: MI_32FC2 of the image size.  Pixels whose
 * structure tensor is singular or whose track leaves the image keep the value their (u, v) buffer held (the reference returns
 * without writing; buffers start at zero). */
MI_API int mi_densepyrlk_calc(mi_densepyrlk *h, const mi_mat *prev, const mi_mat *next, mi_mat *flow, void *stream);
MI_API void mi_densepyrlk_destroy(mi_densepyrlk *h);

/* ======================================================== StereoSGM ===== */

enum { MI_SGM_MODE_HH = 1, MI_SGM_MODE_HH4 = 3 };   /* cv::StereoSGBM::MODE_HH (8 paths), MODE_HH4 (4 paths) */
/* cv::cuda::createStereoSGM(minDisparity = 0, numDisparities = 128, P1 = 10, P2 = 120, uniquenessRatio = 5, mode = MODE_HH4),
 * cudastereo.hpp; numDisparities in {64, 128, 256}.  emulate_cuda_quirks (default 1): the 8-bit sorting of the 16-bit median's
 * scalar columns (stereosgm.cu:1871-1896) and the width/16 x height/16 launch of the consistency check (:1985); 0 = clean. */
typedef struct mi_stereosgm_params {
    int min_disparity, num_disparities, P1, P2, uniqueness_ratio, mode, emulate_cuda_quirks;
} mi_stereosgm_params;
typedef struct mi_stereosgm mi_stereosgm;
MI_API void mi_stereosgm_default_params(mi_stereosgm_params *p);
MI_API int mi_stereosgm_create(const mi_stereosgm_params *p, mi_stereosgm **out);
MI_API int mi_stereosgm_set_params(mi_stereosgm *h, const mi_stereosgm_params *p);
MI_API int mi_stereosgm_get_params(const mi_stereosgm *h, mi_stereosgm_params *p);
/* Replaces: StereoSGMImpl::compute, cudastereo/src/stereosgm.cpp:96-144.  left, right: MI_8UC1 or MI_16UC1; disparity: MI_16SC1
 * of the image size, fixed point with 4 fractional bits, (minDisparity - 1) * 16 where invalid. */
MI_API int mi_stereosgm_compute(mi_stereosgm *h, const mi_mat *left, const mi_mat *right, mi_mat *disparity, void *stream);
MI_API void mi_stereosgm_destroy(mi_stereosgm *h);
/* Stage level = the functions the reference unit-tests against CPU twins (cudastereo/test/test_sgm_funcs.cpp):
 * census_transform::censusTransform (cuda/stereosgm.cu:460-480): src MI_8UC1 | MI_16UC1 -> dst MI_32SC1, 0 on the border */
MI_API int mi_sgm_census(const mi_mat *src, mi_mat *dst, void *stream);
/* path_aggregation::{horizontal,vertical,oblique}::aggregate*Path (cuda/stereosgm.cu:483-1330): dense MI_32SC1 census images ->
 * dst MI_8UC1 1 x (width * height * num_disparities), layout [pixel][disparity]; (dx, dy) = the step of the path */
MI_API int mi_sgm_aggregate_path(const mi_mat *left_census, const mi_mat *right_census, mi_mat *dst, int num_disparities,
                                 int min_disparity, int p1, int p2, int dx, int dy, void *stream);
/* winner_takes_all::winnerTakesAll (cuda/stereosgm.cu:1433-1616): src MI_8UC1 1 x (w * h * D * num_paths) -> left, right MI_16SC1 */
MI_API int mi_sgm_winner_takes_all(const mi_mat *src, mi_mat *left, mi_mat *right, int num_disparities, int num_paths, float uniqueness,
                                   int subpixel, void *stream);

/* ========================================== DisparityBilateralFilter (first part) ===== */

/* cv::cuda::createDisparityBilateralFilter(ndisp = 64, radius = 3, iters = 1), cudastereo.hpp + defaults of
 * cudastereo/src/disparity_bilateral_filter.cpp:125-136 (edge threshold 0.1, max disc threshold 0.2, sigma range 10). */
typedef struct mi_disp_bilateral_params {
    int ndisp, radius, iters;
    float edge_threshold, max_disc_threshold, sigma_range;
} mi_disp_bilateral_params;
typedef struct mi_disp_bilateral mi_disp_bilateral;
MI_API void mi_disp_bilateral_default_params(mi_disp_bilateral_params *p);
MI_API int mi_disp_bilateral_create(const mi_disp_bilateral_params *p, mi_disp_bilateral **out);
MI_API int mi_disp_bilateral_set_params(mi_disp_bilateral *h, const mi_disp_bilateral_params *p);   /* setRadius / setSigmaRange rebuild the tables, .cpp:138-148 */
MI_API int mi_disp_bilateral_get_params(const mi_disp_bilateral *h, mi_disp_bilateral_params *p);
/* Replaces: DispBilateralFilterImpl::apply + device::disp_bilateral_filter<T>, disparity_bilateral_filter.cpp:150-190,
 * cuda/disparity_bilateral_filter.cu:76-199.  disp: MI_8UC1 or MI_16SC1; img: MI_8UC1 or MI_8UC3 of the same size; dst: type and
 * size of disp (may be disp itself).  Every red/black pass reads the map as it was when the pass started (the reference's
 * in-place passes race on same-colour pixels inside the window; see oracle/dbf_ref.c). */
MI_API int mi_disp_bilateral_apply(mi_disp_bilateral *h, const mi_mat *disp, const mi_mat *img, mi_mat *dst, void *stream);
MI_API void mi_disp_bilateral_destroy(mi_disp_bilateral *h);

/* ================================================= superres optical-flow adapters ===== */

/* Replaces: cv::superres::convertToType(GpuMat, CV_8UC1) as used by GpuOpticalFlow::calc, superres/src/optical_flow.cpp:469-470
 * = convertToCn (cuda::cvtColor BGR2GRAY / BGRA2GRAY) then convertToDepth (GpuMat::convertTo(CV_8U, 255 / maxVal(depth))),
 * superres/src/input_array_utility.cpp:165-234,291-314.  src: 8U / 16U / 32F with 1, 3 (BGR) or 4 (BGRA) channels;
 * dst: MI_8UC1 of the same size.  One kernel, no intermediate image. */
MI_API int mi_superres_to_gray8(const mi_mat *src, mi_mat *dst, void *stream);
/* Replaces: cuda::split(flow, flows) in Farneback_CUDA::impl / DualTVL1_CUDA::impl, superres/src/optical_flow.cpp:737-741,834-838.
 * flow: MI_32FC2; u, v: MI_32FC1 of the same size. */
MI_API int mi_split_flow(const mi_mat *flow, mi_mat *u, mi_mat *v, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MIFLOW_C_API_H */
