// Internal helpers shared by the miflow translation units (not part of the C-ABI).
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "miflow/c_api.h"

namespace mi {

void set_error(const char *fmt, ...);

#define MI_HIP_TRY(expr)                                                                  \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            mi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? MI_ERR_OOM : MI_ERR_HIP;                   \
        }                                                                                 \
    } while (0)

#define MI_REQUIRE(cond, code, ...)        \
    do {                                   \
        if (!(cond)) {                     \
            mi::set_error(__VA_ARGS__);    \
            return (code);                 \
        }                                  \
    } while (0)

// Tuning switches (environment, read ONCE for the process under std::call_once; DESIGN.md lists them).  None is needed
// in production: every default is the measured best.
struct Tuning {
    int x_skip;              // MIFLOW_X_SKIP (EXPERIMENTS BUILD ONLY; wrong results): 1 = no warp launches after a level's first, 2 = no iteration launches; always 0 in the release library
    int warp_legacy;         // MIFLOW_WARP=pk: packed-float4 gather warp (the round-1 kernel) instead of the fused-gradient one
    int warp_tile;           // MIFLOW_WARP_TILE: pixels of a wave along x in the warp kernels (64 | 32 | 16)
    int warp_lds;            // MIFLOW_WARP_LDS: windows of the fused-gradient warp read from an LDS-staged region of I1 (1) or gathered from global memory (0)
    int warp_fast;           // MIFLOW_WARP_FAST: fast-math calcs form the warp's bicubic sums separably (1: +4.7 % pairs/s, 2-3 x the EPE against the oracle) or tap by tap in the reference's order (0); -1 (default): separably under MI_SEM_CUDA_COMPAT, whose map is not quantised (EPE 1.2e-5 either way), tap by tap under MI_SEM_CPU_REF
    int warp_np;             // MIFLOW_WARP_NP: patches a wave of the fused-gradient warp kernel walks (1 | 2 | 4)
    int tb_swz;             // MIFLOW_TB_SWZ: XCD-aware workgroup remap of the blocked iteration kernels
    int warp_zoom;          // MIFLOW_WARP_ZOOM: the first warp of a scale zooms the coarser flow itself instead of a resize launch (1; default 0: measured slower)
    int tb_p16;             // MIFLOW_TB_P16=1 (opt-in, CHANGES RESULTS within the fast path's tolerance): p between the passes of a scale as signed 16-bit fixed point
    int tb_nograd;          // MIFLOW_TB_NOGRAD: the blocked pass forms |grad|^2 itself and the warp does not store the plane (1, default) / stored plane (0)
    int tb_hist;            // MIFLOW_TB_HIST: block lengths of the convergence-checked path from the handle's previous calc (1, default)
    int fb_poll;            // MIFLOW_FB_POLL: host feedback through flags the deciding launch writes into pinned host memory when it STARTS (1, default) or a copy of the control slots behind the launch (0, round 3)
    int fb_ahead;           // MIFLOW_FB_AHEAD: polled host feedback enqueues the next warp's kernel (gated on the device) before it polls (1, default)
    int tb_skip_p;          // MIFLOW_TB_SKIP_P: the last pass of a scale does not store p (1, default)
    int tb_fw;              // MIFLOW_TB_FW (experiments build): 1 = a warp whose iterations are one pass of the T = 10 kernel runs INSIDE that pass (producer waves); 0 (default, and always in the release library) = its own launch
    int tb_jw;              // MIFLOW_TB_JW (experiments build): joined-wave form of the T = 10 blocked iteration kernel; the release library holds form 2 only
    int tb_jw_spec;         // MIFLOW_TB_JW_SPEC: ... of the speculative steps as well
    int tb_ppl, tb_wps, tb_pf;   // MIFLOW_TB_VARIANT=ppl,wps,pf (-1: table default)
    int tb_force;            // MIFLOW_TB_FORCE: greedy blocks of exactly the cap (tuning sweeps)
    int tb_plan_wps;         // MIFLOW_TB_WPS: waves/SIMD the band planner assumes (0: table)
    int tb_rows;             // MIFLOW_TB_ROWS: band height (0: planner)
    int tb_verbose;          // MIFLOW_TB_VERBOSE
    long long tile_maxpx;    // MIFLOW_TILE_MAXPX: levels of at most this many pixels x pairs iterate on the register-tile kernel (0: never)
    int tile_spec;           // MIFLOW_TILE_SPEC: speculative steps of the convergence-checked path on the register-tile kernel where it is eligible (1) or always on the streaming kernel (0)
    int tile_variant;        // MIFLOW_TILE_VARIANT: index into the (rows per wave, waves) table of tvl1_tile_kernels.hip; -1 (default) = by grid size
    int tile_fb_block;       // MIFLOW_TILE_FB_BLOCK: block length of the speculative steps of a one-or-two-pair calc on the register-tile kernel
    int tile_fb_model;       // MIFLOW_TILE_FB_MODEL: a one-or-two-pair calc picks each warp's block length / tile margin (4 | 7 | 10) from the previous calc's count by a cost model; the value = its microseconds per pass (7, default; 0 = off)
    int tile_swz, warp_swz;  // MIFLOW_TILE_SWZ / MIFLOW_WARP_SWZ: XCD-contiguous tile order of the register-tile kernel / the fused-gradient warp kernel (1, default)
    int tile_small_wgs;      // MIFLOW_TILE_SMALL_WGS: grids of fewer 64-row tiles than this run on the 48-row / 8-wave variant (1024)
    int lanes;               // MIFLOW_LANES: internal streams a TV-L1 batch is split over (0: automatic)
    int exact_tb;            // MIFLOW_EXACT_TB: exact math, fixed work: fused blocks (1) or one launch per iteration (0)
    int spec;                // MIFLOW_SPEC: speculative blocked convergence path (1) or one launch per iteration (0)
    int sbm_texfuse;         // MIFLOW_SBM_TEXFUSE: StereoBM textureness post-filter in one launch without the |Sobel| plane (1, default) or as two passes (0)
    int sbm_swz;             // MIFLOW_SBM_SWZ: StereoBM tiles in XCD-contiguous order (1, default)
    int sbm_wt;              // MIFLOW_SBM_WT: StereoBM winner-take-all through an LDS transposition (1, default) or the transposed DPP reduction (0)
    int fb_rows, fb_swz;     // MIFLOW_FB_ROWS (4 | 8 rows per workgroup), MIFLOW_FB_SWZ (XCD-contiguous tile order) of the tiled kernel
    int fb_group_mb;         // MIFLOW_FB_GROUP_MB: a batched Farneback level runs its matrix update + all iterations group by group of pairs whose 22 planes fit this many MB (the last-level cache; 0 = whole batch per launch)
    int fb_fuse;             // MIFLOW_FB_FUSE: Farneback few-launch forms (resize sampled inside poly_exp / update_matrices, merge in the last iteration): -1 = small calls (default), 0, 1
    int fb_pair;             // MIFLOW_FB_PAIR: Farneback two iterations per launch (k_iterate2_t): -1 = levels that underfill the device (default), 0, 1
    int fb_narrow;           // MIFLOW_FB_NARROW: Farneback iteration kernel on 64 x 4 tiles: -1 = where the 256-column grid underfills the device (default), 0 = never, 1 = always
    int fb_group_streams;    // MIFLOW_FB_GROUP_STREAMS (experiments build): the pair groups of a batched Farneback level run as two chains on two streams (2) or one after the other (1)
    int fb_poly_tiled;       // MIFLOW_FB_POLY_TILED (experiments build): Farneback polynomial expansion on 8-row tiles (1) or one row per workgroup (0)
    int fb_direct;           // MIFLOW_FB_DIRECT (experiments build): the Farneback pre-blur reads the caller's CV_8UC1 / CV_32FC1 matrices (1) or converted f32 planes (0)
    int fb_blur_tiled;       // MIFLOW_FB_BLUR_TILED (experiments build): Farneback pyramid pre-blur tiled over 8-14 rows (1) or one row per workgroup (0)
    int fb_tiled;            // MIFLOW_FB_TILED: Farneback iteration kernel tiled over 4 rows (1) or one row per workgroup (0)
};
const Tuning &tuning();
// switches outside the Tuning table that only the experiments build reads (see mi_common.cpp)
#ifdef MIFLOW_EXPERIMENTS
#define MI_EXP_ENV(name) getenv(name)
#else
#define MI_EXP_ENV(name) ((const char *)nullptr)
#endif

// SIMDs of the current device (4 per CU; 1024 on MI355X), queried once per device
// Temporary device buffers of the stage-level / self-test entry points: freed on every return path.
struct DevTmp {
    std::vector<void *> bufs;
    DevTmp() = default;
    DevTmp(const DevTmp &) = delete;
    DevTmp &operator=(const DevTmp &) = delete;
    ~DevTmp() { for (void *p : bufs) (void)hipFree(p); }
    template <class T>
    hipError_t alloc(T **out, size_t n)
    {
        void *p = nullptr;
        const hipError_t e = hipMalloc(&p, sizeof(T) * n);
        if (e == hipSuccess) bufs.push_back(p);
        *out = (T *)p;
        return e;
    }
};
// Large device blocks (the per-lane TV-L1 arenas) come from a small process-wide cache: a block released by a handle is kept
// (at most 4 blocks / MIFLOW_CACHE_GB = 24 GB PER DEVICE and MIFLOW_CACHE_TOTAL_GB = 4 x that per process; with an event behind
// its previous owner's last work, which the next taker waits for: nothing of the previous owner is still in flight when the block is
// used again, and no other handle's stream is stalled by a destroy) and handed to the next request of a similar size on the same device.  Measured on
// MI355X / ROCm 7.2 (r02q): an arena obtained by hipMalloc right after a hipFree of the same size runs the same kernels 25 %
// slower than the freed one did (390 vs 520 pairs/s, class defaults), i.e. create / destroy cycles of handles must not go
// through the driver.  mi_release_cached_memory() returns everything to the driver.
int big_alloc(void **p, size_t bytes, size_t *capacity);
void big_free(void *p, size_t capacity, hipEvent_t ready = nullptr, bool busy = true);
void big_trim();
int device_simds();

static inline int div_up(int a, int b) { return (a + b - 1) / b; }
static inline int align_up(int a, int b) { return div_up(a, b) * b; }

// Dense-row float plane in scratch memory: row pitch `ld` floats (multiple of 64 = 256 B),
// base 256-B aligned, so every row start is dwordx4-aligned.
struct PlaneF {
    float *p;
    int ld;
};

}  // namespace mi
