// Internal launch API of the TV-L1 HIP kernels (tvl1_kernels.hip).  Not part of the C-ABI.
#pragma once
#include "mi_common.h"

namespace mi {
namespace tvl1 {

// All scratch planes of one pyramid level are dense float planes, `ld` floats per row
// (multiple of 64), `ps` floats between consecutive pairs of the batch.
struct Geo {
    int w, h, ld;
    long long ps;  // pair stride (floats)
    int batch;
};

// Device-side loop control (epsilon > 0).  One slot per iteration launch q and pair b:
//   S[b*Q + q] = {cur_in, active};  E[b*Q + q] = sum(du^2) of launch q, 2^-24 fixed point.
struct Ctl {
    int2 *S;
    unsigned long long *E;
    int Q;           // slots per pair
    int q;           // this launch's slot
    int q_prev;      // previous iteration launch's slot (-1: none, cur_in = 0)
    int first_of_warp;   // error := max  => always active
    int reset_cur;       // first launch of a scale: cur_in := 0
    double thr;      // scaledEpsilon (already rounded through float for CPU_REF)
    // cv::cuda's check schedule (cudaoptflow/src/tvl1flow.cpp:357-377): the error is summed only at odd iterations n and only
    // while prevError < scaledEpsilon; after a failed or skipped check prevError shrinks by scaledEpsilon per iteration.
    double *P;       // per slot: prevError as seen by that launch (sched only)
    int sched;       // 0: check every iteration (CPU class), 1: the schedule above (MI_SEM_CUDA_COMPAT)
    int n;           // iteration index inside the warp
    int need_done;   // warp kernels only: run only where slot q_prev says the previous warp has stopped (a launch enqueued AHEAD of
                     // the host's knowledge of that, tvl1_api.cpp host feedback); elsewhere the launch leaves everything untouched
};

// (host code reads the flags too: mi_tvl1_last_iterations, the host feedback of tvl1_api.cpp)
// flags of a control slot (S[..].y): bit 0 = the launch moved the state to the other buffer set, bit 1 = it summed the
// error (one-iteration launches), bit 2 = the warp has converged, bits 8..15 = iterations this launch contributed
#define MI_SLOT_FLIP 1
#define MI_SLOT_CHECKED 2
#define MI_SLOT_DONE 4
#define MI_SLOT_ITERS(y) (((y) >> 8) & 0xff)

struct PtrTab {          // per-pair external image pointers (device array)
    const void *a, *b;   // I0, I1 (convert) or unused
    void *out;           // flow (pack)
    long long step_a, step_b, step_out;  // bytes
};

struct IterPlanes {
    const float *ix, *iy, *g, *rc;   // I1wx, I1wy, grad, rho_c
    float *u[2][3];                  // [set][u1,u2,u3]   (u3 only when gamma != 0)
    float *p[2][6];                  // [set][p11,p12,p21,p22,p31,p32]
    float gamma;                     // illumination-change weight; 0 = the 2-channel model
    int err_u3;                      // 1: the convergence error includes (du3)^2 (CPU class, optflow tvl1flow.cpp:1110); 0: cv::cuda
};

// type: MI_8UC1 (x1) or MI_32FC1 (x255)
int convert(const PtrTab *tab_dev, int type, float *I0, float *I1, const Geo &g, hipStream_t s);
int zero_planes4(float *const p[4], size_t n, hipStream_t s);   // four planes of n floats := 0, one launch
// flow (MI_32FC2, tab.out) -> u1,u2
int unpack_flow(const PtrTab *tab_dev, float *u1, float *u2, const Geo &g, hipStream_t s);
// u (set resolved by ctl) -> flow (MI_32FC2)
int pack_flow(const PtrTab *tab_dev, const float *u1[2], const float *u2[2], const Geo &g,
              const Ctl *ctl, int cur_host, hipStream_t s);
// nplanes (<=3) planes resized in one launch; src set resolved via ctl when src_sets == 2
int resize(int semantics, int nplanes, const float *const src[3][2], int src_sets, float *const dst[3],
           const Geo &gs, const Geo &gd, double inv_scale_x, double inv_scale_y,
           const float post_scale[3], const Ctl *ctl, int cur_host, hipStream_t s);
int gradient(const float *src, float *dx, float *dy, const Geo &g, hipStream_t s);
// cv::medianBlur(ksize 3|5) of u1,u2 in place (through tmp planes); device-side loop control like iterate()
int median_flow(int ksize, float *const u1[2], float *const u2[2], float *tmp1, float *tmp2, const Geo &g, const Ctl *ctl, int cur_host,
                hipStream_t s);
// pk = one float4 {I1, I1x, I1y, 0} per pixel (4 * g.ps floats per pair, 16-B aligned)
int gradient_pack(const float *src, float *pk, const Geo &g, hipStream_t s);
int pack3(const float *a, const float *b, const float *c, float *pk, const Geo &g, hipStream_t s);
int warp(int semantics, const float *I0, const float *pk, const float *u1[2], const float *u2[2], float *I1w,
         float *I1wx, float *I1wy, float *grad, float *rho, const float *cubic_tab_dev, const Geo &g, const Ctl *ctl,
         int cur_host, hipStream_t s);
// the same with the centred gradient of I1 derived inside the kernel from a 6 x 6 window of I1 (tvl1_warp_kernels.hip): no packed
// plane, half the gathered bytes, bit-identical results
// fast: the three bicubic sums in separable form (rounding differs from the reference's tap-by-tap order; fast-math paths only)
// lds: windows read from an LDS-staged region of I1 (1) or gathered from global memory (0); -1 = the tuning default
// zoom: the first warp of a scale samples the coarser scale's flow itself (k_resize's arithmetic), writes it to (u1o, u2o) and uses it
struct WarpZoom { const float *u1c, *u2c; float *u1o, *u2o; Geo gc; double inv_scale_x, inv_scale_y; float post; };
bool warp_zoom_ok();
int warp_fused(int semantics, bool fast, int lds, const float *I0, const float *I1, const float *u1[2], const float *u2[2], float *I1w, float *I1wx,
               float *I1wy, float *grad, float *rho, const float *cubic_tab_dev, const Geo &g, const Ctl *ctl, int cur_host,
               hipStream_t s, const WarpZoom *zoom = nullptr);
// one fused iteration (estimateU + estimateDualVariables), set cur -> set cur^1.
// p_zero: p_in is known to be all-zero (first iteration of a scale) and is not read.
int iterate(bool exact, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut,
            bool p_zero, const Ctl *ctl, int cur_host, hipStream_t s);

// Temporally blocked fast-math iteration (tvl1_tbr_kernels.hip): T fused iterations in one HBM pass,
// set cur -> cur^1.  Supported T: 1,2,3,4,5,6,8,10.  rows_per_band <= 0: auto.
// skip_p_out: the launch stores u only (the last pass of a scale: nobody reads its p).  pl.g == nullptr: no |grad|^2 plane, the kernel
// forms it from I1wx, I1wy (only where tb_nograd_ok says so)
// independent_waves (test hook of the stage-level entry): the kernel whose waves each own a 64-column strip, never the joined form
int iterate_tb(int T, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut, bool p_zero,
               int cur, int rows_per_band, hipStream_t s, bool skip_p_out = false, bool independent_waves = false);
bool tb_nograd_ok(int T, const Geo &g);
// the warp of a one-pass warp INSIDE that pass (k_iterate_tbr FW): no warp launch, no I1wx / I1wy / rho_c planes in HBM; bit-identical
bool tb_fused_ok(int T, const Geo &g, int semantics, bool fast_warp);
int iterate_tb_fused(int semantics, const float *I0, const float *I1, const float *cubic_tab_dev, int T, const IterPlanes &pl, const Geo &g,
                     float l_t, float theta, float taut, bool p_zero, int cur, hipStream_t s, bool skip_p_out);
bool tb_spec_nograd_ok(const Geo &g);   // the same for the speculative steps of the convergence-checked path
int tb_max_block();
// Register-tile formulation of the same fused iterations for the small pyramid levels (tvl1_tile_kernels.hip): nit in
// 1..tile_max_block() iterations per launch, bit-identical to iterate_tb.  variant < 0: default of the table.
int iterate_tile(int variant, int nit, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut, bool p_zero, int cur,
                 hipStream_t s);
int tile_max_block();
int tile_owned_rows();   // rows a tile of the default variant owns
int tile_rows_for(const Geo &g);   // rows (owned + margins) of the tiles level g runs on
int tb_query_plan(int T, const Geo &g, int *kernel, int *rows);   // kernel 0 = streaming (band height), 1 = register tile
int tile_variants();
bool tile_eligible(const Geo &g);   // this level (pixels x pairs) runs on the register-tile kernel
// the same in exact math (bit-identical to T one-iteration launches of iterate(exact = true)); T in 1..tb_exact_max_block()
int iterate_tb_exact(int T, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut, bool p_zero, int cur, hipStream_t s);
int tb_exact_max_block();
// speculative steps of the convergence-checked path (epsilon > 0, fast math): see k_iterate_tbr MODE 1
// Speculative step of the convergence-checked path (k_iterate_tbr MODE 1), host-side constants of one launch.
struct SpecK {
    int4 *X;            // per slot: {iterations accepted in this warp, iterations this launch ran speculatively (0: none),
                        //            bits of the last accepted iteration's error / threshold, unused}
    int e0_prev;        // first error-sum index of the previous launch's block
    int final_launch;   // 1: settle the previous block (accept / replay) only -- the last launch of a warp
    int iters;          // iteration limit of the warp (iterations x inner iterations)
    int t_after;        // iterations the later launches of this warp can still run (keeps the limit reachable)
    int q_hist;         // slot whose X.x = iteration count of an earlier warp: the first block's estimate is
    int hist_num, hist_den;   //   floor(count * num / den), at least 1 (-1: none)
    int slack;          // mi_tvl1_params.stop_slack
    // History of the handle's previous calc (same geometry and batch): the iterations THIS warp of THIS pair slot needed then --
    // consecutive frame pairs of a video stop within an iteration of each other, so it is the best estimate of a block length
    // there is.  Like every estimate it cannot change a result, only the number of passes.  Double-buffered by call parity (a
    // launch's workgroups read h_in while its writer thread stores h_out).  nullptr: no usable history.
    const int *h_in;
    int *h_out;
    // Host feedback without a copy: the launch's writer thread stores {fb_seq << 2 | was-done-before << 1 | done, iterations accepted
    // so far} per pair into pinned host memory right after its decision, i.e. when the launch STARTS; the host polls it.  nullptr: none.
    int *fb_flag;
    int fb_seq;
};

int tb_spec_plan(int n, int warp_index, bool large_level, int *blocks, int max_blocks);
// the same step on register tiles (tvl1_tile_kernels.hip); iterate_tb_spec dispatches to it where tile_eligible(g)
int iterate_tile_spec(int T, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut, const Ctl &ctl, const SpecK &sk, int e0,
                      hipStream_t s);
int iterate_tb_spec(int T, const IterPlanes &pl, const Geo &g, float l_t, float theta, float taut, bool p_zero, const Ctl &ctl,
                    const SpecK &sk, int e0, hipStream_t s);
// cost-model decomposition of n iterations into supported blocks (largest first); returns the count
int tb_plan(int n, int cap, int *blocks, int max_blocks);
// the same for level g: greedy blocks of tile_max_block() where the level runs on the register-tile kernel
int tb_plan_level(const Geo &g, int n, int cap, int *blocks, int max_blocks, bool gam = false);   // gam: the illumination channel's block set on streaming levels (10, 5, 2, 1)
// largest supported block <= n (n >= 1)
inline int tb_pick_block(int n, int cap)
{
    static const int sup[] = {10, 8, 6, 5, 4, 3, 2, 1};
    for (int t : sup) if (t <= n && t <= cap) return t;
    return 1;
}
int dbg_lane_shift(int *out_dev, hipStream_t s);
int tb_jw_fault(int *fault_host);   // sticky fault flag of the joined-wave blocked kernels (synchronises the device)

void host_cubic_table(float tab[128]);  // cv::remap INTER_CUBIC phase table (a = -0.75)

}  // namespace tvl1
}  // namespace mi
