// The PLAN of one mi_farneback_calc_batch call (farneback_api.cpp, round 6): which pyramid levels exist, their geometry, and the form
// each level's launches take -- how the coarsest flow starts, how a finer level takes the coarser flow, pair groups, two iterations per
// launch.  Pure host arithmetic over the call's shape and the tuning knobs, no HIP types: tests/cpp/fb_plan_test.cpp compiles it alone.
// The level loop itself (enqueue_level) only EXECUTES a plan.  Reference: FarnebackOpticalFlowImpl::calcImpl, cudaoptflow/src/farneback.cpp:314-482.
#pragma once
#include "fb_groups.h"
#include <cmath>
#include <vector>

namespace mi {
namespace fb {

struct FbShape {
    int W, H, B;                     // frame size, pairs of the batch
    int num_levels; double pyr_scale; bool fast_pyramids;
    int num_iters;
    bool use_init;                   // OPTFLOW_USE_INITIAL_FLOW
};
struct FbKnobs {
    int fuse, pair;                  // MIFLOW_FB_FUSE / MIFLOW_FB_PAIR: -1 = automatic, 0 / 1 = forced
    int group_mb;                    // MIFLOW_FB_GROUP_MB: last-level-cache budget of the pair groups in flight
    int chains;                      // streams the groups of a level may alternate over: 2, or 1 (tuning; always 1 while the caller's stream is captured)
    int simds;                       // SIMDs of the device (the underfill thresholds are in workgroups per CU)
    bool iterate2_ok;                // a two-iteration kernel exists for the window size
};
struct FbLevel {
    int w, h, ld;                    // level image (farneback.cpp:374-395); ld = floats per plane row
    double scale, sigma;
    int smooth;                      // pre-blur kernel size (odd, >= 3)
    // how the level's flow planes start
    bool coarsest;
    bool init_resize;                // coarsest, initial flow given, not level 0: resized from the caller's flow (:398-404)
    bool zero_flow;                  // coarsest, no initial flow, iterations > 0: no fill -- the first matrix update takes the flow as zero and the iterations write every pixel
    bool clear_flow;                 // coarsest, no initial flow, no iterations: the planes are cleared
    bool zoom_fused;                 // finer level of another size: the coarser flow is sampled inside the first matrix update (:412-417, same arithmetic)
    bool zoom;                       // finer level otherwise: a resize (or copy) launch
    // how its launches are cut
    GroupPlan groups;                // pairs per launch chain (fb_groups.h); groups.pairs == B: the whole batch
    bool staged;                     // groups.pairs < B: the zoom and the frames' side (blur, expansion) run group by group too
    bool pair_it;                    // two iterations per launch (levels whose 64 x 4 grid underfills the device)
};
struct FbPlan {
    int levels;                      // the level loop runs k = levels .. 0
    bool fuse_small;                 // launch-latency-bound call: the few-launch forms (resize sampled in the consumers, merge written by the last iteration)
    std::vector<FbLevel> lv;         // lv[k]
};

inline int fb_round(double v) { return (int)std::lrint(v); }   // cvRound
inline int fb_div_up(int a, int b) { return (a + b - 1) / b; }

inline FbPlan fb_make_plan(const FbShape &S, const FbKnobs &K)
{
    FbPlan p;
    // crop unnecessary levels, farneback.cpp:330-340 (MIN_SIZE = 32, :54)
    double scale = 1;
    p.levels = 0;
    for (; p.levels < S.num_levels; p.levels++) {
        scale *= S.pyr_scale;
        if (S.W * scale < 32 || S.H * scale < 32) break;
    }
    p.fuse_small = K.fuse >= 0 ? K.fuse != 0 : (long long)S.W * S.H * S.B <= 1500000;
    p.lv.resize(p.levels + 1);
    // fastPyramids: the levels are the pyrDown chain's sizes (:346-359), not the rounded scales
    std::vector<int> fw(p.levels + 1), fh(p.levels + 1);
    fw[0] = S.W; fh[0] = S.H;
    for (int i = 1; i <= p.levels; ++i) { fw[i] = (fw[i - 1] + 1) / 2; fh[i] = (fh[i - 1] + 1) / 2; }
    for (int k = p.levels; k >= 0; k--) {
        FbLevel &L = p.lv[k];
        scale = 1;
        for (int i = 0; i < k; i++) scale *= S.pyr_scale;
        L.scale = scale;
        L.sigma = (1. / scale - 1) * 0.5;
        int smooth = fb_round(L.sigma * 5) | 1;
        L.smooth = smooth > 3 ? smooth : 3;
        L.w = S.fast_pyramids ? fw[k] : fb_round(S.W * scale);
        L.h = S.fast_pyramids ? fh[k] : fb_round(S.H * scale);
        L.ld = (L.w + 63) / 64 * 64;
        L.coarsest = k == p.levels;
        L.init_resize = L.coarsest && S.use_init && k > 0;
        L.zero_flow = L.coarsest && !S.use_init && S.num_iters > 0;
        L.clear_flow = L.coarsest && !S.use_init && S.num_iters <= 0;
        const bool same_size = !L.coarsest && p.lv[k + 1].w == L.w && p.lv[k + 1].h == L.h;
        L.zoom_fused = !L.coarsest && !same_size && K.fuse != 0;
        L.zoom = !L.coarsest && !L.zoom_fused;
        L.groups = GroupPlan{S.B, 1, false};
        if (!p.fuse_small) L.groups = plan_pair_groups(S.B, 22LL * (long long)L.ld * L.h * (long long)sizeof(float), K.group_mb, K.chains);
        L.staged = L.groups.pairs < S.B;
        L.pair_it = p.fuse_small && K.iterate2_ok &&
                    (K.pair >= 0 ? K.pair != 0 : (long long)fb_div_up(L.w, 64) * fb_div_up(L.h, 4) * S.B <= 2LL * (K.simds / 4));
    }
    return p;
}

}  // namespace fb
}  // namespace mi
