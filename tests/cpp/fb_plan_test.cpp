// The plan of a mi_farneback_calc_batch call (csrc/fb_plan.h): levels, geometry and the form of each level's launches.  Plain C++, no device.
// Geometry follows FarnebackOpticalFlowImpl::calcImpl (cudaoptflow/src/farneback.cpp:330-340 level crop, :374-395 per-level sizes).
#include "fb_plan.h"
#include <cstdio>

using namespace mi::fb;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static FbKnobs knobs(int fuse = -1, int pair = -1, int mb = 240, int chains = 2, bool it2 = true)
{
    FbKnobs k;
    k.fuse = fuse; k.pair = pair; k.group_mb = mb; k.chains = chains; k.simds = 1024; k.iterate2_ok = it2;
    return k;
}

int main()
{
    {   // the class defaults on one 640 x 480 pair (BASELINE configs[0]'s shape): numLevels 5, pyrScale 0.5 -> 40 x 30 is below MIN_SIZE 32
        const FbShape s = {640, 480, 1, 5, 0.5, false, 10, false};
        const FbPlan p = fb_make_plan(s, knobs());
        CHECK(p.levels == 3 && (int)p.lv.size() == 4 && p.fuse_small);
        const int w[4] = {640, 320, 160, 80}, h[4] = {480, 240, 120, 60};
        for (int k = 0; k <= 3; ++k) {
            CHECK(p.lv[k].w == w[k] && p.lv[k].h == h[k] && p.lv[k].ld % 64 == 0 && p.lv[k].ld >= w[k]);
            CHECK(p.lv[k].coarsest == (k == 3));
            CHECK(p.lv[k].groups.pairs == 1 && !p.lv[k].staged);          // a single pair is never cut into groups
            CHECK(p.lv[k].zoom_fused == (k < 3) && !p.lv[k].zoom);          // every finer level has another size: sampled in the first update
        }
        // sigma = (1 / scale - 1) / 2, smoothSize = max(cvRound(5 sigma) | 1, 3)  (:381-384)
        CHECK(p.lv[0].sigma == 0.0 && p.lv[0].smooth == 3);
        CHECK(p.lv[1].sigma == 0.5 && p.lv[1].smooth == 3);
        CHECK(p.lv[2].sigma == 1.5 && p.lv[2].smooth == 9);               // cvRound(7.5) = 8 (ties to even), | 1 = 9
        CHECK(p.lv[3].sigma == 3.5 && p.lv[3].smooth == 19);              // cvRound(17.5) = 18, | 1 = 19
        // no initial flow, ten iterations: the coarsest flow is never filled
        CHECK(p.lv[3].zero_flow && !p.lv[3].clear_flow && !p.lv[3].init_resize);
        for (int k = 0; k < 3; ++k) CHECK(!p.lv[k].zero_flow && !p.lv[k].clear_flow && !p.lv[k].init_resize);
        // the coarse levels underfill 256 CUs with 64 x 4 tiles: two iterations per launch there, not at level 0 (10 x 120 tiles)
        CHECK(p.lv[3].pair_it && p.lv[2].pair_it && !p.lv[0].pair_it);
    }
    {   // the initial flow of the caller, zero iterations, forced switches
        const FbShape s = {640, 480, 1, 5, 0.5, false, 0, true};
        FbPlan p = fb_make_plan(s, knobs());
        CHECK(p.lv[3].init_resize && !p.lv[3].zero_flow && !p.lv[3].clear_flow);
        const FbShape s0 = {640, 480, 1, 0, 0.5, false, 0, true};         // numLevels 0: level 0 is the coarsest and IS the initial flow
        p = fb_make_plan(s0, knobs());
        CHECK(p.levels == 0 && !p.lv[0].init_resize && !p.lv[0].clear_flow && !p.lv[0].zoom && !p.lv[0].zoom_fused);
        const FbShape s1 = {640, 480, 1, 5, 0.5, false, 0, false};
        p = fb_make_plan(s1, knobs());
        CHECK(p.lv[3].clear_flow && !p.lv[3].zero_flow);
        p = fb_make_plan(s1, knobs(0, 0));                                  // MIFLOW_FB_FUSE=0, MIFLOW_FB_PAIR=0: the plain forms
        CHECK(!p.fuse_small);
        for (int k = 0; k < 3; ++k) CHECK(p.lv[k].zoom && !p.lv[k].zoom_fused && !p.lv[k].pair_it);
        p = fb_make_plan(s1, knobs(-1, 1, 240, 2, false));                  // no two-iteration kernel for this window: never paired
        for (int k = 0; k <= 3; ++k) CHECK(!p.lv[k].pair_it);
    }
    {   // the bench's batch: 32 pairs of 640 x 480 -> not a small call; level 0 in 8 groups of 4 on two chains, the coarse levels whole
        const FbShape s = {640, 480, 32, 5, 0.5, false, 10, false};
        FbPlan p = fb_make_plan(s, knobs());
        CHECK(!p.fuse_small);
        CHECK(p.lv[0].groups.pairs == 4 && p.lv[0].groups.groups == 8 && p.lv[0].groups.two && p.lv[0].staged);
        CHECK(p.lv[2].groups.pairs == 32 && !p.lv[2].staged && !p.lv[2].groups.two);
        for (int k = 0; k <= 3; ++k) CHECK(!p.lv[k].pair_it);               // pairing is a small-call form
        for (int k = 0; k < 3; ++k) CHECK(p.lv[k].zoom_fused);               // the zoom inside the first update is for every call (round 5)
        p = fb_make_plan(s, knobs(-1, -1, 240, 1));                          // while the caller's stream is captured: one chain
        CHECK(p.lv[0].groups.pairs == 9 && !p.lv[0].groups.two);
        p = fb_make_plan(s, knobs(-1, -1, 0, 2));                            // MIFLOW_FB_GROUP_MB=0: no groups
        for (int k = 0; k <= 3; ++k) CHECK(p.lv[k].groups.pairs == 32 && !p.lv[k].staged);
    }
    {   // fastPyramids: the pyrDown chain's sizes ((w + 1) / 2), odd sizes; a scale that repeats a size -> copy, not a fused zoom
        const FbShape s = {641, 483, 2, 3, 0.5, true, 3, false};
        FbPlan p = fb_make_plan(s, knobs());
        CHECK(p.levels == 3 && p.lv[1].w == 321 && p.lv[1].h == 242 && p.lv[2].w == 161 && p.lv[2].h == 121 && p.lv[3].w == 81 && p.lv[3].h == 61);
        const FbShape t = {641, 483, 2, 3, 0.5, false, 3, false};          // without: cvRound of the scaled size
        p = fb_make_plan(t, knobs());
        CHECK(p.lv[1].w == 320 && p.lv[1].h == 242 && p.lv[3].w == 80 && p.lv[3].h == 60);
        const FbShape u = {100, 100, 1, 3, 0.999, false, 3, false};        // 100 * 0.999^k rounds to 100 for every level: same size
        p = fb_make_plan(u, knobs());
        CHECK(p.levels == 3);
        for (int k = 0; k < 3; ++k) CHECK(p.lv[k].w == 100 && p.lv[k].zoom && !p.lv[k].zoom_fused);
    }
    // invariants over a sweep
    for (int W : {33, 64, 200, 641, 1920})
        for (int H : {32, 100, 483, 1080})
            for (int B : {1, 3, 32, 70})
                for (int nl : {0, 1, 5, 9})
                    for (double ps : {0.5, 0.8})
                        for (int init = 0; init < 2; ++init) {
                            const FbShape s = {W, H, B, nl, ps, false, 2, init != 0};
                            const FbPlan p = fb_make_plan(s, knobs());
                            CHECK(p.levels >= 0 && p.levels <= nl && (int)p.lv.size() == p.levels + 1);
                            CHECK(p.lv[0].w == W && p.lv[0].h == H);
                            int ncoarsest = 0;
                            for (int k = 0; k <= p.levels; ++k) {
                                const FbLevel &L = p.lv[k];
                                ncoarsest += L.coarsest;
                                CHECK(L.w >= 1 && L.h >= 1 && (L.smooth & 1) && L.smooth >= 3);
                                CHECK((int)L.init_resize + (int)L.zero_flow + (int)L.clear_flow + (int)L.zoom + (int)L.zoom_fused <= 1);   // one way to start
                                CHECK(L.coarsest || L.zoom || L.zoom_fused);
                                CHECK(L.groups.pairs >= 1 && L.groups.pairs <= B && L.staged == (L.groups.pairs < B));
                                CHECK(!(L.pair_it && !p.fuse_small));
                                if (k > 0) CHECK(p.lv[k].w <= p.lv[k - 1].w && p.lv[k].h <= p.lv[k - 1].h);
                            }
                            CHECK(ncoarsest == 1 && p.lv[p.levels].coarsest);
                            if (p.levels < nl) {   // the crop stopped at the first level below MIN_SIZE
                                double sc = 1;
                                for (int i = 0; i <= p.levels; ++i) sc *= ps;
                                CHECK(W * sc < 32 || H * sc < 32);
                            }
                        }
    if (fails) { std::printf("fb_plan_test: %d FAILED\n", fails); return 1; }
    std::printf("fb_plan_test: ok\n");
    return 0;
}
