// Pair groups of a batched Farneback level (farneback_api.cpp, round 5): how many pairs a launch of the level covers and whether
// the groups run as two chains on two streams.  Pure host arithmetic, no HIP types: tests/cpp/fb_groups_test.cpp compiles it alone.
#pragma once
#include <algorithm>

namespace mi {
namespace fb {

struct GroupPlan {
    int pairs;      // pairs per group (== batch: one launch chain over the whole batch)
    int groups;     // ceil(batch / pairs)
    bool two;       // groups alternate between the caller's stream and the handle's internal one
};

// batch pairs of per_pair_bytes each (the 22 level-sized planes an iteration streams); budget_mb = the share of the last-level cache the
// groups in flight may use (MIFLOW_FB_GROUP_MB; <= 0: no groups); chains = 1 | 2 (streams available to run groups side by side).
inline GroupPlan plan_pair_groups(int batch, long long per_pair_bytes, int budget_mb, int chains)
{
    GroupPlan p = {batch, 1, false};
    if (batch <= 1 || budget_mb <= 0 || per_pair_bytes <= 0) return p;
    long long G = std::max(1LL, std::min((long long)batch, ((long long)budget_mb << 20) / per_pair_bytes));
    if (G < 2) return p;                                   // not even two pairs' planes fit: nothing would stay cached
    if ((long long)batch * 8 <= G * 9) return p;           // a budget missed by an eighth: no one- or two-pair tail group
    if (chains >= 2) {
        G = std::max(1LL, G / 2);                          // two groups are in flight
        const long long n = ((batch + G - 1) / G + 1) & ~1LL;   // an even number of (almost) equal groups: both chains end together
        G = (batch + n - 1) / n;
        p.two = true;
    }
    p.pairs = (int)G;
    p.groups = (int)((batch + G - 1) / G);
    return p;
}

}  // namespace fb
}  // namespace mi
