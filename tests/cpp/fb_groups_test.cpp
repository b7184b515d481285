// Pair-group plan of the batched Farneback level loop (csrc/fb_groups.h): plain C++, no device.
#include "fb_groups.h"
#include <cstdio>
#include <cstdlib>

using mi::fb::GroupPlan;
using mi::fb::plan_pair_groups;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main()
{
    const long long MB = 1 << 20;
    const long long vga = 22LL * 640 * 480 * 4, qvga = 22LL * 320 * 240 * 4, hd = 22LL * 1920 * 1080 * 4;
    // the bench's shape: 32 pairs of 640 x 480, 240 MB, two chains -> 8 groups of 4
    GroupPlan p = plan_pair_groups(32, vga, 240, 2);
    CHECK(p.pairs == 4 && p.groups == 8 && p.two);
    // one chain: 240 MB / 25.8 MB = 9 pairs per group
    p = plan_pair_groups(32, vga, 240, 1);
    CHECK(p.pairs == 9 && p.groups == 4 && !p.two);
    // the level above (320 x 240): everything fits within an eighth -> whole batch
    p = plan_pair_groups(32, qvga, 200, 2);
    CHECK(p.pairs == 32 && p.groups == 1 && !p.two);
    // 1080p: one pair per budget -> no groups
    p = plan_pair_groups(8, hd, 240, 2);
    CHECK(p.pairs == 8 && p.groups == 1 && !p.two);
    // switched off, single pair, degenerate sizes
    CHECK(plan_pair_groups(32, vga, 0, 2).pairs == 32);
    CHECK(plan_pair_groups(1, vga, 240, 2).pairs == 1);
    CHECK(plan_pair_groups(32, 0, 240, 2).pairs == 32);
    // properties over a sweep: groups cover the batch, never exceed the budget in flight, even count on two chains
    for (int B = 2; B <= 130; B += 3)
        for (int mb = 20; mb <= 600; mb += 35)
            for (long long pp : {3 * MB, 7 * MB, 26 * MB, 60 * MB, 190 * MB})
                for (int ch = 1; ch <= 2; ++ch) {
                    const GroupPlan q = plan_pair_groups(B, pp, mb, ch);
                    CHECK(q.pairs >= 1 && q.pairs <= B);
                    CHECK(q.groups == (B + q.pairs - 1) / q.pairs);
                    CHECK((long long)q.groups * q.pairs >= B && (long long)(q.groups - 1) * q.pairs < B);
                    if (q.pairs < B) {
                        CHECK((long long)q.pairs * (q.two ? 2 : 1) * pp <= (long long)mb * MB);
                        CHECK(q.two == (ch == 2));
                        // an even number of groups was planned; rounding the group size up can only drop whole groups from the end
                        if (q.two) CHECK(q.groups >= 2);
                    } else
                        CHECK(!q.two && q.groups == 1);
                }
    if (fails) { std::printf("fb_groups_test: %d failure(s)\n", fails); return 1; }
    std::printf("fb_groups_test: ok\n");
    return 0;
}
