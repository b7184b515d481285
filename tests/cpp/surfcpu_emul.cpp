// Host build of opencv_contrib_amd/csrc/surfcpu_dev.h: the phases of the HIP kernels run as loops over the thread index, so the
// kernel logic is checked bit for bit against oracle/surfcpu_ref.c on the CPU (tests/test_surfcpu_emulation.py).  Test code only.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "surfcpu_dev.h"

using namespace mi::surfcpu;

// kp: n rows {x, y, size, angle, ...} with row stride 7 floats (the oracle's layout); sum: (rows + 1) x (cols + 1) ints, dense
extern "C" int emul_orientation(const int *sum, int rows, int cols, float *kp, int n, int upright)
{
    Tables T;
    make_tables(T);
    OriShared *sm = new OriShared();
    const Integral S = {sum, cols + 1};
    for (int k = 0; k < n; ++k) {
        float *K = kp + (size_t)k * 7;
        orientation_block(S, rows, cols, K[0], K[1], K[2], upright, T, *sm, &K[3], &K[2]);
    }
    delete sm;
    return 0;
}

extern "C" int emul_descriptors(const unsigned char *img, int rows, int cols, const float *kp, int n, int extended, int upright, float *desc)
{
    Tables T;
    make_tables(T);
    DescShared *sm = new DescShared();
    const Image I = {img, cols, rows, cols};
    const int dsize = extended ? 128 : 64;
    int bad = 0;
    for (int k = 0; k < n; ++k) {
        const float *K = kp + (size_t)k * 7;
        if (!(K[2] > 0)) { memset(desc + (size_t)k * dsize, 0, sizeof(float) * dsize); continue; }
        descriptor_block(I, K[0], K[1], K[2], K[3], upright, extended, T, *sm, desc + (size_t)k * dsize);
        bad += sm->bad;
    }
    delete sm;
    return bad;
}
