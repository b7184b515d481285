// cv::cuda::StereoSGM for gfx950: census transform -> 4 / 8 path aggregation -> winner takes all (left with
// sub-pixel + uniqueness, right) -> 3x3 median -> left-right check + range correction.  Replaces
// cudastereo/src/cuda/stereosgm.cu:366-2078 behind the C-ABI; all integer, bit-exact against oracle/sgm_ref.c, whose census,
// path aggregation and left winner-takes-all are the reference's own CPU test twins (cudastereo/test/test_sgm_funcs.cpp).
//
// Construction (not a translation of the reference's sub-group dynamic programming):
//   * path aggregation: ONE launch for all paths; a wave owns one scan line of one path, LANE = DISPARITY (V = D/64 consecutive
//     disparities per lane), so the per-pixel minimum over disparities is a 6-step DPP max-reduction on complemented costs,
//     the d+-1 neighbours are DPP wave shifts, and the 64 x V cost bytes of a pixel leave as one contiguous store;
//   * winner takes all: a wave owns an image row, lane = disparity; the right-image minimum over the anti-diagonal
//     cost(x = p + d, d) is kept in a lane-rotating register file fed by one ds_bpermute per step.
#include "mi_common.h"
#include <cstdint>

struct mi_stereosgm {
    mi_stereosgm_params P;
    void *buf = nullptr;
    size_t buf_bytes = 0;
};

namespace mi {
namespace sgm {

// ------------------------------------------------------------------ census 9x7 (test_sgm_funcs.cpp:131-151)
template <typename T>
__global__ __launch_bounds__(256) void k_census(const unsigned char *src, size_t sstep, int *dst, size_t dstep, int rows, int cols)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    int value = 0;
    if (y >= 3 && y < rows - 3 && x >= 4 && x < cols - 4) {
        for (int dy = -3; dy <= 0; ++dy) {
            const T *ra = reinterpret_cast<const T *>(src + (size_t)(y + dy) * sstep);
            const T *rb = reinterpret_cast<const T *>(src + (size_t)(y - dy) * sstep);
            const int dxe = dy == 0 ? -1 : 4;
            for (int dx = -4; dx <= dxe; ++dx) value = (int)(((unsigned)value << 1) | (unsigned)(ra[x + dx] > rb[x - dx]));
        }
    }
    reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(dst) + (size_t)y * dstep)[x] = value;
}

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) { return ~wave_max_u32(~v); }

// ------------------------------------------------------------------ path aggregation (test_sgm_funcs.cpp:225-260)
struct PathArgs {
    const int *left, *right;     // census images, dense rows of `cols` ints
    unsigned char *dst;          // [path][pixel][disparity]
    int rows, cols, min_disp, p1, p2, npaths;
    int dx[8], dy[8];
    int line0[9];                // prefix sums of the number of scan lines per path
};

template <int D>
__global__ __launch_bounds__(256) void k_paths(PathArgs A)
{
    constexpr int V = D / 64;
    const int lane = threadIdx.x & 63;
    const int gl = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));   // global line index, pinned wave-uniform (scalar loop control)
    if (gl >= A.line0[A.npaths]) return;
    int q = 0;
    while (gl >= A.line0[q + 1]) ++q;
    const int L = gl - A.line0[q];
    const int dx = A.dx[q], dy = A.dy[q], W = A.cols, H = A.rows;
    // first pixel of scan line L: a pixel whose predecessor (i - dy, j - dx) is outside the image
    int i, j;
    if (dx == 0) { j = L; i = dy > 0 ? 0 : H - 1; }
    else if (dy == 0) { i = L; j = dx > 0 ? 0 : W - 1; }
    else if (L < W) { j = L; i = dy > 0 ? 0 : H - 1; }
    else { const int s = L - W + 1; j = dx > 0 ? 0 : W - 1; i = dy > 0 ? s : H - 1 - s; }
    unsigned char *out = A.dst + (size_t)q * W * H * D;
    int prev[V];
#pragma unroll
    for (int v = 0; v < V; ++v) prev[v] = 0;
    const int BIG = 1 << 20;
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    // census values of a pixel do not depend on the recurrence: they are fetched ahead of it (the loop is otherwise a chain of
    // dependent load -> min -> store steps with nothing to overlap the load latency inside a wave)
    auto fetch = [&](int ii, int jj, unsigned &l_, unsigned (&r_)[V]) {
        const bool in = ii >= 0 && ii < H && jj >= 0 && jj < W;
        const int ic = min(max(ii, 0), H - 1), jc = min(max(jj, 0), W - 1);
        // read through the VECTOR memory path (vz is an opaque zero VGPR): as a scalar load the compiler must wait lgkmcnt(0) at its
        // use, which also waits for the scalar loads just issued for later steps and defeats the prefetch
        l_ = in ? (unsigned)A.left[(size_t)ic * W + jc + vz] : 0u;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int k = lane * V + v;
            const int jr = jj - k - A.min_disp;
            const bool ok = in && !(k + A.min_disp > jj) && jr < W;   // (jr >= W: negative minDisparity)
            r_[v] = ok ? (unsigned)A.right[(size_t)ic * W + min(max(jr, 0), W - 1)] : 0u;
        }
    };
    // PD steps ahead: a vertical or diagonal step touches a new image row (new cache lines), whose latency is several step times
    constexpr int PD = 4;
    unsigned lq[PD], rq[PD][V];
#pragma unroll
    for (int s = 0; s < PD; ++s) fetch(i + s * dy, j + s * dx, lq[s], rq[s]);
    while (i >= 0 && i < H && j >= 0 && j < W) {
#pragma unroll
        for (int s = 0; s < PD; ++s) {
            if (!(i >= 0 && i < H && j >= 0 && j < W)) break;   // wave-uniform
            const size_t pix = (size_t)i * W + j;
            const unsigned l = lq[s];
            unsigned rr[V];
#pragma unroll
            for (int v = 0; v < V; ++v) rr[v] = rq[s][v];
            fetch(i + PD * dy, j + PD * dx, lq[s], rq[s]);
            unsigned mn = (unsigned)prev[0];
#pragma unroll
            for (int v = 1; v < V; ++v) mn = min(mn, (unsigned)prev[v]);
            const int m = (int)wave_min_u32(mn);
            // d - 1 / d + 1 neighbours across the lane boundary
            int lo = __builtin_amdgcn_update_dpp(0, prev[V - 1], 0x138, 0xf, 0xf, true);   // wave_shr:1  (lane n <- lane n-1)
            int hi = __builtin_amdgcn_update_dpp(0, prev[0], 0x130, 0xf, 0xf, true);       // wave_shl:1  (lane n <- lane n+1)
            if (lane == 0) lo = BIG;
            if (lane == 63) hi = BIG;
            int cur[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const unsigned r = rr[v];
                const int left_n = v > 0 ? prev[v > 0 ? v - 1 : 0] : lo;
                const int right_n = v + 1 < V ? prev[v + 1 < V ? v + 1 : v] : hi;
                int cost = min(prev[v] - m, A.p2);
                cost = min(cost, left_n - m + A.p1);
                cost = min(cost, right_n - m + A.p1);
                cost += __popc(l ^ r);
                cur[v] = cost & 0xff;   // static_cast<uint8_t>(cost)
            }
            unsigned char *o = out + pix * D + lane * V;
            if (V == 1) o[0] = (unsigned char)cur[0];
            else if (V == 2) *reinterpret_cast<unsigned short *>(o) = (unsigned short)(cur[0] | (cur[V > 1 ? 1 : 0] << 8));
            else *reinterpret_cast<unsigned *>(o) = (unsigned)cur[0] | ((unsigned)cur[V > 1 ? 1 : 0] << 8) | ((unsigned)cur[V > 2 ? 2 : 0] << 16) |
                                                     ((unsigned)cur[V > 3 ? 3 : 0] << 24);
#pragma unroll
            for (int v = 0; v < V; ++v) prev[v] = cur[v];
            i += dy; j += dx;
        }
    }
}

// ------------------------------------------------------------------ winner takes all (test_sgm_funcs.cpp:354-403; stereosgm.cu:1524-1568)
template <int D>
__global__ __launch_bounds__(256) void k_wta(const unsigned char *src, short *left, size_t lstep, short *right, size_t rstep, int width,
                                             int height, int npaths, float uniqueness, int subpixel, int seg)
{
    constexpr int V = D / 64;
    const int lane = threadIdx.x & 63;
    const int y = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));   // wave-uniform
    if (y >= height) return;
    // A row is cut into segments of `seg` pixels (one wave each; a single wave per row leaves the chip at one wave per SIMD
    // with every load latency exposed).  The left disparity is per pixel; a right pixel p needs the left pixels p .. p + D - 1,
    // so a segment also walks the D - 1 pixels after its end, for the right minima only.
    const int xs = blockIdx.y * seg, xe = min(xs + seg, width), x_end = min(xe + D - 1, width);
    const size_t cost_step = (size_t)D * width * height;
    short *lrow = reinterpret_cast<short *>(reinterpret_cast<unsigned char *>(left) + (size_t)y * lstep);
    short *rrow = reinterpret_cast<short *>(reinterpret_cast<unsigned char *>(right) + (size_t)y * rstep);
    unsigned rb[V];
#pragma unroll
    for (int s = 0; s < V; ++s) rb[s] = 0xffffffffu;
    auto load_sums = [&](int xx, unsigned (&sm)[V]) {
#pragma unroll
        for (int v = 0; v < V; ++v) sm[v] = 0;
        const unsigned char *p0 = src + ((size_t)y * width + xx) * D + lane * V;
        for (int q = 0; q < npaths; ++q) {
            const unsigned char *pp = p0 + (size_t)q * cost_step;
            if (V == 1) sm[0] += pp[0];
            else if (V == 2) { const unsigned t = *reinterpret_cast<const unsigned short *>(pp); sm[0] += t & 0xff; sm[V > 1 ? 1 : 0] += t >> 8; }
            else { const unsigned t = *reinterpret_cast<const unsigned *>(pp);
                   sm[0] += t & 0xff; sm[V > 1 ? 1 : 0] += (t >> 8) & 0xff; sm[V > 2 ? 2 : 0] += (t >> 16) & 0xff; sm[V > 3 ? 3 : 0] += t >> 24; }
        }
    };
    unsigned nsum[V];
    load_sums(xs, nsum);
    for (int x0 = xs; x0 < x_end; x0 += V) {
#pragma unroll
        for (int x1 = 0; x1 < V; ++x1) {
            const int x = x0 + x1;
            if (x >= x_end) break;   // wave-uniform
            unsigned sum[V];
#pragma unroll
            for (int v = 0; v < V; ++v) sum[v] = nsum[v];
            load_sums(min(x + 1, width - 1), nsum);   // one pixel ahead (independent of this pixel's reductions)
            unsigned packed[V];
            unsigned bl = 0xffffffffu;
#pragma unroll
            for (int v = 0; v < V; ++v) { packed[v] = (sum[v] << 16) | (unsigned)(lane * V + v); bl = min(bl, packed[v]); }
            const unsigned best = wave_min_u32(bl);
            // right image: holder (lane, s) owns the right pixel p with p mod D == lane * V + s; its candidate of this step is
            // disparity k = (x - p) mod D, held by lane k / V in slot k % V == (x1 - s) mod V
#pragma unroll
            for (int s = 0; s < V; ++s) {
                const int k = (x - (lane * V + s)) & (D - 1);
                const unsigned recv = (unsigned)__shfl((int)packed[(x1 - s + V) % V], k / V);
                rb[s] = min(rb[s], recv);
                if (k == D - 1) {
                    const int p = x - k;
                    if (p >= xs) rrow[p] = (short)(rb[s] & 0xffffu);   // (p < xe by construction; p < xs: started mid-way, other segment's)
                    rb[s] = 0xffffffffu;
                }
            }
            if (x >= xe) continue;   // wave-uniform: the overlap pixels feed the right minima only
            // left image: uniqueness + sub-pixel
            const unsigned best_cost = best >> 16;
            const int best_disp = (int)(best & 0xffffu);
            bool uniq = true;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const bool u1 = (float)sum[v] * uniqueness >= (float)best_cost;
                const bool u2 = abs(lane * V + v - best_disp) <= 1;
                uniq = uniq && (u1 || u2);
            }
            const bool all_uniq = __ballot(uniq) == ~0ull;
            int ans = best_disp;
            if (subpixel) {
                ans <<= 4;   // StereoMatcher::DISP_SHIFT
                if (best_disp > 0 && best_disp < D - 1) {
                    // summed costs at best_disp -+ 1 (wave-uniform lanes / slots)
                    const int kl = best_disp - 1, kr = best_disp + 1;
                    int lc = 0, rc = 0;
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const int a = __shfl((int)sum[v], kl / V), b = __shfl((int)sum[v], kr / V);
                        if (kl % V == v) lc = a;
                        if (kr % V == v) rc = b;
                    }
                    const int numer = lc - rc, denom = lc - 2 * (int)best_cost + rc;
                    ans += ((numer << 4) + denom) / (2 * denom);
                }
            }
            if (lane == 0) lrow[x] = all_uniq ? (short)ans : (short)-1;
        }
    }
#pragma unroll
    for (int s = 0; s < V; ++s) {   // flush: right pixels in (width - D, width)
        const unsigned k0 = (unsigned)(lane * V + s);
        const int p = (int)((((unsigned)width - k0) & ~(unsigned)(D - 1)) + k0);
        if (p >= xs && p < xe && p < width) rrow[p] = (short)(rb[s] & 0xffffu);
    }
}

// ------------------------------------------------------------------ 3x3 median of the 16-bit patterns (stereosgm.cu:1699-1922)
__global__ __launch_bounds__(256) void k_median(const unsigned char *src, size_t sstep, unsigned char *dst, size_t dstep, int rows, int cols,
                                                int emulate_quirks)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const unsigned short *r1 = reinterpret_cast<const unsigned short *>(src + (size_t)y * sstep);
    unsigned out = r1[x];
    if (x >= 1 && x < cols - 1 && y >= 1 && y < rows - 1) {
        const unsigned short *r0 = reinterpret_cast<const unsigned short *>(src + (size_t)(y - 1) * sstep);
        const unsigned short *r2 = reinterpret_cast<const unsigned short *>(src + (size_t)(y + 1) * sstep);
        unsigned b[9] = {r0[x - 1], r0[x], r0[x + 1], r1[x - 1], r1[x], r1[x + 1], r2[x - 1], r2[x], r2[x + 1]};
        // Q-a: the scalar fallback columns of median_kernel_3x3_16u_v2 sort through a uint8_t buffer (stereosgm.cu:1871-1896)
        const int x2 = x & ~1;
        if (emulate_quirks && (x2 == 0 || !(x2 >= 2 && x2 + 3 < cols)))
#pragma unroll
            for (int i = 0; i < 9; ++i) b[i] &= 0xffu;
#define SW(i, j) { const unsigned lo_ = min(b[i], b[j]), hi_ = max(b[i], b[j]); b[i] = lo_; b[j] = hi_; }
#define MN(i, j) { b[i] = min(b[i], b[j]); }
#define MX(i, j) { b[j] = max(b[i], b[j]); }
        SW(0, 1); SW(3, 4); SW(6, 7);
        SW(1, 2); SW(4, 5); SW(7, 8);
        SW(0, 1); SW(3, 4); SW(6, 7);
        MX(0, 3); MX(3, 6);
        SW(1, 4); MN(4, 7); MX(1, 4);
        MN(5, 8); MN(2, 5);
        SW(2, 4); MN(4, 6); MX(2, 4);
#undef SW
#undef MN
#undef MX
        out = b[4];
    }
    reinterpret_cast<unsigned short *>(dst + (size_t)y * dstep)[x] = (unsigned short)out;
}

// ------------------------------------------------------------------ left-right check + range correction (stereosgm.cu:1959-2055)
template <typename T>
__global__ __launch_bounds__(256) void k_check_range(unsigned char *ldisp, size_t lstep, const unsigned char *rdisp, size_t rstep,
                                                     const unsigned char *img, size_t istep, int rows, int cols, int subpixel, int min_disp,
                                                     int emulate_quirks)
{
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= cols || i >= rows) return;
    unsigned short *lp = reinterpret_cast<unsigned short *>(ldisp + (size_t)i * lstep) + j;
    unsigned d16 = *lp;
    // Q-b: the reference launches width/16 x height/16 blocks of 16 x 16 (stereosgm.cu:1985)
    const bool checked = !emulate_quirks || (j < (cols / 16) * 16 && i < (rows / 16) * 16);
    if (checked) {
        const unsigned mask = reinterpret_cast<const T *>(img + (size_t)i * istep)[j];
        int d = (int)d16;
        if (subpixel) d >>= 4;
        const int k = j - d;
        bool bad = mask == 0 || d16 == 0xffffu;
        if (!bad && k >= 0 && k < cols) {
            const int rd = reinterpret_cast<const unsigned short *>(rdisp + (size_t)i * rstep)[k];
            bad = abs(rd - d) > 1;
        }
        if (bad) d16 = 0xffffu;
    }
    const int scale = subpixel ? 16 : 1;
    d16 = d16 == 0xffffu ? (unsigned)((min_disp - 1) * scale) : d16 + (unsigned)(min_disp * scale);
    *lp = (unsigned short)d16;
}

static void fill_paths(PathArgs &A, int npaths)
{
    static const int DX[8] = {0, 0, 1, -1, 1, -1, -1, 1}, DY[8] = {1, -1, 0, 0, 1, 1, -1, -1};   // PathAggregation::operator(), :1352-1362
    A.npaths = npaths;
    A.line0[0] = 0;
    for (int q = 0; q < 8; ++q) {
        A.dx[q] = DX[q]; A.dy[q] = DY[q];
        if (q < npaths) A.line0[q + 1] = A.line0[q] + (DX[q] == 0 ? A.cols : DY[q] == 0 ? A.rows : A.cols + A.rows - 1);
        else A.line0[q + 1] = A.line0[q];
    }
}

static int launch_paths(const PathArgs &A, int D, hipStream_t st)
{
    const dim3 grid(div_up(A.line0[A.npaths], 4));
    if (D == 64) hipLaunchKernelGGL((k_paths<64>), grid, dim3(256), 0, st, A);
    else if (D == 128) hipLaunchKernelGGL((k_paths<128>), grid, dim3(256), 0, st, A);
    else hipLaunchKernelGGL((k_paths<256>), grid, dim3(256), 0, st, A);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

static int launch_wta(const unsigned char *src, short *l, size_t ls, short *r, size_t rs, int w, int h, int D, int np, float uniq, int subpixel,
                      hipStream_t st)
{
    const int seg = D <= 128 ? 256 : 512;   // multiple of V; (seg + D - 1) / seg = 1.25 .. 1.5 of redundant walking buys 4 - 8 waves per SIMD
    const dim3 grid(div_up(h, 4), div_up(w, seg));
    if (D == 64) hipLaunchKernelGGL((k_wta<64>), grid, dim3(256), 0, st, src, l, ls, r, rs, w, h, np, uniq, subpixel, seg);
    else if (D == 128) hipLaunchKernelGGL((k_wta<128>), grid, dim3(256), 0, st, src, l, ls, r, rs, w, h, np, uniq, subpixel, seg);
    else hipLaunchKernelGGL((k_wta<256>), grid, dim3(256), 0, st, src, l, ls, r, rs, w, h, np, uniq, subpixel, seg);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

static int census(const mi_mat *src, int *dst, size_t dstep, hipStream_t st)
{
    const dim3 grid(div_up(src->cols, 64), div_up(src->rows, 4));
    if (src->type == MI_8UC1)
        hipLaunchKernelGGL((k_census<unsigned char>), grid, dim3(256), 0, st, (const unsigned char *)src->data, src->step, dst, dstep, src->rows, src->cols);
    else
        hipLaunchKernelGGL((k_census<unsigned short>), grid, dim3(256), 0, st, (const unsigned char *)src->data, src->step, dst, dstep, src->rows, src->cols);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

static int have_device()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device available: the miflow product path has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    return MI_OK;
}

}  // namespace sgm
}  // namespace mi

using namespace mi;

extern "C" {

void mi_stereosgm_default_params(mi_stereosgm_params *p)
{
    if (!p) return;
    // createStereoSGM(minDisparity = 0, numDisparities = 128, P1 = 10, P2 = 120, uniquenessRatio = 5, mode = MODE_HH4), cudastereo.hpp
    p->min_disparity = 0; p->num_disparities = 128; p->P1 = 10; p->P2 = 120; p->uniqueness_ratio = 5; p->mode = MI_SGM_MODE_HH4;
    p->emulate_cuda_quirks = 1;
}

int mi_stereosgm_create(const mi_stereosgm_params *p, mi_stereosgm **out)
{
    MI_REQUIRE(out, MI_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int rc = sgm::have_device();
    if (rc) return rc;
    mi_stereosgm *h = new mi_stereosgm();
    if (p) h->P = *p; else mi_stereosgm_default_params(&h->P);
    *out = h;
    return MI_OK;
}

int mi_stereosgm_set_params(mi_stereosgm *h, const mi_stereosgm_params *p)
{
    MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument");
    h->P = *p;   // validated at compute(), like the reference (stereosgm.cpp:102-131)
    return MI_OK;
}

int mi_stereosgm_get_params(const mi_stereosgm *h, mi_stereosgm_params *p)
{
    MI_REQUIRE(h && p, MI_ERR_BAD_ARG, "null argument");
    *p = h->P;
    return MI_OK;
}

void mi_stereosgm_destroy(mi_stereosgm *h)
{
    if (!h) return;
    if (h->buf) (void)hipFree(h->buf);
    delete h;
}

int mi_stereosgm_compute(mi_stereosgm *h, const mi_mat *left, const mi_mat *right, mi_mat *disp, void *stream)
{
    MI_REQUIRE(h && left && right && disp && left->data && right->data && disp->data, MI_ERR_BAD_ARG, "null argument");
    const mi_stereosgm_params &P = h->P;
    MI_REQUIRE(P.mode == MI_SGM_MODE_HH || P.mode == MI_SGM_MODE_HH4, MI_ERR_BAD_ARG, "Unsupported mode");   // stereosgm.cpp:102-105
    MI_REQUIRE(left->type == MI_8UC1 || left->type == MI_16UC1, MI_ERR_BAD_TYPE, "left.type() == CV_8UC1 || left.type() == CV_16UC1");
    MI_REQUIRE(right->type == left->type && right->rows == left->rows && right->cols == left->cols, MI_ERR_BAD_SIZE,
               "size == right.size() && left.type() == right.type()");
    MI_REQUIRE(P.num_disparities == 64 || P.num_disparities == 128 || P.num_disparities == 256, MI_ERR_BAD_ARG,
               "Unsupported num of disparities");                                                          // stereosgm.cpp:138
    MI_REQUIRE(disp->type == MI_16SC1 && disp->rows == left->rows && disp->cols == left->cols, MI_ERR_BAD_SIZE,
               "disparity must be CV_16SC1 of the image size");
    hipStream_t st = (hipStream_t)stream;
    const int rows = left->rows, cols = left->cols, D = P.num_disparities, np = P.mode == MI_SGM_MODE_HH4 ? 4 : 8;
    const size_t n = (size_t)rows * cols;
    // scratch: census L/R (int32), aggregated costs [path][pixel][d] (u8), three int16 maps
    const size_t off_cr = n * 4, off_agg = off_cr + n * 4, off_lt = off_agg + n * D * np;
    const size_t off_rt = off_lt + ((n * 2 + 255) / 256) * 256, off_rm = off_rt + ((n * 2 + 255) / 256) * 256;
    const size_t need = off_rm + ((n * 2 + 255) / 256) * 256;
    if (h->buf_bytes < need) {
        if (h->buf) { (void)hipFree(h->buf); h->buf = nullptr; h->buf_bytes = 0; }
        MI_HIP_TRY(hipMalloc(&h->buf, need));
        h->buf_bytes = need;
    }
    unsigned char *base = (unsigned char *)h->buf;
    int *cl = (int *)base, *cr = (int *)(base + off_cr);
    unsigned char *agg = base + off_agg;
    short *lt = (short *)(base + off_lt), *rt = (short *)(base + off_rt), *rm = (short *)(base + off_rm);
    int rc;
    if ((rc = sgm::census(left, cl, (size_t)cols * 4, st))) return rc;
    if ((rc = sgm::census(right, cr, (size_t)cols * 4, st))) return rc;
    sgm::PathArgs A;
    A.left = cl; A.right = cr; A.dst = agg; A.rows = rows; A.cols = cols; A.min_disp = P.min_disparity; A.p1 = P.P1; A.p2 = P.P2;
    sgm::fill_paths(A, np);
    if ((rc = sgm::launch_paths(A, D, st))) return rc;
    if ((rc = sgm::launch_wta(agg, lt, (size_t)cols * 2, rt, (size_t)cols * 2, cols, rows, D, np, (float)(100 - P.uniqueness_ratio) / 100, 1, st)))
        return rc;
    const dim3 grid(div_up(cols, 64), div_up(rows, 4));
    hipLaunchKernelGGL(sgm::k_median, grid, dim3(256), 0, st, (const unsigned char *)lt, (size_t)cols * 2, (unsigned char *)disp->data, disp->step,
                       rows, cols, P.emulate_cuda_quirks);
    hipLaunchKernelGGL(sgm::k_median, grid, dim3(256), 0, st, (const unsigned char *)rt, (size_t)cols * 2, (unsigned char *)rm, (size_t)cols * 2,
                       rows, cols, P.emulate_cuda_quirks);
    if (left->type == MI_8UC1)
        hipLaunchKernelGGL((sgm::k_check_range<unsigned char>), grid, dim3(256), 0, st, (unsigned char *)disp->data, disp->step,
                           (const unsigned char *)rm, (size_t)cols * 2, (const unsigned char *)left->data, left->step, rows, cols, 1,
                           P.min_disparity, P.emulate_cuda_quirks);
    else
        hipLaunchKernelGGL((sgm::k_check_range<unsigned short>), grid, dim3(256), 0, st, (unsigned char *)disp->data, disp->step,
                           (const unsigned char *)rm, (size_t)cols * 2, (const unsigned char *)left->data, left->step, rows, cols, 1,
                           P.min_disparity, P.emulate_cuda_quirks);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

// ---- stage level (the granularity of the reference's own unit tests, cudastereo/test/test_sgm_funcs.cpp)
int mi_sgm_census(const mi_mat *src, mi_mat *dst, void *stream)
{
    MI_REQUIRE(src && dst && src->data && dst->data, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(src->type == MI_8UC1 || src->type == MI_16UC1, MI_ERR_BAD_TYPE, "src.type() == CV_8UC1 || src.type() == CV_16UC1");
    MI_REQUIRE(dst->type == MI_32SC1 && dst->rows == src->rows && dst->cols == src->cols, MI_ERR_BAD_SIZE, "dst must be CV_32SC1 of src size");
    int rc = sgm::have_device();
    if (rc) return rc;
    return sgm::census(src, (int *)dst->data, dst->step, (hipStream_t)stream);
}

int mi_sgm_aggregate_path(const mi_mat *left_census, const mi_mat *right_census, mi_mat *dst, int num_disparities, int min_disparity, int p1,
                          int p2, int dx, int dy, void *stream)
{
    MI_REQUIRE(left_census && right_census && dst && left_census->data && right_census->data && dst->data, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(left_census->type == MI_32SC1 && right_census->type == MI_32SC1, MI_ERR_BAD_TYPE, "census images must be CV_32SC1");
    const int rows = left_census->rows, cols = left_census->cols;
    MI_REQUIRE(right_census->rows == rows && right_census->cols == cols, MI_ERR_BAD_SIZE, "left.size() == right.size()");
    MI_REQUIRE(left_census->step == (size_t)cols * 4 && right_census->step == (size_t)cols * 4, MI_ERR_BAD_SIZE, "census images must be dense");
    MI_REQUIRE(num_disparities == 64 || num_disparities == 128 || num_disparities == 256, MI_ERR_BAD_ARG, "num_disparities: 64, 128, 256");
    MI_REQUIRE((dx == 0 || dx == 1 || dx == -1) && (dy == 0 || dy == 1 || dy == -1) && (dx || dy), MI_ERR_BAD_ARG, "bad path direction");
    MI_REQUIRE(dst->type == MI_8UC1 && dst->rows == 1 && (long long)dst->cols == (long long)rows * cols * num_disparities, MI_ERR_BAD_SIZE,
               "dst must be CV_8UC1, 1 x (width * height * num_disparities)");
    int rc = sgm::have_device();
    if (rc) return rc;
    sgm::PathArgs A;
    A.left = (const int *)left_census->data; A.right = (const int *)right_census->data; A.dst = (unsigned char *)dst->data;
    A.rows = rows; A.cols = cols; A.min_disp = min_disparity; A.p1 = p1; A.p2 = p2;
    sgm::fill_paths(A, 1);
    A.dx[0] = dx; A.dy[0] = dy;
    A.line0[1] = dx == 0 ? cols : dy == 0 ? rows : cols + rows - 1;
    for (int q = 2; q <= 8; ++q) A.line0[q] = A.line0[1];
    return sgm::launch_paths(A, num_disparities, (hipStream_t)stream);
}

int mi_sgm_winner_takes_all(const mi_mat *src, mi_mat *left, mi_mat *right, int num_disparities, int num_paths, float uniqueness, int subpixel,
                            void *stream)
{
    MI_REQUIRE(src && left && right && src->data && left->data && right->data, MI_ERR_BAD_ARG, "null argument");
    MI_REQUIRE(num_disparities == 64 || num_disparities == 128 || num_disparities == 256, MI_ERR_BAD_ARG, "num_disparities: 64, 128, 256");
    MI_REQUIRE(num_paths == 4 || num_paths == 8, MI_ERR_BAD_ARG, "num_paths: 4 (MODE_HH4) or 8 (MODE_HH)");
    MI_REQUIRE(left->type == MI_16SC1 && right->type == MI_16SC1 && right->rows == left->rows && right->cols == left->cols, MI_ERR_BAD_TYPE,
               "left, right: CV_16SC1 of the same size");
    MI_REQUIRE(src->type == MI_8UC1 && src->rows == 1 &&
                   (long long)src->cols == (long long)left->rows * left->cols * num_disparities * num_paths, MI_ERR_BAD_SIZE,
               "src.rows == 1 && src.cols == width * height * MAX_DISPARITY * num_paths");                  // stereosgm.cu:1590
    int rc = sgm::have_device();
    if (rc) return rc;
    return sgm::launch_wta((const unsigned char *)src->data, (short *)left->data, left->step, (short *)right->data, right->step, left->cols,
                           left->rows, num_disparities, num_paths, uniqueness, subpixel, (hipStream_t)stream);
}

}  // extern "C"
