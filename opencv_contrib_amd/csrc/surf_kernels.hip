// SURF detect + describe (cv::cuda::SURF_CUDA) -- HIP kernels for gfx950 (MI355X, CDNA4), wave64.
//
// Reference: modules/xfeatures2d/src/cuda/surf.cu:122-942 driven by src/surf.cuda.cpp:134-255.  What surf.cu reaches through main-repo
// device headers is written out here as those headers define it (and as oracle/refshim/cudashim restates them to RUN surf.cu on the
// host, tests/test_ref_pin_cuda.py): point-sampled clamp-addressed texture reads = texel (floor(x), floor(y)); core/cuda/filters.hpp's
// LinearFilter / AreaFilter return saturate_cast<uchar> for WinReader (8-bit patch samples) and AreaFilter normalises by the window's
// remainder in its last row / column; core/cuda/utility.hpp's solve3x3 takes 1 / det in double; reduce<N> is the shfl_down tree per
// warp, then over the warps' partials.  (Rounds 1-2 followed the OpenCL twin surf.cl:55-68,413-441,873-952 for these; the CUDA-source
// pin of round 3 showed where the CUDA class differs.)  CDNA has no sampler path worth relying on: plain loads.
// MI355X formulation:
//   * no __constant__ state (the reference loads 9 symbols per call and serialises every call behind a mutex,
//     surf.cuda.cpp:117,371): parameters are kernel arguments, tables live in the handle;
//   * DETERMINISTIC compaction: the reference appends maxima / features with atomicInc (surf.cu:341,476), so order
//     and, on overflow, the surviving subset change run to run.  Here maxima are flagged per 64-sample chunk with
//     a wave ballot, row counts are scanned, and candidates / features are written in (layer, row, column) scan
//     order; the first maxCandidates / maxFeatures survive;
//   * no host round trips inside the octave loop (the reference copies two counters to the host per octave,
//     surf.cuda.cpp:193,206): every stage reads the counts from device memory and bounds itself;
//   * integral image: band-local column prefix in registers + one wave-wide DPP scan per 64 columns;
//   * orientation / descriptor reductions are the reference's 32-lane shfl_down trees (16,8,4,2,1), two per wave64.
// Haar responses accumulate in double like the reference (u32 integral differences do not fit a float mantissa).
#include "surf_dev.h"
#include <cfloat>
#include <algorithm>

namespace mi {
namespace surf {

#define CV_PI_F 3.14159265f

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ int rn(float v) { return __float2int_rn(v); }
__host__ __device__ __forceinline__ int calc_size(int octave, int layer) { return (9 + 6 * layer) << octave; }   // surf.cu:161-173

// ------------------------------------------------------------------ integral image
// pass A: band-local vertical prefix V[y][x] = sum of img[y0..y][x] (u32) and band totals BT[band][x]
__global__ __launch_bounds__(256) void k_int_cols(const unsigned char *img, long long istep, int rows, int cols, int clamp1,
                                                  unsigned *V, int vld, unsigned *BT, int band_rows)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int band = blockIdx.y;
    if (x >= cols) return;
    const int y0 = band * band_rows, y1 = min(y0 + band_rows, rows);
    unsigned acc = 0;
    for (int y = y0; y < y1; ++y) {
        unsigned v = img[(long long)y * istep + x];
        if (clamp1) v = min(v, 1u);   // cuda::min(mask, 1.0), surf.cuda.cpp:167
        acc += v;
        V[(long long)y * vld + x] = acc;
    }
    BT[(long long)band * vld + x] = acc;
}
// pass A2: exclusive prefix of the band totals per column (in place)
__global__ __launch_bounds__(256) void k_int_bands(unsigned *BT, int vld, int cols, int nbands)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= cols) return;
    unsigned acc = 0;
    for (int b = 0; b < nbands; ++b) {
        const unsigned t = BT[(long long)b * vld + x];
        BT[(long long)b * vld + x] = acc;
        acc += t;
    }
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp0(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v)
{
    v += dpp0<0x111, 0xf>(v);   // row_shr:1
    v += dpp0<0x112, 0xf>(v);   // row_shr:2
    v += dpp0<0x114, 0xf>(v);   // row_shr:4
    v += dpp0<0x118, 0xf>(v);   // row_shr:8  -> inclusive scan inside each row of 16
    v += dpp0<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3 += total of the previous row
    v += dpp0<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3 += total of rows 0..1
    return v;
}
// pass B: horizontal scan of (V + band offset) per row -> sum[(y+1)][(x+1)]; first row/column 0
__global__ __launch_bounds__(256) void k_int_rows(const unsigned *V, const unsigned *BT, int vld, int rows, int cols, int band_rows,
                                                  unsigned *sum, int sld)
{
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per row; y == rows: the zero top row
    if (y > rows) return;
    if (y == rows) {
        for (int x = lane; x <= cols; x += 64) sum[x] = 0;
        return;
    }
    const unsigned *bt = BT + (long long)(y / band_rows) * vld;
    unsigned carry = 0;
    if (lane == 0) sum[(long long)(y + 1) * sld] = 0;
    for (int x0 = 0; x0 < cols; x0 += 64) {
        const int x = x0 + lane;
        unsigned v = x < cols ? V[(long long)y * vld + x] + bt[x] : 0u;
        v = wave_incl_scan(v) + carry;
        if (x < cols) sum[(long long)(y + 1) * sld + x + 1] = v;
        carry = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
    }
}
// pass B, wide form (round 3): ONE WORKGROUP of 8 waves per row, each wave a contiguous run of up to kRowCpw 64-column chunks whose loads
// are all in flight together; the waves' totals meet in LDS and each wave adds the sum of the totals to its left before it stores.  The
// one-wave-per-row form walks a 4K row in 60 dependent round trips (27 us per 4K frame); rows wider than 8 x kRowCpw chunks keep it.
constexpr int kRowWaves = 8, kRowCpw = 16;
__global__ __launch_bounds__(64 * kRowWaves) void k_int_rows_wide(const unsigned *V, const unsigned *BT, int vld, int rows, int cols, int band_rows,
                                                                  unsigned *sum, int sld, int cpw)
{
    __shared__ unsigned tot[kRowWaves];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y = blockIdx.x;   // y == rows: the zero top row
    if (y == rows) {
        for (int x = threadIdx.x; x <= cols; x += 64 * kRowWaves) sum[x] = 0;
        return;
    }
    const unsigned *bt = BT + (long long)(y / band_rows) * vld;
    const unsigned *vr = V + (long long)y * vld;
    unsigned v[kRowCpw];
    const int c0 = wv * cpw;
#pragma unroll
    for (int k = 0; k < kRowCpw; ++k) {
        const int x = (c0 + k) * 64 + lane;
        v[k] = (k < cpw && x < cols) ? vr[x] + bt[x] : 0u;
    }
    unsigned carry = 0;
#pragma unroll
    for (int k = 0; k < kRowCpw; ++k) {
        v[k] = wave_incl_scan(v[k]) + carry;
        carry = (unsigned)__builtin_amdgcn_readlane((int)v[k], 63);
    }
    if (lane == 0) tot[wv] = carry;
    __syncthreads();
    unsigned off = 0;
    for (int w = 0; w < wv; ++w) off += tot[w];
    unsigned *out = sum + (long long)(y + 1) * sld;
    if (threadIdx.x == 0) out[0] = 0;
#pragma unroll
    for (int k = 0; k < kRowCpw; ++k) {
        const int x = (c0 + k) * 64 + lane;
        if (k < cpw && x < cols) out[x + 1] = v[k] + off;
    }
}
__global__ void k_dbg_scan(const unsigned *in, unsigned *out) { out[threadIdx.x] = wave_incl_scan(in[threadIdx.x]); }

// ------------------------------------------------------------------ Haar responses
struct SumTex { const unsigned *s; int sld, rows, cols; };   // image size; the integral is (rows+1) x (cols+1)
__device__ __forceinline__ unsigned tex(const SumTex &t, int y, int x)
{
    return t.s[(long long)clampi(y, 0, t.rows) * t.sld + clampi(x, 0, t.cols)];   // surf.cl:55-60 (clamp addressing)
}
// surf.cu:122-152
template <int N>
__device__ __forceinline__ float haar(const SumTex &t, const float (&src)[N][5], int oldSize, int newSize, int y, int x)
{
    const float ratio = (float)newSize / oldSize;
    double d = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int dx1 = rn(ratio * src[k][0]), dy1 = rn(ratio * src[k][1]), dx2 = rn(ratio * src[k][2]), dy2 = rn(ratio * src[k][3]);
        double tt = 0;
        tt += tex(t, y + dy1, x + dx1);
        tt -= tex(t, y + dy2, x + dx1);
        tt -= tex(t, y + dy1, x + dx2);
        tt += tex(t, y + dy2, x + dx2);
        d += tt * src[k][4] / ((dx2 - dx1) * (dy2 - dy1));
    }
    return (float)d;
}

// surf.cu:175-203.  Grid covers the whole layer region; samples outside the valid window are written as 0
// (the reference leaves them unwritten = stale memory).
//
// Round 3 formulation (the round-2 kernel was the reference's shape -- 40 clamped taps and ten double-precision
// divisions per sample, ~700 VALU slots per sample, and that arithmetic, not the gathers, was its time):
//   * a VALID sample never needs the clamp of the reference's texture fetch: its filter lies inside the image by the definition of
//     samples_i / samples_j, so every tap address is base(sample) + a wave-uniform offset -- one 32-bit lane offset per sample,
//     the tap's offset rides the scalar base of the load (`global_load_dword v, voff, s[base]`): no address arithmetic per tap;
//   * the ten boxes share corners: Dxx and Dyy are three boxes on a 4 x 2 / 2 x 4 corner grid, Dxy four boxes on a 4 x 4 grid:
//     32 distinct taps instead of 40;
//   * a box sum is formed in u32 (exact: the reference adds the same four integers, each < 2^32, in double, which is exact too),
//     converted once, and divided by its area as q = a y, q += fma(-q, b, a) y with y = RN(1 / b): for integer |a| < 2^35 and
//     integer b < 2^24 this is the correctly rounded quotient a / b -- the value the reference's division gives -- checked on
//     4.1e8 cases incl. every area that can occur (tests/test_surf.py::test_box_division_by_reciprocal_is_exact);
//   * workgroups walk the layer in XCD-contiguous row bands (the eight L2s each keep their own rows of the integral table, all
//     layers of an octave back to back) -- profiles/r02u counted 3.2 x the table in FETCH_SIZE with the plain order.
// det / trace planes stay bit-identical to the oracle (hence to surf.cl / surf.cu).
struct HaarGeo {       // per layer: tap offsets (elements, relative to the sample's top-left corner) and 1 / area, area of the 10 boxes
    int xx[4][2];      // Dxx corners [x edge 0..3][y edge 0..1]
    int yy[2][4];      // Dyy corners [x edge 0..1][y edge 0..3]
    int xy[4][4];      // Dxy corners [y edge][x edge]
    double ry[10], area[10];   // boxes: Dxx 0..2, Dyy 3..5, Dxy 6..9
};
// host side (the geometry of a layer does not depend on the sample): rintf = round-half-even = __float2int_rn of the device code
constexpr int kMaxFusedOctaves = 6;
template <class Off>   // off(ey, ex): word offset of the tap (ey rows, ex columns) from the sample's top-left corner
static HaarGeo haar_geo_off(int size, Off off)
{
    HaarGeo g;
    const float ratio = (float)size / 9;
    const auto rnh = [](float v) { return (int)rintf(v); };
    const int e0369[4] = {rnh(ratio * 0.f), rnh(ratio * 3.f), rnh(ratio * 6.f), rnh(ratio * 9.f)};
    const int e27[2] = {rnh(ratio * 2.f), rnh(ratio * 7.f)};
    const int e1458[4] = {rnh(ratio * 1.f), rnh(ratio * 4.f), rnh(ratio * 5.f), rnh(ratio * 8.f)};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j) { g.xx[i][j] = off(e27[j], e0369[i]); g.yy[j][i] = off(e0369[i], e27[j]); }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) g.xy[i][j] = off(e1458[i], e1458[j]);
    for (int k = 0; k < 3; ++k) {
        g.area[k] = (double)((e0369[k + 1] - e0369[k]) * (e27[1] - e27[0]));
        g.area[3 + k] = g.area[k];   // Dyy is Dxx transposed: the same edge differences
    }
    g.area[6] = (double)((e1458[1] - e1458[0]) * (e1458[1] - e1458[0]));
    g.area[7] = (double)((e1458[3] - e1458[2]) * (e1458[1] - e1458[0]));
    g.area[8] = g.area[7];
    g.area[9] = (double)((e1458[3] - e1458[2]) * (e1458[3] - e1458[2]));
    for (int k = 0; k < 10; ++k) g.ry[k] = 1.0 / g.area[k];
    return g;
}
static HaarGeo haar_geo(int size, int sld) { return haar_geo_off(size, [sld](int ey, int ex) { return ey * sld + ex; }); }

// ---- polyphase copies of the integral image for octaves >= 1 (round 5).  A sample of octave o sits at S[(i << o)][(j << o)] and its
// taps at fixed offsets (ey, ex) from there: the 64 lanes of a wave (consecutive j) read words 2^o apart -- 8 .. 32 lines of 64 B per
// load, the L1 tag-lookup rate that bounds the gather path (profiles/surf_counters.json).  With the integral image also stored as 4^o
// PHASE PLANES per octave, plane (y & m, x & m) holding S[y][x] at (y >> o, x >> o), the same tap is
//     plane(ey & m, ex & m)[i + (ey >> o)][j + (ex >> o)]
// i.e. consecutive lanes read CONSECUTIVE words (4-5 lines per load), and the tap is still "lane offset + wave-uniform offset": only the
// geometry table and the lane offset change, the integers read -- and with them every det / trace value -- are the same.
struct PolyGeo { int prows, pld; long long plane_words, base; };   // per octave (octave 0: unused)
static PolyGeo poly_geo(int rows, int cols, int o, long long base)
{
    PolyGeo g;
    g.prows = (rows >> o) + 2; g.pld = align_up((cols >> o) + 2, 64);
    g.plane_words = (long long)g.prows * g.pld; g.base = base;
    return g;
}
static long long poly_total_words(int rows, int cols, int n_octaves)
{
    long long w = 0;
    for (int o = 1; o < n_octaves; ++o) w += poly_geo(rows, cols, o, 0).plane_words << (2 * o);
    return w;
}
struct PolyArgs { int n; int prows[kMaxFusedOctaves], pld[kMaxFusedOctaves]; long long plane_words[kMaxFusedOctaves], base[kMaxFusedOctaves]; };
// one thread per word of the integral image: coalesced read, one store per octave (runs of 64 >> o consecutive words per plane)
__global__ __launch_bounds__(256) void k_poly_build(SumTex t, unsigned *poly, PolyArgs A)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x > t.cols || y > t.rows) return;
    const unsigned v = t.s[(long long)y * t.sld + x];
#pragma unroll
    for (int o = 1; o < kMaxFusedOctaves; ++o) {
        if (o >= A.n) break;
        const int m = (1 << o) - 1;
        poly[A.base[o] + (long long)(((y & m) << o) + (x & m)) * A.plane_words[o] + (long long)(y >> o) * A.pld[o] + (x >> o)] = v;
    }
}
constexpr int kDetLayers = 6;   // layers of one launch (nOctaveLayers + 2 <= 6; more layers: several launches)
struct HaarGeoSet { HaarGeo l[kDetLayers]; };
// (tt * w) / area, correctly rounded (w = +-1, +-2: the product is exact)
__device__ __forceinline__ double box_div(unsigned tt, double w, double area, double ry)
{
    const double a = (double)tt * w;
    const double q = a * ry;
    return fma(fma(-q, area, a), ry, q);
}
// Dxx, Dyy, Dxy of the sample whose top-left corner sits at lane byte offset voff, then det and trace (surf.cu:175-203)
template <class Geo>   // HaarGeo in kernel-argument space, or the same struct behind a constant-address-space reference
__device__ __forceinline__ void haar_det_trace(const SumTex &t, const Geo &g, unsigned voff, float &d, float &tr)
{
    const auto T = [&](int o) { return *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(t.s + o) + voff); };
    // the reference's order (surf.cu:133-150): per box +(y1,x1) -(y2,x1) -(y1,x2) +(y2,x2); d accumulates box by box in double
    unsigned cxx[4][2], cyy[2][4], cxy[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { cxx[a][b] = T(g.xx[a][b]); cyy[b][a] = T(g.yy[b][a]); }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) cxy[a][b] = T(g.xy[a][b]);
    const double wxx[3] = {1.0, -2.0, 1.0};
    double sx = 0, sy = 0, sxy = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        sx += box_div(cxx[k][0] - cxx[k][1] - cxx[k + 1][0] + cxx[k + 1][1], wxx[k], g.area[k], g.ry[k]);
        sy += box_div(cyy[0][k] - cyy[0][k + 1] - cyy[1][k] + cyy[1][k + 1], wxx[k], g.area[3 + k], g.ry[3 + k]);
    }
    // c_DXY (surf.cu:159): {1,1,4,4,+1}, {5,1,8,4,-1}, {1,5,4,8,-1}, {5,5,8,8,+1}  (x1, y1, x2, y2, w); cxy[y edge][x edge]
    sxy += box_div(cxy[0][0] - cxy[1][0] - cxy[0][1] + cxy[1][1], 1.0, g.area[6], g.ry[6]);
    sxy += box_div(cxy[0][2] - cxy[1][2] - cxy[0][3] + cxy[1][3], -1.0, g.area[7], g.ry[7]);
    sxy += box_div(cxy[2][0] - cxy[3][0] - cxy[2][1] + cxy[3][1], -1.0, g.area[8], g.ry[8]);
    sxy += box_div(cxy[2][2] - cxy[3][2] - cxy[2][3] + cxy[3][3], 1.0, g.area[9], g.ry[9]);
    const float dx = (float)sx, dy = (float)sy, dxy = (float)sxy;
    d = dx * dy - 0.81f * dxy * dxy;
    tr = dx + dy;
}
__global__ __launch_bounds__(256) void k_det_trace(SumTex t, float *det, float *trace, int dld, int octave, int layer0, int nlayers2, int nbx, int nby, HaarGeoSet G)
{
    // XCD-contiguous order: workgroup id -> XCD id % 8 (MI355X_MICROARCH.md); each XCD takes a contiguous run of row bands, and inside
    // it the layers of a band follow each other
    const unsigned nwg = nbx * nby * nlayers2, orig = blockIdx.x;
    const unsigned xcd = orig & 7u, qq = nwg >> 3, rr = nwg & 7u;
    const unsigned lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    const int bx = lid % nbx, ll = (lid / nbx) % nlayers2, by = lid / (nbx * nlayers2), layer = layer0 + ll;
    const int layer_rows = t.rows >> octave, layer_cols = t.cols >> octave;
    const int jj = bx * 64 + (threadIdx.x & 63);
    const int ii = by * 4 + (threadIdx.x >> 6);
    if (jj >= layer_cols || ii >= layer_rows) return;
    const int size = calc_size(octave, layer);
    const int samples_i = 1 + ((t.rows - size) >> octave), samples_j = 1 + ((t.cols - size) >> octave);
    const int margin = (size >> 1) >> octave;
    const int i = ii - margin, j = jj - margin;
    float d = 0.f, tr = 0.f;
    if (size <= t.rows && size <= t.cols && i >= 0 && j >= 0 && i < samples_i && j < samples_j) {
        const HaarGeo &g = G.l[ll];   // kernel argument: scalar registers
        unsigned voff = 4u * ((unsigned)(i << octave) * (unsigned)t.sld + (unsigned)(j << octave));
        asm volatile("" : "+v"(voff));             // keep the lane offset a 32-bit VGPR: tap offsets stay on the scalar base
        haar_det_trace(t, g, voff, d, tr);
    }
    const long long o = (long long)(layer * layer_rows + ii) * dld + jj;
    det[o] = d;
    trace[o] = tr;
}

// ------------------------------------------------------------------ non-maximum suppression
// Mask::check surf.cu:229-261
__device__ __forceinline__ bool mask_check(const SumTex &m, int sum_i, int sum_j, int size)
{
    const float ratio = (float)size / 9.0f;
    const int dx1 = rn(ratio * 0.f), dy1 = rn(ratio * 0.f), dx2 = rn(ratio * 9.f), dy2 = rn(ratio * 9.f);
    float tt = 0, d = 0;
    tt += (float)tex(m, sum_i + dy1, sum_j + dx1);
    tt -= (float)tex(m, sum_i + dy2, sum_j + dx1);
    tt -= (float)tex(m, sum_i + dy1, sum_j + dx2);
    tt += (float)tex(m, sum_i + dy2, sum_j + dx2);
    d += tt * 1.f / ((dx2 - dx1) * (dy2 - dy1));
    return d >= 0.5f;
}

struct NmsArgs {
    const float *det, *trace;
    int dld, rows, cols, octave, nlayers;   // nlayers = nOctaveLayers
    float thr;
    SumTex mask;                            // mask.s == nullptr: no mask
    unsigned long long *bits;               // [nlayers][layer_rows][chunks] ballot of the maxima
    const unsigned long long *sbits;        // octave 0 of the all-octave plan with the maxima flagged inside the det kernel (fuse0): sign bit of the trace at each maximum (same layout); else nullptr: nms_write_row reads the trace plane
    unsigned *rowcnt;                       // [nlayers * layer_rows (+1)]  exclusive offsets of the rows (k_scan_counts), total at the end
    unsigned *segcnt;                       // [nlayers * layer_rows][nseg]  maxima per row segment (k_nms_flag)
    int chunks, nseg;                       // 64-sample chunks per row; row segments of kNmsSeg chunks
};

// surf.cu:263-355: one wave per (layer, row); flags per 64-column chunk.  The centre values of four chunks are loaded together (the
// round-2 loop issued one load per chunk and waited for it: 60 dependent round trips per 4K row, 176 us per octave-0 launch for
// 66 MB); a chunk without a value above the threshold -- nearly all -- costs nothing more.
#ifndef MI_SURF_NMS_SEG
#define MI_SURF_NMS_SEG 8
#endif
constexpr int kNmsSeg = MI_SURF_NMS_SEG;   // chunks of one wave: a 4K row is 8 waves (one wave per row left the loop at 60 dependent round trips)
__device__ __forceinline__ void nms_flag_row(const NmsArgs &A, int r, int seg)
{
    const int lane = threadIdx.x & 63;
    const int cbeg = seg * kNmsSeg, cend = min(cbeg + kNmsSeg, A.chunks);
    const int layer_rows = A.rows >> A.octave, layer_cols = A.cols >> A.octave;
    if (r >= A.nlayers * layer_rows) return;
    const int layer = r / layer_rows + 1, i = r % layer_rows;
    const int size = calc_size(A.octave, layer);
    const int margin = ((calc_size(A.octave, layer + 1) >> 1) >> A.octave) + 1;
    unsigned cnt = 0;
    const bool row_ok = i >= margin && i < layer_rows - margin;
    constexpr int CB = 4;
    const float *row = A.det + (long long)(layer * layer_rows + min(i, layer_rows - 1)) * A.dld;   // the centre row of this wave's samples
    for (int c0 = cbeg; c0 < cend; c0 += CB) {
        float v[CB];
        bool in[CB];
        // Round 5: every lane loads ITS OWN centre value, candidate or not (columns up to dld - 1 lie inside the row's allocation; what
        // is there beyond the layer's samples is compared by nobody), plus the two values left and right of the group: the in-row
        // neighbours of a candidate are then the neighbouring LANES' registers -- the first of the four stages below costs no load
        // and no round trip (it ends nearly every chunk: in-plane maxima are rare), where it used to cost one per chunk.
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int j = (c0 + k) * 64 + lane;
            in[k] = row_ok && c0 + k < cend && j >= margin && j < layer_cols - margin;
            v[k] = (row_ok && c0 + k < cend) ? row[j] : -FLT_MAX;
        }
        const float eL = (row_ok && c0 > 0) ? row[c0 * 64 - 1] : -FLT_MAX;
        const float eR = (row_ok && c0 + CB < A.chunks) ? row[(c0 + CB) * 64] : -FLT_MAX;
        // The 26 strict comparisons of surf.cu:316-343 in FOUR stages -- same row, the two other rows of the layer, the layer below, the
        // layer above -- STAGE-MAJOR over the group's chunks (round 5): a stage's loads of all four chunks are issued together and a
        // stage is entered only while some lane of some chunk is still a candidate.  In-row maxima are common on a textured frame (one
        // sample in five to ten), so chunk-major order -- the round-3 form -- made three to four dependent round trips PER CHUNK.
        // margin >= 1 and 1 <= layer <= nlayers keep every neighbour inside the planes.
        bool ismax[CB];
        bool any = false;
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int j = (c0 + k) * 64 + lane;
            ismax[k] = in[k] && v[k] > A.thr;
            if (ismax[k] && A.mask.s) {
                const int sum_i = (i - ((size >> 1) >> A.octave)) << A.octave, sum_j = (j - ((size >> 1) >> A.octave)) << A.octave;
                ismax[k] = mask_check(A.mask, sum_i, sum_j, size);
            }
            // ctr[-1], ctr[1]: the neighbouring lanes' values; lane 0 / lane 63 take the previous / next chunk's edge lane (or eL / eR)
            const float fromL = k > 0 ? __shfl(v[k > 0 ? k - 1 : 0], 63) : eL, fromR = k + 1 < CB ? __shfl(v[k + 1 < CB ? k + 1 : k], 0) : eR;
            const float up = __shfl_up(v[k], 1), dn = __shfl_down(v[k], 1);
            const float left = lane == 0 ? fromL : up, right = lane == 63 ? fromR : dn;
            ismax[k] = ismax[k] && v[k] > left && v[k] > right;
            any = any || ismax[k];
        }
        if (__ballot(any) != 0ull) {
            // the two other rows of the layer: six values per candidate, all chunks' loads in flight together
            float n[CB][6];
#pragma unroll
            for (int k = 0; k < CB; ++k)
                if (ismax[k]) {
                    const float *ctr = row + (c0 + k) * 64 + lane, *qa = ctr - A.dld, *qb = ctr + A.dld;
                    n[k][0] = qa[-1]; n[k][1] = qa[0]; n[k][2] = qa[1]; n[k][3] = qb[-1]; n[k][4] = qb[0]; n[k][5] = qb[1];
                }
            any = false;
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                if (ismax[k]) ismax[k] = v[k] > n[k][0] && v[k] > n[k][1] && v[k] > n[k][2] && v[k] > n[k][3] && v[k] > n[k][4] && v[k] > n[k][5];
                any = any || ismax[k];
            }
#pragma unroll
            for (int dl = -1; dl <= 1; dl += 2) {   // the layer below, then the layer above: nine values per candidate
                if (__ballot(any) == 0ull) break;
                float q9[CB][9];
#pragma unroll
                for (int k = 0; k < CB; ++k)
                    if (ismax[k]) {
                        const float *ctr = row + (c0 + k) * 64 + lane;
#pragma unroll
                        for (int di = -1; di <= 1; ++di) {
                            const float *q = ctr + (long long)(dl * layer_rows + di) * A.dld;
                            q9[k][3 * (di + 1) + 0] = q[-1]; q9[k][3 * (di + 1) + 1] = q[0]; q9[k][3 * (di + 1) + 2] = q[1];
                        }
                    }
                any = false;
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    if (ismax[k]) {
                        bool m = true;
#pragma unroll
                        for (int e = 0; e < 9; ++e) m = m && v[k] > q9[k][e];
                        ismax[k] = m;
                    }
                    any = any || ismax[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            if (c0 + k >= cend) break;
            const unsigned long long m = __ballot(ismax[k]);
            if (lane == 0) A.bits[(long long)r * A.chunks + c0 + k] = m;
            cnt += __popcll(m);
        }
    }
#undef DET
    if (lane == 0) A.segcnt[(long long)r * A.nseg + seg] = cnt;
}
__global__ __launch_bounds__(256) void k_nms_flag(NmsArgs A) { nms_flag_row(A, blockIdx.x * 4 + (threadIdx.x >> 6), blockIdx.y); }

// exclusive scan of n (<= 1024 * per) counts by ONE block; total -> cnt[n]
__device__ __forceinline__ void scan_counts_body(unsigned *cnt, const unsigned *seg, int nseg, int n)
{
    __shared__ unsigned part[1024];
    const int per = (n + 1023) / 1024;
    const int b = threadIdx.x * per, e = min(b + per, n);
    const auto row_total = [&](int k) { unsigned c = 0; for (int q = 0; q < nseg; ++q) c += seg[(long long)k * nseg + q]; return c; };
    // a thread's row totals are kept for the second pass when they fit (8 rows per thread: every frame up to 4096 rows x 2 layers);
    // all their loads are issued before the first add (round 5: the two dependent passes over global memory were most of the launch)
    constexpr int kKeep = 8;
    unsigned keep[kKeep];
    unsigned s = 0;
    if (per <= kKeep) {
#pragma unroll
        for (int q = 0; q < kKeep; ++q) keep[q] = (b + q < e) ? row_total(b + q) : 0u;
#pragma unroll
        for (int q = 0; q < kKeep; ++q) s += keep[q];
    } else
    for (int k = b; k < e; ++k) s += row_total(k);
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
    if (per <= kKeep) {
#pragma unroll
        for (int q = 0; q < kKeep; ++q) if (b + q < e) { cnt[b + q] = run; run += keep[q]; }
    } else
    for (int k = b; k < e; ++k) { const unsigned c = row_total(k); cnt[k] = run; run += c; }
    if (threadIdx.x == 1023) cnt[n] = part[1023];
}
__global__ __launch_bounds__(1024) void k_scan_counts(unsigned *cnt, const unsigned *seg, int nseg, int n) { scan_counts_body(cnt, seg, nseg, n); }

// write candidates {j, i, layer, laplacian} in scan order; count (clamped) -> ncand
__device__ __forceinline__ void nms_write_row(const NmsArgs &A, int r, int4 *cand, int max_candidates, unsigned *ncand)
{
    const int lane = threadIdx.x & 63;
    const int layer_rows = A.rows >> A.octave;
    const int nrows = A.nlayers * layer_rows;
    if (r == 0 && lane == 0) *ncand = min(A.rowcnt[nrows], (unsigned)max_candidates);
    if (r >= nrows) return;
    const int layer = r / layer_rows + 1, i = r % layer_rows;
    unsigned base = A.rowcnt[r];
    // 64 chunk words per trip: lane c holds word c0 + c, a wave-wide exclusive sum of the popcounts gives every word's first index, and
    // only the non-empty words (few) are walked (round 5: the row's words were read one after the other, 60 dependent loads at 4K)
    for (int c0 = 0; c0 < A.chunks; c0 += 64) {
        const unsigned long long mine = (c0 + lane < A.chunks) ? A.bits[(long long)r * A.chunks + c0 + lane] : 0ull;
        const unsigned pc = (unsigned)__popcll(mine);
        unsigned incl = pc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        const unsigned excl = incl - pc;
        unsigned long long nz = __ballot(pc != 0u);
        while (nz) {
            const int cl = __builtin_ctzll(nz);
            nz &= nz - 1;
            const unsigned long long m = __shfl(mine, cl);
            const unsigned wbase = base + __shfl(excl, cl);
            if ((m >> lane) & 1ull) {
                const unsigned idx = wbase + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                if (idx < (unsigned)max_candidates) {
                    const int c = c0 + cl, j = c * 64 + lane;
                    const int lap = A.sbits ? (((A.sbits[(long long)r * A.chunks + c] >> lane) & 1ull) ? -1 : 1)
                                            : (int)copysignf(1.0f, A.trace[(long long)(layer * layer_rows + i) * A.dld + j]);
                    cand[idx] = make_int4(j, i, layer, lap);
                }
            }
        }
        base += __shfl(incl, 63);
    }
}
__global__ __launch_bounds__(256) void k_nms_write(NmsArgs A, int4 *cand, int max_candidates, unsigned *ncand)
{
    nms_write_row(A, blockIdx.x * 4 + (threadIdx.x >> 6), cand, max_candidates, ncand);
}

// ------------------------------------------------------------------ sub-pixel interpolation
// core/cuda/utility.hpp solve3x3<float>: the reciprocal of the determinant in double, the components rounded back to float
__device__ __forceinline__ bool solve3x3(const float A[3][3], const float b[3], float x[3])
{
    const float det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                      A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    if (det != 0) {
        const double invdet = 1.0 / (double)det;
        x[0] = (float)(invdet * (double)(b[0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (b[1] * A[2][2] - A[1][2] * b[2]) +
                                         A[0][2] * (b[1] * A[2][1] - A[1][1] * b[2])));
        x[1] = (float)(invdet * (double)(A[0][0] * (b[1] * A[2][2] - A[1][2] * b[2]) - b[0] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                                         A[0][2] * (A[1][0] * b[2] - b[1] * A[2][0])));
        x[2] = (float)(invdet * (double)(A[0][0] * (A[1][1] * b[2] - b[1] * A[2][1]) - A[0][1] * (A[1][0] * b[2] - b[1] * A[2][0]) +
                                         b[0] * (A[1][0] * A[2][1] - A[1][1] * A[2][0])));
        return true;
    }
    return false;
}

// surf.cu:391-493, pass 1: one thread per candidate (any number of workgroups) evaluates the 3-D quadratic refinement and
// stores {ok, x, y, size, hessian} at the candidate's index; pass 2 (k_interp_compact, one workgroup) scans the ok flags and
// appends the accepted features in candidate order behind the features of the previous octaves.
struct InterpOut { float px, py, psize, hess; int lap, ok; };

// RECOMPUTE (octave 0 with the maxima flagged inside the det kernel): there is no det plane -- the 27 values are evaluated again from the
// integral image (det0_at: the same taps, the same box_div, the same order as the tile kernel: the same bits)
__device__ float det0_at(const SumTex &t, int layer, int ii, int jj);
__device__ __forceinline__ bool interp_eval_one(const float *det, int dld, int rows, int cols, int octave, const int4 *cand,
                                                const unsigned *ncand_p, InterpOut *tmp, int c, const SumTex *recompute = nullptr)
{
    if (c >= (int)*ncand_p) return false;
    const int layer_rows = rows >> octave;
    const int4 mp = cand[c];
    float N9[3][3][3];
    if (recompute) {
#pragma unroll 1
        for (int z = 0; z < 3; ++z)
#pragma unroll 1
            for (int y = 0; y < 3; ++y)
#pragma unroll
                for (int x = 0; x < 3; ++x)
                    N9[z][y][x] = det0_at(*recompute, mp.z - 1 + z, mp.y - 1 + y, mp.x - 1 + x);
    } else {
#pragma unroll
    for (int z = 0; z < 3; ++z)
#pragma unroll
        for (int y = 0; y < 3; ++y)
#pragma unroll
            for (int x = 0; x < 3; ++x)
                N9[z][y][x] = det[(long long)(layer_rows * (mp.z - 1 + z) + mp.y - 1 + y) * dld + mp.x - 1 + x];
    }
    float dD[3], H[3][3], xs[3];
    dD[0] = -0.5f * (N9[1][1][2] - N9[1][1][0]);
    dD[1] = -0.5f * (N9[1][2][1] - N9[1][0][1]);
    dD[2] = -0.5f * (N9[2][1][1] - N9[0][1][1]);
    H[0][0] = N9[1][1][0] - 2.0f * N9[1][1][1] + N9[1][1][2];
    H[0][1] = 0.25f * (N9[1][2][2] - N9[1][2][0] - N9[1][0][2] + N9[1][0][0]);
    H[0][2] = 0.25f * (N9[2][1][2] - N9[2][1][0] - N9[0][1][2] + N9[0][1][0]);
    H[1][0] = H[0][1];
    H[1][1] = N9[1][0][1] - 2.0f * N9[1][1][1] + N9[1][2][1];
    H[1][2] = 0.25f * (N9[2][2][1] - N9[2][0][1] - N9[0][2][1] + N9[0][0][1]);
    H[2][0] = H[0][2];
    H[2][1] = H[1][2];
    H[2][2] = N9[0][1][1] - 2.0f * N9[1][1][1] + N9[2][1][1];
    bool ok = solve3x3(H, dD, xs);
    ok = ok && fabsf(xs[0]) <= 1.f && fabsf(xs[1]) <= 1.f && fabsf(xs[2]) <= 1.f;
    InterpOut o;
    o.px = o.py = o.psize = 0.f; o.hess = N9[1][1][1]; o.lap = mp.w; o.ok = 0;
    if (ok) {
        const int size = calc_size(octave, mp.z);
        const int sum_i = (mp.y - ((size >> 1) >> octave)) << octave, sum_j = (mp.x - ((size >> 1) >> octave)) << octave;
        const float center_i = sum_i + (float)(size - 1) / 2, center_j = sum_j + (float)(size - 1) / 2;
        o.px = center_j + xs[0] * (1 << octave);
        o.py = center_i + xs[1] * (1 << octave);
        const int ds = size - calc_size(octave, mp.z - 1);
        o.psize = roundf(size + xs[2] * ds);
        const float sc = o.psize * 1.2f / 9.0f;
        const int grad_wav_size = 2 * rn(2.0f * sc);
        ok = (rows + 1) >= grad_wav_size && (cols + 1) >= grad_wav_size;
    }
    o.ok = ok ? 1 : 0;
    tmp[c] = o;
    return ok;
}
__global__ __launch_bounds__(256) void k_interp_eval(const float *det, int dld, int rows, int cols, int octave, const int4 *cand,
                                                     const unsigned *ncand_p, InterpOut *tmp)
{
    interp_eval_one(det, dld, rows, cols, octave, cand, ncand_p, tmp, blockIdx.x * 256 + threadIdx.x);
}

// appends the accepted candidates of one octave behind feature nfeat0; returns the new feature count (the same in every thread)
__device__ __forceinline__ unsigned interp_compact_body(const InterpOut *tmp, const unsigned *ncand_p, int octave, float *kp, int kld,
                                                        int max_features, unsigned nfeat0)
{
    __shared__ unsigned part[1024];
    const unsigned ncand = *ncand_p;
    const int per = (int)(ncand + 1023) / 1024;
    const int b = threadIdx.x * per, e = min(b + per, (int)ncand);
    unsigned cnt = 0;
    for (int c = b; c < e; ++c) cnt += (unsigned)tmp[c].ok;
    part[threadIdx.x] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned ind = nfeat0 + (threadIdx.x ? part[threadIdx.x - 1] : 0u);
    for (int c = b; c < e; ++c) {
        const InterpOut o = tmp[c];
        if (!o.ok) continue;
        if (ind < (unsigned)max_features) {
            kp[0 * kld + ind] = o.px;
            kp[1 * kld + ind] = o.py;
            reinterpret_cast<int *>(kp)[2 * kld + ind] = o.lap;     // LAPLACIAN_ROW holds int bit patterns (cuda.hpp:89-99)
            reinterpret_cast<int *>(kp)[3 * kld + ind] = octave;
            kp[4 * kld + ind] = o.psize;
            kp[6 * kld + ind] = o.hess;
        }
        ++ind;
    }
    __syncthreads();
    const unsigned total = min(nfeat0 + part[1023], (unsigned)max_features);
    __syncthreads();   // part[] is reused by the next octave of k_interp_compact_all
    return total;
}
__global__ __launch_bounds__(1024) void k_interp_compact(const InterpOut *tmp, const unsigned *ncand_p, int octave, float *kp, int kld,
                                                         int max_features, unsigned *nfeat_p)
{
    const unsigned total = interp_compact_body(tmp, ncand_p, octave, kp, kld, max_features, *nfeat_p);
    if (threadIdx.x == 0) *nfeat_p = total;
}

// ------------------------------------------------------------------ orientation
__device__ __forceinline__ float reduce32(float v)   // device::reduce<32>, plus<float>: shfl_down 16,8,4,2,1 inside a 32-lane half
{
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) v = v + __shfl_down(v, off, 32);
    return v;
}

// surf.cu:527-658: one wave per feature; the two 32-lane halves play threadIdx.y = {0,2} and {1,3}
__global__ __launch_bounds__(256) void k_orientation(SumTex t, float *kp, int kld, const unsigned *nfeat_p, int n_fixed, const float *apt)
{
    __shared__ float sh[4][3][128];
    __shared__ float best[4][4][3];
    const float c_NX[2][5] = {{0, 0, 2, 4, -1}, {2, 0, 4, 4, 1}};   // surf.cu:524-525
    const float c_NY[2][5] = {{0, 0, 4, 2, 1}, {0, 2, 4, 4, -1}};
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nfeat = nfeat_p ? min((int)*nfeat_p, n_fixed) : n_fixed;
    // a wave walks the features grid-cyclically (round 5: with the count on the device the grid is a fixed few workgroups per CU instead of
    // maxFeatures / 4 workgroups, four fifths of them empty at 4K); a wave's LDS operations execute in order, so its rows of sh / best
    // are free again when the next feature's stores issue
    for (int f = blockIdx.x * 4 + wv; f < nfeat; f += (int)gridDim.x * 4) {
    const float fx = kp[f], fy = kp[kld + f], fsz = kp[4 * kld + f];
    const float s = fsz * 1.2f / 9.0f;
    const int grad_wav_size = 2 * rn(2.0f * s);
    if ((t.rows + 1) < grad_wav_size || (t.cols + 1) < grad_wav_size) continue;   // the reference returns without writing
    float *sX = sh[wv][0], *sY = sh[wv][1], *sA = sh[wv][2];
    for (int tid = lane; tid < 128; tid += 64) {
        float X = 0.f, Y = 0.f, angle = 0.f;
        if (tid < 113) {
            const float margin = (float)(grad_wav_size - 1) / 2.0f;
            const int x = rn(fx + apt[tid] * s - margin), y = rn(fy + apt[113 + tid] * s - margin);
            if (y >= 0 && y < (t.rows + 1) - grad_wav_size && x >= 0 && x < (t.cols + 1) - grad_wav_size) {
                X = apt[226 + tid] * haar<2>(t, c_NX, 4, grad_wav_size, y, x);
                Y = apt[226 + tid] * haar<2>(t, c_NY, 4, grad_wav_size, y, x);
                angle = atan2f(Y, X);
                if (angle < 0) angle += 2.0f * CV_PI_F;
                angle *= 180.0f / CV_PI_F;
            }
        }
        sX[tid] = X; sY[tid] = Y; sA[tid] = angle;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const int tx = lane & 31, half = lane >> 5;
    int a4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a4[q] = rn(sA[tx + 32 * q]);
    for (int rep = 0; rep < 2; ++rep) {
        const int ty = half + 2 * rep;
        float bestx = 0, besty = 0, best_mod = 0;
        for (int i = 0; i < 18; ++i) {
            const int dir = (i * 4 + ty) * 5;
            float sumx = 0.0f, sumy = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = abs(a4[q] - dir);
                if (d < 30 || d > 330) {
                    if (q == 0) { sumx = sX[tx]; sumy = sY[tx]; }
                    else { sumx += sX[tx + 32 * q]; sumy += sY[tx + 32 * q]; }
                }
            }
            sumx = reduce32(sumx);
            sumy = reduce32(sumy);
            const float temp_mod = sumx * sumx + sumy * sumy;   // meaningful in lane tx == 0 of each half
            if (temp_mod > best_mod) { best_mod = temp_mod; bestx = sumx; besty = sumy; }
        }
        if (tx == 0) { best[wv][ty][0] = bestx; best[wv][ty][1] = besty; best[wv][ty][2] = best_mod; }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        int bi = 0;
        if (best[wv][1][2] > best[wv][bi][2]) bi = 1;
        if (best[wv][2][2] > best[wv][bi][2]) bi = 2;
        if (best[wv][3][2] > best[wv][bi][2]) bi = 3;
        float kp_dir = atan2f(best[wv][bi][1], best[wv][bi][0]);
        if (kp_dir < 0) kp_dir += 2.0f * CV_PI_F;
        kp_dir *= 180.0f / CV_PI_F;
        kp_dir = 360.0f - kp_dir;
        if (fabsf(kp_dir - 360.f) < FLT_EPSILON) kp_dir = 0.f;
        kp[5 * kld + f] = kp_dir;
    }
    }
}

__global__ __launch_bounds__(256) void k_fill_angle(float *kp, int kld, const unsigned *nfeat_p, int n_fixed, float value)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int nfeat = nfeat_p ? (int)*nfeat_p : n_fixed;
    if (f < nfeat) kp[5 * kld + f] = value;   // upright: 360 - 90 (surf.cuda.cpp:211-212)
}

// ------------------------------------------------------------------ descriptors
struct Win { const unsigned char *img; long long step; int rows, cols, win; float cx, cy, off, c, s; };
__device__ __forceinline__ float win_get(const Win &w, int i, int j)
{
    // WinReader (surf.cu:709-731) over a point-sampled, clamp-addressed texture: texel (floor(x), floor(y))
    const float px = w.cx + (w.off + j) * w.c + (w.off + i) * w.s;
    const float py = w.cy - (w.off + j) * w.s + (w.off + i) * w.c;
    const int x = clampi(__float2int_rd(px), 0, w.cols - 1), y = clampi(__float2int_rd(py), 0, w.rows - 1);
    return (float)w.img[(long long)y * w.step + x];
}
// saturate_cast<uchar>(float) = cvt.rni.sat.u8.f32: round to nearest even, saturate -- the filters below return WinReader::elem_type
__device__ __forceinline__ float sat_u8(float v) { return fminf(fmaxf(rintf(v), 0.f), 255.f); }
__device__ float linear_filter(const Win &w, float y, float x)   // core/cuda/filters.hpp LinearFilter (s <= 1: provided keypoints only)
{
    float out = 0.0f;
    const int x1 = __float2int_rd(x), y1 = __float2int_rd(y), x2 = x1 + 1, y2 = y1 + 1;
    out = out + win_get(w, y1, x1) * ((x2 - x) * (y2 - y));
    out = out + win_get(w, y1, x2) * ((x - x1) * (y2 - y));
    out = out + win_get(w, y2, x1) * ((x2 - x) * (y - y1));
    out = out + win_get(w, y2, x2) * ((x - x1) * (y - y1));
    return sat_u8(out);
}
// sum_{dx in [a, b)} win_get(w, dy, dx) * wgt accumulated onto `out` in ascending dx order (the reference's order);
// the texel reads of 8 consecutive dx are issued together so their latencies overlap (the adds stay sequential)
__device__ __forceinline__ float row_accum(const Win &w, int dy, int a, int b, float wgt, float out)
{
    int dx = a;
    for (; dx + 16 <= b; dx += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = win_get(w, dy, dx + k);
#pragma unroll
        for (int k = 0; k < 16; ++k) out = out + v[k] * wgt;
    }
    for (; dx + 8 <= b; dx += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = win_get(w, dy, dx + k);
#pragma unroll
        for (int k = 0; k < 8; ++k) out = out + v[k] * wgt;
    }
    for (; dx < b; ++dx) out = out + win_get(w, dy, dx) * wgt;
    return out;
}

__device__ float area_filter(const Win &w, float x, float y, float s)   // core/cuda/filters.hpp AreaFilter with scale_x = scale_y = s
{
    const float fsx1 = x * s, fsx2 = fsx1 + s;
    const int sx1 = (int)ceilf(fsx1), sx2 = (int)floorf(fsx2);
    const float fsy1 = y * s, fsy2 = fsy1 + s;
    const int sy1 = (int)ceilf(fsy1), sy2 = (int)floorf(fsy2);
    const float scale = 1.f / (fminf(s, w.win - fsx1) * fminf(s, w.win - fsy1));   // src.width = src.height = win_size (surf.cu:754)
    float out = 0.f;
    for (int dy = sy1; dy < sy2; ++dy) {
        out = row_accum(w, dy, sx1, sx2, scale, out);
        if (sx1 > fsx1) out = out + win_get(w, dy, sx1 - 1) * ((sx1 - fsx1) * scale);
        if (sx2 < fsx2) out = out + win_get(w, dy, sx2) * ((fsx2 - sx2) * scale);
    }
    if (sy1 > fsy1) out = row_accum(w, sy1 - 1, sx1, sx2, (sy1 - fsy1) * scale, out);
    if (sy2 < fsy2) out = row_accum(w, sy2, sx1, sx2, (fsy2 - sy2) * scale, out);
    if ((sy1 > fsy1) && (sx1 > fsx1)) out = out + win_get(w, sy1 - 1, sx1 - 1) * ((sy1 - fsy1) * (sx1 - fsx1) * scale);
    if ((sy1 > fsy1) && (sx2 < fsx2)) out = out + win_get(w, sy1 - 1, sx2) * ((sy1 - fsy1) * (fsx2 - sx2) * scale);
    if ((sy2 < fsy2) && (sx2 < fsx2)) out = out + win_get(w, sy2, sx2) * ((fsy2 - sy2) * (fsx2 - sx2) * scale);
    if ((sy2 < fsy2) && (sx1 > fsx1)) out = out + win_get(w, sy2, sx1 - 1) * ((fsy2 - sy2) * (sx1 - fsx1) * scale);
    return sat_u8(out);
}

// The 21 x 21 patch P -> 16 sub-regions of 25 weighted Haar responses, one 32-lane tree per half-wave -> 64 or 128 sums -> L2
// normalisation (surf.cu:786-912); shared by the two patch kernels below.  P and D are the workgroup's shared arrays.
template <bool EXT>
__device__ __forceinline__ void desc_tail(const float (&P)[21][21], float (&D)[128], float (&part)[4], const float *dw, float *desc_row)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tx = lane & 31, half = lane >> 5;
    {
        const int ty = wv * 2 + half;    // sub-region 0..15 (threadIdx.y of the reference)
        const int xb = ty % 4, yb = ty / 4;
        float dx = 0.f, dy = 0.f;
        const int xp = tx % 5, yp = tx / 5;
        if (yp < 5) {
            const int xi = xb * 5 + xp, yi = yb * 5 + yp;
            const float wgt = dw[yi * 20 + xi];
            dx = (P[yi][xi + 1] - P[yi][xi] + P[yi + 1][xi + 1] - P[yi + 1][xi]) * wgt;
            dy = (P[yi + 1][xi] - P[yi][xi] + P[yi + 1][xi + 1] - P[yi][xi + 1]) * wgt;
        }
        if (!EXT) {
            const float a = reduce32(dx), b = reduce32(dy), c = reduce32(fabsf(dx)), d = reduce32(fabsf(dy));
            if (tx == 0) { D[ty * 4 + 0] = a; D[ty * 4 + 1] = b; D[ty * 4 + 2] = c; D[ty * 4 + 3] = d; }
        } else {
            const bool py = dy >= 0;
            const float a = reduce32(py ? dx : 0.f), b = reduce32(py ? fabsf(dx) : 0.f), c = reduce32(py ? 0.f : dx), d = reduce32(py ? 0.f : fabsf(dx));
            const bool pxp = dx >= 0;
            const float e = reduce32(pxp ? dy : 0.f), g = reduce32(pxp ? fabsf(dy) : 0.f), h = reduce32(pxp ? 0.f : dy), k = reduce32(pxp ? 0.f : fabsf(dy));
            if (tx == 0) {
                D[ty * 8 + 0] = a; D[ty * 8 + 1] = b; D[ty * 8 + 2] = c; D[ty * 8 + 3] = d;
                D[ty * 8 + 4] = e; D[ty * 8 + 5] = g; D[ty * 8 + 6] = h; D[ty * 8 + 7] = k;
            }
        }
    }
    __syncthreads();
    // normalize_descriptors<N> (surf.cu:891-912): len = device::reduce<N> of the squares = the 32-lane tree in each of the N / 32 warps,
    // then the tree over the warps' partials (core/cuda/detail/reduce.hpp, GenericOptimized32); val / sqrt(len)
    constexpr int N = EXT ? 128 : 64;
    if (threadIdx.x < N) {
        const float v = D[threadIdx.x] * D[threadIdx.x];
        const float r = reduce32(v);
        if ((threadIdx.x & 31) == 0) part[threadIdx.x >> 5] = r;
    }
    __syncthreads();
    const float len = sqrtf(EXT ? (part[0] + part[2]) + (part[1] + part[3]) : part[0] + part[1]);
    if (threadIdx.x < N) desc_row[threadIdx.x] = D[threadIdx.x] / len;
}

// which of the two patch kernels builds feature f's patch: the staged one from cell side s_stage on, as long as one patch row of the
// feature -- ceil(s) + 3 window rows of floor(21 s) + 2 texels -- fits the tile (a caller may provide keypoints of any size)
__device__ __forceinline__ bool desc_staged(float s, float s_stage, int tile_bytes)
{
    return s >= s_stage && ((long long)ceilf(s) + 3) * ((long long)floorf(20.f * s + s) + 2) <= (long long)tile_bytes;
}
// window geometry of feature f (surf.cu:733-760)
__device__ __forceinline__ float desc_window(Win &w, const unsigned char *img, long long istep, int rows, int cols, const float *kp, int kld, int f)
{
    w.img = img; w.step = istep; w.rows = rows; w.cols = cols; w.cx = kp[f]; w.cy = kp[kld + f];
    const float s = kp[4 * kld + f] * 1.2f / 9.0f;
    const int win_size = (int)(21 * s);
    w.win = win_size;
    w.off = -(win_size - 1.0f) / 2.0f;
    float ddir = 360.0f - kp[5 * kld + f];
    if (fabsf(ddir - 360.f) < FLT_EPSILON) ddir = 0.f;
    ddir *= CV_PI_F / 180.0f;
    sincosf(ddir, &w.s, &w.c);
    return s;
}

// surf.cu:733-912: one workgroup (8 waves) per feature: 21 x 21 patch (one sample per thread, every thread walking its own s x s cell of the
// rotated window through global memory) -> desc_tail.  Features whose cell side s reaches s_stage are left to k_descriptors_staged.
template <bool EXT>
__global__ __launch_bounds__(512) void k_descriptors(const unsigned char *img, long long istep, int rows, int cols, const float *kp,
                                                     int kld, int nfeat_host, const unsigned *nfeat_dev, float *desc, long long dstep /* floats */, const float *dw,
                                                     float s_stage, int tile_bytes)
{
    __shared__ float P[21][21];
    __shared__ float D[128];
    __shared__ float part[4];
    // nfeat_dev: the count is read on the device (detect + describe without a host round trip between them, round 5); nfeat_host then
    // bounds it (maxFeatures) and the grid is a fixed number of workgroups that walk the features block-cyclically
    const int nfeat = nfeat_dev ? min((int)*nfeat_dev, nfeat_host) : nfeat_host;
    // features are ordered by octave, i.e. by patch cost (441 x s^2 texel reads, s up to ~29 at 4 octaves): launch the
    // expensive ones first so they do not form the tail of the grid
    for (int f = nfeat - 1 - (int)blockIdx.x; f >= 0; f -= (int)gridDim.x) {
        Win w;
        const float s = desc_window(w, img, istep, rows, cols, kp, kld, f);
        if (desc_staged(s, s_stage, tile_bytes)) continue;
        if (threadIdx.x < 441) {
            const int tid = threadIdx.x, xl = tid % 21, yl = tid / 21;
            P[yl][xl] = s > 1 ? area_filter(w, (float)xl, (float)yl, s) : linear_filter(w, yl * s, xl * s);
        }
        __syncthreads();
        desc_tail<EXT>(P, D, part, dw, desc + (long long)f * dstep);
        __syncthreads();   // (a workgroup that walks several features: P, D, part are free again)
    }
}

// The same for LARGE features (round 3, r06w: the 908 octave-3 features of the 4K frame took 743 of the 1 146 us, 818 ns each -- a
// thread's s x s cell is a rotated lattice, and the 64 cells a wave reads in step lie s pixels apart: 64 cache lines per load, a third
// of the L1 line rate).  Here the workgroup first STAGES the texels of a strip of patch rows into LDS, lanes arranged as 8 x 8 blocks
// of the window lattice (a rotated 8 x 8 block covers ~12 image rows: 12-16 lines per load instead of 64), then every sample of the
// strip accumulates its cell from LDS.  Texel = the same win_get (same float expression per (dy, dx)), accumulation = the same
// area_filter order: bit-identical patch values.  The strip height is what fits the tile.
__device__ __forceinline__ float area_filter_lds(const unsigned char *tile, int nc, int dy_lo, float x, float y, float s, int win)
{
    const auto T = [&](int dy, int dx) { return (float)tile[(dy - dy_lo) * nc + dx + 1]; };
    const float fsx1 = x * s, fsx2 = fsx1 + s;
    const int sx1 = (int)ceilf(fsx1), sx2 = (int)floorf(fsx2);
    const float fsy1 = y * s, fsy2 = fsy1 + s;
    const int sy1 = (int)ceilf(fsy1), sy2 = (int)floorf(fsy2);
    const float scale = 1.f / (fminf(s, win - fsx1) * fminf(s, win - fsy1));
    float out = 0.f;
    // a window row of the cell: the texel reads of 8 consecutive dx are issued together, the adds stay in the reference's order
    const auto row = [&](int dy, float wgt) {
        const unsigned char *q = tile + (dy - dy_lo) * nc + 1;
        int dx = sx1;
        for (; dx + 8 <= sx2; dx += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (float)q[dx + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) out = out + v[k] * wgt;
        }
        for (; dx < sx2; ++dx) out = out + (float)q[dx] * wgt;
    };
    for (int dy = sy1; dy < sy2; ++dy) {
        row(dy, scale);
        if (sx1 > fsx1) out = out + T(dy, sx1 - 1) * ((sx1 - fsx1) * scale);
        if (sx2 < fsx2) out = out + T(dy, sx2) * ((fsx2 - sx2) * scale);
    }
    if (sy1 > fsy1) row(sy1 - 1, (sy1 - fsy1) * scale);
    if (sy2 < fsy2) row(sy2, (fsy2 - sy2) * scale);
    if ((sy1 > fsy1) && (sx1 > fsx1)) out = out + T(sy1 - 1, sx1 - 1) * ((sy1 - fsy1) * (sx1 - fsx1) * scale);
    if ((sy1 > fsy1) && (sx2 < fsx2)) out = out + T(sy1 - 1, sx2) * ((sy1 - fsy1) * (fsx2 - sx2) * scale);
    if ((sy2 < fsy2) && (sx2 < fsx2)) out = out + T(sy2, sx2) * ((fsy2 - sy2) * (fsx2 - sx2) * scale);
    if ((sy2 < fsy2) && (sx1 > fsx1)) out = out + T(sy2, sx1 - 1) * ((fsy2 - sy2) * (sx1 - fsx1) * scale);
    return sat_u8(out);
}
#ifdef MIFLOW_EXPERIMENTS
// ---- row-coalesced staging of a strip (round 6) -- BUILT, bit-identical, and SLOWER: experiments build
// only (MIFLOW_SURF_COAL_S = the cell side from which a feature takes it; profiles/r19/README.md: at 4K 907 frames/s with the per-texel
// gather against 831 / 788 / 736 with this path from cell side 24 / 16 / 8 on, although it issues a fifteenth of the L1 tag lookups --
// the descriptor launch is bound by the dependent phases of a feature's strips, five barriers each here against two, not by lookups).
// Until round 5 the lattice tile was filled by one byte
// gather per lattice point -- 8 x 8 blocks of the rotated window lattice, 41 L1 tag lookups per wave load for 64 useful bytes, the kernel
// at 0.84 of the L1 lookup rate (profiles/surf_counters.json).  Now a strip is staged in two steps: (1) the image rows the strip's
// lattice touches are copied into an IMAGE tile with aligned dword loads along the rows -- per image row only the segment the rotated
// strip covers (a conservative x-range from the strip's four corners), all segments at one pitch: a wave load touches 4-5 lines for 256
// useful bytes; (2) every lattice point takes its texel from that tile (LDS -> LDS), with win_get's own coordinate expression.  A texel
// the conservative range should ever miss is read from global memory instead: the lattice tile holds the same bytes as before whatever
// the geometry code does, and area_filter_lds / desc_tail are untouched -- bit-identical descriptors by construction.
constexpr int kStripRowsMax = 1024;   // image rows one strip may touch (a strip that touches more takes the per-texel gather)
struct StripRows { int y0, nrows; };
// texel rows the lattice region i in [i0, i1], j in [j0, j1] touches: py is monotone in i and in j (also in float arithmetic: every
// operation of win_get's expression is monotone), so its extremes sit at the four corners
__device__ __forceinline__ StripRows strip_rows(const Win &w, int i0, int i1, int j0, int j1)
{
    float lo = 3.0e38f, hi = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = (k & 1) ? i1 : i0, j = (k & 2) ? j1 : j0;
        const float py = w.cy - (w.off + j) * w.s + (w.off + i) * w.c;
        lo = fminf(lo, py); hi = fmaxf(hi, py);
    }
    StripRows r;
    r.y0 = clampi(__float2int_rd(fmaxf(lo, -1.0e6f)), 0, w.rows - 1);
    r.nrows = clampi(__float2int_rd(fminf(hi, 1.0e6f)), 0, w.rows - 1) - r.y0 + 1;
    return r;
}
// conservative range [xl, xr] of the texel columns the region touches on texel row y (clamped to the image like win_get clamps):
// extremes of px over (rectangle of the region) x (band py in [y, y + 1), widened by 0.05; open-ended at the first / last image row,
// where clamped rows collect) = over the rectangle's corners inside the band and its edges' crossings of the band's two lines
__device__ __forceinline__ void strip_row_range(const Win &w, int i0, int i1, int j0, int j1, int y, int &xl, int &xr)
{
    const float blo = y == 0 ? -3.0e38f : (float)y - 0.05f, bhi = y == w.rows - 1 ? 3.0e38f : (float)y + 1.05f;
    float cx[4], cy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // corners in cyclic order: (i0, j0), (i0, j1), (i1, j1), (i1, j0)
        const int i = (k == 0 || k == 1) ? i0 : i1, j = (k == 1 || k == 2) ? j1 : j0;
        cx[k] = w.cx + (w.off + j) * w.c + (w.off + i) * w.s;
        cy[k] = w.cy - (w.off + j) * w.s + (w.off + i) * w.c;
    }
    float lo = 3.0e38f, hi = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = (k + 1) & 3;
        if (cy[k] >= blo && cy[k] <= bhi) { lo = fminf(lo, cx[k]); hi = fmaxf(hi, cx[k]); }
        const float dy = cy[n] - cy[k];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float Y = e ? bhi : blo;
            if (fabsf(Y) < 1.0e37f && dy != 0.f && (cy[k] - Y) * (cy[n] - Y) <= 0.f) {
                const float x = cx[k] + (Y - cy[k]) / dy * (cx[n] - cx[k]);
                lo = fminf(lo, x); hi = fmaxf(hi, x);
            }
        }
    }
    if (lo > hi) { xl = 0; xr = -1; return; }   // the region does not touch this row
    xl = clampi(__float2int_rd(fmaxf(lo, -1.0e6f)) - 1, 0, w.cols - 1);
    xr = clampi(__float2int_rd(fminf(hi, 1.0e6f)) + 1, 0, w.cols - 1);
}
#endif   // MIFLOW_EXPERIMENTS

#ifndef MI_SURF_ORI_WGS_PER_CU
#define MI_SURF_ORI_WGS_PER_CU 8
#endif
#ifndef MI_SURF_DESC_WGS_PER_CU
#define MI_SURF_DESC_WGS_PER_CU 16   // workgroups per CU of the descriptor launch when the feature count is on the device (r15c at 4K: 3 | 6 | 16 | 48 = 818 | 841 | 905 | 902 frames/s)
#endif
#ifndef MI_SURF_STAGE_U
#define MI_SURF_STAGE_U 8   // r14n: 16 and 24 in flight change nothing (818 / 816 against 820 frames/s): the staging is not bound by its depth
#endif
constexpr int kStageU = MI_SURF_STAGE_U;   // staged texel loads in flight per lane
template <bool EXT>
__global__ __launch_bounds__(512) void k_descriptors_staged(const unsigned char *img, long long istep, int rows, int cols, const float *kp,
                                                            int kld, int nfeat_host, const unsigned *nfeat_dev, float *desc, long long dstep /* floats */,
                                                            const float *dw, float s_stage, int tile_bytes, int img_bytes, float s_coal)
{
    __shared__ float P[21][21];
    __shared__ float D[128];
    __shared__ float part[4];
#ifdef MIFLOW_EXPERIMENTS
    __shared__ int rowx0[kStripRowsMax];   // first staged column of each image row of the strip (its address is 4-byte aligned)
    __shared__ int s_pitch;                // bytes staged per row (the longest row's), before rounding up to dwords
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char tile[];   // [0, tile_bytes): the lattice tile; behind it the image tile
#ifdef MIFLOW_EXPERIMENTS
    unsigned char *const itile = tile + tile_bytes;
#endif
    const int nfeat = nfeat_dev ? min((int)*nfeat_dev, nfeat_host) : nfeat_host;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, ly = lane >> 3, lx = lane & 7;
    for (int f = nfeat - 1 - (int)blockIdx.x; f >= 0; f -= (int)gridDim.x) {
        Win w;
        const float s = desc_window(w, img, istep, rows, cols, kp, kld, f);
        if (!desc_staged(s, s_stage, tile_bytes)) {
            // round 5: ONE launch builds every feature's patch -- a small feature straight from global memory (k_descriptors' body), a
            // large one through the tile -- so that the two kinds share the grid (no second launch waiting for the first one's tail)
            if (threadIdx.x < 441) {
                const int tid = threadIdx.x, xl = tid % 21, yl = tid / 21;
                P[yl][xl] = s > 1 ? area_filter(w, (float)xl, (float)yl, s) : linear_filter(w, yl * s, xl * s);
            }
            __syncthreads();
            desc_tail<EXT>(P, D, part, dw, desc + (long long)f * dstep);
            __syncthreads();
            continue;
        }
        const int dxmax = (int)floorf(20.f * s + s);              // the largest sx2 of area_filter (x = 20)
        const int nc = dxmax + 2;                                 // tile columns: dx = -1 .. dxmax
        const int nbc = (nc + 7) >> 3;
        for (int ya = 0; ya < 21;) {
            // the strip: patch rows ya .. yb - 1, as many as the tile holds (at least one: the host sizes the tile for that)
            const int dy_lo = (int)ceilf((float)ya * s) - 1;
            int yb = ya + 1;
            // features from cell side s_coal on stage their strips row-coalesced (lattice tile + image tile); smaller ones -- a strip or two
            // per feature, bound by the latency of their phases, not by L1 lookups -- keep the per-texel gather into ONE tile of both sizes
#ifdef MIFLOW_EXPERIMENTS
            const bool coal = img_bytes > 0 && s >= s_coal;
#else
            constexpr bool coal = false;   // (the release library passes img_bytes = 0: one lattice tile)
#endif
            const int tile_cap = coal ? tile_bytes : tile_bytes + img_bytes;
            while (yb < 21 && ((int)floorf((float)yb * s + s) - dy_lo + 1) * nc <= tile_cap) ++yb;
            const int nr = (int)floorf((float)(yb - 1) * s + s) - dy_lo + 1;
            const int nblk = ((nr + 7) >> 3) * nbc;
#ifdef MIFLOW_EXPERIMENTS
            // (1) the image rows of the strip -> the image tile, row-coalesced
            const StripRows SR = strip_rows(w, dy_lo, dy_lo + nr - 1, -1, dxmax);
            const bool rows_fit = coal && SR.nrows <= kStripRowsMax;
            int pitch4 = 0;   // dwords per staged row
            if (rows_fit) {   // (workgroup-uniform)
                if (threadIdx.x == 0) s_pitch = 0;
                __syncthreads();
                for (int r = threadIdx.x; r < SR.nrows; r += 512) {
                    int xl, xr;
                    strip_row_range(w, dy_lo, dy_lo + nr - 1, -1, dxmax, SR.y0 + r, xl, xr);
                    // start the row's segment where its ADDRESS is dword aligned (the caller's matrix may be a ROI at any byte offset)
                    const int a = (int)((unsigned long long)(img + (long long)(SR.y0 + r) * istep + xl) & 3ull);
                    rowx0[r] = xl - a;
                    if (xr >= xl) atomicMax(&s_pitch, xr - (xl - a) + 1);
                }
                __syncthreads();
                pitch4 = (s_pitch + 3) >> 2;
            }
            const bool staged = rows_fit && pitch4 > 0 && (long long)SR.nrows * pitch4 * 4 <= (long long)img_bytes;   // workgroup-uniform
#else
            constexpr bool staged = false;
            (void)coal;
#endif
            if (staged) {
#ifdef MIFLOW_EXPERIMENTS
                unsigned *const it4 = reinterpret_cast<unsigned *>(itile);
                const int ndw = SR.nrows * pitch4;
                {
                    // kStageU dwords in flight per lane (a trip's row starts are read first, then its loads are issued together, then stored:
                    // one load per trip made the staging a chain of LDS + memory round trips, r19d: 687 against 900 frames/s)
                    int r = (int)threadIdx.x / pitch4, k = (int)threadIdx.x - r * pitch4;
                    const int dr = 512 / pitch4, dk = 512 - dr * pitch4;
                    for (int id = threadIdx.x; id < ndw; id += 512 * kStageU) {
                        const unsigned char *q[kStageU];
                        int x[kStageU];
                        bool ok[kStageU], fast[kStageU];
                        unsigned v[kStageU];
#pragma unroll
                        for (int u = 0; u < kStageU; ++u) {
                            ok[u] = id + 512 * u < ndw;
                            const int ru = ok[u] ? r : 0;
                            x[u] = rowx0[ru] + 4 * k;
                            q[u] = img + (long long)(SR.y0 + ru) * istep;
                            fast[u] = ok[u] && x[u] >= 0 && x[u] + 3 < cols;
                            r += dr; k += dk;
                            if (k >= pitch4) { k -= pitch4; ++r; }
                        }
#pragma unroll
                        for (int u = 0; u < kStageU; ++u)   // aligned by the choice of rowx0; a lane without a full dword inside the row loads row 0's first
                            v[u] = *reinterpret_cast<const unsigned *>(fast[u] ? q[u] + x[u] : img);
#pragma unroll
                        for (int u = 0; u < kStageU; ++u) {
                            if (ok[u] && !fast[u])   // the segment's head before column 0 / its tail beyond the last column: the clamped bytes (never looked up)
                                v[u] = (unsigned)q[u][clampi(x[u], 0, cols - 1)] | ((unsigned)q[u][clampi(x[u] + 1, 0, cols - 1)] << 8) |
                                       ((unsigned)q[u][clampi(x[u] + 2, 0, cols - 1)] << 16) | ((unsigned)q[u][clampi(x[u] + 3, 0, cols - 1)] << 24);
                            if (ok[u]) it4[id + 512 * u] = v[u];
                        }
                    }
                }
                __syncthreads();
                // (2) the lattice tile from the image tile: win_get's coordinates, the texel's byte from LDS (from global memory should the
                //     conservative ranges ever miss it); kGatherU points per trip so that their two dependent LDS reads overlap
                constexpr int kGatherU = 8;
                const int npt = nr * nc, pitch = pitch4 * 4;
                int rr = (int)threadIdx.x / nc, cc = (int)threadIdx.x - rr * nc;
                const int drr = 512 / nc, dcc = 512 - drr * nc;
                for (int ti = threadIdx.x; ti < npt; ti += 512 * kGatherU) {
                    int xs[kGatherU], ys[kGatherU], o[kGatherU];
                    bool ok[kGatherU], in[kGatherU];
                    unsigned char v[kGatherU];
#pragma unroll
                    for (int u = 0; u < kGatherU; ++u) {
                        ok[u] = ti + 512 * u < npt;
                        const int i = dy_lo + rr, j = cc - 1;
                        const float px = w.cx + (w.off + j) * w.c + (w.off + i) * w.s;
                        const float py = w.cy - (w.off + j) * w.s + (w.off + i) * w.c;
                        xs[u] = clampi(__float2int_rd(px), 0, w.cols - 1); ys[u] = clampi(__float2int_rd(py), 0, w.rows - 1);
                        rr += drr; cc += dcc;
                        if (cc >= nc) { cc -= nc; ++rr; }
                    }
#pragma unroll
                    for (int u = 0; u < kGatherU; ++u) {
                        const int ry = ys[u] - SR.y0;
                        const bool rin = (unsigned)ry < (unsigned)SR.nrows;
                        o[u] = xs[u] - rowx0[rin ? ry : 0];
                        in[u] = rin && (unsigned)o[u] < (unsigned)pitch;
                        o[u] = in[u] ? ry * pitch + o[u] : 0;
                    }
#pragma unroll
                    for (int u = 0; u < kGatherU; ++u) v[u] = itile[o[u]];
#pragma unroll
                    for (int u = 0; u < kGatherU; ++u) {
                        if (ok[u] && !in[u]) v[u] = w.img[(long long)ys[u] * w.step + xs[u]];
                        if (ok[u]) tile[ti + 512 * u] = v[u];
                    }
                }
#endif
            } else {
            // 8 x 8 blocks of the window lattice, block row / column carried along (no division per block); kStageU blocks per trip so that
            // their loads are in flight together (the per-texel gather of rounds 3-5: strips whose rows do not fit the image tile)
            int br = wv / nbc, bc = wv - br * nbc;
            for (int blk = wv; blk < nblk; blk += 8 * kStageU) {
                float v[kStageU];
                int ti[kStageU];
#pragma unroll
                for (int u = 0; u < kStageU; ++u) {
                    const int rr = br * 8 + ly, cc = bc * 8 + lx;
                    const bool in = blk + 8 * u < nblk && rr < nr && cc < nc;
                    ti[u] = in ? rr * nc + cc : -1;
                    v[u] = in ? win_get(w, dy_lo + rr, cc - 1) : 0.f;
                    bc += 8;
                    while (bc >= nbc) { bc -= nbc; ++br; }
                }
#pragma unroll
                for (int u = 0; u < kStageU; ++u)
                    if (ti[u] >= 0) tile[ti[u]] = (unsigned char)v[u];
            }
            }
            __syncthreads();
            const int nsmp = (yb - ya) * 21;
            if ((int)threadIdx.x < nsmp) {
                const int xl = threadIdx.x % 21, yl = ya + threadIdx.x / 21;
                P[yl][xl] = area_filter_lds(tile, nc, dy_lo, (float)xl, (float)yl, s, w.win);
            }
            __syncthreads();
            ya = yb;
        }
        desc_tail<EXT>(P, D, part, dw, desc + (long long)f * dstep);
        __syncthreads();   // (a workgroup that walks several features)
    }
}

// ------------------------------------------------------------------ octave 0 of the fused launch on an LDS tile (round 3)
// Three quarters of a frame's samples are octave 0, where the four (nOctaveLayers + 2 <= 4) layers of a sample read 4 x 32 taps from
// the same neighbourhood of the integral image.  A workgroup stages the (16 + 27) x (64 + 27) patch of a 16 x 64 sample tile ONCE
// (3.9 K words for 131 K taps) and every layer reads its taps from LDS: lanes are consecutive columns (bank-conflict free) and the tap
// offsets of octave 0 are COMPILE-TIME constants (sizes 9, 15, 21, 27 and the fixed patch stride), so a tap is a `ds_read_b32` with
// an immediate offset -- no address arithmetic.  Same integers, same box_div, same order: the planes stay bit-identical.
constexpr int kLdsTX = 64, kLdsTY = 16, kLdsSMax = 27, kLdsPW = 92, kLdsPH = kLdsTY + kLdsSMax;   // patch 43 rows x 92 words (91 used)
constexpr int kLdsLayers = 4;
__host__ __device__ constexpr int lds_rn(float v)   // round to nearest, ties to even (= __float2int_rn / rintf), v >= 0
{
    const int f = (int)v;
    const float r = v - (float)f;
    return r > 0.5f ? f + 1 : (r < 0.5f ? f : ((f & 1) ? f + 1 : f));
}
template <int L>
struct LdsGeo {   // geometry of octave 0, layer L for the patch stride: the compile-time twin of haar_geo(9 + 6 L, kLdsPW)
    static constexpr int size = 9 + 6 * L;
    static constexpr float ratio = (float)size / 9;
    static constexpr int e(int c) { return lds_rn(ratio * (float)c); }
    static constexpr int xx(int i, int j) { return e(j ? 7 : 2) * kLdsPW + e(3 * i); }
    static constexpr int yy(int j, int i) { return e(3 * i) * kLdsPW + e(j ? 7 : 2); }
    static constexpr int xy(int i, int j) { return e(i == 0 ? 1 : i == 1 ? 4 : i == 2 ? 5 : 8) * kLdsPW + e(j == 0 ? 1 : j == 1 ? 4 : j == 2 ? 5 : 8); }
    static constexpr double axx(int k) { return (double)((e(3 * k + 3) - e(3 * k)) * (e(7) - e(2))); }
    static constexpr double a6 = (double)((e(4) - e(1)) * (e(4) - e(1))), a7 = (double)((e(8) - e(5)) * (e(4) - e(1))), a9 = (double)((e(8) - e(5)) * (e(8) - e(5)));
};
// host check that the compile-time geometry is haar_geo's (surf_api.cpp calls it once; tests/test_surf.py through the C-ABI self-test)
template <int L>
static bool lds_geo_matches()
{
    typedef LdsGeo<L> G;
    const HaarGeo h = haar_geo(G::size, kLdsPW);
    bool ok = true;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j) ok = ok && h.xx[i][j] == G::xx(i, j) && h.yy[j][i] == G::yy(j, i);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) ok = ok && h.xy[i][j] == G::xy(i, j);
    for (int k = 0; k < 3; ++k) ok = ok && h.area[k] == G::axx(k);
    return ok && h.area[6] == G::a6 && h.area[7] == G::a7 && h.area[9] == G::a9;
}
bool lds_geometry_self_check() { return lds_geo_matches<0>() && lds_geo_matches<1>() && lds_geo_matches<2>() && lds_geo_matches<3>(); }

// Dxx, Dyy, Dxy of one sample of octave 0, layer L from the staged patch; q = the LDS word of the sample's top-left corner
template <int L>
__device__ __forceinline__ void haar_det_trace_lds(const unsigned *q, float &d, float &tr)
{
    typedef LdsGeo<L> G;
    unsigned cxx[4][2], cyy[2][4], cxy[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { cxx[a][b] = q[G::xx(a, b)]; cyy[b][a] = q[G::yy(b, a)]; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) cxy[a][b] = q[G::xy(a, b)];
    const double wxx[3] = {1.0, -2.0, 1.0};
    double sx = 0, sy = 0, sxy = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        sx += box_div(cxx[k][0] - cxx[k][1] - cxx[k + 1][0] + cxx[k + 1][1], wxx[k], G::axx(k), 1.0 / G::axx(k));
        sy += box_div(cyy[0][k] - cyy[0][k + 1] - cyy[1][k] + cyy[1][k + 1], wxx[k], G::axx(k), 1.0 / G::axx(k));
    }
    sxy += box_div(cxy[0][0] - cxy[1][0] - cxy[0][1] + cxy[1][1], 1.0, G::a6, 1.0 / G::a6);
    sxy += box_div(cxy[0][2] - cxy[1][2] - cxy[0][3] + cxy[1][3], -1.0, G::a7, 1.0 / G::a7);
    sxy += box_div(cxy[2][0] - cxy[3][0] - cxy[2][1] + cxy[3][1], -1.0, G::a7, 1.0 / G::a7);
    sxy += box_div(cxy[2][2] - cxy[3][2] - cxy[2][3] + cxy[3][3], 1.0, G::a9, 1.0 / G::a9);
    const float dx = (float)sx, dy = (float)sy, dxy = (float)sxy;
    d = dx * dy - 0.81f * dxy * dxy;
    tr = dx + dy;
}
// The same sample evaluated from the integral image itself (stride sld): p = the word of the sample's top-left corner.  The taps are
// LdsGeo<L>'s edges with the image's stride instead of the patch's -- same integers, same box_div, same order: the same bits.
template <int L>
__device__ __forceinline__ void haar_det_trace_g0(const unsigned *p, int sld, float &d, float &tr)
{
    typedef LdsGeo<L> G;
    const auto Q = [&](int ey, int ex) { return p[ey * sld + ex]; };
    const auto E4 = [](int i) { return G::e(i == 0 ? 1 : i == 1 ? 4 : i == 2 ? 5 : 8); };
    unsigned cxx[4][2], cyy[2][4], cxy[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { cxx[a][b] = Q(G::e(b ? 7 : 2), G::e(3 * a)); cyy[b][a] = Q(G::e(3 * a), G::e(b ? 7 : 2)); }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) cxy[a][b] = Q(E4(a), E4(b));
    const double wxx[3] = {1.0, -2.0, 1.0};
    double sx = 0, sy = 0, sxy = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        sx += box_div(cxx[k][0] - cxx[k][1] - cxx[k + 1][0] + cxx[k + 1][1], wxx[k], G::axx(k), 1.0 / G::axx(k));
        sy += box_div(cyy[0][k] - cyy[0][k + 1] - cyy[1][k] + cyy[1][k + 1], wxx[k], G::axx(k), 1.0 / G::axx(k));
    }
    sxy += box_div(cxy[0][0] - cxy[1][0] - cxy[0][1] + cxy[1][1], 1.0, G::a6, 1.0 / G::a6);
    sxy += box_div(cxy[0][2] - cxy[1][2] - cxy[0][3] + cxy[1][3], -1.0, G::a7, 1.0 / G::a7);
    sxy += box_div(cxy[2][0] - cxy[3][0] - cxy[2][1] + cxy[3][1], -1.0, G::a7, 1.0 / G::a7);
    sxy += box_div(cxy[2][2] - cxy[3][2] - cxy[2][3] + cxy[3][3], 1.0, G::a9, 1.0 / G::a9);
    const float dx = (float)sx, dy = (float)sy, dxy = (float)sxy;
    d = dx * dy - 0.81f * dxy * dxy;
    tr = dx + dy;
}
template <int L>
__device__ __forceinline__ float det0_layer(const SumTex &t, int ii, int jj)
{
    constexpr int size = 9 + 6 * L, m = size >> 1;
    const int samples_i = 1 + (t.rows - size), samples_j = 1 + (t.cols - size);
    const int i = ii - m, j = jj - m;
    float d = 0.f, tr = 0.f;
    if (size <= t.rows && size <= t.cols && i >= 0 && j >= 0 && i < samples_i && j < samples_j)
        haar_det_trace_g0<L>(t.s + (long long)i * t.sld + j, t.sld, d, tr);
    return d;
}
__device__ float det0_at(const SumTex &t, int layer, int ii, int jj)
{
    switch (layer) {
    case 0: return det0_layer<0>(t, ii, jj);
    case 1: return det0_layer<1>(t, ii, jj);
    case 2: return det0_layer<2>(t, ii, jj);
    default: return det0_layer<3>(t, ii, jj);
    }
}

// ---- octave 0 with the maxima flagged INSIDE the det kernel (round 5; surf.cu:205-222 feeding :263-355 without the planes): a
// workgroup's 16 x 64 tile of det values, all layers, stays in LDS; the 26 strict comparisons of the 14 x 62 interior samples of the
// middle layers read their neighbours there, and what leaves the kernel is the flag word of each (layer, row) -- OR-ed into the
// 64-column chunk words the scan and the candidate writer expect (a tile's 62 columns straddle two of them) -- with the sign of the
// trace at each maximum beside it.  Tiles overlap by one sample on every side (stride 14 x 62): 18 % more samples evaluated, and the
// 265 MB of octave-0 det / trace planes of a 4K frame are neither written nor read again.  The sub-pixel refinement evaluates its 27
// values from the integral image (det0_at).
constexpr int kFuseTY = kLdsTY - 2, kFuseTX = kLdsTX - 2;
struct Fuse0Args {
    float thr;
    SumTex mask;                       // mask.s == nullptr: none
    unsigned long long *bits, *sbits;  // octave 0's regions (offset 0), zeroed by the caller
    unsigned *segcnt;
    int chunks, nseg;
};
template <int L>
__device__ __forceinline__ void fuse_layer(const SumTex &t, const unsigned *patch, int ii0, int jj0, int w4, int lane, float (*detT)[kLdsTY][kLdsTX],
                                           unsigned long long (*sgn)[kLdsTY])
{
    constexpr int size = 9 + 6 * L, m = size >> 1, mmax = kLdsSMax >> 1;
    const int samples_i = 1 + (t.rows - size), samples_j = 1 + (t.cols - size);
    const int j = jj0 + lane - m;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int il = w4 * 4 + r4, i = ii0 + il - m;
        float d = 0.f, tr = 0.f;
        if (size <= t.rows && size <= t.cols && i >= 0 && j >= 0 && i < samples_i && j < samples_j)
            haar_det_trace_lds<L>(patch + (il + mmax - m) * kLdsPW + (lane + mmax - m), d, tr);
        detT[L][il][lane] = d;
        const unsigned long long sm = __ballot((__float_as_uint(tr) >> 31) != 0u);
        if (lane == 0) sgn[L][il] = sm;
    }
}
__device__ __forceinline__ void fuse_nms(const SumTex &t, const Fuse0Args &F, int nlayers, int ii0, int jj0, int w4, int lane,
                                         const float (*detT)[kLdsTY][kLdsTX], const unsigned long long (*sgn)[kLdsTY])
{
    const int lm = max(lane - 1, 0), lp = min(lane + 1, kLdsTX - 1);
    const int jj = jj0 + lane;
    for (int q = w4; q < nlayers * kFuseTY; q += 4) {
        const int layer = q / kFuseTY + 1, il = q % kFuseTY + 1, ii = ii0 + il;
        if (ii >= t.rows) continue;   // wave-uniform
        const int size = calc_size(0, layer);
        const int margin = (calc_size(0, layer + 1) >> 1) + 1;
        const float v = detT[layer][il][lane];
        bool ismax = lane >= 1 && lane <= kFuseTX && ii >= margin && ii < t.rows - margin && jj >= margin && jj < t.cols - margin && v > F.thr;
        if (ismax && F.mask.s) ismax = mask_check(F.mask, ii - (size >> 1), jj - (size >> 1), size);
        if (__ballot(ismax) != 0ull) {
            if (ismax) {
                const float *c0 = detT[layer][il];
                ismax = v > c0[lm] && v > c0[lp];
#pragma unroll
                for (int di = -1; di <= 1; di += 2) { const float *c = detT[layer][il + di]; ismax = ismax && v > c[lm] && v > c[lane] && v > c[lp]; }
#pragma unroll
                for (int dl = -1; dl <= 1; dl += 2)
#pragma unroll
                    for (int di = -1; di <= 1; ++di) { const float *c = detT[layer + dl][il + di]; ismax = ismax && v > c[lm] && v > c[lane] && v > c[lp]; }
            }
        }
        const unsigned long long m = __ballot(ismax);
        if (m == 0ull || lane != 0) continue;
        // lane k of the tile row = column jj0 + k (jj0 >= -1): chunk words c_lo (bits from s on) and c_lo + 1
        const int c_lo = ((jj0 + 64) >> 6) - 1, s = jj0 - 64 * c_lo;
        const long long r = (long long)(layer - 1) * t.rows + ii;
        const unsigned long long sg = sgn[layer][il] & m;
        const unsigned long long lo = m << s, hi = s ? m >> (64 - s) : 0ull;
        if (lo && c_lo >= 0) {
            atomicOr(F.bits + r * F.chunks + c_lo, lo);
            if (sg << s) atomicOr(F.sbits + r * F.chunks + c_lo, sg << s);
            atomicAdd(F.segcnt + r * F.nseg + c_lo / kNmsSeg, (unsigned)__popcll(lo));
        }
        if (hi && c_lo + 1 < F.chunks) {
            atomicOr(F.bits + r * F.chunks + c_lo + 1, hi);
            if (s && (sg >> (64 - s))) atomicOr(F.sbits + r * F.chunks + c_lo + 1, sg >> (64 - s));
            atomicAdd(F.segcnt + r * F.nseg + (c_lo + 1) / kNmsSeg, (unsigned)__popcll(hi));
        }
    }
}

// one layer of the tile: the four sample rows of this wave
template <int L>
__device__ __forceinline__ void lds_layer(const SumTex &t, const unsigned *patch, int ii0, int jj, int w4, int lane, float *det, float *trace, long long plane0,
                                          int layer_rows, int layer_cols, int dld)
{
    constexpr int size = 9 + 6 * L, m = size >> 1, mmax = kLdsSMax >> 1;
    const int samples_i = 1 + (t.rows - size), samples_j = 1 + (t.cols - size);
    const int j = jj - m;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int il = w4 * 4 + r4, ii = ii0 + il, i = ii - m;
        float d = 0.f, tr = 0.f;
        if (size <= t.rows && size <= t.cols && i >= 0 && j >= 0 && i < samples_i && j < samples_j)
            haar_det_trace_lds<L>(patch + (il + mmax - m) * kLdsPW + (lane + mmax - m), d, tr);
        if (jj < layer_cols && ii < layer_rows) {
            const long long o = plane0 + (long long)(L * layer_rows + ii) * dld + jj;
            det[o] = d;
            trace[o] = tr;
        }
    }
}

// ------------------------------------------------------------------ all octaves of a frame in one launch per stage (round 3)
// The reference runs its five detector kernels once per octave (surf.cuda.cpp:182-204) and copies two counters to the host in
// between; round 2 kept the per-octave launches (24 per 4-octave frame, the coarse octaves launch-bound: octave 3 is 1/64 of octave
// 0's samples and took 1/6 of its time).  Here every stage covers all octaves: the planes, flag words, counts and candidate lists of
// an octave live at their own offsets (OctSet), a workgroup finds its octave from the cumulative workgroup counts, and the last
// stage walks the octaves in order so that the features of octave o still follow those of octave o - 1 (deterministic order).
struct OctSet {
    int n;                                   // octaves
    int nlayers;                             // nOctaveLayers
    int rows, cols, dld;
    long long plane0[kMaxFusedOctaves];      // float offset of the octave's first det / trace plane
    long long bits0[kMaxFusedOctaves];       // u64 offset of its flag words
    int row0[kMaxFusedOctaves];              // first (layer, row) index in rowcnt space (rowcnt has one extra entry per octave)
    long long seg0[kMaxFusedOctaves];        // offset of its row-segment counts
    int blk_dt[kMaxFusedOctaves + 1];        // cumulative workgroup counts of k_det_trace_all
    int blk_nms[kMaxFusedOctaves + 1];       //   ... of k_nms_flag_all (row groups x segments)
    int blk_wr[kMaxFusedOctaves + 1];        //   ... of k_nms_write_all (row groups)
    int nbx[kMaxFusedOctaves], nby[kMaxFusedOctaves], nseg[kMaxFusedOctaves], chunks[kMaxFusedOctaves];
    int lds0;                                // octave 0 of k_det_trace_all on LDS tiles (all its layers per workgroup): nby[0] counts 16-row tiles
    int poly;                                // octaves >= 1 read their taps from the polyphase planes (pld / pbase per octave; geometry table built for them)
    int pld[kMaxFusedOctaves];
    long long pbase[kMaxFusedOctaves];
    int fuse0;                               // ... and its maxima flagged in that kernel (no planes; tiles of 14 x 62 interior samples; no k_nms_flag_all workgroups)
};
__device__ __forceinline__ int find_octave(const int *cum, int n, int id)
{
    int o = 0;
#pragma unroll
    for (int k = 1; k < kMaxFusedOctaves; ++k) o += (k < n && id >= cum[k]) ? 1 : 0;
    return o;
}
// Octave 0 with its maxima flagged in the kernel (fuse0): its own launch, so that the 32 KB of LDS a tile needs do not cost the
// latency-bound global-tap workgroups of the other octaves their occupancy (r14i: in one launch the frame was 8 % slower).
__global__ __launch_bounds__(256) void k_det_nms0(SumTex t, OctSet S, Fuse0Args F)
{
    const int nbx = S.nbx[0], nl2 = S.nlayers + 2;
    const unsigned q = (unsigned)blockIdx.x, nwg = (unsigned)S.blk_dt[1];   // nwg % 8 == 0; XCD-contiguous tile order as in k_det_trace_all
    const int loc = (int)((q & 7u) * (nwg >> 3) + (q >> 3));
    __shared__ unsigned patch[kLdsPH * kLdsPW];
    {
        __shared__ float detT[kLdsLayers][kLdsTY][kLdsTX];
        __shared__ unsigned long long sgn[kLdsLayers][kLdsTY];
        if (loc >= nbx * S.nby[0]) return;   // padding (the whole workgroup)
        const int bx = __builtin_amdgcn_readfirstlane(loc % nbx), by = __builtin_amdgcn_readfirstlane(loc / nbx);
        const int lane = threadIdx.x & 63, w4 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int ii0 = by * kFuseTY - 1, jj0 = bx * kFuseTX - 1, mmax = kLdsSMax >> 1;
        for (int r = w4; r < kLdsPH; r += 4) {
            const unsigned *row = t.s + (long long)min(max(ii0 - mmax + r, 0), t.rows) * t.sld;
            for (int c = lane; c < kLdsPW; c += 64) patch[r * kLdsPW + c] = row[min(max(jj0 - mmax + c, 0), t.cols)];
        }
        __syncthreads();
        fuse_layer<0>(t, patch, ii0, jj0, w4, lane, detT, sgn);
        fuse_layer<1>(t, patch, ii0, jj0, w4, lane, detT, sgn);
        if (nl2 > 2) fuse_layer<2>(t, patch, ii0, jj0, w4, lane, detT, sgn);
        if (nl2 > 3) fuse_layer<3>(t, patch, ii0, jj0, w4, lane, detT, sgn);
        __syncthreads();
        fuse_nms(t, F, S.nlayers, ii0, jj0, w4, lane, detT, sgn);
        return;
    }
}

// (blk0: workgroups of octave 0 that run in k_det_nms0 instead -- fuse0 -- and are not part of this launch's grid)
__global__ __launch_bounds__(256) void k_det_trace_all(SumTex t, float *det, float *trace, OctSet S, const HaarGeo *geo, int blk0, const unsigned *poly)
{
    // Everything derived from the workgroup id is wave-uniform.  The octave comes from the ORIGINAL id (every octave's range is padded
    // to a multiple of 8 workgroups), the XCD-contiguous remap is applied INSIDE the octave: a sample of octave o costs up to 10 x one of
    // octave 0 (its 64 lanes read taps 4 << o bytes apart: 4 .. 32 cache lines per wave load), so an order that hands whole octaves
    // to single XCDs leaves the other six idle (r03o / r03p: 1 285 us against 395 us for the four per-octave launches)
    const int orig = blockIdx.x + blk0;
    const int octave = __builtin_amdgcn_readfirstlane(find_octave(S.blk_dt, S.n, orig));
    const int nbx = S.nbx[octave], nl2 = S.nlayers + 2;
    const unsigned q = (unsigned)(orig - S.blk_dt[octave]), nwg = (unsigned)(S.blk_dt[octave + 1] - S.blk_dt[octave]);   // nwg % 8 == 0
    const int loc = (int)((q & 7u) * (nwg >> 3) + (q >> 3));
    __shared__ unsigned patch[kLdsPH * kLdsPW];
    if (octave == 0 && S.lds0) {
        if (loc >= nbx * S.nby[0]) return;   // padding (the whole workgroup)
        const int bx = __builtin_amdgcn_readfirstlane(loc % nbx), by = __builtin_amdgcn_readfirstlane(loc / nbx);
        const int lane = threadIdx.x & 63, w4 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int ii0 = by * kLdsTY, jj0 = bx * kLdsTX, mmax = kLdsSMax >> 1;
        // stage the patch: sum-image rows ii0 - 13 .. ii0 + 29, columns jj0 - 13 .. jj0 + 77, clamped into the table (what lies outside is
        // read by samples that are not valid and are written as 0)
        for (int r = w4; r < kLdsPH; r += 4) {
            const unsigned *row = t.s + (long long)min(max(ii0 - mmax + r, 0), t.rows) * t.sld;
            for (int c = lane; c < kLdsPW; c += 64) patch[r * kLdsPW + c] = row[min(max(jj0 - mmax + c, 0), t.cols)];
        }
        __syncthreads();
        const int jj = jj0 + lane;
        lds_layer<0>(t, patch, ii0, jj, w4, lane, det, trace, S.plane0[0], t.rows, t.cols, S.dld);
        lds_layer<1>(t, patch, ii0, jj, w4, lane, det, trace, S.plane0[0], t.rows, t.cols, S.dld);
        if (nl2 > 2) lds_layer<2>(t, patch, ii0, jj, w4, lane, det, trace, S.plane0[0], t.rows, t.cols, S.dld);
        if (nl2 > 3) lds_layer<3>(t, patch, ii0, jj, w4, lane, det, trace, S.plane0[0], t.rows, t.cols, S.dld);
        return;
    }
    if (loc >= nbx * S.nby[octave] * nl2) return;   // padding
    const int bx = __builtin_amdgcn_readfirstlane(loc % nbx), layer = __builtin_amdgcn_readfirstlane((loc / nbx) % nl2),
              by = __builtin_amdgcn_readfirstlane(loc / (nbx * nl2));
    const int layer_rows = t.rows >> octave, layer_cols = t.cols >> octave;
    const int jj = bx * 64 + (threadIdx.x & 63);
    const int ii = by * 4 + (threadIdx.x >> 6);
    if (jj >= layer_cols || ii >= layer_rows) return;
    const int size = calc_size(octave, layer);
    const int samples_i = 1 + ((t.rows - size) >> octave), samples_j = 1 + ((t.cols - size) >> octave);
    const int margin = (size >> 1) >> octave;
    const int i = ii - margin, j = jj - margin;
    float d = 0.f, tr = 0.f;
    if (size <= t.rows && size <= t.cols && i >= 0 && j >= 0 && i < samples_i && j < samples_j) {
        // read-only for the life of the launch and wave-uniform: through the constant address space = scalar loads (s_load_dwordx16)
        typedef const __attribute__((address_space(4))) HaarGeo cgeo_t;
        cgeo_t &g = *((cgeo_t *)(unsigned long long)geo + (octave * kDetLayers + layer));
        if (S.poly && octave >= 1) {   // taps from the octave's phase planes: consecutive lanes read consecutive words (octave 0 without its LDS tiles keeps the image itself: stride 1 already)
            SumTex tp = t;
            tp.s = poly + S.pbase[octave];
            unsigned voff = 4u * ((unsigned)i * (unsigned)S.pld[octave] + (unsigned)j);
            asm volatile("" : "+v"(voff));
            haar_det_trace(tp, g, voff, d, tr);
        } else {
        unsigned voff = 4u * ((unsigned)(i << octave) * (unsigned)t.sld + (unsigned)(j << octave));
        asm volatile("" : "+v"(voff));
        haar_det_trace(t, g, voff, d, tr);
        }
    }
    const long long o = S.plane0[octave] + (long long)(layer * layer_rows + ii) * S.dld + jj;
    det[o] = d;
    trace[o] = tr;
}

// the octave's view of the shared argument block
__device__ __forceinline__ NmsArgs oct_args(const NmsArgs &B, const OctSet &S, int octave)
{
    NmsArgs A = B;
    A.octave = octave;
    A.det = B.det + S.plane0[octave]; A.trace = B.trace + S.plane0[octave];
    A.bits = B.bits + S.bits0[octave];
    A.sbits = (octave == 0 && S.fuse0) ? B.sbits : nullptr;
    A.rowcnt = B.rowcnt + S.row0[octave] + octave;            // one extra entry (the total) per octave
    A.segcnt = B.segcnt + S.seg0[octave];
    A.chunks = S.chunks[octave]; A.nseg = S.nseg[octave];
    return A;
}
__global__ __launch_bounds__(256) void k_nms_flag_all(NmsArgs B, OctSet S)
{
    const int octave = __builtin_amdgcn_readfirstlane(find_octave(S.blk_nms, S.n, blockIdx.x));
    const int loc = blockIdx.x - S.blk_nms[octave], nseg = S.nseg[octave];
    nms_flag_row(oct_args(B, S, octave), (loc / nseg) * 4 + (threadIdx.x >> 6), loc % nseg);
}
__global__ __launch_bounds__(1024) void k_scan_counts_all(NmsArgs B, OctSet S)
{
    const int octave = blockIdx.x;
    const NmsArgs A = oct_args(B, S, octave);
    scan_counts_body(A.rowcnt, A.segcnt, A.nseg, A.nlayers * (A.rows >> octave));
}
__global__ __launch_bounds__(256) void k_nms_write_all(NmsArgs B, OctSet S, int4 *cand, int max_candidates, unsigned *ncand)
{
    const int octave = __builtin_amdgcn_readfirstlane(find_octave(S.blk_wr, S.n, blockIdx.x));
    nms_write_row(oct_args(B, S, octave), (blockIdx.x - S.blk_wr[octave]) * 4 + (threadIdx.x >> 6), cand + (long long)octave * max_candidates,
                  max_candidates, ncand + octave);
}
__global__ __launch_bounds__(256) void k_interp_eval_all(const float *det, OctSet S, const int4 *cand, const unsigned *ncand, InterpOut *tmp,
                                                         int max_candidates, unsigned *okcnt, SumTex t0)
{
    const int octave = blockIdx.y;
    const bool ok = interp_eval_one(det + S.plane0[octave], S.dld, S.rows, S.cols, octave, cand + (long long)octave * max_candidates, ncand + octave,
                                    tmp + (long long)octave * max_candidates, blockIdx.x * 256 + threadIdx.x, (octave == 0 && S.fuse0) ? &t0 : nullptr);
    // accepted candidates of the octave (one atomic per wave): k_interp_compact_all's workgroup of octave o starts at the sum over o' < o
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(okcnt + octave, (unsigned)__popcll(m));
}
// One workgroup per octave (round 3; one workgroup walked the octaves in turn: 33 us per 4K frame).  The features of octave o follow
// those of octave o - 1: its first index is the number of accepted candidates of the octaves before it (okcnt, counted by
// k_interp_eval_all); the feature counter itself starts at 0 (the caller zeroes it) and is written by the last octave's workgroup.
__global__ __launch_bounds__(1024) void k_interp_compact_all(const InterpOut *tmp, const unsigned *ncand, int n_octaves, int max_candidates,
                                                             float *kp, int kld, int max_features, unsigned *nfeat_p, const unsigned *okcnt)
{
    const int octave = blockIdx.x;
    unsigned nfeat0 = 0;
    for (int o = 0; o < octave; ++o) nfeat0 += okcnt[o];
    const unsigned nfeat = interp_compact_body(tmp + (long long)octave * max_candidates, ncand + octave, octave, kp, kld, max_features, nfeat0);
    if (octave == n_octaves - 1 && threadIdx.x == 0) *nfeat_p = nfeat;
}

// ------------------------------------------------------------------ host launchers
int integral(const unsigned char *img, long long istep, int rows, int cols, bool clamp1, unsigned *V, unsigned *BT, int vld,
             unsigned *sum, int sld, hipStream_t s)
{
    const int band_rows = 32, nbands = div_up(rows, band_rows);
    hipLaunchKernelGGL(k_int_cols, dim3(div_up(cols, 256), nbands), dim3(256), 0, s, img, istep, rows, cols, clamp1 ? 1 : 0, V, vld, BT, band_rows);
    hipLaunchKernelGGL(k_int_bands, dim3(div_up(cols, 256)), dim3(256), 0, s, BT, vld, cols, nbands);
    const int cpw = div_up(div_up(cols, 64), kRowWaves);
    if (cpw <= kRowCpw)
        hipLaunchKernelGGL(k_int_rows_wide, dim3(rows + 1), dim3(64 * kRowWaves), 0, s, V, BT, vld, rows, cols, band_rows, sum, sld, cpw);
    else
        hipLaunchKernelGGL(k_int_rows, dim3(div_up(rows + 1, 4)), dim3(256), 0, s, V, BT, vld, rows, cols, band_rows, sum, sld);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}
int integral_bands(int rows) { return div_up(rows, 32); }

int det_trace(const unsigned *sum, int sld, int rows, int cols, int octave, int nOctaveLayers, float *det, float *trace, int dld,
              hipStream_t s)
{
    SumTex t = {sum, sld, rows, cols};
    const int lr = rows >> octave, lc = cols >> octave;
    const int nbx = div_up(lc, 64), nby = div_up(lr, 4);
    for (int l0 = 0; l0 < nOctaveLayers + 2; l0 += kDetLayers) {
        const int nl = std::min(kDetLayers, nOctaveLayers + 2 - l0);
        HaarGeoSet G;
        memset(&G, 0, sizeof(G));
        for (int l = 0; l < nl; ++l) G.l[l] = haar_geo(calc_size(octave, l0 + l), sld);
        hipLaunchKernelGGL(k_det_trace, dim3(nbx * nby * nl), dim3(256), 0, s, t, det, trace, dld, octave, l0, nl, nbx, nby, G);
    }
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int nms_segments(int cols) { return div_up(div_up(cols, 64), kNmsSeg); }
int find_maxima(const float *det, const float *trace, int dld, const unsigned *mask_sum, int sld, int rows, int cols, int octave,
                int nOctaveLayers, float thr, unsigned long long *bits, unsigned *rowcnt, unsigned *segcnt, int4 *cand, int max_candidates,
                unsigned *ncand, hipStream_t s)
{
    NmsArgs A;
    A.det = det; A.trace = trace; A.dld = dld; A.rows = rows; A.cols = cols; A.octave = octave; A.nlayers = nOctaveLayers; A.thr = thr;
    A.mask.s = mask_sum; A.mask.sld = sld; A.mask.rows = rows; A.mask.cols = cols;
    A.bits = bits; A.rowcnt = rowcnt;
    const int lr = rows >> octave, lc = cols >> octave;
    A.chunks = div_up(lc, 64);
    A.segcnt = segcnt; A.nseg = div_up(A.chunks, kNmsSeg);
    const int nrows = nOctaveLayers * lr;
    hipLaunchKernelGGL(k_nms_flag, dim3(div_up(nrows, 4), A.nseg), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, rowcnt, (const unsigned *)segcnt, A.nseg, nrows);
    hipLaunchKernelGGL(k_nms_write, dim3(div_up(nrows, 4)), dim3(256), 0, s, A, cand, max_candidates, ncand);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int interpolate(const float *det, int dld, int rows, int cols, int octave, const int4 *cand, const unsigned *ncand, int max_candidates,
                void *tmp, float *kp, int kld, int max_features, unsigned *nfeat, hipStream_t s)
{
    hipLaunchKernelGGL(k_interp_eval, dim3(div_up(max_candidates, 256)), dim3(256), 0, s, det, dld, rows, cols, octave, cand, ncand,
                       (InterpOut *)tmp);
    hipLaunchKernelGGL(k_interp_compact, dim3(1), dim3(1024), 0, s, (const InterpOut *)tmp, ncand, octave, kp, kld, max_features, nfeat);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}
size_t interp_tmp_bytes(int max_candidates) { return sizeof(InterpOut) * (size_t)max_candidates; }

// ---- all octaves per launch (k_*_all).  Sizes of the per-octave regions for a frame of rows x cols (the handle allocates them):
// lds: octave 0 of the det / trace launch on LDS tiles (the HANDLE's decision, MIFLOW_SURF_LDS=0 switches it off: no process-global state)
static OctSet make_octset(int rows, int cols, int dld, int n_octaves, int nOctaveLayers, int lds)
{
    OctSet S;
    memset(&S, 0, sizeof(S));
    S.n = n_octaves; S.nlayers = nOctaveLayers; S.rows = rows; S.cols = cols; S.dld = dld;
    S.lds0 = (nOctaveLayers + 2 <= kLdsLayers && (lds & 3)) ? 1 : 0;
    S.fuse0 = (S.lds0 && (lds & 3) >= 2) ? 1 : 0;   // lds & 3 = 2: octave 0's maxima flagged inside the det kernel
    S.poly = (lds & 4) && n_octaves > 1 ? 1 : 0;     // lds & 4: octaves >= 1 on the polyphase planes
    {
        long long base = 0;
        for (int o = 1; o < n_octaves; ++o) {
            const PolyGeo pg = poly_geo(rows, cols, o, base);
            S.pld[o] = pg.pld; S.pbase[o] = base;
            base += pg.plane_words << (2 * o);
        }
    }
    long long plane = 0, bits = 0, seg = 0;
    int row = 0;
    for (int o = 0; o < n_octaves; ++o) {
        const int lr = rows >> o, lc = cols >> o;
        S.plane0[o] = plane; S.bits0[o] = bits; S.row0[o] = row; S.seg0[o] = seg;
        S.chunks[o] = div_up(lc, 64); S.nseg[o] = div_up(S.chunks[o], kNmsSeg);
        S.nbx[o] = div_up(lc, 64); S.nby[o] = div_up(lr, 4);
        if (o == 0 && S.fuse0) {   // one workgroup per 14 x 62 tile of interior samples (16 x 64 evaluated) and ALL layers
            S.nbx[0] = div_up(lc, kFuseTX); S.nby[0] = div_up(lr, kFuseTY);
            S.blk_dt[1] = align_up(S.nbx[0] * S.nby[0], 8);
        } else if (o == 0 && S.lds0) {   // one workgroup per 16 x 64 tile and ALL layers
            S.nby[0] = div_up(lr, kLdsTY);
            S.blk_dt[1] = align_up(S.nbx[0] * S.nby[0], 8);
        } else
        S.blk_dt[o + 1] = S.blk_dt[o] + align_up(S.nbx[o] * S.nby[o] * (nOctaveLayers + 2), 8);   // padded: see k_det_trace_all
        S.blk_nms[o + 1] = S.blk_nms[o] + ((o == 0 && S.fuse0) ? 0 : div_up(nOctaveLayers * lr, 4) * S.nseg[o]);
        S.blk_wr[o + 1] = S.blk_wr[o] + div_up(nOctaveLayers * lr, 4);
        plane += (long long)(nOctaveLayers + 2) * lr * dld;
        bits += (long long)nOctaveLayers * lr * S.chunks[o];
        seg += (long long)nOctaveLayers * lr * S.nseg[o];
        row += nOctaveLayers * lr;
    }
    return S;
}
bool fused_supported(int n_octaves, int nOctaveLayers) { return n_octaves <= kMaxFusedOctaves && nOctaveLayers + 2 <= kDetLayers; }
void fused_sizes(int rows, int cols, int dld, int n_octaves, int nOctaveLayers, FusedSizes *z)
{
    const OctSet S = make_octset(rows, cols, dld, n_octaves, nOctaveLayers, 0);   // the region sizes do not depend on the octave-0 path
    const int last = n_octaves - 1, lr = rows >> last;
    z->plane_floats = (size_t)(S.plane0[last] + (long long)(nOctaveLayers + 2) * lr * dld);
    z->bits_words = (size_t)(S.bits0[last] + (long long)nOctaveLayers * lr * S.chunks[last]);
    z->seg_counts = (size_t)(S.seg0[last] + (long long)nOctaveLayers * lr * S.nseg[last]);
    z->row_counts = (size_t)(S.row0[last] + nOctaveLayers * lr + n_octaves);
    z->geo_bytes = sizeof(HaarGeo) * (size_t)n_octaves * kDetLayers;
    z->poly_words = (size_t)poly_total_words(rows, cols, n_octaves);
}
// geometry of every (octave, layer) of the frame size: uploaded by the handle when the size changes
void fused_geometry(int sld, int n_octaves, int nOctaveLayers, void *geo_host, int rows, int cols, bool poly)
{
    HaarGeo *g = (HaarGeo *)geo_host;
    memset(g, 0, sizeof(HaarGeo) * (size_t)n_octaves * kDetLayers);
    for (int o = 0; o < n_octaves; ++o)
        for (int l = 0; l < nOctaveLayers + 2; ++l) {
            if (poly && o >= 1) {   // tap (ey, ex) of octave o = phase plane (ey & m, ex & m), position shifted by (ey >> o, ex >> o)
                const PolyGeo pg = poly_geo(rows, cols, o, 0);
                const int m = (1 << o) - 1;
                g[o * kDetLayers + l] = haar_geo_off(calc_size(o, l), [&](int ey, int ex) {
                    return (int)((long long)(((ey & m) << o) + (ex & m)) * pg.plane_words + (long long)(ey >> o) * pg.pld + (ex >> o));
                });
            } else g[o * kDetLayers + l] = haar_geo(calc_size(o, l), sld);
        }
}
// surf.cuda.cpp:182-204 for all octaves: six launches.  ncand: n_octaves counters; nfeat: the feature counter (zeroed by the caller)
int detect_fused(const unsigned *sum, const unsigned *mask_sum, int sld, int rows, int cols, int n_octaves, int nOctaveLayers, float thr,
                 float *det, float *trace, int dld, unsigned long long *bits, unsigned *rowcnt, unsigned *segcnt, int4 *cand, int max_candidates,
                 unsigned *ncand, void *tmp, const void *geo_dev, float *kp, int kld, int max_features, unsigned *nfeat, int lds_tiles, hipStream_t s,
                 unsigned long long *sbits, unsigned *poly)
{
    int ldsf = lds_tiles;
    if ((ldsf & 3) >= 2 && !sbits) ldsf = (ldsf & ~3) | 1;
    if (!poly) ldsf &= ~4;
    const OctSet S = make_octset(rows, cols, dld, n_octaves, nOctaveLayers, ldsf);
    SumTex t = {sum, sld, rows, cols};
    NmsArgs B;
    memset(&B, 0, sizeof(B));
    B.det = det; B.trace = trace; B.dld = dld; B.rows = rows; B.cols = cols; B.nlayers = nOctaveLayers; B.thr = thr;
    B.mask.s = mask_sum; B.mask.sld = sld; B.mask.rows = rows; B.mask.cols = cols;
    B.bits = bits; B.sbits = sbits; B.rowcnt = rowcnt; B.segcnt = segcnt;
    Fuse0Args F;
    memset(&F, 0, sizeof(F));
    if (S.fuse0) {
        // octave 0's flag words, sign words and segment counts are accumulated by atomics: zero them (2 x 2 MB + 0.1 MB at 4K)
        const size_t words = (size_t)nOctaveLayers * rows * S.chunks[0];
        MI_HIP_TRY(hipMemsetAsync(bits, 0, sizeof(unsigned long long) * words, s));
        MI_HIP_TRY(hipMemsetAsync(sbits, 0, sizeof(unsigned long long) * words, s));
        MI_HIP_TRY(hipMemsetAsync(segcnt, 0, sizeof(unsigned) * (size_t)nOctaveLayers * rows * S.nseg[0], s));
        F.thr = thr; F.mask = B.mask; F.bits = bits; F.sbits = sbits; F.segcnt = segcnt; F.chunks = S.chunks[0]; F.nseg = S.nseg[0];
    }
    if (S.poly) {
        PolyArgs PA;
        memset(&PA, 0, sizeof(PA));
        PA.n = n_octaves;
        for (int o = 1; o < n_octaves; ++o) {
            const PolyGeo pg = poly_geo(rows, cols, o, S.pbase[o]);
            PA.prows[o] = pg.prows; PA.pld[o] = pg.pld; PA.plane_words[o] = pg.plane_words; PA.base[o] = pg.base;
        }
        hipLaunchKernelGGL(k_poly_build, dim3(div_up(cols + 1, 256), rows + 1), dim3(256), 0, s, t, poly, PA);
    }
    const int blk0 = S.fuse0 ? S.blk_dt[1] : 0;
    if (S.fuse0) hipLaunchKernelGGL(k_det_nms0, dim3(blk0), dim3(256), 0, s, t, S, F);
    if (S.blk_dt[n_octaves] > blk0)
        hipLaunchKernelGGL(k_det_trace_all, dim3(S.blk_dt[n_octaves] - blk0), dim3(256), 0, s, t, det, trace, S, (const HaarGeo *)geo_dev, blk0, (const unsigned *)poly);
    hipLaunchKernelGGL(k_nms_flag_all, dim3(S.blk_nms[n_octaves]), dim3(256), 0, s, B, S);
    hipLaunchKernelGGL(k_scan_counts_all, dim3(n_octaves), dim3(1024), 0, s, B, S);
    hipLaunchKernelGGL(k_nms_write_all, dim3(S.blk_wr[n_octaves]), dim3(256), 0, s, B, S, cand, max_candidates, ncand);
    hipLaunchKernelGGL(k_interp_eval_all, dim3(div_up(max_candidates, 256), n_octaves), dim3(256), 0, s, (const float *)det, S, (const int4 *)cand,
                       (const unsigned *)ncand, (InterpOut *)tmp, max_candidates, nfeat + 32, t);   // counters[32 + octave]: zeroed with the others
    hipLaunchKernelGGL(k_interp_compact_all, dim3(n_octaves), dim3(1024), 0, s, (const InterpOut *)tmp, (const unsigned *)ncand, n_octaves, max_candidates, kp,
                       kld, max_features, nfeat, (const unsigned *)(nfeat + 32));
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int orientation(const unsigned *sum, int sld, int rows, int cols, float *kp, int kld, const unsigned *nfeat_dev, int n_or_max,
                bool upright, const float *apt, hipStream_t s)
{
    if (n_or_max <= 0) return MI_OK;
    if (upright) {
        hipLaunchKernelGGL(k_fill_angle, dim3(div_up(n_or_max, 256)), dim3(256), 0, s, kp, kld, nfeat_dev, n_or_max, 360.0f - 90.0f);
    } else {
        SumTex t = {sum, sld, rows, cols};
        const int grid = nfeat_dev ? std::min(div_up(n_or_max, 4), MI_SURF_ORI_WGS_PER_CU * (device_simds() / 4)) : div_up(n_or_max, 4);
        hipLaunchKernelGGL(k_orientation, dim3(grid), dim3(256), 0, s, t, kp, kld, nfeat_dev, n_or_max, apt);
    }
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

// Cell side (pixels) from which a feature's patch is built by the staged kernel; MIFLOW_SURF_STAGE_S (0 = never).  The tile must hold
// one patch row of the largest feature the detector can produce: size <= 27 << 5 would be octave 5; s = size * 1.2 / 9 -- 64 KB
// holds (s + 3) x (21 s + 2) bytes up to s = 53 (size 400: octave 3's largest is 216).
static float surf_stage_s()
{
    static const float v = [] { const char *e = getenv("MIFLOW_SURF_STAGE_S"); const float x = e ? (float)atof(e) : 5.0f; return x > 0.f ? x : 1e30f; }();
    return v;
}
int descriptors(const unsigned char *img, long long istep, int rows, int cols, const float *kp, int kld, int nfeat, bool extended,
                float *desc, long long dstep_floats, const float *dw, hipStream_t s, const unsigned *nfeat_dev)
{
    if (nfeat <= 0) return MI_OK;
    const float ss = surf_stage_s();
    // dynamic LDS of the staged kernel: the lattice tile of a strip.  (Experiments build: behind it the image tile of the row-coalesced
    // staging, MIFLOW_SURF_TILE_KB / _IMG_KB / _COAL_S; the release library runs the per-texel gather into one 48 KB tile.)
#ifdef MIFLOW_EXPERIMENTS
    static const int kTileBytes = [] { const char *e = getenv("MIFLOW_SURF_TILE_KB"); const int kb = e ? atoi(e) : 48; return (kb >= 8 && kb <= 100 ? kb : 48) * 1024; }();
    static const int kImgBytes = [] { const char *e = getenv("MIFLOW_SURF_IMG_KB"); const int kb = e ? atoi(e) : 0; return (kb >= 0 && kb <= 100 ? kb : 0) * 1024; }();
    static const float kCoalS = [] { const char *e = getenv("MIFLOW_SURF_COAL_S"); return e ? (float)atof(e) : 16.0f; }();
#else
    constexpr int kTileBytes = 48 * 1024, kImgBytes = 0;
    constexpr float kCoalS = 1e30f;
#endif
    // host-known count: one workgroup per feature.  Count on the device (nfeat = its upper bound, maxFeatures): a fixed grid of a few
    // workgroups per CU walks the features block-cyclically, most expensive first -- no launch of tens of thousands of empty workgroups
    const int grid = nfeat_dev ? std::min(nfeat, MI_SURF_DESC_WGS_PER_CU * (device_simds() / 4)) : nfeat;
    if (!(ss < 1e29f)) {   // staging switched off (MIFLOW_SURF_STAGE_S=0): every patch straight from global memory
        if (extended) hipLaunchKernelGGL(k_descriptors<true>, dim3(grid), dim3(512), 0, s, img, istep, rows, cols, kp, kld, nfeat, nfeat_dev, desc, dstep_floats, dw, ss, kTileBytes);
        else hipLaunchKernelGGL(k_descriptors<false>, dim3(grid), dim3(512), 0, s, img, istep, rows, cols, kp, kld, nfeat, nfeat_dev, desc, dstep_floats, dw, ss, kTileBytes);
    } else {
        static const hipError_t attr_rc = [] {
            hipError_t e = hipFuncSetAttribute((const void *)k_descriptors_staged<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kTileBytes + kImgBytes);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_descriptors_staged<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kTileBytes + kImgBytes);
            return e;
        }();
        MI_HIP_TRY(attr_rc);
        if (extended) hipLaunchKernelGGL(k_descriptors_staged<true>, dim3(grid), dim3(512), kTileBytes + kImgBytes, s, img, istep, rows, cols, kp, kld, nfeat, nfeat_dev, desc, dstep_floats, dw, ss, kTileBytes, kImgBytes, kCoalS);
        else hipLaunchKernelGGL(k_descriptors_staged<false>, dim3(grid), dim3(512), kTileBytes + kImgBytes, s, img, istep, rows, cols, kp, kld, nfeat, nfeat_dev, desc, dstep_floats, dw, ss, kTileBytes, kImgBytes, kCoalS);
    }
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int dbg_scan(const unsigned *in_dev, unsigned *out_dev, hipStream_t s)
{
    hipLaunchKernelGGL(k_dbg_scan, dim3(1), dim3(64), 0, s, in_dev, out_dev);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace surf
}  // namespace mi
