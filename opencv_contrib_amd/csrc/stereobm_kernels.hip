// Block-matching stereo (cv::cuda::StereoBM) -- HIP kernels for gfx950 (MI355X, CDNA4), wave64.
//
// Reference: modules/cudastereo/src/cuda/stereobm.cu:66-711 (SSD block matching in batches of 8
// disparities with column sums in shared memory, prefilters, textureness post-filter) driven by
// modules/cudastereo/src/stereobm.cpp:139-191.  Results are bit-identical to that code (integer
// arithmetic; tie-breaking of stereobm.cu:118-127,349; the truncated right half-window of the 128-wide
// block mapping, SURVEY Appendix B Q1, is reproduced when emulate_edge is set).
//
// MI355X formulation (not the reference's thread-per-column / smem-column-sum layout):
//   * LANE = DISPARITY.  A wave holds 64 consecutive disparities of one image tile; a workgroup is
//     ndisp/64 waves working on the same tile.  The left pixel is wave-uniform (LDS broadcast read), the
//     right pixels of the 64 lanes are 64 consecutive bytes read backwards (x - d).
//   * each lane keeps the vertical column sums of its disparity for all TW + 2R columns of the tile in
//     VGPRs and slides them down the rows (add the entering row, subtract the leaving one); the
//     horizontal window is a register sliding sum -- no shared-memory column sums, no per-batch barriers.
//   * winner-take-all = wave-wide min over lanes (DPP row_shr / row_bcast) + a ballot that reproduces
//     the reference's "last index inside a batch of 8, first batch across batches" rule with scalar bit
//     operations.
//   * uniqueness check (non-default) = closed form of the reference's sequential logic, evaluated by a
//     second pass of the same kernel (MODE 1) against the winner of pass 0.
// Rows of both images for the tile are staged in LDS once (byte copies: no alignment assumption on the
// caller's GpuMat), then read as aligned dwords + v_alignbyte.
#include "stereobm_dev.h"
#include <climits>
#include <cstdlib>
#include <type_traits>

namespace mi {
namespace sbm {

// ------------------------------------------------------------------ wave helpers
// DPP source fetch with "0 for lanes without a source" (bound_ctrl:1 / old = 0): 0 is the identity of an
// unsigned MAX, so the winner-take-all below runs as a max-reduction over complemented SSDs and every
// step is a single v_max_u32_dpp with no preparatory v_mov.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
// max over the 64 lanes, returned wave-uniform
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = max(v, dpp_u32<0x111, 0xf>(v));  // row_shr:1
    v = max(v, dpp_u32<0x112, 0xf>(v));  // row_shr:2
    v = max(v, dpp_u32<0x114, 0xf>(v));  // row_shr:4
    v = max(v, dpp_u32<0x118, 0xf>(v));  // row_shr:8   -> lane 15 of each row = row max
    v = max(v, dpp_u32<0x142, 0xa>(v));  // row_bcast:15 into rows 1,3
    v = max(v, dpp_u32<0x143, 0xc>(v));  // row_bcast:31 into rows 2,3 -> lane 63 = wave max
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) { return ~wave_max_u32(~v); }
// four independent reductions, step-interleaved so the DPP read-after-write wait states of one chain are
// filled by the other three
__device__ __forceinline__ void wave_max_u32x4(const unsigned in[4], unsigned out[4])
{
    unsigned v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = in[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = max(v[k], dpp_u32<0x111, 0xf>(v[k]));
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = max(v[k], dpp_u32<0x112, 0xf>(v[k]));
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = max(v[k], dpp_u32<0x114, 0xf>(v[k]));
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = max(v[k], dpp_u32<0x118, 0xf>(v[k]));
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = max(v[k], dpp_u32<0x142, 0xa>(v[k]));
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = max(v[k], dpp_u32<0x143, 0xc>(v[k]));
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = (unsigned)__builtin_amdgcn_readlane((int)v[k], 63);
}
// Transposed max-reduction: 16 per-lane values (one per tile column) -> lane j (of every 16-lane row) holds the
// maximum over all 64 lanes of value j.  Level L pairs lanes that differ in bit L: each lane keeps the half of its
// values selected by its own bit and receives the partner's copy of that half, so the value count halves while
// the lanes covered double: 16+8+4+2 exchanges instead of 16 x 6 for one-at-a-time reductions.
template <int CTRL, int BANK>
__device__ __forceinline__ unsigned dppb(unsigned old, unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, BANK, true);
}
__device__ __forceinline__ unsigned tmax16(const unsigned *v, int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    unsigned w[8], x[4], y[2];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const unsigned keep = b0 ? v[2 * m + 1] : v[2 * m], give = b0 ? v[2 * m] : v[2 * m + 1];
        w[m] = max(keep, dppb<0xB1, 0xf>(0u, give));          // quad_perm:[1,0,3,2]  (lane ^ 1)
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const unsigned keep = b1 ? w[2 * m + 1] : w[2 * m], give = b1 ? w[2 * m] : w[2 * m + 1];
        x[m] = max(keep, dppb<0x4E, 0xf>(0u, give));          // quad_perm:[2,3,0,1]  (lane ^ 2)
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const unsigned keep = b2 ? x[2 * m + 1] : x[2 * m], give = b2 ? x[2 * m] : x[2 * m + 1];
        const unsigned lo = dppb<0x104, 0x5>(0u, give);       // row_shl:4 into banks 0,2 (lanes 0-3, 8-11 <- lane + 4)
        y[m] = max(keep, dppb<0x114, 0xa>(lo, give));         // row_shr:4 into banks 1,3 (lanes 4-7, 12-15 <- lane - 4)
    }
    const unsigned keep = b3 ? y[1] : y[0], give = b3 ? y[0] : y[1];
    const unsigned lo = dppb<0x108, 0x3>(0u, give);           // row_shl:8 into banks 0,1
    unsigned z = max(keep, dppb<0x118, 0xc>(lo, give));       // row_shr:8 into banks 2,3      (lane ^ 8)
    z = max(z, (unsigned)__shfl_xor((int)z, 16));             // the four rows of 16 lanes
    z = max(z, (unsigned)__shfl_xor((int)z, 32));
    return z;
}

__global__ void k_dbg_tmax16(const unsigned *in, unsigned *out)
{
    unsigned v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = in[k * 64 + threadIdx.x];
    out[threadIdx.x] = tmax16(v, threadIdx.x);
}

// Among the lanes flagged in `mask` (bit = lane = disparity - 64*set): the reference keeps the LAST
// index inside a batch of 8 (stereobm.cu:120-125) and the FIRST batch (strict <, :349,426).
__device__ __forceinline__ int pick_lane(unsigned long long mask)
{
    const int first = __builtin_ctzll(mask);
    const int batch = first >> 3;
    const unsigned byte = (unsigned)(mask >> (batch * 8)) & 0xffu;
    return batch * 8 + (31 - __builtin_clz(byte));
}

// lane `i` of `acc` := wave-uniform value v (v_writelane_b32 semantics)
__device__ __forceinline__ unsigned put_lane(unsigned acc, unsigned v, int i, int lane)
{
    return lane == i ? v : acc;
}

__global__ void k_dbg_wave_min(const unsigned *in, unsigned *out)
{
    const unsigned v = in[threadIdx.x];
    const unsigned m = wave_min_u32(v);   // = ~max(~v)
    const unsigned long long mk = __ballot(v == m);
    out[threadIdx.x] = m;
    if (threadIdx.x == 0) out[64] = (unsigned)pick_lane(mk);
}

// LDS pointers keep their address space (a generic pointer would turn the accesses into flat_load / flat_store)
typedef __attribute__((address_space(3))) unsigned MI_LDSU;
typedef unsigned mi_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) mi_u4 MI_LDSU4;
// ------------------------------------------------------------------ block matching
struct BmArgs {
    const unsigned char *left, *right;
    long long lstep, rstep;
    unsigned char *disp;
    long long dstep;
    unsigned *minssd;      // [rows][mstep] winner SSD (pass 0 writes, pass 1 reads); may be null in pass 0
    long long mstep;       // elements
    int rows, cols, ndisp, nsets, rb;
    int emulate_edge;
    float thresh_scale;    // (float)(1.0 + uniquenessRatio / 100.0f)  stereobm.cu:273
    int batch;
    int swz;               // XCD-contiguous tile order (MIFLOW_SBM_SWZ, default 1)
    const BmPair *tab;     // batch: blockIdx.z = pair, image pointers from this device table, minssd advances by mpair elements
    long long mpair;
};

template <int R>
struct Cfg {
    static constexpr int TWr = (64 - 2 * R) & ~3;
    // R <= 12: max SSD (2R+1)^2 * 255^2 < 2^26, so (2^26-1 - SSD) << 6 | tie-break key fits one u32 and the winner is a
    // single max-reduction, done 16 columns at a time (tile width 48 = 3 groups); larger windows: generic path
    static constexpr bool PACKED = R <= 12;
    static constexpr int TW = PACKED ? 48 : (TWr < 16 ? 16 : TWr);   // output columns per tile
    static constexpr int NC = TW + 2 * R;            // column sums per lane
    static constexpr int LS = (NC + 15) / 16 * 16;   // LDS bytes per staged left row
    static constexpr int NLW = LS / 4;               // dwords per left row
    static constexpr int NRW = (NC + 3) / 4 + 1;     // aligned dwords a lane reads per right row
};

// MODE 0: winner-take-all.  MODE 1: uniqueness verification of the pass-0 winner.
// WT (MODE 0, packed windows R <= 12; round 4): winner-take-all through an LDS TRANSPOSITION instead of the transposed DPP reduction.
// A lane (= disparity) writes the complemented window SSDs of the tile's columns to T[column][lane]; lane i then reads column i's 64
// values and scans them with one v_lshl_or + v_max per disparity (the tie-break key is a compile-time constant of the scan position).
// In groups of 16 columns (4.3 KB of LDS per wave; the whole tile at once cost the kernel its occupancy: 4 226 against 6 089 pairs/s,
// r08j): lane (c, q) scans the 16 disparities of quarter q of column c -- 32 operations per group, 2 cross-row exchanges -- instead of
// 49 shifts + 49 bit-ops to pack and 141 selects + 69 DPP / max operations to reduce (profiles/static_mix_sbm.json); same winners bit
// for bit (same score, same key).  Row pitch of T = 68 dwords: the write of a column is 64 consecutive banks, a b128 read of 16
// lanes covers the 64 banks once.
template <int R, int MODE, bool WT = false>
__global__ __launch_bounds__(256) void k_block_match(BmArgs A)
{
    // XCD-contiguous tile order (A.swz): workgroups are dealt round-robin over the 8 XCDs in launch order, so neighbouring tiles of
    // a row band -- which read the same rows of both images -- used to sit behind eight different L2s and every L2 fetched nearly the
    // whole pair (r10p: 25 MB per pair fetched for 4.1 MB of images).  XCD k now takes the k-th contiguous eighth of the
    // (pair, y, x) tiles.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (A.swz) {
        const unsigned nxy = gridDim.x * gridDim.y, nwg = nxy * gridDim.z, orig = (bz * gridDim.y + by) * gridDim.x + bx;
        const unsigned xcd = orig & 7u, qq = nwg >> 3, rr = nwg & 7u;
        const unsigned lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
        bz = lid / nxy;
        const unsigned rem = lid - bz * nxy;
        by = rem / gridDim.x; bx = rem - by * gridDim.x;
    }
    if (A.tab) {
        const BmPair p = A.tab[bz];
        A.left = p.left; A.right = p.right; A.disp = p.disp; A.lstep = p.lstep; A.rstep = p.rstep; A.dstep = p.dstep;
        if (A.minssd) A.minssd += (long long)bz * A.mpair;
    }
    using C = Cfg<R>;
    constexpr int TW = C::TW, NC = C::NC, LS = C::LS, NLW = C::NLW, NRW = C::NRW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wset = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nsets = A.nsets;
    const int ndp = nsets * 64;                        // padded disparity range
    const int d = wset * 64 + lane;
    const bool active = d < A.ndisp;
    const int X0 = A.ndisp + R + (int)bx * TW;         // first output column of the tile
    const int Y0 = R + (int)by * A.rb;                 // first output row
    const int nrows = min(A.rb, A.rows - R - Y0);
    const int ncols = min(TW, A.cols - R - X0);        // valid output columns (X < cols - R)
    if (nrows <= 0 || ncols <= 0) return;
    const int srows = nrows + 2 * R;                   // staged rows: Y0-R .. Y0+nrows+R-1
    const int RS = (NC + ndp - 1 + 3) / 4 * 4 + 4;     // LDS bytes per staged right row
    unsigned char *Ls = smem;
    unsigned char *Rs = smem + (size_t)(A.rb + 2 * R) * LS;
    unsigned *comb = reinterpret_cast<unsigned *>(Rs + (size_t)(A.rb + 2 * R) * RS);  // [2][nsets][2][64]
    constexpr int TP = 68;                                                               // dwords per row of the transposition buffer
    // WT: [16][TP] per wave (one group of 16 columns at a time), behind comb at the next 16-byte boundary (b128 reads)
    unsigned *Tb = reinterpret_cast<unsigned *>(smem + (((size_t)(A.rb + 2 * R) * (LS + RS) + (size_t)2 * nsets * 128 * sizeof(unsigned) + 15) / 16 * 16)) +
                   (WT ? (size_t)wset * 16 * TP : 0);

    // ---- stage the tile rows (byte copies; columns outside the image read as 0 and are never used
    //      by an active lane / valid column)
    {
        const int xl0 = X0 - R;                 // left columns xl0 .. xl0+NC-1
        const int xr0 = X0 - R - (ndp - 1);     // right columns xr0 .. xl0+NC-1
        const int nthreads = blockDim.x;
        const int lw = LS, rw = RS;
        for (int i = threadIdx.x; i < srows * lw; i += nthreads) {
            const int r = i / lw, c = i - r * lw;
            const int x = xl0 + c, y = Y0 - R + r;
            Ls[i] = (x < A.cols) ? A.left[(long long)y * A.lstep + x] : (unsigned char)0;
        }
        for (int i = threadIdx.x; i < srows * rw; i += nthreads) {
            const int r = i / rw, c = i - r * rw;
            const int x = xr0 + c, y = Y0 - R + r;
            Rs[i] = (x >= 0 && x < A.cols) ? A.right[(long long)y * A.rstep + x] : (unsigned char)0;
        }
    }
    __syncthreads();

    // lane's byte offset inside a staged right row for column c: c + (ndp-1-d)
    const int off = ndp - 1 - d;
    const int q0 = off >> 2, sh = off & 3;
    const unsigned *Lw = reinterpret_cast<const unsigned *>(Ls);
    const unsigned *Rw = reinterpret_cast<const unsigned *>(Rs) + q0;
    const int RSW = RS / 4;

    unsigned cs[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cs[c] = 0;

    // inactive lanes (d >= ndisp) carry a bias above any real SSD (max 51*51*255^2 < 2^28) so they never win
    const unsigned bias = active ? 0u : 0x40000000u;
    // wave-uniform facts about the tile
#ifdef MI_STATIC_MIX_INTERIOR_TILES   // tools/static_mix.py sbm: count the row loop of a tile away from the right image edge
    const bool edge_tile = false;
#else
    const bool edge_tile = A.emulate_edge && (X0 + ncols + R > A.cols - R);   // some X + R >= cols - R
#endif

    unsigned vd = 0, vo = 0;   // MODE 1: per-lane (lane = tile column) winner disparity / SSD of the row

    const int nsteps = nrows + 2 * R;
    for (int s = 0; s < nsteps; ++s) {
        // ---- vertical: add staged row s, subtract staged row s-(2R+1)
        {
            unsigned rw[NRW];
            const unsigned *rp = Rw + s * RSW;
#pragma unroll
            for (int i = 0; i < NRW; ++i) rw[i] = rp[i];
            const unsigned *lp = Lw + s * NLW;
#pragma unroll
            for (int i = 0; i < NRW - 1; ++i) {
                const unsigned rr = __builtin_amdgcn_alignbyte(rw[i + 1], rw[i], sh);
                const unsigned ll = lp[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = i * 4 + k;
                    if (c < NC) {
                        const int e = (int)((ll >> (8 * k)) & 0xff) - (int)((rr >> (8 * k)) & 0xff);
                        cs[c] += (unsigned)(e * e);
                    }
                }
            }
        }
        if (s >= 2 * R + 1) {
            const int so = s - (2 * R + 1);
            unsigned rw[NRW];
            const unsigned *rp = Rw + so * RSW;
#pragma unroll
            for (int i = 0; i < NRW; ++i) rw[i] = rp[i];
            const unsigned *lp = Lw + so * NLW;
#pragma unroll
            for (int i = 0; i < NRW - 1; ++i) {
                const unsigned rr = __builtin_amdgcn_alignbyte(rw[i + 1], rw[i], sh);
                const unsigned ll = lp[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = i * 4 + k;
                    if (c < NC) {
                        const int e = (int)((ll >> (8 * k)) & 0xff) - (int)((rr >> (8 * k)) & 0xff);
                        cs[c] -= (unsigned)(e * e);
                    }
                }
            }
        }
        if (s < 2 * R) continue;
        const int y = Y0 + s - 2 * R;     // output row

        if (MODE == 1) {
            // winner of pass 0 for this row: lane i <- tile column i
            const bool ok = lane < ncols;
            vd = ok ? A.disp[(long long)y * A.dstep + X0 + lane] : 0u;
            vo = ok ? A.minssd[(long long)y * A.mstep + X0 + lane] : 0u;
        }

        // ---- horizontal window + winner per output column (all TW columns: results of columns past
        //      ncols are discarded at the store, so the TW reductions form one branch-free block the
        //      scheduler can interleave)
        unsigned resm = UINT_MAX, resd = 0;   // lane i collects the result of tile column i
        auto row_stage = [&](auto edge_tag) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            if (MODE == 0 && C::PACKED && WT) {
                // ---- winner-take-all through LDS (see the kernel's header), 16 columns at a time: lane d writes the window values
                //      of the group's columns to T[column][d]; lane (c = lane % 16, q = lane / 16) scans disparities 16 q .. 16 q + 15 of
                //      column c with the same score as the DPP path, (2^26-1 - SSD) << 6 | (d ^ 0x38); the four quarters of a column are
                //      combined across the rows of 16 lanes.  An inactive lane's value is written as 0, which no active lane's (> 0) ties.
                // per-lane constant mask (all ones for an active lane): ONE full-rate v_and per column -- a select on the wave-uniform
                // "every lane is active" cost a half-rate v_cndmask per column on top of it
                unsigned amask = active ? 0xffffffffu : 0u;
                asm volatile("" : "+v"(amask));
                unsigned nw = 0x3ffffffu, nh = 0x3ffffffu;
#pragma unroll
                for (int c = 0; c < 2 * R; ++c) nw -= cs[c];
                if (EDGE) {
#pragma unroll
                    for (int c = 0; c < R; ++c) nh -= cs[c];
                }
                MI_LDSU *tw = (MI_LDSU *)(Tb + lane);
                const int sc = lane & 15, sq = lane >> 4;
                const MI_LDSU4 *tr = (const MI_LDSU4 *)(Tb + sc * TP + sq * 16);
                const unsigned kq = (unsigned)(sq * 16);
                unsigned zr[TW / 16];
                // window values of one group of 16 columns (registers); the sliding sums continue from group to group
                auto group_vals = [&](int g, unsigned (&v)[16]) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int i = g * 16 + k;
                        nw -= cs[i + 2 * R];
                        unsigned val = nw;
                        if (EDGE) {
                            nh -= cs[i + R];
                            const int X = X0 + i;
                            const int t = (X - A.ndisp - R) & 127;
                            val = (t < 128 - R && X + R >= A.cols - R) ? nh : nw;   // stereobm.cu:77-89
                            nh += cs[i];
                        }
                        nw += cs[i];
                        v[k] = val & amask;
                    }
                };
                // Order of a row: write group g, issue its reads, compute group g + 1's values WHILE the reads are in flight, scan g.
                // (LDS executes a wave's operations in order: the reads of group g see its writes, and the writes of group g + 1 -- same
                // buffer -- come after them.)
                unsigned cur[16], nxt[16];
                group_vals(0, cur);
#pragma unroll
                for (int g = 0; g < TW / 16; ++g) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) tw[k * TP] = cur[k];
                    mi_u4 rd[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) rd[q] = tr[q];
                    if (g + 1 < TW / 16) group_vals(g + 1, nxt);
                    unsigned z = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        z = max(z, (rd[q].x << 6) | ((kq + 4 * q + 0) ^ 0x38u));
                        z = max(z, (rd[q].y << 6) | ((kq + 4 * q + 1) ^ 0x38u));
                        z = max(z, (rd[q].z << 6) | ((kq + 4 * q + 2) ^ 0x38u));
                        z = max(z, (rd[q].w << 6) | ((kq + 4 * q + 3) ^ 0x38u));
                    }
                    z = max(z, (unsigned)__shfl_xor((int)z, 16));   // the four quarters of the column
                    z = max(z, (unsigned)__shfl_xor((int)z, 32));
                    zr[g] = z;
#pragma unroll
                    for (int k = 0; k < 16; ++k) cur[k] = nxt[k];
                }
                // lane i <- column i of the tile: row r of 16 lanes takes group r
                unsigned z = zr[0];
#pragma unroll
                for (int g = 1; g < TW / 16; ++g) z = (lane >> 4) == g ? zr[g] : z;
                resm = 0x3ffffffu - (z >> 6);
                resd = (unsigned)(wset * 64) + ((z & 63u) ^ 0x38u);
                return;
            }
            if (MODE == 0 && C::PACKED) {
                // ---- packed winner-take-all: score = (2^26-1 - SSD) << 6 | key, key = lane ^ 0x38 (earlier batch of 8
                //      wins, then the LAST index inside the batch: stereobm.cu:120-125,349); inactive lanes score 0
                const unsigned key = (unsigned)(lane ^ 0x38), amask = active ? 0xffffffffu : 0u;
                unsigned nw = 0x3ffffffu, nh = 0x3ffffffu;
#pragma unroll
                for (int c = 0; c < 2 * R; ++c) nw -= cs[c];
                if (EDGE) {
#pragma unroll
                    for (int c = 0; c < R; ++c) nh -= cs[c];
                }
                unsigned zr[TW / 16];
#pragma unroll
                for (int g = 0; g < TW / 16; ++g) {
                    unsigned pk[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int i = g * 16 + k;
                        nw -= cs[i + 2 * R];
                        unsigned val = nw;
                        if (EDGE) {
                            nh -= cs[i + R];
                            const int X = X0 + i;
                            const int t = (X - A.ndisp - R) & 127;
                            val = (t < 128 - R && X + R >= A.cols - R) ? nh : nw;   // stereobm.cu:77-89 (see below)
                            nh += cs[i];
                        }
                        nw += cs[i];
                        pk[k] = ((val << 6) | key) & amask;
                    }
                    zr[g] = tmax16(pk, lane);
                }
                // lane i <- column i of the tile: row r of 16 lanes takes group r
                unsigned z = zr[0];
#pragma unroll
                for (int g = 1; g < TW / 16; ++g) z = (lane >> 4) == g ? zr[g] : z;
                resm = 0x3ffffffu - (z >> 6);
                resd = (unsigned)(wset * 64) + ((z & 63u) ^ 0x38u);
                return;
            }
            // nwin = ~(window SSD) = UINT_MAX - SSD, slid in complemented form (same op count)
            unsigned nwin = ~bias;
#pragma unroll
            for (int c = 0; c < 2 * R; ++c) nwin -= cs[c];
            unsigned nhalf = ~bias;   // ~(columns i .. i+R: left half + centre), only for the edge emulation
            if (EDGE) {
#pragma unroll
                for (int c = 0; c < R; ++c) nhalf -= cs[c];
            }
#pragma unroll
            for (int i0 = 0; i0 < TW; i0 += 4) {
                unsigned nsd[4];   // complemented SSD of the 4 columns
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k;
                    nwin -= cs[i + 2 * R];
                    nsd[k] = nwin;
                    if (EDGE) {
                        nhalf -= cs[i + R];
                        // stereobm.cu:77-89: threads t < 128-R take the right half-window from thread t+R, whose
                        // own test X+R < cols-R fails near the right image edge -> the right half is dropped
                        const int X = X0 + i;
                        const int t = (X - A.ndisp - R) & 127;
                        nsd[k] = (t < 128 - R && X + R >= A.cols - R) ? nhalf : nwin;
                        nhalf += cs[i];
                    }
                    nwin += cs[i];
                }
                if (MODE == 0) {
                    unsigned m[4];
                    wave_max_u32x4(nsd, m);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned long long mk = __ballot(nsd[k] == m[k]);
                        const int dl = wset * 64 + pick_lane(mk);
                        resm = put_lane(resm, ~m[k], i0 + k, lane);
                        resd = put_lane(resd, (unsigned)dl, i0 + k, lane);
                    }
                } else {
                    // closed form of the sequential uniqueness logic (stereobm.cu:311-346,390-422): with
                    // b* = batch of the final winner, the winner is rejected iff
                    //   (a) the best candidate of the batches before b* (same tie rules) lies outside
                    //       dtest+-1 and its SSD <= thresh, or
                    //   (b) some disparity d >= 8b*-2 outside dtest+-1 has SSD <= thresh,
                    // thresh = thresh_scale * (float)opt.
                    unsigned pv[4], pm[4];
                    bool before[4];
                    unsigned long long ex[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int dtest = __builtin_amdgcn_readlane((int)vd, i0 + k);
                        const unsigned opt = (unsigned)__builtin_amdgcn_readlane((int)vo, i0 + k);
                        const float thresh = A.thresh_scale * (float)opt;
                        const int b8 = dtest & ~7;
                        const bool outside = (d < dtest - 1) || (d > dtest + 1);
                        const bool le = (float)(~nsd[k]) <= thresh;
                        ex[k] = __ballot(active && d >= b8 - 2 && outside && le);
                        before[k] = active && d < b8;
                        pv[k] = before[k] ? nsd[k] : 0u;
                    }
                    wave_max_u32x4(pv, pm);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned long long pk = __ballot(before[k] && pv[k] == pm[k]);
                        const int pd = pk ? wset * 64 + pick_lane(pk ? pk : 1ull) : -1;
                        // resm: previous-best SSD of this set (UINT_MAX: none); resd: its disparity | exists flag << 16
                        resm = put_lane(resm, ~pm[k], i0 + k, lane);
                        resd = put_lane(resd, (unsigned)((pd & 0xffff) | (ex[k] ? 0x10000 : 0)), i0 + k, lane);
                    }
                }
            }
        };
        if (edge_tile) row_stage(std::true_type{});
        else row_stage(std::false_type{});

        // ---- combine the disparity sets of the workgroup (earlier set wins ties) and store
        unsigned *cb = comb + (size_t)((s & 1) * nsets) * 128;
        if (nsets > 1) {
            cb[wset * 128 + lane] = resm;
            cb[wset * 128 + 64 + lane] = resd;
            __syncthreads();
        }
        if (wset == 0 && lane < ncols) {
            unsigned bm = resm, bd = resd;
            if (MODE == 0) {
                for (int w = 1; w < nsets; ++w) {
                    const unsigned m = cb[w * 128 + lane];
                    if (m < bm) { bm = m; bd = cb[w * 128 + 64 + lane]; }
                }
                A.disp[(long long)y * A.dstep + X0 + lane] = (unsigned char)bd;
                if (A.minssd) A.minssd[(long long)y * A.mstep + X0 + lane] = bm;
            } else {
                bool ex = (bd >> 16) & 1;
                int pd = (int)(short)(bd & 0xffff);
                for (int w = 1; w < nsets; ++w) {
                    const unsigned m = cb[w * 128 + lane];
                    const unsigned dd = cb[w * 128 + 64 + lane];
                    ex = ex || ((dd >> 16) & 1);
                    if (m < bm) { bm = m; pd = (int)(short)(dd & 0xffff); }
                }
                const int dtest = (int)vd;
                const float thresh = A.thresh_scale * (float)vo;
                bool reject = ex;
                if (pd >= 0 && (pd < dtest - 1 || pd > dtest + 1) && ((float)bm <= thresh)) reject = true;
                if (reject) A.disp[(long long)y * A.dstep + X0 + lane] = 0;
            }
        }
    }
}

// ------------------------------------------------------------------ prefilters
// stereobm.cu:522-536
__global__ __launch_bounds__(256) void k_prefilter_xsobel(const unsigned char *src, long long sstep, unsigned char *dst,
                                                          long long dstep, int rows, int cols, int cap)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const unsigned char *r0 = src + (long long)max(0, y - 1) * sstep;
    const unsigned char *r1 = src + (long long)y * sstep;
    const unsigned char *r2 = src + (long long)min(y + 1, rows - 1) * sstep;
    const int xl = max(0, x - 1), xr = min(x + 1, cols - 1);
    int conv = r0[xl] * (-1) + r0[xr] * (1) + r1[xl] * (-2) + r1[xr] * (2) + r2[xl] * (-1) + r2[xr] * (1);
    conv = min(min(max(-cap, conv), cap) + cap, 255);
    dst[(long long)y * dstep + x] = (unsigned char)(conv & 0xFF);
}

// stereobm.cu:557-581 (x+1 used twice in the 5-point term, :568-570).  One wave per 64 columns walks
// down a band of rows keeping the winsize x winsize box sum as column sums in LDS-free registers:
// column sums are recomputed per row from a sliding vertical sum held by the lane.
__global__ __launch_bounds__(256) void k_prefilter_norm(const unsigned char *src, long long sstep, unsigned char *dst,
                                                        long long dstep, int rows, int cols, int cap, int scale_g,
                                                        int scale_s, int winsize)
{
    __shared__ int colsum[4][64 + 64];   // per wave: 64 own columns + up to 32 halo columns each side
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int W2 = winsize / 2;          // host guarantees W2 <= 32
    const int x = blockIdx.x * 64 + lane;
    const int yb = (blockIdx.y * 4 + wv) * 32;
    if (yb >= rows) return;
    const int ye = min(yb + 32, rows);
    // lane owns column sums of x (at index 32 + lane) and of one halo column
    const int xh = lane < 32 ? blockIdx.x * 64 - 32 + lane : blockIdx.x * 64 + 64 + (lane - 32);
    const int xc = min(max(x, 0), cols - 1), xhc = min(max(xh, 0), cols - 1);
    int s0 = 0, s1 = 0;
    for (int i = -W2; i <= W2; ++i) {
        const long long r = (long long)min(max(yb + i, 0), rows - 1) * sstep;
        s0 += src[r + xc];
        s1 += src[r + xhc];
    }
    int *cs = colsum[wv];
    for (int y = yb; y < ye; ++y) {
        cs[32 + lane] = s0;
        cs[lane < 32 ? lane : 64 + lane] = s1;
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave are visible to its own reads
        __builtin_amdgcn_wave_barrier();
        if (x < cols) {
            int cov2 = 0;
            for (int j = -W2; j <= W2; ++j) cov2 += cs[32 + lane + j];
            const unsigned char *ru = src + (long long)max(y - 1, 0) * sstep;
            const unsigned char *rc = src + (long long)y * sstep;
            const unsigned char *rd = src + (long long)min(y + 1, rows - 1) * sstep;
            const int xr = min(x + 1, cols - 1);
            const int cov1 = ru[x] * 1 + rc[xr] * 1 + rc[x] * 4 + rc[xr] * 1 + rd[x] * 1;
            int res = (cov1 * scale_g - cov2 * scale_s) >> 10;
            res = min(max(res, -cap), cap) + cap;
            dst[(long long)y * dstep + x] = (unsigned char)res;
        }
        __builtin_amdgcn_wave_barrier();
        // slide the column sums one row down (clamped rows, like the reference's clamp())
        const long long ra = (long long)min(max(y + 1 + W2, 0), rows - 1) * sstep;
        const long long rs = (long long)min(max(y - W2, 0), rows - 1) * sstep;
        s0 += (int)src[ra + xc] - (int)src[rs + xc];
        s1 += (int)src[ra + xhc] - (int)src[rs + xhc];
    }
}

// ------------------------------------------------------------------ textureness post-filter
// stereobm.cu:606-711 under the exact-integer definition of oracle/stereobm_ref.c (orc_sbm_textureness):
// B = 2x2 box sum (the reference's half-pixel bilinear texture fetch x 4 x 255), S = |x-Sobel of B|,
// zero the disparity when (float)(window sum of S) * 0.25f < avgTexThreshold * winsz^2.
__device__ __forceinline__ int tex_box4(const unsigned char *img, long long step, int rows, int cols, int x, int y)
{
    const int x0 = min(max(x - 1, 0), cols - 1), x1 = min(max(x, 0), cols - 1);
    const long long r0 = (long long)min(max(y - 1, 0), rows - 1) * step, r1 = (long long)min(max(y, 0), rows - 1) * step;
    return img[r0 + x0] + img[r0 + x1] + img[r1 + x0] + img[r1 + x1];
}
__device__ __forceinline__ int tex_sobel(const unsigned char *img, long long step, int rows, int cols, int x, int y)
{
    const int c = -tex_box4(img, step, rows, cols, x - 1, y - 1) + tex_box4(img, step, rows, cols, x + 1, y - 1)
                  - 2 * tex_box4(img, step, rows, cols, x - 1, y) + 2 * tex_box4(img, step, rows, cols, x + 1, y)
                  - tex_box4(img, step, rows, cols, x - 1, y + 1) + tex_box4(img, step, rows, cols, x + 1, y + 1);
    return abs(c);
}

// pass 1: S = |x-Sobel of B| on the extended domain x in [-TEX_MX, ...), y in [-TEX_MY, rows + TEX_MY) (coordinates are
// clamped texel-wise inside tex_box4, exactly like reading the clamp-addressed texture out of range)
#define TEX_MX 32
#define TEX_MY 26
__global__ __launch_bounds__(256) void k_tex_sobel(const unsigned char *img, long long istep, int rows, int cols, int *S, int sld, int sh)
{
    const int xe = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ye = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xe >= sld || ye >= sh) return;
    S[(long long)ye * sld + xe] = tex_sobel(img, istep, rows, cols, xe - TEX_MX, ye - TEX_MY);
}

// pass 2: winsz x winsz window sums of S by sliding column sums (one wave per 64 columns x TEX_RPW rows) + threshold.
// 8 rows per wave: 32 put one wave on each SIMD at 1080p, every row a dependent LDS / global round trip (36 us); integer sums, so
// the strip height does not change the result.
#define TEX_RPW 8
__global__ __launch_bounds__(256) void k_textureness(const int *S, int sld, unsigned char *disp, long long dstep, int rows, int cols,
                                                     int winsz, float threshold)
{
    __shared__ int colsum[4][64 + 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int W2 = winsz / 2;   // <= 25
    const int x = blockIdx.x * 64 + lane;
    const int yb = (blockIdx.y * 4 + wv) * TEX_RPW;
    if (yb >= rows) return;
    const int ye = min(yb + TEX_RPW, rows);
    const int xh = lane < 32 ? blockIdx.x * 64 - 32 + lane : blockIdx.x * 64 + 64 + (lane - 32);
    const int *S0 = S + (long long)TEX_MY * sld + TEX_MX + x, *S1 = S + (long long)TEX_MY * sld + TEX_MX + xh;
    int s0 = 0, s1 = 0;
    for (int i = -W2; i <= W2; ++i) {
        s0 += S0[(long long)(yb + i) * sld];
        s1 += S1[(long long)(yb + i) * sld];
    }
    int *cs = colsum[wv];
    for (int y = yb; y < ye; ++y) {
        cs[32 + lane] = s0;
        cs[lane < 32 ? lane : 64 + lane] = s1;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (x < cols) {
            const long long o = (long long)y * dstep + x;
            if (disp[o]) {
                long long sum = 0;
                for (int j = -W2; j <= W2; ++j) sum += cs[32 + lane + j];
                if ((float)sum * 0.25f < threshold) disp[o] = 0;
            }
        }
        __builtin_amdgcn_wave_barrier();
        s0 += S0[(long long)(y + 1 + W2) * sld] - S0[(long long)(y - W2) * sld];
        s1 += S1[(long long)(y + 1 + W2) * sld] - S1[(long long)(y - W2) * sld];
    }
}

// The two passes above in ONE launch and without the |Sobel| plane (round 4): the S plane cost 8.8 MB written and 28 MB read per 1080p
// pair (its rows fetched once per XCD) and the post-filter two launches per pair -- 27 of the ~180 us a pair of a batch took.  A workgroup
// owns 64 x 32 pixels: it forms B on the tile extended by W2 + 1 (texel-wise clamped reads, tex_box4), S = |x-Sobel of B| on the tile
// extended by W2, both as 16-bit words in LDS (B <= 1020, S <= 4080), then every wave slides the column sums of its 8 rows exactly
// as k_textureness does.  Integer arithmetic throughout: the same sums, the same comparison, the same disparities.  blockIdx.z = pair
// of a batch (images and maps from the block matcher's table).
#define TEXF_ROWS 32
__global__ __launch_bounds__(256) void k_textureness_fused(const unsigned char *img, long long istep, unsigned char *disp, long long dstep,
                                                           const BmPair *tab, int rows, int cols, int winsz, float threshold)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (tab) {
        const BmPair p = tab[blockIdx.z];
        img = p.left; istep = p.lstep; disp = p.disp; dstep = p.dstep;
    }
    const int W2 = winsz / 2;                          // <= 25
    const int TC = 64 + 2 * W2, TRR = TEXF_ROWS + 2 * W2;   // S tile
    const int BC = TC + 2, BR = TRR + 2;               // B tile
    unsigned short *Bs = reinterpret_cast<unsigned short *>(smem);
    unsigned short *Ss = Bs + BR * BC;
    int *cs = reinterpret_cast<int *>(Ss + ((TRR * TC + 1) & ~1)) + (threadIdx.x >> 6) * 128;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * TEXF_ROWS;
    for (int i = threadIdx.x; i < BR * BC; i += 256) {
        const int r = i / BC, c = i - r * BC;
        Bs[i] = (unsigned short)tex_box4(img, istep, rows, cols, x0 - W2 - 1 + c, y0 - W2 - 1 + r);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TRR * TC; i += 256) {
        const int r = i / TC, c = i - r * TC;
        const unsigned short *b = Bs + r * BC + c;     // B(x - 1, y - 1) of S(x, y)
        const int v = -(int)b[0] + (int)b[2] - 2 * (int)b[BC] + 2 * (int)b[BC + 2] - (int)b[2 * BC] + (int)b[2 * BC + 2];
        Ss[i] = (unsigned short)abs(v);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = x0 + lane;
    const int rb = wv * (TEXF_ROWS / 4);               // this wave's first row inside the tile
    const int yb = y0 + rb;
    if (yb >= rows) return;
    const int ye = min(yb + TEXF_ROWS / 4, rows);
    // column sums of S columns `lane` and `64 + lane` of the tile (the window of output column lane covers tile columns lane .. lane + 2 W2)
    const bool has1 = 64 + lane < TC;
    int s0 = 0, s1 = 0;
    for (int i = 0; i <= 2 * W2; ++i) {
        s0 += Ss[(rb + i) * TC + lane];
        if (has1) s1 += Ss[(rb + i) * TC + 64 + lane];
    }
    for (int y = yb; y < ye; ++y) {
        const int r = y - y0;
        cs[lane] = s0;
        cs[64 + lane] = s1;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (x < cols) {
            const long long o = (long long)y * dstep + x;
            if (disp[o]) {
                long long sum = 0;
                for (int j = 0; j <= 2 * W2; ++j) sum += cs[lane + j];
                if ((float)sum * 0.25f < threshold) disp[o] = 0;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (y + 1 < ye) {
            s0 += (int)Ss[(r + 1 + 2 * W2) * TC + lane] - (int)Ss[r * TC + lane];
            if (has1) s1 += (int)Ss[(r + 1 + 2 * W2) * TC + 64 + lane] - (int)Ss[r * TC + 64 + lane];
        }
    }
}

// ------------------------------------------------------------------ host launchers
template <int R>
static int launch_bm(const BmArgs &A, int mode, hipStream_t s)
{
    using C = Cfg<R>;
    const int RS = (C::NC + A.nsets * 64 - 1 + 3) / 4 * 4 + 4;
    size_t lds = (size_t)(A.rb + 2 * R) * (C::LS + RS) + (size_t)2 * A.nsets * 128 * sizeof(unsigned);
    lds = (lds + 15) / 16 * 16;   // the transposition buffer behind it is read as b128
    const dim3 grid(div_up(A.cols - A.ndisp - 2 * R, C::TW), div_up(A.rows - 2 * R, A.rb), A.tab ? A.batch : 1);
    const dim3 block(64 * A.nsets);
    // one first-pass kernel per radius in the release library: the transposed winner-take-all where the column sums are packed (C::PACKED),
    // the plain form elsewhere; the experiments build keeps both for MIFLOW_SBM_WT
#ifdef MIFLOW_EXPERIMENTS
    const bool wt = C::PACKED && tuning().sbm_wt;
#else
    constexpr bool wt = C::PACKED;
#endif
    if (mode == 0 && wt) {
        if constexpr (C::PACKED) {
            const size_t ldst = lds + (size_t)A.nsets * 16 * 68 * sizeof(unsigned);
            (void)hipFuncSetAttribute((const void *)k_block_match<R, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldst);
            hipLaunchKernelGGL((k_block_match<R, 0, true>), grid, block, ldst, s, A);
        }
    } else if (mode == 0) {
#ifndef MIFLOW_EXPERIMENTS
        if constexpr (!C::PACKED)
#endif
        {
            (void)hipFuncSetAttribute((const void *)k_block_match<R, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k_block_match<R, 0>), grid, block, lds, s, A);
        }
    } else {
        (void)hipFuncSetAttribute((const void *)k_block_match<R, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_block_match<R, 1>), grid, block, lds, s, A);
    }
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

template <int R>
static int tile_w() { return Cfg<R>::TW; }

typedef int (*bm_launch_t)(const BmArgs &, int, hipStream_t);
#define L1(r) launch_bm<r>
static const bm_launch_t g_bm[26] = {nullptr, L1(1), L1(2), L1(3), L1(4), L1(5), L1(6), L1(7), L1(8), L1(9), L1(10),
                                     L1(11), L1(12), L1(13), L1(14), L1(15), L1(16), L1(17), L1(18), L1(19), L1(20),
                                     L1(21), L1(22), L1(23), L1(24), L1(25)};

static int tile_w_of(int R)
{
    const int t = (64 - 2 * R) & ~3;
    return t < 16 ? 16 : t;
}

static int block_match_impl(BmArgs &A, int winsz, int uniqueness_ratio, int pairs, hipStream_t s)
{
    const int R = winsz >> 1;
    MI_REQUIRE(R >= 1 && R <= 25, MI_ERR_BAD_ARG, "Unsupported window size");   // stereobm.cu:503-504
    A.nsets = div_up(A.ndisp, 64);
    A.thresh_scale = (float)(1.0 + uniqueness_ratio / 100.0f);
    // rows per band: enough waves for the 1024 SIMDs (target ~5 per SIMD) but keep the 2R-row start-up of every band (column
    // sums of the first window) a modest fraction of the work.  A batch supplies the waves, so its bands can be taller.
    const int xt = div_up(A.cols - A.ndisp - 2 * R, tile_w_of(R));
    const int vrows = A.rows - 2 * R;
    int bands = div_up(5120, xt * A.nsets * pairs);   // r01f sweep: ~16 rows per band is the optimum for ONE 1080p / 128-disparity pair
    int rb = div_up(vrows, bands > 0 ? bands : 1);
    rb = rb < 2 * R + 2 ? 2 * R + 2 : rb;
    rb = rb > 48 ? 48 : rb;   // taller bands cost occupancy (LDS per workgroup grows with rb): r02w at 1080p x 16: 32 | 48 | 64 | 96 rows = 4990 | 5070 | . | 4575 pairs/s
    if (const char *e = MI_EXP_ENV("MIFLOW_SBM_ROWS")) rb = atoi(e) > 0 ? atoi(e) : rb;
    A.rb = rb;
    A.swz = tuning().sbm_swz != 0 ? 1 : 0;
    MI_REQUIRE(uniqueness_ratio <= 0 || A.minssd, MI_ERR_BAD_ARG, "the uniqueness test needs the winners' SSD plane");
    int rc = g_bm[R](A, 0, s);
    if (rc) return rc;
    if (uniqueness_ratio > 0) rc = g_bm[R](A, 1, s);
    return rc;
}

int block_match(const unsigned char *left, long long lstep, const unsigned char *right, long long rstep, unsigned char *disp,
                long long dstep, unsigned *minssd, long long mstep, int rows, int cols, int ndisp, int winsz,
                int uniqueness_ratio, int emulate_edge, hipStream_t s)
{
    BmArgs A;
    memset(&A, 0, sizeof(A));
    A.left = left; A.right = right; A.lstep = lstep; A.rstep = rstep; A.disp = disp; A.dstep = dstep;
    A.minssd = minssd; A.mstep = mstep; A.rows = rows; A.cols = cols; A.ndisp = ndisp;
    A.emulate_edge = emulate_edge;
    return block_match_impl(A, winsz, uniqueness_ratio, 1, s);
}

int block_match_batch(const BmPair *tab_dev, int pairs, unsigned *minssd, long long mstep, long long mpair, int rows, int cols, int ndisp,
                      int winsz, int uniqueness_ratio, int emulate_edge, hipStream_t s)
{
    BmArgs A;
    memset(&A, 0, sizeof(A));
    A.tab = tab_dev; A.batch = pairs; A.mpair = mpair;
    A.minssd = minssd; A.mstep = mstep; A.rows = rows; A.cols = cols; A.ndisp = ndisp;
    A.emulate_edge = emulate_edge;
    return block_match_impl(A, winsz, uniqueness_ratio, pairs, s);
}

int prefilter_xsobel(const unsigned char *src, long long sstep, unsigned char *dst, long long dstep, int rows, int cols,
                     int cap, hipStream_t s)
{
    hipLaunchKernelGGL(k_prefilter_xsobel, dim3(div_up(cols, 64), div_up(rows, 4)), dim3(256), 0, s, src, sstep, dst, dstep,
                       rows, cols, cap);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int prefilter_norm(const unsigned char *src, long long sstep, unsigned char *dst, long long dstep, int rows, int cols,
                   int cap, int winsize, hipStream_t s)
{
    MI_REQUIRE(winsize >= 1 && winsize / 2 <= 32, MI_ERR_BAD_ARG, "preFilterSize must be in [1,65]");
    MI_REQUIRE(winsize * winsize / 8 > 0, MI_ERR_BAD_ARG, "preFilterSize too small (integer scale is 0)");
    int scale_g = winsize * winsize / 8, scale_s = (1024 + scale_g) / (scale_g * 2);   // stereobm.cu:591-592
    scale_g *= scale_s;
    hipLaunchKernelGGL(k_prefilter_norm, dim3(div_up(cols, 64), div_up(div_up(rows, 32), 4)), dim3(256), 0, s, src, sstep, dst,
                       dstep, rows, cols, cap, scale_g, scale_s, winsize);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

void textureness_scratch_dims(int rows, int cols, int *sld, int *sh)
{
    *sld = align_up(cols, 64) + 2 * TEX_MX;
    *sh = rows + 2 * TEX_MY;
}

int textureness(const unsigned char *img, long long istep, unsigned char *disp, long long dstep, int rows, int cols,
                int winsz, float avg_threshold, int *S, hipStream_t s)
{
    const float threshold = avg_threshold * (float)(winsz * winsz);   // stereobm.cu:700
    int sld, sh;
    textureness_scratch_dims(rows, cols, &sld, &sh);
    hipLaunchKernelGGL(k_tex_sobel, dim3(div_up(sld, 64), div_up(sh, 4)), dim3(256), 0, s, img, istep, rows, cols, S, sld, sh);
    hipLaunchKernelGGL(k_textureness, dim3(div_up(cols, 64), div_up(div_up(rows, TEX_RPW), 4)), dim3(256), 0, s, (const int *)S, sld, disp,
                       dstep, rows, cols, winsz, threshold);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

// disp := 0 for every pair of a batch in one launch (stereobm.cu:506 does it per pair; n small fills were n dependent launches)
__global__ __launch_bounds__(256) void k_zero_disp_batch(const BmPair *tab, int rows, int cols)
{
    const BmPair p = tab[blockIdx.z];
    const int y = blockIdx.y;
    unsigned char *row = p.disp + (long long)y * p.dstep;
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 16;
    if (x4 >= cols) return;
    if (x4 + 16 <= cols && ((reinterpret_cast<unsigned long long>(row + x4) & 15) == 0)) {
        *reinterpret_cast<uint4 *>(row + x4) = make_uint4(0, 0, 0, 0);
    } else {
        for (int k = 0; k < 16 && x4 + k < cols; ++k) row[x4 + k] = 0;
    }
}
int zero_disp_batch(const BmPair *tab_dev, int pairs, int rows, int cols, hipStream_t s)
{
    hipLaunchKernelGGL(k_zero_disp_batch, dim3(div_up(cols, 256 * 16), rows, pairs), dim3(256), 0, s, tab_dev, rows, cols);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

// the fused form (k_textureness_fused); tab_dev != null: `pairs` pairs of the block matcher's table in one launch
int textureness_fused(const unsigned char *img, long long istep, unsigned char *disp, long long dstep, const BmPair *tab_dev, int pairs,
                      int rows, int cols, int winsz, float avg_threshold, hipStream_t s)
{
    const float threshold = avg_threshold * (float)(winsz * winsz);   // stereobm.cu:700
    const int W2 = winsz / 2;
    MI_REQUIRE(W2 >= 0 && W2 <= 25, MI_ERR_BAD_ARG, "Unsupported window size");
    const int TC = 64 + 2 * W2, TRR = TEXF_ROWS + 2 * W2;
    const size_t lds = sizeof(unsigned short) * ((size_t)(TRR + 2) * (TC + 2) + (((size_t)TRR * TC + 1) & ~(size_t)1)) + sizeof(int) * 4 * 128;
    (void)hipFuncSetAttribute((const void *)k_textureness_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_textureness_fused, dim3(div_up(cols, 64), div_up(rows, TEXF_ROWS), tab_dev ? pairs : 1), dim3(256), lds, s, img, istep, disp,
                       dstep, tab_dev, rows, cols, winsz, threshold);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int dbg_tmax16(const unsigned *in_dev, unsigned *out_dev, hipStream_t s)
{
    hipLaunchKernelGGL(k_dbg_tmax16, dim3(1), dim3(64), 0, s, in_dev, out_dev);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

int dbg_wave_min(const unsigned *in_dev, unsigned *out_dev, hipStream_t s)
{
    hipLaunchKernelGGL(k_dbg_wave_min, dim3(1), dim3(64), 0, s, in_dev, out_dev);
    MI_HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace sbm
}  // namespace mi
