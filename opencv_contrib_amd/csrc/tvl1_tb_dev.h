// Shared device helpers of the temporally blocked TV-L1 kernels (tvl1_tb_kernels.hip, tvl1_tbr_kernels.hip).
#pragma once
#include "tvl1_dev.h"

namespace mi {
namespace tvl1 {

// lane n <- lane n-1 (wave_shr:1) / lane n <- lane n+1 (wave_shl:1); semantics verified on HW
// by tests/test_tvl1_gpu.py::test_dpp_wave_shift_semantics through mi_dbg_lane_shift.
__device__ __forceinline__ float dpp_from_prev(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_next(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

// Dynamic state of one pipeline stage: u_t(a) and p_(t-1)(a) of the row it holds.
template <int PPL>
struct Dyn {
    float u1[PPL], u2[PPL], p11[PPL], p12[PPL], p21[PPL], p22[PPL];
};
template <int PPL>
struct Stat {  // static planes of one row: I1wx, I1wy, 1/grad, rho_c
    float ix[PPL], iy[PPL], rg[PPL], rc[PPL];
};

struct TbArgs {
    IterPlanes pl;
    Geo g;
    float l_t, theta, taut;
    int rows_per_band;
    int cur;  // input set
    int swz, nstrips;
};

// per-wave LDS ring of static rows: slot layout [plane 0..3][64*PPL floats]
template <int PPL>
__device__ __forceinline__ void lds_put(float *slot, int lane, const Stat<PPL> &s)
{
    float *q = slot + lane * PPL;
    if (PPL == 1) {
        q[0] = s.ix[0]; q[64] = s.iy[0]; q[128] = s.rg[0]; q[192] = s.rc[0];
    } else if (PPL == 2) {
        *reinterpret_cast<float2 *>(q) = make_float2(s.ix[0], s.ix[1]);
        *reinterpret_cast<float2 *>(q + 128) = make_float2(s.iy[0], s.iy[1]);
        *reinterpret_cast<float2 *>(q + 256) = make_float2(s.rg[0], s.rg[1]);
        *reinterpret_cast<float2 *>(q + 384) = make_float2(s.rc[0], s.rc[1]);
    } else {
        *reinterpret_cast<float4 *>(q) = make_float4(s.ix[0], s.ix[1], s.ix[2], s.ix[3]);
        *reinterpret_cast<float4 *>(q + 256) = make_float4(s.iy[0], s.iy[1], s.iy[2], s.iy[3]);
        *reinterpret_cast<float4 *>(q + 512) = make_float4(s.rg[0], s.rg[1], s.rg[2], s.rg[3]);
        *reinterpret_cast<float4 *>(q + 768) = make_float4(s.rc[0], s.rc[1], s.rc[2], s.rc[3]);
    }
}
template <int PPL>
__device__ __forceinline__ void lds_get(const float *slot, int lane, Stat<PPL> &s)
{
    const float *q = slot + lane * PPL;
    if (PPL == 1) {
        s.ix[0] = q[0]; s.iy[0] = q[64]; s.rg[0] = q[128]; s.rc[0] = q[192];
    } else if (PPL == 2) {
        float2 a = *reinterpret_cast<const float2 *>(q), b = *reinterpret_cast<const float2 *>(q + 128);
        float2 c = *reinterpret_cast<const float2 *>(q + 256), d = *reinterpret_cast<const float2 *>(q + 384);
        s.ix[0] = a.x; s.ix[1] = a.y; s.iy[0] = b.x; s.iy[1] = b.y;
        s.rg[0] = c.x; s.rg[1] = c.y; s.rc[0] = d.x; s.rc[1] = d.y;
    } else {
        float4 a = *reinterpret_cast<const float4 *>(q), b = *reinterpret_cast<const float4 *>(q + 256);
        float4 c = *reinterpret_cast<const float4 *>(q + 512), d = *reinterpret_cast<const float4 *>(q + 768);
        s.ix[0] = a.x; s.ix[1] = a.y; s.ix[2] = a.z; s.ix[3] = a.w;
        s.iy[0] = b.x; s.iy[1] = b.y; s.iy[2] = b.z; s.iy[3] = b.w;
        s.rg[0] = c.x; s.rg[1] = c.y; s.rg[2] = c.z; s.rg[3] = c.w;
        s.rc[0] = d.x; s.rc[1] = d.y; s.rc[2] = d.z; s.rc[3] = d.w;
    }
}

// Row prefetch.  The loads are UNCONDITIONAL (clamped row / column, uniform row base + 32-bit lane offset): a load inside a
// divergent `if` sits in its own basic block, and the waitcnt pass then has to assume vmcnt(0) at every later use, which
// serialises the prefetch (r01k ISA).  Out-of-image rows/columns are zeroed when the row is consumed (mask_input_row).
// Pins a wave-uniform row pointer into an SGPR pair so that the access is emitted as `global_* v, voffset, s[base:base+1]`
// (otherwise base + lane offset is reassociated into per-plane 64-bit VGPR addresses hoisted out of the row loop: 32 VGPRs
// and two VALU adds per access).  The integer round trip drops the inferred address space, hence the explicit global one.
#define MI_GLOBAL __attribute__((address_space(1)))
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ MI_GLOBAL char *sgpr_row(const void *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (MI_GLOBAL char *)(((unsigned long long)hi << 32) | lo);
}
// Row prefetch.  The loads are UNCONDITIONAL (clamped row / column): a load inside a divergent `if` sits in its own basic
// block, and the waitcnt pass then has to assume vmcnt(0) at every later use, which serialises the prefetch (r01k ISA).
// Out-of-image rows/columns are zeroed when the row is consumed (mask_input_row).  xb = lane offset in BYTES.
template <int PPL>
__device__ __forceinline__ void ldu(float dst[PPL], const float *rowp, unsigned xb)
{
    const MI_GLOBAL char *q = sgpr_row(rowp) + xb;
    if (PPL == 1) {
        dst[0] = *(const MI_GLOBAL float *)q;
    } else if (PPL == 2) {
        const f2v v = *(const MI_GLOBAL f2v *)q;
        dst[0] = v.x; dst[1] = v.y;
    } else {
        const f4v v = *(const MI_GLOBAL f4v *)q;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
}
template <int PPL>
__device__ __forceinline__ void stu(float *rowp, unsigned xb, const float v[PPL])
{
    MI_GLOBAL char *q = sgpr_row(rowp) + xb;
    if (PPL == 1) *(MI_GLOBAL float *)q = v[0];
    else if (PPL == 2) *(MI_GLOBAL f2v *)q = f2v{v[0], v[1]};
    else *(MI_GLOBAL f4v *)q = f4v{v[0], v[1], v[2], v[3]};
}

// 1/grad; grad == 0 -> huge, so that clamp() yields -+l_t*sign(rho) like the reference's first two branches
template <int PPL>
__device__ __forceinline__ void finish_static(Stat<PPL> &st)
{
#pragma unroll
    for (int j = 0; j < PPL; ++j) st.rg[j] = __builtin_amdgcn_rcpf(fmaxf(st.rg[j], 1e-30f));
}

// rotating-slot kernels (tvl1_tbr_kernels.hip): first entry of time block T, or the one matching (ppl, wps, pf) when >= 0
typedef void (*TbLaunch)(const TbArgs &, bool, hipStream_t);
TbLaunch tbr_pick(int T, int want_ppl, int want_wps, int want_pf, int *ppl, int *wps, int *pf);

}  // namespace tvl1
}  // namespace mi
