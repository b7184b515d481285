// Micro-benchmark (tuning aid, not part of the product): what paces a wave whose instruction stream looks like a stage of the
// blocked TV-L1 kernel -- VALU pipe throughput, the per-wave issue interval, or the non-VALU instructions (SALU, LDS, s_waitcnt)
// riding in the same stream -- and at what CLOCK the chip runs such a stream.  Every wave brackets its loop with s_memtime (shader
// clock ticks) and s_memrealtime (constant 100 MHz), so cycles are measured, not derived from a nominal 2.4 GHz:
//     effective clock = d(memtime) / d(memrealtime) x 100 MHz.
//   hipcc --offload-arch=gfx950 -O2 -o issue_mix issue_mix.hip && ./issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

struct Stamp { unsigned long long c0, c1, r0, r1; };

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, Stamp *st, int iters, float seed)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x + i;
    const float b = seed * 0.5f + 1.0f, c = seed * 0.25f;
    extern __shared__ float lds[];
    lds[threadIdx.x] = seed;
    __syncthreads();
    unsigned la = (threadIdx.x & 63) * 16;
    unsigned s0 = iters, s1 = 3, s2 = 5, s3 = 7;
    unsigned long long msk = (threadIdx.x & 63) == 63 ? 0x8000000000000000ull : 0ull;
    msk = __builtin_amdgcn_readfirstlane((unsigned)(iters == 12345)) ? ~0ull : 0x8000000000000000ull;
    float4 q = make_float4(seed, seed, seed, seed);
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v pk[8];
    for (int i = 0; i < 8; ++i) pk[i] = f2v{seed + i, seed - i};
    const f2v pkb = {1.0f + seed * 1e-3f, 1.0f - seed * 1e-3f}, pkc = {seed * 1e-3f, -seed * 1e-3f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define SADD(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
#define SSEL(i) asm volatile("s_cmp_eq_u32 %1, %2\n s_cselect_b32 %0, %1, %2" : "=s"(s3) : "s"(s0), "s"(s2) : "scc");
#define FMA_S(i) FMA(i) SADD(i)
#define FMA2_S(i) FMA(i) FMA((i + 1) & 7) SADD(i)
#define FMA_SS(i) FMA(i) SADD(i) SSEL(i)
#define NOP(i) asm volatile("s_nop 0");
#define FMA_NOP(i) FMA(i) NOP(i)
#define WAIT0(i) asm volatile("s_waitcnt lgkmcnt(0)");
#define FMA_WAIT(i) FMA(i) WAIT0(i)
#define DSR4 asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(la));
#define DSW2M asm volatile("s_mov_b64 exec, %2\n ds_write_b64 %0, %1 offset:4096\n s_mov_b64 exec, -1" : : "v"(la), "v"(*(double *)&q), "s"(msk) : "memory");
#define DSW2 asm volatile("ds_write_b64 %0, %1 offset:4096" : : "v"(la), "v"(*(double *)&q) : "memory");
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
// one "stage": 36 plain VALU, 4 DPP moves, 2 sqrt, 2 rcp = 44 VALU in a mostly dependent order
#define STAGE_VALU REP8(FMA) DPP(0) DPP(1) REP8(FMA) DPP(2) DPP(3) REP8(FMA) FMA(0) FMA(1) FMA(2) FMA(3) SQRT(4) FMA(5) SQRT(6) FMA(7) FMA(4) RCP(4) FMA(6) RCP(6) REP8(FMA)
// the same chain with ILP 1: every instruction depends on its predecessor
#define F1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
#define F8 F1 F1 F1 F1 F1 F1 F1 F1
        if (OP == 0) { REP8(FMA) REP8(FMA) }                                       // 16 VALU
        if (OP == 1) { REP8(FMA2_S) }                                              // 16 VALU + 8 SALU
        if (OP == 2) { REP8(FMA_S) REP8(FMA_S) }                                   // 16 VALU + 16 SALU
        if (OP == 3) { REP8(FMA_SS) REP8(FMA_SS) }                                 // 16 VALU + 48 SALU
        if (OP == 4) { REP8(FMA_NOP) REP8(FMA_NOP) }                               // 16 VALU + 16 s_nop
        if (OP == 5) { REP8(FMA_WAIT) REP8(FMA_WAIT) }                             // 16 VALU + 16 s_waitcnt (nothing outstanding)
        if (OP == 6) { F8 F8 }                                                     // 16 VALU, ONE dependent chain
        if (OP == 7) { STAGE_VALU }                                                // 44 VALU, stage-like
        if (OP == 8) { STAGE_VALU REP8(SADD) REP8(SSEL) }                          // + 24 SALU at the end
        if (OP == 9) { DSR4 REP8(FMA) WAIT0(0) REP8(FMA) DSW2 REP8(FMA) DSW2 REP8(FMA) FMA(0) FMA(1) FMA(2) FMA(3) SQRT(4) FMA(5) SQRT(6) FMA(7) FMA(4) RCP(4) FMA(6) RCP(6) }   // 44 VALU + 1 LDS read + 2 LDS writes (full wave)
        if (OP == 10) { DSR4 REP8(FMA) WAIT0(0) REP8(FMA) DSW2M REP8(FMA) DSW2M REP8(FMA) FMA(0) FMA(1) FMA(2) FMA(3) SQRT(4) FMA(5) SQRT(6) FMA(7) FMA(4) RCP(4) FMA(6) RCP(6) }   // the same with exec-masked one-lane writes
        if (OP == 11) { REP8(SADD) REP8(SADD) }                                    // 16 SALU only
#define G0 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
#define G1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(b), "v"(c));
#define G2 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[2]) : "v"(b), "v"(c));
#define G3 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(b), "v"(c));
        if (OP == 12) { G0 G1 G0 G1 G0 G1 G0 G1 G0 G1 G0 G1 G0 G1 G0 G1 }          // two chains alternating
        if (OP == 13) { G0 G1 G2 G0 G1 G2 G0 G1 G2 G0 G1 G2 G0 G1 G2 G0 }          // three chains
        if (OP == 14) { G0 G1 G2 G3 G0 G1 G2 G3 G0 G1 G2 G3 G0 G1 G2 G3 }          // four chains
        // transcendental result consumed after k independent fillers (chains 1..3 are the fillers)
#define SQ0 asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[0]));
#define RC0 asm volatile("v_rcp_f32 %0, %0" : "+v"(a[0]));
        if (OP == 15) { SQ0 G0 SQ0 G0 SQ0 G0 SQ0 G0 SQ0 G0 SQ0 G0 SQ0 G0 SQ0 G0 }                          // k = 0
        if (OP == 16) { SQ0 G1 G0 G2 SQ0 G1 G0 G2 SQ0 G1 G0 G2 SQ0 G1 G0 G2 }                              // k = 1 (and one after)
        if (OP == 17) { SQ0 G1 G2 G0 SQ0 G1 G2 G0 SQ0 G1 G2 G0 SQ0 G1 G2 G0 }                              // k = 2
        if (OP == 18) { SQ0 G1 G2 G3 G0 G1 G2 G3 SQ0 G1 G2 G3 G0 G1 G2 G3 }                                // k = 3
        // sqrt -> fma -> rcp -> mul as in the stage tail, two components interleaved
#define SQ1 asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[1]));
#define RC1 asm volatile("v_rcp_f32 %0, %0" : "+v"(a[1]));
        if (OP == 19) { SQ0 G0 RC0 G0 SQ1 G1 RC1 G1 SQ0 G0 RC0 G0 SQ1 G1 RC1 G1 }                          // serial, as compiled today
        if (OP == 20) { SQ0 SQ1 G0 G1 RC0 RC1 G0 G1 SQ0 SQ1 G0 G1 RC0 RC1 G0 G1 }                          // components interleaved
        if (OP == 21) { SQ0 SQ1 G2 G3 G0 G1 G2 G3 RC0 RC1 G2 G3 G0 G1 G2 G3 }                              // + two fillers per gap
        // DPP move of a value the previous VALU wrote (the compiler's s_nop included), and of an older value
#define DP0 asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[0]) : "v"(a[0]));
#define DP01 asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[0]) : "v"(a[1]));
        if (OP == 22) { G0 DP0 G0 DP0 G0 DP0 G0 DP0 G0 DP0 G0 DP0 G0 DP0 G0 DP0 }                          // fma -> dpp(dep) -> fma(dep)
        if (OP == 23) { G1 G2 DP01 G0 G1 G2 DP01 G0 G1 G2 DP01 G0 G1 G2 DP01 G0 }                          // dpp source two instructions old
        // SALU result consumed by the next VALU
#define SELF asm volatile("s_cmp_eq_u32 %1, %2\n s_cselect_b32 %0, %1, %2" : "=s"(s3) : "s"(s0), "s"(s2) : "scc");
#define FMSG(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s3), "v"(c));
#define DSRF(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(la), "n"(i * 1024));
        if (OP == 25) { DSRF(0) FMA(0) DSRF(1) FMA(1) DSRF(2) FMA(2) DSRF(3) FMA(3) DSRF(4) FMA(4) DSRF(5) FMA(5) DSRF(6) FMA(6) DSRF(7) FMA(7)
                        DSRF(0) FMA(0) DSRF(1) FMA(1) DSRF(2) FMA(2) DSRF(3) FMA(3) DSRF(4) FMA(4) DSRF(5) FMA(5) DSRF(6) FMA(6) DSRF(7) FMA(7) WAIT0(0) }
        if (OP == 26) { REP8(DPP) REP8(DPP) }
        if (OP == 27) { REP8(SQRT) REP8(RCP) }
#define PKF(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[i]) : "v"(pkb), "v"(pkc));
        if (OP == 28) { REP8(PKF) REP8(PKF) }
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 29) { REP8(MUL) REP8(MUL) }
        if (OP == 30) { REP8(ADD) REP8(ADD) }
        // integer / byte ops of the StereoBM row loop (each: 16 instructions over 8 independent registers)
        unsigned *ia = (unsigned *)a;
        const unsigned ib = __float_as_uint(b), ic = __float_as_uint(c);
#define IOP(name, text) if (OP == name) { for (int rpt = 0; rpt < 2; ++rpt) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(text : "+v"(ia[i]) : "v"(ib), "v"(ic), "s"(msk)); } }
        IOP(40, "v_sub_u32_e32 %0, %0, %1")
        IOP(41, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2")
        IOP(42, "v_mad_i32_i24 %0, %0, %1, %2")
        IOP(43, "v_mul_i32_i24_e32 %0, %0, %1")
        IOP(44, "v_add_u32_e32 %0, %0, %1")
        IOP(45, "v_lshl_or_b32 %0, %0, 6, %1")
        IOP(46, "v_max_u32_e32 %0, %0, %1")
        IOP(47, "v_max3_u32 %0, %0, %1, %2")
        IOP(48, "v_alignbyte_b32 %0, %0, %1, 1")
        IOP(49, "v_cndmask_b32_e64 %0, %0, %1, %3")
        IOP(50, "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        IOP(51, "v_mad_u32_u24 %0, %0, %1, %2")
        IOP(52, "v_and_b32_e32 %0, %0, %1")
        IOP(53, "v_bfe_u32 %0, %0, 8, 8")
        IOP(54, "v_perm_b32 %0, %0, %1, %2")
        IOP(55, "v_dot4_u32_u8 %0, %0, %1, %2")
        IOP(56, "v_sad_u8 %0, %0, %1, %2")
        IOP(57, "v_pk_mul_lo_u16 %0, %0, %1")
        IOP(58, "v_pk_sub_i16 %0, %0, %1")
        IOP(59, "v_mad_u32_u16 %0, %0, %1, %2")
        IOP(60, "v_cvt_f32_ubyte1 %0, %0")
        IOP(61, "v_cvt_u32_f32_e32 %0, %0")
        IOP(62, "v_lshrrev_b32_e32 %0, 8, %0")
        IOP(63, "v_lshlrev_b32_e32 %0, 6, %0")
        IOP(64, "v_or_b32_e32 %0, %0, %1")
        IOP(65, "v_sub_f32_e32 %0, %0, %1")
        IOP(66, "v_fma_f32 %0, -%0, %1, %2")
        IOP(67, "v_cvt_f32_u32_e32 %0, %0")
        IOP(68, "v_sub_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")
        IOP(69, "v_min_u32_e32 %0, %0, %1")
        IOP(70, "v_cvt_f32_ubyte0_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2")
        IOP(71, "v_mul_u32_u24_e32 %0, %0, %1")
        IOP(72, "v_mul_lo_u32 %0, %0, %1")
        IOP(73, "v_max_f32_e32 %0, %0, %1")
        IOP(74, "v_xor_b32_e32 %0, %0, %1")
        IOP(75, "v_subrev_u32_e32 %0, %0, %1")
        if (OP == 24) { SELF FMSG(0) SELF FMSG(1) SELF FMSG(2) SELF FMSG(3) SELF FMSG(4) SELF FMSG(5) SELF FMSG(6) SELF FMSG(7) }   // 8 x (cmp, cselect, dependent fma)
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = q.x + q.y;
    for (int i = 0; i < 8; ++i) s += pk[i].x + pk[i].y;
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 123.456f || s0 == 77 || s3 == 99) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1};
}

template <int OP>
static void run(const char *name, int valu, int other)
{
    float *out;
    Stamp *st;
    hipMalloc(&out, 4096);
    hipMalloc(&st, sizeof(Stamp) * 256 * 8 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000 * 16 / (valu + other > 0 ? valu + other : 16);
    printf("%-34s (%2d VALU + %2d other per iteration)\n", name, valu, other);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 16384, 0, out, st, 64, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 16384, 0, out, st, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<Stamp> h(blocks * 4);
        hipMemcpy(h.data(), st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        for (auto &s : h) { cyc += (double)(s.c1 - s.c0); rt += (double)(s.r1 - s.r0); }
        cyc /= h.size(); rt /= h.size();
        const double ghz = cyc / (rt / 100e6) / 1e9;                  // ticks per second of the 100 MHz real-time counter
        const double n_instr = (double)iters * (valu + other);
        // SIMD cycles per instruction of ONE wave's stream = wave cycles / instructions; per SIMD = / waves per SIMD
        printf("    wps%d: clock %.3f GHz | wave: %6.2f cyc/instr | SIMD: %5.2f cyc/instr, %5.2f cyc/VALU | wall %.1f us (nominal-2.4 view %5.2f cyc/instr)\n",
               wps, ghz, cyc / n_instr, cyc / n_instr / wps, valu ? cyc / ((double)iters * valu) / wps : 0.0, ms * 1e3,
               ms * 1e-3 * 2.4e9 / n_instr / wps);
    }
    hipFree(out); hipFree(st);
}

// sustained mode: `issue_mix sustain <op> <wps> <seconds>` keeps one stream on the chip long enough for the power management to settle;
// poll rocm-smi beside it (tools/clock_poll.sh) for the socket power: energy per lane-instruction = (P - P_idle) / (lane-instructions / s)
template <int OP>
static void sustain(const char *name, int valu, int other, int wps, double seconds)
{
    float *out;
    Stamp *st;
    hipMalloc(&out, 4096);
    hipMalloc(&st, sizeof(Stamp) * 256 * 8 * 4);
    const int iters = 200000 * 16 / (valu + other);
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 16384, 0, out, st, 64, 1.0f);
    hipDeviceSynchronize();
    double total_ms = 0; long launches = 0;
    double ghz = 0;
    while (total_ms < seconds * 1e3) {
        hipEventRecord(e0);
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 16384, 0, out, st, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms; launches += 4;
    }
    std::vector<Stamp> h(blocks * 4);
    hipMemcpy(h.data(), st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (auto &s : h) { cyc += (double)(s.c1 - s.c0); rt += (double)(s.r1 - s.r0); }
    ghz = cyc / (rt / 100e6) / 1e9;
    const double winstr = (double)launches * blocks * 4 * iters * (valu + other);   // wave-instructions
    printf("sustain %-30s wps%d: %.1f s, clock %.3f GHz (last launch), %.2f T lane-instr/s (%.2f T VALU lane-instr/s), %.2f cyc/instr per SIMD at that clock\n",
           name, wps, total_ms * 1e-3, ghz, winstr * 64 / (total_ms * 1e-3) / 1e12, winstr * 64 * valu / (valu + other) / (total_ms * 1e-3) / 1e12,
           (total_ms * 1e-3) * ghz * 1e9 * 1024 / winstr);
}

int main(int argc, char **argv)
{
    if (argc >= 5 && !strcmp(argv[1], "sustain")) {
        const int op = atoi(argv[2]), wps = atoi(argv[3]);
        const double sec = atof(argv[4]);
        switch (op) {
        case 0: sustain<0>("16 fma, 8 chains", 16, 0, wps, sec); break;
        case 2: sustain<2>("16 fma + 16 s_add", 16, 16, wps, sec); break;
        case 7: sustain<7>("stage-like VALU", 44, 0, wps, sec); break;
        case 8: sustain<8>("stage-like VALU + 24 SALU", 44, 24, wps, sec); break;
        case 11: sustain<11>("16 s_add only", 0, 16, wps, sec); break;
        case 25: sustain<25>("16 ds_read_b128 + 16 fma", 16, 16, wps, sec); break;
        case 26: sustain<26>("16 v_mov_dpp", 16, 0, wps, sec); break;
        case 27: sustain<27>("8 sqrt + 8 rcp", 16, 0, wps, sec); break;
        case 28: sustain<28>("16 v_pk_fma_f32", 16, 0, wps, sec); break;
        case 29: sustain<29>("16 v_mul_f32", 16, 0, wps, sec); break;
        case 30: sustain<30>("16 v_add_f32", 16, 0, wps, sec); break;
#define SUS(n, nm) case n: sustain<n>(nm, 16, 0, wps, sec); break;
        SUS(40, "v_sub_u32") SUS(41, "v_sub_u32_sdwa bytes") SUS(42, "v_mad_i32_i24") SUS(43, "v_mul_i32_i24") SUS(44, "v_add_u32") SUS(45, "v_lshl_or_b32")
        SUS(46, "v_max_u32") SUS(47, "v_max3_u32") SUS(48, "v_alignbyte_b32") SUS(49, "v_cndmask_b32_e64 sgpr mask") SUS(50, "v_max_u32_dpp row_shr")
        SUS(51, "v_mad_u32_u24") SUS(52, "v_and_b32") SUS(53, "v_bfe_u32") SUS(54, "v_perm_b32") SUS(55, "v_dot4_u32_u8") SUS(56, "v_sad_u8")
        SUS(57, "v_pk_mul_lo_u16") SUS(58, "v_pk_sub_i16") SUS(59, "v_mad_u32_u16")
        SUS(60, "v_cvt_f32_ubyte1") SUS(61, "v_cvt_u32_f32") SUS(62, "v_lshrrev_b32") SUS(63, "v_lshlrev_b32") SUS(64, "v_or_b32") SUS(65, "v_sub_f32")
        SUS(66, "v_fma_f32 neg") SUS(67, "v_cvt_f32_u32") SUS(68, "v_sub_f32_sdwa dword") SUS(69, "v_min_u32") SUS(70, "v_cvt_f32_ubyte0_sdwa") SUS(71, "v_mul_u32_u24")
        SUS(72, "v_mul_lo_u32") SUS(73, "v_max_f32") SUS(74, "v_xor_b32") SUS(75, "v_subrev_u32")
        default: printf("no such op\n");
        }
        return 0;
    }
    run<0>("16 fma, 8 chains", 16, 0);
    run<6>("16 fma, ONE dependent chain", 16, 0);
    run<1>("16 fma + 8 s_add", 16, 8);
    run<2>("16 fma + 16 s_add", 16, 16);
    run<3>("16 fma + 16 s_add + 16 cmp/cselect", 16, 48);
    run<4>("16 fma + 16 s_nop", 16, 16);
    run<5>("16 fma + 16 s_waitcnt", 16, 16);
    run<11>("16 s_add only", 0, 16);
    run<7>("stage-like VALU", 44, 0);
    run<8>("stage-like VALU + 24 SALU", 44, 24);
    run<9>("stage VALU + 1 ds_read + 2 ds_write", 44, 4);
    run<10>("same, exec-masked one-lane writes", 44, 8);
    run<12>("fma: two chains alternating", 16, 0);
    run<13>("fma: three chains", 16, 0);
    run<14>("fma: four chains", 16, 0);
    run<15>("sqrt -> dependent fma, 0 between", 16, 0);
    run<16>("sqrt -> dependent fma, 1 between", 16, 0);
    run<17>("sqrt -> dependent fma, 2 between", 16, 0);
    run<18>("sqrt -> dependent fma, 3 between", 16, 0);
    run<19>("tail sqrt,fma,rcp,mul serial", 16, 0);
    run<20>("tail, components interleaved", 16, 0);
    run<21>("tail, interleaved + 2 fillers", 16, 0);
    run<22>("fma -> s_nop 1 + dpp(dep) -> fma", 16, 8);
    run<23>("dpp of a 2-old value", 16, 0);
    run<24>("8 x (s_cmp, s_cselect, dep fma)", 8, 16);
    return 0;
}
