// Micro-benchmark (tuning aid, not part of the product): issue cost of the VALU / LDS instructions the temporally blocked
// TV-L1 kernel is made of, on gfx950.  Prints SIMD cycles per wave-instruction at 1, 2, 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
typedef float float2v __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a[8];
    float2v p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x + i; p[i] = float2v{a[i], a[i] + 1.f}; }
    const float b = seed * 0.5f + 1.0f, c = seed * 0.25f;
    const float2v pb = {b, b}, pc = {c, c};
    extern __shared__ float lds[];
    lds[threadIdx.x] = seed;
    __syncthreads();
    const unsigned la = (threadIdx.x & 63) * 4;
    unsigned long long msk = 0x5555555555555555ull + (unsigned long long)iters, m2 = 0; unsigned sr = 0; float sb = seed * 3.f;
    for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define RSQ(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
#define DPPSUB(i) asm volatile("v_sub_f32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
#define DPPROW(i) asm volatile("v_sub_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define DSR(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[i]) : "v"(la), "n"(i * 256));
#define DSR2(i) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(p[i]) : "v"(la), "n"(i), "n"(i + 1));
#define DSR4(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(*(float4 *)&p[(i & 3) * 2]) : "v"(la * 4), "n"(i * 1024));
#define MIXTR(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_rcp_f32 %3, %3" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(b), "v"(c) : );
#define CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(msk));
#define CNDZ(i) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(a[i]) : "s"(msk));
#define CNDI(i) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b), "v"(c), "s"(msk));
#define CNDVI(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b), "v"(c));
#define FMAS(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c));
#define FMACS(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c));
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define CMPS(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(a[i]), "v"(b));
#define RDL(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sr) : "v"(a[i]));
#define MULM(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ADDF(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define SUBF(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(b));
#define FMAI(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b), "v"(c), "v"(b));
#define PKFMAI(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=v"(p[i]) : "v"(pb), "v"(pc));
#define PKFMAOPS(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pb), "v"(pc));
#define MULLEG(i) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 0) { REP8(FMA) REP8(FMA) }
        if (OP == 20) { REP8(CNDS) REP8(CNDS) }
        if (OP == 21) { REP8(CNDZ) REP8(CNDZ) }
        if (OP == 22) { REP8(CNDI) REP8(CNDI) }
        if (OP == 23) { REP8(CNDVI) REP8(CNDVI) }
        if (OP == 24) { REP8(FMAS) REP8(FMAS) }
        if (OP == 25) { REP8(FMACS) REP8(FMACS) }
        if (OP == 26) { REP8(MAXF) REP8(MAXF) }
        if (OP == 27) { REP8(CMP) REP8(CMP) }
        if (OP == 28) { REP8(CMPS) REP8(CMPS) }
        if (OP == 29) { REP8(RDL) REP8(RDL) }
        if (OP == 30) { REP8(MULM) REP8(MULM) }
        if (OP == 31) { REP8(ADDF) REP8(ADDF) }
        if (OP == 32) { REP8(MOVDPP) REP8(MOVDPP) }
        if (OP == 33) { REP8(FMAI) REP8(FMAI) }
        if (OP == 34) { REP8(PKFMAI) REP8(PKFMAI) }
        if (OP == 35) { REP8(PKFMAOPS) REP8(PKFMAOPS) }
        if (OP == 36) { REP8(SUBF) REP8(SUBF) }
        if (OP == 1) { REP8(PKFMA) REP8(PKFMA) }
        if (OP == 2) { REP8(PKMUL) REP8(PKMUL) }
        if (OP == 3) { REP8(PKADD) REP8(PKADD) }
        if (OP == 4) { REP8(RCP) REP8(RCP) }
        if (OP == 5) { REP8(SQRT) REP8(SQRT) }
        if (OP == 6) { REP8(RSQ) REP8(RSQ) }
        if (OP == 7) { REP8(DPPSUB) REP8(DPPSUB) }
        if (OP == 8) { REP8(DPPROW) REP8(DPPROW) }
        if (OP == 9) { REP8(CND) REP8(CND) }
        if (OP == 10) { REP8(MOV) REP8(MOV) }
        if (OP == 11) { REP8(MED3) REP8(MED3) }
        if (OP == 12) { REP8(DSR) REP8(DSR) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 13) { REP8(DSR2) REP8(DSR2) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 14) { REP8(MIXTR) REP8(MIXTR) }   // 64 instr: 48 fma + 16 rcp interleaved 3:1
        if (OP == 15) { REP8(DSR4) REP8(DSR4) asm volatile("s_waitcnt lgkmcnt(0)"); }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f || m2 == 77 || sr == 99) out[threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, int instr_per_iter)
{
    float *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    printf("%-28s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;   // 256 CUs x wps blocks of 4 waves = wps waves per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 65536 / 4 * 0 + 16384, 0, out, 16, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 16384, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9;   // nominal 2.4 GHz
        printf("  wps%d: %6.2f cyc/instr", wps, cyc / ((double)iters * instr_per_iter * wps));
    }
    printf("\n");
    hipFree(out);
}

int main()
{
    run<0>("v_fma_f32", 16);
    run<1>("v_pk_fma_f32", 16);
    run<2>("v_pk_mul_f32", 16);
    run<3>("v_pk_add_f32", 16);
    run<4>("v_rcp_f32", 16);
    run<5>("v_sqrt_f32", 16);
    run<6>("v_rsq_f32", 16);
    run<7>("v_sub_f32_dpp wave_shr:1", 16);
    run<8>("v_sub_f32_dpp row_shr:1", 16);
    run<9>("v_cndmask_b32", 16);
    run<10>("v_mov_b32", 16);
    run<11>("v_med3_f32", 16);
    run<12>("ds_read_b32", 16);
    run<13>("ds_read2st64_b32", 16);
    run<14>("mix 3 fma : 1 rcp", 64);
    run<15>("ds_read_b128", 16);
    run<20>("v_cndmask_e64 v,v,s[] dep", 16);
    run<21>("v_cndmask_e64 0,v,s[] dep", 16);
    run<22>("v_cndmask_e64 indep", 16);
    run<23>("v_cndmask vcc indep", 16);
    run<24>("v_fma_f32 sgpr src", 16);
    run<25>("v_fmac_f32 sgpr src", 16);
    run<26>("v_max_f32", 16);
    run<27>("v_cmp_lt_f32 vcc", 16);
    run<28>("v_cmp_lt_f32_e64 sgpr", 16);
    run<29>("v_readlane_b32", 16);
    run<30>("v_mul_f32", 16);
    run<31>("v_add_f32", 16);
    run<36>("v_sub_f32", 16);
    run<32>("v_mov_b32_dpp wave_shr", 16);
    run<33>("v_fma_f32 indep", 16);
    run<34>("v_pk_fma_f32 indep", 16);
    run<35>("v_pk_fma_f32 op_sel bcast", 16);
    return 0;
}
