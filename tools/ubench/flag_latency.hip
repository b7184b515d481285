// How soon does the host see a word a running kernel stores into pinned host memory?  (polled host feedback of tvl1_api.cpp)
// kernel: thread 0 stores the word (system-scope release) at its START, then the grid spins for `us` microseconds.
// host: launches, polls the word, prints (time to see the word) and (time to kernel end), for coherent and default pinned memory,
// and with a second kernel queued behind the first.
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench/flag_latency.hip -o /tmp/flag_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_flag_spin(int *flag, int v, long long ticks)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}
__global__ void k_spin(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}
int main()
{
    hipStream_t st; hipStreamCreate(&st);
    for (int mode = 0; mode < 2; ++mode) {
        int *flag = nullptr;
        hipHostMalloc((void **)&flag, 64, mode == 0 ? (hipHostMallocCoherent | hipHostMallocMapped) : hipHostMallocDefault);
        *flag = 0;
        for (int behind = 0; behind < 2; ++behind)
            for (int rep = 0; rep < 6; ++rep) {
                const int v = 100 * mode + 10 * behind + rep + 1;
                hipStreamSynchronize(st);
                const double t0 = now_us();
                hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, st, 2000LL);              // 20 us ahead of it (100 MHz clock)
                hipLaunchKernelGGL(k_flag_spin, dim3(64), dim3(256), 0, st, flag, v, 5000LL);   // 50 us
                if (behind) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, st, 5000LL);    // another 50 us queued behind
                const double t1 = now_us();
                while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != v) __builtin_ia32_pause();
                const double t2 = now_us();
                hipStreamSynchronize(st);
                const double t3 = now_us();
                if (rep >= 2) printf("%s behind=%d: enqueue %.1f us, word seen %.1f us after enqueue, stream idle %.1f us after enqueue\n",
                                     mode == 0 ? "coherent" : "default ", behind, t1 - t0, t2 - t1, t3 - t1);
            }
        hipHostFree(flag);
    }
    return 0;
}
