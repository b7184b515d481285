/*
 * oracle/refshim/cudashim/cudashim.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Just enough of the CUDA execution model to RUN the reference's own .cu kernel sources on the CPU: thread blocks as
 * cooperatively scheduled contexts (the fibers of oclrt.c: __syncthreads() = yield until every live thread of the block has
 * arrived), __shared__ as static storage (blocks run one after the other), threadIdx / blockIdx / blockDim / gridDim, the dynamic
 * shared-memory array, and the few runtime calls the launch wrappers make.  The .cu file itself is compiled from where it lies
 * under /root/reference after ONE mechanical rewrite done at build time by cu2host.py into oracle/_ref/ (never committed):
 * `k<<<g, b, s, st>>>(args);` -> `cudashim::launch(g, b, s, [&] { k(args); });` and `extern __shared__ T a[];` ->
 * `T *a = (T *)cudashim::dynamic_smem();` -- no arithmetic is touched.  Nothing here restates reference arithmetic.
 */
#ifndef ORACLE_CUDASHIM_H
#define ORACLE_CUDASHIM_H
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

extern "C" {
#include "../oclrt.h"
size_t oclrt_local_id(unsigned d);
size_t oclrt_group_id(unsigned d);
size_t oclrt_local_size(unsigned d);
size_t oclrt_num_groups(unsigned d);
void oclrt_barrier_plain(void);
}

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __constant__ static
#define __restrict__
#define __launch_bounds__(...)

struct uint3 { unsigned x, y, z; };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint2 { unsigned x, y; };
struct int2 { int x, y; };
struct float2 { float x, y; };
struct int4 { int x, y, z, w; };
struct float4 { float x, y, z, w; };
typedef unsigned int uint;
typedef unsigned char uchar;
typedef unsigned short ushort;
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r = {x, y}; return r; }
static inline int2 make_int2(int x, int y) { int2 r = {x, y}; return r; }
static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
static inline int4 make_int4(int x, int y, int z, int w) { int4 r = {x, y, z, w}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }

namespace cudashim {
struct Idx { unsigned x, y, z; };
inline Idx tid() { Idx i = {(unsigned)oclrt_local_id(0), (unsigned)oclrt_local_id(1), (unsigned)oclrt_local_id(2)}; return i; }
inline Idx bid() { Idx i = {(unsigned)oclrt_group_id(0), (unsigned)oclrt_group_id(1), (unsigned)oclrt_group_id(2)}; return i; }
inline Idx bdim() { Idx i = {(unsigned)oclrt_local_size(0), (unsigned)oclrt_local_size(1), (unsigned)oclrt_local_size(2)}; return i; }
inline Idx gdim() { Idx i = {(unsigned)oclrt_num_groups(0), (unsigned)oclrt_num_groups(1), (unsigned)oclrt_num_groups(2)}; return i; }
void *dynamic_smem();
void set_dynamic_smem(size_t bytes);
template <typename F> void body_tramp(void *p) { (*static_cast<F *>(p))(); }
template <typename F> void launch(dim3 grid, dim3 block, size_t smem, F f)
{
    set_dynamic_smem(smem);
    const size_t g[3] = {(size_t)grid.x * block.x, (size_t)grid.y * block.y, (size_t)grid.z * block.z};
    const size_t l[3] = {block.x, block.y, block.z};
    oclrt_run(3, g, l, 1, 0, &body_tramp<F>, &f);
}
template <typename F> void launch(dim3 grid, dim3 block, size_t smem, void *, F f) { launch(grid, block, smem, f); }
template <typename F> void launch(dim3 grid, dim3 block, F f) { launch(grid, block, 0, f); }   // k<<<grid, block>>>(...)
}  // namespace cudashim

#define threadIdx (cudashim::tid())
#define blockIdx (cudashim::bid())
#define blockDim (cudashim::bdim())
#define gridDim (cudashim::gdim())
static inline void __syncthreads() { oclrt_barrier_plain(); }

// device math used by the kernels in global scope (CUDA puts min / max / abs overloads there)
static inline int min(int a, int b) { return a < b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline int __mul24(int a, int b) { return a * b; }
// float -> int conversions with an explicit rounding mode (PTX cvt.r{n,m,p}i.s32.f32)
static inline int __float2int_rn(float v) { return (int)nearbyintf(v); }
static inline int __float2int_rd(float v) { return (int)floorf(v); }
static inline int __float2int_ru(float v) { return (int)ceilf(v); }
// atomicInc(p, limit): old = *p; *p = old >= limit ? 0 : old + 1 (threads of a block are fibers of one host thread: no race)
static inline unsigned atomicInc(unsigned *p, unsigned limit) { const unsigned old = *p; *p = old >= limit ? 0u : old + 1u; return old; }
#ifndef CV_PI_F
#define CV_PI_F 3.14159265f   /* main repo core/cuda/common.hpp */
#endif
static inline float __fdividef(float a, float b) { return a / b; }

// runtime calls of the launch wrappers
typedef void *cudaStream_t;
typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaTextureFilterMode { cudaFilterModePoint = 0, cudaFilterModeLinear = 1 };
enum cudaTextureAddressMode { cudaAddressModeWrap = 0, cudaAddressModeClamp = 1 };
enum cudaTextureReadMode { cudaReadModeElementType = 0, cudaReadModeNormalizedFloat = 1 };
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline cudaError_t cudaMemset2DAsync(void *p, size_t pitch, int v, size_t width, size_t height, cudaStream_t)
{
    for (size_t y = 0; y < height; ++y) memset((char *)p + y * pitch, v, width);
    return cudaSuccess;
}
template <typename T> static inline cudaError_t cudaMemcpyToSymbol(T &symbol, const void *src, size_t count)
{
    memcpy((void *)&symbol, src, count);   // __constant__ objects are plain statics here
    return cudaSuccess;
}
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3 };
static inline cudaError_t cudaMemcpy(void *dst, const void *src, size_t count, cudaMemcpyKind) { memcpy(dst, src, count); return cudaSuccess; }
#define cudaSafeCall(expr) ((void)(expr))
#define CV_Error(code, msg) throw std::runtime_error(msg)
namespace cv { namespace Error { enum { StsBadArg = -5 }; } }
#endif
