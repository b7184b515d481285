/* oracle/refshim/cudashim: stand-in for the main-repo opencv2/core/cuda/reduce.hpp (+ detail/reduce.hpp) as surf.cu uses it on
 * __CUDA_ARCH__ >= 300.  TEST INFRASTRUCTURE.
 *
 * What that header does there (restated; the header is not under /root/reference):
 *   reduce<N>(smem, val, tid, op), N <= 32 (WarpOptimized): smem unused; val = op(val, shfl_down(val, delta, N)) for
 *       delta = N/2, N/4, ..., 1 -- every lane runs it, lane 0 ends with the total;
 *   32 < N <= 1024 (GenericOptimized32, M = N / 32): the same tree of width 32 in every warp, lane 0 of warp w stores its value to
 *       smem[w], __syncthreads, every thread loads smem[tid], threads < 32 run the tree of width M (delta = M/2 ... 1): thread 0
 *       ends with the total;
 *   the tuple form reduce<N>(smem_tuple(p0, p1), thrust::tie(v0, v1), tid, thrust::make_tuple(op0, op1)) does this per component.
 * shfl_down(v, delta, width) returns lane (l + delta)'s v when that lane is inside the same width-segment, else the caller's own.
 *
 * On the fiber shim a warp = 32 consecutive linear thread ids of the block.  Every thread publishes its value, waits at a block
 * barrier, replays the tree of its own segment on the published snapshot and keeps its own lane's result; a second barrier keeps
 * the snapshot alive until every thread has read it.  All live threads of the block must call reduce together -- they do in
 * surf.cu (icvCalcOrientation, compute_descriptors_64/128, normalize_descriptors).
 */
#ifndef ORACLE_CUDASHIM_REDUCE_HPP
#define ORACLE_CUDASHIM_REDUCE_HPP
#include "opencv2/core/cuda/common.hpp"
namespace cv { namespace cuda { namespace device {
namespace reduce_shim {
enum { MAX_THREADS = 1024 };
inline unsigned linear_tid() { return threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z); }
// the shfl_down tree of `width` lanes starting at `base` of snapshot s, result of lane `lane`
template <typename T, class Op> inline T tree(const T *s, unsigned base, unsigned width, unsigned lane, Op op)
{
    T v[32];
    for (unsigned i = 0; i < width; ++i) v[i] = s[base + i];
    for (unsigned delta = width / 2; delta >= 1; delta /= 2) {
        T n[32];
        for (unsigned i = 0; i < width; ++i) n[i] = op(v[i], i + delta < width ? v[i + delta] : v[i]);
        for (unsigned i = 0; i < width; ++i) v[i] = n[i];
    }
    return v[lane];
}
template <unsigned N, typename T, class Op> inline void run(volatile T *smem, T &val, unsigned tid, Op op)
{
    static T snap[MAX_THREADS];
    const unsigned lt = linear_tid();
    snap[lt] = val;
    __syncthreads();
    if (N <= 32) {
        const unsigned base = lt - lt % N;          // the N-lane segment of this thread (N = 32: its warp)
        val = tree<T>(snap, base, N, lt % N, op);
        __syncthreads();
        return;
    }
    const unsigned M = N / 32;
    T w = tree<T>(snap, lt - lt % 32, 32, lt % 32, op);
    __syncthreads();
    if (lt % 32 == 0) smem[tid / 32] = w;
    __syncthreads();
    snap[lt] = tid < N ? (T)smem[tid] : w;          // loadFromSmem(smem, val, tid)
    __syncthreads();
    val = tid < 32 ? tree<T>(snap, lt - lt % M, M, lt % M, op) : snap[lt];   // shfl_down(., ., M): M-lane segments; thread 0 holds the total
    __syncthreads();
}
}  // namespace reduce_shim

template <unsigned N, typename T, class Op> inline void reduce(volatile T *smem, T &val, unsigned tid, const Op &op)
{
    reduce_shim::run<N, T, Op>(smem, val, tid, op);
}
template <typename P0, typename P1> struct SmemTuple2 { P0 p0; P1 p1; };
template <typename T0, typename T1> inline SmemTuple2<volatile T0 *, volatile T1 *> smem_tuple(T0 *a, T1 *b)
{
    SmemTuple2<volatile T0 *, volatile T1 *> t = {a, b};
    return t;
}
}}}
// the two thrust spellings surf.cu:614-615 uses
namespace thrust {
template <typename A, typename B> struct RefPair { A &a; B &b; };
template <typename A, typename B> inline RefPair<A, B> tie(A &a, B &b) { RefPair<A, B> r = {a, b}; return r; }
template <typename A, typename B> struct ValPair { A a; B b; };
template <typename A, typename B> inline ValPair<A, B> make_tuple(A a, B b) { ValPair<A, B> r = {a, b}; return r; }
}
namespace cv { namespace cuda { namespace device {
template <unsigned N, typename P0, typename P1, typename R0, typename R1, class Op0, class Op1>
inline void reduce(const SmemTuple2<P0, P1> &smem, const thrust::RefPair<R0, R1> &val, unsigned tid, const thrust::ValPair<Op0, Op1> &op)
{
    reduce_shim::run<N, R0, Op0>(smem.p0, val.a, tid, op.a);
    reduce_shim::run<N, R1, Op1>(smem.p1, val.b, tid, op.b);
}
}}}
#endif
