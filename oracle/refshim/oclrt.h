/*
 * oracle/refshim/oclrt.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A minimal host-side OpenCL-C execution environment: just enough to RUN the reference's own OpenCL
 * kernel sources (compiled verbatim for x86-64 by clang's OpenCL C front end, see oracle/Makefile.ref)
 * on the CPU, so that the restated oracles (oracle/ *_ref.c) can be pinned against reference code that
 * executes here.  Nothing in this directory restates reference arithmetic: it provides the NDRange
 * (work-item ids, work-groups, barrier, __local via the compiler's static storage, atomic_inc) and
 * the handful of OpenCL built-ins the kernels call, by their Itanium-mangled names.
 */
#ifndef ORACLE_REFSHIM_OCLRT_H
#define ORACLE_REFSHIM_OCLRT_H
#include <stddef.h>

typedef void (*oclrt_body)(void *args);   /* calls the compiled kernel with its unpacked arguments */

/* Runs `body` once per work-item of a dim-D NDRange.  gsz = global size, lsz = local size (NULL: the
 * whole range is one row of independent items, no barrier allowed).  uses_barrier != 0: the items of a
 * work-group run as cooperatively scheduled contexts (barrier() = yield), work-groups one after the
 * other in row-major order (so atomic_inc appends are deterministic); otherwise items run as plain
 * loops, parallel over the slowest dimension when `parallel` is set (kernels without __local only). */
void oclrt_run(int dim, const size_t *gsz, const size_t *lsz, int uses_barrier, int parallel, oclrt_body body, void *args);

/* read-only image2d_t stand-in for read_imagef/read_imageui with an unnormalised, clamp-to-edge,
 * nearest sampler (the only sampler the reference kernels use) */
typedef struct oclrt_image2d {
    const void *data;
    long step;       /* bytes */
    int width, height;
    int elem;        /* 0: float, 1: uint32, 2: uint8 */
} oclrt_image2d;

#endif
