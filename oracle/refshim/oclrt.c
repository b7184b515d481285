/*
 * oracle/refshim/oclrt.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See oclrt.h.
 * Plain C (gcc, OpenMP); the built-ins with OpenCL vector types in their signatures live in oclrt_vec.c.
 */
#define _GNU_SOURCE
#include "oclrt.h"
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

typedef struct wi_state {
    size_t gid[3], lid[3], grp[3], lsz[3], gsz[3], ngrp[3];
} wi_state;

static __thread wi_state g_wi;

/* ---- work-item functions (OpenCL C 6.12.1), mangled as clang's OpenCL front end references them */
#define WI_FN(name, mangled, field, dflt)                                   \
    size_t oclrt_##name(unsigned d) __asm__(mangled);                       \
    size_t oclrt_##name(unsigned d) { return d < 3 ? g_wi.field[d] : dflt; }
WI_FN(get_global_id, "_Z13get_global_idj", gid, 0)
WI_FN(get_local_id, "_Z12get_local_idj", lid, 0)
WI_FN(get_group_id, "_Z12get_group_idj", grp, 0)
WI_FN(get_local_size, "_Z14get_local_sizej", lsz, 1)
WI_FN(get_global_size, "_Z15get_global_sizej", gsz, 1)
WI_FN(get_num_groups, "_Z14get_num_groupsj", ngrp, 1)

/* ---- cooperative work-group execution */
#define FIBER_STACK (256 * 1024)
typedef struct fiber {
    ucontext_t ctx;
    wi_state wi;
    int done;
} fiber;
static ucontext_t g_sched;
static fiber *g_fibers;
static char *g_stacks;
static size_t g_nfibers_cap;
static fiber *g_cur;
static oclrt_body g_body;
static void *g_args;

static void fiber_main(void)
{
    g_body(g_args);
    g_cur->done = 1;
    swapcontext(&g_cur->ctx, &g_sched);
}

void oclrt_barrier(unsigned flags) __asm__("_Z7barrierj");
void oclrt_barrier(unsigned flags)
{
    (void)flags;
    if (!g_cur) { fprintf(stderr, "oclrt: barrier() in a kernel launched without uses_barrier\n"); abort(); }
    g_cur->wi = g_wi;
    swapcontext(&g_cur->ctx, &g_sched);
    g_wi = g_cur->wi;
}

static void run_group_fibers(size_t nitems, const wi_state *items)
{
    if (nitems > g_nfibers_cap) {
        free(g_fibers); free(g_stacks);
        g_fibers = (fiber *)calloc(nitems, sizeof(fiber));
        g_stacks = (char *)malloc(nitems * (size_t)FIBER_STACK);
        g_nfibers_cap = nitems;
    }
    for (size_t i = 0; i < nitems; ++i) {
        fiber *f = &g_fibers[i];
        getcontext(&f->ctx);
        f->ctx.uc_stack.ss_sp = g_stacks + i * (size_t)FIBER_STACK;
        f->ctx.uc_stack.ss_size = FIBER_STACK;
        f->ctx.uc_link = NULL;
        f->wi = items[i];
        f->done = 0;
        makecontext(&f->ctx, fiber_main, 0);
    }
    size_t live = nitems;
    while (live) {   /* every live item runs to its next barrier (or its end) once per round */
        for (size_t i = 0; i < nitems; ++i) {
            fiber *f = &g_fibers[i];
            if (f->done) continue;
            g_cur = f;
            g_wi = f->wi;
            swapcontext(&g_sched, &f->ctx);
            if (f->done) --live;
        }
    }
    g_cur = NULL;
}

void oclrt_run(int dim, const size_t *gsz_, const size_t *lsz_, int uses_barrier, int parallel, oclrt_body body, void *args)
{
    size_t gsz[3] = {1, 1, 1}, lsz[3] = {1, 1, 1}, ngrp[3];
    for (int d = 0; d < dim; ++d) { gsz[d] = gsz_[d]; lsz[d] = lsz_ ? lsz_[d] : 1; }
    for (int d = 0; d < 3; ++d) {
        if (gsz[d] % lsz[d]) { fprintf(stderr, "oclrt: global size not a multiple of the local size\n"); abort(); }
        ngrp[d] = gsz[d] / lsz[d];
    }
    if (!uses_barrier) {
        const long n2 = (long)gsz[2], n1 = (long)gsz[1], n0 = (long)gsz[0];
#pragma omp parallel for collapse(2) schedule(static) if (parallel)
        for (long z = 0; z < n2; ++z)
            for (long y = 0; y < n1; ++y) {
                wi_state w;
                memcpy(w.lsz, lsz, sizeof(lsz)); memcpy(w.gsz, gsz, sizeof(gsz)); memcpy(w.ngrp, ngrp, sizeof(ngrp));
                w.gid[2] = (size_t)z; w.gid[1] = (size_t)y;
                w.grp[2] = w.gid[2] / lsz[2]; w.lid[2] = w.gid[2] % lsz[2];
                w.grp[1] = w.gid[1] / lsz[1]; w.lid[1] = w.gid[1] % lsz[1];
                for (long x = 0; x < n0; ++x) {
                    w.gid[0] = (size_t)x; w.grp[0] = w.gid[0] / lsz[0]; w.lid[0] = w.gid[0] % lsz[0];
                    g_wi = w;
                    body(args);
                }
            }
        return;
    }
    const size_t nitems = lsz[0] * lsz[1] * lsz[2];
    wi_state *items = (wi_state *)malloc(nitems * sizeof(wi_state));
    g_body = body; g_args = args;
    for (size_t gz = 0; gz < ngrp[2]; ++gz)
        for (size_t gy = 0; gy < ngrp[1]; ++gy)
            for (size_t gx = 0; gx < ngrp[0]; ++gx) {
                size_t k = 0;
                for (size_t lz = 0; lz < lsz[2]; ++lz)
                    for (size_t ly = 0; ly < lsz[1]; ++ly)
                        for (size_t lx = 0; lx < lsz[0]; ++lx) {
                            wi_state *w = &items[k++];
                            memcpy(w->lsz, lsz, sizeof(lsz)); memcpy(w->gsz, gsz, sizeof(gsz)); memcpy(w->ngrp, ngrp, sizeof(ngrp));
                            w->grp[0] = gx; w->grp[1] = gy; w->grp[2] = gz;
                            w->lid[0] = lx; w->lid[1] = ly; w->lid[2] = lz;
                            w->gid[0] = gx * lsz[0] + lx; w->gid[1] = gy * lsz[1] + ly; w->gid[2] = gz * lsz[2] + lz;
                        }
                run_group_fibers(nitems, items);
            }
    free(items);
}

/* ---- built-ins used by optical_flow_tvl1.cl and surf.cl.  Math: the host libm's float functions
 * (OpenCL leaves their last bits to the implementation; none of them is reference arithmetic). */

#define F1(name, mangled, expr) float oclrt_##name(float x) __asm__(mangled); float oclrt_##name(float x) { return expr; }
F1(ceil, "_Z4ceilf", ceilf(x))
F1(floor, "_Z5floorf", floorf(x))
F1(fabs, "_Z4fabsf", fabsf(x))
F1(sqrt, "_Z4sqrtf", sqrtf(x))
F1(round, "_Z5roundf", roundf(x))
float oclrt_hypot(float x, float y) __asm__("_Z5hypotff");
float oclrt_hypot(float x, float y) { return hypotf(x, y); }
float oclrt_atan2(float y, float x) __asm__("_Z5atan2ff");
float oclrt_atan2(float y, float x) { return atan2f(y, x); }
float oclrt_copysign(float x, float y) __asm__("_Z8copysignff");
float oclrt_copysign(float x, float y) { return copysignf(x, y); }
float oclrt_sincos(float x, float *c) __asm__("_Z6sincosfPU9CLprivatef");
float oclrt_sincos(float x, float *c) { *c = cosf(x); return sinf(x); }
int oclrt_clamp(int x, int lo, int hi) __asm__("_Z5clampiii");
int oclrt_clamp(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
int oclrt_abs(int x) __asm__("_Z3absi");
int oclrt_abs(int x) { return x < 0 ? -x : x; }   /* OpenCL abs returns the unsigned type; same bits */
int oclrt_max(int a, int b) __asm__("_Z3maxii");
int oclrt_max(int a, int b) { return a > b ? a : b; }
int oclrt_min(int a, int b) __asm__("_Z3minii");
int oclrt_min(int a, int b) { return a < b ? a : b; }
int oclrt_mad24(int a, int b, int c) __asm__("_Z5mad24iii");
int oclrt_mad24(int a, int b, int c) { return a * b + c; }
int oclrt_atomic_inc(volatile int *p) __asm__("_Z10atomic_incPU8CLglobalVi");
int oclrt_atomic_inc(volatile int *p) { return __atomic_fetch_add(p, 1, __ATOMIC_RELAXED); }

/* convert_int_rte / _rtn / _rtp: float -> int, round to nearest even / toward -inf / toward +inf */
int oclrt_convert_int_rte(float x) __asm__("_Z15convert_int_rtef");
int oclrt_convert_int_rte(float x) { return (int)rintf(x); }
int oclrt_convert_int_rtn(float x) __asm__("_Z15convert_int_rtnf");
int oclrt_convert_int_rtn(float x) { return (int)floorf(x); }
int oclrt_convert_int_rtp(float x) __asm__("_Z15convert_int_rtpf");
int oclrt_convert_int_rtp(float x) { return (int)ceilf(x); }

/* plain-C names of the same state for the CUDA-on-host shim (cudashim.h) */
size_t oclrt_local_id(unsigned d) { return d < 3 ? g_wi.lid[d] : 0; }
size_t oclrt_group_id(unsigned d) { return d < 3 ? g_wi.grp[d] : 0; }
size_t oclrt_local_size(unsigned d) { return d < 3 ? g_wi.lsz[d] : 1; }
size_t oclrt_num_groups(unsigned d) { return d < 3 ? g_wi.ngrp[d] : 1; }
void oclrt_barrier_plain(void) { oclrt_barrier(0); }
