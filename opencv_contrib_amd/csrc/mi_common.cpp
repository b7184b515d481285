// Error channel, device selection and device-memory helpers of the C-ABI.
#include "mi_common.h"
#include <vector>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <mutex>

namespace mi {
static Tuning g_tuning;
static std::once_flag g_tuning_once;
static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
// Release builds read a handful of switches only (each one covered by a digest test); every other tuning variable --
// all alternatives that lost their A/B, and the two that skip work (MIFLOW_X_SKIP) or change results (MIFLOW_TB_P16) -- exists in
// the experiments build alone (MIFLOW_BUILD_VARIANT=exp MIFLOW_EXTRA_FLAGS=-DMIFLOW_EXPERIMENTS python -m opencv_contrib_amd.build):
// in the shipped library the names are not even present as strings and the fields hold their defaults.
#ifdef MIFLOW_EXPERIMENTS
#define EXP_INT(name, dflt) env_int(name, dflt)
#define EXP_ENV(name) getenv(name)
#else
#define EXP_INT(name, dflt) (dflt)
#define EXP_ENV(name) ((const char *)nullptr)
#endif
#ifndef MIFLOW_EXPERIMENTS
// A deployment that sets a variable only the experiments build reads would be ignored silently: say so, once, on stderr.
// (The names this library does read are listed; anything else that starts with the prefix is reported.)
extern "C" char **environ;
static void warn_unread_switches()
{
    static const char *const known[] = {"MIFLOW_BF_W", "MIFLOW_CACHE_GB", "MIFLOW_CACHE_TOTAL_GB", "MIFLOW_FB_FUSE", "MIFLOW_FB_GROUP_MB", "MIFLOW_FB_NARROW",
                                        "MIFLOW_FB_PAIR", "MIFLOW_LANES", "MIFLOW_MULTI_RCCL", "MIFLOW_SURF_NMS0", "MIFLOW_SURF_POLY", "MIFLOW_SURF_STAGE_S",
                                        "MIFLOW_TB_HIST", "MIFLOW_TB_VERBOSE", "MIFLOW_TILE_MAXPX"};
    // read by the Python side, the bench, the tests or the build -- not by the library
    static const char *const foreign[] = {"MIFLOW_LIB", "MIFLOW_BENCH_", "MIFLOW_SWEEP_", "MIFLOW_BUILD_", "MIFLOW_EXTRA_", "MIFLOW_SLP_"};
    if (!environ) return;
    for (char **e = environ; *e; ++e) {
        if (strncmp(*e, "MIFLOW_", 7) != 0) continue;
        const char *eq = strchr(*e, '=');
        const size_t n = eq ? (size_t)(eq - *e) : strlen(*e);
        bool ok = false;
        for (const char *k : known) ok = ok || (strlen(k) == n && strncmp(k, *e, n) == 0);
        for (const char *k : foreign) ok = ok || strncmp(k, *e, strlen(k)) == 0;
        if (!ok) fprintf(stderr, "miflow: %.*s is not read by this library (tuning experiments exist in the -DMIFLOW_EXPERIMENTS build only); ignored\n", (int)n, *e);
    }
}
#endif
const Tuning &tuning()
{
    std::call_once(g_tuning_once, [] {
        Tuning &t = g_tuning;
#ifndef MIFLOW_EXPERIMENTS
        warn_unread_switches();
#endif
        const char *w = EXP_ENV("MIFLOW_WARP");
        t.warp_legacy = (w && w[0] == 'p') ? 1 : 0;
        t.x_skip = EXP_INT("MIFLOW_X_SKIP", 0);
        t.warp_tile = EXP_INT("MIFLOW_WARP_TILE", 32);
        if (t.warp_tile != 64 && t.warp_tile != 32 && t.warp_tile != 16) t.warp_tile = 32;
        t.warp_lds = EXP_INT("MIFLOW_WARP_LDS", 0);   // r02z3 at 1080p x 16: LDS-staged windows 1 140 vs 1 180 pairs/s with two lanes, 1 037 vs 1 012 with one
        t.warp_fast = EXP_INT("MIFLOW_WARP_FAST", -1);   // -1: automatic = cv::cuda semantics only (window_sums, tvl1_warp_kernels.hip)
        t.warp_np = EXP_INT("MIFLOW_WARP_NP", 2);   // r02e at 1080p x 16: np 1 | 2 | 4 = 904 | 1042 | 1034 pairs/s
        if (t.warp_np != 1 && t.warp_np != 4) t.warp_np = 2;
        t.tb_swz = EXP_INT("MIFLOW_TB_SWZ", 1);
        // joined-wave form of the T = 10 blocked iteration kernel (tvl1_tbr_kernels.hip): 2 (default since r03w) = hand-over with one
        // workgroup barrier per stage, 1 = with tags and bounded waits, 0 = independent 64-column waves; all three bit-identical
        t.warp_zoom = EXP_INT("MIFLOW_WARP_ZOOM", 0);   // r08k: bit-identical, 1 388 against 1 406 pairs/s: the per-pixel (double precision) coordinates of cv::resize cost more than the resize launch and its 8 B/px
        t.tb_p16 = EXP_INT("MIFLOW_TB_P16", 0);
        t.tb_nograd = EXP_INT("MIFLOW_TB_NOGRAD", 1);
        t.tb_skip_p = EXP_INT("MIFLOW_TB_SKIP_P", 1);
        t.tb_hist = env_int("MIFLOW_TB_HIST", 1);
        t.fb_poll = EXP_INT("MIFLOW_FB_POLL", 1);
        t.fb_ahead = EXP_INT("MIFLOW_FB_AHEAD", 1);
        t.tb_fw = EXP_INT("MIFLOW_TB_FW", 0);   // bit-identical; measured slower (r14b: 1 190 against 1 372 pairs/s at 64 pairs): experiments build only since round 6
        t.tb_jw = EXP_INT("MIFLOW_TB_JW", 2);   // the release library contains the barrier form (2) only
        if (t.tb_jw < 0 || t.tb_jw > 4) t.tb_jw = 2;   // 3: eight joined waves (experiment); 4: barrier form, branch-free publishes, mask-free interior blocks
        // the speculative steps (MODE 1, class defaults) as joined waves too (barrier form only): r04a at 1080p x 32, 300 iterations,
        // epsilon 0.01: 533 -> 593 pairs/s, the same flows
        t.tb_jw_spec = EXP_INT("MIFLOW_TB_JW_SPEC", 1);
        t.tb_ppl = t.tb_wps = t.tb_pf = -1;
        if (const char *v = EXP_ENV("MIFLOW_TB_VARIANT")) (void)sscanf(v, "%d,%d,%d", &t.tb_ppl, &t.tb_wps, &t.tb_pf);
        t.tb_force = EXP_ENV("MIFLOW_TB_FORCE") != nullptr;
        t.tb_plan_wps = EXP_INT("MIFLOW_TB_WPS", 0);
        t.tb_rows = EXP_INT("MIFLOW_TB_ROWS", 0);
        t.tb_verbose = getenv("MIFLOW_TB_VERBOSE") != nullptr;
        {
            const char *e = getenv("MIFLOW_TILE_MAXPX");
            t.tile_maxpx = e && *e ? atoll(e) : 2300000;   // the three coarsest levels of a 1080p pyramid at 8..16 pairs per lane
        }
        t.tile_variant = EXP_INT("MIFLOW_TILE_VARIANT", -1);
        t.tile_small_wgs = EXP_INT("MIFLOW_TILE_SMALL_WGS", 1024);
        t.tile_swz = EXP_INT("MIFLOW_TILE_SWZ", 1);
        t.warp_swz = EXP_INT("MIFLOW_WARP_SWZ", 1);
        t.tile_fb_block = EXP_INT("MIFLOW_TILE_FB_BLOCK", 10);
        t.tile_fb_model = EXP_INT("MIFLOW_TILE_FB_MODEL", 7);
        t.tile_spec = EXP_INT("MIFLOW_TILE_SPEC", 1);
        t.lanes = env_int("MIFLOW_LANES", 0);
        t.spec = EXP_INT("MIFLOW_SPEC", 1);
        t.exact_tb = EXP_INT("MIFLOW_EXACT_TB", 1);
        t.fb_tiled = EXP_INT("MIFLOW_FB_TILED", 1);
        t.sbm_wt = EXP_INT("MIFLOW_SBM_WT", 1);
        t.sbm_swz = EXP_INT("MIFLOW_SBM_SWZ", 1);
        t.sbm_texfuse = EXP_INT("MIFLOW_SBM_TEXFUSE", 1);
        t.fb_rows = EXP_INT("MIFLOW_FB_ROWS", 4) == 8 ? 8 : 4;
        t.fb_group_mb = env_int("MIFLOW_FB_GROUP_MB", 240);   // 640 x 480 x 32 pairs, two chains (r16i / r16j): 160 | 200 | 240 | 270 | 300 | 330 | 440 MB = 9 630 | 9 630-9 740 | 10 030-10 170 | 9 370 | 9 300 | 8 890 | 8 560 pairs/s; one chain (r15k): 0 | 200 | 320 = 7 190 | 7 755 | 7 185
        t.fb_fuse = env_int("MIFLOW_FB_FUSE", -1);
        t.fb_pair = env_int("MIFLOW_FB_PAIR", -1);
        t.fb_narrow = env_int("MIFLOW_FB_NARROW", -1);
        t.fb_swz = EXP_INT("MIFLOW_FB_SWZ", 1);
        t.fb_group_streams = EXP_INT("MIFLOW_FB_GROUP_STREAMS", 2);   // pair groups of a batched level as two chains on two streams (1: one chain)
        t.fb_poly_tiled = EXP_INT("MIFLOW_FB_POLY_TILED", 1);   // polynomial expansion on 8-row tiles (0: one row per workgroup)
        t.fb_direct = EXP_INT("MIFLOW_FB_DIRECT", 1);   // pyramid pre-blur reads the caller's matrices (0: through converted f32 planes)
        t.fb_blur_tiled = EXP_INT("MIFLOW_FB_BLUR_TILED", 1);   // r16e: pyramid pre-blur on 8-14 row tiles from the LDS (0: one row per workgroup)
    });
    return g_tuning;
}

namespace {
struct BigBlock { void *p; size_t cap; int dev; hipEvent_t ready; };   // ready: the previous owner's last work on the block (may be null)
std::mutex g_big_mu;
std::vector<BigBlock> g_big;
const size_t kBigMaxBlocks = 4;   // PER DEVICE (two lanes of two handles): eight GPUs of a node each keep their own
// upper bounds of what the cache may hold: MIFLOW_CACHE_GB per DEVICE (0 = cache nothing, for hosts whose own allocator wants the
// memory back) and MIFLOW_CACHE_TOTAL_GB for the whole process (all devices together; default 4 x the per-device bound)
size_t big_max_bytes()
{
    static const size_t v = [] {
        const char *e = getenv("MIFLOW_CACHE_GB");
        const long long gb = e && *e ? atoll(e) : 24;
        return (size_t)(gb < 0 ? 0 : gb) << 30;
    }();
    return v;
}
size_t big_max_bytes_total()
{
    static const size_t v = [] {
        const char *e = getenv("MIFLOW_CACHE_TOTAL_GB");
        if (e && *e) { const long long gb = atoll(e); return (size_t)(gb < 0 ? 0 : gb) << 30; }
        return 4 * big_max_bytes();
    }();
    return v;
}
}  // namespace

int big_alloc(void **p, size_t bytes, size_t *capacity)
{
    int dev = 0;
    MI_HIP_TRY(hipGetDevice(&dev));
    {
        hipEvent_t ready = nullptr;
        bool hit = false;
        {
            std::lock_guard<std::mutex> lk(g_big_mu);
            int best = -1;
            for (size_t i = 0; i < g_big.size(); ++i)
                if (g_big[i].dev == dev && g_big[i].cap >= bytes && g_big[i].cap <= bytes + bytes / 2 + (64u << 20) &&
                    (best < 0 || g_big[i].cap < g_big[best].cap))
                    best = (int)i;
            if (best >= 0) {
                *p = g_big[best].p; *capacity = g_big[best].cap; ready = g_big[best].ready;
                g_big.erase(g_big.begin() + best);
                hit = true;
            }
        }
        if (hit) {
            // nothing the block's previous owner enqueued may still touch it: wait for ITS last work only (an event recorded when the
            // block was returned) -- no device-wide synchronisation, other handles' streams keep running
            if (ready) {
                const hipError_t we = hipEventSynchronize(ready);
                (void)hipEventDestroy(ready);
                if (we != hipSuccess) {
                    // The event could not be waited for.  Seen on ROCm 7.2 when the stream it was recorded on has since been
                    // destroyed (hipErrorCapturedEvent: the runtime looks at the dead stream).  Clear the thread's sticky error -- it
                    // would surface at the next hipGetLastError() after an unrelated launch -- and fall back to what vouches for the
                    // block without the event: the whole device idle.  If even that fails the block is dropped for a fresh one.
                    (void)hipGetLastError();
                    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); (void)hipFree(*p); *p = nullptr; hit = false; }
                }
            }
            if (hit) return MI_OK;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {   // make room and retry once
        big_trim();
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) { set_error("hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e)); return MI_ERR_OOM; }
    *capacity = bytes;
    return MI_OK;
}

// `ready`: an event the owner recorded behind its last work on the block WHILE ITS STREAM WAS ALIVE (ownership passes to the cache; the
// next taker waits for it and destroys it).  No event and `busy`: the owner cannot vouch for the block (e.g. its last calc was
// enqueued under stream capture) -- the device is synchronised instead, which keeps the old guarantee.  Nothing here touches a
// caller's stream: at destroy time it may no longer exist.
void big_free(void *p, size_t capacity, hipEvent_t ready, bool busy)
{
    if (!p) { if (ready) (void)hipEventDestroy(ready); return; }
    int dev = 0;
    void *drop = nullptr;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipPointerAttribute_t at;
        const int cur = dev;
        if (hipPointerGetAttributes(&at, p) == hipSuccess) dev = at.device;
        if (dev != cur) (void)hipSetDevice(dev);
        bool idle = true;
        if (!ready && busy) idle = hipDeviceSynchronize() == hipSuccess;
        if (dev != cur) (void)hipSetDevice(cur);
        std::lock_guard<std::mutex> lk(g_big_mu);
        size_t total = capacity, all = capacity, blocks = 0;
        for (const BigBlock &b : g_big) {
            all += b.cap;
            if (b.dev == dev) { total += b.cap; ++blocks; }
        }
        if (idle && blocks < kBigMaxBlocks && total <= big_max_bytes() && all <= big_max_bytes_total()) {
            g_big.push_back({p, capacity, dev, ready});
            return;
        }
        if (ready) { (void)hipEventSynchronize(ready); (void)hipEventDestroy(ready); }
        drop = p;
    } else {
        if (ready) (void)hipEventDestroy(ready);   // the cache does not take the block: its event goes with it
        drop = p;
    }
    const hipError_t fe = hipFree(drop);
    if (fe != hipSuccess) set_error("hipFree of a %zu-byte arena failed: %s", capacity, hipGetErrorString(fe));
}

void big_trim()
{
    std::vector<BigBlock> all;
    {
        std::lock_guard<std::mutex> lk(g_big_mu);
        all.swap(g_big);
    }
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (const BigBlock &b : all) {
        if (b.ready) { (void)hipEventSynchronize(b.ready); (void)hipEventDestroy(b.ready); }
        (void)hipFree(b.p);   // hipFree takes pointers of any device
    }
    if (have) (void)hipSetDevice(cur);
}

int device_simds()
{
    static std::mutex mu;
    static int cache[64];   // 0 = not queried yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1024;
    std::lock_guard<std::mutex> lk(mu);
    if (!cache[dev]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cache[dev] = 4 * cus;
    }
    return cache[dev];
}

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mi

extern "C" {

const char *mi_last_error(void) { return mi::g_err; }
#ifdef MIFLOW_EXPERIMENTS
const char *mi_version(void) { return "miflow 0.1 (gfx950) +experiments"; }   // the tuning variants are compiled in (tests that exercise them ask)
#else
const char *mi_version(void) { return "miflow 0.1 (gfx950)"; }
#endif

int mi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int mi_set_device(int device) { MI_HIP_TRY(hipSetDevice(device)); return MI_OK; }
int mi_release_cached_memory(void) { mi::big_trim(); return MI_OK; }
int mi_get_device(int *device) { MI_REQUIRE(device, MI_ERR_BAD_ARG, "null device"); MI_HIP_TRY(hipGetDevice(device)); return MI_OK; }

int mi_malloc(void **dptr, size_t bytes)
{
    MI_REQUIRE(dptr, MI_ERR_BAD_ARG, "null dptr");
    MI_HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return MI_OK;
}
int mi_malloc_pitch(void **dptr, size_t *step, size_t width_bytes, int rows)
{
    MI_REQUIRE(dptr && step && rows >= 0, MI_ERR_BAD_ARG, "bad mi_malloc_pitch arguments");
    // GpuMat default allocator: pitched when rows > 1 && cols > 1; 256-B pitch here
    const size_t st = rows > 1 ? (width_bytes + 255) / 256 * 256 : width_bytes;
    MI_HIP_TRY(hipMalloc(dptr, st * (size_t)(rows ? rows : 1) + 16));
    *step = st;
    return MI_OK;
}
int mi_free(void *dptr) { MI_HIP_TRY(hipFree(dptr)); return MI_OK; }
int mi_memcpy_h2d(void *dst, size_t dstep, const void *src, size_t sstep, size_t width_bytes, int rows, void *stream)
{
    MI_HIP_TRY(hipMemcpy2DAsync(dst, dstep, src, sstep, width_bytes, (size_t)rows, hipMemcpyHostToDevice, (hipStream_t)stream));
    return MI_OK;
}
int mi_memcpy_d2h(void *dst, size_t dstep, const void *src, size_t sstep, size_t width_bytes, int rows, void *stream)
{
    MI_HIP_TRY(hipMemcpy2DAsync(dst, dstep, src, sstep, width_bytes, (size_t)rows, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MI_OK;
}
int mi_memset(void *dst, size_t dstep, int value, size_t width_bytes, int rows, void *stream)
{
    MI_HIP_TRY(hipMemset2DAsync(dst, dstep, value, width_bytes, (size_t)rows, (hipStream_t)stream));
    return MI_OK;
}
int mi_stream_create(void **stream)
{
    MI_REQUIRE(stream, MI_ERR_BAD_ARG, "null stream");
    hipStream_t s;
    MI_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return MI_OK;
}
int mi_stream_destroy(void *stream) { MI_HIP_TRY(hipStreamDestroy((hipStream_t)stream)); return MI_OK; }
int mi_stream_synchronize(void *stream) { MI_HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return MI_OK; }

}  // extern "C"
