// Batched-frames mode over several GPUs of one node, host side.  The worker state machine (sharding, persistent
// per-device threads, double-buffered peer-to-peer staging, error draining) lives in tvl1_multi_sm.h, written against a backend
// policy; this file is its HIP backend and the C-ABI around it.  Results are bit-identical to mi_tvl1_calc_batch.
// (per chunk of 16 1080p pairs: 2 x 133 MB in + 265 MB out against ~15 ms of compute: ~2.6 ms per direction at one xGMI link's
// ~153 GB/s.)
#include "mi_common.h"
#include "tvl1_multi_sm.h"

using namespace mi;

namespace {

int hip_rc(hipError_t e, const char *what)
{
    if (e == hipSuccess) return MI_OK;
    set_error("%s failed: %s", what, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? MI_ERR_OOM : MI_ERR_HIP;
}
#define HB(expr) hip_rc((expr), #expr)

struct HipBackend {
    static int set_device(int dev) { return HB(hipSetDevice(dev)); }
    static int can_access_peer(int *can, int dev, int peer) { return HB(hipDeviceCanAccessPeer(can, dev, peer)); }
    static int enable_peer(int peer)
    {
        const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return MI_OK; }
        return hip_rc(e, "hipDeviceEnablePeerAccess");
    }
    static int stream_create(void **s) { return HB(hipStreamCreateWithFlags((hipStream_t *)s, hipStreamNonBlocking)); }
    static int stream_destroy(void *s) { return HB(hipStreamDestroy((hipStream_t)s)); }
    static int stream_sync(void *s) { return HB(hipStreamSynchronize((hipStream_t)s)); }
    static int stream_wait_event(void *s, void *e) { return HB(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0)); }
    static int event_create(void **e) { return HB(hipEventCreateWithFlags((hipEvent_t *)e, hipEventDisableTiming)); }
    static int event_destroy(void *e) { return HB(hipEventDestroy((hipEvent_t)e)); }
    static int event_record(void *e, void *s) { return HB(hipEventRecord((hipEvent_t)e, (hipStream_t)s)); }
    static int dev_malloc(void **p, size_t bytes) { return HB(hipMalloc(p, bytes)); }
    static int dev_free(void *p) { return HB(hipFree(p)); }
    static int copy2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows, void *s)
    {
        return HB(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyDeviceToDevice, (hipStream_t)s));
    }
    static int tvl1_create(const mi_tvl1_params *p, void **h) { return mi_tvl1_create(p, (mi_tvl1 **)h); }
    static void tvl1_destroy(void *h) { mi_tvl1_destroy((mi_tvl1 *)h); }
    static int tvl1_calc_batch(void *h, int n, const mi_mat *a, const mi_mat *b, mi_mat *f, void *s)
    {
        return mi_tvl1_calc_batch((mi_tvl1 *)h, n, a, b, f, s);
    }
    static const char *last_error() { return mi_last_error(); }
};
#undef HB

}  // namespace

struct mi_tvl1_multi {
    multi::Machine<HipBackend> M;
};

extern "C" {

int mi_tvl1_multi_create(const mi_tvl1_params *p, int n_devices, const int *device_ids, mi_tvl1_multi **out)
{
    MI_REQUIRE(out, MI_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int avail = 0;
    if (hipGetDeviceCount(&avail) != hipSuccess || avail == 0) {
        set_error("no HIP device available: the miflow product path has no CPU fallback");
        return MI_ERR_NO_DEVICE;
    }
    if (n_devices <= 0) n_devices = avail;
    MI_REQUIRE(n_devices <= 64, MI_ERR_BAD_ARG, "too many devices");
    mi_tvl1_params d;
    if (!p) { mi_tvl1_default_params(&d); p = &d; }
    std::vector<int> devs(n_devices);
    for (int i = 0; i < n_devices; ++i) {
        devs[i] = device_ids ? device_ids[i] : i;
        MI_REQUIRE(devs[i] >= 0 && devs[i] < avail, MI_ERR_BAD_ARG, "device id %d out of range (%d devices)", devs[i], avail);
    }
    mi_tvl1_multi *m = new mi_tvl1_multi();
    const int rc = m->M.init(*p, devs);   // the calling thread's current device is never touched: workers set their own
    if (rc) { set_error("%s", m->M.error().c_str()); delete m; return rc; }
    *out = m;
    return MI_OK;
}

int mi_tvl1_multi_device_count(const mi_tvl1_multi *m) { return m ? m->M.device_count() : 0; }

int mi_tvl1_multi_set_chunk(mi_tvl1_multi *m, int pairs_per_chunk)
{
    MI_REQUIRE(m && pairs_per_chunk >= 1 && pairs_per_chunk <= 4096, MI_ERR_BAD_ARG, "chunk must be in [1, 4096]");
    m->M.set_chunk(pairs_per_chunk);
    return MI_OK;
}

int mi_tvl1_multi_calc_batch(mi_tvl1_multi *m, int n, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows)
{
    MI_REQUIRE(m, MI_ERR_BAD_ARG, "null handle");
    const int rc = m->M.calc_batch(n, I0s, I1s, flows);
    if (rc) set_error("%s", m->M.error().c_str());
    return rc;
}

void mi_tvl1_multi_destroy(mi_tvl1_multi *m)
{
    delete m;   // ~Machine joins the worker threads; each releases its own device's resources
}

}  // extern "C"
