// Batched-frames mode over several GPUs of one node, host side.  The worker state machine (sharding, persistent
// per-device threads, double-buffered peer-to-peer staging, error draining) lives in tvl1_multi_sm.h, written against a backend
// policy; this file is its HIP backend and the C-ABI around it.  Results are bit-identical to mi_tvl1_calc_batch.
// (per chunk of 16 1080p pairs: 2 x 133 MB in + 265 MB out against ~15 ms of compute: ~2.6 ms per direction at one xGMI link's
// ~153 GB/s.)
#include "mi_common.h"
#include "tvl1_multi_sm.h"
#include "mi_selftest.h"
#include <dlfcn.h>
#include <cstdlib>
#include <mutex>

using namespace mi;

namespace {

// ---- RCCL, bound at run time (librccl.so.1 of the ROCm installation; no link-time dependency: a host without it, or
// MIFLOW_MULTI_RCCL=0, leaves the peer copies).  Only the point-to-point subset is used: north_star asks for "RCCL over xGMI for the
// scatter/gather only" -- the pairs are independent, there is no reduction.  Prototypes as in rccl.h (ncclResult_t / ncclDataType_t
// are ints, ncclComm_t an opaque pointer, ncclChar = ncclInt8 = 0).
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) = nullptr;
    int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};
const Rccl &rccl()
{
    static const Rccl R = [] {
        Rccl r;
        const char *e = getenv("MIFLOW_MULTI_RCCL");
        if (e && *e && atoi(e) == 0) { r.why = "MIFLOW_MULTI_RCCL=0"; return r; }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!r.lib) { r.why = "librccl.so not found"; return r; }
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
        r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");
        r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
        if (!r.ok) r.why = "librccl.so lacks the point-to-point entry points";
        return r;
    }();
    return R;
}
int nccl_rc(int e, const char *what)
{
    if (e == 0) return MI_OK;
    set_error("%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error");
    return MI_ERR_HIP;
}
#define NB(expr) nccl_rc((expr), #expr)

// The transport between the root device and one worker's device: a two-rank communicator (rank 0 = root), the root-side stream the
// root's half of every send / recv pair is enqueued on, and the worker's device for restoring the thread's device.  One thread -- the
// worker's -- drives both ranks, so both halves of a chunk's transfers sit in ONE ncclGroup (the documented single-thread,
// multi-device pattern).
// (Every link shares the ROOT device: link_create and the ncclGroupStart .. ncclGroupEnd brackets are serialised process-wide by the
// state machine, tvl1_multi_sm.h root_mu -- ADVICE r05.)
// why the most recent link fell back to peer copies ("" = none did); mi_tvl1_multi_transport_why
std::mutex g_why_mu;
std::string g_link_why;
void note_fallback(const std::string &why)
{
    std::lock_guard<std::mutex> lk(g_why_mu);
    g_link_why = why;
}

struct Link {
    void *comm_root = nullptr, *comm_dev = nullptr;
    hipStream_t root_stream = nullptr;
    int root = 0, dev = 0;
    bool in_group = false;
};

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
    // RCCL link (see Link).  Dense planes (pitch == row bytes on both sides) travel as one ncclSend / ncclRecv pair; a pitched plane
    // (a ROI of a larger matrix) as a 2-D peer copy on the worker's stream, inside the same bracket.
    static int link_create(void **out, int root, int dev)
    {
        *out = nullptr;
        const Rccl &R = rccl();
        if (!R.ok) { note_fallback(R.why); return MI_OK; }   // no library (or switched off): peer copies
        if (root == dev) return MI_OK;                        // two workers on one GPU (RCCL refuses a device twice): peer copies, by design
        Link *L = new Link();
        L->root = root; L->dev = dev;
        void *comms[2] = {nullptr, nullptr};
        const int devs[2] = {root, dev};
        if (const int e = R.CommInitAll(comms, 2, devs)) {   // not fatal: the pair keeps its peer copies; the reason is kept (mi_tvl1_multi_transport_why)
            char buf[256];
            snprintf(buf, sizeof buf, "ncclCommInitAll({%d, %d}) failed: %s", root, dev, R.GetErrorString ? R.GetErrorString(e) : "RCCL error");
            note_fallback(buf);
            delete L;
            (void)hipGetLastError();
            (void)hipSetDevice(dev);
            return MI_OK;
        }
        L->comm_root = comms[0]; L->comm_dev = comms[1];
        int rc = HB(hipSetDevice(root));
        if (!rc) rc = HB(hipStreamCreateWithFlags(&L->root_stream, hipStreamNonBlocking));
        const int rc2 = HB(hipSetDevice(dev));
        if (rc || rc2) { (void)link_destroy(L); return rc ? rc : rc2; }
        *out = L;
        return MI_OK;
    }
    static int link_destroy(void *l)
    {
        Link *L = (Link *)l;
        if (!L) return MI_OK;
        const Rccl &R = rccl();
        if (L->comm_dev) (void)R.CommDestroy(L->comm_dev);
        if (L->comm_root) (void)R.CommDestroy(L->comm_root);
        if (L->root_stream) { (void)hipSetDevice(L->root); (void)hipStreamDestroy(L->root_stream); (void)hipSetDevice(L->dev); }
        delete L;
        return MI_OK;
    }
    static int link_sync(void *l)
    {
        Link *L = (Link *)l;
        int rc = HB(hipSetDevice(L->root));
        if (!rc) rc = HB(hipStreamSynchronize(L->root_stream));
        const int rc2 = HB(hipSetDevice(L->dev));
        return rc ? rc : rc2;
    }
    static int link_begin(void *l) { Link *L = (Link *)l; const int rc = NB(rccl().GroupStart()); L->in_group = rc == MI_OK; return rc; }
    static int link_end(void *l)
    {
        Link *L = (Link *)l;
        if (!L->in_group) return MI_OK;
        L->in_group = false;
        return NB(rccl().GroupEnd());
    }
    static int link_plane(void *l, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows, int to_worker, void *worker_stream)
    {
        Link *L = (Link *)l;
        if (dpitch != width_bytes || spitch != width_bytes)   // pitched: not one contiguous message
            return HB(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyDeviceToDevice, (hipStream_t)worker_stream));
        const Rccl &R = rccl();
        const size_t bytes = width_bytes * rows;
        if (to_worker) {   // rank 0 (root) sends on its stream, rank 1 receives on the worker's copy stream
            if (int rc = NB(R.Send(src, bytes, 0 /* ncclChar */, 1, L->comm_root, L->root_stream))) return rc;
            return NB(R.Recv(dst, bytes, 0, 0, L->comm_dev, (hipStream_t)worker_stream));
        }
        if (int rc = NB(R.Send(src, bytes, 0, 0, L->comm_dev, (hipStream_t)worker_stream))) return rc;
        return NB(R.Recv(dst, bytes, 0, 1, L->comm_root, L->root_stream));
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
#undef NB

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

int mi_tvl1_multi_transport(const mi_tvl1_multi *m, int *rccl_links, int *peer_copy_links)
{
    MI_REQUIRE(m, MI_ERR_BAD_ARG, "null handle");
    const int links = m->M.link_count(), workers = m->M.device_count() - 1;
    if (rccl_links) *rccl_links = links;
    if (peer_copy_links) *peer_copy_links = workers - links;
    return MI_OK;
}

const char *mi_tvl1_multi_transport_why(void)
{
    static thread_local std::string out;
    std::lock_guard<std::mutex> lk(g_why_mu);
    out = g_link_why;
    return out.c_str();
}

// Hardware self-test of the RCCL binding on ONE device (tests/): a one-rank communicator, a grouped send-to-self / receive-from-self of
// `bytes` bytes between two device buffers -- the entry points, argument orders and constants the multi-device link uses.
// Returns MI_OK with *available = 0 where the library is absent (or MIFLOW_MULTI_RCCL=0).
int miflow_selftest_rccl_self_copy(const unsigned char *in_host, unsigned char *out_host, size_t bytes, int *available)
{
    MI_REQUIRE(in_host && out_host && available && bytes > 0, MI_ERR_BAD_ARG, "null argument");
    const Rccl &R = rccl();
    *available = R.ok ? 1 : 0;
    if (!R.ok) return MI_OK;
    int dev = 0;
    MI_HIP_TRY(hipGetDevice(&dev));
    void *comm = nullptr;
    if (int rc = nccl_rc(R.CommInitAll(&comm, 1, &dev), "ncclCommInitAll")) return rc;
    DevTmp tmp;
    unsigned char *a = nullptr, *b = nullptr;
    hipStream_t st = nullptr;
    int rc = MI_OK;
    auto body = [&]() -> int {
        MI_HIP_TRY(tmp.alloc(&a, bytes));
        MI_HIP_TRY(tmp.alloc(&b, bytes));
        MI_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        MI_HIP_TRY(hipMemcpy(a, in_host, bytes, hipMemcpyHostToDevice));
        MI_HIP_TRY(hipMemset(b, 0, bytes));
        if (int r = nccl_rc(R.GroupStart(), "ncclGroupStart")) return r;
        int r1 = R.Send(a, bytes, 0, 0, comm, st), r2 = R.Recv(b, bytes, 0, 0, comm, st);
        const int r3 = R.GroupEnd();
        if (int r = nccl_rc(r1 ? r1 : r2 ? r2 : r3, "ncclSend / ncclRecv to self")) return r;
        MI_HIP_TRY(hipStreamSynchronize(st));
        MI_HIP_TRY(hipMemcpy(out_host, b, bytes, hipMemcpyDeviceToHost));
        return MI_OK;
    };
    rc = body();
    if (st) (void)hipStreamDestroy(st);
    (void)R.CommDestroy(comm);
    return rc;
}

void mi_tvl1_multi_destroy(mi_tvl1_multi *m)
{
    delete m;   // ~Machine joins the worker threads; each releases its own device's resources
}

}  // extern "C"
