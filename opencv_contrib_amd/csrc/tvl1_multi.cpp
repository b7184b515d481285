// Batched-frames mode over several GPUs of one node, host side: the pairs of a batch are independent, so the batch
// is cut into contiguous shards, one per device (pair i -> worker i / ceil(n / G)), with NO data-path collective.  One host
// thread per device -- the reference's own multi-device idiom is cv::cuda::setDevice per thread
// (modules/cudaoptflow/test/test_optflow.cpp:62, :468-527 for the concurrent-instances model) -- owns a mi_tvl1 handle, a compute
// stream and a copy stream on its device.  The caller's matrices live on the ROOT device (the first id): a worker on another
// device pulls its shard over xGMI with peer-to-peer 2-D copies into dense staging planes, computes, and pushes the flows back,
// in chunks and double buffered, so that the copy-in of chunk k + 1 and the copy-out of chunk k - 1 overlap the compute of
// chunk k (per chunk of 16 1080p pairs: 2 x 133 MB in + 265 MB out against ~15 ms of compute: ~2.6 ms per direction at one
// link's ~153 GB/s).  The worker on the root device computes in place.  Results are bit-identical to mi_tvl1_calc_batch.
#include "mi_common.h"
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

using namespace mi;

namespace {

struct Slot {
    void *in0 = nullptr, *in1 = nullptr, *out = nullptr;   // chunk x dense planes on the worker's device
    hipEvent_t in_done = nullptr, calc_done = nullptr, out_done = nullptr;
};

struct Worker {
    int dev = 0;
    bool is_root = false;
    mi_tvl1 *h = nullptr;
    hipStream_t compute = nullptr, copy = nullptr;
    Slot slot[2];
    int cap_chunk = 0, cap_w = 0, cap_h = 0, cap_type = -1;
    // per call
    int first = 0, count = 0, rc = MI_OK;
    std::string err;
};

size_t elem_size(int type) { return type == MI_8UC1 ? 1 : 4; }

}  // namespace

struct mi_tvl1_multi {
    mi_tvl1_params P;
    int chunk = 16;
    std::vector<Worker> W;
};

static void free_slots(Worker &w)
{
    for (Slot &s : w.slot) {
        if (s.in0) (void)hipFree(s.in0);
        if (s.in1) (void)hipFree(s.in1);
        if (s.out) (void)hipFree(s.out);
        s.in0 = s.in1 = s.out = nullptr;
    }
    w.cap_chunk = 0;
}

static int worker_init(Worker &w, const mi_tvl1_params &P, int root_dev)
{
    MI_HIP_TRY(hipSetDevice(w.dev));
    if (!w.is_root && w.dev != root_dev) {
        int can = 0;
        MI_HIP_TRY(hipDeviceCanAccessPeer(&can, w.dev, root_dev));
        MI_REQUIRE(can, MI_ERR_HIP, "device %d cannot access device %d peer-to-peer", w.dev, root_dev);
        const hipError_t e = hipDeviceEnablePeerAccess(root_dev, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) MI_HIP_TRY(e);
        (void)hipGetLastError();
    }
    int rc = mi_tvl1_create(&P, &w.h);
    if (rc) return rc;
    MI_HIP_TRY(hipStreamCreateWithFlags(&w.compute, hipStreamNonBlocking));
    MI_HIP_TRY(hipStreamCreateWithFlags(&w.copy, hipStreamNonBlocking));
    for (Slot &s : w.slot) {
        MI_HIP_TRY(hipEventCreateWithFlags(&s.in_done, hipEventDisableTiming));
        MI_HIP_TRY(hipEventCreateWithFlags(&s.calc_done, hipEventDisableTiming));
        MI_HIP_TRY(hipEventCreateWithFlags(&s.out_done, hipEventDisableTiming));
    }
    return MI_OK;
}

static int ensure_slots(Worker &w, int chunk, int W_, int H_, int type)
{
    if (w.cap_chunk >= chunk && w.cap_w == W_ && w.cap_h == H_ && w.cap_type == type) return MI_OK;
    free_slots(w);
    const size_t es = elem_size(type), in_bytes = (size_t)W_ * H_ * es * chunk, out_bytes = (size_t)W_ * H_ * 8 * chunk;
    for (Slot &s : w.slot) {
        MI_HIP_TRY(hipMalloc(&s.in0, in_bytes));
        MI_HIP_TRY(hipMalloc(&s.in1, in_bytes));
        MI_HIP_TRY(hipMalloc(&s.out, out_bytes));
    }
    w.cap_chunk = chunk; w.cap_w = W_; w.cap_h = H_; w.cap_type = type;
    return MI_OK;
}

// One worker's shard: pairs [first, first + count) of the caller's arrays.
static int worker_run(Worker &w, int chunk, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows)
{
    MI_HIP_TRY(hipSetDevice(w.dev));
    if (w.count == 0) return MI_OK;
    const int W_ = I0s[0].cols, H_ = I0s[0].rows, type = I0s[0].type;
    if (w.is_root) {   // the caller's matrices are already here: compute in place, chunked only to bound the arena
        for (int c = 0; c < w.count; c += chunk) {
            const int n = std::min(chunk, w.count - c);
            const int rc = mi_tvl1_calc_batch(w.h, n, I0s + w.first + c, I1s + w.first + c, flows + w.first + c, w.compute);
            if (rc) return rc;
        }
        MI_HIP_TRY(hipStreamSynchronize(w.compute));
        return MI_OK;
    }
    int rc = ensure_slots(w, chunk, W_, H_, type);
    if (rc) return rc;
    const size_t es = elem_size(type), in_pitch = (size_t)W_ * es, out_pitch = (size_t)W_ * 8;
    const size_t in_plane = in_pitch * H_, out_plane = out_pitch * H_;
    const int nchunks = (w.count + chunk - 1) / chunk;
    std::vector<mi_mat> a(chunk), b(chunk), f(chunk);
    auto copy_in = [&](int k) -> int {
        Slot &s = w.slot[k & 1];
        const int c0 = k * chunk, n = std::min(chunk, w.count - c0);
        if (k >= 2) MI_HIP_TRY(hipStreamWaitEvent(w.copy, s.calc_done, 0));   // the slot's previous compute has read its inputs
        for (int j = 0; j < n; ++j) {
            const mi_mat &m0 = I0s[w.first + c0 + j], &m1 = I1s[w.first + c0 + j];
            MI_HIP_TRY(hipMemcpy2DAsync((char *)s.in0 + j * in_plane, in_pitch, m0.data, m0.step, in_pitch, (size_t)H_, hipMemcpyDeviceToDevice, w.copy));
            MI_HIP_TRY(hipMemcpy2DAsync((char *)s.in1 + j * in_plane, in_pitch, m1.data, m1.step, in_pitch, (size_t)H_, hipMemcpyDeviceToDevice, w.copy));
        }
        MI_HIP_TRY(hipEventRecord(s.in_done, w.copy));
        return MI_OK;
    };
    if ((rc = copy_in(0))) return rc;
    for (int k = 0; k < nchunks; ++k) {
        Slot &s = w.slot[k & 1];
        const int c0 = k * chunk, n = std::min(chunk, w.count - c0);
        if (k + 1 < nchunks && (rc = copy_in(k + 1))) return rc;        // overlaps the compute of chunk k
        MI_HIP_TRY(hipStreamWaitEvent(w.compute, s.in_done, 0));
        if (k >= 2) MI_HIP_TRY(hipStreamWaitEvent(w.compute, s.out_done, 0));   // the slot's previous flows have left
        for (int j = 0; j < n; ++j) {
            a[j] = {(char *)s.in0 + j * in_plane, in_pitch, H_, W_, type};
            b[j] = {(char *)s.in1 + j * in_plane, in_pitch, H_, W_, type};
            f[j] = {(char *)s.out + j * out_plane, out_pitch, H_, W_, MI_32FC2};
            if (flows[w.first + c0 + j].type != MI_32FC2) { set_error("flow must be CV_32FC2"); return MI_ERR_BAD_TYPE; }
        }
        if ((rc = mi_tvl1_calc_batch(w.h, n, a.data(), b.data(), f.data(), w.compute))) return rc;
        MI_HIP_TRY(hipEventRecord(s.calc_done, w.compute));
        MI_HIP_TRY(hipStreamWaitEvent(w.copy, s.calc_done, 0));
        for (int j = 0; j < n; ++j) {
            mi_mat &mf = flows[w.first + c0 + j];
            MI_HIP_TRY(hipMemcpy2DAsync(mf.data, mf.step, (char *)s.out + j * out_plane, out_pitch, out_pitch, (size_t)H_, hipMemcpyDeviceToDevice, w.copy));
        }
        MI_HIP_TRY(hipEventRecord(s.out_done, w.copy));
    }
    MI_HIP_TRY(hipStreamSynchronize(w.copy));
    MI_HIP_TRY(hipStreamSynchronize(w.compute));
    return MI_OK;
}

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
    int prev = 0;
    (void)hipGetDevice(&prev);
    mi_tvl1_multi *m = new mi_tvl1_multi();
    m->P = *p;
    m->W.resize(n_devices);
    int rc = MI_OK;
    for (int i = 0; i < n_devices && !rc; ++i) {
        Worker &w = m->W[i];
        w.dev = device_ids ? device_ids[i] : i;
        if (w.dev < 0 || w.dev >= avail) { set_error("device id %d out of range (%d devices)", w.dev, avail); rc = MI_ERR_BAD_ARG; break; }
        w.is_root = (i == 0);
        rc = worker_init(w, *p, m->W[0].dev);
    }
    (void)hipSetDevice(prev);
    if (rc) { mi_tvl1_multi_destroy(m); return rc; }
    *out = m;
    return MI_OK;
}

int mi_tvl1_multi_device_count(const mi_tvl1_multi *m) { return m ? (int)m->W.size() : 0; }

int mi_tvl1_multi_set_chunk(mi_tvl1_multi *m, int pairs_per_chunk)
{
    MI_REQUIRE(m && pairs_per_chunk >= 1 && pairs_per_chunk <= 4096, MI_ERR_BAD_ARG, "chunk must be in [1, 4096]");
    m->chunk = pairs_per_chunk;
    return MI_OK;
}

int mi_tvl1_multi_calc_batch(mi_tvl1_multi *m, int n, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows)
{
    MI_REQUIRE(m && n > 0 && I0s && I1s && flows, MI_ERR_BAD_ARG, "empty batch");
    for (int i = 0; i < n; ++i) {
        MI_REQUIRE(I0s[i].data && I1s[i].data && flows[i].data, MI_ERR_BAD_ARG, "null data pointer");
        MI_REQUIRE(I0s[i].rows == I0s[0].rows && I0s[i].cols == I0s[0].cols && I0s[i].type == I0s[0].type && I1s[i].type == I0s[0].type &&
                   I1s[i].rows == I0s[0].rows && I1s[i].cols == I0s[0].cols && flows[i].rows == I0s[0].rows && flows[i].cols == I0s[0].cols,
                   MI_ERR_BAD_SIZE, "all pairs of a batch must share size and type");
    }
    MI_REQUIRE(I0s[0].type == MI_8UC1 || I0s[0].type == MI_32FC1, MI_ERR_BAD_TYPE, "I0 must be CV_8UC1 or CV_32FC1");
    const int G = (int)m->W.size();
    const int per = (n + G - 1) / G;   // static block partition
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int i = 0; i < G; ++i) {
        Worker &w = m->W[i];
        w.first = std::min(n, i * per);
        w.count = std::min(per, n - w.first);
        w.rc = MI_OK; w.err.clear();
    }
    std::vector<std::thread> th;
    for (int i = 0; i < G; ++i)
        th.emplace_back([m, i, I0s, I1s, flows] {
            Worker &w = m->W[i];
            w.rc = worker_run(w, m->chunk, I0s, I1s, flows);
            if (w.rc) w.err = mi_last_error();   // the error channel is thread local: carry the text back
        });
    for (auto &t : th) t.join();
    (void)hipSetDevice(prev);
    for (int i = 0; i < G; ++i)
        if (m->W[i].rc) { set_error("device %d: %s", m->W[i].dev, m->W[i].err.c_str()); return m->W[i].rc; }
    return MI_OK;
}

void mi_tvl1_multi_destroy(mi_tvl1_multi *m)
{
    if (!m) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (Worker &w : m->W) {
        (void)hipSetDevice(w.dev);
        free_slots(w);
        for (Slot &s : w.slot) {
            if (s.in_done) (void)hipEventDestroy(s.in_done);
            if (s.calc_done) (void)hipEventDestroy(s.calc_done);
            if (s.out_done) (void)hipEventDestroy(s.out_done);
        }
        if (w.h) mi_tvl1_destroy(w.h);
        if (w.compute) (void)hipStreamDestroy(w.compute);
        if (w.copy) (void)hipStreamDestroy(w.copy);
    }
    (void)hipSetDevice(prev);
    delete m;
}

}  // extern "C"
