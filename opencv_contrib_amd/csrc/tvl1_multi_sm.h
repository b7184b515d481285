// Batched-frames mode over several GPUs of one node: the worker state machine, written against a BACKEND policy so that the
// same source runs over HIP (tvl1_multi.cpp: HipBackend) and over a recording fake with DISTINCT device ids on a host without a
// GPU (tests/cpp/multi_sm_test.cpp).  No HIP header is included here.
//
// Shape: the pairs of a batch are independent, so the batch is cut into contiguous shards, one per device (pair i ->
// worker i / ceil(n / G)), with NO data-path collective.  One PERSISTENT host thread per device -- the reference's own
// multi-device idiom is cv::cuda::setDevice per thread (modules/cudaoptflow/test/test_optflow.cpp:62, :468-527 for the
// concurrent-instances model) -- owns a TV-L1 handle, a compute stream and a copy stream on its device.  The caller's matrices
// live on the ROOT device (the first id): a worker on another device pulls its shard over xGMI with peer-to-peer 2-D copies into
// dense staging planes, computes, and pushes the flows back, in chunks and double buffered, so that the copy-in of chunk k + 1
// and the copy-out of chunk k - 1 overlap the compute of chunk k.  The worker on the root device computes in place.
//
// Backend contract (all functions static, int status = MI_OK or an MI_ERR_* with the text available from last_error()):
//   set_device(dev)                                   -- makes dev current for the CALLING thread
//   can_access_peer(&can, dev, peer), enable_peer(peer)   -- enable_peer acts on the current device; "already enabled" is MI_OK
//   stream_create(&s) / stream_destroy(s) / stream_sync(s) / stream_wait_event(s, e)
//   event_create(&e) / event_destroy(e) / event_record(e, s)
//   dev_malloc(&p, bytes) / dev_free(p)
//   copy2d_async(dst, dpitch, src, spitch, width_bytes, rows, stream)
//   link_create(&link, root_dev, dev)                 -- the transport between the root and this worker's device, created by the worker's
//                                                        thread (device `dev` current on entry and on return).  *link == nullptr: no
//                                                        collective library / not usable for this pair -- the planes travel as peer copies
//   link_destroy(link) / link_sync(link)              -- sync drains the link's ROOT-side stream
//   link_begin(link) / link_end(link)                 -- bracket the planes of one chunk and direction (an RCCL group)
//   link_plane(link, dst, dpitch, src, spitch, width_bytes, rows, to_worker, worker_stream)
//                                                     -- one plane root -> worker (to_worker) or worker -> root, ordered on the
//                                                        worker's stream like copy2d_async
//   tvl1_create(&params, &h) / tvl1_destroy(h) / tvl1_calc_batch(h, n, I0s, I1s, flows, stream)
//   last_error()                                      -- thread-local text of the calling thread's last failure
#pragma once
#include "miflow/c_api.h"
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mi {
namespace multi {

inline size_t elem_size(int type) { return type == MI_8UC1 ? 1 : 4; }

template <class B>
class Machine {
public:
    struct Slot {
        void *in0 = nullptr, *in1 = nullptr, *out = nullptr;   // chunk x dense planes on the worker's device
        void *in_done = nullptr, *calc_done = nullptr, *out_done = nullptr;
    };
    struct Worker {
        int dev = 0;
        bool is_root = false;
        void *h = nullptr;
        void *compute = nullptr, *copy = nullptr;
        void *link = nullptr;   // RCCL transport to the root (nullptr: peer copies)
        Slot slot[2];
        int cap_chunk = 0, cap_w = 0, cap_h = 0, cap_type = -1;
        // per call
        int first = 0, count = 0, rc = MI_OK;
        std::string err;
        // persistent thread
        std::thread th;
        unsigned long long job = 0, done = 0;   // generation counters, guarded by Machine::mu_
    };

    Machine() = default;
    Machine(const Machine &) = delete;
    Machine &operator=(const Machine &) = delete;
    ~Machine() { shutdown(); }

    // devices[0] is the root.  On failure everything created so far is released and the text is in error().
    int init(const mi_tvl1_params &P, const std::vector<int> &devices)
    {
        P_ = P;
        W_.resize(devices.size());
        for (size_t i = 0; i < devices.size(); ++i) { W_[i].dev = devices[i]; W_[i].is_root = (i == 0); }
        // resources are created by the thread that will use them (device currency is per thread), one worker at a time so that a
        // failure names its device; the threads then stay alive for the life of the object
        for (size_t i = 0; i < W_.size(); ++i) {
            Worker &w = W_[i];
            w.th = std::thread([this, i] { thread_main((int)i); });
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait(lk, [&] { return W_[i].done == 1 || W_[i].rc != MI_OK; });
            if (w.rc) { err_ = "device " + std::to_string(w.dev) + ": " + w.err; const int rc = w.rc; lk.unlock(); shutdown(); return rc; }
        }
        return MI_OK;
    }

    int device_count() const { return (int)W_.size(); }
    void set_chunk(int c) { chunk_ = c; }
    void set_use_links(bool on) { use_links_ = on; }   // before init(): false = peer copies only
    int link_count() const { int n = 0; for (const Worker &w : W_) n += w.link != nullptr; return n; }
    const std::string &error() const { return err_; }
    const Worker &worker(int i) const { return W_[i]; }

    // Validates every pair up front (nothing is enqueued for a batch that would fail halfway), shards, runs, waits for ALL workers.
    int calc_batch(int n, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows)
    {
        if (n <= 0 || !I0s || !I1s || !flows) { err_ = "empty batch"; return MI_ERR_BAD_ARG; }
        const int type = I0s[0].type, rows = I0s[0].rows, cols = I0s[0].cols;
        if (type != MI_8UC1 && type != MI_32FC1) { err_ = "I0 must be CV_8UC1 or CV_32FC1"; return MI_ERR_BAD_TYPE; }
        if (rows <= 0 || cols <= 0) { err_ = "empty image"; return MI_ERR_BAD_SIZE; }
        const size_t in_row = (size_t)cols * elem_size(type), out_row = (size_t)cols * 8;
        for (int i = 0; i < n; ++i) {
            if (!I0s[i].data || !I1s[i].data || !flows[i].data) { err_ = "null data pointer"; return MI_ERR_BAD_ARG; }
            if (I0s[i].rows != rows || I0s[i].cols != cols || I1s[i].rows != rows || I1s[i].cols != cols || flows[i].rows != rows ||
                flows[i].cols != cols) { err_ = "all pairs of a batch must share size and type"; return MI_ERR_BAD_SIZE; }
            if (I0s[i].type != type || I1s[i].type != type) { err_ = "all pairs of a batch must share size and type"; return MI_ERR_BAD_TYPE; }
            if (flows[i].type != MI_32FC2) { err_ = "flow must be CV_32FC2"; return MI_ERR_BAD_TYPE; }
            // the steps are the pitches of the peer-to-peer 2-D copies
            if (I0s[i].step < in_row || I1s[i].step < in_row || flows[i].step < out_row) { err_ = "matrix step smaller than a row"; return MI_ERR_BAD_SIZE; }
        }
        const int G = (int)W_.size();
        const int per = (n + G - 1) / G;   // static block partition
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (int i = 0; i < G; ++i) {
                Worker &w = W_[i];
                w.first = std::min(n, i * per);
                w.count = std::min(per, n - w.first);
                w.rc = MI_OK; w.err.clear();
                ++w.job;
            }
            I0s_ = I0s; I1s_ = I1s; flows_ = flows;
        }
        cv_job_.notify_all();
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait(lk, [&] { for (const Worker &w : W_) if (w.done != w.job) return false; return true; });
        }
        for (const Worker &w : W_)
            if (w.rc) { err_ = "device " + std::to_string(w.dev) + ": " + w.err; return w.rc; }
        return MI_OK;
    }

    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_job_.notify_all();
        for (Worker &w : W_) if (w.th.joinable()) w.th.join();
        W_.clear();
    }

private:
    static int fail(Worker &w, int rc) { if (rc && w.err.empty()) w.err = B::last_error(); return rc; }
    // one per backend type and process: all machines of a process share the root devices' RCCL state
    static std::mutex &root_mu() { static std::mutex m; return m; }
#define MI_MSM_TRY(expr) do { const int _rc = (expr); if (_rc) return fail(w, _rc); } while (0)

    int worker_init(Worker &w)
    {
        const int root = W_[0].dev;
        MI_MSM_TRY(B::set_device(w.dev));
        if (!w.is_root && w.dev != root) {
            // both directions: the worker's copy stream reads the root's inputs and writes the root's flows, and a root-side
            // engine may be chosen for either copy
            int can = 0;
            MI_MSM_TRY(B::can_access_peer(&can, w.dev, root));
            if (!can) { w.err = "device " + std::to_string(w.dev) + " cannot access device " + std::to_string(root) + " peer-to-peer"; return MI_ERR_HIP; }
            MI_MSM_TRY(B::enable_peer(root));
            MI_MSM_TRY(B::can_access_peer(&can, root, w.dev));
            if (can) {
                MI_MSM_TRY(B::set_device(root));
                const int rc = B::enable_peer(w.dev);
                MI_MSM_TRY(B::set_device(w.dev));   // the thread's device is restored whatever the enable returned
                MI_MSM_TRY(rc);
            }
        }
        MI_MSM_TRY(B::tvl1_create(&P_, &w.h));
        MI_MSM_TRY(B::stream_create(&w.compute));
        MI_MSM_TRY(B::stream_create(&w.copy));
        // north_star: "RCCL over xGMI for the scatter/gather only" -- a two-rank communicator per (root, worker) pair, owned by this
        // worker's thread alone; absent library or an unusable pair (e.g. the same physical device twice) leave the peer copies
        // Every link shares the ROOT device: communicator creation on it, and the brackets that enqueue on its per-link streams, are
        // serialised process-wide (root_mu: RCCL guarantees no progress for concurrent communicator creation on one device, nor for
        // concurrent groups from several threads that all contain it).  Only the ENQUEUE is serialised -- the transfers of different
        // links still overlap on their own streams.
        if (!w.is_root && use_links_) {
            std::lock_guard<std::mutex> lk(root_mu());
            MI_MSM_TRY(B::link_create(&w.link, root, w.dev));
        }
        for (Slot &s : w.slot) {
            MI_MSM_TRY(B::event_create(&s.in_done));
            MI_MSM_TRY(B::event_create(&s.calc_done));
            MI_MSM_TRY(B::event_create(&s.out_done));
        }
        return MI_OK;
    }

    void free_slots(Worker &w)
    {
        for (Slot &s : w.slot) {
            if (s.in0) (void)B::dev_free(s.in0);
            if (s.in1) (void)B::dev_free(s.in1);
            if (s.out) (void)B::dev_free(s.out);
            s.in0 = s.in1 = s.out = nullptr;
        }
        w.cap_chunk = 0;
    }

    void worker_release(Worker &w)
    {
        (void)B::set_device(w.dev);
        if (w.copy) (void)B::stream_sync(w.copy);
        if (w.compute) (void)B::stream_sync(w.compute);
        if (w.link) { (void)B::link_sync(w.link); (void)B::link_destroy(w.link); w.link = nullptr; }
        free_slots(w);
        for (Slot &s : w.slot) {
            if (s.in_done) (void)B::event_destroy(s.in_done);
            if (s.calc_done) (void)B::event_destroy(s.calc_done);
            if (s.out_done) (void)B::event_destroy(s.out_done);
            s.in_done = s.calc_done = s.out_done = nullptr;
        }
        if (w.h) B::tvl1_destroy(w.h);
        if (w.compute) (void)B::stream_destroy(w.compute);
        if (w.copy) (void)B::stream_destroy(w.copy);
        w.h = w.compute = w.copy = nullptr;
    }

    int ensure_slots(Worker &w, int chunk, int Wd, int Ht, int type)
    {
        if (w.cap_chunk >= chunk && w.cap_w == Wd && w.cap_h == Ht && w.cap_type == type) return MI_OK;
        // the staging planes of the previous size may still be in flight only after a failed call; both streams were drained then
        free_slots(w);
        const size_t es = elem_size(type), in_bytes = (size_t)Wd * Ht * es * chunk, out_bytes = (size_t)Wd * Ht * 8 * chunk;
        for (Slot &s : w.slot) {
            MI_MSM_TRY(B::dev_malloc(&s.in0, in_bytes));
            MI_MSM_TRY(B::dev_malloc(&s.in1, in_bytes));
            MI_MSM_TRY(B::dev_malloc(&s.out, out_bytes));
        }
        w.cap_chunk = chunk; w.cap_w = Wd; w.cap_h = Ht; w.cap_type = type;
        return MI_OK;
    }

    // One worker's shard: pairs [first, first + count) of the caller's arrays (enqueue only; worker_run drains the streams).
    int enqueue_shard(Worker &w, int chunk, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows)
    {
        const int Wd = I0s[0].cols, Ht = I0s[0].rows, type = I0s[0].type;
        if (w.is_root) {   // the caller's matrices are already here: compute in place, chunked only to bound the arena
            for (int c = 0; c < w.count; c += chunk) {
                const int n = std::min(chunk, w.count - c);
                MI_MSM_TRY(B::tvl1_calc_batch(w.h, n, I0s + w.first + c, I1s + w.first + c, flows + w.first + c, w.compute));
            }
            return MI_OK;
        }
        MI_MSM_TRY(ensure_slots(w, chunk, Wd, Ht, type));
        const size_t es = elem_size(type), in_pitch = (size_t)Wd * es, out_pitch = (size_t)Wd * 8;
        const size_t in_plane = in_pitch * Ht, out_plane = out_pitch * Ht;
        const int nchunks = (w.count + chunk - 1) / chunk;
        std::vector<mi_mat> a(chunk), b(chunk), f(chunk);
        auto copy_in = [&](int k) -> int {
            Slot &s = w.slot[k & 1];
            const int c0 = k * chunk, n = std::min(chunk, w.count - c0);
            if (k >= 2) MI_MSM_TRY(B::stream_wait_event(w.copy, s.calc_done));   // the slot's previous compute has read its inputs
            std::unique_lock<std::mutex> bracket(root_mu(), std::defer_lock);   // held from link_begin to link_end (see worker_init)
            if (w.link) { bracket.lock(); MI_MSM_TRY(B::link_begin(w.link)); }
            int rc_planes = MI_OK;
            for (int j = 0; j < n && !rc_planes; ++j) {
                const mi_mat &m0 = I0s[w.first + c0 + j], &m1 = I1s[w.first + c0 + j];
                if (w.link) {
                    rc_planes = B::link_plane(w.link, (char *)s.in0 + j * in_plane, in_pitch, m0.data, m0.step, in_pitch, (size_t)Ht, 1, w.copy);
                    if (!rc_planes) rc_planes = B::link_plane(w.link, (char *)s.in1 + j * in_plane, in_pitch, m1.data, m1.step, in_pitch, (size_t)Ht, 1, w.copy);
                } else {
                    rc_planes = B::copy2d_async((char *)s.in0 + j * in_plane, in_pitch, m0.data, m0.step, in_pitch, (size_t)Ht, w.copy);
                    if (!rc_planes) rc_planes = B::copy2d_async((char *)s.in1 + j * in_plane, in_pitch, m1.data, m1.step, in_pitch, (size_t)Ht, w.copy);
                }
            }
            if (rc_planes) (void)fail(w, rc_planes);
            if (w.link) { const int rc_end = B::link_end(w.link); bracket.unlock(); if (!rc_planes) MI_MSM_TRY(rc_end); }   // a group once begun is always closed
            if (rc_planes) return rc_planes;
            MI_MSM_TRY(B::event_record(s.in_done, w.copy));
            return MI_OK;
        };
        MI_MSM_TRY(copy_in(0));
        for (int k = 0; k < nchunks; ++k) {
            Slot &s = w.slot[k & 1];
            const int c0 = k * chunk, n = std::min(chunk, w.count - c0);
            if (k + 1 < nchunks) MI_MSM_TRY(copy_in(k + 1));                       // overlaps the compute of chunk k
            MI_MSM_TRY(B::stream_wait_event(w.compute, s.in_done));
            if (k >= 2) MI_MSM_TRY(B::stream_wait_event(w.compute, s.out_done));   // the slot's previous flows have left
            for (int j = 0; j < n; ++j) {
                a[j] = {(char *)s.in0 + j * in_plane, in_pitch, Ht, Wd, type};
                b[j] = {(char *)s.in1 + j * in_plane, in_pitch, Ht, Wd, type};
                f[j] = {(char *)s.out + j * out_plane, out_pitch, Ht, Wd, MI_32FC2};
            }
            MI_MSM_TRY(B::tvl1_calc_batch(w.h, n, a.data(), b.data(), f.data(), w.compute));
            MI_MSM_TRY(B::event_record(s.calc_done, w.compute));
            MI_MSM_TRY(B::stream_wait_event(w.copy, s.calc_done));
            std::unique_lock<std::mutex> bracket(root_mu(), std::defer_lock);
            if (w.link) { bracket.lock(); MI_MSM_TRY(B::link_begin(w.link)); }
            int rc_planes = MI_OK;
            for (int j = 0; j < n && !rc_planes; ++j) {
                mi_mat &mf = flows[w.first + c0 + j];
                rc_planes = w.link ? B::link_plane(w.link, mf.data, mf.step, (char *)s.out + j * out_plane, out_pitch, out_pitch, (size_t)Ht, 0, w.copy)
                                   : B::copy2d_async(mf.data, mf.step, (char *)s.out + j * out_plane, out_pitch, out_pitch, (size_t)Ht, w.copy);
            }
            if (rc_planes) (void)fail(w, rc_planes);
            if (w.link) { const int rc_end = B::link_end(w.link); bracket.unlock(); if (!rc_planes) MI_MSM_TRY(rc_end); }
            if (rc_planes) return rc_planes;
            MI_MSM_TRY(B::event_record(s.out_done, w.copy));
        }
        return MI_OK;
    }

    // Whatever enqueue_shard returned, BOTH streams are drained before the worker reports: after an error peer copies may still
    // be in flight against the caller's buffers and the staging slots.
    int worker_run(Worker &w, int chunk, const mi_mat *I0s, const mi_mat *I1s, mi_mat *flows)
    {
        MI_MSM_TRY(B::set_device(w.dev));
        if (w.count == 0) return MI_OK;
        const int rc = enqueue_shard(w, chunk, I0s, I1s, flows);
        const int s1 = B::stream_sync(w.copy), s2 = B::stream_sync(w.compute);
        const int s3 = w.link ? B::link_sync(w.link) : MI_OK;   // the root side of the link: the flows it received are in the caller's matrices
        if (rc) return rc;
        MI_MSM_TRY(s1);
        MI_MSM_TRY(s2);
        MI_MSM_TRY(s3);
        return MI_OK;
    }
#undef MI_MSM_TRY

    void thread_main(int i)
    {
        Worker &w = W_[i];
        {
            const int rc = worker_init(w);
            std::lock_guard<std::mutex> lk(mu_);
            w.rc = rc;
            if (!rc) w.done = w.job = 1;   // generation 1 = initialised
        }
        cv_done_.notify_all();
        if (w.rc) { worker_release(w); return; }
        for (;;) {
            const mi_mat *a, *b;
            mi_mat *f;
            int chunk;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_job_.wait(lk, [&] { return quit_ || w.job != w.done; });
                if (quit_ && w.job == w.done) break;
                a = I0s_; b = I1s_; f = flows_; chunk = chunk_;
            }
            const int rc = worker_run(w, chunk, a, b, f);
            {
                std::lock_guard<std::mutex> lk(mu_);
                w.rc = rc;
                w.done = w.job;
            }
            cv_done_.notify_all();
        }
        worker_release(w);
    }

    mi_tvl1_params P_{};
    int chunk_ = 16;
    bool use_links_ = true;
    std::vector<Worker> W_;
    std::mutex mu_;
    std::condition_variable cv_job_, cv_done_;
    bool quit_ = false;
    const mi_mat *I0s_ = nullptr, *I1s_ = nullptr;
    mi_mat *flows_ = nullptr;
    std::string err_;
};

}  // namespace multi
}  // namespace mi
