// Host-only test of the multi-device worker state machine (opencv_contrib_amd/csrc/tvl1_multi_sm.h) against a FAKE backend whose
// device table has distinct, non-contiguous ids.  No GPU, no HIP: `g++ -std=c++17 -pthread -Iinclude -Iopencv_contrib_amd/csrc`.
//
// The fake models what the machine relies on:
//   * device currency is PER THREAD; every stream / event / allocation / handle remembers the device that was current when it was
//     created and every later use is checked against the calling thread's current device;
//   * streams are queues executed LAZILY (only when somebody synchronises) by a randomised scheduler that honours nothing but
//     stream order and event waits with HIP's semantics (a wait captures the event's latest record at enqueue time; an event
//     never recorded is no dependency) -- so a missing wait shows up as wrong bytes for some seed;
//   * copies move real bytes between "device" buffers; the fake TV-L1 writes flow = (I0 + 1, 2 * I1) per pixel;
//   * one chosen call can be made to fail.
// Exit code 0 = all checks passed; otherwise the failing check's line is printed.
#include "tvl1_multi_sm.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <random>
#include <set>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

namespace fake {

struct Stream;
struct Event { int dev; Stream *rec_stream = nullptr; long rec_ticket = -1; };   // latest record enqueued
struct Op {
    std::function<void()> run;      // payload (copy / calc); empty for pure ordering ops
    Stream *wait_stream = nullptr;  // wait: the stream whose op `wait_ticket` must have executed
    long wait_ticket = -1;
};
struct Stream { int dev; std::vector<Op> q; size_t next = 0; long base = 0; };   // ticket of q[i] = base + i
struct Handle { int dev; };

static thread_local int t_dev = -1;
static thread_local std::string t_err;

struct World {
    std::mutex mu;
    std::set<int> valid{0, 1, 2, 3, 4, 5, 6, 7, 9};
    std::map<std::pair<int, int>, int> peer_enabled;   // (device, peer) -> count
    std::map<void *, int> owner;                       // allocation -> device
    std::vector<std::string> log;                      // "dev:op"
    std::map<int, std::vector<Stream *>> streams;      // per device, for the scheduler
    std::mt19937 rng{1};
    // failure injection
    std::string fail_op; int fail_dev = -1; int fail_after = 0;   // fail the (fail_after+1)-th matching call
    int live_streams = 0, live_events = 0, live_allocs = 0, live_handles = 0, live_links = 0;
    bool links_enabled = false;   // the RCCL transport between the root and the other workers (off: peer copies)
    long groups = 0, messages = 0, pitched_planes = 0;
    int open_brackets = 0, max_open_brackets = 0;   // brackets (RCCL groups) open right now / the most ever open at once
} G;

static int maybe_fail(const char *op)
{
    if (G.fail_dev == t_dev && G.fail_op == op) {
        if (G.fail_after-- == 0) { G.fail_dev = -1; t_err = std::string("injected failure in ") + op; return MI_ERR_HIP; }
    }
    return MI_OK;
}
static void note(const char *op) { G.log.push_back(std::to_string(t_dev) + ":" + op); }

// run ready ops of the device's streams in random order until `target` has drained
static void drain(Stream *target)
{
    std::vector<Stream *> &S = G.streams[target->dev];
    while (target->next < target->q.size()) {
        std::vector<Stream *> ready;
        for (Stream *s : S) {
            if (s->next >= s->q.size()) continue;
            const Op &o = s->q[s->next];
            if (o.wait_stream && o.wait_stream->base + (long)o.wait_stream->next <= o.wait_ticket) continue;   // dependency not executed yet
            ready.push_back(s);
        }
        CHECK(!ready.empty());   // a cycle of waits would be a deadlock on the GPU as well
        Stream *s = ready[G.rng() % ready.size()];
        Op &o = s->q[s->next];
        if (o.run) o.run();
        ++s->next;
    }
    // compact
    target->base += (long)target->q.size();
    target->q.clear();
    target->next = 0;
}

struct Backend {
    static int set_device(int dev)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        if (!G.valid.count(dev)) { t_err = "invalid device"; return MI_ERR_HIP; }
        t_dev = dev;
        note("set_device");
        return maybe_fail("set_device");
    }
    static int can_access_peer(int *can, int dev, int peer)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        *can = (dev != 9 && peer != 9);   // device 9 has no link
        return MI_OK;
    }
    static int enable_peer(int peer)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        CHECK(t_dev >= 0 && peer != t_dev);
        ++G.peer_enabled[{t_dev, peer}];   // a second enable is "already enabled" = MI_OK by contract
        note("enable_peer");
        return maybe_fail("enable_peer");
    }
    static int stream_create(void **s)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        if (int rc = maybe_fail("stream_create")) return rc;
        Stream *st = new Stream{t_dev};
        G.streams[t_dev].push_back(st);
        ++G.live_streams;
        *s = st;
        return MI_OK;
    }
    static int stream_destroy(void *s)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        Stream *st = (Stream *)s;
        CHECK(st->dev == t_dev && st->next == st->q.size());
        auto &v = G.streams[t_dev];
        v.erase(std::find(v.begin(), v.end(), st));
        delete st;
        --G.live_streams;
        return MI_OK;
    }
    static int stream_sync(void *s)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        Stream *st = (Stream *)s;
        CHECK(st->dev == t_dev);
        note("stream_sync");
        drain(st);
        return MI_OK;
    }
    static int stream_wait_event(void *s, void *e)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        Stream *st = (Stream *)s; Event *ev = (Event *)e;
        CHECK(st->dev == t_dev && ev->dev == t_dev);
        if (int rc = maybe_fail("stream_wait_event")) return rc;
        Op o;
        if (ev->rec_stream && ev->rec_stream != st) { o.wait_stream = ev->rec_stream; o.wait_ticket = ev->rec_ticket; }
        st->q.push_back(o);
        return MI_OK;
    }
    static int event_create(void **e)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        *e = new Event{t_dev};
        ++G.live_events;
        return MI_OK;
    }
    static int event_destroy(void *e)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        CHECK(((Event *)e)->dev == t_dev);
        delete (Event *)e;
        --G.live_events;
        return MI_OK;
    }
    static int event_record(void *e, void *s)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        Stream *st = (Stream *)s; Event *ev = (Event *)e;
        CHECK(st->dev == t_dev && ev->dev == t_dev);
        ev->rec_stream = st;
        ev->rec_ticket = st->base + (long)st->q.size();
        st->q.push_back(Op{});
        return MI_OK;
    }
    static int dev_malloc(void **p, size_t bytes)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        if (int rc = maybe_fail("dev_malloc")) return rc;
        *p = malloc(bytes ? bytes : 1);
        memset(*p, 0xCD, bytes);
        G.owner[*p] = t_dev;
        ++G.live_allocs;
        return MI_OK;
    }
    static int dev_free(void *p)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        CHECK(G.owner.count(p) && G.owner[p] == t_dev);
        G.owner.erase(p);
        free(p);
        --G.live_allocs;
        return MI_OK;
    }
    static int copy2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t wbytes, size_t rows, void *s)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        Stream *st = (Stream *)s;
        CHECK(st->dev == t_dev);
        if (int rc = maybe_fail("copy2d_async")) return rc;
        Op o;
        o.run = [=] { for (size_t r = 0; r < rows; ++r) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, wbytes); };
        st->q.push_back(o);
        return MI_OK;
    }
    // ---- the RCCL link of a (root, worker) pair.  Model: a dense plane is a MESSAGE of two halves, one on the worker's stream and one
    // on the link's root-side stream.  The bytes move when the WORKER's half executes (a send does not complete before its data has
    // left, a receive not before it has arrived, so the worker-side stream order is what protects the staging slots); the root's half
    // is a marker behind it that somebody has to wait for (link_sync) before the call may return -- all_streams_idle() checks that.
    // Pitched planes are peer copies on the worker's stream.
    struct Link { int root, dev; Stream *root_stream; bool in_group = false; };
    static int link_create(void **out, int root, int dev)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        CHECK(t_dev == dev);
        *out = nullptr;
        if (!G.links_enabled || root == dev) return MI_OK;
        if (int rc = maybe_fail("link_create")) return rc;
        CHECK(G.open_brackets == 0);   // communicator creation on the root device does not overlap a group either
        Stream *st = new Stream{root};
        G.streams[root].push_back(st);
        ++G.live_streams;
        *out = new Link{root, dev, st};
        ++G.live_links;
        note("link_create");
        return MI_OK;
    }
    static int link_destroy(void *l)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        Link *L = (Link *)l;
        CHECK(t_dev == L->dev && !L->in_group && L->root_stream->next == L->root_stream->q.size());
        auto &v = G.streams[L->root];
        v.erase(std::find(v.begin(), v.end(), L->root_stream));
        delete L->root_stream;
        --G.live_streams;
        delete L;
        --G.live_links;
        return MI_OK;
    }
    static int link_sync(void *l)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        Link *L = (Link *)l;
        CHECK(t_dev == L->dev);
        note("link_sync");
        drain(L->root_stream);
        return MI_OK;
    }
    static int link_begin(void *l)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        Link *L = (Link *)l;
        CHECK(t_dev == L->dev && !L->in_group);
        if (int rc = maybe_fail("link_begin")) return rc;
        // what the real RCCL groups need: never two brackets open at once that contain the same root device
        CHECK(G.open_brackets == 0);
        ++G.open_brackets;
        G.max_open_brackets = std::max(G.max_open_brackets, G.open_brackets);
        L->in_group = true;
        ++G.groups;
        return MI_OK;
    }
    static int link_end(void *l)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        Link *L = (Link *)l;
        CHECK(t_dev == L->dev);
        if (L->in_group) --G.open_brackets;
        L->in_group = false;
        return MI_OK;
    }
    static int link_plane(void *l, void *dst, size_t dpitch, const void *src, size_t spitch, size_t wbytes, size_t rows, int to_worker, void *ws)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        Link *L = (Link *)l;
        Stream *st = (Stream *)ws;
        CHECK(t_dev == L->dev && st->dev == L->dev && L->in_group);   // every plane of a chunk inside ONE bracket
        if (int rc = maybe_fail("link_plane")) return rc;
        const auto copy = [=] { for (size_t r = 0; r < rows; ++r) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, wbytes); };
        if (dpitch != wbytes || spitch != wbytes) {   // pitched: a peer copy on the worker's stream
            ++G.pitched_planes;
            Op o; o.run = copy;
            st->q.push_back(o);
            return MI_OK;
        }
        ++G.messages;
        if (to_worker) {   // sender half (marker) on the root-side stream, receiver half on the worker's stream waits for it
            const long t = L->root_stream->base + (long)L->root_stream->q.size();
            L->root_stream->q.push_back(Op{});
            Op o; o.run = copy; o.wait_stream = L->root_stream; o.wait_ticket = t;
            st->q.push_back(o);
            // the worker-side drain must be able to run the root-side marker: it lives on another device's stream list, so execute
            // markers eagerly (a send has no precondition in this model: the caller's inputs are resident)
            L->root_stream->next = L->root_stream->q.size();
        } else {        // sender half (the bytes) on the worker's stream, receiver half on the root-side stream behind it
            const long t = st->base + (long)st->q.size();
            Op o; o.run = copy;
            st->q.push_back(o);
            Op r; r.wait_stream = st; r.wait_ticket = t;
            L->root_stream->q.push_back(r);
        }
        return MI_OK;
    }
    static int tvl1_create(const mi_tvl1_params *, void **h)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        if (int rc = maybe_fail("tvl1_create")) return rc;
        *h = new Handle{t_dev};
        ++G.live_handles;
        note("tvl1_create");
        return MI_OK;
    }
    static void tvl1_destroy(void *h)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        CHECK(((Handle *)h)->dev == t_dev);
        delete (Handle *)h;
        --G.live_handles;
    }
    static int tvl1_calc_batch(void *h, int n, const mi_mat *a, const mi_mat *b, mi_mat *f, void *s)
    {
        std::lock_guard<std::mutex> lk(G.mu);
        Stream *st = (Stream *)s;
        CHECK(((Handle *)h)->dev == t_dev && st->dev == t_dev && n > 0);
        if (int rc = maybe_fail("tvl1_calc_batch")) return rc;
        note("tvl1_calc_batch");
        for (int i = 0; i < n; ++i) {
            const mi_mat A = a[i], B = b[i], F = f[i];
            Op o;
            o.run = [=] {
                for (int y = 0; y < A.rows; ++y)
                    for (int x = 0; x < A.cols; ++x) {
                        const float v0 = ((const float *)((const char *)A.data + y * A.step))[x];
                        const float v1 = ((const float *)((const char *)B.data + y * B.step))[x];
                        float *o2 = (float *)((char *)F.data + y * F.step) + 2 * x;
                        o2[0] = v0 + 1.f; o2[1] = 2.f * v1;
                    }
            };
            st->q.push_back(o);
        }
        return MI_OK;
    }
    static const char *last_error() { return t_err.c_str(); }
};

}  // namespace fake

using M = mi::multi::Machine<fake::Backend>;

// every stream of every device has executed everything that was enqueued on it (a call that returned left nothing in flight)
static bool all_streams_idle()
{
    std::lock_guard<std::mutex> lk(fake::G.mu);
    for (auto &kv : fake::G.streams)
        for (fake::Stream *s : kv.second)
            if (s->next != s->q.size()) return false;
    return true;
}

struct Batch {
    int n, rows, cols;
    size_t in_step, out_step;   // pitched "root device" matrices
    std::vector<std::vector<float>> I0, I1, F;
    std::vector<mi_mat> a, b, f;
    int pi, po;   // pitch padding, in elements (0: dense matrices)
    Batch(int n_, int rows_, int cols_, unsigned seed, bool dense = false) : n(n_), rows(rows_), cols(cols_), pi(dense ? 0 : 3), po(dense ? 0 : 5)
    {
        in_step = (cols + pi) * 4; out_step = (cols + po) * 8;
        std::mt19937 r(seed);
        I0.resize(n); I1.resize(n); F.resize(n); a.resize(n); b.resize(n); f.resize(n);
        for (int i = 0; i < n; ++i) {
            I0[i].resize(rows * (cols + pi)); I1[i].resize(rows * (cols + pi)); F[i].assign(rows * (cols + po) * 2, -7.f);
            for (float &v : I0[i]) v = (float)(r() % 1000);
            for (float &v : I1[i]) v = (float)(r() % 1000);
            a[i] = {I0[i].data(), in_step, rows, cols, MI_32FC1};
            b[i] = {I1[i].data(), in_step, rows, cols, MI_32FC1};
            f[i] = {F[i].data(), out_step, rows, cols, MI_32FC2};
        }
    }
    bool correct(int i) const
    {
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x) {
                const float v0 = I0[i][y * (cols + pi) + x], v1 = I1[i][y * (cols + pi) + x];
                const float *o = &F[i][y * (cols + po) * 2 + 2 * x];
                if (o[0] != v0 + 1.f || o[1] != 2.f * v1) return false;
            }
        // the pitch padding of the caller's flow matrices is never written
        for (int y = 0; y < rows; ++y)
            for (int x = cols * 2; x < (cols + po) * 2; ++x)
                if (F[i][y * (cols + po) * 2 + x] != -7.f) return false;
        return true;
    }
};

static int count_log(const std::string &entry)
{
    int c = 0;
    for (const std::string &s : fake::G.log) c += (s == entry);
    return c;
}

int main()
{
    mi_tvl1_params P;
    memset(&P, 0, sizeof(P));
    const std::vector<int> devs{5, 3, 6};   // root = 5; distinct, unordered ids

    // 1. construction: every worker's resources on its own device, peer access enabled both ways between each worker and the root
    {
        M m;
        CHECK(m.init(P, devs) == MI_OK);
        CHECK(m.device_count() == 3);
        CHECK((fake::G.peer_enabled[{3, 5}] == 1) && (fake::G.peer_enabled[{5, 3}] == 1));
        CHECK((fake::G.peer_enabled[{6, 5}] == 1) && (fake::G.peer_enabled[{5, 6}] == 1));
        CHECK(fake::G.peer_enabled.size() == 4);
        CHECK(count_log("5:tvl1_create") == 1 && count_log("3:tvl1_create") == 1 && count_log("6:tvl1_create") == 1);
        CHECK(fake::G.live_handles == 3 && fake::G.live_streams == 6 && fake::G.live_events == 18);
        CHECK(fake::t_dev == -1);   // the calling thread's device is never touched

        // 2. data path under randomised stream scheduling: shard sizes that need 1..4 chunks, ragged last chunk, a batch
        //    smaller than the device count (an idle worker), repeated calls through the same persistent threads
        int call = 0;
        for (int chunk : {1, 2, 3, 16})
            for (int n : {1, 2, 7, 10, 23}) {
                fake::G.rng.seed(1000 + call);
                m.set_chunk(chunk);
                Batch B(n, 5, 9, 77 + call);
                CHECK(m.calc_batch(n, B.a.data(), B.b.data(), B.f.data()) == MI_OK);
                for (int i = 0; i < n; ++i) CHECK(B.correct(i));
                ++call;
            }
        // the shards: worker i gets pairs [i * ceil(n / G), ...) -- 10 pairs over 3 devices = 4 + 4 + 2
        {
            Batch B(10, 3, 4, 5);
            CHECK(m.calc_batch(10, B.a.data(), B.b.data(), B.f.data()) == MI_OK);
            CHECK(m.worker(0).first == 0 && m.worker(0).count == 4 && m.worker(1).first == 4 && m.worker(1).count == 4 &&
                  m.worker(2).first == 8 && m.worker(2).count == 2);
        }

        // 3. up-front validation: nothing is enqueued for a batch that would fail halfway
        {
            Batch B(6, 3, 4, 9);
            const size_t before = fake::G.log.size();
            B.f[5].type = MI_32FC1;
            CHECK(m.calc_batch(6, B.a.data(), B.b.data(), B.f.data()) == MI_ERR_BAD_TYPE);
            B.f[5].type = MI_32FC2;
            B.f[4].step = 4 * 8 - 1;   // smaller than a row: would be used as the pitch of a peer copy
            CHECK(m.calc_batch(6, B.a.data(), B.b.data(), B.f.data()) == MI_ERR_BAD_SIZE);
            B.f[4].step = B.out_step;
            B.b[3].cols = 5;
            CHECK(m.calc_batch(6, B.a.data(), B.b.data(), B.f.data()) == MI_ERR_BAD_SIZE);
            CHECK(fake::G.log.size() == before);
        }

        // 4. a failing worker: the error names its device, BOTH its streams are drained before the call returns, the other
        //    workers finish their shards, and the machine stays usable
        for (const char *op : {"tvl1_calc_batch", "copy2d_async", "stream_wait_event", "dev_malloc"}) {
            const bool is_malloc = std::string(op) == "dev_malloc";
            if (is_malloc) m.set_chunk(4); else m.set_chunk(2);   // a larger chunk forces new staging planes
            Batch B(12, 4, 6, 31);
            fake::G.fail_op = op; fake::G.fail_dev = 6; fake::G.fail_after = 1;
            const size_t before = fake::G.log.size();
            const int rc = m.calc_batch(12, B.a.data(), B.b.data(), B.f.data());
            CHECK(rc == MI_ERR_HIP);
            CHECK(m.error().find("device 6") != std::string::npos && m.error().find("injected failure") != std::string::npos);
            int syncs6 = 0;
            for (size_t i = before; i < fake::G.log.size(); ++i) syncs6 += (fake::G.log[i] == "6:stream_sync");
            CHECK(syncs6 == 2);
            for (int i = 0; i < 8; ++i) CHECK(B.correct(i));   // devices 5 and 3
            Batch C(12, 4, 6, 32);
            CHECK(m.calc_batch(12, C.a.data(), C.b.data(), C.f.data()) == MI_OK);
            for (int i = 0; i < 12; ++i) CHECK(C.correct(i));
        }
    }
    // 5. destruction released everything, each object on its own device (checked inside the fake)
    CHECK(fake::G.live_handles == 0 && fake::G.live_streams == 0 && fake::G.live_events == 0 && fake::G.live_allocs == 0);

    // 5b. the RCCL transport (round 5): one link per non-root worker on a DIFFERENT device, every plane of a chunk and direction inside
    //     one bracket (= one ncclGroup), dense planes as messages whose receiving half runs on the root-side stream for the flows --
    //     so the results are only right if the root-side stream is drained after the worker's streams; pitched planes as peer copies;
    //     a worker on the root's own device keeps its peer copies; failures inside a bracket close it and drain everything
    {
        fake::G.links_enabled = true;
        M m;
        CHECK(m.init(P, {5, 3, 6, 5}) == MI_OK);   // the fourth worker sits on the root's device
        CHECK(m.link_count() == 2 && fake::G.live_links == 2 && count_log("3:link_create") == 1 && count_log("6:link_create") == 1);
        CHECK(fake::G.live_streams == 8 + 2);      // two streams per worker + one root-side stream per link
        int call = 0;
        for (int chunk : {1, 3, 16})
            for (int n : {2, 9, 23}) {
                fake::G.rng.seed(2000 + call);
                m.set_chunk(chunk);
                // dense inputs and flows: messages
                Batch B(n, 5, 9, 177 + call, true);
                const long g0 = fake::G.groups, m0 = fake::G.messages, p0 = fake::G.pitched_planes;
                CHECK(m.calc_batch(n, B.a.data(), B.b.data(), B.f.data()) == MI_OK);
                CHECK(all_streams_idle());   // including the root-side halves of the flows' messages
                for (int i = 0; i < n; ++i) CHECK(B.correct(i));
                const int per = (n + 3) / 4, linked = std::max(0, std::min(per, n - per)) + std::max(0, std::min(per, n - 2 * per));   // pairs of workers 1 and 2
                CHECK(fake::G.messages - m0 == 3L * linked && fake::G.pitched_planes == p0);
                long want_groups = 0;
                for (int wk = 1; wk <= 2; ++wk) { const int cnt = std::max(0, std::min(per, n - wk * per)); want_groups += 2L * ((cnt + chunk - 1) / chunk); }
                CHECK(fake::G.groups - g0 == want_groups);
                // pitched matrices through the same links: peer copies inside the brackets
                Batch C(n, 5, 9, 277 + call);
                CHECK(m.calc_batch(n, C.a.data(), C.b.data(), C.f.data()) == MI_OK);
                for (int i = 0; i < n; ++i) CHECK(C.correct(i));
                ++call;
            }
        for (const char *op : {"link_plane", "link_begin"}) {
            m.set_chunk(2);
            Batch B(16, 4, 6, 41, true);
            fake::G.fail_op = op; fake::G.fail_dev = 6; fake::G.fail_after = 2;
            const size_t before = fake::G.log.size();
            CHECK(m.calc_batch(16, B.a.data(), B.b.data(), B.f.data()) == MI_ERR_HIP);
            CHECK(m.error().find("device 6") != std::string::npos);
            int syncs6 = 0, lsync6 = 0;
            for (size_t i = before; i < fake::G.log.size(); ++i) { syncs6 += (fake::G.log[i] == "6:stream_sync"); lsync6 += (fake::G.log[i] == "6:link_sync"); }
            CHECK(syncs6 == 2 && lsync6 == 1 && all_streams_idle());
            for (int i = 0; i < 8; ++i) CHECK(B.correct(i));      // workers 0 and 1 (devices 5 and 3)
            for (int i = 12; i < 16; ++i) CHECK(B.correct(i));    // worker 3 (device 5 again)
            Batch C(16, 4, 6, 42, true);
            CHECK(m.calc_batch(16, C.a.data(), C.b.data(), C.f.data()) == MI_OK);
            for (int i = 0; i < 16; ++i) CHECK(C.correct(i));
        }
    }
    CHECK(fake::G.live_handles == 0 && fake::G.live_streams == 0 && fake::G.live_events == 0 && fake::G.live_allocs == 0 && fake::G.live_links == 0);
    // 5c. a full node: eight workers, SEVEN links that all contain the root device.  The worker threads run
    //     concurrently; the brackets (the real backend's ncclGroupStart .. ncclGroupEnd) must never overlap -- the fake checks that in
    //     link_begin -- while the copies themselves still interleave on the per-link streams.  Ragged shards, several chunk sizes.
    {
        fake::G.max_open_brackets = 0;
        M m;
        CHECK(m.init(P, {5, 0, 1, 2, 3, 4, 6, 7}) == MI_OK);
        CHECK(m.link_count() == 7 && fake::G.live_links == 7);
        int call = 0;
        for (int chunk : {1, 2, 16})
            for (int n : {8, 37, 64}) {
                fake::G.rng.seed(3000 + call);
                m.set_chunk(chunk);
                Batch B(n, 5, 9, 400 + call, true);
                const long g0 = fake::G.groups;
                CHECK(m.calc_batch(n, B.a.data(), B.b.data(), B.f.data()) == MI_OK);
                CHECK(all_streams_idle());
                for (int i = 0; i < n; ++i) CHECK(B.correct(i));
                CHECK(fake::G.groups > g0);
                ++call;
            }
        CHECK(fake::G.max_open_brackets == 1 && fake::G.open_brackets == 0);
        // a failure inside one worker's bracket releases the serialisation: the other six links finish, the machine stays usable
        m.set_chunk(2);
        Batch B(64, 4, 6, 51, true);
        fake::G.fail_op = "link_plane"; fake::G.fail_dev = 3; fake::G.fail_after = 1;
        CHECK(m.calc_batch(64, B.a.data(), B.b.data(), B.f.data()) == MI_ERR_HIP);
        CHECK(m.error().find("device 3") != std::string::npos && fake::G.open_brackets == 0 && all_streams_idle());
        Batch C(64, 4, 6, 52, true);
        CHECK(m.calc_batch(64, C.a.data(), C.b.data(), C.f.data()) == MI_OK);
        for (int i = 0; i < 64; ++i) CHECK(C.correct(i));
    }
    CHECK(fake::G.live_handles == 0 && fake::G.live_streams == 0 && fake::G.live_events == 0 && fake::G.live_allocs == 0 && fake::G.live_links == 0);
    {   // a failing link_create is a failing worker initialisation
        M m;
        fake::G.fail_op = "link_create"; fake::G.fail_dev = 6; fake::G.fail_after = 0;
        CHECK(m.init(P, {5, 3, 6}) == MI_ERR_HIP);
        CHECK(m.error().find("device 6") != std::string::npos);
    }
    CHECK(fake::G.live_handles == 0 && fake::G.live_streams == 0 && fake::G.live_links == 0);
    fake::G.links_enabled = false;

    // 6. construction failures: a device without a peer link, and a failure in the middle of a later worker's initialisation --
    //    init reports the device and leaves nothing behind
    {
        M m;
        CHECK(m.init(P, {5, 9}) == MI_ERR_HIP);
        CHECK(m.error().find("device 9") != std::string::npos && m.error().find("peer-to-peer") != std::string::npos);
    }
    CHECK(fake::G.live_handles == 0 && fake::G.live_streams == 0 && fake::G.live_events == 0);
    {
        M m;
        fake::G.fail_op = "stream_create"; fake::G.fail_dev = 6; fake::G.fail_after = 1;
        CHECK(m.init(P, {3, 5, 6}) == MI_ERR_HIP);
        CHECK(m.error().find("device 6") != std::string::npos);
    }
    CHECK(fake::G.live_handles == 0 && fake::G.live_streams == 0 && fake::G.live_events == 0 && fake::G.live_allocs == 0);
    // a single device = the root computing in place
    {
        M m;
        CHECK(m.init(P, {6}) == MI_OK);
        Batch B(5, 4, 4, 3);
        CHECK(m.calc_batch(5, B.a.data(), B.b.data(), B.f.data()) == MI_OK);
        for (int i = 0; i < 5; ++i) CHECK(B.correct(i));
        CHECK(fake::G.live_allocs == 0);   // no staging planes on the root
    }
    printf("multi_sm_test: ok\n");
    return 0;
}
