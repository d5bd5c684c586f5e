// mbk_feeder.h -- the tile worker's protocol loop in native code (host only; no HIP in this file).
//
// What it replaces: the worker loop of the reference, DistributedMandelbrotWorkerCUDA.py:111-184 ("W.py"):
//     lease   connect, send 0x00, read 0x10 + level,mrd,indexReal,indexImag (4 x u32 LE) | 0x11   W.py:115-134
//     compute process_workload(level, mrd, indexReal, indexImag) -> 16 777 216 bytes              W.py:142
//     return  connect, send 0x01 + the four u32, read 0x20 | 0x21, on 0x20 send the raw bytes     W.py:148-172
// against Distributer.cs:226-297 (accept loop), :358-392 (request), :397-458 (response).  The bytes on the wire are
// the reference's; only the timing of the two exchanges of DIFFERENT tiles overlaps (any connection may return any
// leased tile), exactly like distributedmandelbrot_amd/worker.py: run_pipelined, whose structure this
// is:
//     this thread    lease tile n+1 | backend: tile n on slot n % k (its D2H overlaps the kernels of the other slots; k = 4
//                    for mbk_worker_run, MBK_NET_FEEDER_SLOTS for mbk_feeder_run)
//     sender threads tile n-1 ... on their own connections; `senders + k` result buffers circulate, so a slow server
//                    back-pressures the lease rate instead of growing a queue
// Tiles the backend reports as uniform (all 0 = "Never", all 1 = "Immediate": DataChunk.cs:82,87) were not copied
// off the GPU (MBK_LAZY_UNIFORM); their 16 MiB payload is one of two shared constant buffers.
// Why native: at ~4 000 tiles/s per context the Python loop's per-tile interpreter work and the GIL
// hand-offs between its lease, wait and sender threads were the bound (647 tiles/s on loopback).
//
// Against the REAL server (round 4).  The reference Distributer accepts on ONE thread, one connection at a time, with
// `listenBacklog = 16` (Distributer.cs:16,221,226-297), reads with 100 ms receive timeouts (:17,196-202) and needs >= 10 ms
// per 16 MiB payload; the reference worker opened one connection at a time.  A farm of 8 feeders x (4 senders + 1
// lease connection) would present it with up to 40 concurrent connects: on a Windows/.NET host a full backlog answers
// with RST, on Linux the SYN is dropped and retransmitted after 1 s.  So, process-wide (every feeder of a farm shares
// them; mbk_net_set_option):
//   * at most MBK_NET_MAX_CONNECTIONS (default 8 < 16) connections are open or being opened at any time;
//   * connects and socket reads / writes have timeouts (a blackholed address or a server that accepts and never
//     answers cannot hang the call);
//   * both exchanges are retried with exponential backoff on transient failures (refused, reset, timed out, closed
//     before the reply) as long as the server has not answered -- a computed tile is never dropped on the first failed
//     connect.  After 0x20 the payload is sent once: the server removed the lease when it accepted (Distributer.cs:404-423).
//   * MBK_NET_STOP asks every running loop to stop leasing and drain.
#pragma once

#include <arpa/inet.h>
#include <fcntl.h>
#include <netdb.h>
#include <poll.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mbkf {

constexpr uint8_t kRequest = 0x00, kResponse = 0x01;          // Distributer.cs:30-31, W.py:10-11
constexpr uint8_t kAvailable = 0x10, kNotAvailable = 0x11;    // Distributer.cs:35-38
constexpr uint8_t kAccept = 0x20, kReject = 0x21;             // Distributer.cs:42-45
constexpr size_t kChunkBytes = (size_t)MBK_CHUNK_BYTES;

struct Tile {
    uint32_t w[4];   // level, mrd, indexReal, indexImag
    uint8_t *buf;
    mbk_stats st;
};

template <typename T>
class Channel {   // unbounded MPMC queue; the buffer pool bounds what is in flight
public:
    void put(T v)
    {
        {
            std::lock_guard<std::mutex> g(m_);
            q_.push_back(v);
        }
        cv_.notify_one();
    }
    T get()
    {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return !q_.empty(); });
        T v = q_.front();
        q_.pop_front();
        return v;
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<T> q_;
};

// ---- process-wide network settings (mbk_net_set_option) and the connection gate -------------------------------
struct NetConfig {
    std::atomic<uint32_t> max_connections{8};       // < the reference's listenBacklog of 16 (Distributer.cs:16)
    std::atomic<uint32_t> connect_timeout_ms{10000};
    std::atomic<uint32_t> io_timeout_ms{30000};     // per send / recv call without progress
    std::atomic<uint32_t> retries{6};               // attempts after the first, per exchange
    std::atomic<uint32_t> backoff_ms{50};           // first pause; doubles per attempt, capped at 2 s, + jitter
    std::atomic<uint32_t> stop{0};
    std::atomic<uint32_t> feeder_slots{2};          // mbk_feeder_run: tiles on the backend at once
    std::mutex m;
    std::condition_variable cv;
    uint32_t open = 0;                              // connections open or being opened, all feeders of the process
    uint32_t peak = 0;                              // the most that were ever open at once (diagnostic; MBK_NET_PEAK_CONNECTIONS)
};
static inline NetConfig &net()
{
    static NetConfig c;
    return c;
}
struct ConnSlot {   // RAII: one of the process's max_connections
    ConnSlot()
    {
        NetConfig &c = net();
        std::unique_lock<std::mutex> g(c.m);
        c.cv.wait(g, [&] { return c.open < std::max<uint32_t>(1u, c.max_connections.load()); });
        ++c.open;
        if (c.open > c.peak) c.peak = c.open;
    }
    ~ConnSlot()
    {
        NetConfig &c = net();
        {
            std::lock_guard<std::mutex> g(c.m);
            --c.open;
        }
        c.cv.notify_one();
    }
};

static inline std::string errno_text(const char *what, int e)
{
    return std::string(what) + ": " + (e == 0 ? "connection closed by peer" : std::strerror(e));
}

// A failure that says "not now" rather than "never": the server's backlog was full (RST / refused), it was busy past
// a timeout, or it closed before answering.  0 = orderly close by the peer before the expected byte.
static inline bool transient_errno(int e)
{
    return e == 0 || e == ECONNREFUSED || e == ECONNRESET || e == ECONNABORTED || e == ETIMEDOUT || e == EAGAIN ||
           e == EWOULDBLOCK || e == EPIPE || e == EHOSTUNREACH || e == ENETUNREACH || e == EINPROGRESS || e == EINTR;
}

static inline void backoff_sleep(uint32_t attempt)
{
    static thread_local uint32_t rng = (uint32_t)std::hash<std::thread::id>()(std::this_thread::get_id()) | 1u;
    rng ^= rng << 13;
    rng ^= rng >> 17;
    rng ^= rng << 5;
    uint64_t ms = (uint64_t)std::max<uint32_t>(1u, net().backoff_ms.load()) << std::min<uint32_t>(attempt, 6u);
    if (ms > 2000) ms = 2000;
    ms += rng % (ms / 2 + 1);
    std::this_thread::sleep_for(std::chrono::milliseconds(ms));
}

// Connect with a timeout; the socket comes back blocking, with send / receive timeouts.  *e = errno of the failure.
static int connect_to(const char *addr, uint16_t port, std::string *err, int *e)
{
    addrinfo hints;
    std::memset(&hints, 0, sizeof(hints));
    hints.ai_family = AF_UNSPEC;
    hints.ai_socktype = SOCK_STREAM;
    addrinfo *res = nullptr;
    const std::string service = std::to_string(port);
    const int g = getaddrinfo(addr, service.c_str(), &hints, &res);
    if (g != 0) {
        *err = std::string("getaddrinfo(") + addr + "): " + gai_strerror(g);
        *e = EINVAL;   // not transient
        return -1;
    }
    const int timeout_ms = (int)net().connect_timeout_ms.load();
    int fd = -1;
    *e = EADDRNOTAVAIL;
    for (addrinfo *a = res; a; a = a->ai_next) {
        fd = socket(a->ai_family, a->ai_socktype, a->ai_protocol);
        if (fd < 0) continue;
        const int fl = fcntl(fd, F_GETFL, 0);
        (void)fcntl(fd, F_SETFL, fl | O_NONBLOCK);
        int rc = connect(fd, a->ai_addr, a->ai_addrlen);
        int ce = rc == 0 ? 0 : errno;
        if (rc != 0 && (ce == EINPROGRESS || ce == EINTR)) {
            pollfd pf = {fd, POLLOUT, 0};
            int pr;
            do pr = poll(&pf, 1, timeout_ms > 0 ? timeout_ms : -1);
            while (pr < 0 && errno == EINTR);
            if (pr == 0) {
                ce = ETIMEDOUT;
            } else if (pr < 0) {
                ce = errno;
            } else {
                socklen_t len = sizeof(ce);
                if (getsockopt(fd, SOL_SOCKET, SO_ERROR, &ce, &len) != 0) ce = errno;
            }
        }
        if (ce == 0) {
            (void)fcntl(fd, F_SETFL, fl);
            break;
        }
        *e = ce;
        *err = errno_text("connect", ce);
        close(fd);
        fd = -1;
    }
    freeaddrinfo(res);
    if (fd < 0) {
        if (err->empty()) *err = "connect: no usable address";
        return -1;
    }
    err->clear();   // (an earlier address of the list may have failed: that text must not outlive the success)
    *e = 0;
    const int one = 1, sndbuf = 4 << 20;   // a 16 MiB payload follows: fewer, larger sends
    (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    (void)setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &sndbuf, sizeof(sndbuf));
    const uint32_t io_ms = net().io_timeout_ms.load();
    if (io_ms > 0) {
        timeval tv;
        tv.tv_sec = io_ms / 1000;
        tv.tv_usec = (io_ms % 1000) * 1000;
        (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        (void)setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
    }
    return fd;
}

// *e: errno of the failure (a send / receive timeout is EAGAIN)
static bool send_all(int fd, const uint8_t *p, size_t n, size_t *sent, int *e)
{
    size_t done = 0;
    while (done < n) {
        const ssize_t k = send(fd, p + done, n - done, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            *e = errno;
            if (sent) *sent = done;
            return false;
        }
        done += (size_t)k;
    }
    if (sent) *sent = done;
    return true;
}

// *e: errno, or 0 when the peer closed the connection in an orderly way before n bytes had arrived
static bool recv_exact(int fd, uint8_t *p, size_t n, int *e)
{
    size_t done = 0;
    while (done < n) {
        const ssize_t k = recv(fd, p + done, n - done, 0);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) {
            *e = k == 0 ? 0 : errno;
            return false;
        }
        done += (size_t)k;
    }
    return true;
}

// First connection of W.py:115-134.  1 = a workload, 0 = 0x11 (none available), -1 = error.  Transient failures are
// retried with backoff while the hand-out code cannot have run: a refused / reset / timed-out CONNECT, a failed send, a
// close or reset before the reply byte.  A reply that times out after the request was sent is NOT retried (a lease the
// server registered for a reply nobody read is lost to everybody for its hour -- Distributer.cs:22).
static int lease(const char *addr, uint16_t port, uint32_t w[4], std::string *err, uint64_t *retried)
{
    const uint32_t attempts = 1u + net().retries.load();
    for (uint32_t attempt = 0; attempt < attempts; ++attempt) {
        if (attempt > 0) {
            backoff_sleep(attempt - 1);
            if (retried) ++*retried;
        }
        int e = 0, rc = -1;
        bool again = false;
        {
            ConnSlot slot_guard;
            const int fd = connect_to(addr, port, err, &e);
            if (fd < 0) {
                again = transient_errno(e) && e != 0;
            } else {
                uint8_t op = kRequest, reply = 0, raw[16];
                if (!send_all(fd, &op, 1, nullptr, &e)) {
                    *err = errno_text("workload request", e);
                    again = transient_errno(e);
                } else if (!recv_exact(fd, &reply, 1, &e)) {
                    // The request is out.  An orderly close or a reset before the reply byte never reached the hand-out
                    // code: ask again.  A TIMEOUT is different: the server registers the lease as it sends 0x10 + the
                    // workload (Distributer.cs HandleWorkloadRequest), so a reply that was sent but not read in time
                    // would orphan that tile for its hour and the retry would take a second one: an error.
                    *err = errno_text("workload request (no reply)", e);
                    again = e == 0 || e == ECONNRESET || e == ECONNABORTED || e == EPIPE;
                } else if (reply == kNotAvailable) {
                    rc = 0;
                } else if (reply != kAvailable) {
                    *err = "Unknown response code to request: " + std::to_string((int)reply);   // W.py:131-132
                } else if (!recv_exact(fd, raw, 16, &e)) {   // four 4-byte sends on the server side (DistributerWorkload.cs:53-77)
                    *err = errno_text("workload fields", e);   // the lease exists on the server: do not ask for another
                } else {
                    for (int k = 0; k < 4; ++k)
                        w[k] = (uint32_t)raw[4 * k] | ((uint32_t)raw[4 * k + 1] << 8) | ((uint32_t)raw[4 * k + 2] << 16) | ((uint32_t)raw[4 * k + 3] << 24);
                    rc = 1;
                }
                close(fd);
            }
        }
        if (!again) return rc;
    }
    return -1;   // *err holds the last failure
}

enum { kSubmitRejected = 0, kSubmitAccepted = 1, kSubmitReset = 2, kSubmitError = -1 };

// Second connection of W.py:148-172.  Until the server has answered 0x20 / 0x21 a transient failure is retried with
// backoff (the lease is still there: Distributer.cs:404 matches and removes it only when it answers 0x20).
static int give_back(const char *addr, uint16_t port, const uint32_t w[4], const uint8_t *payload, std::string *err,
                     uint64_t *retried)
{
    // opcode + header in ONE segment: the server reads them with separate 100 ms-timeout receives
    // (Distributer.cs:17,243-245,400), so do not dribble them
    uint8_t head[17];
    head[0] = kResponse;
    for (int k = 0; k < 4; ++k)
        for (int b = 0; b < 4; ++b) head[1 + 4 * k + b] = (uint8_t)(w[k] >> (8 * b));
    const uint32_t attempts = 1u + net().retries.load();
    for (uint32_t attempt = 0; attempt < attempts; ++attempt) {
        if (attempt > 0) {
            backoff_sleep(attempt - 1);
            if (retried) ++*retried;
        }
        int e = 0, rc = kSubmitError;
        bool again = false;
        {
            ConnSlot slot_guard;
            const int fd = connect_to(addr, port, err, &e);
            if (fd < 0) {
                again = transient_errno(e) && e != 0;
            } else {
                uint8_t reply = 0;
                if (!send_all(fd, head, sizeof(head), nullptr, &e) || !recv_exact(fd, &reply, 1, &e)) {
                    *err = errno_text("workload response", e);
                    again = transient_errno(e);
                } else if (reply == kReject) {
                    rc = kSubmitRejected;
                } else if (reply != kAccept) {
                    *err = "Unknown response code to request: " + std::to_string((int)reply);   // W.py:165-166
                } else {
                    size_t sent = 0;
                    if (send_all(fd, payload, kChunkBytes, &sent, &e)) {   // exactly 16 777 216 raw bytes, no header (W.py:168)
                        rc = kSubmitAccepted;
                    } else if (e == ECONNRESET || e == EPIPE) {
                        // the reference server reads the payload with ONE Socket.Receive and closes (Distributer.cs:416-423):
                        // with unread bytes in flight that close is a reset, after the tile was marked complete
                        rc = kSubmitReset;
                    } else {
                        *err = errno_text("payload send", e);
                    }
                }
                close(fd);
            }
        }
        if (!again) return rc;
    }
    return kSubmitError;
}

// The loop.  `ops` is the compute backend (mbk_worker_run binds it to a GPU context).
// nslots: tiles on the backend at once (slot numbers 0 .. nslots-1, in turn)
static int run(const mbk_feeder_ops *ops, const char *addr, uint16_t port, uint64_t max_tiles, uint32_t senders,
               mbk_worker_report *rep, std::string *err, uint32_t nslots = 2)
{
    if (senders == 0) senders = 1;
    if (senders > 64) senders = 64;
    nslots = std::min<uint32_t>(std::max<uint32_t>(nslots, 1u), 8u);
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t nbuf = senders + nslots;
    std::vector<uint8_t *> bufs;
    for (uint32_t k = 0; k < nbuf; ++k) {
        uint8_t *b = (uint8_t *)ops->alloc(ops->user, kChunkBytes);
        if (!b) {
            for (uint8_t *x : bufs) ops->release(ops->user, x);
            *err = "feeder: could not allocate the result buffers";
            return MBK_ERR_NOMEM;
        }
        bufs.push_back(b);
    }
    // the payload of a uniform tile: one shared buffer per constant, filled on first use
    std::vector<uint8_t> constant[2];
    std::mutex constant_lock;
    Channel<uint8_t *> free_bufs;
    for (uint8_t *b : bufs) free_bufs.put(b);
    Channel<Tile *> outbox;
    std::mutex rep_lock;
    std::string sender_err;
    mbk_worker_report r;
    std::memset(&r, 0, sizeof(r));

    auto sender = [&]() {
        for (;;) {
            Tile *t = outbox.get();
            if (!t) return;
            const uint8_t *payload = t->buf;
            const int uniform = t->st.all_bytes_zero ? 0 : t->st.all_bytes_one ? 1 : -1;
            if (uniform >= 0) {
                std::lock_guard<std::mutex> g(constant_lock);
                if (constant[uniform].empty()) constant[uniform].assign(kChunkBytes, (uint8_t)uniform);
                payload = constant[uniform].data();
            }
            std::string e;
            uint64_t retried = 0;
            const int status = give_back(addr, port, t->w, payload, &e, &retried);
            {
                std::lock_guard<std::mutex> g(rep_lock);
                r.net_retries += retried;
                if (status == kSubmitAccepted) ++r.accepted;
                else if (status == kSubmitRejected) ++r.rejected;   // the tile is dropped, carry on (W.py:161-163)
                else if (status == kSubmitReset) ++r.resets;
                else if (sender_err.empty()) sender_err = e;
                if (uniform >= 0) ++r.uniform_tiles;
                r.pixel_iterations += t->st.pixel_iterations;
                r.kernel_ms_sum += t->st.kernel_ms;
            }
            if (ops->on_tile) ops->on_tile(ops->user, t->w, &t->st, status);
            free_bufs.put(t->buf);
            delete t;
        }
    };
    std::vector<std::thread> threads;
    for (uint32_t k = 0; k < senders; ++k) threads.emplace_back(sender);

    Tile *inflight[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = MBK_OK;
    bool more = true;
    int slot = 0;
    auto any_inflight = [&]() {
        for (uint32_t k = 0; k < nslots; ++k)
            if (inflight[k]) return true;
        return false;
    };
    while (more || any_inflight()) {
        uint32_t w[4];
        bool have = false;
        {
            std::lock_guard<std::mutex> g(rep_lock);
            if (!sender_err.empty()) more = false;
        }
        if (more && rc == MBK_OK && (max_tiles == 0 || r.leased < max_tiles) && net().stop.load() == 0) {
            uint64_t retried = 0;
            const int l = lease(addr, port, w, err, &retried);
            if (retried) {
                std::lock_guard<std::mutex> g(rep_lock);
                r.net_retries += retried;
            }
            if (l < 0) {          // stop leasing, but finish (wait for + send) the tiles already leased
                rc = MBK_ERR_NET;
                more = false;
            } else if (l == 0) {  // "No workload was available, ending program" (W.py:127-129)
                more = false;
            } else {
                have = true;
            }
        } else {
            more = false;
        }
        if (inflight[slot]) {   // retire the tile that occupies the slot we are about to reuse (or drain)
            Tile *t = inflight[slot];
            inflight[slot] = nullptr;
            const int wr = ops->wait(ops->user, slot, &t->st);
            if (wr != MBK_OK) {
                if (rc == MBK_OK) {
                    rc = wr;
                    *err = "feeder: backend wait failed";
                }
                more = false;
                free_bufs.put(t->buf);
                delete t;
            } else {
                outbox.put(t);
            }
        }
        if (have) {
            Tile *t = new Tile();
            std::memcpy(t->w, w, sizeof(w));
            t->buf = free_bufs.get();   // blocks while every buffer is with a sender: back-pressure
            const int sr = ops->submit(ops->user, slot, w[0], w[1], w[2], w[3], t->buf);
            {
                std::lock_guard<std::mutex> g(rep_lock);
                ++r.leased;
            }
            if (sr != MBK_OK) {   // e.g. level 0 / index >= level from a broken server: drop the tile, stop
                if (rc == MBK_OK) {
                    rc = sr;
                    *err = "feeder: backend submit failed";
                }
                more = false;
                free_bufs.put(t->buf);
                delete t;
            } else {
                inflight[slot] = t;
            }
        }
        slot = (slot + 1) % (int)nslots;
    }
    for (size_t k = 0; k < threads.size(); ++k) outbox.put(nullptr);
    for (std::thread &t : threads) t.join();
    for (uint8_t *b : bufs) ops->release(ops->user, b);
    if (rc == MBK_OK && !sender_err.empty()) {
        rc = MBK_ERR_NET;
        *err = sender_err;
    }
    r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rep) *rep = r;
    return rc;
}

}  // namespace mbkf
