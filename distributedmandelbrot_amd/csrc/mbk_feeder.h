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
//     this thread    lease tile n+1 | backend: tile n on slot n % 2 (its D2H overlaps the kernel of the other slot)
//     sender threads tile n-1 ... on their own connections; `senders + 2` result buffers circulate, so a slow server
//                    back-pressures the lease rate instead of growing a queue
// Tiles the backend reports as uniform (all 0 = "Never", all 1 = "Immediate": DataChunk.cs:82,87) were not copied
// off the GPU (MBK_LAZY_UNIFORM); their 16 MiB payload is one of two shared constant buffers.
// Why native: at ~4 000 tiles/s per context the Python loop's per-tile interpreter work and the GIL
// hand-offs between its lease, wait and sender threads were the bound (647 tiles/s on loopback).
#pragma once

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

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

static inline std::string errno_text(const char *what)
{
    return std::string(what) + ": " + std::strerror(errno);
}

static int connect_to(const char *addr, uint16_t port, std::string *err)
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
        return -1;
    }
    int fd = -1;
    for (addrinfo *a = res; a; a = a->ai_next) {
        fd = socket(a->ai_family, a->ai_socktype, a->ai_protocol);
        if (fd < 0) continue;
        if (connect(fd, a->ai_addr, a->ai_addrlen) == 0) break;
        *err = errno_text("connect");
        close(fd);
        fd = -1;
    }
    freeaddrinfo(res);
    if (fd < 0) {
        if (err->empty()) *err = "connect: no usable address";
        return -1;
    }
    const int one = 1, sndbuf = 4 << 20;   // a 16 MiB payload follows: fewer, larger sends
    (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    (void)setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &sndbuf, sizeof(sndbuf));
    return fd;
}

static bool send_all(int fd, const uint8_t *p, size_t n, size_t *sent)
{
    size_t done = 0;
    while (done < n) {
        const ssize_t k = send(fd, p + done, n - done, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            if (sent) *sent = done;
            return false;
        }
        done += (size_t)k;
    }
    if (sent) *sent = done;
    return true;
}

static bool recv_exact(int fd, uint8_t *p, size_t n)
{
    size_t done = 0;
    while (done < n) {
        const ssize_t k = recv(fd, p + done, n - done, 0);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return false;
        done += (size_t)k;
    }
    return true;
}

// First connection of W.py:115-134.  1 = a workload, 0 = 0x11 (none available), -1 = error.
static int lease(const char *addr, uint16_t port, uint32_t w[4], std::string *err)
{
    const int fd = connect_to(addr, port, err);
    if (fd < 0) return -1;
    int rc = -1;
    uint8_t op = kRequest, reply = 0, raw[16];
    if (!send_all(fd, &op, 1, nullptr) || !recv_exact(fd, &reply, 1)) {
        *err = errno_text("workload request");
    } else if (reply == kNotAvailable) {
        rc = 0;
    } else if (reply != kAvailable) {
        *err = "Unknown response code to request: " + std::to_string((int)reply);   // W.py:131-132
    } else if (!recv_exact(fd, raw, 16)) {   // four 4-byte sends on the server side (DistributerWorkload.cs:53-77)
        *err = errno_text("workload fields");
    } else {
        for (int k = 0; k < 4; ++k)
            w[k] = (uint32_t)raw[4 * k] | ((uint32_t)raw[4 * k + 1] << 8) | ((uint32_t)raw[4 * k + 2] << 16) | ((uint32_t)raw[4 * k + 3] << 24);
        rc = 1;
    }
    close(fd);
    return rc;
}

enum { kSubmitRejected = 0, kSubmitAccepted = 1, kSubmitReset = 2, kSubmitError = -1 };

// Second connection of W.py:148-172.
static int give_back(const char *addr, uint16_t port, const uint32_t w[4], const uint8_t *payload, std::string *err)
{
    const int fd = connect_to(addr, port, err);
    if (fd < 0) return kSubmitError;
    // opcode + header in ONE segment: the server reads them with separate 100 ms-timeout receives
    // (Distributer.cs:17,243-245,400), so do not dribble them
    uint8_t head[17];
    head[0] = kResponse;
    for (int k = 0; k < 4; ++k)
        for (int b = 0; b < 4; ++b) head[1 + 4 * k + b] = (uint8_t)(w[k] >> (8 * b));
    uint8_t reply = 0;
    int rc = kSubmitError;
    if (!send_all(fd, head, sizeof(head), nullptr) || !recv_exact(fd, &reply, 1)) {
        *err = errno_text("workload response");
    } else if (reply == kReject) {
        rc = kSubmitRejected;
    } else if (reply != kAccept) {
        *err = "Unknown response code to request: " + std::to_string((int)reply);   // W.py:165-166
    } else {
        size_t sent = 0;
        if (send_all(fd, payload, kChunkBytes, &sent)) {   // exactly 16 777 216 raw bytes, no header (W.py:168)
            rc = kSubmitAccepted;
        } else if (errno == ECONNRESET || errno == EPIPE) {
            // the reference server reads the payload with ONE Socket.Receive and closes (Distributer.cs:416-423):
            // with unread bytes in flight that close is a reset, after the tile was marked complete
            rc = kSubmitReset;
        } else {
            *err = errno_text("payload send");
        }
    }
    close(fd);
    return rc;
}

// The loop.  `ops` is the compute backend (mbk_worker_run binds it to a GPU context).
static int run(const mbk_feeder_ops *ops, const char *addr, uint16_t port, uint64_t max_tiles, uint32_t senders,
               mbk_worker_report *rep, std::string *err)
{
    if (senders == 0) senders = 1;
    if (senders > 64) senders = 64;
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t nbuf = senders + 2;
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
            const int status = give_back(addr, port, t->w, payload, &e);
            {
                std::lock_guard<std::mutex> g(rep_lock);
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

    Tile *inflight[2] = {nullptr, nullptr};
    int rc = MBK_OK;
    bool more = true;
    int slot = 0;
    while (more || inflight[0] || inflight[1]) {
        uint32_t w[4];
        bool have = false;
        {
            std::lock_guard<std::mutex> g(rep_lock);
            if (!sender_err.empty()) more = false;
        }
        if (more && rc == MBK_OK && (max_tiles == 0 || r.leased < max_tiles)) {
            const int l = lease(addr, port, w, err);
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
        slot ^= 1;
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
