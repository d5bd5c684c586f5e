// sink_distributer.cpp -- a minimal NATIVE stand-in for the reference's Distributer, for measuring what the worker
// loop can sustain when the server is not the bottleneck (scripts/worker_e2e.py).  MEASUREMENT TOOL, not product code.
//
// Protocol (Distributer.cs:30-45,358-458; DistributerWorkload.cs:53-100), one thread per connection:
//   0x00           -> 0x10 + level,mrd,indexReal,indexImag (4 x u32 LE) for the next tile of ONE level (indexReal-major,
//                     Distributer.cs:335-353) | 0x11 when every tile has been handed out
//   0x01 + 4 x u32 -> 0x20 if that tile is under lease (then read exactly 16 777 216 bytes and drop them) | 0x21
// Prints "PORT <n>" on stdout, then "DONE <tiles> <seconds>" when every tile has arrived, and exits.
//   g++ -O2 -pthread -o sink_distributer sink_distributer.cpp ;  ./sink_distributer LEVEL MRD
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

static const size_t kChunk = 4096u * 4096u;
static uint32_t g_level, g_mrd;
static std::mutex g_lock;
static uint64_t g_next = 0;                       // next tile to hand out
static std::set<uint64_t> g_leased;               // tile ids under lease
static std::atomic<uint64_t> g_done{0};
static uint64_t g_checksum = 0;                   // so that the received bytes are really read

static bool recv_exact(int fd, uint8_t *p, size_t n)
{
    size_t got = 0;
    while (got < n) {
        const ssize_t k = recv(fd, p + got, n - got, 0);
        if (k <= 0) return false;
        got += (size_t)k;
    }
    return true;
}

static void put_u32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t get_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

static void serve(int fd)
{
    static thread_local std::vector<uint8_t> buf(kChunk);
    const int rcv = 4 << 20;
    (void)setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &rcv, sizeof(rcv));
    uint8_t op = 0xff;
    if (recv_exact(fd, &op, 1)) {
        if (op == 0x00) {
            uint64_t id = ~0ull;
            {
                std::lock_guard<std::mutex> g(g_lock);
                if (g_next < (uint64_t)g_level * g_level) {
                    id = g_next++;
                    g_leased.insert(id);
                }
            }
            if (id == ~0ull) {
                const uint8_t none = 0x11;
                (void)send(fd, &none, 1, MSG_NOSIGNAL);
            } else {
                uint8_t msg[17];
                msg[0] = 0x10;
                put_u32(msg + 1, g_level);
                put_u32(msg + 5, g_mrd);
                put_u32(msg + 9, (uint32_t)(id / g_level));
                put_u32(msg + 13, (uint32_t)(id % g_level));
                (void)send(fd, msg, sizeof(msg), MSG_NOSIGNAL);
            }
        } else if (op == 0x01) {
            uint8_t h[16];
            if (recv_exact(fd, h, 16)) {
                const uint32_t level = get_u32(h), mrd = get_u32(h + 4), ir = get_u32(h + 8), ii = get_u32(h + 12);
                bool ok = false;
                if (level == g_level && mrd == g_mrd && ir < g_level && ii < g_level) {
                    std::lock_guard<std::mutex> g(g_lock);
                    ok = g_leased.erase((uint64_t)ir * g_level + ii) == 1;
                }
                const uint8_t reply = ok ? 0x20 : 0x21;
                (void)send(fd, &reply, 1, MSG_NOSIGNAL);
                if (ok && recv_exact(fd, buf.data(), kChunk)) {
                    uint64_t sum = 0;
                    for (size_t k = 0; k < kChunk; k += 4096) sum += buf[k];   // touch every page
                    {
                        std::lock_guard<std::mutex> g(g_lock);
                        g_checksum += sum;
                    }
                    g_done.fetch_add(1);
                }
            }
        }
    }
    close(fd);
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s LEVEL MRD\n", argv[0]);
        return 2;
    }
    g_level = (uint32_t)std::atoi(argv[1]);
    g_mrd = (uint32_t)std::atoi(argv[2]);
    const int ls = socket(AF_INET, SOCK_STREAM, 0);
    const int one = 1;
    (void)setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in a;
    std::memset(&a, 0, sizeof(a));
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    a.sin_port = 0;
    if (bind(ls, (sockaddr *)&a, sizeof(a)) != 0 || listen(ls, 64) != 0) {
        std::perror("bind/listen");
        return 1;
    }
    socklen_t len = sizeof(a);
    getsockname(ls, (sockaddr *)&a, &len);
    std::printf("PORT %d\n", (int)ntohs(a.sin_port));
    std::fflush(stdout);
    const uint64_t total = (uint64_t)g_level * g_level;
    std::chrono::steady_clock::time_point t0;
    bool started = false;
    std::thread([&] {
        for (;;) {
            const int fd = accept(ls, nullptr, nullptr);
            if (fd < 0) return;
            if (!started) {
                started = true;
                t0 = std::chrono::steady_clock::now();
            }
            std::thread(serve, fd).detach();
        }
    }).detach();
    while (g_done.load() < total) usleep(200);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("DONE %llu %.6f checksum %llu\n", (unsigned long long)total, dt, (unsigned long long)g_checksum);
    std::fflush(stdout);
    return 0;
}
