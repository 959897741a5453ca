// Driver of the sanitizer builds of qgs_amd/csrc/host_bridge.cpp (tests/test_host_bridge_sanitizers_cpu.py): the record-window, member-group and
// shard-thread traffic patterns of the library, against the stub runtime of tests/stub_hip (streams = asynchronous worker threads).
// Every scenario checks the bytes that arrive; ThreadSanitizer / AddressSanitizer check how they got there.  Prints "OK <scenarios>".
#include "host_bridge.h"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using qgs::bridge_d2h;
using qgs::bridge_d2h_rows_async;
using qgs::bridge_h2d;

static std::atomic<int> g_fail{0};
#define CHECK(cond)                                                                        \
    do {                                                                                   \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; return; } \
    } while (0)

static unsigned char pattern(size_t i, unsigned seed) { return (unsigned char)((i * 2654435761u + seed * 40503u) >> 7); }

// blocking copies: sizes around the bounce-block and task-grain boundaries
static void roundtrip(int dev, unsigned seed)
{
    hipSetDevice(dev);
    hipStream_t st = nullptr;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const size_t sizes[] = {1, 7, 4096, (1u << 20) - 1, (1u << 20) + 3, (8u << 20), (8u << 20) + 17, (20u << 20) + 5};
    for (size_t n : sizes) {
        std::vector<unsigned char> src(n), back(n, 0);
        for (size_t i = 0; i < n; ++i) src[i] = pattern(i, seed);
        void *d = nullptr;
        hipMalloc(&d, n);
        std::string err;
        CHECK(bridge_h2d(d, src.data(), n, st, &err) == 0);
        CHECK(bridge_d2h(back.data(), d, n, st, &err) == 0);
        CHECK(std::memcmp(src.data(), back.data(), n) == 0);
        hipFree(d);
    }
    std::string err;
    CHECK(bridge_h2d(nullptr, nullptr, 0, st, &err) == 0);
}

// record windows: a "kernel" fills device buffer w % 2 with the records of window w, the window is handed to the drain thread with a
// `ready` event, and buffer w % 2 is overwritten by window w + 2 only after bridge_wait_copied of window w (qgs_hip_api's window loop)
static void windows(int dev, unsigned seed, size_t row_bytes, size_t rows_per_window, int n_windows)
{
    hipSetDevice(dev);
    hipStream_t compute = nullptr;
    hipStreamCreateWithFlags(&compute, hipStreamNonBlocking);
    const size_t win_bytes = row_bytes * rows_per_window, pitch = row_bytes * (size_t)n_windows + 24;      // host rows: window w at column w * row_bytes
    char *buf[2];
    for (auto &b : buf) { void *p; hipMalloc(&p, win_bytes); b = (char *)p; }
    std::vector<char> host(pitch * rows_per_window, 0x5a);
    hipEvent_t ready[2];
    for (auto &e : ready) hipEventCreate(&e);
    int64_t ticket[2] = {0, 0};
    std::vector<int64_t> all;
    std::string err;
    for (int w = 0; w < n_windows; ++w) {
        const int q = w % 2;
        if (ticket[q]) CHECK(qgs::bridge_wait_copied(ticket[q], &err) == 0);
        char *b = buf[q];
        compute->push([=] { for (size_t i = 0; i < win_bytes; ++i) b[i] = (char)pattern(i + (size_t)w * win_bytes, seed); });
        hipEventRecord(ready[q], compute);
        ticket[q] = bridge_d2h_rows_async(host.data() + (size_t)w * row_bytes, pitch, b, row_bytes, rows_per_window, ready[q], &err);
        CHECK(ticket[q] > 0);
        all.push_back(ticket[q]);
        if (w % 3 == 1) (void)qgs::bridge_poll_copied(ticket[q]);
    }
    for (int64_t t : all) CHECK(qgs::bridge_wait_done(t, &err) == 0);
    for (int w = 0; w < n_windows; ++w)
        for (size_t r = 0; r < rows_per_window; ++r)
            for (size_t i = 0; i < row_bytes; i += 97)
                CHECK(host[r * pitch + (size_t)w * row_bytes + i] == (char)pattern(r * row_bytes + i + (size_t)w * win_bytes, seed));
    for (size_t r = 0; r < rows_per_window; ++r) CHECK(host[r * pitch + pitch - 1] == 0x5a);                 // nothing written past a row
    hipStreamSynchronize(compute);
    for (auto &b : buf) hipFree(b);
}

// a failing transfer: the job reports it, the next job of the device works
static void failure(int dev)
{
    hipSetDevice(dev);
    const size_t n = 3u << 20;
    void *d = nullptr;
    hipMalloc(&d, n);
    std::memset(d, 7, n);
    std::vector<char> host(n, 0);
    std::string err;
    {
        std::lock_guard<std::mutex> lock(stub_hip::fail_mu());
        stub_hip::fail_memcpy_in() = 1;
    }
    const int64_t t1 = bridge_d2h_rows_async(host.data(), n, (const char *)d, n, 1, nullptr, &err);
    CHECK(t1 > 0);
    const int rc = qgs::bridge_wait_done(t1, &err);
    CHECK(rc == -1 && !err.empty());
    const int64_t t2 = bridge_d2h_rows_async(host.data(), n, (const char *)d, n, 1, nullptr, &err);
    CHECK(t2 > 0 && qgs::bridge_wait_done(t2, &err) == 0);
    CHECK(host[0] == 7 && host[n - 1] == 7);
    CHECK(qgs::bridge_wait_done(-5, &err) == -1);
    CHECK(bridge_d2h_rows_async(host.data(), n, (const char *)d, 0, 1, nullptr, &err) == -1);
    hipFree(d);
}

int main(int argc, char **argv)
{
    const int shards = argc > 1 ? std::atoi(argv[1]) : 4;
    // one device, one thread
    roundtrip(0, 1);
    windows(0, 2, 40000, 36, 9);                      // rows smaller than a bounce block (several rows per piece)
    windows(0, 3, (size_t)18 << 20, 2, 3);            // a row larger than a bounce block (segments of a row)
    windows(0, 4, 8, 1000, 5);                        // 8-byte runs a page apart: the window-of-records layout
    failure(0);
    // shard threads: every shard its own device, blocking copies and windows at once; device 0 also serves a second thread
    // (the Python binding's to_device / to_host next to a running estimator)
    std::vector<std::thread> th;
    for (int s = 0; s < shards; ++s)
        th.emplace_back([s] {
            roundtrip(s, 10 + s);
            windows(s, 20 + s, 60000 + 8 * s, 24, 7);
            for (int k = 0; k < 50; ++k) windows(s, 100 + k, 512, 16, 4);          // member groups: many short jobs
        });
    th.emplace_back([] { for (int k = 0; k < 3; ++k) roundtrip(0, 50 + k); });
    th.emplace_back([] { windows(0, 60, 100000, 8, 6); });
    for (auto &t : th) t.join();
    const qgs::BridgeStats bs = qgs::bridge_stats();
    if (g_fail.load()) { std::printf("FAILED %d checks\n", g_fail.load()); return 1; }
    std::printf("OK shards=%d h2d=%llu d2h=%llu jobs=%llu threads=%d\n", shards, (unsigned long long)bs.h2d_bytes, (unsigned long long)bs.d2h_bytes,
                (unsigned long long)bs.row_jobs, qgs::host_copy_threads());
    return 0;
}
