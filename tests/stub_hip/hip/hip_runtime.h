// Test double of the small part of the HIP runtime qgs_amd/csrc/host_bridge.cpp uses, for the CPU sanitizer builds of
// tests/test_host_bridge_sanitizers_cpu.py (GPU-side sanitizers are not available on the pool; the window / drain / bounce-ring logic
// is host arithmetic plus copies).  "Device memory" is heap memory; a stream is a worker thread that executes its operations in
// order and ASYNCHRONOUSLY to the caller -- so the bridge's own synchronisation (events, stream waits, tickets) is what orders the
// copies, and ThreadSanitizer sees every byte the "copy engine" and the host threads touch.  Not a product file.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
enum { hipHostMallocPortable = 1, hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };

struct StubEvent {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t recorded = 0, completed = 0;
};
typedef StubEvent *hipEvent_t;

struct StubStream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool busy = false;
    std::thread worker;
    StubStream() : worker([this] { run(); }) { worker.detach(); }
    void push(std::function<void()> f)
    {
        std::lock_guard<std::mutex> lock(mu);
        q.push_back(std::move(f));
        cv.notify_all();
    }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return !q.empty(); });
                f = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            f();
            std::lock_guard<std::mutex> lock(mu);
            busy = false;
            cv.notify_all();
        }
    }
    void drain()
    {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return q.empty() && !busy; });
    }
};
typedef StubStream *hipStream_t;

namespace stub_hip {
inline StubStream *null_stream(int dev)
{
    static std::mutex mu;
    static StubStream *table[64] = {nullptr};
    std::lock_guard<std::mutex> lock(mu);
    if (!table[dev & 63]) table[dev & 63] = new StubStream();
    return table[dev & 63];
}
inline int &current() { thread_local int d = 0; return d; }
inline StubStream *resolve(hipStream_t s) { return s ? s : null_stream(current()); }
// fault injection for the error paths: the n-th hipMemcpyAsync from now fails (0: never)
inline long &fail_memcpy_in() { static long n = 0; return n; }
inline std::mutex &fail_mu() { static std::mutex m; return m; }
}  // namespace stub_hip

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "stub failure"; }
inline hipError_t hipSetDevice(int d) { stub_hip::current() = d; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = stub_hip::current(); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new StubStream(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new StubEvent(); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new StubEvent(); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t s)
{
    {
        std::lock_guard<std::mutex> lock(stub_hip::fail_mu());
        long &k = stub_hip::fail_memcpy_in();
        if (k > 0 && --k == 0) return hipErrorInvalidValue;
    }
    stub_hip::resolve(s)->push([=] { std::memcpy(dst, src, n); });
    return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    uint64_t target;
    {
        std::lock_guard<std::mutex> lock(e->mu);
        target = ++e->recorded;
    }
    stub_hip::resolve(s)->push([=] {
        std::lock_guard<std::mutex> lock(e->mu);
        if (e->completed < target) e->completed = target;
        e->cv.notify_all();
    });
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t e)
{
    std::unique_lock<std::mutex> lock(e->mu);
    const uint64_t target = e->recorded;
    e->cv.wait(lock, [&] { return e->completed >= target; });
    return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    uint64_t target;
    {
        std::lock_guard<std::mutex> lock(e->mu);
        target = e->recorded;
    }
    stub_hip::resolve(s)->push([=] {
        std::unique_lock<std::mutex> lock(e->mu);
        e->cv.wait(lock, [&] { return e->completed >= target; });
    });
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t s) { stub_hip::resolve(s)->drain(); return hipSuccess; }
