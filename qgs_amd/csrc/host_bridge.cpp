// host_bridge.cpp -- see host_bridge.h: page-locked bounce rings between the device and the caller's (pageable) host memory.
#include "host_bridge.h"

#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>
#include <vector>

namespace qgs {
namespace {

constexpr int MAX_DEVICES = 64;
constexpr size_t SYNC_BLOCK = (size_t)8 << 20, ASYNC_BLOCK = (size_t)16 << 20;
constexpr int SYNC_BLOCKS = 2, ASYNC_BLOCKS = 4;
constexpr size_t PARALLEL_MIN = (size_t)1 << 20;        // below this a gather / scatter is done by the calling thread alone
constexpr size_t TASK_BYTES = (size_t)512 << 10;        // grain of the parallel gather / scatter

std::atomic<uint64_t> g_h2d_bytes{0}, g_d2h_bytes{0}, g_row_jobs{0};

int set_err(std::string *err, const std::string &msg)
{
    if (err) *err = msg;
    return -1;
}

#define BRCHK(expr)                                                                                                         \
    do {                                                                                                                    \
        hipError_t e__ = (expr);                                                                                            \
        if (e__ != hipSuccess) {                                                                                            \
            std::ostringstream os__;                                                                                        \
            os__ << #expr << " failed: " << hipGetErrorString(e__) << " (" << __FILE__ << ":" << __LINE__ << ")";          \
            return set_err(err, os__.str());                                                                                \
        }                                                                                                                   \
    } while (0)

// CPUs this process may use: its affinity mask, cut down to a cgroup CPU quota when there is one
int available_cpus()
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string quota;
    long long period = 0;
    if (f && (f >> quota >> period) && quota != "max" && period > 0) {
        const long long q = std::atoll(quota.c_str());
        if (q > 0) n = std::min<long long>(n, std::max<long long>(1, (q + period - 1) / period));
    }
    return std::max(1, n);
}

// A few persistent host threads that run one parallel loop at a time.  Never destroyed: the process ends with its threads
// parked (a static destructor that joined them would run after the interpreter that loaded this library has begun to tear down).
class CpuPool {
public:
    static CpuPool &get()
    {
        static CpuPool *p = new CpuPool();
        return *p;
    }
    int threads() const { return n_threads_; }
    // f(i) for i in [0, n): on the pool's threads and the calling one; returns when all are done
    void run(size_t n, const std::function<void(size_t)> &f)
    {
        if (n == 0) return;
        if (n == 1 || n_threads_ <= 1) {
            for (size_t i = 0; i < n; ++i) f(i);
            return;
        }
        std::lock_guard<std::mutex> region(region_mu_);
        start_workers();
        {
            std::lock_guard<std::mutex> lock(mu_);
            fn_ = &f;
            n_items_ = n;
            next_.store(0);
            active_ = (int)workers_.size();
            ++generation_;
        }
        cv_work_.notify_all();
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= n) break;
            f(i);
        }
        std::unique_lock<std::mutex> lock(mu_);
        cv_done_.wait(lock, [&] { return active_ == 0; });
        fn_ = nullptr;
    }

private:
    CpuPool()
    {
        int n = std::min(16, available_cpus());
        if (const char *e = std::getenv("QGS_HIP_HOST_THREADS")) n = std::atoi(e);
        n_threads_ = std::max(1, std::min(64, n));
    }
    void start_workers()
    {
        if (started_pid_ == getpid()) return;
        // (first use, or first use in a forked child: the parent's threads do not exist here)
        workers_.clear();
        started_pid_ = getpid();
        for (int t = 0; t + 1 < n_threads_; ++t) {
            workers_.push_back(std::make_unique<std::thread>([this] { work(); }));
            workers_.back()->detach();
        }
    }
    void work()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_work_.wait(lock, [&] { return generation_ != seen; });
                seen = generation_;
            }
            for (;;) {
                const size_t i = next_.fetch_add(1);
                if (i >= n_items_) break;
                (*fn_)(i);
            }
            std::lock_guard<std::mutex> lock(mu_);
            if (--active_ == 0) cv_done_.notify_one();
        }
    }
    int n_threads_ = 1;
    pid_t started_pid_ = -1;
    std::vector<std::unique_ptr<std::thread>> workers_;
    std::mutex region_mu_, mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_items_ = 0;
    std::atomic<size_t> next_{0};
    int active_ = 0;
    uint64_t generation_ = 0;
};

// rows of `width` bytes between a packed block (row r at r * width) and strided memory (row r at r * pitch)
void move_rows(char *strided, size_t pitch, char *packed, size_t width, size_t rows, bool to_strided)
{
    const size_t total = width * rows;
    auto one = [&](size_t r_lo, size_t r_hi) {
        if (pitch == width) {
            if (to_strided) std::memcpy(strided + r_lo * width, packed + r_lo * width, (r_hi - r_lo) * width);
            else std::memcpy(packed + r_lo * width, strided + r_lo * width, (r_hi - r_lo) * width);
            return;
        }
        for (size_t r = r_lo; r < r_hi; ++r) {
            if (to_strided) std::memcpy(strided + r * pitch, packed + r * width, width);
            else std::memcpy(packed + r * width, strided + r * pitch, width);
        }
    };
    if (total < PARALLEL_MIN || CpuPool::get().threads() <= 1) {
        if (rows == 1 || pitch == width) one(0, rows);
        else one(0, rows);
        return;
    }
    if (rows == 1) {                                     // one long run: split it by bytes
        const size_t n_tasks = (width + TASK_BYTES - 1) / TASK_BYTES;
        CpuPool::get().run(n_tasks, [&](size_t t) {
            const size_t lo = t * TASK_BYTES, hi = std::min(width, lo + TASK_BYTES);
            if (to_strided) std::memcpy(strided + lo, packed + lo, hi - lo);
            else std::memcpy(packed + lo, strided + lo, hi - lo);
        });
        return;
    }
    const size_t rows_per = std::max<size_t>(1, TASK_BYTES / width);
    const size_t n_tasks = (rows + rows_per - 1) / rows_per;
    CpuPool::get().run(n_tasks, [&](size_t t) { one(t * rows_per, std::min(rows, (t + 1) * rows_per)); });
}

struct Blocks {
    std::vector<char *> blk;
    std::vector<hipEvent_t> ev;
    size_t bytes = 0;
    int ensure(int count, size_t block_bytes, std::string *err)
    {
        if (!blk.empty()) return 0;
        bytes = block_bytes;
        for (int i = 0; i < count; ++i) {
            void *p = nullptr;
            hipEvent_t e = nullptr;
            BRCHK(hipHostMalloc(&p, block_bytes, hipHostMallocPortable));
            blk.push_back((char *)p);
            BRCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ev.push_back(e);
        }
        return 0;
    }
};

// piece c of a (rows x row_bytes) transfer cut to blocks of `cap` bytes: whole rows when a row fits, else segments of one row
struct Pieces {
    size_t row_bytes, rows, cap, rows_per = 0, segs = 0, count = 0;
    Pieces(size_t rb, size_t r, size_t c) : row_bytes(rb), rows(r), cap(c)
    {
        if (row_bytes <= cap) { rows_per = std::max<size_t>(1, cap / row_bytes); count = (rows + rows_per - 1) / rows_per; }
        else { segs = (row_bytes + cap - 1) / cap; count = rows * segs; }
    }
    // packed-side byte offset, first row, rows, bytes per row, offset inside the row
    void piece(size_t c, size_t *packed_off, size_t *r0, size_t *nr, size_t *width, size_t *col) const
    {
        if (rows_per) {
            *r0 = c * rows_per; *nr = std::min(rows_per, rows - *r0); *width = row_bytes; *col = 0; *packed_off = *r0 * row_bytes;
        } else {
            *r0 = c / segs; *nr = 1; *col = (c % segs) * cap; *width = std::min(cap, row_bytes - *col); *packed_off = *r0 * row_bytes + *col;
        }
    }
};

// device (rows packed) -> host (rows `dst_pitch` apart): DMA into block c % N while the host threads scatter block c - N + 1
int pump_d2h(Blocks &bs, hipStream_t st, char *dst, size_t dst_pitch, const char *src, size_t row_bytes, size_t rows,
             const std::function<void()> &copied, std::string *err)
{
    const Pieces P(row_bytes, rows, bs.bytes);
    const size_t N = bs.blk.size();
    for (size_t c = 0; c < P.count + N - 1; ++c) {
        size_t off, r0, nr, width, col;
        if (c < P.count) {
            P.piece(c, &off, &r0, &nr, &width, &col);
            BRCHK(hipMemcpyAsync(bs.blk[c % N], src + off, nr * width, hipMemcpyDeviceToHost, st));
            BRCHK(hipEventRecord(bs.ev[c % N], st));
        }
        if (c + 1 >= N && c + 1 - N < P.count) {
            const size_t d = c + 1 - N;
            BRCHK(hipEventSynchronize(bs.ev[d % N]));
            if (d == P.count - 1 && copied) copied();          // the device block has been read to its end
            P.piece(d, &off, &r0, &nr, &width, &col);
            move_rows(dst + r0 * dst_pitch + col, dst_pitch, bs.blk[d % N], width, nr, true);
        }
    }
    g_d2h_bytes += (uint64_t)row_bytes * rows;
    return 0;
}

// host (contiguous) -> device: the host threads fill block c while the DMA of block c - 1 runs
int pump_h2d(Blocks &bs, hipStream_t st, char *dst_dev, const char *src, size_t bytes, std::string *err)
{
    const Pieces P(bytes, 1, bs.bytes);
    const size_t N = bs.blk.size();
    for (size_t c = 0; c < P.count; ++c) {
        size_t off, r0, nr, width, col;
        P.piece(c, &off, &r0, &nr, &width, &col);
        if (c >= N) BRCHK(hipEventSynchronize(bs.ev[c % N]));
        move_rows(const_cast<char *>(src) + off, width, bs.blk[c % N], width, 1, false);
        BRCHK(hipMemcpyAsync(dst_dev + off, bs.blk[c % N], width, hipMemcpyHostToDevice, st));
        BRCHK(hipEventRecord(bs.ev[c % N], st));
    }
    BRCHK(hipStreamSynchronize(st));
    g_h2d_bytes += bytes;
    return 0;
}

struct Job {
    int64_t seq;
    char *dst;
    size_t dst_pitch;
    const char *src;
    size_t row_bytes, rows;
    hipEvent_t ready;
};

struct Device {
    int dev = 0;
    std::mutex sync_mu;                 // blocking copies: one at a time per device
    Blocks sync_blocks, async_blocks;
    // drain thread: jobs in order
    std::mutex mu;
    std::condition_variable cv_job, cv_state;
    std::deque<Job> queue;
    int64_t next_seq = 1, copied_upto = 0, done_upto = 0;
    std::map<int64_t, std::string> failed;
    bool running = false;
    pid_t pid = -1;
    hipStream_t dma = nullptr;

    void drain_loop()
    {
        std::string err;
        bool ok = hipSetDevice(dev) == hipSuccess;
        if (ok && !dma) ok = hipStreamCreateWithFlags(&dma, hipStreamNonBlocking) == hipSuccess;
        if (ok) ok = async_blocks.ensure(ASYNC_BLOCKS, ASYNC_BLOCK, &err) == 0;
        if (!ok && err.empty()) err = "host bridge: the drain thread could not set up its stream";
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv_job.wait(lock, [&] { return !queue.empty(); });
                j = queue.front();
                queue.pop_front();
            }
            std::string jerr = err;
            bool signalled = false;
            auto copied = [&] {
                std::lock_guard<std::mutex> lock(mu);
                copied_upto = j.seq;
                signalled = true;
                cv_state.notify_all();
            };
            int rc = ok ? 0 : -1;
            if (rc == 0 && j.ready && hipStreamWaitEvent(dma, j.ready, 0) != hipSuccess) { rc = -1; jerr = "host bridge: hipStreamWaitEvent failed"; }
            if (rc == 0) rc = pump_d2h(async_blocks, dma, j.dst, j.dst_pitch, j.src, j.row_bytes, j.rows, copied, &jerr);
            if (rc != 0) (void)hipStreamSynchronize(dma);       // nothing of a failed job stays in flight
            std::lock_guard<std::mutex> lock(mu);
            if (rc != 0) failed[j.seq] = jerr.empty() ? "host bridge: transfer failed" : jerr;
            if (!signalled) copied_upto = j.seq;
            done_upto = j.seq;
            cv_state.notify_all();
        }
    }
    void ensure_thread()
    {
        // (mu held)  a forked child starts its own thread: the parent's does not exist there
        if (running && pid == getpid()) return;
        running = true;
        pid = getpid();
        std::thread([this] { drain_loop(); }).detach();
    }
};

Device *device_state(int dev)
{
    static std::mutex mu;
    static Device *table[MAX_DEVICES] = {nullptr};
    if (dev < 0 || dev >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!table[dev]) { table[dev] = new Device(); table[dev]->dev = dev; }     // (never freed: see CpuPool)
    return table[dev];
}

Device *current_device(std::string *err)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_err(err, "host bridge: no current device"); return nullptr; }
    Device *d = device_state(dev);
    if (!d) set_err(err, "host bridge: device index out of range");
    return d;
}

}  // namespace

int host_copy_threads() { return CpuPool::get().threads(); }

int bridge_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t st, std::string *err)
{
    if (bytes == 0) return 0;
    Device *d = current_device(err);
    if (!d) return -1;
    BRCHK(hipStreamSynchronize(st));
    std::lock_guard<std::mutex> lock(d->sync_mu);
    if (d->sync_blocks.ensure(SYNC_BLOCKS, SYNC_BLOCK, err)) return -1;
    return pump_h2d(d->sync_blocks, st, (char *)dst_dev, (const char *)src_host, bytes, err);
}

int bridge_d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t st, std::string *err)
{
    if (bytes == 0) return 0;
    Device *d = current_device(err);
    if (!d) return -1;
    BRCHK(hipStreamSynchronize(st));
    std::lock_guard<std::mutex> lock(d->sync_mu);
    if (d->sync_blocks.ensure(SYNC_BLOCKS, SYNC_BLOCK, err)) return -1;
    return pump_d2h(d->sync_blocks, st, (char *)dst_host, bytes, (const char *)src_dev, bytes, 1, nullptr, err);
}

int64_t bridge_d2h_rows_async(char *dst_host, size_t dst_pitch, const char *src_dev, size_t row_bytes, size_t rows, hipEvent_t ready,
                              std::string *err)
{
    if (row_bytes == 0 || rows == 0) { set_err(err, "host bridge: empty transfer"); return -1; }
    Device *d = current_device(err);
    if (!d) return -1;
    std::lock_guard<std::mutex> lock(d->mu);
    d->ensure_thread();
    const int64_t seq = d->next_seq++;
    d->queue.push_back(Job{seq, dst_host, dst_pitch, src_dev, row_bytes, rows, ready});
    d->cv_job.notify_one();
    ++g_row_jobs;
    return seq * MAX_DEVICES + d->dev;
}

static int wait_for(int64_t ticket, bool done, std::string *err)
{
    if (ticket <= 0) return set_err(err, "host bridge: bad ticket");
    Device *d = device_state((int)(ticket % MAX_DEVICES));
    if (!d) return set_err(err, "host bridge: bad ticket");
    const int64_t seq = ticket / MAX_DEVICES;
    std::unique_lock<std::mutex> lock(d->mu);
    d->cv_state.wait(lock, [&] { return (done ? d->done_upto : d->copied_upto) >= seq; });
    auto it = d->failed.find(seq);
    if (it == d->failed.end()) return 0;
    const std::string msg = it->second;
    if (done) d->failed.erase(it);
    return set_err(err, msg);
}

int bridge_poll_copied(int64_t ticket)
{
    if (ticket <= 0) return 1;
    Device *d = device_state((int)(ticket % MAX_DEVICES));
    if (!d) return 1;
    std::lock_guard<std::mutex> lock(d->mu);
    return d->copied_upto >= ticket / MAX_DEVICES ? 1 : 0;
}

int bridge_wait_copied(int64_t ticket, std::string *err) { return wait_for(ticket, false, err); }
int bridge_wait_done(int64_t ticket, std::string *err) { return wait_for(ticket, true, err); }

BridgeStats bridge_stats() { return BridgeStats{g_h2d_bytes.load(), g_d2h_bytes.load(), g_row_jobs.load()}; }

}  // namespace qgs
