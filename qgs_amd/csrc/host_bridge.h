// host_bridge.h -- every byte that moves between the device and host memory the CALLER owns goes through here.
//
// Rule (round 5): the GPU -- kernels and the copy engines alike -- only ever touches host memory this library
// allocated itself (hipHostMalloc), or memory the caller explicitly handed over with qgs_host_register.  Anything else is
// pageable as far as the library knows and is reached by the CPU only: a ring of page-locked bounce blocks per device, DMA
// between the device and a bounce block, and a gather / scatter between the bounce block and the caller's memory by a small
// pool of host threads, pipelined so that the copy engine stays busy.  The runtime is never asked to pin the caller's pages
// (hipMemcpy with a pageable operand does that in place, read-only for sources; hipHostRegister of heap blocks was where every
// GPU write fault of round 4 was found), so no allocator assumption is left in the path.
//
// The reference's results are plain host arrays (qgs/integrators/integrator.py:386-395); this is how they are filled.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

namespace qgs {

// host threads the gather / scatter uses (QGS_HIP_HOST_THREADS; default min(16, CPUs this process may run on))
int host_copy_threads();

// Blocking copies of `bytes` contiguous bytes; work already queued on `st` is waited for first, the copy has completed on return.
// Return 0, or -1 with *err set.
int bridge_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t st, std::string *err);
int bridge_d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t st, std::string *err);

// `rows` runs of `row_bytes` bytes, contiguous in device memory at `src_dev`, to dst_host + r * dst_pitch (the records of one
// window on their way into the caller's (member, variable, record) array).  Asynchronous: the job is queued on the device's
// drain thread and starts when `ready` (an event recorded on the producing stream; may be null: start at once) has completed.
// Returns a ticket (> 0), or -1 with *err set.
//   bridge_wait_copied(ticket): the device block has been read completely (it may be overwritten);
//   bridge_wait_done(ticket):   the bytes are in the caller's memory.
// Both return 0, or -1 with *err set when that job or an earlier one of the device failed.
int64_t bridge_d2h_rows_async(char *dst_host, size_t dst_pitch, const char *src_dev, size_t row_bytes, size_t rows, hipEvent_t ready,
                              std::string *err);
int bridge_wait_copied(int64_t ticket, std::string *err);
int bridge_poll_copied(int64_t ticket);          // 1: the device block of that job has been read completely (or the job failed), 0: not yet
int bridge_wait_done(int64_t ticket, std::string *err);

// counters for tests and measurements (process-wide): bytes that went through bounce blocks in each direction, jobs queued
struct BridgeStats {
    uint64_t h2d_bytes, d2h_bytes, row_jobs;
};
BridgeStats bridge_stats();

}  // namespace qgs
