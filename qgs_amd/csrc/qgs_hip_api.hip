// qgs_hip_api.hip -- implementation of the C-ABI declared in include/qgs_hip.h.  gfx950 only.
//
// ONE translation unit (the pieces share the model struct and a handful of file-local helpers), kept in one file per concern
// and included below in this order:
//   api_kernel_cache.inc     which compiler builds the specialised kernels (helper process / in-process hiprtc) and the on-disk kernel cache
//   api_model_state.inc      what a model owns on the device: tensors in the generic kernels' layouts, scratch buffers, upload ring, tuning knobs; struct qgs_model
//   api_launch.inc           model classification, generator options, kernel selection by ensemble size / tableau / system size, and the launchers of every kernel family
//   api_model.inc            C-ABI: backend information, model creation / destruction / inspection, kernel pre-build
//   api_device.inc           C-ABI: device-layout entry points (the caller holds device pointers; nothing crosses PCIe)
//   api_host.inc             C-ABI: host-layout entry points -- single-state f / Df, and the integrations whose records leave in windows through the host bridge
//   api_contraction.inc      C-ABI: the general contraction (sparse_mul3 / 5 / 2 / 4 with any vectors), host registration
//   api_group.inc            C-ABI: all GPUs of the node behind one handle (qgs_group): one host thread per shard
//
//
// Host side of the MI355X path: stages the model tensors on the device, generates and compiles the
// tensor-specialised kernels (codegen.cpp + hiprtc, cached on disk), owns the scratch buffers and
// launches either the specialised or the generic kernels.  No CPU compute path exists here: every
// entry point fails if no GPU is visible.
#include "../../include/qgs_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <limits.h>
#include <spawn.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "codegen.h"
#include "generic_kernels.h"
#include "host_bridge.h"

extern char **environ;

#ifndef QGS_LDS_STATE_BYTES
#define QGS_LDS_STATE_BYTES (152 * 1024)   // stage state of the LDS-resident stepper: ndim * 64 members * 8 B (160 KB LDS per CU)
#endif
#ifndef QGS_SPEC_MAX_NDIM
#define QGS_SPEC_MAX_NDIM 64      // register-resident specialised kernels up to this many variables
#endif
#ifndef QGS_SPEC_MAX_DERIVED
#define QGS_SPEC_MAX_DERIVED 256  // ... and (rank-5 tensors) this many derived monomials per tendency evaluation
#endif
namespace {


thread_local std::string g_err;

int fail(const std::string &msg)
{
    g_err = msg;
    return -1;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            std::ostringstream os__;                                                                   \
            os__ << #expr << " failed: " << hipGetErrorString(e__) << " (" << __FILE__ << ":" << __LINE__ << ")"; \
            return fail(os__.str());                                                                   \
        }                                                                                              \
    } while (0)

uint64_t fnv1a(const std::string &s, uint64_t h = 1469598103934665603ull)
{
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

// 128 bits from two differently mixed 64-bit lanes over 8-byte words (names and integrity checks of the kernel cache; not
// cryptographic: it guards against collisions of the 64-bit file name, truncated files and bit rot, not against an adversary)
struct Hash128 {
    uint64_t a = 0, b = 0;
    bool operator==(const Hash128 &o) const { return a == o.a && b == o.b; }
    std::string hex() const
    {
        char buf[40];
        std::snprintf(buf, sizeof buf, "%016llx%016llx", (unsigned long long)a, (unsigned long long)b);
        return buf;
    }
};
struct Hasher {
    uint64_t a = 1469598103934665603ull, b = 0x9e3779b97f4a7c15ull;
    void word(uint64_t w)
    {
        a = (a ^ w) * 1099511628211ull;
        a ^= a >> 31;
        b = ((b << 29) | (b >> 35)) ^ (w * 0xc2b2ae3d27d4eb4full);
        b *= 0x165667b19e3779f9ull;
        b ^= b >> 33;
    }
    void add(const void *p, size_t n)
    {
        const unsigned char *q = (const unsigned char *)p;
        word((uint64_t)n);
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, q + i, 8); word(w); }
        if (i < n) { uint64_t w = 0; std::memcpy(&w, q + i, n - i); word(w); }
    }
    void add(const std::string &s) { add(s.data(), s.size()); }
    Hash128 done() const
    {
        Hasher h = *this;
        h.word(0x51ed270b1f2c3d4eull);
        Hash128 r;
        r.a = h.a ^ (h.b >> 17);
        r.b = h.b ^ (h.a << 23);
        return r;
    }
};

std::string lib_dir()
{
    Dl_info info;
    if (dladdr((void *)&fnv1a, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t k = p.find_last_of('/');
        return k == std::string::npos ? std::string(".") : p.substr(0, k);
    }
    return ".";
}

std::string cache_dir()
{
    const char *e = std::getenv("QGS_HIP_CACHE_DIR");
    std::string d = e && *e ? std::string(e) : lib_dir() + "/kcache";
    ::mkdir(d.c_str(), 0777);
    return d;
}

std::string target_arch(int device)
{
    const char *e = std::getenv("QGS_HIP_ARCH");
    if (e && *e) return e;
    if (device >= 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
            std::string a(prop.gcnArchName);
            size_t c = a.find(':');
            return c == std::string::npos ? a : a.substr(0, c);
        }
    }
    return "gfx950";
}

// Compiler flags beyond -O3 that every specialised kernel is built with (part of the cache key).
std::vector<std::string> default_extra_flags() { return {}; }

// source -> code object (hsaco), through the on-disk cache
// (developer build: extra compiler flags, e.g. QGS_HIP_EXTRA_FLAGS="-mllvm -amdgpu-sched-strategy=max-ilp")
std::vector<std::string> extra_flags()
{
#ifdef QGS_HIP_DEV_KNOBS
    if (const char *e = std::getenv("QGS_HIP_EXTRA_FLAGS")) {
        std::vector<std::string> extra;
        std::istringstream is(e);
        for (std::string tok; is >> tok;) extra.push_back(tok);
        return extra;
    }
#endif
    return default_extra_flags();
}

}  // namespace

#include "api_kernel_cache.inc"
#include "api_model_state.inc"
#include "api_launch.inc"
#include "api_model.inc"
#include "api_device.inc"
#include "api_host.inc"
#include "api_contraction.inc"
#include "api_group.inc"
