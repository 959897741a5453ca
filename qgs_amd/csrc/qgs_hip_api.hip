// qgs_hip_api.hip -- implementation of the C-ABI declared in include/qgs_hip.h.  gfx950 only.
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

// ---- which compiler builds the specialised kernels -------------------------------------------------------------------
// Default: the helper process qgs_kcompile next to this library (kcompile.cpp), which is bound to the system ROCm's hiprtc /
// comgr.  A process that imported PyTorch has torch's older bundled pair mapped under the same sonames, and that pair
// generates worse code for the fused stepper (324 instead of 282 VGPRs); going through the helper gives the same code
// object everywhere.  QGS_HIP_INPROC_RTC=1 (or a missing helper) compiles with whatever hiprtc this process has mapped.
// The identity of the compiler is part of the cache key, so objects of one never pass for the other's.
std::string helper_path()
{
    if (const char *e = std::getenv("QGS_HIP_HELPER")) if (*e) return e;      // (tests: a helper that fails)
    return lib_dir() + "/qgs_kcompile";
}

// environment of the helper: no preloaded tool libraries (profilers), it must stay a plain compiler process
std::vector<std::string> helper_env()
{
    std::vector<std::string> env;
    for (char **e = environ; e && *e; ++e) {
        const std::string kv(*e);
        if (kv.rfind("LD_PRELOAD=", 0) == 0 || kv.rfind("HSA_TOOLS_LIB=", 0) == 0 || kv.rfind("ROCP_", 0) == 0 ||
            kv.rfind("ROCPROFILER_", 0) == 0 || kv.rfind("LD_LIBRARY_PATH=", 0) == 0) continue;
        env.push_back(kv);
    }
    return env;
}

// run the helper; stdout + stderr of the child end up in *output.
// Returns 0: exit code 0; 1: the helper ran and reported a failure (exit code 1: a compile error, its log is in *output);
// -1: the helper could not be run or did not end normally (spawn failure, signal, any other exit code, waitpid failure).
std::mutex g_helper_mutex;
int run_helper(const std::vector<std::string> &args, std::string *output)
{
    std::lock_guard<std::mutex> lock(g_helper_mutex);        // one spawn + read at a time: no other spawn of ours inherits the pipe
    int fds[2];
    if (pipe2(fds, O_CLOEXEC) != 0) return -1;                // (the dup2 targets 1 and 2 lose CLOEXEC in the child)
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_adddup2(&fa, fds[1], 1);
    posix_spawn_file_actions_adddup2(&fa, fds[1], 2);
    std::vector<char *> argv;
    for (const auto &a : args) argv.push_back(const_cast<char *>(a.c_str()));
    argv.push_back(nullptr);
    const std::vector<std::string> env = helper_env();
    std::vector<char *> envp;
    for (const auto &e : env) envp.push_back(const_cast<char *>(e.c_str()));
    envp.push_back(nullptr);
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, args[0].c_str(), &fa, nullptr, argv.data(), envp.data());
    posix_spawn_file_actions_destroy(&fa);
    close(fds[1]);
    if (rc != 0) { close(fds[0]); return -1; }
    char buf[4096];
    ssize_t n;
    while ((n = read(fds[0], buf, sizeof buf)) > 0 || (n < 0 && errno == EINTR)) if (n > 0 && output) output->append(buf, (size_t)n);
    close(fds[0]);
    int status = 0;
    pid_t w;
    while ((w = waitpid(pid, &status, 0)) < 0 && errno == EINTR) {}
    if (w != pid) return -1;                                   // e.g. ECHILD when the application ignores SIGCHLD: the exit status is lost
    if (!WIFEXITED(status)) return -1;
    return WEXITSTATUS(status) == 0 ? 0 : (WEXITSTATUS(status) == 1 ? 1 : -1);
}

// e.g. "inproc-hiprtc7.0-libhiprtc.so.7.0.51831": the hiprtc this process has mapped
std::string inproc_compiler_id()
{
    int major = 0, minor = 0;
    (void)hiprtcVersion(&major, &minor);
    std::string file = "?";
    Dl_info info;
    if (dladdr((void *)&hiprtcVersion, &info) && info.dli_fname) {
        char real[PATH_MAX];
        file = realpath(info.dli_fname, real) ? real : info.dli_fname;
        const size_t k = file.find_last_of('/');
        if (k != std::string::npos) file = file.substr(k + 1);
    }
    return "inproc-hiprtc" + std::to_string(major) + "." + std::to_string(minor) + "-" + file;
}

// Which compiler this process uses, decided once: the helper only when it exists AND answers `--version` (its answer, e.g.
// "hiprtc9.0-libhiprtc.so.7.2.70200", is the compiler identity in the cache key); otherwise the in-process hiprtc under its
// own identity.  A helper that stops working later (see obtain_blob) switches the process to the in-process compiler for
// good -- `helper` and `id` always change together.
struct CompilerChoice {
    bool helper = false;
    std::string id;
};
std::mutex g_compiler_mutex;
CompilerChoice &compiler_choice_locked()
{
    static CompilerChoice c;
    static bool decided = false;
    if (!decided) {
        decided = true;
        c.id = inproc_compiler_id();
        const char *e = std::getenv("QGS_HIP_INPROC_RTC");
        if (!(e && *e == '1')) {
            if (access(helper_path().c_str(), X_OK) != 0) {
                std::fprintf(stderr, "libqgs_hip: %s is missing (make -C qgs_amd/csrc); kernels that miss the cache are compiled by the "
                                     "hiprtc this process has mapped\n", helper_path().c_str());
            } else {
                std::string out;
                if (run_helper({helper_path(), "--version"}, &out) == 0 && !out.empty()) {
                    while (!out.empty() && (out.back() == '\n' || out.back() == '\r')) out.pop_back();
                    c.helper = true;
                    c.id = out;
                } else {
                    std::fprintf(stderr, "libqgs_hip: %s --version failed (%s); compiling in-process\n", helper_path().c_str(), out.c_str());
                }
            }
        }
    }
    return c;
}
CompilerChoice compiler_choice()
{
    std::lock_guard<std::mutex> lock(g_compiler_mutex);
    return compiler_choice_locked();
}
void disable_helper(const std::string &why)
{
    std::lock_guard<std::mutex> lock(g_compiler_mutex);
    CompilerChoice &c = compiler_choice_locked();
    if (c.helper) {
        std::fprintf(stderr, "libqgs_hip: %s no longer usable (%s); compiling in-process from now on\n", helper_path().c_str(), why.c_str());
        c.helper = false;
        c.id = inproc_compiler_id();
    }
}

// ---- kernel cache --------------------------------------------------------------------------------------------------------------
// Coefficient values are never part of a cache entry: every model fills the tables of its own loaded module with its own
// values, so a parameter sweep over one model shares its entries -- as the reference compiles sparse_mul3 once whatever `val`
// holds (sparse_mul.py:48-81).  Two kinds of entries, both named by the first 64 bits of a 128-bit key:
//
//   <structure key>.qgst   what one (kernel kind, tensor STRUCTURE, generator options) needs besides code: the layout of its
//                          coefficient tables in canonical form (magnitude-class ids, codegen.h Canonical) and the key of its
//                          code object.  The structure key is computed from the canonical tensor alone: a hit costs no
//                          generator run (0.9 s per kernel at ndim 228).
//   <code key>.hsaco       the code object of one generated SOURCE (+ compiler, flags, architecture).  Different structures
//                          often generate the same source -- a coincidence of two magnitudes only changes the source when
//                          the generator exploits it (same row, same de-duplication window) -- and then share the object.
//
// Files: payload | 64-byte footer (magic, own 128-bit key, payload length, 128-bit payload hash, and for a structure entry the
// code key).  A code entry starts with the ELF, so llvm-objdump reads it as it is.  A file that is truncated, damaged or
// belongs to a colliding key is not a hit: it is rebuilt and replaced.
struct KernelBlob {
    std::shared_ptr<const std::vector<char>> code;
    std::vector<qgs::CoefTable> tables;
};

#ifndef QGS_CODEGEN_HASH
#define QGS_CODEGEN_HASH "unversioned"      // the Makefile passes a hash of codegen.cpp + codegen.h: a changed generator never hits old entries
#endif
const char CODE_MAGIC[8] = {'Q', 'G', 'S', 'K', 'C', '0', '0', '2'};
const char STRUCT_MAGIC[8] = {'Q', 'G', 'S', 'K', 'T', '0', '0', '2'};

std::string serialise_tables(const std::vector<qgs::CoefTable> &tables)
{
    std::string out;
    auto put64 = [&](uint64_t v) { out.append((const char *)&v, 8); };
    put64(tables.size());
    for (const auto &t : tables) {
        put64(t.symbol.size());
        out += t.symbol;
        put64(t.values.size());
        out.append((const char *)t.values.data(), t.values.size() * sizeof(double));
    }
    return out;
}

bool parse_tables(const char *p, size_t n, std::vector<qgs::CoefTable> &tables)
{
    size_t pos = 0;
    auto get64 = [&](uint64_t *v) { if (pos + 8 > n) return false; std::memcpy(v, p + pos, 8); pos += 8; return true; };
    uint64_t nt;
    if (!get64(&nt) || nt > 4096) return false;
    tables.clear();
    for (uint64_t i = 0; i < nt; ++i) {
        uint64_t len, cnt;
        if (!get64(&len) || len > 256 || pos + len > n) return false;
        qgs::CoefTable t;
        t.symbol.assign(p + pos, (size_t)len);
        pos += (size_t)len;
        if (!get64(&cnt) || cnt > (n - pos) / sizeof(double)) return false;
        t.values.resize((size_t)cnt);
        std::memcpy(t.values.data(), p + pos, (size_t)cnt * sizeof(double));
        pos += (size_t)cnt * sizeof(double);
        tables.push_back(std::move(t));
    }
    return pos == n;
}

std::string cache_entry_path(const Hash128 &key, const char *ext)
{
    char name[32];
    std::snprintf(name, sizeof name, "%016llx", (unsigned long long)key.a);
    return cache_dir() + "/" + name + ext;
}

// payload of a verified entry (magic, key, length and payload hash all match), or false; *link: the second key of the footer
bool read_cache_file(const std::string &path, const char *magic, const Hash128 &key, std::vector<char> &payload, Hash128 *link)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<char> all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (all.size() < 64) return false;
    const char *ft = all.data() + all.size() - 64;
    uint64_t w[7];
    std::memcpy(w, ft + 8, sizeof w);
    if (std::memcmp(ft, magic, 8) != 0 || w[0] != key.a || w[1] != key.b) return false;
    if (w[2] + 64 != all.size()) return false;
    Hasher h;
    h.add(all.data(), (size_t)w[2]);
    const Hash128 sum = h.done();
    if (sum.a != w[3] || sum.b != w[4]) return false;
    if (link) { link->a = w[5]; link->b = w[6]; }
    all.resize((size_t)w[2]);
    payload.swap(all);
    (void)utimensat(AT_FDCWD, path.c_str(), nullptr, 0);          // last use, for the eviction order (fails quietly on a read-only cache)
    return true;
}

// The cache directory is bounded: QGS_HIP_CACHE_MAX_MB (default 2048; 0 = unbounded).  After a publish that takes it over the
// bound the least recently used entries (hits refresh the modification time) are removed down to 80 % of it.
void enforce_cache_limit(const std::string &dir, const std::string &keep)
{
    double limit_mb = 2048.0;
    if (const char *e = std::getenv("QGS_HIP_CACHE_MAX_MB")) limit_mb = std::atof(e);
    if (!(limit_mb > 0.0)) return;
    const double limit = limit_mb * 1048576.0;
    DIR *d = opendir(dir.c_str());
    if (!d) return;
    struct Ent { double mtime; double size; std::string path; };
    std::vector<Ent> ents;
    double total = 0.0;
    auto ends_with = [](const std::string &s, const char *e) { const size_t n = std::strlen(e); return s.size() > n && s.compare(s.size() - n, n, e) == 0; };
    while (struct dirent *de = readdir(d)) {
        const std::string name(de->d_name);
        if (!ends_with(name, ".hsaco") && !ends_with(name, ".qgst")) continue;
        const std::string path = dir + "/" + name;
        struct stat sb;
        if (stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) continue;
        ents.push_back({(double)sb.st_mtim.tv_sec + 1e-9 * (double)sb.st_mtim.tv_nsec, (double)sb.st_size, path});
        total += (double)sb.st_size;
    }
    closedir(d);
    if (total <= limit) return;
    std::sort(ents.begin(), ents.end(), [](const Ent &x, const Ent &y) { return x.mtime < y.mtime; });
    for (const Ent &e : ents) {
        if (total <= 0.8 * limit) break;
        if (e.path == keep) continue;
        if (std::remove(e.path.c_str()) == 0) total -= e.size;
    }
}

// Best effort: put an entry into the kernel cache (write next to the final name, then rename = atomic publish).  A
// cache directory that is read-only (shared install) just means the next process builds it again.
void publish_cache_file(const std::string &path, const char *magic, const Hash128 &key, const char *payload, size_t n, const Hash128 &link)
{
    Hasher h;
    h.add(payload, n);
    const Hash128 sum = h.done();
    char footer[64];
    std::memset(footer, 0, sizeof footer);
    std::memcpy(footer, magic, 8);
    const uint64_t w[7] = {key.a, key.b, (uint64_t)n, sum.a, sum.b, link.a, link.b};
    std::memcpy(footer + 8, w, sizeof w);
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    {
        std::ofstream f(tmp, std::ios::binary);
        if (!f) return;
        f.write(payload, (std::streamsize)n);
        f.write(footer, sizeof footer);
        f.close();
        if (!f) { std::remove(tmp.c_str()); return; }
    }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) { std::remove(tmp.c_str()); return; }
    const size_t k = path.find_last_of('/');
    enforce_cache_limit(k == std::string::npos ? std::string(".") : path.substr(0, k), path);
}

int compile_in_process(const std::string &src, const std::string &arch, const std::vector<std::string> &extra, std::vector<char> &code)
{
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "qgs_spec.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail("hiprtcCreateProgram failed");
    const std::string archopt = "--offload-arch=" + arch;
    std::vector<const char *> opts = {archopt.c_str(), "-O3", "-std=c++17"};
    for (const auto &x : extra) opts.push_back(x.c_str());
    hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return fail(std::string("hiprtc compile failed: ") + hiprtcGetErrorString(r) + "\n" + log.substr(0, 4000));
    }
    size_t n = 0;
    hiprtcGetCodeSize(prog, &n);
    code.resize(n);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return 0;
}

std::string scratch_dir()
{
    if (const char *e = std::getenv("TMPDIR")) if (*e && access(e, W_OK) == 0) return e;
    return "/tmp";
}

// a fresh private file under the scratch directory (mkstemp); "" on failure
std::string make_temp(const std::string &suffix_hint)
{
    std::string templ = scratch_dir() + "/qgs_hip_" + suffix_hint + "_XXXXXX";
    std::vector<char> buf(templ.begin(), templ.end());
    buf.push_back('\0');
    const int fd = mkstemp(buf.data());
    if (fd < 0) return "";
    close(fd);
    return std::string(buf.data());
}

bool read_file(const std::string &path, std::vector<char> &out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return !out.empty();
}

// Several processes that miss the same cache entry at the same time (8 ranks creating the same model on a cold cache)
// compile it once: the first takes an advisory lock on <entry>.lock, the others wait on it and then find the entry.  No lock
// (read-only cache directory): everybody compiles, nothing is shared, nothing breaks.
struct CacheLock {
    int fd = -1;
    std::string path;
    explicit CacheLock(const std::string &entry) : path(entry + ".lock")
    {
        fd = open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0666);
        if (fd >= 0) while (flock(fd, LOCK_EX) != 0 && errno == EINTR) {}
    }
    ~CacheLock()
    {
        if (fd >= 0) {
            std::remove(path.c_str());
            flock(fd, LOCK_UN);
            close(fd);
        }
    }
};

// source -> code object with the helper process.  0: done; 1: the helper is not usable (any more), *why says so; -1: a real
// compile error (g_err holds the compiler's log)
int compile_with_helper(const std::string &src, const std::string &arch, const std::vector<std::string> &extra, std::vector<char> &code,
                        std::string *why)
{
    // source and object travel through private temp files under $TMPDIR, not through the cache directory
    const std::string srcfile = make_temp("src"), objfile = make_temp("obj");
    if (srcfile.empty() || objfile.empty()) *why = "cannot create temp files under " + scratch_dir();
    if (why->empty()) {
        std::ofstream f(srcfile, std::ios::binary);
        f << src;
        f.close();
        if (!f) *why = "cannot write " + srcfile;
    }
    int rc = -1;
    std::string out;
    if (why->empty()) {
        std::vector<std::string> args = {helper_path(), arch, srcfile, objfile};
        args.insert(args.end(), extra.begin(), extra.end());
        rc = run_helper(args, &out);
        if (rc < 0) *why = "helper did not run to completion: " + out.substr(0, 400);
    }
    const bool got = (rc == 0) && read_file(objfile, code);
    if (rc == 0 && !got) *why = "helper produced no code object";
    if (!srcfile.empty()) std::remove(srcfile.c_str());
    if (!objfile.empty()) std::remove(objfile.c_str());
    if (rc == 1) { fail("kernel compilation failed (" + helper_path() + "):\n" + out.substr(0, 4000)); return -1; }    // a real compile error
    return got ? 0 : 1;
}

// entries this process has already read or built: a second model of the same structure (the next point of a parameter sweep,
// the other shards of a device group) costs neither a file read nor a generator run
std::mutex g_memo_mutex;
std::map<std::string, std::shared_ptr<const KernelBlob>> g_memo;                 // structure key -> blob
std::map<std::string, std::shared_ptr<const std::vector<char>>> g_code_memo;     // code key -> code object

std::shared_ptr<const std::vector<char>> code_memo_get(const std::string &k)
{
    std::lock_guard<std::mutex> lock(g_memo_mutex);
    auto it = g_code_memo.find(k);
    return it == g_code_memo.end() ? nullptr : it->second;
}
void code_memo_put(const std::string &k, std::shared_ptr<const std::vector<char>> c)
{
    std::lock_guard<std::mutex> lock(g_memo_mutex);
    if (g_code_memo.size() >= 1024) g_code_memo.clear();
    g_code_memo[k] = c;
}

// The blob of the kernel identified by `what` (everything that decides the generated source: kernel kind, generator options,
// canonical tensor), through memo -> structure entry + code entry on disk -> generate (+ compile) + publish.  `gen` is only
// called when the structure is new to the cache.  mode: Use (whatever serves the blob fastest), Lookup (never generate or
// compile: 1 when the entry exists nowhere), Publish (pre-build: both entries must also be on disk when the call returns -- a
// memo hit whose files are gone, e.g. evicted or another cache directory, is written again).
enum class BlobMode { Use, Lookup, Publish };
int obtain_blob(const std::string &what, const std::string &arch, const std::vector<std::string> &kernel_flags,
                const std::function<qgs::GeneratedKernel()> &gen, std::shared_ptr<const KernelBlob> *out, bool *from_cache,
                BlobMode mode = BlobMode::Use)
{
    std::vector<std::string> extra = extra_flags();
    extra.insert(extra.end(), kernel_flags.begin(), kernel_flags.end());
    for (int attempt = 0; attempt < 2; ++attempt) {
        const CompilerChoice cc = compiler_choice();
        std::string common = "qgs-kernel-cache-v4|" QGS_CODEGEN_HASH "|" + arch + "|O3|c++17|" + cc.id;
        for (const auto &x : extra) common += "|" + x;
        Hasher hk;
        hk.add(common);
        hk.add(what);
        const Hash128 skey = hk.done();
        const std::string memo_key = skey.hex();
        const std::string spath = cache_entry_path(skey, ".qgst");
        auto code_key_of = [&](const std::string &source) {
            Hasher h;
            h.add(common);
            h.add(source);
            return h.done();
        };
        auto publish_struct = [&](const KernelBlob &b, const Hash128 &ckey) {
            const std::string tab = serialise_tables(b.tables);
            publish_cache_file(spath, STRUCT_MAGIC, skey, tab.data(), tab.size(), ckey);
        };
        {
            std::shared_ptr<const KernelBlob> hit;
            {
                std::lock_guard<std::mutex> lock(g_memo_mutex);
                auto it = g_memo.find(memo_key);
                if (it != g_memo.end()) hit = it->second;
            }
            if (hit && mode == BlobMode::Publish && access(spath.c_str(), R_OK) != 0) hit = nullptr;     // rebuilt (below) into this directory
            if (hit) { *out = hit; if (from_cache) *from_cache = true; return 0; }
        }
        auto remember = [&](std::shared_ptr<KernelBlob> b) {
            std::lock_guard<std::mutex> lock(g_memo_mutex);
            if (g_memo.size() >= 4096) g_memo.clear();
            g_memo[memo_key] = b;
            *out = b;
        };
        // structure entry -> code entry
        auto try_disk = [&](std::shared_ptr<KernelBlob> blob) {
            std::vector<char> tab;
            Hash128 ckey;
            if (!read_cache_file(spath, STRUCT_MAGIC, skey, tab, &ckey)) return false;
            if (!parse_tables(tab.data(), tab.size(), blob->tables)) return false;
            blob->code = code_memo_get(ckey.hex());
            if (!blob->code) {
                auto code = std::make_shared<std::vector<char>>();
                if (!read_cache_file(cache_entry_path(ckey, ".hsaco"), CODE_MAGIC, ckey, *code, nullptr)) return false;
                blob->code = code;
                code_memo_put(ckey.hex(), code);
            } else if (mode == BlobMode::Publish && access(cache_entry_path(ckey, ".hsaco").c_str(), R_OK) != 0) {
                publish_cache_file(cache_entry_path(ckey, ".hsaco"), CODE_MAGIC, ckey, blob->code->data(), blob->code->size(), Hash128());
            }
            return true;
        };
        auto blob = std::make_shared<KernelBlob>();
        if (try_disk(blob)) { remember(blob); if (from_cache) *from_cache = true; return 0; }
        if (mode == BlobMode::Lookup) return 1;
        CacheLock slock(spath);
        if (try_disk(blob)) { remember(blob); if (from_cache) *from_cache = true; return 0; }   // somebody else built it meanwhile
        qgs::GeneratedKernel g;
        try {
            g = gen();
        } catch (const std::exception &e) {
            return fail(std::string("kernel generator: ") + e.what());
        }
        blob->tables = std::move(g.tables);
        // the code object of this source: memo -> disk -> compile
        const Hash128 ckey = code_key_of(g.source);
        const std::string cpath = cache_entry_path(ckey, ".hsaco");
        blob->code = code_memo_get(ckey.hex());
        if (blob->code && mode == BlobMode::Publish && access(cpath.c_str(), R_OK) != 0)
            publish_cache_file(cpath, CODE_MAGIC, ckey, blob->code->data(), blob->code->size(), Hash128());
        if (!blob->code) {
            CacheLock clock(cpath);
            auto code = std::make_shared<std::vector<char>>();
            if (read_cache_file(cpath, CODE_MAGIC, ckey, *code, nullptr)) {
                if (from_cache) *from_cache = true;
            } else {
                if (from_cache) *from_cache = false;
#ifdef QGS_HIP_DEV_KNOBS
                if (const char *d = std::getenv("QGS_HIP_DUMP_SRC")) {          // keep the generated source
                    std::ofstream f(std::string(d) + "/" + cpath.substr(cpath.find_last_of('/') + 1) + ".hip");
                    f << g.source;
                }
#endif
                if (cc.helper) {
                    std::string why;
                    const int rc = compile_with_helper(g.source, arch, extra, *code, &why);
                    if (rc < 0) return -1;
                    if (rc == 1) {
                        // the helper is not usable (any more): this process compiles in-process from here on, under that compiler's identity
                        disable_helper(why);
                        continue;
                    }
                } else if (compile_in_process(g.source, arch, extra, *code)) return -1;
                publish_cache_file(cpath, CODE_MAGIC, ckey, code->data(), code->size(), Hash128());
            }
            blob->code = code;
            code_memo_put(ckey.hex(), code);
        } else if (from_cache) *from_cache = true;
        publish_struct(*blob, ckey);
        remember(blob);
        return 0;
    }
    return fail("no usable kernel compiler");
}

struct HostCsr {          // row-grouped tensor on the host, see generic_kernels.h DevTensor
    std::vector<int32_t> rowptr;
    std::vector<uint32_t> idx;
    std::vector<double> val;
    std::vector<uint32_t> idx2;   // rank 5 only
};

// One entry of the caller's tensor, rank 3 (l = m = 0) or rank 5: coordinates (i, j, k, l, m)
struct Entry {
    int i, j, k, l, m;
    double v;
};

// group entries by `row(t)`, keep the incoming (reference) order inside a row
template <class RowFn, class IdxFn>
HostCsr build_csr(int ndim, const std::vector<Entry> &ts, bool rank5, RowFn row, IdxFn idx)
{
    HostCsr c;
    c.rowptr.assign(ndim + 2, 0);
    for (const auto &t : ts) c.rowptr[row(t) + 1]++;
    for (int i = 0; i <= ndim; ++i) c.rowptr[i + 1] += c.rowptr[i];
    c.idx.resize(ts.size());
    c.val.resize(ts.size());
    if (rank5) c.idx2.resize(ts.size());
    std::vector<int32_t> pos(c.rowptr.begin(), c.rowptr.end() - 1);
    for (const auto &t : ts) {
        int p = pos[row(t)]++;
        c.idx[p] = idx(t);
        c.val[p] = t.v;
        if (rank5) c.idx2[p] = ((uint32_t)t.l << 16) | (uint32_t)t.m;
    }
    return c;
}

struct DevCsr {
    int32_t *rowptr = nullptr;
    uint32_t *idx = nullptr;
    double *val = nullptr;
    uint32_t *idx2 = nullptr;
    qgs::DevTensor view() const { return qgs::DevTensor{rowptr, idx, val, idx2}; }
};

struct Buffer {           // grow-only device scratch
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) return fail("hipMalloc of " + std::to_string(bytes) + " bytes failed");
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    double *f64() const { return (double *)p; }
};

// Small host-to-device uploads that must not stall the caller (the directed time grid and the tableau of every integration
// call; the Benettin loop makes one such call per re-orthonormalisation interval): the source is copied into a slot of a ring
// of page-locked blocks and leaves from there, so the copy is truly asynchronous, its source stays valid whatever the caller
// does with its own memory, and the host only ever waits for the upload made N uploads ago.
struct UploadRing {
    static const int N = 16;
    void *slot[N];
    size_t cap[N];
    hipEvent_t ev[N];
    bool used[N];
    unsigned next = 0;
    int last = -1;
    UploadRing() { for (int i = 0; i < N; ++i) { slot[i] = nullptr; cap[i] = 0; ev[i] = nullptr; used[i] = false; } }
    int stage(const void *src, size_t bytes, void *dst_dev, hipStream_t st)
    {
        const int i = (int)(next++ % N);
        if (!ev[i]) HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        if (used[i]) HIPCHK(hipEventSynchronize(ev[i]));
        if (cap[i] < bytes) {
            if (slot[i]) (void)hipHostFree(slot[i]);
            slot[i] = nullptr;
            cap[i] = 0;
            const size_t want = std::max<size_t>(bytes, 4096);
            HIPCHK(hipHostMalloc(&slot[i], want, hipHostMallocDefault));
            cap[i] = want;
        }
        std::memcpy(slot[i], src, bytes);
        HIPCHK(hipMemcpyAsync(dst_dev, slot[i], bytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(ev[i], st));
        used[i] = true;
        last = i;
        return 0;
    }
    void release()
    {
        for (int i = 0; i < N; ++i) {
            if (ev[i]) { if (used[i]) (void)hipEventSynchronize(ev[i]); (void)hipEventDestroy(ev[i]); }
            if (slot[i]) (void)hipHostFree(slot[i]);
            slot[i] = nullptr; ev[i] = nullptr; cap[i] = 0; used[i] = false;
        }
    }
};

// Host blocks THIS library has page-locked and mapped (qgs_host_alloc, qgs_host_register): the only host memory a kernel may store
// into.  What hipPointerGetAttributes says about a host address cannot be used for that decision: the runtime pins the pages
// of pageable hipMemcpy operands on its own and keeps those pins cached -- the source of a host-to-device copy READ-ONLY --
// and reports any later allocation that reuses such an address as "host" memory with a device pointer.  A result block that
// landed there was taken for page-locked, the unpack kernel stored into it and the process died with "Memory access fault by
// GPU ... Write access to a read-only page" (once in about ten runs of the GPU suite; pytest's capture hid the message).
std::mutex g_registered_mutex;
std::map<uintptr_t, size_t> g_registered;              // start -> bytes

void registry_add(const void *p, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_registered_mutex);
    g_registered[(uintptr_t)p] = bytes;
}
void registry_remove(const void *p)
{
    std::lock_guard<std::mutex> lock(g_registered_mutex);
    g_registered.erase((uintptr_t)p);
}
bool registry_covers(const void *p, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_registered_mutex);
    auto it = g_registered.upper_bound((uintptr_t)p);
    if (it == g_registered.begin()) return false;
    --it;
    return (uintptr_t)p >= it->first && (uintptr_t)p + bytes <= it->first + it->second;
}

// Copies between device memory and the CALLER's host memory.  A block this library page-locked itself (or the caller handed over
// with qgs_host_register) is copied asynchronously on `st`.  Any other host memory is pageable as far as the library knows and is
// never shown to the runtime: it is reached through the page-locked bounce blocks of host_bridge.cpp (DMA to / from a bounce
// block, gather / scatter by host threads), blocking -- which a copy with a pageable operand is anyway.  Rounds 1-4 handed such
// operands to hipMemcpy, which pins the caller's pages in place (read-only when they are the source); concurrent copies of
// that kind from the shard threads of a device group produced "Memory access fault by GPU ... Write access to a read-only page"
// about once in ten to twenty runs of the GPU suite.
int copy_with_host(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, const void *host_side, hipStream_t st)
{
    if (bytes == 0) return 0;
    if (registry_covers(host_side, bytes)) {
        // page-locked: asynchronous on the caller's stream (which the caller synchronises); a call without a stream of its
        // own is a blocking entry point and gets the blocking copy
        if (st) HIPCHK(hipMemcpyAsync(dst, src, bytes, kind, st));
        else HIPCHK(hipMemcpy(dst, src, bytes, kind));
        return 0;
    }
    std::string err;
    const int rc = kind == hipMemcpyHostToDevice ? qgs::bridge_h2d(dst, src, bytes, st, &err) : qgs::bridge_d2h(dst, src, bytes, st, &err);
    return rc ? fail(err) : 0;
}
int copy_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t st = nullptr)
{
    return copy_with_host(dst_dev, src_host, bytes, hipMemcpyHostToDevice, src_host, st);
}
int copy_d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t st = nullptr)
{
    return copy_with_host(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, dst_host, st);
}

struct KernelInfo {
    std::string name;
    int vgprs = 0, sgprs = 0, lds = 0, scratch = 0;
};

// Kernel-selection overrides (developer knobs), read from the environment ONCE when a model is created -- the
// launch paths only look at these fields.
struct LaunchTuning {
    int64_t wave_max_traj = -1;        // QGS_HIP_WAVE_MAX_TRAJ: largest ensemble on the wavefront-per-trajectory kernels (-1: measured crossovers)
    int64_t lds_tgl_min_pairs = 0;     // QGS_HIP_LDS_TGL_MIN_PAIRS
    int lds_force = -1;                // QGS_HIP_LDS=0|1: never / always the LDS-resident JIT kernels (-1: when cached or worth compiling)
    bool generic_simple = false;       // QGS_HIP_GENERIC=simple: never the tiled generic stepper
    int rk_variant = 0;                // QGS_HIP_RK_VARIANT=plain|split -> 1 | 2 (0: by ensemble size)
    bool rk_spread_rec = true;         // QGS_HIP_RK_SPREAD_REC=0: burst record stores also for write_steps == 1
    int64_t tgls_chunk = 0;            // QGS_HIP_TGLS_CHUNK: steps per trajectory / tangent pass pair (0: by the stage-record size)
    size_t tgl_share_min_bytes = (size_t)256 << 20;   // QGS_HIP_TGL_SHARE_MIN_MB
    bool tgl_plain = false;            // QGS_HIP_TGL_VARIANT=plain: never the shared-stage-state tangent kernel
    size_t window_bytes = (size_t)8 << 30;   // QGS_HIP_RECORD_WINDOW_MB: device memory the host-layout entry points spend on record windows
    bool window_by_hand = false;             //   (set by hand: the records of qgs_rk_integrate stay in windows of records, no member groups)
    int64_t group_members = 0;               // QGS_HIP_RECORD_GROUP_MEMBERS: members per group of qgs_rk_integrate's member groups (0: by rule)
    int d2h_mode = 0;                  // QGS_HIP_D2H=kernel|copy -> 1 | 2: records reach a page-locked host block by stores of the unpack kernel, or by a copy (0: by measurement)
    void read_env()
    {
        if (const char *e = std::getenv("QGS_HIP_WAVE_MAX_TRAJ")) wave_max_traj = std::atoll(e);
#ifdef QGS_HIP_DEV_KNOBS
        if (const char *e = std::getenv("QGS_HIP_LDS_TGL_MIN_PAIRS")) lds_tgl_min_pairs = std::atoll(e);
        if (const char *e = std::getenv("QGS_HIP_RK_SPREAD_REC")) rk_spread_rec = (*e == '1');
#endif
        if (const char *e = std::getenv("QGS_HIP_LDS")) lds_force = (*e == '1') ? 1 : 0;
        if (const char *e = std::getenv("QGS_HIP_GENERIC")) generic_simple = !std::strcmp(e, "simple");
        if (const char *e = std::getenv("QGS_HIP_RK_VARIANT")) rk_variant = !std::strcmp(e, "plain") ? 1 : (!std::strcmp(e, "split") ? 2 : 0);
        if (const char *e = std::getenv("QGS_HIP_TGLS_CHUNK")) tgls_chunk = std::max<int64_t>(1, std::atoll(e));
        if (const char *e = std::getenv("QGS_HIP_TGL_SHARE_MIN_MB")) tgl_share_min_bytes = (size_t)std::atoll(e) << 20;
        if (const char *e = std::getenv("QGS_HIP_TGL_VARIANT")) tgl_plain = !std::strcmp(e, "plain");
        if (const char *e = std::getenv("QGS_HIP_RECORD_WINDOW_MB")) { window_bytes = (size_t)std::max(1.0, std::atof(e) * 1048576.0); window_by_hand = true; }   // fractions allowed
        if (const char *e = std::getenv("QGS_HIP_RECORD_GROUP_MEMBERS")) group_members = std::max<int64_t>(64, (std::atoll(e) + 63) / 64 * 64);
        if (const char *e = std::getenv("QGS_HIP_D2H")) d2h_mode = !std::strcmp(e, "kernel") ? 1 : (!std::strcmp(e, "copy") ? 2 : 0);
    }
};

}  // namespace

struct DrainSlot {
    int64_t ticket = 0;
    hipEvent_t ev = nullptr;
};

struct qgs_model {
    int device = 0;
    int ndim = 0;
    int rank = 3;                 // 3: QgsTensor; 5: QgsTensorDynamicT / QgsTensorT4 (sparse_mul5 / sparse_mul4 path)
    std::string arch;
    std::vector<qgs::Term> T, J;  // terms of the specialised kernels; rank 5: over the derived-monomial index space (codegen.h)
    qgs::Derived der;
    // what the kernel generator sees: the canonical forms of T and J (magnitude-class ids instead of values, codegen.h), and the
    // hashes of those structures -- the model's part of the kernel-cache keys (a kernel reads one of the two tensors)
    qgs::Canonical canon_t, canon_j;
    Hash128 hash_t, hash_j;
    int64_t nnz_in = 0, jnnz_in = 0;
    DevCsr dT, dJ_by_i, dJ_by_j;
    // rank 5: the reduced tensors (two factors per term over variables + derived monomials) and the derived chains, for
    // the wavefront-per-trajectory kernels (generic_kernels.h DerivedChains); rank 3: unused, the kernels take dT / dJ_*
    DevCsr dT_red, dJ_red_by_i, dJ_red_by_j;
    int32_t *d_chain_t = nullptr, *d_chain_j = nullptr;     // [a | b | slot] packed, products sorted by level
    int n_der_t = 0, n_der_j = 0, n_lev_t = 0, n_lev_j = 0;
    int lev_ptr_t[qgs::WAVE_DER_LEVELS + 1] = {0, 0, 0, 0}, lev_ptr_j[qgs::WAVE_DER_LEVELS + 1] = {0, 0, 0, 0};
    bool wave_der_ok_t = true, wave_der_ok_j = true;        // the derived monomials fit the wave kernels' level scheme
    qgs::DerivedChains chains(bool jac) const
    {
        qgs::DerivedChains D{0, {0, 0, 0, 0}, nullptr, nullptr, nullptr};
        const int32_t *b = jac ? d_chain_j : d_chain_t;
        const int n = jac ? n_der_j : n_der_t;
        if (n == 0) return D;
        D.n_levels = jac ? n_lev_j : n_lev_t;
        for (int l = 0; l <= qgs::WAVE_DER_LEVELS; ++l) D.level_ptr[l] = jac ? lev_ptr_j[l] : lev_ptr_t[l];
        D.a = b; D.b = b + n; D.slot = b + 2 * n;
        return D;
    }
    qgs::DevTensor wave_T() const { return rank == 3 ? dT.view() : dT_red.view(); }
    qgs::DevTensor wave_J(bool adjoint) const
    {
        if (rank == 3) return adjoint ? dJ_by_j.view() : dJ_by_i.view();
        return adjoint ? dJ_red_by_j.view() : dJ_red_by_i.view();
    }
    // regrouped tendencies tensor for the tiled generic stepper (generic_kernels.h TiledTensor)
    int32_t *t_row_term = nullptr;
    uint32_t *t_term_joff = nullptr, *t_term_koff = nullptr;
    double *t_term_c = nullptr;
    int max_row_terms = 0;        // longest tendencies-tensor row (selects the register-resident wave kernel)
    int max_jrow_terms = 0;       // longest Jacobian-tensor row, by i or by j
    int t_terms_per_trip = 4, t_rpw = 16;
    int32_t *t_row_map = nullptr;
    qgs::TiledTensor tiled() const
    {
        return qgs::TiledTensor{t_row_term, t_term_joff, t_term_koff, t_term_c, t_terms_per_trip, t_row_map, t_rpw};
    }
    int kernel_kind = 0;          // 0 auto, 1 generic, 2 specialised
    int n_simd = 1024;            // SIMDs on the device (CUs x 4)
    bool spec_possible = false;
    bool spec_jac_possible = false;   // ... and the Jacobian / tangent kernels too (rank 5: their derived monomials fit as well)
    bool lds_spec_possible = false;   // too large for the register file, stage state fits LDS: JIT LDS-resident stepper
    bool prefer_lds = false;          // register-resident kernels exist but would spill (rank 5 with many derived monomials)
    mutable std::map<std::string, bool> lds_on_disk;   // kernel name -> code object found in the kernel cache (checked once)
    qgs::CodegenOptions cg;
    LaunchTuning tune;
    // compiled specialised kernels, one module per kernel (keyed by the kernel name)
    std::map<std::string, hipModule_t> modules;
    std::map<std::string, hipFunction_t> functions;
    std::vector<std::pair<qgs::Kernel, int>> loaded_kernels;      // (kind, stage count) of the specialised kernels loaded so far
    // staged time grid / tableau (uploads go through a ring of page-locked blocks; a stream other than the one of the last
    // upload waits for that upload's event before it reuses the staged tables)
    Buffer d_time, d_tab;
    std::vector<double> h_time, h_tab;
    UploadRing uploads;
    hipStream_t tab_stream = nullptr, time_stream = nullptr, tab_reader = nullptr;   // streams of the tables' last uploads / last readers
    int time_slot = -1, tab_slot = -1;                  // ring slots (events) of those uploads
    bool tab_reader_valid = false;
    // scratch
    Buffer work, stages, b_in_rows, b_in_modes, b_rec_modes, b_rec_rows, b_tg_rows, b_tg_modes, b_fm_modes, b_fm_rows, b_state2, b_tg2, b_ywork, b_vwork, b_mom_part, b_mom_out, b_unit, b_carry, b_win[2], b_fwin[2], b_drain;
    // host-layout pipeline: compute stream, copy stream, "window k computed" / "window k drained" events (created on first use)
    hipStream_t st_comp = nullptr, st_copy = nullptr;
    hipEvent_t ev_comp[2] = {nullptr, nullptr}, ev_copy[2] = {nullptr, nullptr};
    // single-state fast path of f / Df: page-locked staging block the kernels read and write directly
    double *h_pin = nullptr, *d_pin = nullptr;
    size_t pin_cap = 0;
    int magnitude_ulp = qgs::DEFAULT_MAGNITUDE_ULP;     // coefficient classes of the specialised kernels (QGS_HIP_MAGNITUDE_ULP, codegen.h Canonical)
    std::vector<int64_t> drain_tickets;                // windows on their way into pageable host memory (host_bridge.h), oldest first
    std::map<const Buffer *, DrainSlot> drain_slots;   // per staging block: its window in flight, the event that marks its unpack
    std::vector<std::unique_ptr<Buffer>> drain_pool;   // qgs_unpack_window_enqueue: staging blocks, reused as their windows leave the device
    unsigned *d_one_counter = nullptr;                 // "workgroups finished" word of the single-state kernels
    unsigned long long one_seq = 0;                    // sequence number of the last single-state call (the kernel echoes it into h_pin[0])
    // Jacobian tensor grouped by output element (generic_kernels.h OnePairs), models of up to 1024 variables
    int32_t *p_lut = nullptr, *p_ptr = nullptr;
    uint32_t *p_idx = nullptr, *p_idx2 = nullptr;
    double *p_val = nullptr;
    int64_t last_windows = 0;      // windows of the last host-layout integration (qgs_model_info 8)
    int64_t last_groups = 1;       // member groups of the last qgs_rk_integrate (qgs_model_info 9)
    KernelInfo last;
};

namespace {

// bytes of LDS the LDS-resident tangent kernel needs: stage state (+ derived monomials) of the tile's members (16, or 8 when
// 16 do not fit), tangent vector of 64 pairs
size_t lds_tgl_bytes(const qgs_model *m, int members = 0)
{
    if (members == 0) members = m->cg.lds_tgl_members;
    return ((size_t)m->ndim + m->der.j.size()) * 8 * (size_t)members + (size_t)m->ndim * 512;
}

// Which specialised kernel families a model can have (shared by qgs_model_create_rank and qgs_prebuild_rank).
//   register-resident: ndim <= 64 and (rank 5) at most QGS_SPEC_MAX_DERIVED derived monomials per evaluation
//   LDS-resident: stage state + derived monomials fit one workgroup's LDS; used when the register kernels do not exist,
//   or when they exist but must spill (more than QGS_PREFER_LDS_DERIVED derived monomials: the T4 model keeps 111 pair
//   products alive next to its 38 variables; measured 205 ms vs the LDS kernel for 65 536 members x 100 steps)
#ifndef QGS_PREFER_LDS_DERIVED
#define QGS_PREFER_LDS_DERIVED 24
#endif
void classify_model(qgs_model *m)
{
    const size_t ndim = (size_t)m->ndim, nt = m->der.t.size(), nj = m->der.j.size();
    m->spec_possible = (m->ndim <= QGS_SPEC_MAX_NDIM) && nt <= QGS_SPEC_MAX_DERIVED;
    m->spec_jac_possible = m->spec_possible && nj <= QGS_SPEC_MAX_DERIVED;
    m->prefer_lds = m->spec_possible && nt > QGS_PREFER_LDS_DERIVED;
#ifdef QGS_HIP_DEV_KNOBS
    if (const char *e = std::getenv("QGS_HIP_PREFER_LDS")) m->prefer_lds = m->spec_possible && (*e == '1');
#endif
    const bool fits = (ndim + nt) * 512 <= (size_t)QGS_LDS_STATE_BYTES && m->T.size() <= 200000;
    m->lds_spec_possible = fits && (!m->spec_possible || m->prefer_lds);
    if (const char *e = std::getenv("QGS_HIP_LDS_TGL_MEMBERS")) m->cg.lds_tgl_members = (std::atoi(e) == 8) ? 8 : 16;
    else m->cg.lds_tgl_members = (lds_tgl_bytes(m, 16) > (size_t)QGS_LDS_STATE_BYTES) ? 8 : 16;
}


int upload_csr(const HostCsr &h, DevCsr &d)
{
    HIPCHK(hipMalloc((void **)&d.rowptr, sizeof(int32_t) * h.rowptr.size()));
    HIPCHK(hipMalloc((void **)&d.idx, sizeof(uint32_t) * std::max<size_t>(1, h.idx.size())));
    HIPCHK(hipMalloc((void **)&d.val, sizeof(double) * std::max<size_t>(1, h.val.size())));
    if (copy_h2d(d.rowptr, h.rowptr.data(), sizeof(int32_t) * h.rowptr.size())) return -1;
    if (!h.idx.empty()) {
        if (copy_h2d(d.idx, h.idx.data(), sizeof(uint32_t) * h.idx.size())) return -1;
        if (copy_h2d(d.val, h.val.data(), sizeof(double) * h.val.size())) return -1;
    }
    if (!h.idx2.empty()) {
        HIPCHK(hipMalloc((void **)&d.idx2, sizeof(uint32_t) * h.idx2.size()));
        if (copy_h2d(d.idx2, h.idx2.data(), sizeof(uint32_t) * h.idx2.size())) return -1;
    }
    return 0;
}

template <class T>
int upload_vec(const std::vector<T> &h, T **d)
{
    HIPCHK(hipMalloc((void **)d, sizeof(T) * std::max<size_t>(1, h.size())));
    if (!h.empty() && copy_h2d(*d, h.data(), sizeof(T) * h.size())) return -1;
    return 0;
}

// rows -> flat term stream (reference (j,k) order kept), each row padded to a multiple of 4 terms with
// zero-coefficient terms that read slot 0; offsets are LDS byte offsets (generic_kernels.h TiledTensor)
int upload_tiled(qgs_model *m, const std::vector<Entry> &Tr)
{
    const int ndim = m->ndim;
    std::vector<std::vector<const Entry *>> by_row(ndim + 2);
    for (const auto &t : Tr) by_row[t.i].push_back(&t);
    std::vector<int32_t> row_term(ndim + 2, 0);
    std::vector<uint32_t> joff, koff;
    std::vector<double> c;
    // long rows (MAOOAM 6x6: ~120 terms) take 16 terms per loop trip to amortise the scalar-load latency
    const size_t pad = (Tr.size() >= (size_t)32 * ndim) ? 16 : 4;
    m->t_terms_per_trip = (int)pad;
    for (int i = 0; i <= ndim; ++i) {
        row_term[i] = (int32_t)c.size();
        for (const Entry *t : by_row[i]) {
            joff.push_back((uint32_t)t->j * 512u);
            koff.push_back((uint32_t)t->k * 512u);
            c.push_back(t->v);
        }
        while (c.size() % (size_t)m->t_terms_per_trip) { joff.push_back(0); koff.push_back(0); c.push_back(0.0); }
    }
    row_term[ndim + 1] = (int32_t)c.size();
    // rows -> (wavefront, slot): longest-processing-time greedy over the padded term counts, 16 wavefronts
    const int NW = 16;
    int rpw = 2;
    while (rpw < 16 && rpw * NW < ndim) rpw *= 2;
    m->t_rpw = rpw;
    std::vector<int32_t> row_map((size_t)NW * rpw, 0);
    if (rpw * NW >= ndim) {
        std::vector<int> order(ndim);
        for (int i = 0; i < ndim; ++i) order[i] = i + 1;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
            return (row_term[x + 1] - row_term[x]) > (row_term[y + 1] - row_term[y]);
        });
        std::vector<int64_t> load(NW, 0);
        std::vector<int> used(NW, 0);
        for (int row : order) {
            int best = -1;
            for (int w = 0; w < NW; ++w)
                if (used[w] < rpw && (best < 0 || load[w] < load[best])) best = w;
            row_map[(size_t)best * rpw + used[best]++] = row;
            load[best] += row_term[row + 1] - row_term[row];
        }
    }
    if (upload_vec(row_map, &m->t_row_map)) return -1;
    if (upload_vec(row_term, &m->t_row_term) || upload_vec(joff, &m->t_term_joff) || upload_vec(koff, &m->t_term_koff) ||
        upload_vec(c, &m->t_term_c)) return -1;
    return 0;
}

// Jacobian tensor grouped by output element (i, j) for the single-state Df kernel (generic_kernels.h OnePairs): entries keep
// their incoming order inside a pair.  Only for ndim <= QGS_ONE_MAX_NDIM (the lookup table is ndim^2 words).
#ifndef QGS_ONE_MAX_NDIM
#define QGS_ONE_MAX_NDIM 1024
#endif
int upload_pairs(qgs_model *m, const std::vector<Entry> &Jr, bool rank5)
{
    const int ndim = m->ndim;
    if (Jr.empty() || ndim > QGS_ONE_MAX_NDIM) return 0;
    std::vector<int32_t> lut((size_t)ndim * ndim, -1), count;
    for (const auto &t : Jr) {
        int32_t &p = lut[(size_t)(t.i - 1) * ndim + (t.j - 1)];
        if (p < 0) { p = (int32_t)count.size(); count.push_back(0); }
        count[(size_t)p]++;
    }
    // pairs numbered in row-major order of (i, j): neighbouring threads walk neighbouring entries
    std::vector<int32_t> ptr(1, 0);
    for (auto &p : lut)
        if (p >= 0) { const int32_t c = count[(size_t)p]; p = (int32_t)ptr.size() - 1; ptr.push_back(ptr.back() + c); }
    std::vector<int32_t> pos(ptr.begin(), ptr.end() - 1);
    std::vector<uint32_t> idx(Jr.size()), idx2(rank5 ? Jr.size() : 0);
    std::vector<double> val(Jr.size());
    for (const auto &t : Jr) {
        const int32_t e = pos[(size_t)lut[(size_t)(t.i - 1) * ndim + (t.j - 1)]]++;
        idx[(size_t)e] = (uint32_t)t.k;
        val[(size_t)e] = t.v;
        if (rank5) idx2[(size_t)e] = ((uint32_t)t.l << 16) | (uint32_t)t.m;
    }
    if (upload_vec(lut, &m->p_lut) || upload_vec(ptr, &m->p_ptr) || upload_vec(idx, &m->p_idx) || upload_vec(val, &m->p_val)) return -1;
    if (rank5 && upload_vec(idx2, &m->p_idx2)) return -1;
    return 0;
}

// Derived monomials sorted into levels for the wave kernels (generic_kernels.h DerivedChains).  *ok = false when the
// scheme does not fit (more than WAVE_DER_LEVELS levels or more products in a level than the workgroup can hold).
int upload_levels(int ndim, const std::vector<std::pair<int, int>> &der, int32_t **d_out, int *n_levels, int *level_ptr, bool *ok)
{
    const int nd = (int)der.size();
    std::vector<int> level(nd, 1);
    int nl = nd ? 1 : 0;
    for (int n = 0; n < nd; ++n) {                           // a derived value only refers to earlier ones
        for (int f : {der[n].first, der[n].second}) if (f > ndim) level[n] = std::max(level[n], level[f - ndim - 1] + 1);
        nl = std::max(nl, level[n]);
    }
    const int threads = 64 * ((ndim + 63) / 64);
    *ok = nl <= qgs::WAVE_DER_LEVELS;
    std::vector<int32_t> a, b, slot;
    for (int l = 0; l <= qgs::WAVE_DER_LEVELS; ++l) level_ptr[l] = 0;
    for (int l = 1; l <= std::min(nl, (int)qgs::WAVE_DER_LEVELS); ++l) {
        for (int n = 0; n < nd; ++n)
            if (level[n] == l) { a.push_back(der[n].first); b.push_back(der[n].second); slot.push_back(ndim + 1 + n); }
        level_ptr[l] = (int)a.size();
        if (level_ptr[l] - level_ptr[l - 1] > threads * qgs::WAVE_DER_PER) *ok = false;
    }
    for (int l = nl + 1; l <= qgs::WAVE_DER_LEVELS; ++l) level_ptr[l] = level_ptr[std::max(nl, 0)];
    *n_levels = std::min(nl, (int)qgs::WAVE_DER_LEVELS);
    if (!*ok) { a.clear(); b.clear(); slot.clear(); }
    std::vector<int32_t> packed(a);
    packed.resize(3 * (size_t)nd, 0);
    if (*ok) {
        std::copy(b.begin(), b.end(), packed.begin() + nd);
        std::copy(slot.begin(), slot.end(), packed.begin() + 2 * (size_t)nd);
    }
    return upload_vec(packed, d_out);
}

// rank 5: reduced tensors for the wavefront-per-trajectory kernels
int upload_reduced(qgs_model *m)
{
    const int ndim = m->ndim;
    auto pack = [](int a, int b) { return ((uint32_t)a << 16) | (uint32_t)b; };
    std::vector<Entry> Tr, Jr;
    for (const auto &t : m->T) if (t.i >= 1) Tr.push_back(Entry{t.i, t.j, t.k, 0, 0, t.v});
    for (const auto &t : m->J) if (t.i >= 1 && t.j >= 1) Jr.push_back(Entry{t.i, t.j, t.k, 0, 0, t.v});
    m->max_row_terms = m->max_jrow_terms = 0;
    {
        std::vector<int> cnt(ndim + 2, 0), ci(ndim + 2, 0), cj(ndim + 2, 0);
        for (const auto &t : Tr) m->max_row_terms = std::max(m->max_row_terms, ++cnt[t.i]);
        for (const auto &t : Jr) m->max_jrow_terms = std::max(m->max_jrow_terms, std::max(++ci[t.i], ++cj[t.j]));
    }
    HostCsr hT = build_csr(ndim, Tr, false, [](const Entry &t) { return t.i; }, [&](const Entry &t) { return pack(t.j, t.k); });
    HostCsr hJi = build_csr(ndim, Jr, false, [](const Entry &t) { return t.i; }, [&](const Entry &t) { return pack(t.j, t.k); });
    HostCsr hJj = build_csr(ndim, Jr, false, [](const Entry &t) { return t.j; }, [&](const Entry &t) { return pack(t.i, t.k); });
    if (upload_csr(hT, m->dT_red) || upload_csr(hJi, m->dJ_red_by_i) || upload_csr(hJj, m->dJ_red_by_j)) return -1;
    m->n_der_t = (int)m->der.t.size();
    m->n_der_j = (int)m->der.j.size();
    if (upload_levels(ndim, m->der.t, &m->d_chain_t, &m->n_lev_t, m->lev_ptr_t, &m->wave_der_ok_t)) return -1;
    if (upload_levels(ndim, m->der.j, &m->d_chain_j, &m->n_lev_j, m->lev_ptr_j, &m->wave_der_ok_j)) return -1;
    return 0;
}

void free_csr(DevCsr &d)
{
    if (d.rowptr) (void)hipFree(d.rowptr);
    if (d.idx) (void)hipFree(d.idx);
    if (d.val) (void)hipFree(d.val);
    if (d.idx2) (void)hipFree(d.idx2);
    d = DevCsr();
}

// identity of kernel `k` of this model's structure in the kernel cache (obtain_blob adds compiler, flags, architecture)
std::string kernel_key(const qgs_model *m, qgs::Kernel k, int S)
{
    return qgs::kernel_name(k, S, m->cg) + "|" + qgs::options_signature(m->cg) + "|ulp=" + std::to_string(m->magnitude_ulp) + "|ndim=" +
           std::to_string(m->ndim) + "|" +
           (qgs::kernel_uses_jacobian(k) ? "J" + m->hash_j.hex() : "T" + m->hash_t.hex());
}

int model_blob(const qgs_model *m, qgs::Kernel k, int S, std::shared_ptr<const KernelBlob> *out, bool *from_cache, BlobMode mode = BlobMode::Use)
{
    return obtain_blob(kernel_key(m, k, S), m->arch, qgs::kernel_compile_flags(k),
                       [&] { return qgs::generate_kernel(m->ndim, m->canon_t.terms, m->canon_j.terms, k, S, m->cg, m->der); }, out, from_cache,
                       mode);
}

// Load a blob on the current device and store THIS model's coefficients into the module's tables.
int load_blob(qgs_model *m, const std::string &fname, const KernelBlob &blob, const qgs::Canonical &canon, hipFunction_t *fn)
{
    hipModule_t mod;
    HIPCHK(hipModuleLoadData(&mod, blob.code->data()));
    m->modules[fname] = mod;
    std::vector<double> values;
    for (const qgs::CoefTable &t : blob.tables) {
        hipDeviceptr_t dptr = nullptr;
        size_t bytes = 0;
        hipError_t e = hipModuleGetGlobal(&dptr, &bytes, mod, t.symbol.c_str());
        if (e != hipSuccess) return fail("coefficient table " + t.symbol + " not found in its module: " + hipGetErrorString(e));
        if (bytes != t.values.size() * sizeof(double))
            return fail("coefficient table " + t.symbol + ": the module has " + std::to_string(bytes) + " bytes, the generator " +
                        std::to_string(t.values.size() * sizeof(double)));
        try {
            canon.decode(t.values, values);
        } catch (const std::exception &ex) {
            return fail(std::string("coefficient table ") + t.symbol + ": " + ex.what());
        }
        if (copy_h2d((void *)dptr, values.data(), bytes)) return -1;
    }
    hipFunction_t f;
    hipError_t e = hipModuleGetFunction(&f, mod, fname.c_str());
    if (e != hipSuccess) return fail("kernel " + fname + " not found in its module: " + hipGetErrorString(e));
    m->functions[fname] = f;
    *fn = f;
    return 0;
}

// Make sure kernel `k` (for S stages) is generated, compiled (or fetched from the cache) and loaded.
int get_function(qgs_model *m, qgs::Kernel k, int S, hipFunction_t *fn, std::string *name_out = nullptr)
{
    const std::string fname = qgs::kernel_name(k, S, m->cg);
    if (name_out) *name_out = fname;
    auto it = m->functions.find(fname);
    if (it != m->functions.end()) { *fn = it->second; return 0; }
    std::shared_ptr<const KernelBlob> blob;
    bool cached = false;
    if (model_blob(m, k, S, &blob, &cached)) return -1;
    if (load_blob(m, fname, *blob, qgs::kernel_uses_jacobian(k) ? m->canon_j : m->canon_t, fn)) return -1;
    m->loaded_kernels.push_back({k, S});
    return 0;
}

void note_kernel(qgs_model *m, const std::string &name, hipFunction_t f)
{
    m->last.name = name;
    if (f) {
        int v = 0;
        if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_NUM_REGS, f) == hipSuccess) m->last.vgprs = v;
        if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, f) == hipSuccess) m->last.lds = v;
        if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f) == hipSuccess) m->last.scratch = v;
        m->last.sgprs = 0;
    } else {
        m->last.vgprs = m->last.sgprs = m->last.lds = m->last.scratch = 0;
    }
}

// grid of the one-wavefront-per-(64 members, column) tangent kernels (codegen emit_tgl_kernel: XCD-aware order when ld % 64 == 0)
int launch_tgl(hipFunction_t f, int64_t ld, int64_t n_tg, hipStream_t st, void **args)
{
    unsigned blocks = (unsigned)((n_tg * ld + 63) / 64);
    if ((ld & 63) == 0) blocks = (unsigned)(8 * (((ld >> 6) + 7) / 8) * n_tg);
    HIPCHK(hipModuleLaunchKernel(f, blocks, 1, 1, 64, 1, 1, 0, st, args, nullptr));
    return 0;
}

int launch(hipFunction_t f, int64_t lanes, hipStream_t st, void **args)
{
    const unsigned blocks = (unsigned)((lanes + 63) / 64);
    HIPCHK(hipModuleLaunchKernel(f, blocks, 1, 1, 64, 1, 1, 0, st, args, nullptr));
    return 0;
}

// Stage the directed time grid (integrate.py:199-202) and the tableau on the device; both are cached.
//   tab layout: [ b[s], a_sub[s-1] | b[s], a[s*s] ]   (specialised part first, generic part after it)
int stage_time_tab(qgs_model *m, const double *time, int64_t n_time, int direction, int s, const double *b,
                   const double *a, hipStream_t st, const double **d_time, const double **d_tab_spec,
                   const double **d_tab_full)
{
    std::vector<double> dt(time, time + n_time);
    if (direction == -1) std::reverse(dt.begin(), dt.end());
    std::vector<double> tab;
    tab.insert(tab.end(), b, b + s);
    for (int i = 1; i < s; ++i) tab.push_back(a[i * s + (i - 1)]);
    tab.insert(tab.end(), b, b + s);
    tab.insert(tab.end(), a, a + (size_t)s * s);
    const bool new_time = dt != m->h_time, new_tab = tab != m->h_tab;
    // Two streams may use the cached tables (the compute stream of the host-layout calls, a caller's stream of the *_device calls).
    // A table is overwritten only after the kernels of the stream that read it last have finished (they may still be running on
    // another stream than this call's), and a stream that did not stage a table itself waits for that table's upload: each table
    // keeps the ring slot (event) and the stream of its last upload.  (An event of a ring slot that has been reused since is
    // harmless to wait for: the ring reuses a slot only after the host has seen its previous upload complete.)
    if ((new_time || new_tab) && m->tab_reader_valid && m->tab_reader != st) HIPCHK(hipStreamSynchronize(m->tab_reader));
    if (new_time) {
        if (m->d_time.ensure(sizeof(double) * (size_t)n_time)) return -1;
        if (m->uploads.stage(dt.data(), sizeof(double) * (size_t)n_time, m->d_time.p, st)) return -1;
        m->h_time.swap(dt);
        m->time_slot = m->uploads.last;
        m->time_stream = st;
    } else if (m->time_slot >= 0 && st != m->time_stream) {
        HIPCHK(hipStreamWaitEvent(st, m->uploads.ev[m->time_slot], 0));
    }
    if (new_tab) {
        if (m->d_tab.ensure(sizeof(double) * tab.size())) return -1;
        if (m->uploads.stage(tab.data(), sizeof(double) * tab.size(), m->d_tab.p, st)) return -1;
        m->h_tab.swap(tab);
        m->tab_slot = m->uploads.last;
        m->tab_stream = st;
    } else if (m->tab_slot >= 0 && st != m->tab_stream) {
        HIPCHK(hipStreamWaitEvent(st, m->uploads.ev[m->tab_slot], 0));
    }
    m->tab_reader = st;                     // the kernels of this call read both tables on st
    m->tab_reader_valid = true;
    *d_time = m->d_time.f64();
    *d_tab_spec = m->d_tab.f64();
    *d_tab_full = m->d_tab.f64() + (2 * s - 1);
    return 0;
}

// Generator knobs.  The shipped defaults are what was measured fastest.  A normal build reads one of them (exercised by the
// parity tests); the experiment knobs of DESIGN.md section 3 exist only in a developer build (`make DEV=1`,
// -DQGS_HIP_DEV_KNOBS), where their variants can be re-measured.
void apply_env_options(qgs::CodegenOptions &cg)
{
    if (const char *e = std::getenv("QGS_HIP_TGL_PAIR")) cg.tgl_pair = (*e == '1');
#ifdef QGS_HIP_DEV_KNOBS
    if (const char *e = std::getenv("QGS_HIP_ROW_SPLIT")) cg.row_split = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("QGS_HIP_TGL_SHARE_X")) cg.tgl_share_x = std::min(16, std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("QGS_HIP_INTERLEAVE")) cg.interleave = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("QGS_HIP_LDS_ASM")) cg.lds_asm = (*e == '1');
    if (const char *e = std::getenv("QGS_HIP_LDS_ASM_WAVES")) cg.lds_asm_waves = std::min(16, std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("QGS_HIP_LDS_ASM_CAP")) cg.lds_asm_cap = std::max(2, std::atoi(e));
    if (const char *e = std::getenv("QGS_HIP_LDS_ASM_PP")) cg.lds_asm_pingpong = (*e == '1');
    if (const char *e = std::getenv("QGS_HIP_LDS_ASM_LANES")) cg.lds_asm_lanes = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("QGS_HIP_LDS_ASM_CHUNK")) cg.lds_asm_chunk = std::atoi(e);
    if (const char *e = std::getenv("QGS_HIP_LDS_ASM_COEF")) cg.lds_asm_coef = std::atoi(e);
    if (const char *e = std::getenv("QGS_HIP_LDS_ASM_RING")) cg.lds_asm_ring = std::max(2, std::atoi(e));
    if (const char *e = std::getenv("QGS_HIP_LDS_ASM_MINCAP")) cg.lds_asm_mincap = std::max(4, std::atoi(e));
    if (const char *e = std::getenv("QGS_HIP_LDS_ASM_VFREE")) cg.lds_asm_vfree = std::atoi(e);
    if (const char *e = std::getenv("QGS_HIP_LDS_ASM_SFREE")) cg.lds_asm_sfree = std::atoi(e);
    if (const char *e = std::getenv("QGS_HIP_LDS_WAVES")) cg.lds_waves = std::min(16, std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("QGS_HIP_LDS_CAP")) cg.lds_cap = std::max(2, std::atoi(e));
    if (const char *e = std::getenv("QGS_HIP_LDS_GROUP")) cg.lds_group = (*e == '1');
    if (const char *e = std::getenv("QGS_HIP_LDS_YLOAD")) cg.lds_yload_ahead = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("QGS_HIP_LDS_ORDER")) cg.lds_order = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("QGS_HIP_LDS_DEDUPE")) cg.lds_coeff_dedupe = (*e == '1');
    if (const char *e = std::getenv("QGS_HIP_TGL_DEDUPE")) cg.tgl_coeff_dedupe = (*e == '1');
    if (const char *e = std::getenv("QGS_HIP_TGL_PARK_V")) cg.tgl_park_v = (*e == '1');
    if (const char *e = std::getenv("QGS_HIP_TGL_INTERLEAVE")) cg.tgl_interleave = std::max(1, std::atoi(e));
#else
    (void)cg;
#endif
}

// QGS_HIP_MAGNITUDE_ULP (a normal-build knob): coefficients of a tensor within this many units in the last place
// are one magnitude class of the specialised kernels (default 2); 0 = every distinct value is its own class.
int magnitude_ulp_from_env()
{
    if (const char *e = std::getenv("QGS_HIP_MAGNITUDE_ULP")) {
        char *end = nullptr;
        const long v = std::strtol(e, &end, 10);
        if (end != e && v >= 0 && v <= 64) return (int)v;
    }
    return qgs::DEFAULT_MAGNITUDE_ULP;
}

// Layout of the shape-specialised batched QR (codegen.h QrPlan); a developer build can re-measure the alternatives.
qgs::QrPlan qr_plan_for(int n_rows, int n_cols)
{
    int members = 0, slots = 0;
#ifdef QGS_HIP_DEV_KNOBS
    if (const char *e = std::getenv("QGS_HIP_QR_MEMBERS")) members = std::atoi(e);
    if (const char *e = std::getenv("QGS_HIP_QR_SLOTS")) slots = std::max(1, std::min(8, std::atoi(e)));
#endif
    qgs::QrPlan p = qgs::qr_plan(n_rows, n_cols, members, slots);
#ifdef QGS_HIP_DEV_KNOBS
    if (const char *e = std::getenv("QGS_HIP_QR_CHAINS")) p.chains = std::max(1, std::min(8, std::atoi(e)));
    if (const char *e = std::getenv("QGS_HIP_QR_RELOAD")) p.reload = (*e == '1');
#endif
    return p;
}

// explicit scheme: a[i][j] == 0 for j >= i
bool tableau_is_lower_triangular(int s, const double *a)
{
    for (int i = 0; i < s; ++i)
        for (int j = i; j < s; ++j)
            if (a[i * s + j] != 0.0) return false;
    return true;
}

bool use_spec(const qgs_model *m, int s, const double *a)
{
    if (m->kernel_kind == 1) return false;
    if (!m->spec_possible) return false;
    if (s < 1 || s > 8) return false;
    return a == nullptr || qgs::tableau_is_subdiagonal(s, a);
}

// wavefront-per-trajectory stepper for small ensembles: below QGS_HIP_WAVE_MAX_TRAJ members (default 4096, see
// DESIGN.md 3.5) it beats one-member-per-lane because the lanes of the few wavefronts would do all rows serially
bool use_wave(const qgs_model *m, int64_t n_traj, int s, const double *a)
{
    if (m->kernel_kind != 0) return false;                 // explicit generic / specialised request
    // measured crossovers (tools/latency_bench.py): rows in registers (<= 16 / <= 32 terms) 2048 / 1024 members; longer rows
    // run the lane-group kernel (G lanes per row, terms streamed), which beats the LDS-resident JIT stepper up to ~600
    // members (MAOOAM 6x6: 29 vs 102 us per RK4 step for one trajectory, 8.0 vs 20.4 ms per 200 steps at 256 members,
    // 30 vs 20 ms at 1024; T4 MAOOAM: 12 vs 27 us per step)
    int64_t limit = (m->max_row_terms <= 16) ? 2048 : (m->max_row_terms <= 32 ? 1024 : 512);
    if (m->tune.wave_max_traj >= 0) limit = m->tune.wave_max_traj;
    return n_traj <= limit && s >= 1 && s <= 8 && qgs::wave_supported(m->ndim + (int)m->der.t.size()) && m->wave_der_ok_t &&
           qgs::tableau_is_subdiagonal(s, a);
}

// below this many (member, column) pairs the wavefront-per-pair kernel would be preferred to the LDS-resident tangent
// kernel; measured (tools/tgls228.py): the LDS-resident one wins at every size (1 x 228 pairs: 3.8 vs 5.5 ms per 10 steps)
int64_t lds_tgl_min_pairs(const qgs_model *m) { return m->tune.lds_tgl_min_pairs; }

// wavefront-per-(member, column) tangent kernel: against the specialised lane kernel it wins below 4096 pairs,
// against the simple generic kernel (large ndim, latency-bound at ~350 ms per 10 steps) up to ~16k pairs
bool use_tgl_wave(const qgs_model *m, int64_t pairs, int s, const double *a)
{
    if (m->kernel_kind != 0) return false;
    int64_t limit = m->spec_possible ? 4096 : 16384;
    if (m->tune.wave_max_traj >= 0) limit = m->tune.wave_max_traj;
    return pairs <= limit && s >= 1 && s <= 8 && qgs::wave_supported(m->ndim + (int)m->der.j.size()) && m->wave_der_ok_j &&
           qgs::tableau_is_subdiagonal(s, a);
}

// JIT LDS-resident stepper for systems beyond the register file (codegen.cpp emit_rk_lds_kernel).  Compiling it takes
// about 20 s for MAOOAM 6x6 (once: the code object is cached on disk), so in auto mode it is used when the code object
// is already there, for runs long enough to pay for the compilation, or when requested with qgs_model_set_kernel(m, 2).
bool lds_kernel_wanted(const qgs_model *m, qgs::Kernel k, double work)
{
    if (m->tune.lds_force >= 0) return m->tune.lds_force == 1;
    if (m->kernel_kind == 2) return true;
    if (m->prefer_lds) return true;         // the alternative is a register-resident kernel that spills and takes minutes to compile
    const std::string name = qgs::kernel_name(k, 0, m->cg);
    if (m->functions.count(name)) return true;                                              // already loaded
    auto it = m->lds_on_disk.find(name);
    if (it == m->lds_on_disk.end()) {
        std::shared_ptr<const KernelBlob> blob;
        it = m->lds_on_disk.emplace(name, model_blob(m, k, 0, &blob, nullptr, BlobMode::Lookup) == 0).first;      // looks, never compiles
    }
    if (it->second) return true;                                                            // built earlier (qgs_prebuild / a previous run)
    return work >= 2e12;                                                                    // ~10 s of the generic kernels
}

bool use_lds_spec(const qgs_model *m, int64_t n_traj, int64_t n_steps, int s, const double *a)
{
    if (m->kernel_kind == 1 || !m->lds_spec_possible) return false;
    if (s < 1 || s > 64 || !qgs::tableau_is_subdiagonal(s, a)) return false;
    return lds_kernel_wanted(m, qgs::Kernel::RkLds, (double)n_traj * (double)n_steps * (double)s * (double)m->T.size());
}

// LDS-resident tangent / adjoint kernel (codegen.cpp emit_tgl_lds_kernel): stage state of 16 members + tangent vector of
// 64 pairs in LDS, i.e. ndim * 640 B
bool use_lds_tgl(const qgs_model *m, int64_t pairs, int64_t n_steps, int s, const double *a, int adjoint)
{
    if (m->kernel_kind == 1 || !m->lds_spec_possible || m->J.empty()) return false;
    if (lds_tgl_bytes(m) > (size_t)QGS_LDS_STATE_BYTES) return false;
    if (s < 1 || s > 64 || !qgs::tableau_is_subdiagonal(s, a)) return false;
    return lds_kernel_wanted(m, adjoint ? qgs::Kernel::AdjLds : qgs::Kernel::TglLds,
                             (double)pairs * (double)n_steps * (double)s * (double)m->J.size());
}

// tiled generic stepper: sub-diagonal tableau, stage state fits one workgroup's LDS
bool use_tiled(const qgs_model *m, int s, const double *a)
{
    if (m->tune.generic_simple) return false;
    if (m->rank != 3) return false;                                   // the tiled stream holds two factors per term
    return s >= 1 && s <= 8 && qgs::tiled_supported(m->ndim) && qgs::tableau_is_subdiagonal(s, a);
}

// LDS-resident JIT stepper launch (codegen.cpp emit_rk_lds_kernel): W wavefronts per 64 members
int launch_rk_lds(qgs_model *m, int64_t n_traj, int64_t ld, const double *y_in, double *y_out, double *d_rec, double *stages,
                  const double *d_time, const double *d_tab, int64_t step_begin, int64_t step_end, int64_t write_steps,
                  int64_t n_records, int backward, int write_final, int s, hipStream_t st, qgs::Kernel which = qgs::Kernel::RkLds)
{
    hipFunction_t f;
    std::string name;
    if (get_function(m, which, 0, &f, &name)) return -1;
    const int64_t blocks = (n_traj + 63) / 64;
    if (m->b_ywork.ensure(sizeof(double) * (size_t)m->ndim * 64 * (size_t)blocks)) return -1;
    double *yw = m->b_ywork.f64();
    long long nt = n_traj, l = ld, sb = step_begin, se = step_end, ws = write_steps, nr = n_records;
    int bw = backward, wf = write_final, S = s, one = 1;
    void *args[] = {(void *)&y_in, &y_out, &yw, &d_rec, &stages, (void *)&d_time, (void *)&d_tab,
                    &nt, &l, &sb, &se, &ws, &nr, &bw, &wf, &S, &one};      // `one`: the extra argument of the tendencies-only flavour
    note_kernel(m, name, f);
    // (the hand-scheduled stepper has a workgroup shape of its own; the tendencies-only flavour is always the compiler-scheduled one)
    const int waves = (which == qgs::Kernel::RkLds && m->cg.lds_asm) ? m->cg.lds_asm_waves : m->cg.lds_waves;
    HIPCHK(hipModuleLaunchKernel(f, (unsigned)blocks, 1, 1, 64 * waves, 1, 1, 0, st, args, nullptr));
    return 0;
}

// LDS-resident tangent / adjoint launch: one workgroup per tile of 16 members x 4 columns
int launch_tgl_lds(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_tg, const double *w_in, double *w_out, double *d_rec_fm,
                   const double *stages, const double *d_time, const double *d_tab, int64_t step_begin, int64_t step_end,
                   int64_t write_steps, int64_t n_records, int backward, int write_final, int adjoint, double inverse, int s,
                   hipStream_t st)
{
    hipFunction_t f;
    std::string name;
    if (get_function(m, adjoint ? qgs::Kernel::AdjLds : qgs::Kernel::TglLds, 0, &f, &name)) return -1;
    const int MT = m->cg.lds_tgl_members, NC = 64 / MT;
    const int64_t bx = (n_traj + MT - 1) / MT, by = (n_tg + NC - 1) / NC;
    if (by > 65535) return fail("too many tangent columns for the LDS-resident tangent kernel");
    if (m->b_vwork.ensure(sizeof(double) * (size_t)m->ndim * 64 * (size_t)(bx * by))) return -1;
    double *vw = m->b_vwork.f64();
    long long nt = n_traj, l = ld, ntg = n_tg, sb = step_begin, se = step_end, ws = write_steps, nr = n_records;
    int bw = backward, wf = write_final, S = s;
    double inv = inverse;
    void *args[] = {(void *)&w_in, &w_out, &vw, &d_rec_fm, (void *)&stages, (void *)&d_time, (void *)&d_tab,
                    &nt, &l, &ntg, &sb, &se, &ws, &nr, &bw, &wf, &inv, &S};
    note_kernel(m, name, f);
    HIPCHK(hipModuleLaunchKernel(f, (unsigned)bx, (unsigned)by, 1, 64 * m->cg.lds_waves, 1, 1, 0, st, args, nullptr));
    return 0;
}

int check_common(const qgs_model *m, int64_t n_traj, int64_t ld)
{
    if (!m) return fail("null model");
    if (n_traj < 1) return fail("n_traj must be >= 1");
    if (ld < n_traj || (ld % 64) != 0) return fail("ld must be >= n_traj and a multiple of 64");
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// QGS_HIP_PREBUILD_SHARD="i/n": this process compiles only every n-th code object of the pre-build lists (the i-th ones), so
// that n processes walking the same lists share the work (__graft_entry__.build()).  The ordinal runs over all
// qgs_prebuild* calls of the process.
static bool prebuild_mine()
{
    static int ordinal = 0, shard = 0, shards = 1, parsed = 0;
    if (!parsed) {
        parsed = 1;
        if (const char *e = std::getenv("QGS_HIP_PREBUILD_SHARD")) {
            int a = 0, b = 1;
            if (std::sscanf(e, "%d/%d", &a, &b) == 2 && b >= 1 && a >= 0 && a < b) { shard = a; shards = b; }
        }
    }
    return (ordinal++ % shards) == shard;
}

extern "C" {

const char *qgs_last_error(void) { return g_err.c_str(); }

int qgs_backend_info(int *n_devices, char *arch_buf, int buflen)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1) {
        if (n_devices) *n_devices = 0;
        return fail(std::string("no HIP device visible (") + (e == hipSuccess ? "count 0" : hipGetErrorString(e)) +
                    "); libqgs_hip has no CPU path");
    }
    if (n_devices) *n_devices = n;
    if (arch_buf && buflen > 0) {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, 0));
        std::snprintf(arch_buf, (size_t)buflen, "%s", prop.gcnArchName);
    }
    return 0;
}

int64_t qgs_n_records(const double *time, int64_t n_time, int64_t write_steps)
{
    if (write_steps == 0 || n_time < 1) return 1;                 // integrate.py:190-191
    int64_t n = (n_time + write_steps - 1) / write_steps;         // len(time[::write_steps])
    if (time[(n - 1) * write_steps] != time[n_time - 1]) n += 1;  // :195-196
    return n;
}

// Reads the caller's COO arrays (rank 3 or 5) into the model: validated entries for the generic kernels and the
// reduced term lists (+ derived monomials) the code generator works on.
static int load_tensors(qgs_model *m, int rank, int64_t nnz, const int32_t *coo, const double *val, int64_t jnnz,
                        const int32_t *jcoo, const double *jval, std::vector<Entry> *Tr, std::vector<Entry> *Jr)
{
    const int ndim = m->ndim;
    auto read = [&](int64_t n, const int32_t *c, const double *v, std::vector<Entry> &out, bool jac) {
        for (int64_t e = 0; e < n; ++e) {
            const int32_t *q = c + (int64_t)rank * e;
            for (int r = 0; r < rank; ++r) if (q[r] < 0 || q[r] > ndim) return false;
            Entry t{q[0], q[1], q[2], rank == 5 ? q[3] : 0, rank == 5 ? q[4] : 0, v[e]};
            // generic-kernel tensors: row 0 is the constant slot (res[0] = 1), Df drops row / column 0
            if (t.i >= 1 && (!jac || t.j >= 1)) out.push_back(t);
        }
        return true;
    };
    std::vector<Entry> tr, jr;
    if (!read(nnz, coo, val, tr, false)) return fail("tensor coordinate out of range");
    if (!read(jnnz, jcoo, jval, jr, true)) return fail("jacobian coordinate out of range");
    qgs::reduce_polynomial(ndim, rank, nnz, coo, val, false, m->T, m->der.t);
    qgs::reduce_polynomial(ndim, rank, jnnz, jcoo, jval, true, m->J, m->der.j);
    m->magnitude_ulp = magnitude_ulp_from_env();
    qgs::canonicalize(m->T, m->canon_t, m->magnitude_ulp);
    qgs::canonicalize(m->J, m->canon_j, m->magnitude_ulp);
    // structure hash: everything of a canonical form the generated source can depend on
    auto hash_form = [&](const qgs::Canonical &c, const std::vector<std::pair<int, int>> &der, char tag) {
        Hasher h;
        const int64_t head[5] = {ndim, rank, (int64_t)c.terms.size(), (int64_t)der.size(), (int64_t)tag};
        h.add(head, sizeof head);
        for (const qgs::Term &t : c.terms) {
            const int32_t q[3] = {t.i, t.j, t.k};
            h.add(q, sizeof q);
            h.add(&t.v, sizeof t.v);
        }
        for (const auto &pr : der) { const int32_t q[2] = {pr.first, pr.second}; h.add(q, sizeof q); }
        return h.done();
    };
    m->hash_t = hash_form(m->canon_t, m->der.t, 'T');
    m->hash_j = hash_form(m->canon_j, m->der.j, 'J');
    m->rank = rank;
    m->nnz_in = nnz;
    m->jnnz_in = jnnz;
    if (Tr) Tr->swap(tr);
    if (Jr) Jr->swap(jr);
    return 0;
}

int qgs_model_create_rank(int device, int ndim, int rank, int64_t nnz, const int32_t *coo, const double *val, int64_t jnnz,
                          const int32_t *jcoo, const double *jval, qgs_model **out)
{
    if (!out) return fail("out is null");
    *out = nullptr;
    if (rank != 3 && rank != 5) return fail("tensor rank must be 3 or 5");
    if (ndim < 1 || ndim > 65534) return fail("ndim out of range");
    if (nnz < 0 || (nnz > 0 && (!coo || !val))) return fail("bad tensor arguments");
    if (jnnz < 0 || (jnnz > 0 && (!jcoo || !jval))) return fail("bad jacobian tensor arguments");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail("no HIP device visible; libqgs_hip has no CPU path");
    if (device < 0 || device >= n) return fail("device index out of range");
    HIPCHK(hipSetDevice(device));
    qgs_model *m = new qgs_model();
    m->device = device;
    m->ndim = ndim;
    m->arch = target_arch(device);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            m->n_simd = 4 * prop.multiProcessorCount;
    }
    std::vector<Entry> Tr, Jr;
    if (load_tensors(m, rank, nnz, coo, val, jnnz, jcoo, jval, &Tr, &Jr)) { delete m; return -1; }
    const bool r5 = rank == 5;
    auto pack = [](int a, int b) { return ((uint32_t)a << 16) | (uint32_t)b; };
    {
        std::vector<int> cnt(ndim + 2, 0);
        for (const auto &t : Tr) m->max_row_terms = std::max(m->max_row_terms, ++cnt[t.i]);
    }
    {
        std::vector<int> ci(ndim + 2, 0), cj(ndim + 2, 0);
        for (const auto &t : Jr) m->max_jrow_terms = std::max(m->max_jrow_terms, std::max(++ci[t.i], ++cj[t.j]));
    }
    HostCsr hT = build_csr(ndim, Tr, r5, [](const Entry &t) { return t.i; }, [&](const Entry &t) { return pack(t.j, t.k); });
    // Jacobian kernel wants (j,k) per row i; tangent model wants (w=j, x=k) per row i; adjoint (w=i, x=k) per row j
    HostCsr hJi = build_csr(ndim, Jr, r5, [](const Entry &t) { return t.i; }, [&](const Entry &t) { return pack(t.j, t.k); });
    HostCsr hJj = build_csr(ndim, Jr, r5, [](const Entry &t) { return t.j; }, [&](const Entry &t) { return pack(t.i, t.k); });
    if (upload_csr(hT, m->dT) || upload_csr(hJi, m->dJ_by_i) || upload_csr(hJj, m->dJ_by_j)) { qgs_model_destroy(m); return -1; }
    if (upload_pairs(m, Jr, r5)) { qgs_model_destroy(m); return -1; }
    if (!r5 && upload_tiled(m, Tr)) { qgs_model_destroy(m); return -1; }
    if (r5 && upload_reduced(m)) { qgs_model_destroy(m); return -1; }
    classify_model(m);
    apply_env_options(m->cg);
    if (!m->der.t.empty()) m->cg.lds_asm = false;       // the hand-scheduled LDS stepper takes rank-3 tensors only
    m->tune.read_env();
    if (r5) m->cg.row_split = 1;       // the row-split stepper would evaluate the derived monomials once per wavefront
    *out = m;
    return 0;
}

int qgs_model_create(int device, int ndim, int64_t nnz, const int32_t *coo, const double *val, int64_t jnnz,
                     const int32_t *jcoo, const double *jval, qgs_model **out)
{
    return qgs_model_create_rank(device, ndim, 3, nnz, coo, val, jnnz, jcoo, jval, out);
}

int qgs_model_destroy(qgs_model *m)
{
    if (!m) return 0;
    (void)hipSetDevice(m->device);
    for (auto &kv : m->modules) (void)hipModuleUnload(kv.second);
    free_csr(m->dT); free_csr(m->dJ_by_i); free_csr(m->dJ_by_j);
    free_csr(m->dT_red); free_csr(m->dJ_red_by_i); free_csr(m->dJ_red_by_j);
    if (m->d_chain_t) (void)hipFree(m->d_chain_t);
    if (m->d_chain_j) (void)hipFree(m->d_chain_j);
    for (void *q : {(void *)m->t_row_term, (void *)m->t_term_joff, (void *)m->t_term_koff, (void *)m->t_term_c, (void *)m->t_row_map})
        if (q) (void)hipFree(q);
    for (Buffer *b : {&m->d_time, &m->d_tab, &m->work, &m->stages, &m->b_in_rows, &m->b_in_modes, &m->b_rec_modes,
                      &m->b_rec_rows, &m->b_tg_rows, &m->b_tg_modes, &m->b_fm_modes, &m->b_fm_rows, &m->b_state2, &m->b_tg2, &m->b_ywork, &m->b_vwork, &m->b_mom_part, &m->b_mom_out, &m->b_unit,
                      &m->b_carry, &m->b_win[0], &m->b_win[1], &m->b_fwin[0], &m->b_fwin[1], &m->b_drain})
        b->release();
    for (int i = 0; i < 2; ++i) {
        if (m->ev_comp[i]) (void)hipEventDestroy(m->ev_comp[i]);
        if (m->ev_copy[i]) (void)hipEventDestroy(m->ev_copy[i]);
    }
    for (int64_t t : m->drain_tickets) (void)qgs::bridge_wait_done(t, nullptr);
    for (auto &kv : m->drain_slots) if (kv.second.ev) (void)hipEventDestroy(kv.second.ev);
    for (auto &b : m->drain_pool) b->release();
    m->uploads.release();
    if (m->st_comp) (void)hipStreamDestroy(m->st_comp);
    if (m->st_copy) (void)hipStreamDestroy(m->st_copy);
    if (m->h_pin) (void)hipHostFree(m->h_pin);
    for (void *q : {(void *)m->d_one_counter, (void *)m->p_lut, (void *)m->p_ptr, (void *)m->p_idx, (void *)m->p_idx2, (void *)m->p_val})
        if (q) (void)hipFree(q);
    delete m;
    return 0;
}

int64_t qgs_model_info(const qgs_model *m, int which)
{
    if (!m) return -1;
    switch (which) {
    case 0: return m->ndim;
    case 1: return m->nnz_in;
    case 2: return m->jnnz_in;
    case 3: return m->device;
    case 4: return (m->spec_possible || m->lds_spec_possible) ? 1 : 0;
    case 5: return m->rank;
    case 6: return (int64_t)m->der.t.size();
    case 7: return (int64_t)m->der.j.size();
    case 8: return m->last_windows;
    case 9: return m->last_groups;
    default: return -1;
    }
}

int qgs_model_set_kernel(qgs_model *m, int kind)
{
    if (!m) return fail("null model");
    if (kind < 0 || kind > 2) return fail("kind must be 0, 1 or 2");
    if (kind == 2 && !m->spec_possible && !m->lds_spec_possible) return fail("specialised kernels are not available for this ndim");
    m->kernel_kind = kind;
    m->tune = LaunchTuning();             // the selection knobs are read here and at model creation, never in a launch path
    m->tune.read_env();
    return 0;
}

int qgs_kernel_clock(qgs_model *m, double *shader_ghz, double *elapsed_ms)
{
    if (!m) return fail("null model");
    auto it = m->modules.find(m->last.name);
    if (it == m->modules.end()) return fail("the last kernel (" + m->last.name + ") is not a generated one: no clock probe");
    HIPCHK(hipSetDevice(m->device));
    hipDeviceptr_t dptr = nullptr;
    size_t bytes = 0;
    if (hipModuleGetGlobal(&dptr, &bytes, it->second, "qgs_clock_probe") != hipSuccess || bytes != 4 * sizeof(unsigned long long)) {
        (void)hipGetLastError();
        return fail("kernel " + m->last.name + " carries no clock probe");
    }
    HIPCHK(hipDeviceSynchronize());
    unsigned long long h[4];
    if (copy_d2h(h, (const void *)dptr, sizeof h)) return -1;
    if (h[3] <= h[1] || h[2] <= h[0]) return fail("kernel " + m->last.name + ": the clock probe has not been written by a completed launch");
    const double ns = (double)(h[3] - h[1]) * 10.0;                         // s_memrealtime: 100 MHz
    if (shader_ghz) *shader_ghz = (double)(h[2] - h[0]) / ns;
    if (elapsed_ms) *elapsed_ms = ns * 1e-6;
    return 0;
}

int qgs_last_kernel_info(const qgs_model *m, char *name_buf, int buflen, int *vgprs, int *sgprs, int *lds_bytes,
                         int *scratch_bytes)
{
    if (!m) return fail("null model");
    if (name_buf && buflen > 0) std::snprintf(name_buf, (size_t)buflen, "%s", m->last.name.c_str());
    if (vgprs) *vgprs = m->last.vgprs;
    if (sgprs) *sgprs = m->last.sgprs;
    if (lds_bytes) *lds_bytes = m->last.lds;
    if (scratch_bytes) *scratch_bytes = m->last.scratch;
    return 0;
}

int64_t qgs_model_kernel_source(const qgs_model *m, char *buf, int64_t buflen)
{
    if (!m) return -1;
    // generated again on request (cache hits never run the generator): the value-free source of every specialised kernel loaded so far
    std::string src;
    try {
        for (const auto &ks : m->loaded_kernels)
            src += qgs::generate_kernel(m->ndim, m->canon_t.terms, m->canon_j.terms, ks.first, ks.second, m->cg, m->der).source;
        if (src.empty() && m->spec_possible)
            src = qgs::generate_kernel(m->ndim, m->canon_t.terms, m->canon_j.terms, qgs::Kernel::Tend, 0, m->cg, m->der).source;
    } catch (const std::exception &) {
        return -1;
    }
    if (buf && buflen > 0) {
        const size_t n = std::min<size_t>(src.size(), (size_t)buflen - 1);
        std::memcpy(buf, src.data(), n);
        buf[n] = 0;
    }
    return (int64_t)src.size();
}

// ---- device-layout entry points ---------------------------------------------------------------

int qgs_pack_states(qgs_model *m, int64_t n_traj, int64_t ld, const double *d_rows, double *d_modes, void *stream)
{
    if (check_common(m, n_traj, ld)) return -1;
    HIPCHK(hipSetDevice(m->device));
    qgs::launch_pack_states(m->ndim, n_traj, ld, d_rows, d_modes, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

int qgs_unpack_states(qgs_model *m, int64_t n_traj, int64_t ld, const double *d_modes, double *d_rows, void *stream)
{
    if (check_common(m, n_traj, ld)) return -1;
    HIPCHK(hipSetDevice(m->device));
    qgs::launch_unpack_states(m->ndim, n_traj, ld, d_modes, d_rows, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

int qgs_pack_tangent(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_tg, const double *d_rows, double *d_modes, void *stream)
{
    if (check_common(m, n_traj, ld)) return -1;
    if (n_tg < 1 || (int64_t)m->ndim * n_tg > (int64_t)65535 * 64 || !d_rows || !d_modes) return fail("bad n_tg / null pointer");
    HIPCHK(hipSetDevice(m->device));
    qgs::launch_pack_tangent(m->ndim, n_tg, n_traj, ld, d_rows, d_modes, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

int qgs_local_exponents_device(qgs_model *m, int64_t n, const double *d_rdiag, double dt, double *d_out, void *stream)
{
    if (!m) return fail("null model");
    if (n < 1 || !d_rdiag || !d_out) return fail("bad n / null pointer");
    if (!(dt != 0.0)) return fail("dt must not be zero");
    HIPCHK(hipSetDevice(m->device));
    qgs::launch_local_exponents(n, d_rdiag, dt, d_out, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

int qgs_unpack_records(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_inner, int64_t n_records, const double *d_in,
                       double *d_out, void *stream)
{
    if (check_common(m, n_traj, ld)) return -1;
    if (n_inner < 1 || n_inner > (int64_t)65535 * 64 || n_records < 1) return fail("bad n_inner / n_records");
    HIPCHK(hipSetDevice(m->device));
    qgs::launch_unpack_records(n_inner, n_traj, ld, n_records, d_in, d_out, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

int qgs_tendencies_device(qgs_model *m, int64_t n_traj, int64_t ld, const double *d_x, double *d_dx, void *stream)
{
    if (check_common(m, n_traj, ld)) return -1;
    HIPCHK(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    if (use_spec(m, 1, nullptr) && !m->prefer_lds) {
        hipFunction_t f;
        if (get_function(m, qgs::Kernel::Tend, 0, &f)) return -1;
        long long nt = n_traj, l = ld;
        void *args[] = {(void *)&d_x, (void *)&d_dx, &nt, &l};
        note_kernel(m, "qgs_spec_tend", f);
        return launch(f, n_traj, st, args);
    }
    if (m->kernel_kind != 1 && m->lds_spec_possible &&
        lds_kernel_wanted(m, qgs::Kernel::TendLds, (double)n_traj * (double)m->T.size())) {
        // the tendencies-only flavour of the LDS-resident stepper: one step, one stage, leaves after the first evaluation
        if (!m->b_unit.p) {
            const double unit[4] = {0.0, 1.0, 1.0, 0.0};                      // time grid {0, 1}; tableau b = {1}
            if (m->b_unit.ensure(sizeof unit)) return -1;
            if (copy_h2d(m->b_unit.p, unit, sizeof unit)) return -1;
        }
        return launch_rk_lds(m, n_traj, ld, d_x, d_dx, nullptr, nullptr, m->b_unit.f64(), m->b_unit.f64() + 2, 0, 1, 0, 1, 0, 0, 1,
                             st, qgs::Kernel::TendLds);
    }
    if (use_spec(m, 1, nullptr)) {                                    // prefer_lds, but the LDS kernel is not wanted / built
        hipFunction_t f;
        if (get_function(m, qgs::Kernel::Tend, 0, &f)) return -1;
        long long nt = n_traj, l = ld;
        void *args[] = {(void *)&d_x, (void *)&d_dx, &nt, &l};
        note_kernel(m, "qgs_spec_tend", f);
        return launch(f, n_traj, st, args);
    }
    qgs::launch_gen_tend(m->dT.view(), m->ndim, n_traj, ld, d_x, d_dx, st);
    note_kernel(m, "gen_tend_kernel", nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

static int jacobian_device(qgs_model *m, int64_t n_traj, int64_t ld, const double *d_x, double *d_jm, hipStream_t st)
{
    if (m->J.empty()) return fail("model was created without a Jacobian tensor");
    HIPCHK(hipMemsetAsync(d_jm, 0, sizeof(double) * (size_t)m->ndim * m->ndim * ld, st));
    if (use_spec(m, 1, nullptr) && m->spec_jac_possible) {
        hipFunction_t f;
        if (get_function(m, qgs::Kernel::Jac, 0, &f)) return -1;
        long long nt = n_traj, l = ld;
        void *args[] = {(void *)&d_x, (void *)&d_jm, &nt, &l};
        note_kernel(m, "qgs_spec_jac", f);
        return launch(f, n_traj, st, args);
    }
    qgs::launch_gen_jac(m->dJ_by_i.view(), m->ndim, n_traj, ld, d_x, d_jm, st);
    note_kernel(m, "gen_jac_kernel", nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

// One launch of the trajectory stepper over the steps [step_begin, step_end) of a run of n_steps steps: state in from y_in,
// state out to y_out (may be null), records of the steps in the range to d_rec (indexed by the record number of the WHOLE run:
// a caller that keeps only a window of records passes the window's base minus the offset of its first record), the final
// record when write_final.  Which kernel runs depends on the run (ensemble size, tableau, write_steps), never on the range, so
// a run cut into ranges is bitwise the run in one piece.
static int rk_launch(qgs_model *m, int64_t n_traj, int64_t ld, const double *d_ic, double *y_out, double *d_rec,
                     const double *d_time, const double *d_tab_spec, const double *d_tab_full, int64_t step_begin,
                     int64_t step_end, int64_t n_steps, int64_t write_steps, int64_t n_records, int backward, int write_final,
                     int s, const double *a, hipStream_t st)
{
    if (use_wave(m, n_traj, s, a)) {
        // small ensemble: one workgroup per trajectory, lane = tensor row (latency-optimised)
        qgs::RkArgs pw{m->ndim, s, n_traj, ld, step_begin, step_end, write_steps, n_records, backward, write_final};
        HIPCHK(qgs::launch_gen_rk_wave(m->wave_T(), m->max_row_terms, pw, d_ic, y_out, d_rec, nullptr, d_time, d_tab_spec, st, m->chains(false)));
        note_kernel(m, "gen_rk_wave_kernel", nullptr);
        return 0;
    }
    if (m->prefer_lds && use_lds_spec(m, n_traj, n_steps, s, a))
        return launch_rk_lds(m, n_traj, ld, d_ic, y_out, d_rec, nullptr, d_time, d_tab_spec, step_begin, step_end, write_steps,
                             n_records, backward, write_final, s, st);
    if (use_spec(m, s, a)) {
        // Kernel choice by ensemble size (measured, tools/kbench.py / tools/latency_bench.py, MAOOAM-36, ms per 1000 steps):
        //   n <= 2048      wave-per-trajectory kernel (handled above)        0.8-1.5
        //   n <= 40960     row-split stepper, R = 4 wavefronts per 64 members: 2.4 (n <= 16384), 3.7 (n = 32768) -- the
        //                  chip is not full, so splitting the rows over more wavefronts shortens every trajectory
        //   above          plain one-wave-per-64-members stepper: 4.6 at 65 536 members (fp64 VALU ~89 % busy);
        //                  the split needs LDS + a barrier per stage and loses there (5.9)
        const int R = m->cg.row_split;
        const int64_t waves = (n_traj + 63) / 64;
        bool split = R > 1 && m->ndim >= 2 * R && waves * R <= (int64_t)m->n_simd * 5 / 2;
        if (m->tune.rk_variant == 1) split = false;
        if (m->tune.rk_variant == 2 && R > 1 && m->ndim >= 2 * R) split = true;
        // every step is a record (write_steps == 1, the reference's default): the variant with the record stores spread over the step
        const bool spread = !split && m->cg.rk_spread_rec && m->tune.rk_spread_rec && write_steps == 1 && ld >= 64 * waves;
        hipFunction_t f;
        std::string name;
        if (get_function(m, split ? qgs::Kernel::RkSplit : (spread ? qgs::Kernel::RkRec : qgs::Kernel::Rk), s, &f, &name)) return -1;
        double *stg = nullptr;
        long long nt = n_traj, l = ld, sb = step_begin, se = step_end, ws = write_steps, nr = n_records;
        int bw = backward, wf = write_final;
        void *args[] = {(void *)&d_ic, &y_out, &d_rec, &stg, (void *)&d_time, (void *)&d_tab_spec,
                        &nt, &l, &sb, &se, &ws, &nr, &bw, &wf};
        note_kernel(m, name, f);
        const unsigned blocks = (unsigned)waves;
        HIPCHK(hipModuleLaunchKernel(f, blocks, 1, 1, split ? 64 * R : 64, 1, 1, 0, st, args, nullptr));
        return 0;
    }
    if (use_lds_spec(m, n_traj, n_steps, s, a))
        return launch_rk_lds(m, n_traj, ld, d_ic, y_out, d_rec, nullptr, d_time, d_tab_spec, step_begin, step_end, write_steps,
                             n_records, backward, write_final, s, st);
    if (m->kernel_kind != 1 && m->lds_spec_possible && s >= 2 && s <= 64 && !qgs::tableau_is_subdiagonal(s, a) &&
        tableau_is_lower_triangular(s, a) &&
        lds_kernel_wanted(m, qgs::Kernel::RkLdsDense, (double)n_traj * (double)n_steps * (double)s * (double)m->T.size())) {
        // general tableau at LDS-resident sizes: same kernel text, partial stage sums in a private global buffer
        hipFunction_t f;
        std::string name;
        if (get_function(m, qgs::Kernel::RkLdsDense, 0, &f, &name)) return -1;
        const int64_t blocks = (n_traj + 63) / 64;
        if (m->b_ywork.ensure(sizeof(double) * (size_t)m->ndim * 64 * (size_t)blocks)) return -1;
        if (m->b_vwork.ensure(sizeof(double) * (size_t)s * m->ndim * 64 * (size_t)blocks)) return -1;
        double *yw = m->b_ywork.f64(), *pw = m->b_vwork.f64(), *stg = nullptr;
        long long nt = n_traj, l = ld, sb = step_begin, se = step_end, ws = write_steps, nr = n_records;
        int bw = backward, wf = write_final, S = s;
        void *args[] = {(void *)&d_ic, &y_out, &yw, &pw, &d_rec, &stg, (void *)&d_time, (void *)&d_tab_full,
                        &nt, &l, &sb, &se, &ws, &nr, &bw, &wf, &S};
        note_kernel(m, name, f);
        HIPCHK(hipModuleLaunchKernel(f, (unsigned)blocks, 1, 1, 64 * m->cg.lds_waves, 1, 1, 0, st, args, nullptr));
        return 0;
    }
    if (m->kernel_kind != 1 && m->spec_possible && !m->prefer_lds && s >= 2 && s <= 8 && tableau_is_lower_triangular(s, a) &&
        (size_t)(s - 2) * m->ndim * 512 <= (size_t)QGS_LDS_STATE_BYTES) {
        // general explicit tableau: register-resident tendencies, partial stage sums in LDS (codegen emit_rk_dense_kernel)
        hipFunction_t f;
        std::string name;
        if (get_function(m, qgs::Kernel::RkDense, s, &f, &name)) return -1;
        double *stg = nullptr;
        long long nt = n_traj, l = ld, sb = step_begin, se = step_end, ws = write_steps, nr = n_records;
        int bw = backward, wf = write_final;
        void *args[] = {(void *)&d_ic, &y_out, &d_rec, &stg, (void *)&d_time, (void *)&d_tab_full, &nt, &l, &sb, &se, &ws, &nr, &bw, &wf};
        note_kernel(m, name, f);
        return launch(f, n_traj, st, args);
    }
    qgs::RkArgs p{m->ndim, s, n_traj, ld, step_begin, step_end, write_steps, n_records, backward, write_final};
    if (use_tiled(m, s, a)) {
        HIPCHK(qgs::launch_gen_rk_tiled(m->tiled(), p, d_ic, y_out, d_rec, nullptr, d_time, d_tab_spec, st));
        note_kernel(m, "gen_rk_tiled_kernel", nullptr);
        return 0;
    }
    if (m->work.ensure(sizeof(double) * (size_t)(s + 2) * m->ndim * ld)) return -1;
    qgs::launch_gen_rk(m->dT.view(), p, d_ic, y_out, d_rec, nullptr, m->work.f64(), d_time, d_tab_full, st);
    note_kernel(m, "gen_rk_kernel", nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

int qgs_rk_integrate_device(qgs_model *m, int64_t n_traj, int64_t ld, const double *d_ic, const double *time,
                            int64_t n_time, int time_direction, int64_t write_steps, int s, const double *b,
                            const double *c, const double *a, double *d_rec, void *stream)
{
    (void)c;   // autonomous system: f ignores t (tendencies.py:112)
    if (check_common(m, n_traj, ld)) return -1;
    if (!time || n_time < 1 || !b || !a || s < 1) return fail("bad time grid / tableau");
    if (time_direction != 1 && time_direction != -1) return fail("time_direction must be +1 or -1");
    if (write_steps < 0) return fail("write_steps must be >= 0");
    if (!d_ic || !d_rec) return fail("null device pointer");
    HIPCHK(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    const double *d_time, *d_tab_spec, *d_tab_full;
    if (stage_time_tab(m, time, n_time, time_direction, s, b, a, st, &d_time, &d_tab_spec, &d_tab_full)) return -1;
    return rk_launch(m, n_traj, ld, d_ic, nullptr, d_rec, d_time, d_tab_spec, d_tab_full, 0, n_time - 1, n_time - 1, write_steps,
                     qgs_n_records(time, n_time, write_steps), time_direction == -1, 1, s, a, st);
}

// Trajectory + tangent / adjoint passes over the steps [step_lo, step_hi) of a run of n_steps steps.  `first`: the states come
// from d_ic / d_tg_ic, otherwise from the model's carry buffers (b_state2 / b_tg2), where every call leaves the states it
// ended with.  Records are indexed by the record number of the whole run (see rk_launch); the final record is written when
// write_final.  Kernel choices depend on the run, not on the range.
static int tgls_launch(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_tg, const double *d_ic, const double *d_tg_ic,
                       const double *d_time, const double *d_tab_spec, const double *d_tab_full, int64_t step_lo, int64_t step_hi,
                       int64_t n_steps, bool first_range, int write_final, int64_t write_steps, int64_t n_records, int backward,
                       int s, const double *a, int adjoint, double inverse, double *d_rec, double *d_rec_fm, hipStream_t st)
{
    const int64_t A = (int64_t)m->ndim * ld, L = n_tg * ld;
    const bool spec = use_spec(m, s, a);
    // general lower-triangular tableau on the register-resident kernels (partial stage sums in LDS)
    const bool dense = !spec && m->kernel_kind != 1 && m->spec_possible && !m->prefer_lds && s >= 2 && s <= 8 &&
                       !qgs::tableau_is_subdiagonal(s, a) && tableau_is_lower_triangular(s, a) &&
                       (size_t)(s - 2) * m->ndim * 512 <= (size_t)64 * 1024;

    // The reference pre-writes record 0 with the initial conditions (integrate.py:581-582); with at least
    // one step that record is rewritten by the loop, with zero steps the final record covers it.
    // Steps are processed in chunks: trajectory kernel (stores every stage state) -> tangent kernel.
    const size_t stage_bytes_per_step = sizeof(double) * (size_t)s * A;
    int64_t chunk = std::max<int64_t>(1, (int64_t)((size_t)768 << 20) / (int64_t)stage_bytes_per_step);
    if (m->tune.tgls_chunk > 0) chunk = m->tune.tgls_chunk;
    chunk = std::min<int64_t>(chunk, std::max<int64_t>(1, n_steps));
    if (m->stages.ensure(stage_bytes_per_step * (size_t)chunk)) return -1;
    if (m->b_state2.ensure(sizeof(double) * (size_t)A)) return -1;
    if (m->b_tg2.ensure(sizeof(double) * (size_t)m->ndim * L)) return -1;
    if (!spec || !m->spec_jac_possible) {
        if (m->work.ensure(sizeof(double) * (size_t)(s + 2) * m->ndim * std::max<int64_t>(ld, L))) return -1;
    }
    double *y_state = m->b_state2.f64();
    double *w_state = m->b_tg2.f64();
    double *stages = m->stages.f64();
    const qgs::DevTensor Jrow = adjoint ? m->dJ_by_j.view() : m->dJ_by_i.view();

    int64_t begin = step_lo;
    bool first = first_range;
    do {
        const int64_t end = std::min(step_hi, begin + chunk);
        const int final_chunk = (end == step_hi) && write_final;
        const double *y_src = first ? d_ic : y_state;
        const double *w_src = first ? d_tg_ic : w_state;
        // which tangent kernel this chunk takes (decided first: the stepper has to know the layout of the stage record it feeds)
        const bool lds_tgl = (!spec || !m->spec_jac_possible) && use_lds_tgl(m, n_traj * n_tg, n_steps, s, a, adjoint);
        const bool tg_lds = lds_tgl && (m->kernel_kind == 2 || n_traj * n_tg > lds_tgl_min_pairs(m));
        const bool tg_dense = !tg_lds && dense && m->spec_jac_possible;
        const bool tg_wave = !tg_lds && !tg_dense && use_tgl_wave(m, n_traj * n_tg, s, a);
        const bool tg_spec = !tg_lds && !tg_dense && !tg_wave && spec && m->spec_jac_possible;
        // shared-stage-state kernel: C columns of the same 64 members per workgroup, stage states prefetched through LDS
        // Measured (tools/tgls_scale.py, MAOOAM-36, 36 columns, 10 steps): while the stage record of a chunk stays in the
        // 256 MB Infinity Cache every column can afford to read it (one-wavefront kernel 3-8 % ahead: 1.11 vs 1.21 ms at
        // 16 384 members, 189 MB); beyond that the re-reads go to HBM and sharing wins 1.5x (65 536 members, 755 MB:
        // 4.4 vs 6.4 ms).  Rank-5 models keep the plain kernel (their derived monomials already fill the register file).
        // The size that decides is the stage record of a FULL chunk of the run, not of this range: a run cut into record windows
        // (whose last chunks are shorter) takes the same kernels in every range and stays bitwise the run in one piece.
        const int C = m->cg.tgl_share_x;
        const bool share_x = tg_spec && C > 1 && n_tg >= 2 && (size_t)m->ndim * 1024 <= (size_t)64 * 1024 &&
                             (n_tg + C - 1) / C <= 65535 && m->der.j.empty() &&
                             stage_bytes_per_step * (size_t)chunk >= m->tune.tgl_share_min_bytes && !m->tune.tgl_plain;
        const bool st_wave = use_wave(m, n_traj, s, a);
        const bool st_lds_first = !st_wave && m->prefer_lds && use_lds_spec(m, n_traj, n_steps, s, a);
        const bool st_spec = !st_wave && !st_lds_first && !dense && spec;
        // the fused stepper and the one-wavefront-per-column tangent kernel exchange the stage record in mode pairs
        // (128-bit accesses: codegen emit_rk_kernel pair_stages); every other combination uses S[..][mode][member]
        const bool pair = st_spec && tg_spec && !share_x && m->cg.tgl_pair && (ld & 1) == 0;
        // --- trajectory pass (stores every stage input state) ---
        qgs::RkArgs pa{m->ndim, s, n_traj, ld, begin, end, write_steps, n_records, backward, final_chunk};
        long long nt = n_traj, l = ld, sb = begin, se = end, ws = write_steps, nr = n_records, ntg = n_tg;
        int bw = backward, wf = final_chunk, adj = adjoint ? 1 : 0;
        double inv = inverse;
        if (st_wave) {                                    // few members: latency-optimised, lane = tensor row
            HIPCHK(qgs::launch_gen_rk_wave(m->wave_T(), m->max_row_terms, pa, y_src, y_state, d_rec, stages, d_time, d_tab_spec, st, m->chains(false)));
        } else if (st_lds_first) {
            if (launch_rk_lds(m, n_traj, ld, y_src, y_state, d_rec, stages, d_time, d_tab_spec, begin, end, write_steps, n_records,
                              backward, final_chunk, s, st)) return -1;
        } else if (dense) {
            hipFunction_t f1;
            if (get_function(m, qgs::Kernel::RkDense, s, &f1)) return -1;
            void *a1[] = {(void *)&y_src, &y_state, &d_rec, &stages, (void *)&d_time, (void *)&d_tab_full,
                          &nt, &l, &sb, &se, &ws, &nr, &bw, &wf};
            if (launch(f1, n_traj, st, a1)) return -1;
        } else if (spec) {
            hipFunction_t f1;
            if (get_function(m, pair ? qgs::Kernel::RkStagesPair : qgs::Kernel::RkStages, s, &f1)) return -1;
            void *a1[] = {(void *)&y_src, &y_state, &d_rec, &stages, (void *)&d_time, (void *)&d_tab_spec,
                          &nt, &l, &sb, &se, &ws, &nr, &bw, &wf};
            if (launch(f1, n_traj, st, a1)) return -1;
        } else if (use_lds_spec(m, n_traj, n_steps, s, a)) {
            if (launch_rk_lds(m, n_traj, ld, y_src, y_state, d_rec, stages, d_time, d_tab_spec, begin, end, write_steps, n_records,
                              backward, final_chunk, s, st)) return -1;
        } else if (use_tiled(m, s, a)) {
            HIPCHK(qgs::launch_gen_rk_tiled(m->tiled(), pa, y_src, y_state, d_rec, stages, d_time, d_tab_spec, st));
        } else {
            qgs::launch_gen_rk(m->dT.view(), pa, y_src, y_state, d_rec, stages, m->work.f64(), d_time, d_tab_full, st);
        }
        // --- tangent / adjoint pass ---
        if (tg_lds) {
            if (launch_tgl_lds(m, n_traj, ld, n_tg, w_src, w_state, d_rec_fm, stages, d_time, d_tab_spec, begin, end, write_steps,
                               n_records, backward, final_chunk, adjoint ? 1 : 0, inverse, s, st)) return -1;
        } else if (tg_dense) {
            hipFunction_t f2;
            std::string n2;
            if (get_function(m, qgs::Kernel::TglDense, s, &f2, &n2)) return -1;
            void *a2[] = {(void *)&w_src, &w_state, &d_rec_fm, &stages, (void *)&d_time, (void *)&d_tab_full,
                          &nt, &l, &ntg, &sb, &se, &ws, &nr, &bw, &wf, &adj, &inv};
            note_kernel(m, n2, f2);
            if (launch_tgl(f2, ld, n_tg, st, a2)) return -1;
        } else if (tg_wave) {                                    // few (member, column) pairs: lane = row of J / J^T
            HIPCHK(qgs::launch_gen_tgl_wave(m->wave_J(adjoint != 0), m->max_jrow_terms, pa, n_tg, inverse, w_src, w_state, d_rec_fm,
                                            stages, d_time, d_tab_spec, st, m->chains(true)));
            note_kernel(m, "gen_tgl_wave_kernel", nullptr);
        } else if (tg_spec) {
            hipFunction_t f2;
            std::string n2;
            if (get_function(m, share_x ? qgs::Kernel::TglX : (pair ? qgs::Kernel::TglPair : qgs::Kernel::Tgl), s, &f2, &n2)) return -1;
            void *a2[] = {(void *)&w_src, &w_state, &d_rec_fm, &stages, (void *)&d_time, (void *)&d_tab_spec,
                          &nt, &l, &ntg, &sb, &se, &ws, &nr, &bw, &wf, &adj, &inv};
            note_kernel(m, n2, f2);
            if (share_x) {
                HIPCHK(hipModuleLaunchKernel(f2, (unsigned)(ld / 64), (unsigned)((n_tg + C - 1) / C), 1, 64 * C, 1, 1, 0, st, a2, nullptr));
            } else if (launch_tgl(f2, ld, n_tg, st, a2)) return -1;
        } else {
            qgs::launch_gen_tgl(Jrow, pa, n_tg, inverse, w_src, w_state, d_rec_fm, stages, m->work.f64(), d_time, d_tab_full, st);
            note_kernel(m, "gen_tgl_kernel", nullptr);
            HIPCHK(hipGetLastError());
        }
        begin = end;
        first = false;
    } while (begin < step_hi);
    return 0;
}

int qgs_rk_tgls_integrate_device(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_tg, const double *d_ic,
                                 const double *d_tg_ic, const double *time, int64_t n_time, int time_direction,
                                 int64_t write_steps, int s, const double *b, const double *c, const double *a,
                                 int adjoint, double inverse, double *d_rec, double *d_rec_fm, void *stream)
{
    (void)c;
    if (check_common(m, n_traj, ld)) return -1;
    if (m->J.empty()) return fail("model was created without a Jacobian tensor");
    if (n_tg < 1) return fail("n_tg must be >= 1");
    if (!time || n_time < 1 || !b || !a || s < 1) return fail("bad time grid / tableau");
    if (time_direction != 1 && time_direction != -1) return fail("time_direction must be +1 or -1");
    if (write_steps < 0) return fail("write_steps must be >= 0");
    if (!d_ic || !d_tg_ic || !d_rec || !d_rec_fm) return fail("null device pointer");
    HIPCHK(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    const double *d_time, *d_tab_spec, *d_tab_full;
    if (stage_time_tab(m, time, n_time, time_direction, s, b, a, st, &d_time, &d_tab_spec, &d_tab_full)) return -1;
    return tgls_launch(m, n_traj, ld, n_tg, d_ic, d_tg_ic, d_time, d_tab_spec, d_tab_full, 0, n_time - 1, n_time - 1, true, 1,
                       write_steps, qgs_n_records(time, n_time, write_steps), time_direction == -1, s, a, adjoint, inverse,
                       d_rec, d_rec_fm, st);
}

int qgs_batched_qr_device(qgs_model *m, int64_t n_traj, int64_t ld, int n_rows, int n_cols, double *d_a, double *d_rdiag,
                          void *stream)
{
    if (check_common(m, n_traj, ld)) return -1;
    if (n_rows < 1 || n_cols < 1 || n_cols > n_rows) return fail("batched QR needs 1 <= n_cols <= n_rows");
    if (n_rows > 16384) return fail("batched QR: n_rows too large");
    HIPCHK(hipSetDevice(m->device));
    if (n_cols > 64 || n_rows > 300) {
        // beyond one column per lane / the LDS (e.g. the full 228-vector Lyapunov basis of MAOOAM 6x6): global-memory kernel
        if (m->work.ensure(sizeof(double) * (size_t)n_traj * (((size_t)n_rows + 17) * (size_t)n_cols + 256))) return -1;
        note_kernel(m, qgs::launch_batched_qr_global(n_rows, n_cols, n_traj, ld, d_a, d_rdiag, m->work.f64(), (hipStream_t)stream), nullptr);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (m->kernel_kind != 1) {          // (rows <= 300, cols <= 64 here)
        // shape-specialised kernel, 16 members per workgroup, columns in registers (codegen generate_qr_kernel), compiled once per shape
        const qgs::QrPlan plan = qr_plan_for(n_rows, n_cols);
        const std::string fname = "qgs_spec_qr_" + std::to_string(n_rows) + "x" + std::to_string(n_cols);
        const std::string key = fname + "|" + qgs::qr_plan_signature(plan);
        hipFunction_t f = nullptr;
        auto it = m->functions.find(key);
        if (it != m->functions.end()) f = it->second;
        else {
            std::shared_ptr<const KernelBlob> blob;
            bool cached = false;
            if (obtain_blob(key, m->arch, {}, [&] { return qgs::generate_qr_kernel(n_rows, n_cols, plan); }, &blob, &cached)) return -1;
            if (load_blob(m, fname, *blob, m->canon_t, &f)) return -1;          // (no tables)
            m->functions[key] = f;
        }
        long long nt = n_traj, l = ld;
        void *args[] = {(void *)&d_a, (void *)&d_rdiag, &nt, &l};
        note_kernel(m, fname, f);
        // workgroups: 16 members each (row design: 4 wavefronts x 4 members; tile design with 16-member tiles), or 8-member tiles in pairs
        unsigned grid, block;
        if (plan.row_groups > 0) {          // grid design (rows > 64): plan.members members per workgroup, plan.waves wavefronts each
            grid = (unsigned)((n_traj + plan.members - 1) / plan.members);
            block = 64u * (unsigned)(plan.waves * plan.members);
        } else {
            const int64_t tiles = (n_traj + (plan.members == 8 ? 8 : 16) - 1) / (plan.members == 8 ? 8 : 16);
            grid = (unsigned)(plan.members == 8 ? (tiles + 15) / 16 * 16 : tiles);
            block = 64u * (unsigned)plan.waves;
        }
        HIPCHK(hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, 0, (hipStream_t)stream, args, nullptr));
        return 0;
    }
    qgs::launch_batched_qr(n_rows, n_cols, n_traj, ld, d_a, d_rdiag, (hipStream_t)stream);
    note_kernel(m, "batched_qr_kernel", nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

int qgs_batched_matmul_device(qgs_model *m, int64_t n_traj, int64_t ld, int n_rows, int n_inner, int n_cols, int trans_a, int triangular,
                              const double *d_a, const double *d_b, double *d_c, void *stream)
{
    if (check_common(m, n_traj, ld)) return -1;
    if (n_rows < 1 || n_inner < 1 || n_cols < 1 || n_rows > 65535 || n_inner > 65535 || n_cols > 65535)
        return fail("batched matmul: dimensions must lie in 1 ... 65535");
    if (triangular < 0 || triangular > 2) return fail("batched matmul: triangular must be 0, 1 or 2");
    if (triangular == 1 && n_rows != n_cols) return fail("batched matmul: the upper triangle of the product is asked of a square result");
    if (triangular == 2 && n_inner != n_cols) return fail("batched matmul: an upper-triangular B must be square");
    if (!d_a || !d_b || !d_c) return fail("null device pointer");
    if (d_c == d_a || d_c == d_b) return fail("batched matmul: the result must not alias an operand");
    HIPCHK(hipSetDevice(m->device));
    qgs::launch_batched_matmul(n_rows, n_inner, n_cols, trans_a ? 1 : 0, triangular, n_traj, ld, d_a, d_b, d_c, (hipStream_t)stream);
    note_kernel(m, "batched_matmul_kernel", nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

int qgs_clv_backstep_device(qgs_model *m, int64_t n_traj, int64_t ld, int n_vec, const double *d_r, const double *d_a_in, double *d_a_out,
                            double *d_norm, const double *d_noise, double noise_pert, void *stream)
{
    if (check_common(m, n_traj, ld)) return -1;
    if (n_vec < 1 || n_vec > 65535) return fail("backward CLV step: n_vec must lie in 1 ... 65535");
    if (!d_r || !d_a_in || !d_a_out || !d_norm) return fail("null device pointer");
    if (d_a_out == d_a_in || d_a_out == d_r) return fail("backward CLV step: the result must not alias an operand");
    HIPCHK(hipSetDevice(m->device));
    qgs::launch_clv_backstep(n_vec, n_traj, ld, d_r, d_a_in, d_a_out, d_norm, d_noise, noise_pert, (hipStream_t)stream);
    note_kernel(m, "clv_backstep_kernel", nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

int qgs_ensemble_moments_device(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_rows, const double *d_x, double *d_mean,
                                double *d_var, void *stream)
{
    if (check_common(m, n_traj, ld)) return -1;
    if (n_rows < 1 || !d_x || !d_mean) return fail("bad arguments");
    if (n_rows > 0x7fffffff) return fail("too many rows");
    HIPCHK(hipSetDevice(m->device));
    if (m->b_mom_part.ensure(sizeof(double) * 2 * (size_t)n_rows * (size_t)qgs::moments_splits(n_rows, n_traj))) return -1;
    qgs::launch_moments(n_rows, n_traj, ld, d_x, m->b_mom_part.f64(), d_mean, d_var, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- host-layout entry points -------------------------------------------------------------------

static int64_t round_ld(int64_t n) { return (n + 63) / 64 * 64; }

// ---- single state: f(x), Df(x) for one (ndim,) vector ----------------------------------------------------------------------
// The callables the reference hands to SciPy / DiffEq solvers (documentation user_guide.rst:502-517) evaluate ONE state per
// call.  For one member the reference's (ndim,) layout and the mode-major layout coincide (ld = 1), so nothing is packed or
// unpacked: the state is copied (CPU memcpy, 288 bytes at ndim 36) into a page-locked block that the kernel reads over PCIe, the
// kernel writes its result into the same block, one launch, one stream synchronisation.
static int pin_ensure(qgs_model *m, size_t doubles)
{
    if (doubles <= m->pin_cap) return 0;
    if (m->h_pin) (void)hipHostFree(m->h_pin);
    m->h_pin = m->d_pin = nullptr;
    m->pin_cap = 0;
    HIPCHK(hipHostMalloc((void **)&m->h_pin, sizeof(double) * doubles, hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void **)&m->d_pin, m->h_pin, 0));
    std::memset(m->h_pin, 0, sizeof(double) * doubles);       // word 0 is the completion flag: sequence numbers start at 1
    m->pin_cap = doubles;
    return 0;
}

static int streams_ready(qgs_model *m)
{
    if (m->st_comp) return 0;
    HIPCHK(hipStreamCreateWithFlags(&m->st_comp, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&m->st_copy, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipEventCreateWithFlags(&m->ev_comp[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&m->ev_copy[i], hipEventDisableTiming));
    }
    return 0;
}

// wait until the single-state kernel has echoed `seq` into the page-locked block; the stream is only consulted now and then
// (a kernel that died would otherwise never be noticed)
static int one_state_wait(qgs_model *m, hipStream_t st, unsigned long long seq)
{
    volatile unsigned long long *flag = (volatile unsigned long long *)m->h_pin;
    for (unsigned long long spins = 1;; ++spins) {
        if (*flag == seq) return 0;
        if ((spins & 0xfffffull) == 0) {
            const hipError_t e = hipStreamQuery(st);
            if (e == hipSuccess) { if (*flag == seq) return 0; return fail("single-state kernel finished without reporting completion"); }
            if (e != hipErrorNotReady) return fail(std::string("single-state kernel failed: ") + hipGetErrorString(e));
        }
    }
}

static int one_state_ready(qgs_model *m, size_t doubles)
{
    if (streams_ready(m) || pin_ensure(m, doubles + 1)) return -1;
    if (!m->d_one_counter) {
        HIPCHK(hipMalloc((void **)&m->d_one_counter, sizeof(unsigned)));
        // zeroed ON THE STREAM the single-state kernels run on, and waited for: a hipMemset on the null stream is not ordered
        // with that (non-blocking) stream, and a kernel that started on an unset counter never recognises its last workgroup
        // ("single-state kernel finished without reporting completion", seen once in a group of five one-member shards)
        HIPCHK(hipMemsetAsync(m->d_one_counter, 0, sizeof(unsigned), m->st_comp));
        HIPCHK(hipStreamSynchronize(m->st_comp));
    }
    return 0;
}

static bool one_state_possible(const qgs_model *m, bool jac)
{
    if (m->kernel_kind != 0 || m->ndim > 8190) return false;     // an explicit kernel family keeps the batched route of that family
    return !jac || m->p_lut != nullptr;
}

static int tendencies_one(qgs_model *m, const double *x, double *dx)
{
    const int nd = m->ndim;
    if (one_state_ready(m, 2 * (size_t)nd)) return -1;
    hipStream_t st = m->st_comp;
    std::memcpy(m->h_pin + 1, x, sizeof(double) * nd);
    const unsigned long long seq = ++m->one_seq;
    qgs::launch_gen_tend_one(m->dT.view(), nd, m->d_pin + 1, m->d_pin + 1 + nd, m->d_one_counter, (unsigned long long *)m->d_pin, seq, st);
    if (m->last.name != "gen_tend_one_kernel") note_kernel(m, "gen_tend_one_kernel", nullptr);
    HIPCHK(hipGetLastError());
    if (one_state_wait(m, st, seq)) return -1;
    std::memcpy(dx, m->h_pin + 1 + nd, sizeof(double) * nd);
    return 0;
}

static int jacobian_one(qgs_model *m, const double *x, double *jac)
{
    const int nd = m->ndim;
    const size_t nn = (size_t)nd * nd;
    if (one_state_ready(m, (size_t)nd + nn)) return -1;
    hipStream_t st = m->st_comp;
    std::memcpy(m->h_pin + 1, x, sizeof(double) * nd);
    const unsigned long long seq = ++m->one_seq;
    const qgs::OnePairs P{m->p_lut, m->p_ptr, m->p_idx, m->p_val, m->p_idx2};
    qgs::launch_gen_jac_one(P, nd, m->d_pin + 1, m->d_pin + 1 + nd, m->d_one_counter, (unsigned long long *)m->d_pin, seq, st);
    if (m->last.name != "gen_jac_one_kernel") note_kernel(m, "gen_jac_one_kernel", nullptr);
    HIPCHK(hipGetLastError());
    if (one_state_wait(m, st, seq)) return -1;
    std::memcpy(jac, m->h_pin + 1 + nd, sizeof(double) * nn);
    return 0;
}

int qgs_tendencies(qgs_model *m, int64_t n_traj, const double *x, double *dx)
{
    if (!m || !x || !dx || n_traj < 1) return fail("bad arguments");
    HIPCHK(hipSetDevice(m->device));
    if (n_traj == 1 && one_state_possible(m, false)) return tendencies_one(m, x, dx);
    const int64_t ld = round_ld(n_traj);
    const size_t rows_b = sizeof(double) * (size_t)n_traj * m->ndim, modes_b = sizeof(double) * (size_t)ld * m->ndim;
    if (m->b_in_rows.ensure(rows_b) || m->b_in_modes.ensure(modes_b) || m->b_rec_modes.ensure(modes_b)) return -1;
    if (copy_h2d(m->b_in_rows.p, x, rows_b)) return -1;
    if (qgs_pack_states(m, n_traj, ld, m->b_in_rows.f64(), m->b_in_modes.f64(), nullptr)) return -1;
    if (qgs_tendencies_device(m, n_traj, ld, m->b_in_modes.f64(), m->b_rec_modes.f64(), nullptr)) return -1;
    if (qgs_unpack_states(m, n_traj, ld, m->b_rec_modes.f64(), m->b_in_rows.f64(), nullptr)) return -1;
    if (copy_d2h(dx, m->b_in_rows.p, rows_b)) return -1;
    return 0;
}

int qgs_jacobian(qgs_model *m, int64_t n_traj, const double *x, double *jac)
{
    if (!m || !x || !jac || n_traj < 1) return fail("bad arguments");
    HIPCHK(hipSetDevice(m->device));
    if (m->J.empty()) return fail("model was created without a Jacobian tensor");
    if (n_traj == 1 && one_state_possible(m, true)) return jacobian_one(m, x, jac);
    const int64_t ld = round_ld(n_traj), nn = (int64_t)m->ndim * m->ndim;
    const size_t rows_b = sizeof(double) * (size_t)n_traj * m->ndim, modes_b = sizeof(double) * (size_t)ld * m->ndim;
    const size_t jm_b = sizeof(double) * (size_t)ld * nn, jr_b = sizeof(double) * (size_t)n_traj * nn;
    if (m->b_in_rows.ensure(rows_b) || m->b_in_modes.ensure(modes_b) || m->b_fm_modes.ensure(jm_b) || m->b_fm_rows.ensure(jr_b)) return -1;
    if (copy_h2d(m->b_in_rows.p, x, rows_b)) return -1;
    if (qgs_pack_states(m, n_traj, ld, m->b_in_rows.f64(), m->b_in_modes.f64(), nullptr)) return -1;
    if (jacobian_device(m, n_traj, ld, m->b_in_modes.f64(), m->b_fm_modes.f64(), nullptr)) return -1;
    qgs::launch_unpack_records(nn, n_traj, ld, 1, m->b_fm_modes.f64(), m->b_fm_rows.f64(), nullptr);
    HIPCHK(hipGetLastError());
    if (copy_d2h(jac, m->b_fm_rows.p, jr_b)) return -1;
    return 0;
}

// ---- record windows: the host-layout integrations --------------------------------------------------------------------------
// The reference's record is a host array (integrate.py:196, integrator.py:378-395): its size limit is host memory, not HBM.  The
// steppers therefore write W records at a time into one of two mode-major device windows; while window k + 1 is being
// computed (compute stream), window k is transposed into the caller's (n_traj, n_inner, n_records) layout and leaves the device
// (copy stream):
//   * destination the GPU can address (page-locked host block of qgs_host_register, or device memory): the unpack kernel stores
//     straight into it, runs of W doubles at record offset lo -- no second copy of the record on the device at all;
//   * pageable host memory: unpack into a device staging block, then one (strided) device-to-host copy.
// W = what QGS_HIP_RECORD_WINDOW_MB (default 8192) pays for; a record that fits is one window.
struct WindowPlan {
    int64_t n_records = 1, n_steps = 0, write_steps = 0, W = 1, n_windows = 1;
    int backward = 0;
    // window k: directed records [lo, hi), steps [sb, se), final record included?, first stored record index
    void window(int64_t k, int64_t *lo, int64_t *hi, int64_t *sb, int64_t *se, int *wf, int64_t *lo_s) const
    {
        *lo = k * W;
        *hi = std::min(n_records, *lo + W);
        *wf = (*hi == n_records) ? 1 : 0;
        *sb = write_steps > 0 ? std::min(n_steps, *lo * write_steps) : 0;
        *se = *wf ? n_steps : std::min(n_steps, *hi * write_steps);
        *lo_s = backward ? (n_records - *hi) : *lo;
    }
};

static WindowPlan plan_windows(const qgs_model *m, int64_t n_records, int64_t n_steps, int64_t write_steps, int backward,
                               size_t bytes_per_record, int buffers, size_t budget = 0)
{
    WindowPlan p;
    p.n_records = n_records; p.n_steps = n_steps; p.write_steps = write_steps; p.backward = backward;
    const size_t per = std::max<size_t>(1, bytes_per_record * (size_t)buffers);
    p.W = (int64_t)std::max<size_t>(1, (budget ? budget : m->tune.window_bytes) / per);
    if (p.W >= n_records) { p.W = n_records; p.n_windows = 1; }
    else p.n_windows = (n_records + p.W - 1) / p.W;
    return p;
}

// device-side address of a destination block of `bytes` bytes: the block itself when it is device memory, its mapped alias when
// it lies inside a host block this library has page-locked (qgs_host_alloc / qgs_host_register), null for any other host memory
static double *device_alias(qgs_model *m, double *dst, size_t bytes, bool *is_device = nullptr)
{
    if (is_device) *is_device = false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dst) == hipSuccess) {
        if (at.type == hipMemoryTypeDevice) { if (is_device) *is_device = true; return dst; }
    } else (void)hipGetLastError();
    if (m->tune.d2h_mode == 2 || !registry_covers(dst, bytes)) return nullptr;
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, dst, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return (double *)dp;
}

// Which way a page-locked host block is filled (measured, tools/d2h_routes.py, 1.9 GB of records, PCIe floor 33.4 ms): one
// window -> staging + one contiguous DMA copy, 35.6 ms (the unpack kernel's own stores over PCIe: 38.0 ms); several windows ->
// the kernel's stores, whose runs of W doubles per (member, variable) cost 40 / 49 ms at W = 105 / 13, where the strided
// copy of the staged window costs 50 / 119 ms.  So: the copy when the whole record fits the budget three times, else stores.
static bool prefer_copy_route(const qgs_model *m, bool dst_is_device, int64_t n_records, size_t bytes_per_record)
{
    if (dst_is_device || m->tune.d2h_mode == 1) return false;
    return m->tune.window_bytes / std::max<size_t>(1, bytes_per_record * 3) >= (size_t)n_records;
}

// A pageable destination is never page-locked by this library (rounds 3-4 hipHostRegister'ed blocks of 32 MB and more for the
// duration of a call, on the assumption that the C library serves such blocks from mappings of their own; a kernel storing into
// a registered heap block is where every GPU write fault of round 4 was found).  hipHostRegister happens in
// qgs_host_register only -- the caller's explicit request, with the guarantees include/qgs_hip.h lists -- and a pageable
// block is filled by host threads from page-locked bounce blocks (host_bridge.h), window k on its way while window k + 1 is
// computed: 189 GB of records reach pageable memory at tools/big_record.py's rate in profiles/r05_big_record.txt.

// Whatever way a pipelined call ends -- also on an error in the middle of it -- nothing of it may still be in flight when its
// buffers or the caller's blocks go away: both streams, and every window still with the drain thread.
struct DrainGuard {
    qgs_model *m;
    hipStream_t a, b;
    bool armed = true;          // (false: the caller waits itself, after more work has been enqueued -- member groups)
    ~DrainGuard()
    {
        if (!armed) return;
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        for (int64_t t : m->drain_tickets) (void)qgs::bridge_wait_done(t, nullptr);
        m->drain_tickets.clear();
    }
};

// every window handed to the drain thread has arrived in the caller's memory (first failure reported)
static int drain_finish(qgs_model *m)
{
    int rc = 0;
    std::string err, first;
    for (int64_t t : m->drain_tickets)
        if (qgs::bridge_wait_done(t, &err) && !rc) { rc = -1; first = err; }
    m->drain_tickets.clear();
    return rc ? fail(first) : 0;
}

// staging blocks of qgs_unpack_window_enqueue (nothing of them in flight any more)
static void release_drain_staging(qgs_model *m)
{
    for (auto &b : m->drain_pool) {
        auto it = m->drain_slots.find(b.get());
        if (it != m->drain_slots.end()) {
            if (it->second.ev) (void)hipEventDestroy(it->second.ev);
            m->drain_slots.erase(it);
        }
        b->release();
    }
    m->drain_pool.clear();
}

// A staging block of `need` bytes for the next window of qgs_unpack_window_enqueue.  The windows of one flush (vectors, states,
// exponents of a record window) and of consecutive flushes (the next record window, the next member group) want different blocks,
// so that none of them waits for the DMA of another inside the call -- the host would sit through a transfer and enqueue the next
// kernels only afterwards.  In this order: an idle block (its last window has left the device) of about the right size; a new
// block while the pool stays within a quarter of the device's memory; any idle block (grown if too small); else the block whose
// window was handed to the drain thread first (drain_window waits for that window to have left the device).
static Buffer *acquire_drain_staging(qgs_model *m, size_t need)
{
    size_t pool_bytes = 0;
    Buffer *fit = nullptr, *loose = nullptr, *small = nullptr, *oldest = nullptr;
    int64_t oldest_ticket = 0;
    for (auto &b : m->drain_pool) {
        pool_bytes += b->cap;
        const DrainSlot &s = m->drain_slots[b.get()];
        if (s.ticket == 0 || qgs::bridge_poll_copied(s.ticket)) {
            if (b->cap >= need && b->cap / 4 <= need) { if (!fit || b->cap < fit->cap) fit = b.get(); }
            else if (b->cap >= need) { if (!loose || b->cap < loose->cap) loose = b.get(); }
            else if (!small || b->cap > small->cap) small = b.get();
        } else if (!oldest || s.ticket < oldest_ticket) {
            oldest = b.get();
            oldest_ticket = s.ticket;
        }
    }
    if (fit) return fit;
    size_t free_b = 0, total_b = 0;
    const bool room = m->drain_pool.size() < 16 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && pool_bytes + need <= total_b / 4 &&
                      need <= free_b / 2;
    if (room || m->drain_pool.empty()) {
        m->drain_pool.push_back(std::make_unique<Buffer>());
        return m->drain_pool.back().get();
    }
    if (loose) return loose;
    if (small) return small;
    return oldest;
}

// one window of records leaves the device (enqueued on st): alias != null -> stores of the unpack kernel; else staging, then
// either one copy (page-locked destination, or a single window) or the bounce ring of host_bridge.h (pageable destination)
static int drain_window(qgs_model *m, int64_t n_inner, int64_t n_traj, int64_t ld, int64_t Wk, int64_t n_records, int64_t lo_s,
                        const double *d_win, double *alias, double *dst_host, Buffer &staging, hipStream_t st)
{
    if (alias) {
        qgs::launch_unpack_window(n_inner, n_traj, ld, Wk, n_records, d_win, alias + lo_s, st);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const size_t rows = (size_t)n_traj * (size_t)n_inner;
    const bool pinned = registry_covers(dst_host, sizeof(double) * rows * (size_t)n_records);
    // the staging block is read by the drain thread until the last byte of its previous window has left the device
    std::string err;
    DrainSlot &slot = m->drain_slots[&staging];
    if (slot.ticket > 0 && qgs::bridge_wait_copied(slot.ticket, &err)) return fail(err);
    // (and whatever a stream still does with the block -- the copy of a page-locked destination below, enqueued by a caller on
    // another stream: the blocks of qgs_unpack_window_enqueue are shared by all callers of the model)
    if (slot.ev) HIPCHK(hipStreamWaitEvent(st, slot.ev, 0));
    if (staging.cap < sizeof(double) * rows * (size_t)Wk && slot.ev) HIPCHK(hipEventSynchronize(slot.ev));     // (about to be freed)
    if (staging.ensure(sizeof(double) * rows * (size_t)Wk)) return -1;
    qgs::launch_unpack_window(n_inner, n_traj, ld, Wk, Wk, d_win, staging.f64(), st);
    HIPCHK(hipGetLastError());
    if (pinned) {
        // page-locked destination (QGS_HIP_D2H=copy, or one window): one (strided) DMA copy
        if (Wk == n_records) {
            if (copy_d2h(dst_host, staging.p, sizeof(double) * rows * (size_t)Wk, st)) return -1;
        } else {
            HIPCHK(hipMemcpy2DAsync(dst_host + lo_s, sizeof(double) * (size_t)n_records, staging.p, sizeof(double) * (size_t)Wk,
                                    sizeof(double) * (size_t)Wk, rows, hipMemcpyDeviceToHost, st));
        }
        if (!slot.ev) HIPCHK(hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(slot.ev, st));
        return 0;
    }
    // pageable destination: rows of Wk doubles -> runs `n_records` doubles apart, by the device's drain thread.  The event marks
    // the end of the unpack kernel on st; the ticket is waited for by the next window (staging) and at the end of the call.
    // (the slot's event can be recorded again: its previous job has passed its wait, it has even finished reading)
    if (!slot.ev) HIPCHK(hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(slot.ev, st));
    const int64_t ticket = qgs::bridge_d2h_rows_async((char *)(dst_host + lo_s), sizeof(double) * (size_t)n_records, (const char *)staging.p,
                                                      sizeof(double) * (size_t)Wk, rows, slot.ev, &err);
    if (ticket < 0) return fail(err);
    slot.ticket = ticket;
    m->drain_tickets.push_back(ticket);
    return 0;
}

// ic_rows: (n_traj, ndim) in host memory (ic_on_device == 0) or device memory; traj: (n_traj, ndim, n_records) in host memory
// (pageable or page-locked) or device memory.  Blocking.
// (budget != 0: the window budget of this call; defer: return with the last window still on its way -- the caller, who runs the
// next member group first, waits for both streams and the drain thread)
static int rk_windowed(qgs_model *m, int64_t n_traj, const double *ic_rows, int ic_on_device, const double *time, int64_t n_time,
                       int time_direction, int64_t write_steps, int s, const double *b, const double *a, double *traj,
                       size_t budget = 0, bool defer = false)
{
    if (!time || n_time < 1 || !b || !a || s < 1) return fail("bad time grid / tableau");
    if (time_direction != 1 && time_direction != -1) return fail("time_direction must be +1 or -1");
    if (write_steps < 0) return fail("write_steps must be >= 0");
    HIPCHK(hipSetDevice(m->device));
    if (streams_ready(m)) return -1;
    const int nd = m->ndim;
    const int64_t ld = round_ld(n_traj), n_steps = n_time - 1;
    const int64_t n_records = qgs_n_records(time, n_time, write_steps);
    const int backward = time_direction == -1;
    const size_t rows_b = sizeof(double) * (size_t)n_traj * nd, modes_b = sizeof(double) * (size_t)ld * nd;
    hipStream_t sc = m->st_comp, sd = m->st_copy;
    if (m->b_in_modes.ensure(modes_b)) return -1;
    const double *d_rows = ic_rows;
    if (!ic_on_device) {
        if (m->b_in_rows.ensure(rows_b)) return -1;
        if (copy_h2d(m->b_in_rows.p, ic_rows, rows_b, sc)) return -1;
        d_rows = m->b_in_rows.f64();
    }
    qgs::launch_pack_states(nd, n_traj, ld, d_rows, m->b_in_modes.f64(), sc);
    HIPCHK(hipGetLastError());
    const double *d_time, *d_tab_spec, *d_tab_full;
    if (stage_time_tab(m, time, n_time, time_direction, s, b, a, sc, &d_time, &d_tab_spec, &d_tab_full)) return -1;
    bool dst_dev = false;
    double *alias = device_alias(m, traj, sizeof(double) * (size_t)n_traj * nd * (size_t)n_records, &dst_dev);
    if (alias && prefer_copy_route(m, dst_dev, n_records, modes_b)) alias = nullptr;
    DrainGuard drain{m, sc, sd};
    const WindowPlan plan = plan_windows(m, n_records, n_steps, write_steps, backward, modes_b, alias ? 2 : 3, budget);
    m->last_windows = plan.n_windows;
    const int nbuf = plan.n_windows > 1 ? 2 : 1;
    for (int i = 0; i < nbuf; ++i) if (m->b_win[i].ensure(modes_b * (size_t)plan.W)) return -1;
    if (plan.n_windows > 1 && (m->b_state2.ensure(modes_b) || m->b_carry.ensure(modes_b))) return -1;
    const double *y_in = m->b_in_modes.f64();
    for (int64_t k = 0; k < plan.n_windows; ++k) {
        int64_t lo, hi, sb, se, lo_s;
        int wf;
        plan.window(k, &lo, &hi, &sb, &se, &wf, &lo_s);
        const int q = (int)(k & 1);
        double *win = m->b_win[q].f64();
        if (k >= 2) HIPCHK(hipStreamWaitEvent(sc, m->ev_copy[q], 0));            // the window's previous content has left
        double *y_out = (k + 1 < plan.n_windows) ? ((k & 1) ? m->b_carry.f64() : m->b_state2.f64()) : nullptr;
        if (rk_launch(m, n_traj, ld, y_in, y_out, win - lo_s * (int64_t)nd * ld, d_time, d_tab_spec, d_tab_full, sb, se, n_steps,
                      write_steps, n_records, backward, wf, s, a, sc)) return -1;
        if (y_out) y_in = y_out;
        HIPCHK(hipEventRecord(m->ev_comp[q], sc));
        HIPCHK(hipStreamWaitEvent(sd, m->ev_comp[q], 0));
        if (drain_window(m, nd, n_traj, ld, hi - lo, n_records, lo_s, win, alias, traj, m->b_rec_rows, sd)) return -1;
        HIPCHK(hipEventRecord(m->ev_copy[q], sd));
    }
    if (defer) {
        // the next group's kernels go to the compute stream: they must not overwrite this group's window before its unpack has run
        HIPCHK(hipStreamWaitEvent(sc, m->ev_copy[(plan.n_windows - 1) & 1], 0));
        drain.armed = false;
        return 0;
    }
    HIPCHK(hipStreamSynchronize(sc));
    HIPCHK(hipStreamSynchronize(sd));
    return drain_finish(m);
}

// Records of a large ensemble into PAGEABLE host memory leave in member groups: traj is (n_traj, ndim, n_records), so the record of
// a group of members is one contiguous piece of it -- one window per group, streamed front to back by the drain thread while the next
// group is integrated.  Windows of records reach every page of the block once per window, in runs of 8 W bytes (the first window
// takes the page faults of the whole block).  Returns the members per group, or 0: windows of records (few members, a record that
// fits one window, a page-locked or device destination, a window budget set by hand).
static int64_t rk_member_groups(qgs_model *m, int64_t n_traj, size_t per_member, double *dst_a, size_t bytes_a, double *dst_b, size_t bytes_b,
                                size_t *group_budget)
{
    // (per_member: bytes of records per member over all destination blocks; dst_b may be null)
    bool dst_dev = false;
    if (device_alias(m, dst_a, bytes_a, &dst_dev) || dst_dev) return 0;
    if (dst_b && (device_alias(m, dst_b, bytes_b, &dst_dev) || dst_dev)) return 0;
    int64_t g = m->tune.group_members;
    if (g <= 0) {
        if (m->tune.window_by_hand || per_member * (size_t)n_traj * 3 <= m->tune.window_bytes) return 0;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
        // window + staging block of a group within a sixth of the free memory each; sixteen groups or more where that leaves 4 096
        // members per group, never fewer than 1 024
        const int64_t cap = (int64_t)(free_b / 6 / std::max<size_t>(1, per_member)) / 64 * 64;
        g = std::min(cap, std::max<int64_t>(4096, ((n_traj + 15) / 16 + 63) / 64 * 64));
        if (g < 1024) return 0;
    }
    if (g >= n_traj) return 0;
    *group_budget = 3 * per_member * (size_t)round_ld(g) + ((size_t)1 << 20);       // (plan_windows divides by three buffers)
    return g;
}

// after a run in member groups: the windows and staging blocks of a group can be far larger than the budget the model otherwise
// keeps (up to a third of the device's memory together) -- they go back to the device
static void release_group_buffers(qgs_model *m)
{
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    const size_t keep = std::max(m->tune.window_bytes, total_b / 16);          // (a repeated run of the same size finds its blocks again)
    for (Buffer *b : {&m->b_win[0], &m->b_win[1], &m->b_fwin[0], &m->b_fwin[1], &m->b_rec_rows, &m->b_fm_rows})
        if (b->cap > keep) b->release();
}

int qgs_record_window(int64_t n_records, int64_t n_steps, int64_t write_steps, int backward, int64_t W, int64_t k, int64_t *out)
{
    if (!out || n_records < 1 || n_steps < 0 || write_steps < 0 || W < 1 || k < 0) return fail("bad arguments");
    WindowPlan p;
    p.n_records = n_records; p.n_steps = n_steps; p.write_steps = write_steps; p.backward = backward ? 1 : 0;
    p.W = std::min(W, n_records);
    p.n_windows = (n_records + p.W - 1) / p.W;
    if (k >= p.n_windows) return fail("window index out of range");
    int wf;
    p.window(k, &out[0], &out[1], &out[2], &out[3], &wf, &out[5]);
    out[4] = wf;
    return (int)std::min<int64_t>(p.n_windows, 0x7fffffff);
}

int qgs_unpack_window_enqueue(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_inner, int64_t n_window, int64_t n_records,
                              int64_t first_record, const double *d_window, double *dst, void *stream)
{
    if (check_common(m, n_traj, ld)) return -1;
    if (n_inner < 1 || n_inner > (int64_t)65535 * 64 || n_window < 1 || n_records < 1 || first_record < 0 ||
        first_record + n_window > n_records || !d_window || !dst) return fail("bad window arguments");
    HIPCHK(hipSetDevice(m->device));
    double *alias = device_alias(m, dst, sizeof(double) * (size_t)n_traj * (size_t)n_inner * (size_t)n_records);
    if (alias) return drain_window(m, n_inner, n_traj, ld, n_window, n_records, first_record, d_window, alias, dst, m->b_drain, (hipStream_t)stream);
    Buffer *staging = acquire_drain_staging(m, sizeof(double) * (size_t)n_traj * (size_t)n_inner * (size_t)n_window);
    return drain_window(m, n_inner, n_traj, ld, n_window, n_records, first_record, d_window, alias, dst, *staging, (hipStream_t)stream);
}

int qgs_drain_wait(qgs_model *m)
{
    if (!m) return fail("null model");
    const int rc = drain_finish(m);
    release_drain_staging(m);          // the run is over: its staging blocks (a window of records each) go back to the device
    return rc;
}

int qgs_unpack_window(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_inner, int64_t n_window, int64_t n_records,
                      int64_t first_record, const double *d_window, double *dst, void *stream)
{
    const int rc = qgs_unpack_window_enqueue(m, n_traj, ld, n_inner, n_window, n_records, first_record, d_window, dst, stream);
    if (!m) return rc;
    if (rc) { const std::string e = g_err; (void)drain_finish(m); return fail(e); }
    return drain_finish(m);          // (a pageable destination: the window has arrived when the call returns)
}

int qgs_rk_integrate(qgs_model *m, int64_t n_traj, const double *ic, const double *time, int64_t n_time,
                     int time_direction, int64_t write_steps, int s, const double *b, const double *c, const double *a,
                     double *traj)
{
    (void)c;
    if (!m || !ic || !traj || n_traj < 1) return fail("bad arguments");
    if (time && n_time >= 1 && write_steps >= 0) {
        HIPCHK(hipSetDevice(m->device));
        const int64_t n_records = qgs_n_records(time, n_time, write_steps);
        size_t budget = 0;
        const size_t per_member = sizeof(double) * (size_t)m->ndim * (size_t)n_records;
        const int64_t g = rk_member_groups(m, n_traj, per_member, traj, per_member * (size_t)n_traj, nullptr, 0, &budget);
        if (g > 0) {
            int rc = 0;
            int64_t windows = 0;
            for (int64_t lo = 0; lo < n_traj && !rc; lo += g) {
                const int64_t cnt = std::min(g, n_traj - lo);
                rc = rk_windowed(m, cnt, ic + lo * m->ndim, 0, time, n_time, time_direction, write_steps, s, b, a,
                                 traj + lo * m->ndim * n_records, budget, true);
                windows = std::max(windows, m->last_windows);
            }
            const std::string err = rc ? g_err : std::string();
            // whatever way the loop ended: nothing of it is in flight when the caller's block is handed back
            if (m->st_comp) (void)hipStreamSynchronize(m->st_comp);
            if (m->st_copy) (void)hipStreamSynchronize(m->st_copy);
            const int rd = drain_finish(m);
            release_group_buffers(m);
            m->last_windows = windows;
            m->last_groups = (n_traj + g - 1) / g;
            return rc ? fail(err) : rd;
        }
        m->last_groups = 1;
    }
    return rk_windowed(m, n_traj, ic, 0, time, n_time, time_direction, write_steps, s, b, a, traj);
}

int qgs_rk_integrate_rows_device(qgs_model *m, int64_t n_traj, const double *d_ic_rows, const double *time, int64_t n_time,
                                 int time_direction, int64_t write_steps, int s, const double *b, const double *c, const double *a,
                                 double *d_traj_rows)
{
    (void)c;
    if (!m || !d_ic_rows || !d_traj_rows || n_traj < 1) return fail("bad arguments");
    return rk_windowed(m, n_traj, d_ic_rows, 1, time, n_time, time_direction, write_steps, s, b, a, d_traj_rows);
}

int qgs_rk_integrate_moments(qgs_model *m, int64_t n_traj, const double *ic, const double *time, int64_t n_time,
                             int time_direction, int64_t write_steps, int s, const double *b, const double *c, const double *a,
                             double *mean, double *var, double *final_states)
{
    (void)c;
    if (!m || !ic || !mean || n_traj < 1) return fail("bad arguments");
    if (!time || n_time < 1 || !b || !a || s < 1) return fail("bad time grid / tableau");
    if (time_direction != 1 && time_direction != -1) return fail("time_direction must be +1 or -1");
    if (write_steps < 0) return fail("write_steps must be >= 0");
    HIPCHK(hipSetDevice(m->device));
    if (streams_ready(m)) return -1;
    const int nd = m->ndim;
    const int64_t ld = round_ld(n_traj), n_steps = n_time - 1;
    const int64_t n_records = qgs_n_records(time, n_time, write_steps);
    const int64_t n_rows = n_records * nd;
    const int backward = time_direction == -1;
    const size_t rows_b = sizeof(double) * (size_t)n_traj * nd, modes_b = sizeof(double) * (size_t)ld * nd;
    hipStream_t sc = m->st_comp;
    // the record never exists as a whole: window after window is integrated and reduced in place (the rows of the moments are
    // (record, variable) pairs, so a window's rows are its own); one window buffer, one stream
    const WindowPlan plan = plan_windows(m, n_records, n_steps, write_steps, backward, modes_b, 1);
    m->last_windows = plan.n_windows;
    if (m->b_in_rows.ensure(rows_b) || m->b_in_modes.ensure(modes_b) || m->b_win[0].ensure(modes_b * (size_t)plan.W) ||
        m->b_state2.ensure(modes_b) || m->b_carry.ensure(modes_b) || m->b_mom_out.ensure(sizeof(double) * 2 * (size_t)n_rows)) return -1;
    if (copy_h2d(m->b_in_rows.p, ic, rows_b, sc)) return -1;
    qgs::launch_pack_states(nd, n_traj, ld, m->b_in_rows.f64(), m->b_in_modes.f64(), sc);
    HIPCHK(hipGetLastError());
    const double *d_time, *d_tab_spec, *d_tab_full;
    if (stage_time_tab(m, time, n_time, time_direction, s, b, a, sc, &d_time, &d_tab_spec, &d_tab_full)) return -1;
    double *d_mean = m->b_mom_out.f64(), *d_var = d_mean + n_rows, *win = m->b_win[0].f64();
    const double *y_in = m->b_in_modes.f64();
    double *y_last = nullptr;
    for (int64_t k = 0; k < plan.n_windows; ++k) {
        int64_t lo, hi, sb, se, lo_s;
        int wf;
        plan.window(k, &lo, &hi, &sb, &se, &wf, &lo_s);
        double *y_out = (k & 1) ? m->b_carry.f64() : m->b_state2.f64();        // always: the final states may be wanted
        if (rk_launch(m, n_traj, ld, y_in, y_out, win - lo_s * (int64_t)nd * ld, d_time, d_tab_spec, d_tab_full, sb, se, n_steps,
                      write_steps, n_records, backward, wf, s, a, sc)) return -1;
        y_in = y_last = y_out;
        const int64_t rows_k = (hi - lo) * nd;
        if (rows_k > 0x7fffffff) return fail("too many rows in a record window");
        if (m->b_mom_part.ensure(sizeof(double) * 2 * (size_t)rows_k * (size_t)qgs::moments_splits(rows_k, n_traj))) return -1;
        qgs::launch_moments(rows_k, n_traj, ld, win, m->b_mom_part.f64(), d_mean + lo_s * nd, var ? d_var + lo_s * nd : nullptr, sc);
        HIPCHK(hipGetLastError());
    }
    // device rows are (record, mode); the reference's axis order is (mode, record)
    std::vector<double> h((size_t)n_rows * 2);
    if (copy_d2h(h.data(), d_mean, sizeof(double) * (size_t)n_rows * (var ? 2 : 1), sc)) return -1;
    if (final_states) {
        // the state after the last step of the directed run
        qgs::launch_unpack_states(nd, n_traj, ld, y_last, m->b_in_rows.f64(), sc);
        HIPCHK(hipGetLastError());
        if (copy_d2h(final_states, m->b_in_rows.p, rows_b, sc)) return -1;
    }
    HIPCHK(hipStreamSynchronize(sc));
    for (int64_t r = 0; r < n_records; ++r)
        for (int d = 0; d < nd; ++d) {
            mean[(int64_t)d * n_records + r] = h[(size_t)(r * nd + d)];
            if (var) var[(int64_t)d * n_records + r] = h[(size_t)(n_rows + r * nd + d)];
        }
    return 0;
}

} // extern "C"

// (budget, defer: as rk_windowed)
static int tgls_windowed(qgs_model *m, int64_t n_traj, int64_t n_tg, const double *ic, const double *tg_ic,
                         const double *time, int64_t n_time, int time_direction, int64_t write_steps, int s,
                         const double *b, const double *a, int adjoint, double inverse, double *traj,
                         double *fmatrix, size_t budget = 0, bool defer = false)
{
    HIPCHK(hipSetDevice(m->device));
    if (streams_ready(m)) return -1;
    const int nd = m->ndim;
    const int64_t ld = round_ld(n_traj), n_steps = n_time - 1, n_inner = (int64_t)nd * n_tg;
    const int64_t n_records = qgs_n_records(time, n_time, write_steps);
    const int backward = time_direction == -1;
    const size_t rows_b = sizeof(double) * (size_t)n_traj * nd, modes_b = sizeof(double) * (size_t)ld * nd;
    const size_t tg_rows_b = rows_b * (size_t)n_tg, tg_modes_b = modes_b * (size_t)n_tg;
    hipStream_t sc = m->st_comp, sd = m->st_copy;
    if (m->b_in_rows.ensure(rows_b) || m->b_in_modes.ensure(modes_b) || m->b_tg_rows.ensure(tg_rows_b) || m->b_tg_modes.ensure(tg_modes_b)) return -1;
    if (copy_h2d(m->b_in_rows.p, ic, rows_b, sc) || copy_h2d(m->b_tg_rows.p, tg_ic, tg_rows_b, sc)) return -1;
    qgs::launch_pack_states(nd, n_traj, ld, m->b_in_rows.f64(), m->b_in_modes.f64(), sc);
    // padding lanes of the tangent state are read (never stored) by the specialised kernel: the pack defines them
    qgs::launch_pack_tangent(nd, n_tg, n_traj, ld, m->b_tg_rows.f64(), m->b_tg_modes.f64(), sc);
    HIPCHK(hipGetLastError());
    const double *d_time, *d_tab_spec, *d_tab_full;
    if (stage_time_tab(m, time, n_time, time_direction, s, b, a, sc, &d_time, &d_tab_spec, &d_tab_full)) return -1;
    bool dev_t = false, dev_f = false;
    double *alias_t = device_alias(m, traj, rows_b * (size_t)n_records, &dev_t), *alias_f = device_alias(m, fmatrix, tg_rows_b * (size_t)n_records, &dev_f);
    if (prefer_copy_route(m, dev_t || dev_f, n_records, modes_b + tg_modes_b)) alias_t = alias_f = nullptr;
    DrainGuard drain{m, sc, sd};
    const WindowPlan plan = plan_windows(m, n_records, n_steps, write_steps, backward, modes_b + tg_modes_b,
                                         (alias_t && alias_f) ? 2 : 3, budget);
    m->last_windows = plan.n_windows;
    const int nbuf = plan.n_windows > 1 ? 2 : 1;
    for (int i = 0; i < nbuf; ++i)
        if (m->b_win[i].ensure(modes_b * (size_t)plan.W) || m->b_fwin[i].ensure(tg_modes_b * (size_t)plan.W)) return -1;
    for (int64_t k = 0; k < plan.n_windows; ++k) {
        int64_t lo, hi, sb, se, lo_s;
        int wf;
        plan.window(k, &lo, &hi, &sb, &se, &wf, &lo_s);
        const int q = (int)(k & 1);
        double *win = m->b_win[q].f64(), *fwin = m->b_fwin[q].f64();
        if (k >= 2) HIPCHK(hipStreamWaitEvent(sc, m->ev_copy[q], 0));
        if (tgls_launch(m, n_traj, ld, n_tg, m->b_in_modes.f64(), m->b_tg_modes.f64(), d_time, d_tab_spec, d_tab_full, sb, se,
                        n_steps, k == 0, wf, write_steps, n_records, backward, s, a, adjoint, inverse,
                        win - lo_s * (int64_t)nd * ld, fwin - lo_s * n_inner * ld, sc)) return -1;
        HIPCHK(hipEventRecord(m->ev_comp[q], sc));
        HIPCHK(hipStreamWaitEvent(sd, m->ev_comp[q], 0));
        if (drain_window(m, nd, n_traj, ld, hi - lo, n_records, lo_s, win, alias_t, traj, m->b_rec_rows, sd)) return -1;
        if (drain_window(m, n_inner, n_traj, ld, hi - lo, n_records, lo_s, fwin, alias_f, fmatrix, m->b_fm_rows, sd)) return -1;
        HIPCHK(hipEventRecord(m->ev_copy[q], sd));
    }
    if (defer) {
        HIPCHK(hipStreamWaitEvent(sc, m->ev_copy[(plan.n_windows - 1) & 1], 0));
        drain.armed = false;
        return 0;
    }
    HIPCHK(hipStreamSynchronize(sc));
    HIPCHK(hipStreamSynchronize(sd));
    return drain_finish(m);
}

extern "C" {

int qgs_rk_tgls_integrate(qgs_model *m, int64_t n_traj, int64_t n_tg, const double *ic, const double *tg_ic,
                          const double *time, int64_t n_time, int time_direction, int64_t write_steps, int s,
                          const double *b, const double *c, const double *a, int adjoint, double inverse, double *traj,
                          double *fmatrix)
{
    (void)c;
    if (!m || !ic || !tg_ic || !traj || !fmatrix || n_traj < 1 || n_tg < 1) return fail("bad arguments");
    if (!time || n_time < 1 || !b || !a || s < 1) return fail("bad time grid / tableau");
    if (time_direction != 1 && time_direction != -1) return fail("time_direction must be +1 or -1");
    if (write_steps < 0) return fail("write_steps must be >= 0");
    if (m->J.empty()) return fail("model was created without a Jacobian tensor");
    if ((int64_t)m->ndim * n_tg > (int64_t)65535 * 64) return fail("ndim * n_tg too large for the layout conversion kernels");
    HIPCHK(hipSetDevice(m->device));
    // records of a large ensemble into pageable memory: member groups, as in qgs_rk_integrate (both blocks are member-major)
    const int64_t n_records = qgs_n_records(time, n_time, write_steps);
    size_t budget = 0;
    const size_t per_traj = sizeof(double) * (size_t)m->ndim * (size_t)n_records, per_fm = per_traj * (size_t)n_tg;
    const int64_t g = rk_member_groups(m, n_traj, per_traj + per_fm, traj, per_traj * (size_t)n_traj, fmatrix, per_fm * (size_t)n_traj, &budget);
    if (g > 0) {
        int rc = 0;
        int64_t windows = 0;
        const int64_t nd = m->ndim;
        for (int64_t lo = 0; lo < n_traj && !rc; lo += g) {
            const int64_t cnt = std::min(g, n_traj - lo);
            rc = tgls_windowed(m, cnt, n_tg, ic + lo * nd, tg_ic + lo * nd * n_tg, time, n_time, time_direction, write_steps, s, b, a, adjoint,
                               inverse, traj + lo * nd * n_records, fmatrix + lo * nd * n_tg * n_records, budget, true);
            windows = std::max(windows, m->last_windows);
        }
        const std::string err = rc ? g_err : std::string();
        if (m->st_comp) (void)hipStreamSynchronize(m->st_comp);
        if (m->st_copy) (void)hipStreamSynchronize(m->st_copy);
        const int rd = drain_finish(m);
        release_group_buffers(m);
        m->last_windows = windows;
        m->last_groups = (n_traj + g - 1) / g;
        return rc ? fail(err) : rd;
    }
    m->last_groups = 1;
    return tgls_windowed(m, n_traj, n_tg, ic, tg_ic, time, n_time, time_direction, write_steps, s, b, a, adjoint, inverse, traj, fmatrix);
}

// ---- the general contraction: sparse_mul3 / sparse_mul5 / sparse_mul2 / sparse_mul4 with any vectors -----------------------
}  // extern "C"

struct qgs_contraction {
    int device = 0, n_slots = 0, n_fac = 0, n_out = 0;
    int64_t out_len = 0;
    int32_t *d_out_index = nullptr, *d_ptr = nullptr;
    uint32_t *d_fidx = nullptr;
    double *d_val = nullptr, *d_vecs = nullptr, *d_res = nullptr;
};

extern "C" {

int qgs_contraction_create(int device, int n_slots, int rank, int n_out_axes, int64_t nnz, const int32_t *coo, const double *val,
                           qgs_contraction **out)
{
    if (!out) return fail("out is null");
    *out = nullptr;
    if (rank != 3 && rank != 5) return fail("tensor rank must be 3 or 5");
    if (n_out_axes != 1 && n_out_axes != 2) return fail("the result has 1 (vector) or 2 (matrix) axes");
    if (n_slots < 1 || n_slots > 46340) return fail("n_slots out of range");
    if (nnz < 0 || nnz > 0x7fffffff || (nnz > 0 && (!coo || !val))) return fail("bad tensor arguments");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail("no HIP device visible; libqgs_hip has no CPU path");
    if (device < 0 || device >= n) return fail("device index out of range");
    for (int64_t e = 0; e < nnz * rank; ++e)
        if (coo[e] < 0 || coo[e] >= n_slots) return fail("tensor coordinate out of range");
    HIPCHK(hipSetDevice(device));
    const int n_fac = rank - n_out_axes;
    // entries grouped by output element, incoming order kept inside a group (stable counting sort)
    const int64_t out_len = n_out_axes == 1 ? n_slots : (int64_t)n_slots * n_slots;
    auto out_of = [&](int64_t e) { const int32_t *q = coo + e * rank; return n_out_axes == 1 ? (int64_t)q[0] : (int64_t)q[0] * n_slots + q[1]; };
    std::map<int64_t, int32_t> group;                          // output element -> group number, in order of the element
    for (int64_t e = 0; e < nnz; ++e) group.emplace(out_of(e), 0);
    std::vector<int32_t> out_index, ptr(1, 0);
    for (auto &kv : group) { kv.second = (int32_t)out_index.size(); out_index.push_back((int32_t)kv.first); }
    std::vector<int32_t> count(out_index.size(), 0);
    for (int64_t e = 0; e < nnz; ++e) count[(size_t)group[out_of(e)]]++;
    for (int32_t c : count) ptr.push_back(ptr.back() + c);
    std::vector<int32_t> pos(ptr.begin(), ptr.end() - 1);
    std::vector<uint32_t> fidx((size_t)nnz * n_fac);
    std::vector<double> v((size_t)nnz);
    for (int64_t e = 0; e < nnz; ++e) {
        const int32_t at = pos[(size_t)group[out_of(e)]]++;
        for (int f = 0; f < n_fac; ++f) fidx[(size_t)at * n_fac + f] = (uint32_t)coo[e * rank + n_out_axes + f];
        v[(size_t)at] = val[e];
    }
    qgs_contraction *c = new qgs_contraction();
    c->device = device; c->n_slots = n_slots; c->n_fac = n_fac; c->n_out = (int)out_index.size(); c->out_len = out_len;
    if (upload_vec(out_index, &c->d_out_index) || upload_vec(ptr, &c->d_ptr) || upload_vec(fidx, &c->d_fidx) || upload_vec(v, &c->d_val) ||
        hipMalloc((void **)&c->d_vecs, sizeof(double) * (size_t)n_fac * n_slots) != hipSuccess ||
        hipMalloc((void **)&c->d_res, sizeof(double) * (size_t)out_len) != hipSuccess) {
        qgs_contraction_destroy(c);
        return fail("device allocation for the contraction failed");
    }
    *out = c;
    return 0;
}

int qgs_contraction_apply(qgs_contraction *c, const double *vecs, double *res)
{
    if (!c || !vecs || !res) return fail("bad arguments");
    HIPCHK(hipSetDevice(c->device));
    if (copy_h2d(c->d_vecs, vecs, sizeof(double) * (size_t)c->n_fac * c->n_slots)) return -1;
    HIPCHK(hipMemsetAsync(c->d_res, 0, sizeof(double) * (size_t)c->out_len, nullptr));
    qgs::launch_contract(c->n_out, c->d_out_index, c->d_ptr, c->d_fidx, c->d_val, c->n_fac, c->d_vecs, c->n_slots, c->d_res, nullptr);
    HIPCHK(hipGetLastError());
    if (copy_d2h(res, c->d_res, sizeof(double) * (size_t)c->out_len)) return -1;
    return 0;
}

int qgs_contraction_destroy(qgs_contraction *c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    for (void *q : {(void *)c->d_out_index, (void *)c->d_ptr, (void *)c->d_fidx, (void *)c->d_val, (void *)c->d_vecs, (void *)c->d_res})
        if (q) (void)hipFree(q);
    delete c;
    return 0;
}

int qgs_host_alloc(int64_t bytes, void **out)
{
    if (!out || bytes <= 0) return fail("bad arguments");
    *out = nullptr;
    void *p = nullptr;
    HIPCHK(hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable | hipHostMallocMapped));
    registry_add(p, (size_t)bytes);
    *out = p;
    return 0;
}

int qgs_host_free(void *ptr)
{
    if (!ptr) return 0;
    registry_remove(ptr);
    HIPCHK(hipHostFree(ptr));
    return 0;
}

int qgs_host_register(void *ptr, int64_t bytes)
{
    if (!ptr || bytes <= 0) return fail("bad arguments");
    // portable + mapped: every GPU of the node can store into the block (the unpack kernels of all shards write their slices)
    HIPCHK(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    registry_add(ptr, (size_t)bytes);
    return 0;
}

int qgs_memcpy_h2d(int device, void *d_dst, const void *h_src, int64_t bytes, void *stream)
{
    if (bytes < 0 || (bytes > 0 && (!d_dst || !h_src))) return fail("bad arguments");
    HIPCHK(hipSetDevice(device));
    if (copy_h2d(d_dst, h_src, (size_t)bytes, (hipStream_t)stream)) return -1;
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int qgs_memcpy_d2h(int device, void *h_dst, const void *d_src, int64_t bytes, void *stream)
{
    if (bytes < 0 || (bytes > 0 && (!h_dst || !d_src))) return fail("bad arguments");
    HIPCHK(hipSetDevice(device));
    if (copy_d2h(h_dst, d_src, (size_t)bytes, (hipStream_t)stream)) return -1;
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int qgs_host_unregister(void *ptr)
{
    if (!ptr) return fail("bad arguments");
    registry_remove(ptr);
    HIPCHK(hipHostUnregister(ptr));
    return 0;
}

// ---- all GPUs of the node behind one handle ------------------------------------------------------------------------------------
// The reference's integrators fan the trajectories out over every core of the machine by default (integrator.py:79-82,
// 133-142, 386-395).  A qgs_group is the same for GPUs: one qgs_model per listed device (a device may be listed more than once:
// two models, two pipelines on one GPU), members split into contiguous shards (remainder to the first shards), one host
// thread per shard driving that shard's windowed pipeline; every shard reads its slice of the caller's input block and
// delivers its slice of the result block itself -- G parallel device-to-host streams, no gather, no collective.
}  // extern "C"

struct qgs_group {
    std::vector<qgs_model *> models;
};

namespace {

void shard_of(int64_t n_total, int n_shards, int i, int64_t *start, int64_t *count)
{
    const int64_t base = n_total / n_shards, rem = n_total % n_shards;
    *start = (int64_t)i * base + std::min<int64_t>(i, rem);
    *count = base + (i < rem ? 1 : 0);
}

// run fn(i, start, count) for every non-empty shard, each on its own thread; first error wins
template <class Fn>
int for_each_shard(qgs_group *g, int64_t n_total, Fn fn)
{
    const int G = (int)g->models.size();
    std::vector<std::string> errs((size_t)G);
    std::vector<int> rcs((size_t)G, 0);
    std::vector<std::thread> th;
    for (int i = 0; i < G; ++i) {
        int64_t start, count;
        shard_of(n_total, G, i, &start, &count);
        if (count < 1) continue;
        th.emplace_back([&, i, start, count] {
            rcs[(size_t)i] = fn(i, start, count);
            if (rcs[(size_t)i]) errs[(size_t)i] = g_err;           // g_err is thread-local: carry the text over
        });
    }
    for (auto &t : th) t.join();
    for (int i = 0; i < G; ++i)
        if (rcs[(size_t)i]) return fail("shard " + std::to_string(i) + " (device " + std::to_string(g->models[(size_t)i]->device) + "): " + errs[(size_t)i]);
    return 0;
}

}  // namespace

extern "C" {

int qgs_group_create(int n_devices, const int *devices, int ndim, int rank, int64_t nnz, const int32_t *coo, const double *val,
                     int64_t jnnz, const int32_t *jcoo, const double *jval, qgs_group **out)
{
    if (!out) return fail("out is null");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 1024 || !devices) return fail("a group needs 1..1024 devices");
    qgs_group *g = new qgs_group();
    g->models.assign((size_t)n_devices, nullptr);
    // the models are built one after the other: the first compiles (or finds) the code objects, the others load them
    for (int i = 0; i < n_devices; ++i)
        if (qgs_model_create_rank(devices[i], ndim, rank, nnz, coo, val, jnnz, jcoo, jval, &g->models[(size_t)i])) {
            const std::string e = g_err;
            qgs_group_destroy(g);
            return fail("device " + std::to_string(devices[i]) + ": " + e);
        }
    *out = g;
    return 0;
}

int qgs_group_destroy(qgs_group *g)
{
    if (!g) return 0;
    for (qgs_model *m : g->models) qgs_model_destroy(m);
    delete g;
    return 0;
}

int qgs_group_size(const qgs_group *g) { return g ? (int)g->models.size() : -1; }

qgs_model *qgs_group_model(qgs_group *g, int i)
{
    if (!g || i < 0 || i >= (int)g->models.size()) { fail("shard index out of range"); return nullptr; }
    return g->models[(size_t)i];
}

int qgs_group_shard(const qgs_group *g, int64_t n_traj, int i, int64_t *start, int64_t *count)
{
    if (!g || i < 0 || i >= (int)g->models.size() || n_traj < 0 || !start || !count) return fail("bad arguments");
    shard_of(n_traj, (int)g->models.size(), i, start, count);
    return 0;
}

int qgs_group_set_kernel(qgs_group *g, int kind)
{
    if (!g) return fail("null group");
    for (qgs_model *m : g->models) if (qgs_model_set_kernel(m, kind)) return -1;
    return 0;
}

int qgs_group_tendencies(qgs_group *g, int64_t n_traj, const double *x, double *dx)
{
    if (!g || !x || !dx || n_traj < 1) return fail("bad arguments");
    const int64_t nd = g->models[0]->ndim;
    return for_each_shard(g, n_traj, [&](int i, int64_t a, int64_t n) { return qgs_tendencies(g->models[(size_t)i], n, x + a * nd, dx + a * nd); });
}

int qgs_group_jacobian(qgs_group *g, int64_t n_traj, const double *x, double *jac)
{
    if (!g || !x || !jac || n_traj < 1) return fail("bad arguments");
    const int64_t nd = g->models[0]->ndim;
    return for_each_shard(g, n_traj, [&](int i, int64_t a, int64_t n) { return qgs_jacobian(g->models[(size_t)i], n, x + a * nd, jac + a * nd * nd); });
}

int qgs_group_rk_integrate(qgs_group *g, int64_t n_traj, const double *ic, const double *time, int64_t n_time, int time_direction,
                           int64_t write_steps, int s, const double *b, const double *c, const double *a, double *traj)
{
    if (!g || !ic || !traj || n_traj < 1) return fail("bad arguments");
    if (!time || n_time < 1) return fail("bad time grid");
    const int64_t nd = g->models[0]->ndim, nrec = qgs_n_records(time, n_time, write_steps);
    return for_each_shard(g, n_traj, [&](int i, int64_t a0, int64_t n) {
        return qgs_rk_integrate(g->models[(size_t)i], n, ic + a0 * nd, time, n_time, time_direction, write_steps, s, b, c, a,
                                traj + a0 * nd * nrec);
    });
}

int qgs_group_rk_tgls_integrate(qgs_group *g, int64_t n_traj, int64_t n_tg, const double *ic, const double *tg_ic,
                                const double *time, int64_t n_time, int time_direction, int64_t write_steps, int s,
                                const double *b, const double *c, const double *a, int adjoint, double inverse, double *traj,
                                double *fmatrix)
{
    if (!g || !ic || !tg_ic || !traj || !fmatrix || n_traj < 1 || n_tg < 1) return fail("bad arguments");
    if (!time || n_time < 1) return fail("bad time grid");
    const int64_t nd = g->models[0]->ndim, nrec = qgs_n_records(time, n_time, write_steps);
    return for_each_shard(g, n_traj, [&](int i, int64_t a0, int64_t n) {
        return qgs_rk_tgls_integrate(g->models[(size_t)i], n, n_tg, ic + a0 * nd, tg_ic + a0 * nd * n_tg, time, n_time,
                                     time_direction, write_steps, s, b, c, a, adjoint, inverse, traj + a0 * nd * nrec,
                                     fmatrix + a0 * nd * n_tg * nrec);
    });
}

// Compile (and cache) the specialised kernels of a model without touching a device: used by
// __graft_entry__.build() on the GPU-less build host so that the code objects travel with the tree.
int qgs_prebuild_rank(int ndim, int rank, int64_t nnz, const int32_t *coo, const double *val, int64_t jnnz, const int32_t *jcoo,
                      const double *jval, int n_stage_counts, const int *stage_counts, const char *arch)
{
    if (rank != 3 && rank != 5) return fail("tensor rank must be 3 or 5");
    qgs_model m;
    m.ndim = ndim;
    m.arch = (arch && *arch) ? arch : target_arch(-1);
    if (load_tensors(&m, rank, nnz, coo, val, jnnz, jcoo, jval, nullptr, nullptr)) return -1;
    apply_env_options(m.cg);
    if (!m.der.t.empty()) m.cg.lds_asm = false;
    if (rank == 5) m.cg.row_split = 1;
    std::vector<int> stages(stage_counts, stage_counts + n_stage_counts);
    classify_model(&m);
    auto build = [&](qgs::Kernel k, int S) {
        if (!prebuild_mine()) return 0;
        std::shared_ptr<const KernelBlob> blob;
        bool cached;
        return model_blob(&m, k, S, &blob, &cached, BlobMode::Publish);
    };
    if (m.lds_spec_possible) {
        if (build(qgs::Kernel::RkLds, 0) || build(qgs::Kernel::TendLds, 0)) return -1;
        for (int S : stages)
            if (S >= 3) {                                   // some 3+-stage scheme requested: also the general-tableau flavour
                if (build(qgs::Kernel::RkLdsDense, 0)) return -1;
                break;
            }
        if (!m.J.empty() && !m.spec_jac_possible && lds_tgl_bytes(&m) <= (size_t)QGS_LDS_STATE_BYTES)
            for (qgs::Kernel k : {qgs::Kernel::TglLds, qgs::Kernel::AdjLds})
                if (build(k, 0)) return -1;
    }
    if (!m.spec_possible || m.prefer_lds) return 0;       // prefer_lds: the register-resident kernels would only spill (and take minutes to compile)
    const bool jac_spec = !m.J.empty() && m.spec_jac_possible;
    auto list = qgs::kernel_list(m.ndim, jac_spec, stages, m.cg);
    if (!m.J.empty() && !jac_spec)                       // the trajectory pass of the tangent model is still specialised
        for (int S : stages) list.push_back({qgs::Kernel::RkStages, S});
    for (int S : stages)
        if (S >= 3) {                                                            // general lower-triangular tableaus
            list.push_back({qgs::Kernel::RkDense, S});
            if (jac_spec) list.push_back({qgs::Kernel::TglDense, S});
        }
    for (auto &ks : list)
        if (build(ks.first, ks.second)) return -1;
    return 0;
}

int qgs_prebuild_qr(int n_rows, int n_cols, const char *arch)
{
    if (n_rows < 1 || n_cols < 1 || n_cols > n_rows || n_rows > 300 || n_cols > 64) return fail("shape-specialised QR: 1 <= n_cols <= n_rows <= 300, n_cols <= 64");
    if (!prebuild_mine()) return 0;
    std::shared_ptr<const KernelBlob> blob;
    bool cached;
    const qgs::QrPlan plan = qr_plan_for(n_rows, n_cols);
    return obtain_blob("qgs_spec_qr_" + std::to_string(n_rows) + "x" + std::to_string(n_cols) + "|" + qgs::qr_plan_signature(plan),
                       (arch && *arch) ? arch : target_arch(-1), {}, [&] { return qgs::generate_qr_kernel(n_rows, n_cols, plan); }, &blob,
                       &cached, BlobMode::Publish);
}

int64_t qgs_qr_kernel_source(int n_rows, int n_cols, char *buf, int64_t buflen)
{
    if (n_rows < 1 || n_cols < 1 || n_cols > n_rows || n_rows > 300 || n_cols > 64) return fail("shape-specialised QR: 1 <= n_cols <= n_rows <= 300, n_cols <= 64");
    std::string src;
    try {
        const qgs::QrPlan plan = qr_plan_for(n_rows, n_cols);
        src = "// plan " + qgs::qr_plan_signature(plan) + "\n" + qgs::generate_qr_kernel(n_rows, n_cols, plan).source;
    } catch (const std::exception &e) {
        return fail(e.what());
    }
    if (buf && buflen > 0) {
        const size_t n = std::min<size_t>(src.size(), (size_t)buflen - 1);
        std::memcpy(buf, src.data(), n);
        buf[n] = 0;
    }
    return (int64_t)src.size();
}

int qgs_prebuild(int ndim, int64_t nnz, const int32_t *coo, const double *val, int64_t jnnz, const int32_t *jcoo,
                 const double *jval, int n_stage_counts, const int *stage_counts, const char *arch)
{
    return qgs_prebuild_rank(ndim, 3, nnz, coo, val, jnnz, jcoo, jval, n_stage_counts, stage_counts, arch);
}

}  // extern "C"
