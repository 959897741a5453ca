// qgs_kcompile -- out-of-process hiprtc front end of libqgs_hip.so.
//
// The specialised kernels are compiled at run time; which compiler does it decides their quality (the fused stepper:
// 282 VGPRs from the system ROCm's hiprtc / comgr, 324 from the older pair bundled with PyTorch, which a process that
// imported torch has mapped first under the same sonames).  The library therefore compiles through this helper, a fresh
// process linked against /opt/rocm/lib's hiprtc (RPATH): the same code object results on the GPU-less build host, inside a
// torch process on the GPU box and under a profiler.
//
//   qgs_kcompile --version                          -> identity of the compiler: hiprtc API version + the library file it resolved to
//   qgs_kcompile <arch> <source file> <output file> [extra compiler flags ...]
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <limits.h>
#include <stdlib.h>

#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

int main(int argc, char **argv)
{
    if (argc == 2 && std::string(argv[1]) == "--version") {
        int major = 0, minor = 0;
        if (hiprtcVersion(&major, &minor) != HIPRTC_SUCCESS) return 1;
        // hiprtcVersion is the API level (9.0 for ROCm 7.0 and 7.2 alike): the library file name carries the release
        std::string file = "?";
        Dl_info info;
        if (dladdr((void *)&hiprtcVersion, &info) && info.dli_fname) {
            char real[PATH_MAX];
            file = realpath(info.dli_fname, real) ? real : info.dli_fname;
            const size_t k = file.find_last_of('/');
            if (k != std::string::npos) file = file.substr(k + 1);
        }
        std::printf("hiprtc%d.%d-%s\n", major, minor, file.c_str());
        return 0;
    }
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s --version | <arch> <source> <output> [flags...]\n", argv[0]);
        return 2;
    }
    std::ifstream in(argv[2], std::ios::binary);
    if (!in) { std::fprintf(stderr, "qgs_kcompile: cannot read %s\n", argv[2]); return 1; }
    const std::string src((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "qgs_spec.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        std::fprintf(stderr, "qgs_kcompile: hiprtcCreateProgram failed\n");
        return 1;
    }
    const std::string archopt = std::string("--offload-arch=") + argv[1];
    std::vector<const char *> opts = {archopt.c_str(), "-O3", "-std=c++17"};
    for (int a = 4; a < argc; ++a) opts.push_back(argv[a]);
    const hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        std::fprintf(stderr, "qgs_kcompile: %s\n%s\n", hiprtcGetErrorString(r), log.substr(0, 4000).c_str());
        return 1;
    }
    size_t n = 0;
    hiprtcGetCodeSize(prog, &n);
    std::vector<char> code(n);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    std::ofstream out(argv[3], std::ios::binary);
    out.write(code.data(), (std::streamsize)code.size());
    out.close();
    return out ? 0 : 1;
}
