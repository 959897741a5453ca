// Developer micro-benchmark (GPU box): where the time of the shape-specialised batched QR goes.
//   build host:  g++ -O1 -std=c++17 -I qgs_amd/csrc tools/ubench/qr_dump.cpp qgs_amd/csrc/codegen.cpp -o /tmp/qr_dump
//                /tmp/qr_dump 36 36 [members slots chains reload lookahead] > /tmp/q.hip
//                hipcc --offload-arch=gfx950 -O3 -DQGS_QR_PROFILE -DQR_SRC='"/tmp/q.hip"' -DQR_NAME=qgs_spec_qr_36x36 \
//                      -DQR_R=36 -DQR_C=36 -DQR_M=16 -DQR_W=5 tools/ubench/qr_phases.cpp -o tools/ubench/qr_phases_m16p2
//   GPU box:     tools/ubench/qr_phases_m16p2 [members, default 16384]
// The kernel is the generator's, compiled with -DQGS_QR_PROFILE: thread 0 of every workgroup notes the 100 MHz clock at entry,
// when its loads have arrived, after dgeqr2, after dorg2r, after issuing its stores and when they have been acknowledged.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include QR_SRC

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? std::atoll(argv[1]) : 16384, ld = (n + 63) / 64 * 64;
    const int R = QR_R, C = QR_C, M = QR_M, W = QR_W;
    const size_t na = (size_t)R * C * ld;
    std::vector<double> h(na);
    srand(1);
    for (auto &x : h) x = rand() / (double)RAND_MAX - 0.5;
    double *a, *a0, *rd;
    unsigned long long *prof;
    const long long tiles = (n + M - 1) / M;
    const unsigned grid = (unsigned)(M == 16 ? tiles : (tiles + 15) / 16 * 16);
    CHK(hipMalloc(&a, na * 8)); CHK(hipMalloc(&a0, na * 8)); CHK(hipMalloc(&rd, (size_t)C * ld * 8)); CHK(hipMalloc(&prof, (size_t)grid * 160 * 8));
    CHK(hipMemcpy(a0, h.data(), na * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHK(hipMemcpy(a, a0, na * 8, hipMemcpyDeviceToDevice));
        CHK(hipMemset(prof, 0, (size_t)grid * 160 * 8));
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(QR_NAME, dim3(grid), dim3(64 * W), 0, 0, a, rd, n, ld, prof);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> p((size_t)grid * 160);
    CHK(hipMemcpy(p.data(), prof, (size_t)grid * 160 * 8, hipMemcpyDeviceToHost));
    unsigned long long t_min = ~0ull, t_max = 0;
    for (unsigned b = 0; b < grid; ++b) if (p[b * 160]) { t_min = std::min(t_min, p[b * 160]); t_max = std::max(t_max, p[b * 160 + 5]); }
    double ph[5] = {0, 0, 0, 0, 0};
    long long cnt = 0;
    for (unsigned b = 0; b < grid; ++b) {
        if (!p[b * 160]) continue;
        ++cnt;
        for (int k = 0; k < 5; ++k) ph[k] += (double)(p[b * 160 + k + 1] - p[b * 160 + k]) * 0.01;      // 100 MHz -> us
    }
    std::printf("%lld x %dx%d, %u workgroups of %d wavefronts: kernel %.4f ms (events), first entry to last exit %.1f us\n", n, R, C, grid, W, best,
                (double)(t_max - t_min) * 0.01);
    std::printf("per workgroup (us, mean of %lld): loads %.2f | dgeqr2 %.2f | dorg2r %.2f | issue stores %.2f | stores acknowledged %.2f\n", cnt,
                ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt, ph[4] / cnt);
    // start times: how the workgroups of a CU follow each other
    std::vector<double> starts;
    for (unsigned b = 0; b < grid; ++b) if (p[b * 160]) starts.push_back((double)(p[b * 160] - t_min) * 0.01);
    std::sort(starts.begin(), starts.end());
    std::printf("workgroup entry times (us after the first): 10%% %.1f, 25%% %.1f, 50%% %.1f, 75%% %.1f, 90%% %.1f, last %.1f\n", starts[starts.size() / 10],
                starts[starts.size() / 4], starts[starts.size() / 2], starts[starts.size() * 3 / 4], starts[starts.size() * 9 / 10], starts.back());
    // shader cycles of every broadcast step (thread 0's s_memtime before the step's barrier), mean over the workgroups
    const int n_steps = 2 * (C - 1);
    std::printf("cycles per step (mean over workgroups), dgeqr2 j = 0 .. %d then dorg2r j = %d .. 0:\n", C - 2, C - 2);
    for (int st = 0; st < n_steps; ++st) {
        double sum = 0;
        long long k = 0;
        for (unsigned b = 0; b < grid; ++b) {
            if (!p[b * 160] || !p[b * 160 + 8 + st + 1] || !p[b * 160 + 8 + st]) continue;
            sum += (double)(p[b * 160 + 8 + st + 1] - p[b * 160 + 8 + st]);
            ++k;
        }
        std::printf("%s%.0f", st % 12 ? " " : (st ? "\n  " : "  "), k ? sum / k : 0.0);
    }
    std::printf("\n");
    return 0;
}
