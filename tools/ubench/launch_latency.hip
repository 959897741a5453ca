// launch_latency -- what one tiny kernel costs end to end on the host, by completion-wait strategy.
// Build: hipcc --offload-arch=gfx950 -O2 -o launch_latency launch_latency.hip ; run on the GPU box.
// The single-state f(t, x) / Df(t, x) of libqgs_hip (qgs_tendencies with n_traj == 1) is one such launch: 36 loads from a
// page-locked block, ~500 FMAs, 36 stores into the block.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

__global__ void work(const double *x, double *y, int n, volatile unsigned long long *flag, unsigned long long seq)
{
    if (threadIdx.x < n) y[threadIdx.x] = x[threadIdx.x] * 1.5 + 1.0;
    if (flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) *flag = seq;
    }
}

#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { std::printf("%s: %s\n", #e, hipGetErrorString(r)); return 1; } } while (0)

int main()
{
    double *h, *d;
    CK(hipHostMalloc((void **)&h, 4096, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&d, h, 0));
    std::memset(h, 0, 4096);
    volatile unsigned long long *hflag = (volatile unsigned long long *)(h + 256);
    unsigned long long *dflag = (unsigned long long *)(d + 256);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int N = 5000;
    auto bench = [&](const char *name, int mode) {
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 1; i <= N; ++i) {
                const unsigned long long seq = (unsigned long long)(rep * N + i) + (unsigned long long)mode * 1000000ull;
                hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, st, d, d + 64, 36, mode == 2 ? dflag : nullptr, seq);
                if (mode == 0) (void)hipStreamSynchronize(st);
                else if (mode == 1) while (hipStreamQuery(st) == hipErrorNotReady) {}
                else while (*hflag != seq) {}
            }
            (void)hipStreamSynchronize(st);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep == 2) std::printf("%-44s %7.2f us per launch + wait\n", name, us);
        }
    };
    bench("hipStreamSynchronize", 0);
    bench("spin on hipStreamQuery", 1);
    bench("spin on a flag the kernel writes (host memory)", 2);
    unsigned flags = 0;
    (void)hipGetDeviceFlags(&flags);
    std::printf("device flags 0x%x\n", flags);
    return 0;
}
