// issue_ubench.hip -- developer micro-benchmark (not part of the product): how many cycles does a
// single wavefront need per fp64 FMA on gfx950, alone on its SIMD and with co-resident waves, with
// dependent / independent accumulators and with scalar-literal (s_mov) or s_load traffic interleaved?
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_ubench issue_ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define FMA(a) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y))
#define FMAS(a, s) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "s"(s), "v"(y))
#define SMOV2() asm volatile("s_mov_b32 %0, 0x12345678\n\ts_mov_b32 %1, 0x3ff12345" : "=s"(d0), "=s"(d1))
#define SMOV1() asm volatile("s_mov_b32 %0, 0x12345678" : "=s"(d0))

typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, long long *cyc, const double *ctab, int iters)
{
    double x = 1.0000001 + threadIdx.x * 1e-9, y = 0.9999999;
    double a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    int d0 = 0, d1 = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 8 independent chains, 32 FMAs
#pragma unroll
            for (int u = 0; u < 4; ++u) { FMA(a0); FMA(a1); FMA(a2); FMA(a3); FMA(a4); FMA(a5); FMA(a6); FMA(a7); }
        } else if (MODE == 1) {   // 1 dependent chain, 32 FMAs
#pragma unroll
            for (int u = 0; u < 32; ++u) { FMA(a0); }
        } else if (MODE == 2) {   // 2 chains
#pragma unroll
            for (int u = 0; u < 16; ++u) { FMA(a0); FMA(a1); }
        } else if (MODE == 3) {   // 4 chains
#pragma unroll
            for (int u = 0; u < 8; ++u) { FMA(a0); FMA(a1); FMA(a2); FMA(a3); }
        } else if (MODE == 4) {   // 8 chains + 2 s_mov per FMA
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                SMOV2(); FMA(a0); SMOV2(); FMA(a1); SMOV2(); FMA(a2); SMOV2(); FMA(a3);
                SMOV2(); FMA(a4); SMOV2(); FMA(a5); SMOV2(); FMA(a6); SMOV2(); FMA(a7);
            }
        } else if (MODE == 5) {   // 8 chains + 1 s_mov per FMA
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                SMOV1(); FMA(a0); SMOV1(); FMA(a1); SMOV1(); FMA(a2); SMOV1(); FMA(a3);
                SMOV1(); FMA(a4); SMOV1(); FMA(a5); SMOV1(); FMA(a6); SMOV1(); FMA(a7);
            }
        } else if (MODE == 6) {   // 8 chains, coefficients from one s_load_dwordx16 per 8 FMAs (waited immediately)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v16i c;
                asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c) : "s"(ctab));
                double c0 = __builtin_bit_cast(double, __builtin_shufflevector(c, c, 0, 1));
                double c1 = __builtin_bit_cast(double, __builtin_shufflevector(c, c, 2, 3));
                FMAS(a0, c0); FMAS(a1, c1); FMAS(a2, c0); FMAS(a3, c1); FMAS(a4, c0); FMAS(a5, c1); FMAS(a6, c0); FMAS(a7, c1);
            }
        } else if (MODE == 7) {   // as 6 but the load is issued one group ahead (double buffered)
            v16i c, n;
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(c) : "s"(ctab));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 %0, %1, 0x40" : "=s"(n) : "s"(ctab), "s"(c));
                double c0 = __builtin_bit_cast(double, __builtin_shufflevector(c, c, 0, 1));
                double c1 = __builtin_bit_cast(double, __builtin_shufflevector(c, c, 2, 3));
                FMAS(a0, c0); FMAS(a1, c1); FMAS(a2, c0); FMAS(a3, c1); FMAS(a4, c0); FMAS(a5, c1); FMAS(a6, c0); FMAS(a7, c1);
                c = n;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(c));
        } else if (MODE == 8) {   // 8 chains of v_mul/v_fma mix with a v_accvgpr round trip per FMA (AGPR parking cost)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int t;
                asm volatile("v_accvgpr_write_b32 a0, %1\n\tv_accvgpr_read_b32 %0, a0" : "=v"(t) : "v"(d0) : "a0");
                FMA(a0); FMA(a1); FMA(a2); FMA(a3);
                asm volatile("v_accvgpr_write_b32 a1, %1\n\tv_accvgpr_read_b32 %0, a1" : "=v"(t) : "v"(d0) : "a1");
                FMA(a4); FMA(a5); FMA(a6); FMA(a7);
            }
        } else if (MODE == 9) {   // 8 chains + one ds_read_b64 (broadcast address) per FMA
            __shared__ double lds[64];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double t;
                asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(0));
                FMA(a0); FMA(a1); FMA(a2); FMA(a3); FMA(a4); FMA(a5); FMA(a6); FMA(a7);
                asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(t));
            }
            (void)lds;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + d0 + d1;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks, int iters, double *out, long long *cyc, const double *ctab)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out, cyc, ctab, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, cyc, ctab, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double fmas = (double)iters * 32;
    printf("%-34s blocks %5d  cyc/FMA median %6.2f  min %6.2f  max %6.2f | wall %.3f ms -> %.2f GHz-equiv, chip DP TFLOP/s %.1f\n",
           name, blocks, h[blocks / 2] / fmas, h[0] / fmas, h[blocks - 1] / fmas, ms,
           h[blocks / 2] / (ms * 1e6), 2.0 * 64 * fmas * blocks / (ms * 1e-3) / 1e12);
}

int main()
{
    double *out; long long *cyc; double *ctab;
    hipMalloc(&out, 8 * 64 * 8192); hipMalloc(&cyc, 8 * 8192); hipMalloc(&ctab, 4096);
    hipMemset(ctab, 0, 4096);
    const int iters = 2000;
    for (int blocks : {1024, 2048, 4096, 8192}) {
        run<0>("8 indep chains", blocks, iters, out, cyc, ctab);
        run<1>("1 dependent chain", blocks, iters, out, cyc, ctab);
        run<2>("2 chains", blocks, iters, out, cyc, ctab);
        run<3>("4 chains", blocks, iters, out, cyc, ctab);
        run<4>("8 chains + 2 s_mov/FMA", blocks, iters, out, cyc, ctab);
        run<5>("8 chains + 1 s_mov/FMA", blocks, iters, out, cyc, ctab);
        run<6>("8 chains + s_load x16 per 8 (sync)", blocks, iters, out, cyc, ctab);
        run<7>("8 chains + s_load x16 per 8 (ahead)", blocks, iters, out, cyc, ctab);
        run<8>("8 chains + accvgpr rt per 4", blocks, iters, out, cyc, ctab);
        run<9>("8 chains + ds_read_b64 per 8", blocks, iters, out, cyc, ctab);
    }
    return 0;
}
