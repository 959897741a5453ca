// mfma_f64_ubench.hip -- developer micro-benchmark: what is the fp64 MFMA rate of gfx950 next to the fp64
// VALU rate?  (Decides whether a dense-tile formulation of the tendencies can ever beat the sparse VALU
// kernels: it cannot if the two peaks are equal, because the tensor blocks are <5 % dense.)
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_ubench mfma_f64_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_mfma(double *out, int iters)
{
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void __launch_bounds__(256) k_valu(double *out, int iters)
{
    double x = 1.0000001 + threadIdx.x * 1e-9, y = 0.9999999;
    double a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_fma(x, y, a0); a1 = __builtin_fma(x, y, a1); a2 = __builtin_fma(x, y, a2); a3 = __builtin_fma(x, y, a3);
        a4 = __builtin_fma(x, y, a4); a5 = __builtin_fma(x, y, a5); a6 = __builtin_fma(x, y, a6); a7 = __builtin_fma(x, y, a7);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
    double *out;
    hipMalloc(&out, 8 * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 2048;     // 8 waves per SIMD
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0); k_mfma<<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        double flops = 2.0 * 16 * 16 * 4 * 4.0 * iters * (blocks * 4.0);      // per wave: 4 MFMAs of 2*16*16*4
        printf("v_mfma_f64_16x16x4_f64 : %.3f ms  %.1f TFLOP/s\n", ms, flops / (ms * 1e-3) / 1e12);
        hipEventRecord(e0); k_valu<<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        flops = 2.0 * 8 * iters * (blocks * 256.0);
        printf("v_fma_f64 (8 chains)   : %.3f ms  %.1f TFLOP/s\n", ms, flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
