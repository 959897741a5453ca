// Micro-benchmark: rate of v_fmac_f64_dpp (coefficient broadcast from a lane of a VGPR pair, row_newbcast) against
// v_fma_f64 with the coefficient in an SGPR pair, and the meaning of row_newbcast:n (lane n of every row of 16).
//   hipcc --offload-arch=gfx950 -O3 -o dpp_fmac dpp_fmac.hip && ./dpp_fmac
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__constant__ double ctab[16];
#define FM(acc, n) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(cv), "v"(p))
__global__ void k_dpp(double *out, const double *tab, int iters)
{
    const int lane = threadIdx.x & 63;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    double p = 1.0 + lane * 1e-3;
    double cv = tab[lane & 15];
    for (int it = 0; it < iters; ++it) {
        FM(a0, 0); FM(a1, 1); FM(a2, 2); FM(a3, 3); FM(a4, 4); FM(a5, 5); FM(a6, 6); FM(a7, 7);
        FM(a0, 8); FM(a1, 9); FM(a2, 10); FM(a3, 11); FM(a4, 12); FM(a5, 13); FM(a6, 14); FM(a7, 15);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void k_sgpr(double *out, const double *tab, int iters)
{
    const int lane = threadIdx.x & 63;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    double p = 1.0 + lane * 1e-3;
    double c[16];
    for (int q = 0; q < 16; ++q) { c[q] = ctab[q]; asm volatile("" : "+s"(c[q])); }
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(p));
        a0 = __builtin_fma(c[0], p, a0); a1 = __builtin_fma(c[1], p, a1); a2 = __builtin_fma(c[2], p, a2); a3 = __builtin_fma(c[3], p, a3);
        a4 = __builtin_fma(c[4], p, a4); a5 = __builtin_fma(c[5], p, a5); a6 = __builtin_fma(c[6], p, a6); a7 = __builtin_fma(c[7], p, a7);
        a0 = __builtin_fma(c[8], p, a0); a1 = __builtin_fma(c[9], p, a1); a2 = __builtin_fma(c[10], p, a2); a3 = __builtin_fma(c[11], p, a3);
        a4 = __builtin_fma(c[12], p, a4); a5 = __builtin_fma(c[13], p, a5); a6 = __builtin_fma(c[14], p, a6); a7 = __builtin_fma(c[15], p, a7);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// Hazard check: the fmac's plain operands (product, accumulator) written by the instruction right before it.  (The ISA manual
// asks for two wait states between a VALU write and a DPP read of the same VGPR; the compiler does not see inside inline
// asm.  The DPP operand here always comes from a memory load, the other two operands are ordinary interlocked reads.)
__global__ void k_hazard(double *out, const double *tab, int iters, int use_dpp)
{
    const int lane = threadIdx.x & 63;
    double a = 0.0, p = 1.0 + lane * 1e-3;
    double cv = tab[lane & 15];
    const double c5 = ctab[5], c9 = ctab[9];
    if (use_dpp) {
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_mul_f64 %1, %1, %3\n\tv_fmac_f64_dpp %0, %2, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f64 %1, %1, %0\n\tv_fmac_f64_dpp %0, -%2, %1 row_newbcast:9 row_mask:0xf bank_mask:0xf"
                         : "+v"(a), "+v"(p) : "v"(cv), "v"(0.999));
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            p = p * 0.999; a = __builtin_fma(c5, p, a); p = p + a; a = __builtin_fma(-c9, p, a);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main()
{
    const int blocks = 256 * 8, threads = 256, iters = 20000;
    double *out, *tab;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&tab, sizeof(double) * 16);
    std::vector<double> h(16);
    for (int q = 0; q < 16; ++q) h[q] = (q + 1) * 1e-6;
    hipMemcpy(tab, h.data(), 128, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(ctab), h.data(), 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> r1(64), r2(64);
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which == 0) k_dpp<<<blocks, threads>>>(out, tab, iters); else k_sgpr<<<blocks, threads>>>(out, tab, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 16 * iters * (double)blocks * threads;
        printf("%s: %.3f ms  %.1f TFLOP/s\n", which == 0 ? "v_fmac_f64_dpp row_newbcast" : "v_fma_f64 sgpr coefficient ", ms, flop / ms * 1e-9);
        hipMemcpy(which == 0 ? r1.data() : r2.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    }
    double d = 0;
    for (int l = 0; l < 64; ++l) d = fmax(d, fabs(r1[l] - r2[l]) / fabs(r2[l]));
    printf("max relative difference between the two: %.3e\n", d);
    for (int which = 0; which < 2; ++which) {
        k_hazard<<<64, 256>>>(out, tab, 1000, which);
        hipMemcpy(which == 0 ? r1.data() : r2.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    }
    double dh = 0;
    for (int l = 0; l < 64; ++l) dh = fmax(dh, fabs(r1[l] - r2[l]) / fabs(r2[l]));
    printf("back-to-back dependent operands, max relative difference: %.3e (value %.6e)\n", dh, r2[7]);
    return (d < 1e-14 && dh < 1e-14) ? 0 : 1;
}
