// Micro-benchmark: does LDS read traffic take fp64 issue slots away from a full CU?  16 wavefronts per workgroup (4 per SIMD,
// as in qgs_spec_rklds16), every iteration 64 v_fma_f64 per wavefront on 8 independent chains plus NLDS ds_read_b64 whose
// values are folded into the chains one iteration later.  Reports cycles per iteration against the fp64 floor (4 wavefronts x
// 64 x 4 = 1024 cycles per SIMD) and the LDS floor (16 x NLDS x 4 cycles per CU).
//   hipcc --offload-arch=gfx950 -O3 -o lds_valu lds_valu.hip && ./lds_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define F8(c) a0 = __builtin_fma(c, p, a0); a1 = __builtin_fma(c, p, a1); a2 = __builtin_fma(c, p, a2); a3 = __builtin_fma(c, p, a3); \
              a4 = __builtin_fma(c, p, a4); a5 = __builtin_fma(c, p, a5); a6 = __builtin_fma(c, p, a6); a7 = __builtin_fma(c, p, a7);
template <int NLDS>
__global__ void __launch_bounds__(1024) k(double *out, int iters, double c)
{
    __shared__ double xs[228][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < 228; r += 16) xs[r][lane] = 1e-3 * r + lane;
    __syncthreads();
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, p = 1.0 + lane * 1e-3;
    asm volatile("" : "+s"(c));
    double t[NLDS > 0 ? NLDS : 1];
    for (int q = 0; q < NLDS; ++q) t[q] = 0.0;
    unsigned row = wave;
    for (int it = 0; it < iters; ++it) {
        // fold the values requested one iteration ago, then request the next ones (they fly under the 64 FMAs)
        for (int q = 0; q < NLDS; ++q) p += t[q];
        unsigned base = (row % 200) * 512u + lane * 8u;
        asm volatile("" : "+v"(base));
#pragma unroll
        for (int q = 0; q < NLDS; ++q) t[q] = *(const double *)((const char *)xs + base + q * 512);
        __builtin_amdgcn_sched_barrier(0);
        F8(c) F8(c) F8(c) F8(c) F8(c) F8(c) F8(c) F8(c)
        __builtin_amdgcn_sched_barrier(0);
        row += 3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p;
}
template <int NLDS>
void run(double *out, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NLDS><<<256, 1024>>>(out, 10, 1e-9);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<NLDS><<<256, 1024>>>(out, iters, 1e-9);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ns_iter = best * 1e6 / iters;
    std::printf("NLDS %2d per 64 FMAs: %.3f ms, %.1f ns per iteration = %.0f cycles at 2.0 GHz (fp64 floor 1024 + %d fold adds x 16, LDS floor %d)\n",
                NLDS, best, ns_iter, ns_iter * 2.0, NLDS, 64 * NLDS);
}
int main()
{
    double *out;
    hipMalloc(&out, 256 * 1024 * sizeof(double));
    const int iters = 20000;
    run<0>(out, iters); run<4>(out, iters); run<8>(out, iters); run<12>(out, iters); run<16>(out, iters); run<24>(out, iters);
    hipFree(out);
    return 0;
}
