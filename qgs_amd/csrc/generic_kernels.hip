// generic_kernels.hip -- see generic_kernels.h.  gfx950 only.
#include "generic_kernels.h"

#include <algorithm>
#include <mutex>

namespace qgs {

namespace {

constexpr int WAVE = 64;

// Largest dynamic-LDS size a kernel has been configured for (hipFuncAttributeMaxDynamicSharedMemorySize), per device and
// guarded: models of one process live on several GPUs and are driven by one host thread each (qgs_group, Lyapunov shards).
struct DynLdsLimit {
    static constexpr int MAX_DEVICES = 64;
    size_t configured[MAX_DEVICES];
    std::mutex mutex;
    explicit DynLdsLimit(size_t initial = 0) { for (size_t &c : configured) c = initial; }
    static int device()
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
        return d;
    }
    // beyond MAX_DEVICES devices nothing is remembered: the attribute is set on every launch that needs more than the default
    bool needs(size_t lds)
    {
        const int d = device();
        if (d >= MAX_DEVICES) return lds > 64 * 1024;
        std::lock_guard<std::mutex> lock(mutex);
        return lds > configured[d];
    }
    void set(size_t lds)
    {
        const int d = device();
        if (d >= MAX_DEVICES) return;
        std::lock_guard<std::mutex> lock(mutex);
        if (lds > configured[d]) configured[d] = lds;
    }
};

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// One tendency row: r = sum_e val_e * x_a * x_b  with x_0 = 1 (sparse_mul.py:76-80; the row loop
// is the COO loop restricted to coo[n,0] == i, entries kept in the reference's (j,k) order).
__device__ __forceinline__ double row_dot2(const DevTensor &T, int i, const double *__restrict__ x, int64_t ld, int64_t m)
{
    double r = 0.0;
    const int e1 = T.rowptr[i + 1];
    for (int e = T.rowptr[i]; e < e1; ++e) {
        const uint32_t jk = T.idx[e];
        const uint32_t j = jk >> 16, k = jk & 0xffffu;
        const double xj = j ? x[(int64_t)(j - 1) * ld + m] : 1.0;
        const double xk = k ? x[(int64_t)(k - 1) * ld + m] : 1.0;
        double prod = xj * xk;
        if (T.idx2) {                                               // rank 5 (sparse_mul.py:155-157): two more factors
            const uint32_t cd = T.idx2[e];
            const uint32_t c = cd >> 16, d = cd & 0xffffu;
            if (c) prod *= x[(int64_t)(c - 1) * ld + m];
            if (d) prod *= x[(int64_t)(d - 1) * ld + m];
        }
        r = __builtin_fma(prod, T.val[e], r);
    }
    return r;
}

__global__ void __launch_bounds__(WAVE) gen_tend_kernel(DevTensor T, int ndim, int64_t n_traj, int64_t ld,
                                                        const double *__restrict__ x, double *__restrict__ dx)
{
    const int64_t m = (int64_t)blockIdx.x * WAVE + threadIdx.x;
    if (m >= n_traj) return;
    for (int i = 1; i <= ndim; ++i) dx[(int64_t)(i - 1) * ld + m] = row_dot2(T, i, x, ld, m);
}

// Jacobian rows: entries of row i carry idx = (j << 16) | k and contribute val * x_k to J[i][j]
// (sparse_mul.py:40-45).
__global__ void __launch_bounds__(WAVE) gen_jac_kernel(DevTensor Jt, int ndim, int64_t n_traj, int64_t ld,
                                                       const double *__restrict__ x, double *__restrict__ jm)
{
    const int64_t m = (int64_t)blockIdx.x * WAVE + threadIdx.x;
    if (m >= n_traj) return;
    for (int i = 1; i <= ndim; ++i) {
        const int e1 = Jt.rowptr[i + 1];
        for (int e = Jt.rowptr[i]; e < e1; ++e) {
            const uint32_t jk = Jt.idx[e];
            const uint32_t j = jk >> 16, k = jk & 0xffffu;
            if (j == 0) continue;                                   // column 0 is dropped (tendencies.py:121)
            double xk = k ? x[(int64_t)(k - 1) * ld + m] : 1.0;
            if (Jt.idx2) {                                          // rank 5 (sparse_mul4, sparse_mul.py:117-119)
                const uint32_t cd = Jt.idx2[e];
                const uint32_t c = cd >> 16, d = cd & 0xffffu;
                if (c) xk *= x[(int64_t)(c - 1) * ld + m];
                if (d) xk *= x[(int64_t)(d - 1) * ld + m];
            }
            double *p = jm + ((int64_t)(i - 1) * ndim + (j - 1)) * ld + m;
            *p = __builtin_fma(xk, Jt.val[e], *p);
        }
    }
}

// ---- one state: f(x) and Df(x) for a single (ndim,) vector (the callables handed to SciPy / DiffEq solvers) -----------
// x and the results live in a page-locked host block the kernel addresses directly.  The host does not call
// hipStreamSynchronize (12 us per launch + wait on this stack): the workgroup that finishes last writes the call's sequence
// number into the block and the host spins on it (7.8 us; tools/ubench/launch_latency.hip).
__device__ __forceinline__ void one_state_done(unsigned *counter, volatile unsigned long long *flag, unsigned long long seq)
{
    __threadfence_system();                                 // this workgroup's results are visible to the host ...
    __syncthreads();
    if (threadIdx.x == 0) {
        if (gridDim.x == 1) { *flag = seq; return; }        // small systems run in one workgroup: nobody else to wait for
        const unsigned done = atomicAdd(counter, 1u);
        if (done == gridDim.x - 1) {                        // ... and so are everybody else's: they fenced before they counted
            *counter = 0;
            __threadfence_system();
            *flag = seq;
        }
    }
}

// f: one wavefront per tensor row, lanes stride over the row's entries (coalesced), butterfly sum
__global__ void __launch_bounds__(256) gen_tend_one_kernel(DevTensor T, int ndim, const double *__restrict__ x, double *__restrict__ dx,
                                                           unsigned *counter, volatile unsigned long long *flag, unsigned long long seq)
{
    extern __shared__ double xs[];                         // slot 0 = 1 (the constant), slot d = x_d
    for (int d = threadIdx.x; d <= ndim; d += blockDim.x) xs[d] = d ? x[d - 1] : 1.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    for (int i = 1 + blockIdx.x * nwv + wv; i <= ndim; i += gridDim.x * nwv) {
        double r = 0.0;
        const int e1 = T.rowptr[i + 1];
        for (int e = T.rowptr[i] + lane; e < e1; e += 64) {
            const uint32_t jk = T.idx[e];
            double t = xs[jk >> 16] * xs[jk & 0xffffu];
            if (T.idx2) { const uint32_t cd = T.idx2[e]; t *= xs[cd >> 16] * xs[cd & 0xffffu]; }
            r = __builtin_fma(t, T.val[e], r);
        }
        for (int off = 32; off > 0; off >>= 1) r += __shfl_xor(r, off);
        if (lane == 0) dx[i - 1] = r;
    }
    one_state_done(counter, flag, seq);
}

// Df: one thread per element of the (ndim, ndim) output; lut[i * ndim + j] = index of the (i, j) pair in the pair-grouped
// Jacobian tensor (OnePairs) or -1 where the Jacobian is structurally zero.  Every element is written: no zero-fill.
__global__ void __launch_bounds__(256) gen_jac_one_kernel(OnePairs P, int ndim, const double *__restrict__ x, double *__restrict__ jm,
                                                          unsigned *counter, volatile unsigned long long *flag, unsigned long long seq)
{
    extern __shared__ double xs[];
    for (int d = threadIdx.x; d <= ndim; d += blockDim.x) xs[d] = d ? x[d - 1] : 1.0;
    __syncthreads();
    const int total = ndim * ndim;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        double r = 0.0;
        const int p = P.lut[t];
        if (p >= 0) {
            const int e1 = P.ptr[p + 1];
            for (int e = P.ptr[p]; e < e1; ++e) {
                const uint32_t w = P.idx[e];
                double xk = xs[w & 0xffffu];
                if (P.idx2) { const uint32_t cd = P.idx2[e]; xk *= xs[cd >> 16] * xs[cd & 0xffffu]; }
                r = __builtin_fma(xk, P.val[e], r);
            }
        }
        jm[t] = r;
    }
    one_state_done(counter, flag, seq);
}

__device__ __forceinline__ int64_t rec_index(int64_t iw, int64_t n_records, int backward)
{
    return backward ? (n_records - 1 - iw) : iw;
}

__global__ void __launch_bounds__(WAVE) gen_rk_kernel(DevTensor T, RkArgs p, const double *__restrict__ y_in,
                                                      double *__restrict__ y_out, double *__restrict__ rec,
                                                      double *__restrict__ stages, double *__restrict__ work,
                                                      const double *__restrict__ dtime, const double *__restrict__ tab)
{
    const int64_t m = (int64_t)blockIdx.x * WAVE + threadIdx.x;
    if (m >= p.n_traj) return;
    const int ndim = p.ndim, s = p.s;
    const int64_t ld = p.ld, A = (int64_t)ndim * ld;
    double *y = work, *ys = work + A, *k = work + 2 * A;
    const double *b = tab, *a = tab + s;
    for (int d = 0; d < ndim; ++d) y[d * ld + m] = y_in[d * ld + m];
    for (int64_t ti = p.step_begin; ti < p.step_end; ++ti) {
        const double dt = dtime[ti + 1] - dtime[ti];
        if (p.write_steps > 0 && (ti % p.write_steps) == 0) {
            double *r = rec + rec_index(ti / p.write_steps, p.n_records, p.backward) * A + m;
            for (int d = 0; d < ndim; ++d) r[d * ld] = y[d * ld + m];
        }
        for (int i = 0; i < s; ++i) {
            for (int d = 0; d < ndim; ++d) {                              // y_s = y + (dt*a[i]) @ k
                double acc = 0.0;
                for (int j = 0; j < i; ++j) acc = __builtin_fma(dt * a[i * s + j], k[j * A + d * ld + m], acc);
                ys[d * ld + m] = y[d * ld + m] + acc;
            }
            if (stages) {
                double *sp = stages + ((ti - p.step_begin) * s + i) * A + m;
                for (int d = 0; d < ndim; ++d) sp[d * ld] = ys[d * ld + m];
            }
            double *ki = k + i * A;
            for (int r = 1; r <= ndim; ++r) ki[(int64_t)(r - 1) * ld + m] = row_dot2(T, r, ys, ld, m);
        }
        for (int d = 0; d < ndim; ++d) {                                  // y = y + (dt*b) @ k
            double acc = 0.0;
            for (int j = 0; j < s; ++j) acc = __builtin_fma(dt * b[j], k[j * A + d * ld + m], acc);
            y[d * ld + m] += acc;
        }
    }
    if (y_out) for (int d = 0; d < ndim; ++d) y_out[d * ld + m] = y[d * ld + m];
    if (p.write_final) {
        double *r = rec + rec_index(p.n_records - 1, p.n_records, p.backward) * A + m;
        for (int d = 0; d < ndim; ++d) r[d * ld] = y[d * ld + m];
    }
}

__global__ void __launch_bounds__(WAVE) gen_tgl_kernel(DevTensor Jr, RkArgs p, int64_t n_tg, double inverse,
                                                       const double *__restrict__ w_in, double *__restrict__ w_out,
                                                       double *__restrict__ rec, const double *__restrict__ stages,
                                                       double *__restrict__ work, const double *__restrict__ dtime,
                                                       const double *__restrict__ tab)
{
    const int64_t L = n_tg * p.ld;
    const int64_t l = (int64_t)blockIdx.x * WAVE + threadIdx.x;
    if (l >= L) return;
    const int64_t m = l % p.ld;
    if (m >= p.n_traj) return;
    const int ndim = p.ndim, s = p.s;
    const int64_t ld = p.ld, A = (int64_t)ndim * L, AS = (int64_t)ndim * ld;
    double *v = work, *ws = work + A, *km = work + 2 * A;
    const double *b = tab, *a = tab + s;
    for (int d = 0; d < ndim; ++d) v[d * L + l] = w_in[d * L + l];
    for (int64_t ti = p.step_begin; ti < p.step_end; ++ti) {
        const double dt = dtime[ti + 1] - dtime[ti];
        if (p.write_steps > 0 && (ti % p.write_steps) == 0) {
            double *r = rec + rec_index(ti / p.write_steps, p.n_records, p.backward) * A + l;
            for (int d = 0; d < ndim; ++d) r[d * L] = v[d * L + l];
        }
        for (int i = 0; i < s; ++i) {
            for (int d = 0; d < ndim; ++d) {                              // km_s = fm + sum_j dt*a[i,j]*km[j]
                double acc = v[d * L + l];
                for (int j = 0; j < i; ++j) acc = __builtin_fma(dt * a[i * s + j], km[j * A + d * L + l], acc);
                ws[d * L + l] = acc;
            }
            const double *x = stages + ((ti - p.step_begin) * s + i) * AS;
            double *kmi = km + i * A;
            for (int r = 1; r <= ndim; ++r) {                             // hom = inverse * (J or J^T) @ km_s
                double acc = 0.0;
                const int e1 = Jr.rowptr[r + 1];
                for (int e = Jr.rowptr[r]; e < e1; ++e) {
                    const uint32_t wx = Jr.idx[e];
                    const uint32_t wi = wx >> 16, xi = wx & 0xffffu;
                    double xv = xi ? x[(int64_t)(xi - 1) * ld + m] : 1.0;
                    if (Jr.idx2) {
                        const uint32_t cd = Jr.idx2[e];
                        const uint32_t c = cd >> 16, d = cd & 0xffffu;
                        if (c) xv *= x[(int64_t)(c - 1) * ld + m];
                        if (d) xv *= x[(int64_t)(d - 1) * ld + m];
                    }
                    acc = __builtin_fma(xv * ws[(int64_t)(wi - 1) * L + l], Jr.val[e], acc);
                }
                kmi[(int64_t)(r - 1) * L + l] = inverse * acc;
            }
        }
        for (int d = 0; d < ndim; ++d) {                                  // fm += sum_j dt*b[j]*km[j]
            double acc = v[d * L + l];
            for (int j = 0; j < s; ++j) acc = __builtin_fma(dt * b[j], km[j * A + d * L + l], acc);
            v[d * L + l] = acc;
        }
    }
    if (w_out) for (int d = 0; d < ndim; ++d) w_out[d * L + l] = v[d * L + l];
    if (p.write_final) {
        double *r = rec + rec_index(p.n_records - 1, p.n_records, p.backward) * A + l;
        for (int d = 0; d < ndim; ++d) r[d * L] = v[d * L + l];
    }
}

// ---- tiled generic stepper ----------------------------------------------------------------------------
// RPW rows per wavefront (static register slots), NW = 16 wavefronts per workgroup, 64 members per
// workgroup.  The tensor entries of a row are wave-uniform (scalar loads); x_j / x_k are conflict-free
// 512-byte LDS reads (lane-consecutive doubles).
constexpr int TILED_NW = 16;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const int32_t ci32;
typedef __attribute__((address_space(4))) const uint32_t cu32;
typedef __attribute__((address_space(4))) const double cf64;
typedef __attribute__((address_space(4))) const u32x4 cu32x4;
typedef __attribute__((address_space(4))) const f64x4 cf64x4;

template <int RPW, int TPI>
__global__ void __launch_bounds__(64 * TILED_NW) gen_rk_tiled_kernel(TiledTensor T, RkArgs p, const double *__restrict__ y_in,
                                                                      double *__restrict__ y_out, double *__restrict__ rec,
                                                                      double *__restrict__ stages,
                                                                      const double *__restrict__ dtime,
                                                                      const double *__restrict__ tab)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t m0 = (int64_t)blockIdx.x * WAVE + lane;
    const bool live = m0 < p.n_traj;
    const int64_t m = live ? m0 : p.n_traj - 1;
    const int ndim = p.ndim, s = p.s;
    const int64_t ld = p.ld, A = (int64_t)ndim * ld;
    // own rows: slot r of this wavefront evaluates tensor row row_map[wave*RPW + r] (0 = empty slot); the map
    // balances the number of terms per wavefront (rows have 16-237 terms at MAOOAM 6x6)
    const ci32 *row_map = (const ci32 *)T.row_map + wave * RPW;
    const uint32_t lane8 = (uint32_t)lane * 8u;
    char *xs = smem;                                       // slot r at byte r*512, lane l at +l*8
    // the tensor stream is read-only for the whole launch: constant address space => scalar (s_load) fetches
    const ci32 *row_term = (const ci32 *)T.row_term;
    const cu32 *term_joff = (const cu32 *)T.term_joff;
    const cu32 *term_koff = (const cu32 *)T.term_koff;
    const cf64 *term_c = (const cf64 *)T.term_c;
    int rows[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) rows[r] = row_map[r];

    double y[RPW], acc[RPW], xn[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = rows[r];
        y[r] = row ? y_in[(int64_t)(row - 1) * ld + m] : 0.0;
        if (row) *(double *)(xs + (uint32_t)row * 512u + lane8) = y[r];
    }
    if (wave == 0) *(double *)(xs + lane8) = 1.0;          // eta_0 = 1
    __syncthreads();

    int64_t iw = 0, next_rec = -1;
    if (p.write_steps > 0) { iw = (p.step_begin + p.write_steps - 1) / p.write_steps; next_rec = iw * p.write_steps; }

    for (int64_t ti = p.step_begin; ti < p.step_end; ++ti) {
        const double dt = dtime[ti + 1] - dtime[ti];
        if (ti == next_rec) {
            double *q = rec + rec_index(iw, p.n_records, p.backward) * A + m;
            ++iw; next_rec += p.write_steps;
            if (live) {
#pragma unroll
                for (int r = 0; r < RPW; ++r) if (rows[r]) q[(int64_t)(rows[r] - 1) * ld] = y[r];
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[r] = y[r];
        for (int st = 0; st < s; ++st) {
            const double hb = dt * tab[st];
            const double ha = (st + 1 < s) ? dt * tab[s + st] : 0.0;
            if (stages && live) {
                double *sp = stages + ((ti - p.step_begin) * s + st) * A + m;
#pragma unroll
                for (int r = 0; r < RPW; ++r)
                    if (rows[r]) sp[(int64_t)(rows[r] - 1) * ld] = *(double *)(xs + (uint32_t)rows[r] * 512u + lane8);
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = rows[r];
                if (row) {
                    double k0 = 0.0, k1 = 0.0;               // two partial sums: independent FMA chains
                    const int t1 = row_term[row + 1];
                    for (int t = row_term[row]; t < t1; t += TPI) {
                        // TPI terms per trip: one round of scalar loads (j offsets, k offsets, coefficients) ...
                        u32x4 jo[TPI / 4], ko[TPI / 4];
                        f64x4 cc[TPI / 4];
#pragma unroll
                        for (int q = 0; q < TPI / 4; ++q) {
                            jo[q] = *(const cu32x4 *)(term_joff + t + 4 * q);
                            ko[q] = *(const cu32x4 *)(term_koff + t + 4 * q);
                            cc[q] = *(const cf64x4 *)(term_c + t + 4 * q);
                        }
                        // ... then the LDS reads and the FMAs, eight terms (16 ds_read_b64) at a time
#pragma unroll
                        for (int h = 0; h < TPI / 4; h += 2) {
                            double xj[8], xk[8];
#pragma unroll
                            for (int q = 0; q < 2 && h + q < TPI / 4; ++q) {
                                xj[4 * q + 0] = *(const double *)(xs + jo[h + q].x + lane8); xk[4 * q + 0] = *(const double *)(xs + ko[h + q].x + lane8);
                                xj[4 * q + 1] = *(const double *)(xs + jo[h + q].y + lane8); xk[4 * q + 1] = *(const double *)(xs + ko[h + q].y + lane8);
                                xj[4 * q + 2] = *(const double *)(xs + jo[h + q].z + lane8); xk[4 * q + 2] = *(const double *)(xs + ko[h + q].z + lane8);
                                xj[4 * q + 3] = *(const double *)(xs + jo[h + q].w + lane8); xk[4 * q + 3] = *(const double *)(xs + ko[h + q].w + lane8);
                            }
#pragma unroll
                            for (int q = 0; q < 2 && h + q < TPI / 4; ++q) {
                                k0 = __builtin_fma(cc[h + q].x, xj[4 * q + 0] * xk[4 * q + 0], k0);
                                k1 = __builtin_fma(cc[h + q].y, xj[4 * q + 1] * xk[4 * q + 1], k1);
                                k0 = __builtin_fma(cc[h + q].z, xj[4 * q + 2] * xk[4 * q + 2], k0);
                                k1 = __builtin_fma(cc[h + q].w, xj[4 * q + 3] * xk[4 * q + 3], k1);
                            }
                        }
                    }
                    const double k = k0 + k1;
                    acc[r] = __builtin_fma(hb, k, acc[r]);
                    xn[r] = __builtin_fma(ha, k, y[r]);
                }
            }
            __syncthreads();                               // every wavefront is done reading the stage state
            const bool last = (st == s - 1);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = rows[r];
                if (row) {
                    const double v = last ? acc[r] : xn[r];
                    *(double *)(xs + (uint32_t)row * 512u + lane8) = v;
                    if (last) y[r] = acc[r];
                }
            }
            __syncthreads();
        }
    }
    if (live) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = rows[r];
            if (row) {
                if (y_out) y_out[(int64_t)(row - 1) * ld + m] = y[r];
                if (p.write_final) rec[rec_index(p.n_records - 1, p.n_records, p.backward) * A + (int64_t)(row - 1) * ld + m] = y[r];
            }
        }
    }
}

template <int RPW, int TPI>
hipError_t launch_tiled(const TiledTensor &T, const RkArgs &p, const double *y_in, double *y_out, double *rec, double *stages,
                        const double *dtime, const double *tab, hipStream_t st)
{
    const size_t lds = (size_t)(p.ndim + 1) * 512;
    // The raised dynamic-LDS limit is remembered per device (a process may hold models on several GPUs: device groups, the
    // shards of the Lyapunov estimator, each driven by its own thread) -- the attribute is set on the current device.
    static DynLdsLimit configured;
    if (configured.needs(lds)) {
        hipError_t e = hipFuncSetAttribute((const void *)gen_rk_tiled_kernel<RPW, TPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured.set(lds);
    }
    hipLaunchKernelGGL((gen_rk_tiled_kernel<RPW, TPI>), dim3(blocks_for(p.n_traj, WAVE)), dim3(64 * TILED_NW), lds, st, T, p, y_in, y_out,
                       rec, stages, dtime, tab);
    return hipGetLastError();
}

// ---- wavefront-per-trajectory stepper ---------------------------------------------------------------------
// One workgroup (NWV wavefronts, 64*NWV >= ndim) per trajectory, lane = row.  REG_TERMS > 0: the row's terms
// (at most REG_TERMS) live in registers as (LDS byte offset of x_j, of x_k, coefficient); REG_TERMS == 0: streamed
// from the CSR arrays every stage.  S > 0: stage loop unrolled (tableau in registers, LDS buffer parity known at
// compile time so the double-buffer switch is an immediate offset); S == 0: run-time stage count.
constexpr int WAVE_XS_STRIDE = 264;                        // doubles per LDS stage buffer (>= 256 + 1)

template <int REG_TERMS>
__device__ __forceinline__ double wave_row_eval(const char *xb, const uint32_t *jo, const uint32_t *ko, const double *cf,
                                                const DevTensor &T, int e0, int e1)
{
    double k0 = 0.0, k1 = 0.0;
    if (REG_TERMS > 0) {
        double xj[REG_TERMS > 0 ? REG_TERMS : 1], xk[REG_TERMS > 0 ? REG_TERMS : 1];
#pragma unroll
        for (int t = 0; t < REG_TERMS; ++t) { xj[t] = *(const double *)(xb + jo[t]); xk[t] = *(const double *)(xb + ko[t]); }
#pragma unroll
        for (int t = 0; t < REG_TERMS; t += 2) {
            k0 = __builtin_fma(cf[t], xj[t] * xk[t], k0);
            if (t + 1 < REG_TERMS) k1 = __builtin_fma(cf[t + 1], xj[t + 1] * xk[t + 1], k1);
        }
    } else {
        for (int e = e0; e < e1; ++e) {
            const uint32_t q = T.idx[e];
            k0 = __builtin_fma(T.val[e], *(const double *)(xb + (q >> 16) * 8u) * *(const double *)(xb + (q & 0xffffu) * 8u), k0);
        }
    }
    return k0 + k1;
}

// Derived monomials of the stage state (rank-5 models), see generic_kernels.h DerivedChains.  The (a, b, slot) triples of
// this thread are loaded once; unused entries multiply slot 0 by itself into a scratch slot.
constexpr int WAVE_DER_SCRATCH = WAVE_XS_STRIDE - 1;
struct WaveDerived {
    int a[WAVE_DER_LEVELS][WAVE_DER_PER], b[WAVE_DER_LEVELS][WAVE_DER_PER], s[WAVE_DER_LEVELS][WAVE_DER_PER];
    int n_levels;
    __device__ __forceinline__ void load(const DerivedChains &D)
    {
        n_levels = D.n_levels;
#pragma unroll
        for (int l = 0; l < WAVE_DER_LEVELS; ++l)
#pragma unroll
            for (int q = 0; q < WAVE_DER_PER; ++q) {
                const int e = D.level_ptr[l] + (int)threadIdx.x + q * (int)blockDim.x;
                const bool ok = l < D.n_levels && e < D.level_ptr[l + 1];
                a[l][q] = ok ? D.a[e] : 0;
                b[l][q] = ok ? D.b[e] : 0;
                s[l][q] = ok ? D.slot[e] : WAVE_DER_SCRATCH;
            }
    }
    // call with the stage state complete in xbuf (after a barrier); ends with a barrier
    __device__ __forceinline__ void eval(double *xbuf) const
    {
#pragma unroll
        for (int l = 0; l < WAVE_DER_LEVELS; ++l) {
            if (l < n_levels) {
#pragma unroll
                for (int q = 0; q < WAVE_DER_PER; ++q) xbuf[s[l][q]] = xbuf[a[l][q]] * xbuf[b[l][q]];
                __syncthreads();
            }
        }
    }
};

template <int NWV, int REG_TERMS, int S>
__global__ void __launch_bounds__(64 * NWV) gen_rk_wave_kernel(DevTensor T, RkArgs p, const double *__restrict__ y_in,
                                                               double *__restrict__ y_out, double *__restrict__ rec,
                                                               double *__restrict__ stages,
                                                               const double *__restrict__ dtime,
                                                               const double *__restrict__ tab, DerivedChains D)
{
    __shared__ double xs[2 * WAVE_XS_STRIDE];              // two stage buffers, slot 0 of each = eta_0 = 1
    const int ndim = p.ndim, s = (S > 0) ? S : p.s;
    const int row = (int)threadIdx.x + 1;                  // tensor row of this lane
    const bool active = row <= ndim;
    const int64_t m = blockIdx.x;                          // trajectory
    const int64_t ld = p.ld, A = (int64_t)ndim * ld;
    const int e0 = active ? T.rowptr[row] : 0, e1 = active ? T.rowptr[row + 1] : 0;

    uint32_t jo[REG_TERMS > 0 ? REG_TERMS : 1], ko[REG_TERMS > 0 ? REG_TERMS : 1];
    double cf[REG_TERMS > 0 ? REG_TERMS : 1];
    if (REG_TERMS > 0) {
#pragma unroll
        for (int t = 0; t < REG_TERMS; ++t) {              // pad with zero terms reading slot 0
            const bool have = e0 + t < e1;
            const uint32_t q = have ? T.idx[e0 + t] : 0u;
            jo[t] = (q >> 16) * 8u;
            ko[t] = (q & 0xffffu) * 8u;
            cf[t] = have ? T.val[e0 + t] : 0.0;
        }
    }
    double tb[S > 0 ? S : 1], ta[S > 0 ? S : 1];
    if (S > 0) {
#pragma unroll
        for (int st = 0; st < S; ++st) { tb[st] = tab[st]; ta[st] = (st + 1 < S) ? tab[S + st] : 0.0; }
    }
    double y = active ? y_in[(int64_t)(row - 1) * ld + m] : 0.0;
    if (threadIdx.x == 0) { xs[0] = 1.0; xs[WAVE_XS_STRIDE] = 1.0; }
    if (active) xs[row] = y;
    WaveDerived wd;
    wd.load(D);
    __syncthreads();
    wd.eval(xs);

    int64_t iw = 0, next_rec = -1;
    if (p.write_steps > 0) { iw = (p.step_begin + p.write_steps - 1) / p.write_steps; next_rec = iw * p.write_steps; }
    const char *xb0 = (const char *)xs, *xb1 = (const char *)(xs + WAVE_XS_STRIDE);
    int cur = 0;                                           // buffer holding the current stage input (run-time s only)
    for (int64_t ti = p.step_begin; ti < p.step_end; ++ti) {
        const double dt = dtime[ti + 1] - dtime[ti];
        if (ti == next_rec) {
            if (active) rec[rec_index(iw, p.n_records, p.backward) * A + (int64_t)(row - 1) * ld + m] = y;
            ++iw; next_rec += p.write_steps;
        }
        double acc = y;
        if (S > 0) {
            // every step starts and (for even S) ends in buffer (ti*S) & 1; parity per stage is a compile-time pattern
            // only for even S, so odd S toggles `cur` like the run-time path
#pragma unroll
            for (int st = 0; st < S; ++st) {
                const bool in1 = (S % 2 == 0) ? (st & 1) : ((cur + st) & 1);
                const char *xb = in1 ? xb1 : xb0;
                double *xo = in1 ? xs : xs + WAVE_XS_STRIDE;
                if (stages && active) stages[((ti - p.step_begin) * S + st) * A + (int64_t)(row - 1) * ld + m] = *(const double *)(xb + row * 8);
                const double k = wave_row_eval<REG_TERMS>(xb, jo, ko, cf, T, e0, e1);
                acc = __builtin_fma(dt * tb[st], k, acc);
                const bool last = (st == S - 1);
                const double xn = last ? acc : __builtin_fma(dt * ta[st], k, y);
                if (active) xo[row] = xn;
                if (last) y = acc;
                __syncthreads();
                wd.eval(xo);
            }
            if (S % 2) cur ^= 1;
        } else {
            for (int st = 0; st < s; ++st) {
                const char *xb = cur ? xb1 : xb0;
                double *xo = cur ? xs : xs + WAVE_XS_STRIDE;
                if (stages && active) stages[((ti - p.step_begin) * s + st) * A + (int64_t)(row - 1) * ld + m] = *(const double *)(xb + row * 8);
                const double k = wave_row_eval<REG_TERMS>(xb, jo, ko, cf, T, e0, e1);
                acc = __builtin_fma(dt * tab[st], k, acc);
                const bool last = (st == s - 1);
                const double xn = last ? acc : __builtin_fma(dt * tab[s + st], k, y);
                if (active) xo[row] = xn;
                if (last) y = acc;
                cur ^= 1;
                __syncthreads();
                wd.eval(xo);
            }
        }
    }
    if (active) {
        if (y_out) y_out[(int64_t)(row - 1) * ld + m] = y;
        if (p.write_final) rec[rec_index(p.n_records - 1, p.n_records, p.backward) * A + (int64_t)(row - 1) * ld + m] = y;
    }
}

// Long rows (more than 32 terms: MAOOAM 6x6 has up to 237): G lanes share a row.  Thread (row, sub) streams the terms
// sub, sub + G, ... of its row from the CSR arrays, the G partial sums are combined with shuffles (G is a power of two and
// divides 64, so a row never straddles wavefronts), lane sub == 0 does the Runge-Kutta update.  G is chosen so that
// ndim * G threads fill a workgroup of up to 1024: one MAOOAM 6x6 trajectory advances with 912 lanes instead of 228.
__global__ void __launch_bounds__(1024) gen_rk_waveg_kernel(DevTensor T, RkArgs p, int G, const double *__restrict__ y_in,
                                                            double *__restrict__ y_out, double *__restrict__ rec,
                                                            double *__restrict__ stages, const double *__restrict__ dtime,
                                                            const double *__restrict__ tab, DerivedChains D)
{
    __shared__ double xs[2 * WAVE_XS_STRIDE];
    const int ndim = p.ndim, s = p.s;
    const int row = (int)threadIdx.x / G + 1, sub = (int)threadIdx.x % G;
    const bool active = row <= ndim, owner = active && sub == 0;
    const int64_t m = blockIdx.x;
    const int64_t ld = p.ld, A = (int64_t)ndim * ld;
    const int e0 = active ? T.rowptr[row] : 0, e1 = active ? T.rowptr[row + 1] : 0;
    double y = active ? y_in[(int64_t)(row - 1) * ld + m] : 0.0;
    if (threadIdx.x == 0) { xs[0] = 1.0; xs[WAVE_XS_STRIDE] = 1.0; }
    if (owner) xs[row] = y;
    WaveDerived wd;
    wd.load(D);
    __syncthreads();
    wd.eval(xs);
    int64_t iw = 0, next_rec = -1;
    if (p.write_steps > 0) { iw = (p.step_begin + p.write_steps - 1) / p.write_steps; next_rec = iw * p.write_steps; }
    int cur = 0;
    for (int64_t ti = p.step_begin; ti < p.step_end; ++ti) {
        const double dt = dtime[ti + 1] - dtime[ti];
        if (ti == next_rec) {
            if (owner) rec[rec_index(iw, p.n_records, p.backward) * A + (int64_t)(row - 1) * ld + m] = y;
            ++iw; next_rec += p.write_steps;
        }
        double acc = y;
        for (int st = 0; st < s; ++st) {
            const char *xb = (const char *)(xs + cur * WAVE_XS_STRIDE);
            double *xo = xs + (cur ^ 1) * WAVE_XS_STRIDE;
            if (stages && owner) stages[((ti - p.step_begin) * s + st) * A + (int64_t)(row - 1) * ld + m] = *(const double *)(xb + row * 8);
            double k0 = 0.0, k1 = 0.0;
            int e = e0 + sub;
            for (; e + G < e1; e += 2 * G) {               // two independent chains per lane
                const uint32_t qa = T.idx[e], qb = T.idx[e + G];
                k0 = __builtin_fma(T.val[e], *(const double *)(xb + (qa >> 16) * 8u) * *(const double *)(xb + (qa & 0xffffu) * 8u), k0);
                k1 = __builtin_fma(T.val[e + G], *(const double *)(xb + (qb >> 16) * 8u) * *(const double *)(xb + (qb & 0xffffu) * 8u), k1);
            }
            if (e < e1) {
                const uint32_t qa = T.idx[e];
                k0 = __builtin_fma(T.val[e], *(const double *)(xb + (qa >> 16) * 8u) * *(const double *)(xb + (qa & 0xffffu) * 8u), k0);
            }
            double k = k0 + k1;
            for (int off = G >> 1; off > 0; off >>= 1) k += __shfl_xor(k, off);
            acc = __builtin_fma(dt * tab[st], k, acc);
            const bool last = (st == s - 1);
            const double xn = last ? acc : __builtin_fma(dt * tab[s + st], k, y);
            if (owner) xo[row] = xn;
            if (last) y = acc;
            cur ^= 1;
            __syncthreads();
            wd.eval(xo);
        }
    }
    if (owner) {
        if (y_out) y_out[(int64_t)(row - 1) * ld + m] = y;
        if (p.write_final) rec[rec_index(p.n_records - 1, p.n_records, p.backward) * A + (int64_t)(row - 1) * ld + m] = y;
    }
}

template <int NWV, int REG_TERMS>
hipError_t launch_wave(const DevTensor &T, const RkArgs &p, const double *y_in, double *y_out, double *rec, double *stages,
                       const double *dtime, const double *tab, hipStream_t st, const DerivedChains &D)
{
    const dim3 grid((unsigned)p.n_traj), block(64 * NWV);
    switch (p.s) {
    case 2: hipLaunchKernelGGL((gen_rk_wave_kernel<NWV, REG_TERMS, 2>), grid, block, 0, st, T, p, y_in, y_out, rec, stages, dtime, tab, D); break;
    case 4: hipLaunchKernelGGL((gen_rk_wave_kernel<NWV, REG_TERMS, 4>), grid, block, 0, st, T, p, y_in, y_out, rec, stages, dtime, tab, D); break;
    default: hipLaunchKernelGGL((gen_rk_wave_kernel<NWV, REG_TERMS, 0>), grid, block, 0, st, T, p, y_in, y_out, rec, stages, dtime, tab, D); break;
    }
    return hipGetLastError();
}

// ---- wavefront-per-(member, column) tangent model -----------------------------------------------------------
template <int REG_TERMS>
__device__ __forceinline__ double wave_wx_eval(const char *xb, const char *wb, const uint32_t *wo, const uint32_t *xo,
                                               const double *cf, const DevTensor &J, int e0, int e1)
{
    double k0 = 0.0, k1 = 0.0;
    if (REG_TERMS > 0) {
        double xv[REG_TERMS > 0 ? REG_TERMS : 1], wv[REG_TERMS > 0 ? REG_TERMS : 1];
#pragma unroll
        for (int t = 0; t < REG_TERMS; ++t) { xv[t] = *(const double *)(xb + xo[t]); wv[t] = *(const double *)(wb + wo[t]); }
#pragma unroll
        for (int t = 0; t < REG_TERMS; t += 2) {
            k0 = __builtin_fma(cf[t], xv[t] * wv[t], k0);
            if (t + 1 < REG_TERMS) k1 = __builtin_fma(cf[t + 1], xv[t + 1] * wv[t + 1], k1);
        }
    } else {
        for (int e = e0; e < e1; ++e) {
            const uint32_t q = J.idx[e];
            k0 = __builtin_fma(J.val[e], *(const double *)(xb + (q & 0xffffu) * 8u) * *(const double *)(wb + (q >> 16) * 8u), k0);
        }
    }
    return k0 + k1;
}

template <int NWV, int REG_TERMS>
__global__ void __launch_bounds__(64 * NWV) gen_tgl_wave_kernel(DevTensor J, RkArgs p, int64_t n_tg, double inverse,
                                                                const double *__restrict__ w_in, double *__restrict__ w_out,
                                                                double *__restrict__ rec, const double *__restrict__ stages,
                                                                const double *__restrict__ dtime, const double *__restrict__ tab,
                                                                DerivedChains D)
{
    __shared__ double xsh[WAVE_XS_STRIDE];                 // stage state of the member (+ derived monomials), slot 0 = 1
    __shared__ double wsh[2 * WAVE_XS_STRIDE];             // tangent stage vector, double buffered (slot 0 unused = 0)
    const int ndim = p.ndim, s = p.s;
    const int row = (int)threadIdx.x + 1;
    const bool active = row <= ndim;
    const int64_t col = blockIdx.x / p.n_traj, m = blockIdx.x % p.n_traj;
    const int64_t ld = p.ld, L = n_tg * ld, l = col * ld + m;
    const int64_t A = (int64_t)ndim * L, AS = (int64_t)ndim * ld;
    const int e0 = active ? J.rowptr[row] : 0, e1 = active ? J.rowptr[row + 1] : 0;
    uint32_t wo[REG_TERMS > 0 ? REG_TERMS : 1], xo[REG_TERMS > 0 ? REG_TERMS : 1];
    double cf[REG_TERMS > 0 ? REG_TERMS : 1];
    if (REG_TERMS > 0) {
#pragma unroll
        for (int t = 0; t < REG_TERMS; ++t) {
            const bool have = e0 + t < e1;
            const uint32_t q = have ? J.idx[e0 + t] : 0u;
            wo[t] = (q >> 16) * 8u;
            xo[t] = (q & 0xffffu) * 8u;
            cf[t] = have ? J.val[e0 + t] : 0.0;
        }
    }
    double v = active ? w_in[(int64_t)(row - 1) * L + l] : 0.0;
    if (threadIdx.x == 0) { xsh[0] = 1.0; wsh[0] = 0.0; wsh[WAVE_XS_STRIDE] = 0.0; }
    if (active) wsh[row] = v;
    WaveDerived wd;
    wd.load(D);
    int64_t iw = 0, next_rec = -1;
    if (p.write_steps > 0) { iw = (p.step_begin + p.write_steps - 1) / p.write_steps; next_rec = iw * p.write_steps; }
    int cur = 0;
    for (int64_t ti = p.step_begin; ti < p.step_end; ++ti) {
        const double dt = dtime[ti + 1] - dtime[ti];
        if (ti == next_rec) {
            if (active) rec[rec_index(iw, p.n_records, p.backward) * A + (int64_t)(row - 1) * L + l] = v;
            ++iw; next_rec += p.write_steps;
        }
        double acc = v;
        for (int st = 0; st < s; ++st) {
            if (active) xsh[row] = stages[((ti - p.step_begin) * s + st) * AS + (int64_t)(row - 1) * ld + m];
            __syncthreads();                               // x of this stage and w written by the previous stage visible
            wd.eval(xsh);
            const char *wb = (const char *)(wsh + cur * WAVE_XS_STRIDE);
            const double k = inverse * wave_wx_eval<REG_TERMS>((const char *)xsh, wb, wo, xo, cf, J, e0, e1);
            acc = __builtin_fma(dt * tab[st], k, acc);
            const bool last = (st == s - 1);
            const double wn = last ? acc : __builtin_fma(dt * tab[s + st], k, v);
            __syncthreads();                               // everybody has read xsh before the next stage overwrites it
            if (active) wsh[(cur ^ 1) * WAVE_XS_STRIDE + row] = wn;
            if (last) v = acc;
            cur ^= 1;
        }
    }
    if (active) {
        if (w_out) w_out[(int64_t)(row - 1) * L + l] = v;
        if (p.write_final) rec[rec_index(p.n_records - 1, p.n_records, p.backward) * A + (int64_t)(row - 1) * L + l] = v;
    }
}

// ---- batched QR ------------------------------------------------------------------------------------------
// One wavefront per member, lane = column, the matrix lives in LDS as A[row][lane] with a row stride of 65
// doubles (column walks are then bank-conflict free).  Unblocked Householder with LAPACK's conventions
// (dgeqr2 / dlarfg: beta = -sign(alpha)*norm, v_0 = 1; dorg2r for Q).  The reflector is applied to all columns
// c > j at once: every lane accumulates v^T a_c for its own column (v_i is a broadcast LDS read), so the only
// cross-lane reduction per step is the norm of the pivot column.
constexpr int QR_STRIDE = 65;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// A[j][c] + sum_{i>j} A[i][j] * A[i][c] with four independent partial sums (the LDS reads of consecutive rows
// are independent, a single accumulator would serialise them behind the FMA latency)
__device__ __forceinline__ double qr_col_dot(const double *A, int j, int c, int n_rows)
{
    double w0 = A[j * QR_STRIDE + c], w1 = 0.0, w2 = 0.0, w3 = 0.0;
    int i = j + 1;
    for (; i + 3 < n_rows; i += 4) {
        w0 = __builtin_fma(A[i * QR_STRIDE + j], A[i * QR_STRIDE + c], w0);
        w1 = __builtin_fma(A[(i + 1) * QR_STRIDE + j], A[(i + 1) * QR_STRIDE + c], w1);
        w2 = __builtin_fma(A[(i + 2) * QR_STRIDE + j], A[(i + 2) * QR_STRIDE + c], w2);
        w3 = __builtin_fma(A[(i + 3) * QR_STRIDE + j], A[(i + 3) * QR_STRIDE + c], w3);
    }
    for (; i < n_rows; ++i) w0 = __builtin_fma(A[i * QR_STRIDE + j], A[i * QR_STRIDE + c], w0);
    return (w0 + w1) + (w2 + w3);
}

// Member of workgroup b in the batched QR kernels (grid = 8 * ceil(n_traj / 8)).  Workgroups go round-robin to the 8 XCDs,
// each with its own L2, and 16 consecutive members share every 128-byte line of A[row][col][member]: XCD x takes the x-th
// eighth of the ensemble, so the workgroups that share a line run on one XCD at about the same time (36 x 36, 16 384
// members: 0.37 instead of 0.53 ms).  Returns n_traj or more for the padding workgroups.
__device__ __forceinline__ int64_t qr_member(unsigned b, int64_t n_traj)
{
    const int64_t per = (n_traj + 7) / 8;
    return (int64_t)(b & 7) * per + (int64_t)(b >> 3);
}

__global__ void __launch_bounds__(WAVE) batched_qr_kernel(int n_rows, int n_cols, int64_t n_traj, int64_t ld,
                                                          double *__restrict__ a, double *__restrict__ rdiag)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *A = (double *)smem;                            // [n_rows][QR_STRIDE]
    const int64_t m = qr_member(blockIdx.x, n_traj);
    if (m >= n_traj) return;                               // (the whole workgroup)
    const int c = threadIdx.x;                             // own column
    const bool col = c < n_cols;
    for (int i = 0; i < n_rows; ++i) A[i * QR_STRIDE + c] = col ? a[((int64_t)i * n_cols + c) * ld + m] : 0.0;
    __syncthreads();
    const int k = n_cols < n_rows ? n_cols : n_rows;
    double my_tau = 0.0;                                   // lane j keeps tau_j
    for (int j = 0; j < k; ++j) {
        // ---- dlarfg on column j (rows j..n_rows-1): lanes cooperate over the rows below the diagonal
        double part = 0.0;
        for (int i = j + 1 + c; i < n_rows; i += WAVE) { const double v = A[i * QR_STRIDE + j]; part = __builtin_fma(v, v, part); }
        const double xn2 = wave_sum(part);
        const double alpha = A[j * QR_STRIDE + j];
        double t = 0.0, beta = alpha;
        if (xn2 != 0.0) {
            beta = -copysign(sqrt(__builtin_fma(alpha, alpha, xn2)), alpha);
            t = (beta - alpha) / beta;
            const double scale = 1.0 / (alpha - beta);
            for (int i = j + 1 + c; i < n_rows; i += WAVE) A[i * QR_STRIDE + j] *= scale;
        }
        if (c == j) { my_tau = t; rdiag[(int64_t)j * ld + m] = beta; }
        __syncthreads();
        // ---- dlarf: columns c > j get a_c -= tau * v * (v^T a_c), v = (1, A[j+1:, j])
        if (col && c > j) {
            double w = qr_col_dot(A, j, c, n_rows) * t;
            A[j * QR_STRIDE + c] -= w;
#pragma unroll 4
            for (int i = j + 1; i < n_rows; ++i) A[i * QR_STRIDE + c] = __builtin_fma(-w, A[i * QR_STRIDE + j], A[i * QR_STRIDE + c]);
        }
        if (c == j) A[j * QR_STRIDE + j] = beta;
        __syncthreads();
    }
    for (int j = k - 1; j >= 0; --j) {                     // ---- dorg2r
        const double t = __shfl(my_tau, j);
        if (col && c > j) {                                // apply H_j to Q[j:, j+1:], Q[j][j] taken as 1
            double w = qr_col_dot(A, j, c, n_rows) * t;
            A[j * QR_STRIDE + c] -= w;
#pragma unroll 4
            for (int i = j + 1; i < n_rows; ++i) A[i * QR_STRIDE + c] = __builtin_fma(-w, A[i * QR_STRIDE + j], A[i * QR_STRIDE + c]);
        }
        __syncthreads();
        // column j itself: Q[i][j] = -tau v_i (i > j), Q[j][j] = 1 - tau, zeros above; lanes cooperate over rows
        for (int i = c; i < n_rows; i += WAVE) {
            const double v = A[i * QR_STRIDE + j];
            A[i * QR_STRIDE + j] = (i > j) ? -t * v : ((i == j) ? 1.0 - t : 0.0);
        }
        __syncthreads();
    }
    if (col) for (int i = 0; i < n_rows; ++i) a[((int64_t)i * n_cols + c) * ld + m] = A[i * QR_STRIDE + c];
}

// ---- layout conversion ----------------------------------------------------------------------
// 64 members x 64 inner elements per block through an LDS tile so that both sides are coalesced.
constexpr int TILE = 64;

__global__ void __launch_bounds__(256) pack_kernel(int64_t n_inner, int64_t n_traj, int64_t ld,
                                                   const double *__restrict__ rows, double *__restrict__ modes)
{
    __shared__ double tile[TILE][TILE + 1];
    const int64_t m0 = (int64_t)blockIdx.x * TILE, q0 = (int64_t)blockIdx.y * TILE;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;           // 64 x 4
    for (int r = ty; r < TILE; r += 4) {                              // read rows[m][q]: q fastest
        const int64_t m = m0 + r, q = q0 + tx;
        if (m < n_traj && q < n_inner) tile[r][tx] = rows[m * n_inner + q];
    }
    __syncthreads();
    for (int r = ty; r < TILE; r += 4) {                              // write modes[q][m]: m fastest
        const int64_t q = q0 + r, m = m0 + tx;
        if (m < n_traj && q < n_inner) modes[q * ld + m] = tile[tx][r];
    }
}

__global__ void __launch_bounds__(256) unpack_kernel(int64_t n_inner, int64_t n_traj, int64_t ld,
                                                     const double *__restrict__ modes, double *__restrict__ rows)
{
    __shared__ double tile[TILE][TILE + 1];
    const int64_t m0 = (int64_t)blockIdx.x * TILE, q0 = (int64_t)blockIdx.y * TILE;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < TILE; r += 4) {
        const int64_t q = q0 + r, m = m0 + tx;
        if (m < n_traj && q < n_inner) tile[r][tx] = modes[q * ld + m];
    }
    __syncthreads();
    for (int r = ty; r < TILE; r += 4) {
        const int64_t m = m0 + r, q = q0 + tx;
        if (m < n_traj && q < n_inner) rows[m * n_inner + q] = tile[tx][r];
    }
}

// A window of W records R[W][n_inner][ld] -> columns [0, W) of out, where element (m, q, r) of the output sits at
// out[(m * n_inner + q) * out_stride + r] (out already points at the window's first record column; out_stride = the
// record count of the whole run).  64 members x 64 consecutive (q, r) pairs per block through an LDS tile: reads are
// coalesced along the members, writes along r (runs of W doubles, one run of n_inner * W when W == out_stride), which
// is what a destination in host memory (PCIe writes) needs.
__global__ void __launch_bounds__(256) unpack_window_kernel(int64_t n_inner, int64_t n_traj, int64_t ld, int64_t W,
                                                            int64_t out_stride, const double *__restrict__ in,
                                                            double *__restrict__ out)
{
    __shared__ double tile[TILE][TILE + 1];
    const int64_t m0 = (int64_t)blockIdx.x * TILE, total = n_inner * W;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int64_t c0 = (int64_t)blockIdx.y * TILE; c0 < total; c0 += (int64_t)gridDim.y * TILE) {
        for (int i = ty; i < TILE; i += 4) {
            const int64_t c = c0 + i, m = m0 + tx;
            if (c < total && m < n_traj) {
                const int64_t q = c / W, r = c - q * W;                 // uniform per wavefront
                tile[i][tx] = in[(r * n_inner + q) * ld + m];
            }
        }
        __syncthreads();
        const int64_t c = c0 + tx;
        if (c < total) {
            const int64_t q = c / W, r = c - q * W;
            double *o = out + q * out_stride + r;
            for (int i = ty; i < TILE; i += 4) {
                const int64_t m = m0 + i;
                if (m < n_traj) o[m * n_inner * out_stride] = tile[tx][i];
            }
        }
        __syncthreads();
    }
}

}  // namespace

void launch_gen_tend(const DevTensor &T, int ndim, int64_t n_traj, int64_t ld, const double *x, double *dx, hipStream_t st)
{
    hipLaunchKernelGGL(gen_tend_kernel, dim3(blocks_for(n_traj, WAVE)), dim3(WAVE), 0, st, T, ndim, n_traj, ld, x, dx);
}

void launch_gen_jac(const DevTensor &Jt, int ndim, int64_t n_traj, int64_t ld, const double *x, double *jm, hipStream_t st)
{
    hipLaunchKernelGGL(gen_jac_kernel, dim3(blocks_for(n_traj, WAVE)), dim3(WAVE), 0, st, Jt, ndim, n_traj, ld, x, jm);
}

void launch_gen_tend_one(const DevTensor &T, int ndim, const double *x, double *dx, unsigned *counter,
                         unsigned long long *flag, unsigned long long seq, hipStream_t st)
{
    // four rows (wavefronts) per workgroup.  (One workgroup of 16 wavefronts for a whole small system, which needs no
    // cross-workgroup completion count, measured slower: 20.4 instead of 15.1 us per call at ndim 36.)
    const int blocks = std::min(256, (ndim + 3) / 4);
    hipLaunchKernelGGL(gen_tend_one_kernel, dim3(blocks), dim3(256), sizeof(double) * (size_t)(ndim + 1), st, T, ndim, x, dx, counter,
                       flag, seq);
}

void launch_gen_jac_one(const OnePairs &P, int ndim, const double *x, double *jm, unsigned *counter, unsigned long long *flag,
                        unsigned long long seq, hipStream_t st)
{
    const int blocks = std::min(256, (ndim * ndim + 255) / 256);
    hipLaunchKernelGGL(gen_jac_one_kernel, dim3(blocks), dim3(256), sizeof(double) * (size_t)(ndim + 1), st, P, ndim, x, jm, counter,
                       flag, seq);
}

// ---- general contraction with explicit vectors (sparse_mul3 / 5 / 2 / 4 with any arguments) ---------------------------------------
// One thread per non-empty output element, entries in their incoming order, products left to right and then times the
// value, contraction into FMAs switched off for this kernel: the same sequence of IEEE operations as the reference's loops
// `res[i] += a[j] * b[k] * val[n]` (sparse_mul.py:76-78).
__global__ void __launch_bounds__(256) contract_kernel(int n_out, const int32_t *__restrict__ out_index, const int32_t *__restrict__ ptr,
                                                       const uint32_t *__restrict__ fidx, const double *__restrict__ val, int n_fac,
                                                       const double *__restrict__ vecs, int n_slots, double *__restrict__ out)
{
#pragma clang fp contract(off)       // (HIP's __dmul_rn / __dadd_rn are plain * and +: only this keeps `s + p * val` two roundings)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_out) return;
    double s = 0.0;
    for (int e = ptr[t]; e < ptr[t + 1]; ++e) {
        const uint32_t *f = fidx + (size_t)e * n_fac;
        double p = vecs[f[0]];
        for (int q = 1; q < n_fac; ++q) p = p * vecs[(size_t)q * n_slots + f[q]];
        p = p * val[e];
        s = s + p;
    }
    out[out_index[t]] = s;
}

void launch_contract(int n_out, const int32_t *out_index, const int32_t *ptr, const uint32_t *fidx, const double *val, int n_fac,
                     const double *vecs, int n_slots, double *out, hipStream_t st)
{
    if (n_out < 1) return;
    hipLaunchKernelGGL(contract_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, n_out, out_index, ptr, fidx, val, n_fac, vecs,
                       n_slots, out);
}

void launch_gen_rk(const DevTensor &T, const RkArgs &p, const double *y_in, double *y_out, double *rec, double *stages,
                   double *work, const double *dtime, const double *tab_full, hipStream_t st)
{
    hipLaunchKernelGGL(gen_rk_kernel, dim3(blocks_for(p.n_traj, WAVE)), dim3(WAVE), 0, st, T, p, y_in, y_out, rec, stages,
                       work, dtime, tab_full);
}

void launch_gen_tgl(const DevTensor &Jrow, const RkArgs &p, int64_t n_tg, double inverse, const double *w_in, double *w_out,
                    double *rec, const double *stages, double *work, const double *dtime, const double *tab_full,
                    hipStream_t st)
{
    hipLaunchKernelGGL(gen_tgl_kernel, dim3(blocks_for(n_tg * p.ld, WAVE)), dim3(WAVE), 0, st, Jrow, p, n_tg, inverse, w_in,
                       w_out, rec, stages, work, dtime, tab_full);
}

bool wave_supported(int n_slots) { return n_slots <= 256; }

hipError_t launch_gen_rk_wave(const DevTensor &T, int max_row_terms, const RkArgs &p, const double *y_in, double *y_out,
                              double *rec, double *stages, const double *dtime, const double *tab_spec, hipStream_t st,
                              const DerivedChains &D)
{
    const int nwv = (p.ndim + 63) / 64;
    if (max_row_terms > 32) {                              // long rows: G lanes per row
        int G = 1;
        while (G < 16 && p.ndim * (2 * G) <= 1024) G *= 2;
        const int threads = (p.ndim * G + 63) / 64 * 64;
        hipLaunchKernelGGL(gen_rk_waveg_kernel, dim3((unsigned)p.n_traj), dim3(threads), 0, st, T, p, G, y_in, y_out, rec, stages, dtime,
                           tab_spec, D);
        return hipGetLastError();
    }
#define QGS_WAVE_CASE(N)                                                                                          \
    if (nwv == N) {                                                                                               \
        if (max_row_terms <= 16) return launch_wave<N, 16>(T, p, y_in, y_out, rec, stages, dtime, tab_spec, st, D); \
        if (max_row_terms <= 32) return launch_wave<N, 32>(T, p, y_in, y_out, rec, stages, dtime, tab_spec, st, D); \
        return launch_wave<N, 0>(T, p, y_in, y_out, rec, stages, dtime, tab_spec, st, D);                           \
    }
    QGS_WAVE_CASE(1)
    QGS_WAVE_CASE(2)
    QGS_WAVE_CASE(3)
    QGS_WAVE_CASE(4)
#undef QGS_WAVE_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_gen_tgl_wave(const DevTensor &Jrow, int max_row_terms, const RkArgs &p, int64_t n_tg, double inverse,
                               const double *w_in, double *w_out, double *rec, const double *stages, const double *dtime,
                               const double *tab_spec, hipStream_t st, const DerivedChains &D)
{
    const int nwv = (p.ndim + 63) / 64;
    const dim3 grid((unsigned)(p.n_traj * n_tg));
#define QGS_TGLW_CASE(N)                                                                                                 \
    if (nwv == N) {                                                                                                      \
        if (max_row_terms <= 32)                                                                                         \
            hipLaunchKernelGGL((gen_tgl_wave_kernel<N, 32>), grid, dim3(64 * N), 0, st, Jrow, p, n_tg, inverse, w_in, w_out, rec, \
                               stages, dtime, tab_spec, D);                                                              \
        else                                                                                                             \
            hipLaunchKernelGGL((gen_tgl_wave_kernel<N, 0>), grid, dim3(64 * N), 0, st, Jrow, p, n_tg, inverse, w_in, w_out, rec, \
                               stages, dtime, tab_spec, D);                                                              \
        return hipGetLastError();                                                                                        \
    }
    QGS_TGLW_CASE(1)
    QGS_TGLW_CASE(2)
    QGS_TGLW_CASE(3)
    QGS_TGLW_CASE(4)
#undef QGS_TGLW_CASE
    return hipErrorInvalidValue;
}

// ---- ensemble moments: mean and variance over the members of every row of X[row][member] ----------------------
// (reference: np.mean(f(traj), axis=0) in TrajectoriesStatistics.compute_stats, statistics.py:55-63, for the
// observables x and x^2 -- the trajectories stay on the device, only the moments travel.)
// Pass 1: grid (n_rows, n_split), every workgroup sums (x - shift) and (x - shift)^2 over its slice of the members;
// shift = x[row][0] keeps the second moment well conditioned.  Pass 2: one thread per row combines the partial sums.
constexpr int MOM_THREADS = 256;

__global__ void __launch_bounds__(MOM_THREADS) moments_partial_kernel(const double *__restrict__ x, int64_t ld, int64_t n_traj,
                                                                      int n_split, double *__restrict__ part)
{
    __shared__ double sh1[MOM_THREADS / WAVE], sh2[MOM_THREADS / WAVE];
    const int64_t row = blockIdx.x;
    const int split = blockIdx.y;
    const double *xr = x + row * ld;
    const double shift = xr[0];
    // slices in units of 2 members (16-byte loads; ld is a multiple of 64 so every row is 512-byte aligned)
    const int64_t pairs = n_traj / 2, per = (pairs + n_split - 1) / n_split;
    const int64_t p0 = (int64_t)split * per, p1 = p0 + per < pairs ? p0 + per : pairs;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t q = p0 + threadIdx.x; q < p1; q += MOM_THREADS) {
        const double2 v = *(const double2 *)(xr + 2 * q);
        const double a = v.x - shift, b = v.y - shift;
        s1 += a + b;
        s2 = __builtin_fma(a, a, __builtin_fma(b, b, s2));
    }
    if (split == n_split - 1 && (n_traj & 1) && threadIdx.x == 0) {
        const double a = xr[n_traj - 1] - shift;
        s1 += a;
        s2 = __builtin_fma(a, a, s2);
    }
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, WAVE); s2 += __shfl_down(s2, off, WAVE); }
    const int wv = threadIdx.x / WAVE;
    if ((threadIdx.x & (WAVE - 1)) == 0) { sh1[wv] = s1; sh2[wv] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t1 = 0.0, t2 = 0.0;
        for (int w = 0; w < MOM_THREADS / WAVE; ++w) { t1 += sh1[w]; t2 += sh2[w]; }
        part[(row * n_split + split) * 2] = t1;
        part[(row * n_split + split) * 2 + 1] = t2;
    }
}

__global__ void moments_final_kernel(const double *__restrict__ x, int64_t ld, int64_t n_traj, int64_t n_rows, int n_split,
                                     const double *__restrict__ part, double *__restrict__ mean, double *__restrict__ var)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    double t1 = 0.0, t2 = 0.0;
    for (int q = 0; q < n_split; ++q) { t1 += part[(row * n_split + q) * 2]; t2 += part[(row * n_split + q) * 2 + 1]; }
    const double inv = 1.0 / (double)n_traj, d = t1 * inv;
    mean[row] = x[row * ld] + d;
    if (var) var[row] = __builtin_fma(-d, d, t2 * inv);      // population variance (ddof = 0), as np.var
}

int moments_splits(int64_t n_rows, int64_t n_traj)
{
    // enough workgroups to fill the chip (~8 per CU), at least 2048 members per workgroup
    int64_t want = (2048 + n_rows - 1) / n_rows, cap = (n_traj + 2047) / 2048;
    int64_t s = want < cap ? want : cap;
    return (int)(s < 1 ? 1 : (s > 1024 ? 1024 : s));
}

// What this chip sustains on nothing but independent fp64 FMAs (eight chains per lane, eight wavefronts per SIMD): the practical
// ceiling bench.py sets its fp64 kernels against next to the nominal peak (the board lowers the clock under this load).
__global__ void __launch_bounds__(256) fma_rate_kernel(double *out, int iters)
{
    double x = 1.0000001 + threadIdx.x * 1e-9, y = 0.9999999;
    double a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
#pragma nounroll
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_fma(x, y, a0); a1 = __builtin_fma(x, y, a1); a2 = __builtin_fma(x, y, a2); a3 = __builtin_fma(x, y, a3);
        a4 = __builtin_fma(x, y, a4); a5 = __builtin_fma(x, y, a5); a6 = __builtin_fma(x, y, a6); a7 = __builtin_fma(x, y, a7);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

void launch_fma_rate(int blocks, int iters, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(fma_rate_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, iters);
}

void launch_moments(int64_t n_rows, int64_t n_traj, int64_t ld, const double *x, double *part, double *mean, double *var,
                    hipStream_t st)
{
    const int n_split = moments_splits(n_rows, n_traj);
    hipLaunchKernelGGL(moments_partial_kernel, dim3((unsigned)n_rows, (unsigned)n_split), dim3(MOM_THREADS), 0, st, x, ld, n_traj,
                       n_split, part);
    hipLaunchKernelGGL(moments_final_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, x, ld, n_traj, n_rows, n_split,
                       part, mean, var);
}

// Matrices that fit neither the register file nor the LDS (full Lyapunov spectrum of MAOOAM 6x6: 228 x 228 = 416 KB):
// one workgroup of 256 threads per matrix, the matrix in a contiguous scratch copy B[row][col] in global memory (column
// index fastest, so the threads of the column loops read consecutive doubles), the pivot column staged in LDS.  Same
// Householder conventions as above.  Functional rather than fast: these are few-member workloads.
constexpr int QRG_THREADS = 1024, QRG_COLS = 256, QRG_SUB = QRG_THREADS / QRG_COLS;    // 4 threads per column, each a quarter of the rows

__device__ __forceinline__ double block_sum(double v, double *red)
{
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int q = 0; q < QRG_THREADS / WAVE; ++q) t += red[q];
    return t;
}

// a_c -= tau * u (u^T a_c) for the columns c > j, u = (1, v[j+1:]); thread (sub, cc) takes the rows j+1+sub, j+1+sub+4, ...
// of the columns cc, cc + 256, ...; the four partial dot products of a column meet in LDS.
__device__ __forceinline__ void qrg_apply(double *B, const double *v, double *part, int j, int n_rows, int n_cols, double t)
{
    const int cc = threadIdx.x % QRG_COLS, sub = threadIdx.x / QRG_COLS;
    for (int c0 = j + 1; c0 < n_cols; c0 += QRG_COLS) {
        const int c = c0 + cc;
        const bool on = c < n_cols;
        double w0 = (on && sub == 0) ? B[(int64_t)j * n_cols + c] : 0.0, w1 = 0.0;
        if (on) {
            int i = j + 1 + sub;
            for (; i + QRG_SUB < n_rows; i += 2 * QRG_SUB) {
                w0 = __builtin_fma(v[i], B[(int64_t)i * n_cols + c], w0);
                w1 = __builtin_fma(v[i + QRG_SUB], B[(int64_t)(i + QRG_SUB) * n_cols + c], w1);
            }
            if (i < n_rows) w0 = __builtin_fma(v[i], B[(int64_t)i * n_cols + c], w0);
        }
        part[sub * QRG_COLS + cc] = w0 + w1;
        __syncthreads();
        const double w = t * ((part[cc] + part[QRG_COLS + cc]) + (part[2 * QRG_COLS + cc] + part[3 * QRG_COLS + cc]));
        if (on) {
            if (sub == 0) B[(int64_t)j * n_cols + c] -= w;
            for (int i = j + 1 + sub; i < n_rows; i += QRG_SUB) B[(int64_t)i * n_cols + c] = __builtin_fma(-w, v[i], B[(int64_t)i * n_cols + c]);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(QRG_THREADS) batched_qr_global_kernel(int n_rows, int n_cols, int64_t n_traj, int64_t ld,
                                                                        double *__restrict__ a, double *__restrict__ rdiag,
                                                                        double *__restrict__ scratch, double *__restrict__ taus)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *v = (double *)smem;                            // pivot column (n_rows)
    __shared__ double red[QRG_THREADS / WAVE];
    __shared__ double part[QRG_THREADS];
    __shared__ double s_t, s_beta, s_scale;
    const int64_t m = qr_member(blockIdx.x, n_traj);
    if (m >= n_traj) return;                               // (the whole workgroup)
    const int tid = threadIdx.x;
    double *B = scratch + (int64_t)m * n_rows * n_cols;
    double *tau = taus + (int64_t)m * n_cols;
    for (int64_t e = tid; e < (int64_t)n_rows * n_cols; e += QRG_THREADS) B[e] = a[e * ld + m];
    __syncthreads();
    const int k = n_cols < n_rows ? n_cols : n_rows;
    for (int j = 0; j < k; ++j) {
        // ---- dlarfg on column j
        double p2 = 0.0;
        for (int i = j + 1 + tid; i < n_rows; i += QRG_THREADS) { const double x = B[(int64_t)i * n_cols + j]; p2 = __builtin_fma(x, x, p2); }
        const double xn2 = block_sum(p2, red);
        if (tid == 0) {
            const double alpha = B[(int64_t)j * n_cols + j];
            double t = 0.0, beta = alpha, scale = 1.0;
            if (xn2 != 0.0) { beta = -copysign(sqrt(__builtin_fma(alpha, alpha, xn2)), alpha); t = (beta - alpha) / beta; scale = 1.0 / (alpha - beta); }
            s_t = t; s_beta = beta; s_scale = scale;
            tau[j] = t;
            rdiag[(int64_t)j * ld + m] = beta;
        }
        __syncthreads();
        const double t = s_t, scale = s_scale;
        for (int i = j + 1 + tid; i < n_rows; i += QRG_THREADS) {
            const double x = B[(int64_t)i * n_cols + j] * scale;
            B[(int64_t)i * n_cols + j] = x;
            v[i] = x;
        }
        __syncthreads();
        qrg_apply(B, v, part, j, n_rows, n_cols, t);       // ---- dlarf on the columns right of j
        if (tid == 0) B[(int64_t)j * n_cols + j] = s_beta;
        __syncthreads();
    }
    for (int j = k - 1; j >= 0; --j) {                     // ---- dorg2r
        const double t = tau[j];
        for (int i = j + 1 + tid; i < n_rows; i += QRG_THREADS) v[i] = B[(int64_t)i * n_cols + j];
        __syncthreads();
        qrg_apply(B, v, part, j, n_rows, n_cols, t);
        for (int i = tid; i < n_rows; i += QRG_THREADS)
            B[(int64_t)i * n_cols + j] = (i > j) ? -t * v[i] : ((i == j) ? 1.0 - t : 0.0);
        __syncthreads();
    }
    for (int64_t e = tid; e < (int64_t)n_rows * n_cols; e += QRG_THREADS) a[e * ld + m] = B[e];
}

// Blocked Householder QR for matrices beyond the generated kernels (cols > 64: the full 228 x 228 bases of MAOOAM 6x6, the
// reference's default n_vec = n_dim): LAPACK's dgeqrf + dorgqr with panels of NB = 16 columns (np.linalg.qr is these two routines,
// qgs/toolbox/lyapunov.py:524, 603).  The unblocked kernel above reads and writes the whole trailing matrix once per column --
// (2/3) n^3 x 16 bytes = 126 MB per 228 x 228 member through the L2, 30 ms for 1 024 members; here it is touched once per PANEL.
// One workgroup of 256 threads per member, the member's matrix in a contiguous scratch copy B[row][col] in global memory.  Per panel:
//  * the panel (rows j0 .., 16 columns) is factored in LDS, row-major with pitch 18 (128-bit reads of a row's 16 entries; 4-way
//    bank conflicts for the column sweeps of the factorisation, which are a few per cent of the work): every wavefront forms the
//    pivot's norm by itself (the same bits in all four), the remaining panel columns are dealt to the wavefronts, one barrier per step;
//  * T of the compact WY form (dlarft: H_1 .. H_nb = I - V T V^T) from the Gram matrix of the reflectors;
//  * the trailing columns X (rows j0 ..) become X - V T^T (V^T X): a thread owns two columns, accumulates its 2 x 16 entries of V^T X
//    over the rows with V's row broadcast from LDS (16 FMAs per loaded element), multiplies by T^T in registers and makes a second
//    pass for the update.  Consecutive threads hold consecutive columns: every global access is whole lines.
// dorgqr walks the panels backwards with H = I - V T V^T (T kept from the first phase in scratch): the rows of R above and inside a
// panel count as zeros on the way in, the panel's own columns start from the identity.
constexpr int QRB_THREADS = 256, QRB_NB = 16, QRB_PITCH = 18, QRB_AHEAD = 8, QRB_NC = 2;

// X(rows j0 .., columns c_first .. c_end - 1) -= V (U^T X) with U = V op(T) formed once per panel (qrb_form_u): the product with
// the 16 x 16 triangle is then part of the first pass instead of 272 FMAs per thread on 136 LDS operands (which the compiler
// loads all at once: 426 registers).  VIRT 0: X as stored; 1: the first nbk rows of X count as zeros (dorgqr: they hold entries of R).
template <int VIRT>
__device__ __forceinline__ void qrb_apply(double *__restrict__ B, const double *__restrict__ Vr, const double *__restrict__ Ur, int j0, int nbk,
                                           int c_first, int c_end, int n_rows, int n_cols)
{
    // A thread owns QRB_NC consecutive columns and every S-th row of them: S = 1, 2, 4 or 8 lanes of a wavefront share a column
    // group (the fewer groups are left, the more), their partial U^T X meet through shuffles.  Consecutive lanes hold consecutive
    // groups: whole lines.  The loads of QRB_AHEAD rows are issued before their FMAs.
    // (QRB_NC = 2 or 4 and QRB_AHEAD = 4 or 8 all measure the same, 5.2 - 5.3 ms for 1 024 x 228 x 228: the loops are bound by neither
    // the LDS operands (LDS busy 25 %) nor the FMAs (VALU 19 %) but by the trailing matrix itself -- 512 members in flight x 416 KB do
    // not fit the L2, every pass comes from HBM: 9.4 GB read + 4.7 GB written per launch, 2.7 TB/s with two wavefronts per SIMD;
    // profiles/r05_qr.md section 11.)
    const int RR = n_rows - j0, ncg = (c_end - c_first + QRB_NC - 1) / QRB_NC;
    int sb = 0;
    while (sb < 3 && (ncg << (sb + 1)) <= QRB_THREADS) ++sb;
    const int S = 1 << sb, lanes_cg = WAVE >> sb, per_pass = (QRB_THREADS / WAVE) * lanes_cg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slice = lane / lanes_cg;
    const int first = VIRT == 1 ? nbk : 0;
    for (int cg0 = 0; cg0 < ncg; cg0 += per_pass) {
        const int cg = cg0 + wave * lanes_cg + (lane % lanes_cg);
        const int c = c_first + QRB_NC * cg;
        bool on[QRB_NC];
#pragma unroll
        for (int ci = 0; ci < QRB_NC; ++ci) on[ci] = cg < ncg && c + ci < c_end;
        double w[QRB_NC][QRB_NB];
#pragma unroll
        for (int ci = 0; ci < QRB_NC; ++ci)
#pragma unroll
            for (int k = 0; k < QRB_NB; ++k) w[ci][k] = 0.0;
        double *col = B + (int64_t)j0 * n_cols + (on[0] ? c : c_first);
        for (int r0 = first + slice; r0 < RR; r0 += QRB_AHEAD * S) {
            double x[QRB_AHEAD][QRB_NC];
#pragma unroll
            for (int qq = 0; qq < QRB_AHEAD; ++qq) {
                const int r = r0 + qq * S;
#pragma unroll
                for (int ci = 0; ci < QRB_NC; ++ci) x[qq][ci] = (on[ci] && r < RR) ? col[(int64_t)r * n_cols + ci] : 0.0;
            }
#pragma unroll
            for (int qq = 0; qq < QRB_AHEAD; ++qq) {
                const double *u = Ur + min(r0 + qq * S, RR - 1) * QRB_PITCH;
#pragma unroll
                for (int k = 0; k < QRB_NB; ++k) {
                    const double uk = u[k];
#pragma unroll
                    for (int ci = 0; ci < QRB_NC; ++ci) w[ci][k] = __builtin_fma(uk, x[qq][ci], w[ci][k]);
                }
            }
        }
        for (int off = lanes_cg; off < WAVE; off <<= 1) {
#pragma unroll
            for (int ci = 0; ci < QRB_NC; ++ci)
#pragma unroll
                for (int k = 0; k < QRB_NB; ++k) w[ci][k] += __shfl_xor(w[ci][k], off);
        }
        for (int r0 = slice; r0 < RR; r0 += QRB_AHEAD * S) {
            double x[QRB_AHEAD][QRB_NC];
#pragma unroll
            for (int qq = 0; qq < QRB_AHEAD; ++qq) {
                const int r = r0 + qq * S;
                const bool ld_ = r < RR && !(VIRT == 1 && r < nbk);
#pragma unroll
                for (int ci = 0; ci < QRB_NC; ++ci) x[qq][ci] = (on[ci] && ld_) ? col[(int64_t)r * n_cols + ci] : 0.0;
            }
#pragma unroll
            for (int qq = 0; qq < QRB_AHEAD; ++qq) {
                const int r = r0 + qq * S;
                const double *v = Vr + min(r, RR - 1) * QRB_PITCH;
#pragma unroll
                for (int k = 0; k < QRB_NB; ++k) {
                    const double vk = -v[k];
#pragma unroll
                    for (int ci = 0; ci < QRB_NC; ++ci) x[qq][ci] = __builtin_fma(vk, w[ci][k], x[qq][ci]);
                }
#pragma unroll
                for (int ci = 0; ci < QRB_NC; ++ci) if (on[ci] && r < RR) col[(int64_t)r * n_cols + ci] = x[qq][ci];
            }
        }
    }
}

// U = V T (TRANS_T: X - V T^T V^T X = X - V (V T)^T X) or V T^T, row-major like V; T upper triangular
template <bool TRANS_T>
__device__ __forceinline__ void qrb_form_u(const double *__restrict__ Vr, const double *__restrict__ T, double *__restrict__ Ur, int RR)
{
    for (int e = threadIdx.x; e < RR * QRB_NB; e += QRB_THREADS) {
        const int r = e / QRB_NB, k = e % QRB_NB;
        double acc = 0.0;
        if (TRANS_T) { for (int mm = 0; mm <= k; ++mm) acc = __builtin_fma(Vr[r * QRB_PITCH + mm], T[mm * QRB_NB + k], acc); }
        else { for (int mm = k; mm < QRB_NB; ++mm) acc = __builtin_fma(Vr[r * QRB_PITCH + mm], T[k * QRB_NB + mm], acc); }
        Ur[r * QRB_PITCH + k] = acc;
    }
}

__global__ void __launch_bounds__(QRB_THREADS) batched_qr_blocked_kernel(int n_rows, int n_cols, int64_t n_traj, int64_t ld,
                                                                         double *__restrict__ a, double *__restrict__ rdiag,
                                                                         double *__restrict__ scratch, double *__restrict__ t_store, int skip)
{
    // (skip & 8: the matrix is in `scratch` already and stays there -- the launcher's transposes; 1, 2, 4: timing experiments of a
    // developer build: no trailing update / no second phase / no panel factorisation, the results are then wrong)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Vr = (double *)smem;                           // panel, [row - j0][QRB_PITCH]
    double *Ur = Vr + (size_t)n_rows * QRB_PITCH;          // V op(T), same layout
    __shared__ __attribute__((aligned(16))) double T[QRB_NB * QRB_NB];
    __shared__ double G[QRB_NB * QRB_NB];
    __shared__ double taus[QRB_NB];
    const int64_t m = qr_member(blockIdx.x, n_traj);
    if (m >= n_traj) return;                               // (the whole workgroup)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = QRB_THREADS / WAVE;
    double *B = scratch + (int64_t)m * n_rows * n_cols;
    const int n_panels = (n_cols + QRB_NB - 1) / QRB_NB;
    double *Tg = t_store + (int64_t)m * n_panels * (QRB_NB * QRB_NB);
    if (!(skip & 8)) for (int64_t e = tid; e < (int64_t)n_rows * n_cols; e += QRB_THREADS) B[e] = a[e * ld + m];
    __syncthreads();
    for (int p = 0; p < n_panels; ++p) {                   // ---- dgeqrf
        const int j0 = p * QRB_NB, nbk = min(QRB_NB, n_cols - j0), RR = n_rows - j0;
        for (int e = tid; e < RR * QRB_NB; e += QRB_THREADS) {
            const int r = e / QRB_NB, k = e % QRB_NB;
            Vr[r * QRB_PITCH + k] = k < nbk ? B[(int64_t)(j0 + r) * n_cols + j0 + k] : 0.0;
        }
        T[tid] = 0.0;                                      // (QRB_THREADS == QRB_NB * QRB_NB)
        __syncthreads();
        for (int jj = 0; jj < ((skip & 4) ? 0 : nbk); ++jj) {                 // dgeqr2 on the panel
            double p2 = 0.0;
            for (int r = jj + 1 + lane; r < RR; r += WAVE) { const double x = Vr[r * QRB_PITCH + jj]; p2 = __builtin_fma(x, x, p2); }
            const double xn2 = wave_sum(p2);
            const double alpha = Vr[jj * QRB_PITCH + jj];
            double t = 0.0, beta = alpha, scale = 0.0;
            if (xn2 != 0.0) { beta = -copysign(sqrt(__builtin_fma(alpha, alpha, xn2)), alpha); t = (beta - alpha) / beta; scale = 1.0 / (alpha - beta); }
            for (int kc = jj + 1 + wave; kc < nbk; kc += NW) {
                double d = 0.0;
                for (int r = jj + 1 + lane; r < RR; r += WAVE) d = __builtin_fma(Vr[r * QRB_PITCH + jj], Vr[r * QRB_PITCH + kc], d);
                const double wv = t * __builtin_fma(scale, wave_sum(d), Vr[jj * QRB_PITCH + kc]);
                const double nw = -(wv * scale);
                for (int r = jj + 1 + lane; r < RR; r += WAVE) Vr[r * QRB_PITCH + kc] = __builtin_fma(nw, Vr[r * QRB_PITCH + jj], Vr[r * QRB_PITCH + kc]);
                if (lane == 0) Vr[jj * QRB_PITCH + kc] -= wv;
            }
            __syncthreads();                               // every read of the unscaled column jj is done
            if (wave == jj % NW) {
                for (int r = jj + 1 + lane; r < RR; r += WAVE) Vr[r * QRB_PITCH + jj] *= scale;
                if (lane == 0) {
                    Vr[jj * QRB_PITCH + jj] = beta;
                    taus[jj] = t;
                    rdiag[(int64_t)(j0 + jj) * ld + m] = beta;
                }
            }
        }
        __syncthreads();
        // the panel goes back (V below the diagonal, R on and above it); in LDS it becomes V itself: unit diagonal, zeros above
        for (int e = tid; e < RR * nbk; e += QRB_THREADS) {
            const int r = e / nbk, k = e % nbk;
            B[(int64_t)(j0 + r) * n_cols + j0 + k] = Vr[r * QRB_PITCH + k];
        }
        __syncthreads();
        if (tid < QRB_NB * QRB_NB) {
            const int r = tid / QRB_NB, k = tid % QRB_NB;
            if (r < RR && k >= r) Vr[r * QRB_PITCH + k] = (k == r && k < nbk) ? 1.0 : 0.0;
        }
        __syncthreads();
        // G[k][i] = v_k . v_i (k < i), pairs dealt to the wavefronts; then T (dlarft) row by row: thread k keeps row k
        for (int pr = wave; pr < QRB_NB * QRB_NB; pr += NW) {
            const int k = pr / QRB_NB, i = pr % QRB_NB;
            if (k >= i || i >= nbk) continue;
            double d = 0.0;
            for (int r = i + lane; r < RR; r += WAVE) d = __builtin_fma(Vr[r * QRB_PITCH + k], Vr[r * QRB_PITCH + i], d);
            d = wave_sum(d);
            if (lane == 0) G[k * QRB_NB + i] = d;
        }
        __syncthreads();
        if (tid < nbk) {
            const int k = tid;
            T[k * QRB_NB + k] = taus[k];
            for (int i = k + 1; i < nbk; ++i) {
                double acc = 0.0;
                for (int mm = k; mm < i; ++mm) acc = __builtin_fma(T[k * QRB_NB + mm], G[mm * QRB_NB + i], acc);
                T[k * QRB_NB + i] = -taus[i] * acc;
            }
        }
        __syncthreads();
        if (j0 + nbk < n_cols && !(skip & 1)) {
            qrb_form_u<true>(Vr, T, Ur, RR);
            __syncthreads();
            qrb_apply<0>(B, Vr, Ur, j0, nbk, j0 + nbk, n_cols, n_rows, n_cols);
        }
        // (kept for the second phase)
        Tg[(int64_t)p * (QRB_NB * QRB_NB) + tid] = T[tid];
        __syncthreads();
    }
    for (int p = (skip & 2) ? -1 : n_panels - 1; p >= 0; --p) {              // ---- dorgqr
        const int j0 = p * QRB_NB, nbk = min(QRB_NB, n_cols - j0), RR = n_rows - j0;
        for (int e = tid; e < RR * QRB_NB; e += QRB_THREADS) {
            const int r = e / QRB_NB, k = e % QRB_NB;
            Vr[r * QRB_PITCH + k] = (k < nbk && r > k) ? B[(int64_t)(j0 + r) * n_cols + j0 + k] : ((k < nbk && r == k) ? 1.0 : 0.0);
        }
        T[tid] = Tg[(int64_t)p * (QRB_NB * QRB_NB) + tid];
        __syncthreads();
        qrb_form_u<false>(Vr, T, Ur, RR);
        __syncthreads();
        qrb_apply<1>(B, Vr, Ur, j0, nbk, j0 + nbk, n_cols, n_rows, n_cols);
        // the panel's own columns: (I - V T V^T) e_c = e_c - V (row c of V T^T)^T; thread = (column c of the panel, rows sl, sl + 16, ...)
        {
            const int c = tid % QRB_NB, sl = tid / QRB_NB;
            if (c < nbk) {
                double w[QRB_NB];
#pragma unroll
                for (int k = 0; k < QRB_NB; ++k) w[k] = Ur[c * QRB_PITCH + k];
                for (int r = sl; r < RR; r += QRB_THREADS / QRB_NB) {
                    double x = (r == c) ? 1.0 : 0.0;
#pragma unroll
                    for (int k = 0; k < QRB_NB; ++k) x = __builtin_fma(-Vr[r * QRB_PITCH + k], w[k], x);
                    B[(int64_t)(j0 + r) * n_cols + j0 + c] = x;
                }
                for (int r = sl; r < j0; r += QRB_THREADS / QRB_NB) B[(int64_t)r * n_cols + j0 + c] = 0.0;       // (entries of R above the panel)
            }
        }
        __syncthreads();
    }
    if (!(skip & 8)) for (int64_t e = tid; e < (int64_t)n_rows * n_cols; e += QRB_THREADS) a[e * ld + m] = B[e];
}

// rows the blocked kernel holds a panel of in its LDS
constexpr int QRB_MAX_ROWS = 400;

// Returns the name of the kernel that was launched.  The blocked kernel needs 2 * 8 * 18 * n_rows bytes of dynamic LDS next to 4.5 KB of
// static LDS (70 KB at 228 rows, 119 KB at 400): where the device does not grant that (the attribute call fails, or the device's
// opt-in limit is smaller -- any GPU with 64 KB of LDS), the unblocked kernel (8 * n_rows bytes) takes every shape, as before round 5.
const char *launch_batched_qr_global(int n_rows, int n_cols, int64_t n_traj, int64_t ld, double *a, double *rdiag, double *scratch,
                                     hipStream_t st)
{
    constexpr size_t QRB_STATIC_LDS = 4608;
    bool blocked = n_rows <= QRB_MAX_ROWS;
    const size_t lds = 2 * sizeof(double) * (size_t)n_rows * QRB_PITCH;
    if (blocked) {
        static DynLdsLimit configured(64 * 1024 - QRB_STATIC_LDS);          // per device, see launch_tiled
        if (configured.needs(lds)) {
            int dev = 0, optin = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, dev) != hipSuccess) optin = 0;
            if (optin > 0 && lds + QRB_STATIC_LDS > (size_t)optin) blocked = false;
            else if (hipFuncSetAttribute((const void *)batched_qr_blocked_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
                configured.set(lds);
            else {
                (void)hipGetLastError();                                    // the refusal is handled here, not reported by a later call
                blocked = false;
            }
        }
    }
#ifdef QGS_HIP_DEV_KNOBS
    if (const char *e = std::getenv("QGS_HIP_QR_UNBLOCKED")) if (*e == '1') blocked = false;
#endif
    if (blocked) {
        double *t_store = scratch + (size_t)n_traj * n_rows * n_cols;
        int skip = 0;
#ifdef QGS_HIP_DEV_KNOBS
        if (const char *e = std::getenv("QGS_HIP_QRB_SKIP")) skip = std::atoi(e);
#endif
        // A[row][col][member] <-> the members' contiguous copies by the tile transposes (whole lines on both sides: 0.3 ms each way for
        // 1 024 x 228 x 228 instead of 0.7 ms of 8-byte accesses inside the kernel)
        const int64_t n_inner = (int64_t)n_rows * n_cols;
        hipLaunchKernelGGL(unpack_kernel, dim3(blocks_for(n_traj, TILE), blocks_for(n_inner, TILE)), dim3(256), 0, st, n_inner, n_traj, ld, a, scratch);
        hipLaunchKernelGGL(batched_qr_blocked_kernel, dim3((unsigned)(8 * ((n_traj + 7) / 8))), dim3(QRB_THREADS), lds, st, n_rows, n_cols,
                           n_traj, ld, a, rdiag, scratch, t_store, skip | 8);
        hipLaunchKernelGGL(pack_kernel, dim3(blocks_for(n_traj, TILE), blocks_for(n_inner, TILE)), dim3(256), 0, st, n_inner, n_traj, ld, scratch, a);
        return "batched_qr_blocked_kernel";
    }
    double *taus = scratch + (size_t)n_traj * n_rows * n_cols;
    hipLaunchKernelGGL(batched_qr_global_kernel, dim3((unsigned)(8 * ((n_traj + 7) / 8))), dim3(QRG_THREADS), sizeof(double) * (size_t)n_rows, st,
                       n_rows, n_cols, n_traj, ld, a, rdiag, scratch, taus);
    return "batched_qr_global_kernel";
}

void launch_batched_qr(int n_rows, int n_cols, int64_t n_traj, int64_t ld, double *a, double *rdiag, hipStream_t st)
{
    const size_t lds = sizeof(double) * (size_t)n_rows * QR_STRIDE;
    static DynLdsLimit configured(64 * 1024);          // per device, see launch_tiled
    if (configured.needs(lds)) {
        if (hipFuncSetAttribute((const void *)batched_qr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
            configured.set(lds);
    }
    hipLaunchKernelGGL(batched_qr_kernel, dim3((unsigned)(8 * ((n_traj + 7) / 8))), dim3(WAVE), lds, st, n_rows, n_cols, n_traj, ld, a, rdiag);
}

// ---- small dense algebra of the covariant Lyapunov vectors, one matrix per member, layout M[row][col][member] -----------------
// C = A B (or A^T B): one lane per member, a strip of MM_TC columns of one row of C per thread; every access is coalesced over
// the members.  triangular 1: only the upper triangle of C is formed, the rest is written as zero (R = Q^T A of a QR step);
// triangular 2: B is upper triangular, the sum over k stops at the column index (V = Q a).
constexpr int MM_TC = 4;

__global__ void __launch_bounds__(64) batched_matmul_kernel(int n_rows, int n_inner, int n_cols, int trans_a, int triangular, int64_t n_traj,
                                                            int64_t ld, const double *__restrict__ a, const double *__restrict__ b,
                                                            double *__restrict__ c)
{
    const int64_t m = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (m >= n_traj) return;
    const int r = (int)blockIdx.y, c0 = (int)blockIdx.z * MM_TC;
    double acc[MM_TC];
#pragma unroll
    for (int q = 0; q < MM_TC; ++q) acc[q] = 0.0;
    const bool skip = triangular == 1 && c0 + MM_TC - 1 < r;               // the whole strip lies below the diagonal
    if (!skip) {
        const int k_end = triangular == 2 ? min(n_inner, c0 + MM_TC) : n_inner;
        for (int k = 0; k < k_end; ++k) {
            const double av = trans_a ? a[((int64_t)k * n_rows + r) * ld + m] : a[((int64_t)r * n_inner + k) * ld + m];
#pragma unroll
            for (int q = 0; q < MM_TC; ++q) {
                const int cc = c0 + q;
                if (cc < n_cols && !(triangular == 2 && k > cc)) acc[q] = __builtin_fma(av, b[((int64_t)k * n_cols + cc) * ld + m], acc[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < MM_TC; ++q) {
        const int cc = c0 + q;
        if (cc < n_cols) c[((int64_t)r * n_cols + cc) * ld + m] = (triangular == 1 && cc < r) ? 0.0 : acc[q];
    }
}

void launch_batched_matmul(int n_rows, int n_inner, int n_cols, int trans_a, int triangular, int64_t n_traj, int64_t ld, const double *a,
                           const double *b, double *c, hipStream_t st)
{
    hipLaunchKernelGGL(batched_matmul_kernel, dim3(blocks_for(n_traj, 64), (unsigned)n_rows, blocks_for(n_cols, MM_TC)), dim3(64), 0, st,
                       n_rows, n_inner, n_cols, trans_a, triangular, n_traj, ld, a, b, c);
}

// One backward step of the Ginelli recursion (qgs/toolbox/lyapunov.py:1252-1283): column c of a_out = the solution x of
// R x = a_in[:, c] (R upper triangular: back substitution over the leading (c+1) x (c+1) block, what solve_triangular_matrix,
// util.py:78-98, asks of np.linalg.solve), plus noise[c] * pert on its diagonal entry, scaled to unit 2-norm; norm[c] is the
// norm before scaling (normalize_matrix_columns, util.py:56-75).  One lane per (member, column); the partial solution lives in
// a_out itself (a thread re-reads only what it wrote).
__global__ void __launch_bounds__(64) clv_backstep_kernel(int nv, int64_t n_traj, int64_t ld, const double *__restrict__ rm,
                                                          const double *__restrict__ a_in, double *a_out, double *__restrict__ norm,
                                                          const double *__restrict__ noise, double pert)
{
    const int64_t m = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (m >= n_traj) return;
    const int c = (int)blockIdx.y;
    for (int i = c; i >= 0; --i) {
        double s = a_in[((int64_t)i * nv + c) * ld + m];
        for (int k = i + 1; k <= c; ++k) s = __builtin_fma(-rm[((int64_t)i * nv + k) * ld + m], a_out[((int64_t)k * nv + c) * ld + m], s);
        a_out[((int64_t)i * nv + c) * ld + m] = s / rm[((int64_t)i * nv + i) * ld + m];
    }
    if (noise) a_out[((int64_t)c * nv + c) * ld + m] += noise[(int64_t)c * ld + m] * pert;
    double ss = 0.0;
    for (int i = 0; i <= c; ++i) {
        const double x = a_out[((int64_t)i * nv + c) * ld + m];
        ss = __builtin_fma(x, x, ss);
    }
    const double nrm = sqrt(ss);
    for (int i = 0; i <= c; ++i) a_out[((int64_t)i * nv + c) * ld + m] /= nrm;
    for (int i = c + 1; i < nv; ++i) a_out[((int64_t)i * nv + c) * ld + m] = 0.0;
    norm[(int64_t)c * ld + m] = nrm;
}

void launch_clv_backstep(int nv, int64_t n_traj, int64_t ld, const double *rm, const double *a_in, double *a_out, double *norm,
                         const double *noise, double pert, hipStream_t st)
{
    hipLaunchKernelGGL(clv_backstep_kernel, dim3(blocks_for(n_traj, 64), (unsigned)nv), dim3(64), 0, st, nv, n_traj, ld, rm, a_in, a_out,
                       norm, noise, pert);
}

bool tiled_supported(int ndim) { return ndim <= 16 * TILED_NW; }

hipError_t launch_gen_rk_tiled(const TiledTensor &T, const RkArgs &p, const double *y_in, double *y_out, double *rec,
                               double *stages, const double *dtime, const double *tab_spec, hipStream_t st)
{
    const int rpw = T.rpw;
#define QGS_TILED_CASE(R)                                                                                   \
    if (rpw == R) return T.terms_per_trip == 16 ? launch_tiled<R, 16>(T, p, y_in, y_out, rec, stages, dtime, tab_spec, st) \
                                                : launch_tiled<R, 4>(T, p, y_in, y_out, rec, stages, dtime, tab_spec, st);
    QGS_TILED_CASE(2)
    QGS_TILED_CASE(4)
    QGS_TILED_CASE(8)
    QGS_TILED_CASE(16)
#undef QGS_TILED_CASE
    return hipErrorInvalidValue;
}

void launch_pack_states(int ndim, int64_t n_traj, int64_t ld, const double *rows, double *modes, hipStream_t st)
{
    hipLaunchKernelGGL(pack_kernel, dim3(blocks_for(n_traj, TILE), blocks_for(ndim, TILE)), dim3(256), 0, st,
                       (int64_t)ndim, n_traj, ld, rows, modes);
}

void launch_unpack_states(int ndim, int64_t n_traj, int64_t ld, const double *modes, double *rows, hipStream_t st)
{
    hipLaunchKernelGGL(unpack_kernel, dim3(blocks_for(n_traj, TILE), blocks_for(ndim, TILE)), dim3(256), 0, st,
                       (int64_t)ndim, n_traj, ld, modes, rows);
}

void launch_unpack_records(int64_t n_inner, int64_t n_traj, int64_t ld, int64_t n_records, const double *in, double *out,
                           hipStream_t st)
{
    if (n_records == 1) {   // plain 2-D transpose
        hipLaunchKernelGGL(unpack_kernel, dim3(blocks_for(n_traj, TILE), blocks_for(n_inner, TILE)), dim3(256), 0, st,
                           n_inner, n_traj, ld, in, out);
        return;
    }
    // the whole record as one window of the tile kernel: 3.5 TB/s moved at config-2 size (101 records of 65 536 x 36), where one
    // thread per (member, inner) writing its run of records 8 bytes at a time reached 0.94 (tools/unpack_ab.py)
    launch_unpack_window(n_inner, n_traj, ld, n_records, n_records, in, out, st);
}

void launch_unpack_window(int64_t n_inner, int64_t n_traj, int64_t ld, int64_t W, int64_t out_stride, const double *in,
                          double *out, hipStream_t st)
{
    const int64_t cb = (n_inner * W + TILE - 1) / TILE;
    hipLaunchKernelGGL(unpack_window_kernel, dim3(blocks_for(n_traj, TILE), (unsigned)std::min<int64_t>(cb, 65535)), dim3(256), 0, st,
                       n_inner, n_traj, ld, W, out_stride, in, out);
}

// local Lyapunov exponents of one Benettin interval: out = log|diag R| / dt  (qgs/toolbox/lyapunov.py:531, 611)
__global__ void __launch_bounds__(256) local_exponents_kernel(int64_t n, const double *__restrict__ rdiag, double dt, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = log(fabs(rdiag[i])) / dt;
}

void launch_local_exponents(int64_t n, const double *rdiag, double dt, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(local_exponents_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, n, rdiag, dt, out);
}

void launch_pack_tangent(int ndim, int64_t n_tg, int64_t n_traj, int64_t ld, const double *rows, double *modes, hipStream_t st)
{
    const int64_t n_inner = (int64_t)ndim * n_tg;
    hipLaunchKernelGGL(pack_kernel, dim3(blocks_for(n_traj, TILE), blocks_for(n_inner, TILE)), dim3(256), 0, st, n_inner,
                       n_traj, ld, rows, modes);
}

}  // namespace qgs
