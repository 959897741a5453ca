// generic_kernels.h -- tensor-streaming ("generic") HIP kernels: any ndim, any explicit tableau.
//
// These kernels read the COO tensor from memory at run time (row-grouped, i.e. CSR over the
// output index i) instead of having it compiled in.  They serve models whose state does not fit
// the register file (ndim > QGS_SPEC_MAX_NDIM, e.g. MAOOAM 6x6 with ndim 228), tableaus with a
// dense `a` matrix, and as an independent second implementation in the parity tests.
//
// Layout is the library's device layout: one ensemble member per lane, X[mode][member].
// The tensor entry (j,k,val) of a row is wave-uniform, so it is fetched through the scalar unit /
// broadcast, while x_j, x_k are coalesced 512-byte wavefront loads.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace qgs {

// Row-grouped tensor on the device: entries e in [rowptr[i], rowptr[i+1]) belong to output row i
// (i in 0..ndim, row 0 unused).  idx packs the two remaining coordinates: (a << 16) | b.
// Rank-5 tensors (sparse_mul5 / sparse_mul4, qgs/functions/sparse_mul.py:84-158) carry the two further coordinates
// in idx2 = (c << 16) | d (null for rank 3): an entry then stands for val * x_a * x_b * x_c * x_d.  Only the simple
// one-lane-per-member kernels (gen_tend / gen_jac / gen_rk / gen_tgl) read idx2.
struct DevTensor {
    const int32_t *rowptr;   // ndim + 2
    const uint32_t *idx;     // nnz
    const double *val;       // nnz
    const uint32_t *idx2;    // nnz or null
};

// Derived monomials of a rank-5 model for the kernels that keep the stage state in LDS slots (wavefront-per-trajectory
// kernels): slot `slot[e]` = slot `a[e]` * slot `b[e]`, evaluated after every stage-state update.  The products are
// sorted into levels (level 1 uses variables only, level 2 may use level 1, ...): within a level every thread forms at
// most WAVE_DER_PER products, which it keeps (a, b, slot) in registers for the whole run; a barrier separates the levels.
// n_levels == 0: rank 3.
constexpr int WAVE_DER_LEVELS = 3, WAVE_DER_PER = 2;
struct DerivedChains {
    int n_levels;
    int level_ptr[WAVE_DER_LEVELS + 1];     // products of level l are [level_ptr[l], level_ptr[l + 1])
    const int32_t *a, *b, *slot;
};

struct RkArgs {
    int ndim, s;
    int64_t n_traj, ld;
    int64_t step_begin, step_end, write_steps, n_records;
    int backward, write_final;
};

// f(x) -> dx
void launch_gen_tend(const DevTensor &T, int ndim, int64_t n_traj, int64_t ld, const double *x, double *dx, hipStream_t st);
// Df(x) -> jm[(i-1)*ndim + (j-1)][member]   (output must be zero-filled by the caller)
void launch_gen_jac(const DevTensor &Jt, int ndim, int64_t n_traj, int64_t ld, const double *x, double *jm, hipStream_t st);

// One state (n_traj == 1, the f / Df handed to an ODE solver): x, dx, jm are plain (ndim,) / (ndim, ndim) arrays in a
// page-locked host block; when the grid has finished, `seq` appears in *flag (host memory) -- the host spins on it instead of
// synchronising the stream.  `counter` is a zero-initialised device word.  ndim <= 8190 (LDS copy of x).
// Df works on the Jacobian tensor grouped by output element: pair p = entries [ptr[p], ptr[p + 1]) with idx = k (and idx2 for
// rank 5), lut[(i - 1) * ndim + (j - 1)] = p or -1.
struct OnePairs {
    const int32_t *lut;      // ndim * ndim
    const int32_t *ptr;      // n_pairs + 1
    const uint32_t *idx;     // n_entries: k
    const double *val;       // n_entries
    const uint32_t *idx2;    // n_entries or null
};
void launch_gen_tend_one(const DevTensor &T, int ndim, const double *x, double *dx, unsigned *counter, unsigned long long *flag,
                         unsigned long long seq, hipStream_t st);
void launch_gen_jac_one(const OnePairs &P, int ndim, const double *x, double *jm, unsigned *counter, unsigned long long *flag,
                        unsigned long long seq, hipStream_t st);

// General contraction of a COO tensor with explicit vectors (sparse_mul3 / sparse_mul5 / sparse_mul2 / sparse_mul4 called with ANY
// vectors, qgs/functions/sparse_mul.py:13-158): entries grouped by output element (in their incoming order), n_fac factor
// indices per entry; out[out_index[t]] = sum_e val[e] * prod_f vecs[f][fidx[e * n_fac + f]] with the reference's operation
// order ((a*b)*val, no contraction into FMAs): bitwise the reference's loops.  vecs: n_fac vectors of n_slots doubles.
void launch_contract(int n_out, const int32_t *out_index, const int32_t *ptr, const uint32_t *fidx, const double *val, int n_fac,
                     const double *vecs, int n_slots, double *out, hipStream_t st);

// Explicit s-stage RK with the full `a` matrix (integrate.py:204-221).
//   work: (s + 2) * ndim * ld doubles of scratch;  stages: optional S[(step-step_begin)*s+stage][mode][member]
//   tab_full: device array  b[s], a[s*s]
void launch_gen_rk(const DevTensor &T, const RkArgs &p, const double *y_in, double *y_out, double *rec, double *stages,
                   double *work, const double *dtime, const double *tab_full, hipStream_t st);

// Tangent / adjoint propagation along stored stage states, one lane per (member, column)
// (integrate.py:226-231, 593-609).  Jrow is the Jacobian tensor grouped by output row
// (by i for the tangent model, by j for the adjoint), idx = (w index << 16) | x index.
//   work: (s + 2) * ndim * n_tg * ld doubles of scratch
void launch_gen_tgl(const DevTensor &Jrow, const RkArgs &p, int64_t n_tg, double inverse,
                    const double *w_in, double *w_out, double *rec, const double *stages,
                    double *work, const double *dtime, const double *tab_full, hipStream_t st);

// ---- tiled generic stepper ------------------------------------------------------------------------
// Tensor re-laid for the tiled kernel: a flat stream of terms per output row, padded with zero terms to a
// multiple of 4 (small rows) or 16 (long rows) so that one round of wide scalar loads feeds a whole trip:
//     dx_i = sum_terms c * x_j * x_k ,      x_0 = 1 lives in LDS slot 0.
// Offsets are LDS byte offsets of the 64-lane slots (index * 512).
struct TiledTensor {
    const int32_t *row_term;   // ndim + 2 : terms of row i are [row_term[i], row_term[i+1]), both multiples of 4
    const uint32_t *term_joff; // n_terms : j * 512
    const uint32_t *term_koff; // n_terms : k * 512
    const double *term_c;      // n_terms
    int terms_per_trip;        // 4 or 16: rows are padded to a multiple of it
    const int32_t *row_map;    // 16 * rpw : tensor row of (wavefront, slot), 0 = empty; balanced by term count
    int rpw;                   // rows per wavefront the map was built for (2, 4, 8 or 16)
};

// True if the tiled kernel can run this problem (sub-diagonal tableau is checked by the caller).
bool tiled_supported(int ndim);
// One workgroup of 16 wavefronts per 64 members: lanes = members, wavefronts split the rows, the stage
// state lives in LDS ((ndim+1) x 64 doubles), each wavefront keeps y / acc / x_next of its own rows in
// registers.  tab_spec = b[s], a[1][0], a[2][1], ...   Returns hipSuccess or the launch error.
hipError_t launch_gen_rk_tiled(const TiledTensor &T, const RkArgs &p, const double *y_in, double *y_out, double *rec,
                               double *stages, const double *dtime, const double *tab_spec, hipStream_t st);

// ---- wavefront-per-trajectory stepper (small ensembles) --------------------------------------------------
// With fewer members than lanes on the chip (n_traj << 65 536) one-member-per-lane leaves the GPU idle and a
// single trajectory advances at the pace of one in-order wavefront doing ALL rows (7.5 us per RK4 step at
// ndim 36).  Here a trajectory owns a whole workgroup: lane = tensor row, the stage state lives in LDS, each lane
// keeps its row's terms in registers (rows of at most 16 terms) or streams them from memory.
// Tensor = the CSR `DevTensor` (idx = j<<16 | k).  Sub-diagonal tableaus; tab_spec = b[s], a[1][0], a[2][1], ...
// n_slots = ndim + number of derived monomials.  For rank-5 models T is the REDUCED tensor (codegen.h reduce_polynomial:
// two factors per term over the extended index space) and D lists the derived monomials.
bool wave_supported(int n_slots);
hipError_t launch_gen_rk_wave(const DevTensor &T, int max_row_terms, const RkArgs &p, const double *y_in, double *y_out,
                              double *rec, double *stages, const double *dtime, const double *tab_spec, hipStream_t st,
                              const DerivedChains &D = DerivedChains{0, {0, 0, 0, 0}, nullptr, nullptr, nullptr});

// Same idea for the tangent / adjoint model: one workgroup per (member, tangent column), lane = row of J (or J^T),
// the row's entries (w index, x index, value) in registers when there are at most 32, x and w staged in LDS.
// Jrow as for launch_gen_tgl; stages as written by the trajectory pass.
hipError_t launch_gen_tgl_wave(const DevTensor &Jrow, int max_row_terms, const RkArgs &p, int64_t n_tg, double inverse,
                               const double *w_in, double *w_out, double *rec, const double *stages, const double *dtime,
                               const double *tab_spec, hipStream_t st,
                               const DerivedChains &D = DerivedChains{0, {0, 0, 0, 0}, nullptr, nullptr, nullptr});

// Batched Householder QR (LAPACK dgeqr2 + dorg2r conventions: R_jj = -sign(a_jj)*||.||) of one
// (n_rows x n_cols) matrix per member in the device layout A[row][col][member]; A is overwritten by Q,
// rdiag[col][member] receives diag(R).  Used by the Benettin Lyapunov
// estimator (reference: np.linalg.qr in qgs/toolbox/lyapunov.py:540-547, 599-628).
void launch_batched_qr(int n_rows, int n_cols, int64_t n_traj, int64_t ld, double *a, double *rdiag, hipStream_t st);
// C[row][col][member] = A B or A^T B per member (triangular 1: upper triangle of C only, rest zero; 2: B upper triangular)
void launch_batched_matmul(int n_rows, int n_inner, int n_cols, int trans_a, int triangular, int64_t n_traj, int64_t ld, const double *a,
                           const double *b, double *c, hipStream_t st);
// a_out = columns of R^-1 a_in (+ noise * pert on the diagonal) scaled to unit norm, norm[col][member] = the norms
void launch_clv_backstep(int nv, int64_t n_traj, int64_t ld, const double *rm, const double *a_in, double *a_out, double *norm,
                         const double *noise, double pert, hipStream_t st);
// any shape (n_cols > 64 or matrices beyond the LDS): matrix in a global scratch copy, blocked (dgeqrf + dorgqr, 16-column panels) up to
// 400 rows; scratch = n_traj * ((n_rows + 17) * n_cols + 256) doubles
const char *launch_batched_qr_global(int n_rows, int n_cols, int64_t n_traj, int64_t ld, double *a, double *rdiag, double *scratch,
                              hipStream_t st);
// mean / variance over the members of every row of X[row][member]; `part` holds 2 * n_rows * moments_splits() doubles
int moments_splits(int64_t n_rows, int64_t n_traj);
void launch_fma_rate(int blocks, int iters, double *out, hipStream_t st);      // blocks x 256 lanes x iters x 8 independent fp64 FMAs
void launch_moments(int64_t n_rows, int64_t n_traj, int64_t ld, const double *x, double *part, double *mean, double *var,
                    hipStream_t st);

// Layout conversion kernels (host layout <-> device layout), see include/qgs_hip.h
void launch_pack_states(int ndim, int64_t n_traj, int64_t ld, const double *rows, double *modes, hipStream_t st);
void launch_unpack_states(int ndim, int64_t n_traj, int64_t ld, const double *modes, double *rows, hipStream_t st);
// in: R[n_records][n_inner][ld]  ->  out: (n_traj, n_inner, n_records)
void launch_unpack_records(int64_t n_inner, int64_t n_traj, int64_t ld, int64_t n_records, const double *in, double *out,
                           hipStream_t st);
// a window of W records R[W][n_inner][ld] -> out[(m * n_inner + q) * out_stride + r], r in [0, W): `out` points at the
// window's first record column of a (n_traj, n_inner, out_stride) array (device memory or device-accessible host memory)
void launch_unpack_window(int64_t n_inner, int64_t n_traj, int64_t ld, int64_t W, int64_t out_stride, const double *in,
                          double *out, hipStream_t st);
// out[i] = log|rdiag[i]| / dt, i < n: the local Lyapunov exponents of one Benettin interval from the diagonal of its R
void launch_local_exponents(int64_t n, const double *rdiag, double dt, double *out, hipStream_t st);
// tangent IC: host (n_traj, ndim, n_tg) -> F[ndim][n_tg][ld]
void launch_pack_tangent(int ndim, int64_t n_tg, int64_t n_traj, int64_t ld, const double *rows, double *modes, hipStream_t st);
// tangent records: F[n_records][ndim][n_tg][ld] -> (n_traj, ndim, n_tg, n_records): same as unpack_records with
// n_inner = ndim * n_tg.

}  // namespace qgs
