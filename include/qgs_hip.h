/*
 * qgs_hip.h -- C-ABI of libqgs_hip.so: the MI355X (gfx950) implementation of the qgs
 * ensemble spectral-tendency + Runge-Kutta hot path.
 *
 * The reference (Climdyn/qgs) is pure Python and has no FFI of its own for this path; its
 * "native" code is what numba compiles from the functions cited below.  Each entry point
 * here replaces one of those functions for a whole ensemble at once, and is what a ctypes
 * binding inside the reference would call.  Plain pointers and sizes
 * only; no torch / numpy types.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; qgs_last_error() gives the message
 *     (thread-local).
 *   - "host layout"  = the reference's NumPy layout, C-contiguous:
 *         states      (n_traj, ndim)
 *         trajectory  (n_traj, ndim, n_records)
 *         tangent     (n_traj, ndim, n_tg [, n_records])
 *   - "device layout" = mode-major ensemble, one member per wavefront lane:
 *         states      X[mode][member]            element (d, m) at  d*ld + m
 *         trajectory  R[record][mode][member]    element (r, d, m) at (r*ndim + d)*ld + m
 *         tangent     F[record][mode][col][member]
 *     `ld` (leading dimension, in members) is >= n_traj and a multiple of 64.
 *   - tensor coordinates are the reference's: (i, j, k) with index 0 the constant slot
 *     (eta_0 = 1), rows i in 1..ndim  (qgs/tensors/qgtensor.py:19-65).
 *   - all floating point data is IEEE-754 binary64.
 */
#ifndef QGS_HIP_H
#define QGS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qgs_model qgs_model;   /* opaque: tensors staged on one device + compiled kernels */
typedef struct qgs_group qgs_group;   /* opaque: one qgs_model per GPU of a device list; members are sharded over them */

/* Error text of the last failing call on this thread ("" if none). */
const char *qgs_last_error(void);

/* Number of visible HIP devices and the gcnArchName of device 0 (e.g. "gfx950:sramecc+:xnack-").
 * Fails (<0) when no GPU is visible: there is no CPU fallback in this library. */
int qgs_backend_info(int *n_devices, char *arch_buf, int buflen);

/* Stage a model's tensors on `device` and build its kernels.
 *   coo/val   : qgs/functions/tendencies.py:92-93   coo = tensor.coords.T (nnz,3), val = tensor.data
 *   jcoo/jval : qgs/functions/tendencies.py:95-96   jacobian_tensor ditto (may be NULL/0: then
 *               qgs_jacobian and qgs_rk_tgls_integrate are unavailable)
 * Replaces the closure capture of create_tendencies (tendencies.py:111-121).
 * As in the reference (numba compiles sparse_mul3 once; `val` is a run-time operand, sparse_mul.py:48-81) the VALUES are data:
 * the specialised kernels are generated from, and cached under, the STRUCTURE of the tensor (coordinates, which coefficients
 * share a magnitude -- to within 2 ulp --, signs); every model stores its own coefficients into the tables of the module it
 * loads.  Models that differ in parameter values share their code objects. */
int qgs_model_create(int device, int ndim,
                     int64_t nnz, const int32_t *coo, const double *val,
                     int64_t jnnz, const int32_t *jcoo, const double *jval,
                     qgs_model **out);
/* Same for a tensor of rank 3 or 5: coo is (nnz, rank), jcoo (jnnz, rank).  Rank 5 is the tensor of the dynamic-
 * temperature and T^4 models (QgsTensorDynamicT / QgsTensorT4, qgs/tensors/qgtensor.py:843-1363), whose closures
 * contract it with sparse_mul5 / sparse_mul4 (qgs/functions/tendencies.py:98-109, sparse_mul.py:84-158):
 *     f_i = sum T_ijklm x_j x_k x_l x_m ,   Df_ij = sum Tj_ijklm x_k x_l x_m     (index 0 = the constant slot).
 * Every other entry point works on such a model unchanged. */
int qgs_model_create_rank(int device, int ndim, int rank,
                          int64_t nnz, const int32_t *coo, const double *val,
                          int64_t jnnz, const int32_t *jcoo, const double *jval,
                          qgs_model **out);
int qgs_model_destroy(qgs_model *m);

/* Model properties: which=0 ndim, 1 nnz, 2 jnnz, 3 device, 4 specialised-kernel available (0/1), 5 tensor rank,
 * 6 / 7 number of derived monomials of the tendencies / Jacobian code (rank 5), 8 number of record windows the last
 * host-layout integration of this model was cut into, 9 member groups of that integration, 10 / 11 terms of the tendencies /
 * Jacobian polynomial in the bilinear form the generated code evaluates (rank 3: the tensor entries of rows >= 1; rank 5: after
 * the products shared between monomials have become derived monomials). */
int64_t qgs_model_info(const qgs_model *m, int which);

/* Select the kernel family: 0 = automatic (specialised when available, else generic),
 * 1 = force generic (tensor streamed from memory, any ndim), 2 = force specialised.
 * Specialised = code generated from the tensor and compiled at run time (cached on disk): register-resident
 * kernels up to 64 variables; beyond that (stage state <= 152 KB of LDS, i.e. ndim <= 304) the LDS-resident
 * stepper serves the trajectory integrations and f, and (ndim <= 243) LDS-resident tangent / adjoint kernels the
 * tangent pass; Df stays generic.  In automatic mode the LDS-resident kernels are used when their code objects are
 * already cached (the cache that ships with the tree, or a previous run) or when one call is long enough to pay for the
 * compilation: one to four seconds for the rank-3 stepper and tangent / adjoint kernels (hand-scheduled stage body, from 5e11
 * term-stages per call), 10 - 40 s for the rank-5 and general-tableau flavours (from 2e12).  bench.py's cold_start entries
 * quantify each case.
 * The call also re-reads the kernel-selection knobs of INTEGRATION.md from the environment (they are read
 * when a model is created and here, never inside a launch). */
int qgs_model_set_kernel(qgs_model *m, int kind);

/* ---- host-layout entry points (copy in, run on the GPU, copy out; blocking) --------------
 * Results may be larger than the device memory: the integrations keep only a window of records on the device
 * (QGS_HIP_RECORD_WINDOW_MB, default 8192) and hand window k to the host while window k + 1 is computed.  The result
 * block may be page-locked (qgs_host_register: the device stores into it directly) or pageable (staged copy; a pageable
 * block that needs several windows is page-locked for the duration of the call). */

/* f(t, x) for n_traj states at once.   qgs/functions/tendencies.py:111-115 + sparse_mul.py:48-81 */
int qgs_tendencies(qgs_model *m, int64_t n_traj, const double *x, double *dx);

/* Df(t, x) -> (n_traj, ndim, ndim).     qgs/functions/tendencies.py:117-121 + sparse_mul.py:13-45 */
int qgs_jacobian(qgs_model *m, int64_t n_traj, const double *x, double *jac);

/* Number of records the steppers produce.  qgs/integrators/integrate.py:190-196 */
int64_t qgs_n_records(const double *time, int64_t n_time, int64_t write_steps);

/* How the host-layout integrations cut a run into record windows of W records (pure host arithmetic; exported so that the plan can
 * be checked without a GPU): window k holds the directed records [out[0], out[1]) -- record iw is written at the top of step
 * iw * write_steps, the last one after the final step (integrate.py:210-221) --, runs the steps [out[2], out[3]), includes the
 * final record iff out[4], and its first record is stored at index out[5] of the (direction-corrected) record axis.
 * Returns the number of windows (> 0), < 0 on bad arguments. */
int qgs_record_window(int64_t n_records, int64_t n_steps, int64_t write_steps, int backward, int64_t W, int64_t k, int64_t *out);

/* _integrate_runge_kutta_jit(f, time, ic, time_direction, write_steps, b, c, a)
 * qgs/integrators/integrate.py:182-223.  `time` is the undirected grid (the function reverses it
 * for time_direction == -1 exactly like :199-202) and `traj` comes back direction-corrected (:223).
 * traj must hold n_traj*ndim*qgs_n_records(...) doubles. */
int qgs_rk_integrate(qgs_model *m, int64_t n_traj, const double *ic,
                     const double *time, int64_t n_time, int time_direction, int64_t write_steps,
                     int s, const double *b, const double *c, const double *a,
                     double *traj);

/* _integrate_runge_kutta_tgls_jit(f, fjac, time, ic, tg_ic, time_direction, write_steps, b, c, a,
 *                                 adjoint, inverse, boundary=_zeros_func)
 * qgs/integrators/integrate.py:555-614.  `inverse` is the +-1.0 multiplier (integrate.py:521-523).
 * Only the zero boundary term (the reference default, :235-237) runs on the device. */
int qgs_rk_tgls_integrate(qgs_model *m, int64_t n_traj, int64_t n_tg,
                          const double *ic, const double *tg_ic,
                          const double *time, int64_t n_time, int time_direction, int64_t write_steps,
                          int s, const double *b, const double *c, const double *a,
                          int adjoint, double inverse,
                          double *traj, double *fmatrix);

/* Same run as qgs_rk_integrate, but only the ensemble mean and variance of every variable at every record come back:
 * mean, var are (ndim, n_records) row-major (var may be null); final_states (n_traj, ndim) may be null.  The record never
 * exists as a whole, on the device or anywhere else: it is integrated and reduced window by window.
 * Replaces integrate() + get_trajectories() + np.mean / np.var over the member axis
 * (qgs/integrators/statistics.py:33-66) without moving the ensemble trajectories to the host. */
int qgs_rk_integrate_moments(qgs_model *m, int64_t n_traj, const double *ic, const double *time, int64_t n_time,
                             int time_direction, int64_t write_steps, int s, const double *b, const double *c,
                             const double *a, double *mean, double *var, double *final_states);

/* All GPUs of the node (or any list of them) behind one handle: what the reference's integrators do with the cores of the
 * machine (one task per trajectory over `num_threads` processes, qgs/integrators/integrator.py:79-82, 133-142, 386-395).
 * `devices` may name a device more than once (two models and two pipelines on it).  Members are split into contiguous
 * shards, shard i = members [start, start + count) with the remainder going to the first shards (qgs_group_shard); every
 * shard is integrated by its own model from a host thread of its own and delivers its slice of the result block itself
 * (G parallel device-to-host streams; no collective).  Arguments and results exactly as for the qgs_model entry points of
 * the same name.  A shard's results are bitwise what a qgs_model returns for those members alone; against one qgs_model call
 * over the whole ensemble they agree to rounding (the library chooses its kernel by the size of the ensemble it is handed).
 * SURVEY 8(b)'s `device_mask`. */
int qgs_group_create(int n_devices, const int *devices, int ndim, int rank,
                     int64_t nnz, const int32_t *coo, const double *val,
                     int64_t jnnz, const int32_t *jcoo, const double *jval, qgs_group **out);
int qgs_group_destroy(qgs_group *g);
int qgs_group_size(const qgs_group *g);
qgs_model *qgs_group_model(qgs_group *g, int i);              /* borrowed: shard i's model (its device: qgs_model_info(m, 3)) */
int qgs_group_shard(const qgs_group *g, int64_t n_traj, int i, int64_t *start, int64_t *count);
int qgs_group_set_kernel(qgs_group *g, int kind);
int qgs_group_tendencies(qgs_group *g, int64_t n_traj, const double *x, double *dx);
int qgs_group_jacobian(qgs_group *g, int64_t n_traj, const double *x, double *jac);
int qgs_group_rk_integrate(qgs_group *g, int64_t n_traj, const double *ic,
                           const double *time, int64_t n_time, int time_direction, int64_t write_steps,
                           int s, const double *b, const double *c, const double *a, double *traj);
int qgs_group_rk_tgls_integrate(qgs_group *g, int64_t n_traj, int64_t n_tg, const double *ic, const double *tg_ic,
                                const double *time, int64_t n_time, int time_direction, int64_t write_steps,
                                int s, const double *b, const double *c, const double *a,
                                int adjoint, double inverse, double *traj, double *fmatrix);

/* The general contraction behind the reference's sparse_mul3 / sparse_mul5 / sparse_mul2 / sparse_mul4 called with ANY vectors
 * (qgs/functions/sparse_mul.py:13-158: `coo`, `val` and the vectors are all run-time arguments; a[0] need not be 1, the
 * vectors may differ, the tensor may hold entries in row 0).  coo: (nnz, rank) row-major, rank 3 or 5, every index in
 * [0, n_slots).  n_out_axes = 1: res[i] = sum val * a[j] * b[k] (* c[l] * d[m])          -- sparse_mul3 / sparse_mul5
 * n_out_axes = 2: res[i][j] = sum val * a[k] (* b[l] * c[m])                             -- sparse_mul2 / sparse_mul4
 * `vecs`: the rank - n_out_axes vectors, n_slots doubles each, one after the other; `res`: n_slots (or n_slots^2) doubles,
 * zero where the tensor has no entry (the reference's `res[0] = 1` of sparse_mul3 / 5 is the caller's).  Entries are summed
 * per output element in their incoming order with the reference's operation order: bitwise the reference's loops. */
typedef struct qgs_contraction qgs_contraction;
int qgs_contraction_create(int device, int n_slots, int rank, int n_out_axes, int64_t nnz, const int32_t *coo, const double *val,
                           qgs_contraction **out);
int qgs_contraction_apply(qgs_contraction *c, const double *vecs, double *res);
int qgs_contraction_destroy(qgs_contraction *c);

/* Host memory and the GPU (round 5).  The GPU -- kernels and copy engines alike -- only ever touches host memory that this
 * library allocated itself (qgs_host_alloc below, its own bounce blocks) or that the caller handed over explicitly with
 * qgs_host_register.  Every other host pointer an entry point is given (initial conditions, result blocks, tensors) is pageable
 * as far as the library knows: it is read and written by CPU threads only, through page-locked bounce blocks
 * (qgs_amd/csrc/host_bridge.h; 45 GB/s for a 189 GB record with 16 host threads, profiles/r05_big_record.txt), and the runtime
 * is never asked to pin it.
 *
 * qgs_host_register page-locks and maps a caller-owned block so that the unpack kernels store into it directly (51 GB/s for the
 * same record, after 0.04 s per GB of page-locking).  By calling it the caller guarantees, until qgs_host_unregister:
 *   - the block is a mapping of its own (mmap, a page-aligned allocation of whole pages) -- not a piece of an allocator's
 *     heap that shares pages with other objects or that the allocator may remap, trim or hand out again;
 *   - it is not freed, moved (realloc, mremap) or protected (mprotect), and no other thread forks while kernels write to it;
 *   - nothing else registers or unregisters overlapping ranges.
 * Registered heap memory that did not keep these promises is where every GPU write fault of round 4 was found.
 * Replaces nothing in the reference: its results travel between processes through pickling queues
 * (qgs/integrators/integrator.py:388-395). */
int qgs_host_register(void *ptr, int64_t bytes);
/* Blocking copies between device memory of GPU `device` and ANY host memory (pageable: through the bounce blocks; a block of
 * qgs_host_alloc / qgs_host_register: one DMA copy).  Work queued on `stream` (may be NULL) is waited for first.  For host
 * programs that would otherwise hand pageable pointers to hipMemcpy themselves (the Python layer's uploads and downloads). */
int qgs_memcpy_h2d(int device, void *d_dst, const void *h_src, int64_t bytes, void *stream);
int qgs_memcpy_d2h(int device, void *h_dst, const void *d_src, int64_t bytes, void *stream);
/* Page-locked host memory allocated by the runtime itself (portable, mapped): what result blocks should live in when the kernels
 * are to store into them.  Unlike a registered block of the caller's it shares no pages with the C library's heap
 * (every GPU write fault seen in round 4 hit registered heap memory). */
int qgs_host_alloc(int64_t bytes, void **out);
int qgs_host_free(void *ptr);
int qgs_host_unregister(void *ptr);

/* ---- device-layout entry points (pointers are device pointers on the model's device; the work
 *      is enqueued on `stream` (a hipStream_t, NULL = default stream) and NOT synchronised) ------ */

/* qgs_rk_integrate with both blocks in DEVICE memory, in the reference's layouts: d_ic_rows (n_traj, ndim), d_traj_rows
 * (n_traj, ndim, n_records).  Pack, windowed stepper and record unpack run on the model's own streams with only a window
 * of mode-major records as scratch; blocking.  (The shard of a multi-process run whose result is gathered with RCCL,
 * qgs_amd/parallel.py.) */
int qgs_rk_integrate_rows_device(qgs_model *m, int64_t n_traj, const double *d_ic_rows,
                                 const double *time, int64_t n_time, int time_direction, int64_t write_steps,
                                 int s, const double *b, const double *c, const double *a, double *d_traj_rows);

/* (n_traj, ndim) host-layout device buffer  <->  mode-major X[ndim][ld] */
int qgs_pack_states(qgs_model *m, int64_t n_traj, int64_t ld, const double *d_x_rows, double *d_x_modes, void *stream);
int qgs_unpack_states(qgs_model *m, int64_t n_traj, int64_t ld, const double *d_x_modes, double *d_x_rows, void *stream);
/* tangent vectors / matrices: (n_traj, ndim, n_tg) host-layout device buffer -> F[ndim][n_tg][ld] */
int qgs_pack_tangent(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_tg, const double *d_rows, double *d_modes, void *stream);

/* d_out[i] = log|d_rdiag[i]| / dt for i < n (device pointers; in place allowed): the local Lyapunov exponents of one Benettin
 * interval from diag(R) of its QR step, `np.log(np.abs(np.diag(r))) / dt` of qgs/toolbox/lyapunov.py:531, 611 -- on the device, so
 * that the exponents leave in the record windows as they are (the host pass over the finished block took as long as a third of
 * the transfer of a 72 GB record). */
int qgs_local_exponents_device(qgs_model *m, int64_t n, const double *d_rdiag, double dt, double *d_out, void *stream);
/* R[n_records][ndim][ld]  ->  (n_traj, ndim, n_records) */
int qgs_unpack_records(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_inner, int64_t n_records,
                       const double *d_rec_modes, double *d_rec_rows, void *stream);
/* One WINDOW of mode-major records into the record block of the whole run: d_window holds n_window consecutive records
 * R[w][n_inner][ld]; dst is the (n_traj, n_inner, n_records) block in the reference's layout (the host array the reference's
 * loops fill record by record, qgs/toolbox/lyapunov.py:232-358, qgs/integrators/integrate.py:196-223) and receives records
 * [first_record, first_record + n_window).  dst may be device memory or page-locked host memory (qgs_host_register: written
 * by the kernel's own stores) or pageable host memory (staged on the device, then brought over by the bounce ring of
 * qgs_amd/csrc/host_bridge.h: the library never page-locks memory it did not allocate).  Enqueued on `stream`: the window may
 * be overwritten by work enqueued behind it.  With a pageable dst, qgs_unpack_window returns when the records are in dst;
 * qgs_unpack_window_enqueue hands the window to the device's drain thread and returns (the host is only held while the
 * previous window of this model is still being read from its staging block), and qgs_drain_wait blocks until every window
 * enqueued for the model has arrived -- dst must stay valid until then.  For any other dst the two calls are the same. */
int qgs_unpack_window(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_inner, int64_t n_window, int64_t n_records,
                      int64_t first_record, const double *d_window, double *dst, void *stream);
int qgs_unpack_window_enqueue(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_inner, int64_t n_window, int64_t n_records,
                              int64_t first_record, const double *d_window, double *dst, void *stream);
int qgs_drain_wait(qgs_model *m);

int qgs_tendencies_device(qgs_model *m, int64_t n_traj, int64_t ld, const double *d_x, double *d_dx, void *stream);

/* `time` is a HOST pointer (n_time doubles, undirected); it is small and is staged by the library.
 * d_rec: R[n_records][ndim][ld]. */
int qgs_rk_integrate_device(qgs_model *m, int64_t n_traj, int64_t ld, const double *d_ic,
                            const double *time, int64_t n_time, int time_direction, int64_t write_steps,
                            int s, const double *b, const double *c, const double *a,
                            double *d_rec, void *stream);

/* d_tg_ic: F[ndim][n_tg][ld];  d_rec_fm: F[n_records][ndim][n_tg][ld]. */
int qgs_rk_tgls_integrate_device(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_tg,
                                 const double *d_ic, const double *d_tg_ic,
                                 const double *time, int64_t n_time, int time_direction, int64_t write_steps,
                                 int s, const double *b, const double *c, const double *a,
                                 int adjoint, double inverse,
                                 double *d_rec, double *d_rec_fm, void *stream);

/* Mean and population variance (may be null) over the members of every row of X[row][member] (n_rows rows of
 * leading dimension ld, e.g. the n_records*ndim rows of a device record): the ensemble averages of
 * TrajectoriesStatistics.compute_stats (qgs/integrators/statistics.py:55-63) for the observables x and x^2,
 * computed where the trajectories are. */
int qgs_ensemble_moments_device(qgs_model *m, int64_t n_traj, int64_t ld, int64_t n_rows, const double *d_x,
                                double *d_mean, double *d_var, void *stream);

/* Batched QR of one (n_rows x n_cols) matrix per member, device layout A[row][col][member]: A is replaced by Q
 * (LAPACK Householder sign convention), d_rdiag[col][member] receives diag(R).  Replaces the per-trajectory
 * `np.linalg.qr` of the Benettin loops, qgs/toolbox/lyapunov.py:540-547, 599-628. */
/* (n_cols <= 64 and n_rows <= 300: a kernel generated and compiled for the shape, the matrices in registers -- three layouts, see
 * qgs_amd/csrc/codegen.h QrPlan; anything larger, e.g. the 228 x 228 bases of a full spectrum: one workgroup per matrix on a scratch
 * copy in global memory, blocked (dgeqrf + dorgqr with 16-column panels) up to 400 rows, column by column beyond.) */
int qgs_batched_qr_device(qgs_model *m, int64_t n_traj, int64_t ld, int n_rows, int n_cols,
                          double *d_a, double *d_rdiag, void *stream);

/* Small dense algebra of the covariant Lyapunov vectors (qgs/toolbox/lyapunov.py:1174-1288, the loops the reference runs per
 * trajectory on NumPy matrices), one matrix per member in the device layout M[row][col][member]:
 *   qgs_batched_matmul_device   C (n_rows x n_cols) = A B, or A^T B with trans_a (A stored n_inner x n_rows).  triangular 1:
 *                               only the upper triangle of a square C is formed, the rest is zero -- R = Q^T A of a QR step,
 *                               the `qr[1]` of lyapunov.py:1222-1247; triangular 2: B is upper triangular (square) -- the
 *                               `tmp_vec[ti] @ am` of :1279.  C must not alias A or B.
 *   qgs_clv_backstep_device     one backward step of the Ginelli recursion (:1255-1273): a_out = R^-1 a_in column by column
 *                               (solve_triangular_matrix, qgs/functions/util.py:78-98), plus d_noise[col][member] * noise_pert
 *                               on the diagonal (d_noise may be null), columns scaled to unit 2-norm
 *                               (normalize_matrix_columns, util.py:56-75); d_norm[col][member] receives the norms. */
int qgs_batched_matmul_device(qgs_model *m, int64_t n_traj, int64_t ld, int n_rows, int n_inner, int n_cols, int trans_a,
                              int triangular, const double *d_a, const double *d_b, double *d_c, void *stream);
int qgs_clv_backstep_device(qgs_model *m, int64_t n_traj, int64_t ld, int n_vec, const double *d_r, const double *d_a_in,
                            double *d_a_out, double *d_norm, const double *d_noise, double noise_pert, void *stream);

/* Name, VGPR/SGPR/LDS/scratch use of the kernel the last *_device call launched (for profiling
 * reports).  Any pointer may be NULL. */
int qgs_last_kernel_info(const qgs_model *m, char *name_buf, int buflen,
                         int *vgprs, int *sgprs, int *lds_bytes, int *scratch_bytes);

/* What the device sustains on independent fp64 FMAs and nothing else (eight chains per lane, eight wavefronts per SIMD), timed over
 * about `target_ms` milliseconds with HIP events on the default stream: *tflops, *elapsed_ms.  A measurement aid of bench.py (the
 * practical ceiling next to the nominal FP64 peak: the board lowers the clock under this load); no counterpart in the reference. */
int qgs_fp64_fma_rate(int device, double target_ms, double *tflops, double *elapsed_ms);

/* Effective shader clock of the LAST generated kernel this model launched (blocks until the device is idle): lane 0 of
 * workgroup 0 notes the shader-clock counter and the constant 100 MHz counter when it starts and when it has issued its last
 * store.  *shader_ghz = shader cycles per nanosecond over that interval, *elapsed_ms = the interval (one workgroup's life: the
 * whole launch for the one-wavefront-per-SIMD steppers).  For the roofline report of bench.py: the FP64 peak scales with the
 * clock the chip actually ran at.  Fails for the generic (non-generated) kernels, which carry no probe.  Pointers may be NULL. */
int qgs_kernel_clock(qgs_model *m, double *shader_ghz, double *elapsed_ms);

/* Compile and cache the specialised kernels of a model WITHOUT touching a device (build hosts have no
 * GPU; the cached code objects travel with the source tree).  stage_counts lists the RK stage counts
 * to pre-build; arch NULL = $QGS_HIP_ARCH or gfx950. */
int qgs_prebuild(int ndim, int64_t nnz, const int32_t *coo, const double *val,
                 int64_t jnnz, const int32_t *jcoo, const double *jval,
                 int n_stage_counts, const int *stage_counts, const char *arch);

int qgs_prebuild_rank(int ndim, int rank, int64_t nnz, const int32_t *coo, const double *val,
                      int64_t jnnz, const int32_t *jcoo, const double *jval,
                      int n_stage_counts, const int *stage_counts, const char *arch);

/* Same for the shape-specialised batched QR kernel of qgs_batched_qr_device (n_cols <= n_rows <= 300, n_cols <= 64). */
int qgs_prebuild_qr(int n_rows, int n_cols, const char *arch);

/* Generated HIP source of that kernel, first line "// plan <signature>" (inspection, and the build-host test that compiles it and
 * checks the instruction stream around every DPP instruction: tests/test_qr_codegen_cpu.py).  Returns the length, or -1; copies at
 * most buflen-1 bytes. */
int64_t qgs_qr_kernel_source(int n_rows, int n_cols, char *buf, int64_t buflen);

/* Generated HIP source of the specialised kernels of this model (debugging / inspection); value-free: the coefficient
 * tables are declared without initialisers.
 * Returns the length; copies at most buflen-1 bytes. */
int64_t qgs_model_kernel_source(const qgs_model *m, char *buf, int64_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* QGS_HIP_H */
