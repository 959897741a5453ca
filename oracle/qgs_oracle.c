/*
 * qgs_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, scalar, CPU restatement of the reference algorithm (Climdyn/qgs) for the
 * ensemble tendencies + Runge-Kutta hot path.  It exists only as the *checker* that the
 * HIP path in qgs_amd/csrc is compared against (tests/, __graft_entry__.smoke(), and the
 * `cpu_baseline` leg of bench.py).  Nothing under qgs_amd/ imports, links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against
 * tests/golden/{rp20,a36,m36,t228}.npz, which were produced by importing the Python
 * reference itself in the build container (tests/golden/make_golden.py): f(x) is bitwise
 * identical (same loop order, built with -ffp-contract=off); trajectories agree to
 * <= 1e-13 relative (the reference's `@` products go through BLAS, whose summation
 * order/FMA use is not specified, so bitwise equality is not claimed there).
 * Caveat on the pin: numba is not installable in that container, so the goldens are the
 * reference's own loops executed by CPython with `numba.njit` replaced by the identity
 * (oracle/refshim/): same statements and IEEE-754 operation order as the jitted code, but
 * not numba-compiled code.  The reference has no stepper fixtures of its own; its tensor
 * fixtures (model_test/<name>.ref, tests/golden/ref/) cross-pin the tensors.
 *
 * Each function cites the reference file:line (relative to the qgs repository root) whose
 * statements it follows.  Operation order is the reference's: (a*b)*val then +=, no FMA.
 *
 * Build:  make -C oracle        (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* qgs/functions/sparse_mul.py:48-81  sparse_mul3
 *   res = zeros_like(vec_a); for n: res[coo[n,0]] += vec_a[coo[n,1]] * vec_b[coo[n,2]] * value[n]; res[0] = 1.
 * coo is (nnz,3) row-major int32; vectors have length n (= ndim+1). */
void oracle_sparse_mul3(int64_t nnz, const int32_t *coo, const double *value,
                        int64_t n, const double *vec_a, const double *vec_b, double *res)
{
    for (int64_t i = 0; i < n; ++i) res[i] = 0.0;
    for (int64_t e = 0; e < nnz; ++e) {
        double prod = vec_a[coo[3 * e + 1]] * vec_b[coo[3 * e + 2]];
        prod = prod * value[e];
        res[coo[3 * e + 0]] += prod;
    }
    res[0] = 1.0;
}

/* qgs/functions/sparse_mul.py:13-45  sparse_mul2
 *   res = zeros((n,n)); for n: res[coo[n,0], coo[n,1]] += vec[coo[n,2]] * value[n] */
void oracle_sparse_mul2(int64_t nnz, const int32_t *coo, const double *value,
                        int64_t n, const double *vec, double *res /* n*n row-major */)
{
    for (int64_t i = 0; i < n * n; ++i) res[i] = 0.0;
    for (int64_t e = 0; e < nnz; ++e)
        res[(int64_t)coo[3 * e + 0] * n + coo[3 * e + 1]] += vec[coo[3 * e + 2]] * value[e];
}

/* qgs/functions/sparse_mul.py:123-158  sparse_mul5
 *   res[coo[n,0]] += vec_a[coo[n,1]] * vec_b[coo[n,2]] * vec_c[coo[n,3]] * vec_d[coo[n,4]] * value[n]; res[0] = 1.
 * coo is (nnz,5) row-major int32 (left-to-right products, like the Python expression). */
void oracle_sparse_mul5(int64_t nnz, const int32_t *coo, const double *value, int64_t n,
                        const double *vec_a, const double *vec_b, const double *vec_c, const double *vec_d, double *res)
{
    for (int64_t i = 0; i < n; ++i) res[i] = 0.0;
    for (int64_t e = 0; e < nnz; ++e) {
        double prod = vec_a[coo[5 * e + 1]] * vec_b[coo[5 * e + 2]];
        prod = prod * vec_c[coo[5 * e + 3]];
        prod = prod * vec_d[coo[5 * e + 4]];
        prod = prod * value[e];
        res[coo[5 * e + 0]] += prod;
    }
    res[0] = 1.0;
}

/* qgs/functions/sparse_mul.py:84-120  sparse_mul4
 *   res[coo[n,0], coo[n,1]] += vec_a[coo[n,2]] * vec_b[coo[n,3]] * vec_c[coo[n,4]] * value[n] */
void oracle_sparse_mul4(int64_t nnz, const int32_t *coo, const double *value, int64_t n,
                        const double *vec_a, const double *vec_b, const double *vec_c, double *res /* n*n row-major */)
{
    for (int64_t i = 0; i < n * n; ++i) res[i] = 0.0;
    for (int64_t e = 0; e < nnz; ++e) {
        double prod = vec_a[coo[5 * e + 2]] * vec_b[coo[5 * e + 3]];
        prod = prod * vec_c[coo[5 * e + 4]];
        prod = prod * value[e];
        res[(int64_t)coo[5 * e + 0] * n + coo[5 * e + 1]] += prod;
    }
}

typedef struct {
    int64_t ndim;
    int64_t nnz;  const int32_t *coo;  const double *val;
    int64_t jnnz; const int32_t *jcoo; const double *jval;
    int rank;     /* 3: QgsTensor (sparse_mul3 / sparse_mul2); 5: QgsTensorDynamicT / QgsTensorT4 (sparse_mul5 / sparse_mul4),
                     qgs/functions/tendencies.py:98-121 */
} oracle_model;

/* qgs/functions/tendencies.py:111-115  f(t, x): xx = concat(([1.], x)); xr = sparse_mul3(coo,val,xx,xx); return xr[1:]
 * work: 2*(ndim+1) doubles */
static void model_f(const oracle_model *m, const double *x, double *out, double *work)
{
    const int64_t n = m->ndim + 1;
    double *xx = work, *xr = work + n;
    xx[0] = 1.0;
    memcpy(xx + 1, x, sizeof(double) * m->ndim);
    if (m->rank == 5) oracle_sparse_mul5(m->nnz, m->coo, m->val, n, xx, xx, xx, xx, xr);   /* tendencies.py:100-103 */
    else oracle_sparse_mul3(m->nnz, m->coo, m->val, n, xx, xx, xr);
    memcpy(out, xr + 1, sizeof(double) * m->ndim);
}

/* qgs/functions/tendencies.py:117-121  Df(t, x): mul_jac = sparse_mul2(jcoo,jval,xx); return mul_jac[1:,1:]
 * work: (ndim+1) + (ndim+1)^2 doubles; out is ndim*ndim row-major */
static void model_Df(const oracle_model *m, const double *x, double *out, double *work)
{
    const int64_t n = m->ndim + 1, nd = m->ndim;
    double *xx = work, *full = work + n;
    xx[0] = 1.0;
    memcpy(xx + 1, x, sizeof(double) * nd);
    if (m->rank == 5) oracle_sparse_mul4(m->jnnz, m->jcoo, m->jval, n, xx, xx, xx, full);     /* tendencies.py:105-109 */
    else oracle_sparse_mul2(m->jnnz, m->jcoo, m->jval, n, xx, full);
    for (int64_t i = 0; i < nd; ++i)
        memcpy(out + i * nd, full + (i + 1) * n + 1, sizeof(double) * nd);
}

void oracle_tendencies_r(int rank, int64_t ndim, int64_t nnz, const int32_t *coo, const double *val,
                         int64_t n_traj, const double *x /* n_traj*ndim */, double *out)
{
    oracle_model m = {ndim, nnz, coo, val, 0, NULL, NULL, rank};
    double *work = (double *)malloc(sizeof(double) * 2 * (ndim + 1));
    for (int64_t t = 0; t < n_traj; ++t) model_f(&m, x + t * ndim, out + t * ndim, work);
    free(work);
}

void oracle_tendencies(int64_t ndim, int64_t nnz, const int32_t *coo, const double *val,
                       int64_t n_traj, const double *x, double *out)
{
    oracle_tendencies_r(3, ndim, nnz, coo, val, n_traj, x, out);
}

void oracle_jacobian_r(int rank, int64_t ndim, int64_t jnnz, const int32_t *jcoo, const double *jval,
                       int64_t n_traj, const double *x, double *out /* n_traj*ndim*ndim */)
{
    oracle_model m = {ndim, 0, NULL, NULL, jnnz, jcoo, jval, rank};
    double *work = (double *)malloc(sizeof(double) * ((ndim + 1) + (ndim + 1) * (ndim + 1)));
    for (int64_t t = 0; t < n_traj; ++t) model_Df(&m, x + t * ndim, out + t * ndim * ndim, work);
    free(work);
}

void oracle_jacobian(int64_t ndim, int64_t jnnz, const int32_t *jcoo, const double *jval,
                     int64_t n_traj, const double *x, double *out)
{
    oracle_jacobian_r(3, ndim, jnnz, jcoo, jval, n_traj, x, out);
}

/* Number of records, qgs/integrators/integrate.py:190-196 (and integrator.py:378-384):
 *   write_steps == 0 -> 1; else len(time[::ws]) (+1 if time[::ws][-1] != time[-1]) */
int64_t oracle_n_records(const double *time, int64_t n_time, int64_t write_steps)
{
    if (write_steps == 0) return 1;
    int64_t n = (n_time + write_steps - 1) / write_steps;
    int64_t last = (n - 1) * write_steps;
    if (time[last] != time[n_time - 1]) n += 1;
    return n;
}

/* qgs/integrators/integrate.py:182-223  _integrate_runge_kutta_jit
 * time (n_time), ic (n_traj,n_dim), b (s), c (s), a (s,s) -> recorded_traj (n_traj,n_dim,n_records),
 * already reversed along the record axis when time_direction == -1 (:223).
 * `threads` > 1 parallelises the (independent) trajectory loop (:204) with OpenMP. */
void oracle_rk_integrate_r(int rank, int64_t ndim, int64_t nnz, const int32_t *coo, const double *val,
                           int64_t n_traj, const double *ic,
                           const double *time, int64_t n_time, int time_direction, int64_t write_steps,
                           int s, const double *b, const double *c, const double *a,
                           int64_t n_records, double *recorded, int threads)
{
    (void)c; /* autonomous system: f ignores t (tendencies.py:112) */
    oracle_model m = {ndim, nnz, coo, val, 0, NULL, NULL, rank};
    /* directed_time = reverse(time) if backward (:199-202) */
    double *dtime = (double *)malloc(sizeof(double) * n_time);
    for (int64_t i = 0; i < n_time; ++i) dtime[i] = (time_direction == -1) ? time[n_time - 1 - i] : time[i];

#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
#endif
    {
        double *y = (double *)malloc(sizeof(double) * ndim);
        double *ys = (double *)malloc(sizeof(double) * ndim);
        double *k = (double *)malloc(sizeof(double) * s * ndim);
        double *work = (double *)malloc(sizeof(double) * 2 * (ndim + 1));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t it = 0; it < n_traj; ++it) {
            double *rec = recorded + it * ndim * n_records;
            memcpy(y, ic + it * ndim, sizeof(double) * ndim);
            int64_t iw = 0;
            for (int64_t ti = 0; ti + 1 < n_time; ++ti) {
                const double dt = dtime[ti + 1] - dtime[ti];                 /* np.diff(directed_time) */
                if (write_steps > 0 && ti % write_steps == 0) {              /* :210-212 */
                    for (int64_t d = 0; d < ndim; ++d) rec[d * n_records + iw] = y[d];
                    iw++;
                }
                for (int64_t q = 0; q < (int64_t)s * ndim; ++q) k[q] = 0.0;  /* k.fill(0.) */
                for (int i = 0; i < s; ++i) {
                    /* y_s = y + (dt * a[i]) @ k     (:216) */
                    for (int64_t d = 0; d < ndim; ++d) {
                        double acc = 0.0;
                        for (int j = 0; j < s; ++j) acc += (dt * a[i * s + j]) * k[j * ndim + d];
                        ys[d] = y[d] + acc;
                    }
                    model_f(&m, ys, k + (int64_t)i * ndim, work);            /* k[i] = f(tt + c[i]*dt, y_s) */
                }
                /* y_new = y + (dt * b) @ k          (:218) */
                for (int64_t d = 0; d < ndim; ++d) {
                    double acc = 0.0;
                    for (int j = 0; j < s; ++j) acc += (dt * b[j]) * k[j * ndim + d];
                    y[d] = y[d] + acc;
                }
            }
            for (int64_t d = 0; d < ndim; ++d) rec[d * n_records + (n_records - 1)] = y[d];   /* :221 */
            if (time_direction == -1) {                                       /* [:, :, ::-1]  (:223) */
                for (int64_t d = 0; d < ndim; ++d) {
                    double *row = rec + d * n_records;
                    for (int64_t l = 0, r = n_records - 1; l < r; ++l, --r) { double t = row[l]; row[l] = row[r]; row[r] = t; }
                }
            }
        }
        free(y); free(ys); free(k); free(work);
    }
    free(dtime);
}

void oracle_rk_integrate(int64_t ndim, int64_t nnz, const int32_t *coo, const double *val,
                         int64_t n_traj, const double *ic,
                         const double *time, int64_t n_time, int time_direction, int64_t write_steps,
                         int s, const double *b, const double *c, const double *a,
                         int64_t n_records, double *recorded, int threads)
{
    oracle_rk_integrate_r(3, ndim, nnz, coo, val, n_traj, ic, time, n_time, time_direction, write_steps, s, b, c, a,
                          n_records, recorded, threads);
}

/* qgs/integrators/integrate.py:555-614  _integrate_runge_kutta_tgls_jit, with
 * _tangent_linear_system (:226-231) and boundary = _zeros_func (:235-237).
 * tg_ic (n_traj, n_dim, n_tg) -> recorded_traj (n_traj,n_dim,n_records),
 * recorded_fmatrix (n_traj, n_dim, n_tg, n_records); `inverse` is the +-1.0 multiplier. */
void oracle_rk_tgls_integrate_r(int rank, int64_t ndim, int64_t nnz, const int32_t *coo, const double *val,
                                int64_t jnnz, const int32_t *jcoo, const double *jval,
                                int64_t n_traj, int64_t n_tg, const double *ic, const double *tg_ic,
                                const double *time, int64_t n_time, int time_direction, int64_t write_steps,
                                int s, const double *b, const double *c, const double *a,
                                int adjoint, double inverse,
                                int64_t n_records, double *recorded, double *recorded_fm, int threads)
{
    (void)c;
    oracle_model m = {ndim, nnz, coo, val, jnnz, jcoo, jval, rank};
    const int64_t nm = ndim * n_tg;
    double *dtime = (double *)malloc(sizeof(double) * n_time);
    for (int64_t i = 0; i < n_time; ++i) dtime[i] = (time_direction == -1) ? time[n_time - 1 - i] : time[i];

#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
#endif
    {
        double *y = (double *)malloc(sizeof(double) * ndim);
        double *ys = (double *)malloc(sizeof(double) * ndim);
        double *k = (double *)malloc(sizeof(double) * s * ndim);
        double *fm = (double *)malloc(sizeof(double) * nm);
        double *kms = (double *)malloc(sizeof(double) * nm);
        double *km = (double *)malloc(sizeof(double) * s * nm);
        double *J = (double *)malloc(sizeof(double) * ndim * ndim);
        double *work = (double *)malloc(sizeof(double) * ((ndim + 1) + (ndim + 1) * (ndim + 1)));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t it = 0; it < n_traj; ++it) {
            double *rec = recorded + it * ndim * n_records;
            double *recm = recorded_fm + it * nm * n_records;
            memcpy(y, ic + it * ndim, sizeof(double) * ndim);
            memcpy(fm, tg_ic + it * nm, sizeof(double) * nm);
            for (int64_t d = 0; d < ndim; ++d) rec[d * n_records] = y[d];          /* :581 */
            for (int64_t q = 0; q < nm; ++q) recm[q * n_records] = fm[q];          /* :582 */
            int64_t iw = 0;
            for (int64_t ti = 0; ti + 1 < n_time; ++ti) {
                const double dt = dtime[ti + 1] - dtime[ti];
                if (write_steps > 0 && ti % write_steps == 0) {                    /* :588-591 */
                    for (int64_t d = 0; d < ndim; ++d) rec[d * n_records + iw] = y[d];
                    for (int64_t q = 0; q < nm; ++q) recm[q * n_records + iw] = fm[q];
                    iw++;
                }
                for (int64_t q = 0; q < (int64_t)s * ndim; ++q) k[q] = 0.0;
                for (int64_t q = 0; q < (int64_t)s * nm; ++q) km[q] = 0.0;
                for (int i = 0; i < s; ++i) {
                    for (int64_t d = 0; d < ndim; ++d) {                           /* y_s  (:596) */
                        double acc = 0.0;
                        for (int j = 0; j < s; ++j) acc += (dt * a[i * s + j]) * k[j * ndim + d];
                        ys[d] = y[d] + acc;
                    }
                    model_f(&m, ys, k + (int64_t)i * ndim, work);                  /* :597 */
                    memcpy(kms, fm, sizeof(double) * nm);                          /* km_s = fm.copy() */
                    for (int j = 0; j < s; ++j) {                                  /* :599-600 */
                        const double w = dt * a[i * s + j];
                        for (int64_t q = 0; q < nm; ++q) kms[q] += w * km[(int64_t)j * nm + q];
                    }
                    model_Df(&m, ys, J, work);                                     /* fjac(t, xs) */
                    double *kmi = km + (int64_t)i * nm;
                    for (int64_t r = 0; r < ndim; ++r)                             /* :226-231, :601-603 */
                        for (int64_t q = 0; q < n_tg; ++q) {
                            double acc = 0.0;
                            if (adjoint) for (int64_t l = 0; l < ndim; ++l) acc += J[l * ndim + r] * kms[l * n_tg + q];
                            else         for (int64_t l = 0; l < ndim; ++l) acc += J[r * ndim + l] * kms[l * n_tg + q];
                            kmi[r * n_tg + q] = inverse * acc + 0.0;               /* hom + inhom (zeros) */
                        }
                }
                for (int64_t d = 0; d < ndim; ++d) {                               /* :604 */
                    double acc = 0.0;
                    for (int j = 0; j < s; ++j) acc += (dt * b[j]) * k[j * ndim + d];
                    y[d] = y[d] + acc;
                }
                for (int j = 0; j < s; ++j) {                                      /* :605-607 */
                    const double w = dt * b[j];
                    for (int64_t q = 0; q < nm; ++q) fm[q] += w * km[(int64_t)j * nm + q];
                }
            }
            for (int64_t d = 0; d < ndim; ++d) rec[d * n_records + (n_records - 1)] = y[d];   /* :611 */
            for (int64_t q = 0; q < nm; ++q) recm[q * n_records + (n_records - 1)] = fm[q];   /* :612 */
            if (time_direction == -1) {
                for (int64_t d = 0; d < ndim; ++d) {
                    double *row = rec + d * n_records;
                    for (int64_t l = 0, r = n_records - 1; l < r; ++l, --r) { double t = row[l]; row[l] = row[r]; row[r] = t; }
                }
                for (int64_t q = 0; q < nm; ++q) {
                    double *row = recm + q * n_records;
                    for (int64_t l = 0, r = n_records - 1; l < r; ++l, --r) { double t = row[l]; row[l] = row[r]; row[r] = t; }
                }
            }
        }
        free(y); free(ys); free(k); free(fm); free(kms); free(km); free(J); free(work);
    }
    free(dtime);
}

void oracle_rk_tgls_integrate(int64_t ndim, int64_t nnz, const int32_t *coo, const double *val,
                              int64_t jnnz, const int32_t *jcoo, const double *jval,
                              int64_t n_traj, int64_t n_tg, const double *ic, const double *tg_ic,
                              const double *time, int64_t n_time, int time_direction, int64_t write_steps,
                              int s, const double *b, const double *c, const double *a,
                              int adjoint, double inverse,
                              int64_t n_records, double *recorded, double *recorded_fm, int threads)
{
    oracle_rk_tgls_integrate_r(3, ndim, nnz, coo, val, jnnz, jcoo, jval, n_traj, n_tg, ic, tg_ic, time, n_time, time_direction,
                               write_steps, s, b, c, a, adjoint, inverse, n_records, recorded, recorded_fm, threads);
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
