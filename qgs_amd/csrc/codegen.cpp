// codegen.cpp -- see codegen.h.  Emits HIP C++ source; everything is straight-line fp64 code built from the
// tensor, so the compiled kernels contain no tensor index loads at all: the coefficients arrive through the
// scalar unit (per-kernel __constant__ tables walked with s_load_dwordx16) and every VALU slot is a
// v_mul_f64 / v_fma_f64.  The tables are declared without initialisers: their contents are returned next to the source
// (GeneratedKernel::tables) and stored into the loaded module, so the source -- and the code object -- depend on the
// structure of the tensor only, not on its values.
#include "codegen.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

namespace qgs {

namespace {

std::string hexlit(double v)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%a", v);   // exact hex-float literal (C++17)
    return std::string(buf);
}

// Coefficient table mode: instead of a literal (two s_mov_b32 per use) a coefficient can be the next entry of
// a __constant__ table that the kernel walks sequentially (fetched eight at a time by s_load_dwordx16).
// Every stage emits the same rows in the same order, so one table per (kernel, wave partition) serves all stages.
struct KTable {
    std::vector<double> vals;
    size_t cursor = 0;
    size_t pad_to = 0;      // group-64 mode: the run-ahead loads may touch this many entries
};
thread_local KTable *g_ktab = nullptr;
thread_local std::vector<CoefTable> *g_tables = nullptr;     // tables of the kernel being generated (generate_kernel)

std::string lit(double v)
{
    // no literal coefficients: a value in the source would tie the code object to one parameter set
    if (!g_ktab) throw std::logic_error("codegen: coefficient outside a coefficient table");
    return "@K" + hexlit(v) + "@";        // resolved to kt[n] in final text order by resolve_ktab()
}

// acc = fma(c, factor, acc) with a tabulated coefficient c
std::string coef_fma(const std::string &acc, double c, const std::string &factor)
{
    return acc + " = __builtin_fma(" + lit(c) + ", " + factor + ", " + acc + ");";
}

// Replace the @K<value>@ placeholders of one stage's text by sequential table references.
//
// Explicit software pipeline (left to itself the compiler merges the scalar loads into s_load_dwordx16 and issues each ~8
//   instructions ahead of its first use: 5.75 instead of 4.6 ms for the stepper).  SMEM returns out of order, so every wait is lgkmcnt(0) and also
//   waits for whatever was issued last; a lone wavefront then stalls (L - 45) cycles per 8 coefficients (PMC:
//   21 % of the stepper's cycles).  Here the coefficients are consumed in groups of 16 held in two 8-double vectors
//   `kq<2g>`, `kq<2g+1>`; the loads of group g+1 are issued right AFTER the first statement that uses group g
//   (which carries the wait) and pinned there with sched_barriers, so they fly during the 16 FMAs of group g.
// dedupe (LDS-resident kernels): a coefficient whose magnitude already sits in the group of 16 that is being consumed is
// not fetched again, the statement refers to that entry (negated if the sign differs).  MAOOAM 6x6: cos / sin partner
// modes and the psi / theta copies of the advection terms repeat their coefficients in neighbouring statements, 21 657
// fetches per workgroup-stage become ~15 000 -- and the coefficient stream is what bounds that kernel.
std::string resolve_ktab(const std::string &text, KTable &t, bool dedupe = false)
{
    std::string out;
    t.cursor = 0;
    auto next_ref = [&](double v, bool *ok) {
        if (t.cursor == t.vals.size()) t.vals.push_back(v);
        *ok = (t.vals[t.cursor] == v);        // always true: every stage emits the same sequence
        return t.cursor++;
    };
    // count the coefficients of this stage first (number of 8-double blocks that exist)
    size_t total = 0;
    for (size_t p = text.find("@K"); p != std::string::npos; p = text.find("@K", text.find('@', p + 2) + 1)) ++total;
    const size_t nblocks = (total + 7) / 8;
    auto load_group = [&](size_t g, const char *ind) {
        std::string l;
        for (size_t b = 2 * g; b < 2 * g + 2 && b < nblocks; ++b)
            l += std::string(ind) + "const v8d kq" + std::to_string(b) + " = *(const kv8*)(kt + " + std::to_string(8 * b) + ");\n";
        return l;
    };
    const char *ind = "                ";
    out += load_group(0, ind);
    long opened = -1;                         // highest group whose successor has been requested
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        std::string line = text.substr(pos, eol - pos), res;
        long first_group = -1;
        size_t lp = 0;
        while (true) {
            size_t a = line.find("@K", lp);
            if (a == std::string::npos) { res.append(line, lp, std::string::npos); break; }
            size_t b = line.find('@', a + 2);
            res.append(line, lp, a - lp);
            const double v = std::strtod(line.substr(a + 2, b - a - 2).c_str(), nullptr);
            bool ok;
            size_t n = 0;
            bool reused = false;
            if (dedupe && t.cursor > 0) {
                // first entry of the group being consumed.  (Searching the group before it as well -- its registers would have to
                // stay live -- finds 7 % more repeats at ndim 228 and costs more in SGPR spills: 52.2 instead of 50.6 ms.)
                const size_t g0 = (t.cursor - 1) / 16 * 16;
                for (size_t q = t.cursor; q-- > g0;)
                    if (std::fabs(t.vals[q]) == std::fabs(v) && v != 0.0) { n = q; reused = true; break; }
            }
            if (reused) {
                const std::string ref = "kq" + std::to_string(n / 8) + "[" + std::to_string(n % 8) + "]";
                res += (std::signbit(t.vals[n]) == std::signbit(v)) ? ref : "(-" + ref + ")";
            } else {
                n = next_ref(v, &ok);
                if (!ok) throw std::logic_error("codegen: the stages of a kernel emit different coefficient sequences");
                res += "kq" + std::to_string(n / 8) + "[" + std::to_string(n % 8) + "]";
            }
            if (first_group < 0) first_group = (long)(n / 16);
            lp = b + 1;
        }
        out += res + "\n";
        if (first_group > opened) {           // first statement of a new group: now request the next group
            opened = first_group;
            const std::string l = load_group((size_t)first_group + 1, ind);
            if (!l.empty()) out += std::string(ind) + "__builtin_amdgcn_sched_barrier(0);\n" + l + ind + "__builtin_amdgcn_sched_barrier(0);\n";
        }
        pos = eol + 1;
    }
    return out;
}

std::vector<std::string> split_lines(const std::string &text)
{
    std::vector<std::string> lines;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        lines.push_back(text.substr(pos, e - pos));
        pos = e + 1;
    }
    return lines;
}

// Round-robin merge of independent statement lists: consecutive statements then belong to different rows,
// i.e. to independent fp64 dependency chains (a lone v_fma_f64 chain issues every 9 cycles, independent
// ones every 4-5.6).
std::string interleave(const std::vector<std::vector<std::string>> &lists)
{
    std::string out;
    size_t n = 0;
    for (auto &l : lists) n = std::max(n, l.size());
    for (size_t k = 0; k < n; ++k)
        for (auto &l : lists)
            if (k < l.size()) { out += l[k]; out += '\n'; }
    return out;
}

// The table is declared without an initialiser (a HIP __constant__ variable is `externally_initialized`: the compiler
// assumes nothing about its contents); the values go to the caller, who stores them into the loaded module.
void emit_ktable(std::ostringstream &o, const std::string &name, const KTable &t)
{
    const size_t padded = std::max<size_t>(std::max<size_t>(8, t.pad_to), (t.vals.size() + 7) / 8 * 8);      // whole 8-double blocks
    o << "__constant__ __attribute__((aligned(64))) f64 " << name << "[" << padded << "];   // " << t.vals.size()
      << " coefficients, filled after the module is loaded\n";
    if (!g_tables) throw std::logic_error("codegen: coefficient table outside generate_kernel");
    CoefTable ct;
    ct.symbol = name;
    ct.values = t.vals;
    ct.values.resize(padded, 0.0);
    g_tables->push_back(std::move(ct));
}

struct Bil { int j, k; double c; };
struct Lin { int k; double c; };

struct Row {
    double c0 = 0.0;
    bool has_c0 = false;
    std::vector<Lin> lin;
    std::vector<Bil> bil;
};

// rows[i] for i in 1..ndim from COO entries (entries are summed if duplicated)
std::vector<Row> build_rows(int ndim, const std::vector<Term> &tensor)
{
    std::vector<Row> rows(ndim + 1);
    for (const Term &t : tensor) {
        if (t.i < 1 || t.i > ndim) continue;        // row 0 is the constant slot: res[0] = 1 (sparse_mul.py:80)
        Row &r = rows[t.i];
        if (t.j == 0 && t.k == 0) { r.c0 += t.v; r.has_c0 = true; }
        else if (t.j == 0 || t.k == 0) r.lin.push_back({std::max(t.j, t.k), t.v});
        else r.bil.push_back({t.j, t.k, t.v});
    }
    return rows;
}

using NameFn = std::function<std::string(int)>;

// One accumulation "r": tracks whether it has been initialised to emit mul instead of fma.
struct Acc {
    std::ostringstream &o;
    std::string name;
    bool init = false;
    const char *indent;
    Acc(std::ostringstream &os, const std::string &n, const char *ind) : o(os), name(n), indent(ind) {}
    // (one v_mov_b64 from the SGPR pair; the compiler's own copy is two v_mov_b32)
    void set_const(double c) { o << indent << "f64 " << name << " = qgs_mov64(" << lit(c) << ");\n"; init = true; }
    // r += c * expr
    void add(const std::string &c, const std::string &expr)
    {
        if (!init) { o << indent << "f64 " << name << " = " << c << " * " << expr << ";\n"; init = true; }
        else o << indent << name << " = __builtin_fma(" << c << ", " << expr << ", " << name << ");\n";
    }
    void finish() { if (!init) { o << indent << "f64 " << name << " = 0.0;\n"; init = true; } }
};

// Emit the products of a group of (sign, left, right) factors into a temp `g`:
//   g = l0*r0; g = fma(+-l1, r1, g); ...
struct Prod { bool neg; std::string l, r; };

void emit_group(std::ostringstream &o, const char *indent, const std::string &g, const std::vector<Prod> &ps)
{
    for (size_t n = 0; n < ps.size(); ++n) {
        const Prod &p = ps[n];
        if (n == 0)
            o << indent << "f64 " << g << " = " << (p.neg ? "-" : "") << p.l << " * " << p.r << ";\n";
        else
            o << indent << g << " = __builtin_fma(" << (p.neg ? "-" : "") << p.l << ", " << p.r << ", " << g << ");\n";
    }
}

// Group items by |coefficient| (exact equality of the doubles), keeping first-appearance order.
template <class T>
std::vector<std::vector<T>> group_by_abs(const std::vector<T> &items)
{
    std::vector<std::vector<T>> groups;
    std::map<double, size_t> where;
    for (const T &t : items) {
        double a = std::fabs(t.c);
        auto it = where.find(a);
        if (it == where.end()) { where[a] = groups.size(); groups.push_back({t}); }
        else groups[it->second].push_back(t);
    }
    return groups;
}

// f_i(x): emits "f64 <res> = ...;" for row i.  X(k) names the register holding x_k.
void emit_tend_row(std::ostringstream &o, const char *indent, const Row &row, const std::string &res,
                   const NameFn &X, const CodegenOptions &opt, int uid)
{
    Acc acc(o, res, indent);
    if (row.has_c0 && row.c0 != 0.0) acc.set_const(row.c0);
    for (const Lin &l : row.lin) acc.add(lit(l.c), X(l.k));
    auto groups = group_by_abs(row.bil);
    int gi = 0;
    for (auto &g : groups) {
        if (g.size() == 1) {
            acc.add(lit(g[0].c), "(" + X(g[0].j) + " * " + X(g[0].k) + ")");
        } else {
            std::string gname = "g" + std::to_string(uid) + "_" + std::to_string(gi++);
            std::vector<Prod> ps;
            const bool ref_neg = std::signbit(g[0].c);
            for (const Bil &b : g) ps.push_back({std::signbit(b.c) != ref_neg, X(b.j), X(b.k)});
            emit_group(o, indent, gname, ps);
            acc.add(lit(g[0].c), gname);
        }
    }
    acc.finish();
}

// Tangent / adjoint rows are both "sum of c * w_a * x_b" lists built from the Jacobian tensor
// Tj (reference: jacobian_tensor, qgtensor.py:700-722; J[i][j] = sum_k Tj_ijk x_k, sparse_mul.py:40-45):
//   tangent  (J w)_i   = sum_{j,k} Tj_ijk x_k w_j      -> row i   gets {w=j, x=k}
//   adjoint  (J^T w)_j = sum_{i,k} Tj_ijk x_k w_i      -> row j   gets {w=i, x=k}
// (x index 0 is the constant slot: the factor x_0 = 1 is dropped.)  Because Tj holds both (i,j,k) and
// (i,k,j) with equal value, grouping by |c| recovers c*(x_k w_j + x_j w_k) with one final FMA.
struct WX { int w, x; double c; };

std::vector<std::vector<WX>> build_wx_rows(int ndim, const std::vector<Term> &jac, bool adjoint)
{
    std::vector<std::vector<WX>> out(ndim + 1);
    for (const Term &t : jac) {
        if (t.i < 1 || t.j < 1 || t.i > ndim || t.j > ndim) continue;     // Df drops row/column 0 (tendencies.py:121)
        if (adjoint) out[t.j].push_back({t.i, t.k, t.v});
        else out[t.i].push_back({t.j, t.k, t.v});
    }
    return out;
}

void emit_wx_row(std::ostringstream &o, const char *indent, const std::vector<WX> &items,
                 const std::string &res, const NameFn &X, const NameFn &W, const CodegenOptions &opt, int uid)
{
    Acc acc(o, res, indent);
    std::vector<WX> lin, bil;
    for (const WX &a : items) (a.x == 0 ? lin : bil).push_back(a);
    for (const WX &a : lin) acc.add(lit(a.c), W(a.w));
    auto groups = group_by_abs(bil);
    int gi = 0;
    for (auto &g : groups) {
        if (g.size() == 1) {
            acc.add(lit(g[0].c), "(" + X(g[0].x) + " * " + W(g[0].w) + ")");
        } else {
            std::string gname = "g" + std::to_string(uid) + "_" + std::to_string(gi++);
            std::vector<Prod> ps;
            const bool ref_neg = std::signbit(g[0].c);
            for (const WX &a : g) ps.push_back({std::signbit(a.c) != ref_neg, X(a.x), W(a.w)});
            emit_group(o, indent, gname, ps);
            acc.add(lit(g[0].c), gname);
        }
    }
    acc.finish();
}

// Derived monomials (rank-5 tensors, see reduce_polynomial in codegen.h): indices above the model's ndim name
// products of two earlier variables, `q<idx>`, defined at the head of the block that evaluates a stage.
thread_local int g_ext_base = 1 << 30;

NameFn names(const std::string &prefix)
{
    const int base = g_ext_base;
    return [prefix, base](int k) { return (k > base ? std::string("q") : prefix) + std::to_string(k); };
}

void emit_derived(std::ostringstream &o, const char *indent, int ndim, const std::vector<std::pair<int, int>> &der, const NameFn &X)
{
    for (size_t n = 0; n < der.size(); ++n)
        o << indent << "const f64 q" << (ndim + 1 + (int)n) << " = " << X(der[n].first) << " * " << X(der[n].second) << ";\n";
}

std::string decl_list(const std::string &prefix, int ndim)
{
    std::ostringstream o;
    o << "f64 ";
    for (int d = 1; d <= ndim; ++d) o << prefix << d << (d < ndim ? ", " : ";");
    return o.str();
}

// "Use" the freshly loaded values before the step loop.  Without it the compiler waits for the initial loads where they
// are first needed -- inside the loop -- and, vmcnt being one in-order counter, that wait (vmcnt(0) at the loop head)
// also drains the RECORD STORES of the previous step on every iteration: a full store round trip per step
// (65 536 members x 100 steps with write_steps = 1: 0.69 ms, of which 0.23 ms were this stall).
void emit_settle_loads(std::ostringstream &o, const char *indent, const std::string &prefix, const std::vector<int> &idx)
{
    for (size_t a = 0; a < idx.size(); a += 12) {
        o << indent << "asm volatile(\"\" ::";
        for (size_t q = a; q < std::min(idx.size(), a + 12); ++q) o << (q > a ? ", " : " ") << "\"v\"(" << prefix << idx[q] << ")";
        o << ");\n";
    }
}

std::vector<int> all_rows(int ndim)
{
    std::vector<int> v;
    for (int d = 1; d <= ndim; ++d) v.push_back(d);
    return v;
}

const char *PRELUDE = R"(// ---- generated by qgs_amd/csrc/codegen.cpp: tensor-specialised gfx950 kernels -------------
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>   // offline hipcc build; hiprtc predefines the HIP builtins
#endif
typedef double f64;
typedef long long i64;
typedef const double __attribute__((address_space(4))) kf64;   // coefficient tables: scalar (s_load) fetches
typedef double v8d __attribute__((ext_vector_type(8)));
typedef const v8d __attribute__((address_space(4), aligned(64))) kv8;
#define QGS_WAVE 64
// Effective shader clock of a launch: lane 0 of workgroup 0 notes the shader-clock counter (s_memtime) and the constant 100 MHz
// counter (s_memrealtime) when it starts and when it has issued its last store; qgs_kernel_clock reads the four words back.
__device__ unsigned long long qgs_clock_probe[4];
#define QGS_CLOCK_MARK(k) if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { \
    qgs_clock_probe[k] = __builtin_amdgcn_s_memtime(); qgs_clock_probe[(k) + 1] = __builtin_amdgcn_s_memrealtime(); }
)";

// Record bookkeeping shared by the steppers (reference integrate.py:190-223): record `iw` of the directed
// run lands at index iw (forward) or n_records-1-iw (backward, the [::-1] of :223).
const char *RECORD_HELPERS = R"(
__device__ __forceinline__ i64 qgs_rec_index(i64 iw, i64 n_records, int backward)
{
    return backward ? (n_records - 1 - iw) : iw;
}
// step `ti` is recorded (as record iw = ti / write_steps) when ti % write_steps == 0: keep the next such
// step in a counter instead of dividing every step
#define QGS_REC_INIT i64 iw = 0, next_rec = -1; \
    if (write_steps > 0) { iw = (step_begin + write_steps - 1) / write_steps; next_rec = iw * write_steps; }
// a*b + c as the three-address v_fma_f64 (see emit_rk_kernel)
__device__ __forceinline__ f64 qgs_fma3(f64 a, f64 b, f64 c)
{
    f64 d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// v -> row[lane] with `row` a uniform pointer, as the scalar-base form of the store (SGPR pair + 32-bit lane offset).  The
// compiler forms a 64-bit vector address with one v_lshl_add_u64 per store instead (a VALU slot for a lone wavefront), also
// when base and offset are handed to it separately; an opaque offset per store costs a v_mov_b32 each.  The store is not
// tracked by the compiler's vmcnt bookkeeping, which only makes its own waits more conservative.
// The compiler does not look inside inline assembly for the hazards of what is in it, so the two that a store has are closed here:
//  * an SGPR written by a VALU instruction (v_readlane_b32 reloading a spilled SGPR, v_readfirstlane_b32) must not be the address
//    of a vector-memory instruction within five wait states: the address goes through an s_mov_b64 inside the statement -- a
//    scalar instruction may read such an SGPR at once, and what IT writes may be used at once;
//  * the data registers of a store of more than 64 bits must not be written by the next VALU instructions (two wait states on
//    gfx94x / gfx950): the 128-bit form ends with the wait.
__device__ __forceinline__ void qgs_store_row(f64* row, unsigned lane8, f64 v)
{
    f64* r;
    asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx2 %1, %2, %0" : "=&s"(r) : "v"(lane8), "v"(v), "s"(row) : "memory");
}
typedef double qgs_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void qgs_store_row2(f64* row, unsigned lane16, f64 a, f64 b)
{
    qgs_d2 pr;
    pr.x = a; pr.y = b;
    f64* r;
    asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx4 %1, %2, %0\n\ts_nop 1" : "=&s"(r) : "v"(lane16), "v"(pr), "s"(row) : "memory");
}
// a uniform double (SGPR pair) into a vector register with one v_mov_b64
__device__ __forceinline__ f64 qgs_mov64(f64 c)
{
    f64 d;
    asm("v_mov_b64 %0, %1" : "=v"(d) : "s"(c));
    return d;
}
// the double whose low / high word sit in lanes `lane` / `lane + 1` of v
__device__ __forceinline__ f64 qgs_lane_f64(unsigned v, int lane)
{
    const unsigned long long lo = __builtin_amdgcn_readlane(v, lane), hi = __builtin_amdgcn_readlane(v, lane + 1);
    return __builtin_bit_cast(f64, (hi << 32) | lo);
}
// mask ? a : b on the bit patterns (mask is all ones or all zeros), branch-free
__device__ __forceinline__ f64 qgs_bitsel(unsigned long long mask, f64 a, f64 b)
{
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    return __builtin_bit_cast(f64, (ua & mask) | (ub & ~mask));
}
)";

void emit_tend_kernel(std::ostringstream &out, int ndim, const std::vector<Row> &rows, const CodegenOptions &opt,
                      const std::vector<std::pair<int, int>> &der)
{
    std::ostringstream o;
    KTable table;
    o << "\n// f(t,x) for an ensemble: x, dx are X[mode][member] with leading dimension ld\n";
    o << "extern \"C\" __global__ void __launch_bounds__(64) qgs_spec_tend(const f64* __restrict__ x, f64* __restrict__ dx, i64 n_traj, i64 ld)\n{\n";
    o << "    const i64 m = (i64)blockIdx.x * QGS_WAVE + threadIdx.x;\n    if (m >= n_traj) return;\n";
    for (int d = 1; d <= ndim; ++d) o << "    const f64 x" << d << " = x[" << (d - 1) << " * ld + m];\n";
    emit_derived(o, "    ", ndim, der, names("x"));
    // brace-less rows: the coefficient group vectors of the table pipeline stay in scope (the kernel is bound by its
    // 2 * ndim memory accesses per member, not by where its coefficients come from)
    g_ktab = &table;
    o << "    kf64* kt = (kf64*)qgs_spec_tend_kt; asm volatile(\"\" : \"+s\"(kt));\n";
    std::ostringstream so;
    for (int i = 1; i <= ndim; ++i) {
        const std::string rn = "r" + std::to_string(i);
        emit_tend_row(so, "    ", rows[i], rn, names("x"), opt, i);
        so << "    dx[" << (i - 1) << " * ld + m] = " << rn << ";\n";
    }
    o << resolve_ktab(so.str(), table);
    g_ktab = nullptr;
    o << "}\n";
    emit_ktable(out, "qgs_spec_tend_kt", table);
    out << o.str();
}

void emit_jac_kernel(std::ostringstream &out, int ndim, const std::vector<Term> &jac, const std::vector<std::pair<int, int>> &der)
{
    std::ostringstream o;
    KTable table;
    // J[i][j] = sum_k Tj_ijk x_k  (sparse_mul2, sparse_mul.py:40-45); only structural entries are
    // stored, the caller zero-fills the output.  Output layout: Jm[(i-1)*ndim + (j-1)][member].
    std::map<std::pair<int, int>, std::vector<Lin>> ent;
    for (const Term &t : jac)
        if (t.i >= 1 && t.j >= 1) ent[{t.i, t.j}].push_back({t.k, t.v});
    o << "\n// Df(t,x) for an ensemble: only structurally non-zero entries are written\n";
    o << "extern \"C\" __global__ void __launch_bounds__(64) qgs_spec_jac(const f64* __restrict__ x, f64* __restrict__ jm, i64 n_traj, i64 ld)\n{\n";
    o << "    const i64 m = (i64)blockIdx.x * QGS_WAVE + threadIdx.x;\n    if (m >= n_traj) return;\n";
    for (int d = 1; d <= ndim; ++d) o << "    const f64 x" << d << " = x[" << (d - 1) << " * ld + m];\n";
    emit_derived(o, "    ", ndim, der, names("x"));
    const NameFn X = names("x");
    g_ktab = &table;
    o << "    kf64* kt = (kf64*)qgs_spec_jac_kt; asm volatile(\"\" : \"+s\"(kt));\n";
    std::ostringstream so;
    int en = 0;
    for (auto &kv : ent) {
        const std::string name = "e" + std::to_string(en++);
        Acc acc(so, name, "    ");
        double c0 = 0.0; bool has = false;
        for (const Lin &l : kv.second) if (l.k == 0) { c0 += l.c; has = true; }
        if (has) acc.set_const(c0);
        for (const Lin &l : kv.second) if (l.k != 0) acc.add(lit(l.c), X(l.k));
        acc.finish();
        so << "    jm[(i64)" << ((kv.first.first - 1) * ndim + (kv.first.second - 1)) << " * ld + m] = " << name << ";\n";
    }
    o << resolve_ktab(so.str(), table);
    g_ktab = nullptr;
    o << "}\n";
    emit_ktable(out, "qgs_spec_jac_kt", table);
    out << o.str();
}

// Fused S-stage explicit RK stepper for sub-diagonal tableaus, one member per lane, all state in
// registers for the whole run.  Storage: y (step start), acc (running y + dt*sum b_i k_i),
// xa/xb (ping-pong stage inputs).  k_i is consumed row by row as it is produced.
// spread_rec (qgs_spec_rkr_s<S>, launched for write_steps == 1, the reference's default): EVERY step is a record, so
// nothing about the record is conditional.  The burst version stores the 36 rows at the top of a step (36 x 512 B per
// wavefront, 18.9 MB for the whole chip at 65 536 members, all wavefronts in lock step): the store queues fill and the
// in-order wavefront sits behind them (measured 0.74 vs 0.61 ms per 100 steps).  y_i is constant for the whole step, so
// here its store goes out somewhere in the step: row r right after its evaluation in stage (r - 1) mod S, one 512-byte
// store every ~58 FMAs, addressed as scalar row pointer + lane offset (no 64-bit VALU address arithmetic).  Lanes past the
// last member write their own padding column of the record (the buffer has ld >= 64 * gridDim.x columns per row).
// pair_stages (qgs_spec_rkstagesp_s<S>, feeds qgs_spec_tglp_s<S>): the stage record holds the modes in pairs,
// S[..][mode / 2][member][2] (an odd last mode as before), written with one 128-bit store per pair.  The tangent kernel then
// needs half as many vector-memory instructions for the stage states, and each costs a lone wavefront ~3.4 issue slots
// (config 4: 0.932 instead of 0.965 ms per call).  Every other producer / consumer of stage records keeps S[..][mode][member].
void emit_rk_kernel(std::ostringstream &out, int ndim, const std::vector<Row> &rows, int S, bool store_stages,
                    const CodegenOptions &opt, const std::vector<std::pair<int, int>> &der, bool spread_rec = false,
                    bool pair_stages = false)
{
    std::ostringstream o;
    KTable table;
    spread_rec = spread_rec && !store_stages;
    pair_stages = pair_stages && store_stages;
    const std::string kname = std::string(store_stages ? (pair_stages ? "qgs_spec_rkstagesp_s" : "qgs_spec_rkstages_s")
                                                       : (spread_rec ? "qgs_spec_rkr_s" : "qgs_spec_rk_s")) + std::to_string(S);
    o << "\n// " << S << "-stage RK, " << (store_stages ? "also storing every stage input state" : "trajectory only")
      << (spread_rec ? ", every step a record (write_steps == 1)" : "") << "\n";
    o << "extern \"C\" __global__ void __launch_bounds__(64, " << opt.min_waves_per_simd << ") " << kname << "(\n"
      << "    const f64* __restrict__ y_in,   // X[mode][member] state at step `step_begin`\n"
      << "    f64* __restrict__ y_out,        // state after step `step_end-1` (may be null)\n"
      << "    f64* __restrict__ rec,          // R[record][mode][member] (may be null when no record is due)\n"
      << "    f64* __restrict__ stages,       // S[(step-step_begin)*" << S << "+stage][mode][member] (rkstages only)\n"
      << "    const f64* __restrict__ dtime,  // directed time grid\n"
      << "    const f64* __restrict__ tab,    // b[0.." << S - 1 << "], a[1][0], a[2][1], ...\n"
      << "    i64 n_traj, i64 ld, i64 step_begin, i64 step_end, i64 write_steps, i64 n_records, int backward, int write_final)\n{\n";
    o << "    QGS_CLOCK_MARK(0)\n";
    if (spread_rec || store_stages) o << "    const unsigned lane8 = threadIdx.x * 8u;\n";
    o << "    const i64 m0 = (i64)blockIdx.x * QGS_WAVE + threadIdx.x;\n"
      << "    const bool live = m0 < n_traj;\n"
      << "    const i64 m = live ? m0 : (n_traj - 1);   // tail lanes shadow the last member and never store\n";
    o << "    " << decl_list("y", ndim) << "\n";
    for (int d = 1; d <= ndim; ++d) o << "    y" << d << " = y_in[" << (d - 1) << " * ld + m];\n";
    // The 2S - 1 tableau entries sit in the lanes of ONE vector register (lane 2n / 2n + 1 = low / high word of tab[n]) and are
    // read back with two v_readlane where a stage needs them.  As 14 loop-invariant SGPRs next to the coefficient
    // pipeline's 64 they were spilled to lanes by the compiler anyway, and then reloaded as a block at every stage
    // boundary and at the end of every step (86 v_readlane per RK4 step instead of 14).
    o << "    unsigned tabw = 0;\n"
      << "    if (threadIdx.x < " << 2 * (2 * S - 1) << ") tabw = ((const unsigned*)tab)[threadIdx.x];\n";
    emit_settle_loads(o, "    ", "y", all_rows(ndim));
    o << "    QGS_REC_INIT\n";
    // The steps between two records are an inner loop of their own: what only the (cold) record block needs -- record
    // pointer, leading dimension, counters -- is then not part of the hot loop's scalar state.
    const bool nest = !spread_rec;
    if (nest) o << "    i64 ti = step_begin;\n    while (ti < step_end) {\n";
    else {
        o << "    for (i64 ti = step_begin; ti < step_end; ++ti) {\n";
        o << "        const f64 dt = dtime[ti + 1] - dtime[ti];\n";
    }
    if (spread_rec) {
        // write_steps == 1: step ti is record ti; uniform (scalar) pointer to this workgroup's 64 columns of row 0
        o << "        f64* const prow = rec + qgs_rec_index(ti, n_records, backward) * " << ndim << " * ld + (i64)blockIdx.x * QGS_WAVE;\n";
    } else {
        o << "        if (ti == next_rec) {\n"
          << "            f64* p = rec + qgs_rec_index(iw, n_records, backward) * " << ndim << " * ld + m;\n"
          << "            ++iw; next_rec += write_steps;\n"
          << "            if (live) {\n";
        for (int d = 1; d <= ndim; ++d) o << "                p[" << (d - 1) << " * ld] = y" << d << ";\n";
        o << "            }\n        }\n";
    }
    if (nest) {
        o << "        i64 seg_end = step_end;\n"
          << "        if (write_steps > 0 && next_rec < seg_end) seg_end = next_rec;    // next_rec > ti here\n"
          << "        for (; ti < seg_end; ++ti) {\n"
          << "        const f64 dt = dtime[ti + 1] - dtime[ti];\n";
    }
    o << "        " << decl_list("acc", ndim) << "\n";
    if (S > 1) o << "        " << decl_list("xa", ndim) << "\n";
    if (S > 2) o << "        " << decl_list("xb", ndim) << "\n";
    for (int st = 0; st < S; ++st) {
        const std::string in = (st == 0) ? "y" : ((st % 2 == 1) ? "xa" : "xb");
        const std::string outn = (st % 2 == 0) ? "xa" : "xb";
        const bool last = (st == S - 1);
        o << "        {   // stage " << st << "\n";
        o << "            unsigned tw = tabw; asm volatile(\"\" : \"+v\"(tw));   // keeps the v_readlane inside the stage\n";
        o << "            const f64 hb = dt * qgs_lane_f64(tw, " << 2 * st << ");\n";
        if (!last) o << "            const f64 ha = dt * qgs_lane_f64(tw, " << 2 * (S + st) << ");\n";
        if (store_stages) {
            // scalar row pointer + lane offset, the leading dimension opaque per stage: with `sp[d * ld]` the compiler kept the
            // 36 row offsets as loop-invariant SGPR pairs, spilled them to lanes and reloaded them in every stage
            // (982 v_readlane + 933 v_writelane in the kernel)
            o << "            {\n                i64 ldr = ld; asm volatile(\"\" : \"+s\"(ldr));\n"
              << "                f64* const srow = stages + ((ti - step_begin) * " << S << " + " << st << ") * " << ndim << " * ldr + (i64)blockIdx.x * QGS_WAVE;\n"
              << "                if (live) {\n";
            if (pair_stages) {             // pair (d, d + 1) at srow' = stage base + (d - 1) * ld + 128 * workgroup, 16 bytes per lane
                for (int d = 1; d + 1 <= ndim; d += 2)
                    o << "                    qgs_store_row2(srow + " << (d - 1) << " * ldr + (i64)blockIdx.x * QGS_WAVE, lane8 * 2u, " << in << d << ", " << in << (d + 1) << ");\n";
                if (ndim & 1) o << "                    qgs_store_row(srow + " << (ndim - 1) << " * ldr, lane8, " << in << ndim << ");\n";
            } else
            for (int d = 1; d <= ndim; ++d) o << "                    qgs_store_row(srow + " << (d - 1) << " * ldr, lane8, " << in << d << ");\n";
            o << "                }\n            }\n";
        }
        emit_derived(o, "            ", ndim, der, names(in));
        g_ktab = &table;
            o << "            kf64* kt = (kf64*)" << kname << "_kt; asm volatile(\"\" : \"+s\"(kt));\n";
        std::ostringstream so;
        for (int i = 1; i <= ndim; ++i) {
            const std::string rn = "r" + std::to_string(i);
            emit_tend_row(so, "            ", rows[i], rn, names(in), opt, st * 1000 + i);
            // (Record rows in mode pairs, one 128-bit store per pair as in the stage record of rkstagesp: 18 instead of 36 vector-
            // memory instructions per step, but 36 more VALU instructions to bring the pairs into aligned registers; measured
            // 0.599-0.630 against 0.609-0.624 ms per 100 steps: nothing, profiles/r03_record_path.txt.)
            if (spread_rec && (i - 1) % S == st) so << "            qgs_store_row(prow + " << (i - 1) << " * ld, lane8, y" << i << ");\n";
            // Whenever the addend stays live (y_i in every stage but the last) the sum is formed by an explicit three-address
            // v_fma_f64 (qgs_fma3): the compiler otherwise picks the two-address v_fmac_f64 plus a v_mov_b64 copy of the
            // addend (63 copies per RK4 step at ndim 36).  In the last stage y_i is dead (the stage input is xa / xb), so the new
            // state is written straight into it and no end-of-step copy y = acc is left.
            if (!last) so << "            " << outn << i << " = qgs_fma3(ha, " << rn << ", y" << i << ");\n";
            if (st == 0 && !last) so << "            acc" << i << " = qgs_fma3(hb, " << rn << ", y" << i << ");\n";
            else if (last && S > 1) so << "            y" << i << " = qgs_fma3(hb, " << rn << ", acc" << i << ");\n";
            else so << "            acc" << i << " = __builtin_fma(hb, " << rn << ", " << (st == 0 ? "y" : "acc") << i << ");\n";
        }
        o << resolve_ktab(so.str(), table);
        g_ktab = nullptr;
        o << "        }\n";
    }
    if (S == 1) for (int d = 1; d <= ndim; ++d) o << "        y" << d << " = acc" << d << ";\n";
    if (nest) o << "        }\n";
    o << "    }\n";
    o << "    if (live) {\n";
    o << "        if (y_out) {\n";
    for (int d = 1; d <= ndim; ++d) o << "            y_out[" << (d - 1) << " * ld + m] = y" << d << ";\n";
    o << "        }\n        if (write_final) {\n"
      << "            f64* p = rec + qgs_rec_index(n_records - 1, n_records, backward) * " << ndim << " * ld + m;\n";
    for (int d = 1; d <= ndim; ++d) o << "            p[" << (d - 1) << " * ld] = y" << d << ";\n";
    o << "        }\n    }\n    QGS_CLOCK_MARK(2)\n}\n";
    emit_ktable(out, kname + "_kt", table);
    out << o.str();
}

// General explicit tableau (dense lower-triangular `a`, e.g. Kutta's third-order scheme or the 3/8 rule; reference
// integrate.py:214-219 takes any b, c, a).  The input of stage q is P_q = y + dt * sum_{j<q} a_qj k_j.  k_j is still consumed
// row by row as it is produced: the next stage's input P_{j+1} is completed in registers (as in the sub-diagonal kernel)
// and the partial sums of the stages after that are read-modify-written in LDS, psum[q - 2][mode][lane] -- (S - 2) * ndim
// doubles per lane, 36.9 KB per wavefront for a 4-stage scheme at ndim 36, so four wavefronts still fit a CU.  A first
// version kept the k_j in a global scratch array: 170 MB of traffic per step at 65 536 members, 56 ms per 1000 steps;
// this one needs 216 LDS operations per member-step next to 2 076 FMAs.  tab = b[S], a[S*S] (row-major), run-time values.
void emit_rk_dense_kernel(std::ostringstream &out, int ndim, const std::vector<Row> &rows, int S, const CodegenOptions &opt,
                          const std::vector<std::pair<int, int>> &der)
{
    std::ostringstream o;
    KTable table;
    const std::string kname = "qgs_spec_rkd_s" + std::to_string(S);
    o << "\n// " << S << "-stage RK with a general lower-triangular tableau, partial stage sums in LDS\n";
    o << "extern \"C\" __global__ void __launch_bounds__(64, " << opt.min_waves_per_simd << ") " << kname << "(\n"
      << "    const f64* __restrict__ y_in, f64* __restrict__ y_out, f64* __restrict__ rec,\n"
      << "    f64* __restrict__ stages,       // S[(step-step_begin)*" << S << "+stage][mode][member] for the tangent model, or null\n"
      << "    const f64* __restrict__ dtime, const f64* __restrict__ tab,\n"
      << "    i64 n_traj, i64 ld, i64 step_begin, i64 step_end, i64 write_steps, i64 n_records, int backward, int write_final)\n{\n";
    if (S > 2) o << "    __shared__ f64 psum[" << (S - 2) << "][" << ndim << "][QGS_WAVE];\n";
    o << "    const int lane = threadIdx.x;\n"
      << "    const i64 m0 = (i64)blockIdx.x * QGS_WAVE + threadIdx.x;\n"
      << "    const bool live = m0 < n_traj;\n"
      << "    const i64 m = live ? m0 : (n_traj - 1);\n";
    o << "    " << decl_list("y", ndim) << "\n";
    for (int d = 1; d <= ndim; ++d) o << "    y" << d << " = y_in[" << (d - 1) << " * ld + m];\n";
    for (int st = 0; st < S; ++st) o << "    const f64 tb" << st << " = tab[" << st << "];\n";
    for (int i = 1; i < S; ++i)
        for (int j = 0; j < i; ++j) o << "    const f64 ta" << i << "_" << j << " = tab[" << (S + i * S + j) << "];\n";
    emit_settle_loads(o, "    ", "y", all_rows(ndim));
    o << "    QGS_REC_INIT\n";
    o << "    for (i64 ti = step_begin; ti < step_end; ++ti) {\n";
    o << "        const f64 dt = dtime[ti + 1] - dtime[ti];\n";
    o << "        if (ti == next_rec) {\n"
      << "            f64* p = rec + qgs_rec_index(iw, n_records, backward) * " << ndim << " * ld + m;\n"
      << "            ++iw; next_rec += write_steps;\n"
      << "            if (live) {\n";
    for (int d = 1; d <= ndim; ++d) o << "                p[" << (d - 1) << " * ld] = y" << d << ";\n";
    o << "            }\n        }\n";
    o << "        " << decl_list("acc", ndim) << "\n";
    if (S > 1) o << "        " << decl_list("xa", ndim) << "\n";
    if (S > 2) o << "        " << decl_list("xb", ndim) << "\n";
    for (int st = 0; st < S; ++st) {
        const std::string in = (st == 0) ? "y" : ((st % 2 == 1) ? "xa" : "xb");
        const std::string outn = (st % 2 == 0) ? "xa" : "xb";
        const bool last = (st == S - 1);
        o << "        {   // stage " << st << "\n";
        o << "            const f64 hb = dt * tb" << st << ";\n";
        for (int q = st + 1; q < S; ++q) o << "            const f64 h" << q << " = dt * ta" << q << "_" << st << ";\n";
        o << "            if (stages && live) {\n                f64* sp = stages + ((ti - step_begin) * " << S << " + " << st << ") * " << ndim << " * ld + m;\n";
        for (int d = 1; d <= ndim; ++d) o << "                sp[" << (d - 1) << " * ld] = " << in << d << ";\n";
        o << "            }\n";
        emit_derived(o, "            ", ndim, der, names(in));
        g_ktab = &table;
            o << "            kf64* kt = (kf64*)" << kname << "_kt; asm volatile(\"\" : \"+s\"(kt));\n";
        std::ostringstream so;
        for (int i = 1; i <= ndim; ++i) {
            const std::string rn = "r" + std::to_string(i);
            emit_tend_row(so, "            ", rows[i], rn, names(in), opt, st * 1000 + i);
            so << "            acc" << i << " = __builtin_fma(hb, " << rn << ", " << (st == 0 ? "y" : "acc") << i << ");\n";
            if (!last) {
                // input of the next stage, completed in registers
                const std::string base = (st == 0) ? "y" + std::to_string(i)
                                                   : "psum[" + std::to_string(st + 1 - 2) + "][" + std::to_string(i - 1) + "][lane]";
                so << "            " << outn << i << " = __builtin_fma(h" << (st + 1) << ", " << rn << ", " << base << ");\n";
                // partial sums of the stages after the next one
                for (int q = st + 2; q < S; ++q) {
                    const std::string slot = "psum[" + std::to_string(q - 2) + "][" + std::to_string(i - 1) + "][lane]";
                    so << "            " << slot << " = __builtin_fma(h" << q << ", " << rn << ", " << (st == 0 ? "y" + std::to_string(i) : slot) << ");\n";
                }
            }
        }
        o << resolve_ktab(so.str(), table);
        g_ktab = nullptr;
        o << "        }\n";
    }
    for (int d = 1; d <= ndim; ++d) o << "        y" << d << " = acc" << d << ";\n";
    o << "    }\n";
    o << "    if (live) {\n        if (y_out) {\n";
    for (int d = 1; d <= ndim; ++d) o << "            y_out[" << (d - 1) << " * ld + m] = y" << d << ";\n";
    o << "        }\n        if (write_final) {\n"
      << "            f64* p = rec + qgs_rec_index(n_records - 1, n_records, backward) * " << ndim << " * ld + m;\n";
    for (int d = 1; d <= ndim; ++d) o << "            p[" << (d - 1) << " * ld] = y" << d << ";\n";
    o << "        }\n    }\n}\n";
    emit_ktable(out, kname + "_kt", table);
    out << o.str();
}

// Row-split variant of the fused stepper: a workgroup of R wavefronts shares 64 members; wave w evaluates
// only the rows of its partition and the R partitions exchange the new stage state through LDS once per
// stage.  With n_traj/64 wavefronts of work a 1024-SIMD MI355X gets only ONE wave per SIMD from a
// 65 536-member ensemble, and a lone wave cannot issue fp64 FMAs back to back (measured: 5.6 cycles per
// independent v_fma_f64 against 4 with a second wave).  Splitting rows doubles the wave count for the
// same ensemble and shrinks the per-wave register footprint (own rows of y/acc/x_out + the full x_in).
std::vector<int> partition_rows(int ndim, const std::vector<Row> &rows, int R, const CodegenOptions &opt)
{
    std::vector<std::pair<int64_t, int>> cost;
    for (int i = 1; i <= ndim; ++i) {
        int64_t c = 2 + (int64_t)rows[i].lin.size();
        for (auto &g : group_by_abs(rows[i].bil)) c += (int64_t)g.size() + 1;
        cost.push_back({c, i});
    }
    std::sort(cost.begin(), cost.end(), [](const std::pair<int64_t, int> &a, const std::pair<int64_t, int> &b) {
        return a.first != b.first ? a.first > b.first : a.second < b.second;
    });
    std::vector<int64_t> load(R, 0);
    std::vector<int> owner(ndim + 1, 0);
    for (auto &ci : cost) {
        int w = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        owner[ci.second] = w;
        load[w] += ci.first;
    }
    return owner;
}

void emit_rk_split_kernel(std::ostringstream &out, int ndim, const std::vector<Row> &rows, int S, int R,
                          const CodegenOptions &opt, const std::vector<std::pair<int, int>> &der)
{
    const std::vector<int> owner = partition_rows(ndim, rows, R, opt);
    const std::string kname = "qgs_spec_rksplit" + std::to_string(R) + "_s" + std::to_string(S);
    std::ostringstream o;                       // kernel text; the coefficient tables are emitted in front of it
    std::vector<KTable> tables(R);
    o << "\n// " << S << "-stage RK, rows split over " << R << " wavefronts per 64 members (LDS exchange per stage)\n";
    o << "extern \"C\" __global__ void __launch_bounds__(" << 64 * R << ", " << R << ") " << kname << "(\n"
      << "    const f64* __restrict__ y_in, f64* __restrict__ y_out, f64* __restrict__ rec, f64* __restrict__ stages,\n"
      << "    const f64* __restrict__ dtime, const f64* __restrict__ tab,\n"
      << "    i64 n_traj, i64 ld, i64 step_begin, i64 step_end, i64 write_steps, i64 n_records, int backward, int write_final)\n{\n";
    o << "    __shared__ f64 xs[2][" << ndim << "][QGS_WAVE];\n";
    o << "    const int lane = threadIdx.x & 63;\n"
      << "    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n"
      << "    const i64 m0 = (i64)blockIdx.x * QGS_WAVE + lane;\n"
      << "    const bool live = m0 < n_traj;\n"
      << "    const i64 m = live ? m0 : (n_traj - 1);\n";
    for (int st = 0; st < S; ++st) o << "    const f64 tb" << st << " = tab[" << st << "];\n";
    for (int st = 0; st + 1 < S; ++st) o << "    const f64 ta" << st << " = tab[" << (S + st) << "];\n";
    for (int w = 0; w < R; ++w) {
        std::vector<int> own, other;
        for (int i = 1; i <= ndim; ++i) (owner[i] == w ? own : other).push_back(i);
        o << "    " << (w == 0 ? "if" : "else if") << " (wave == " << w << ") {   // rows:";
        for (int i : own) o << " " << i;
        o << "\n";
        g_ktab = &tables[w];
        o << "        " << decl_list("y", ndim) << "\n";
        for (int d = 1; d <= ndim; ++d) o << "        y" << d << " = y_in[" << (d - 1) << " * ld + m];\n";
        emit_settle_loads(o, "        ", "y", all_rows(ndim));
        o << "        QGS_REC_INIT\n";
        o << "        for (i64 ti = step_begin; ti < step_end; ++ti) {\n";
        o << "            const f64 dt = dtime[ti + 1] - dtime[ti];\n";
        o << "            const int par0 = (int)(((ti - step_begin) * " << S << ") & 1);\n";
        o << "            if (ti == next_rec) {\n"
          << "                f64* p = rec + qgs_rec_index(iw, n_records, backward) * " << ndim << " * ld + m;\n"
          << "                ++iw; next_rec += write_steps;\n"
          << "                if (live) {\n";
        for (int d : own) o << "                    p[" << (d - 1) << " * ld] = y" << d << ";\n";
        o << "                }\n            }\n";
        o << "            f64 ";
        for (size_t n = 0; n < own.size(); ++n) o << "acc" << own[n] << (n + 1 < own.size() ? ", " : ";\n");
        if (S > 1) o << "            " << decl_list("xa", ndim) << "\n";
        if (S > 2) o << "            " << decl_list("xb", ndim) << "\n";
        for (int st = 0; st < S; ++st) {
            const std::string in = (st == 0) ? "y" : ((st % 2 == 1) ? "xa" : "xb");
            const std::string out = (st % 2 == 0) ? "xa" : "xb";
            const bool last = (st == S - 1);
            o << "            {   // stage " << st << "\n";
            o << "                const f64 hb = dt * tb" << st << ";\n";
            if (!last) o << "                const f64 ha = dt * ta" << st << ";\n";
            o << "                const int pb = (par0 + " << st << ") & 1;\n";
            emit_derived(o, "                ", ndim, der, names(in));      // unused ones are dead code in this wavefront's branch
            o << "                kf64* kt = (kf64*)" << kname << "_kt" << w << "; asm volatile(\"\" : \"+s\"(kt));\n";
            {
                std::ostringstream so;
                const int W = std::max(1, opt.interleave);
                for (size_t c0 = 0; c0 < own.size(); c0 += W) {
                    std::vector<std::vector<std::string>> lists;
                    for (size_t q = c0; q < std::min(own.size(), c0 + W); ++q) {
                        const int i = own[q];
                        const std::string rn = "r" + std::to_string(i);
                        std::ostringstream ro;
                        emit_tend_row(ro, "                ", rows[i], rn, names(in), opt, w * 10000 + st * 100 + i);
                        ro << "                acc" << i << " = __builtin_fma(hb, " << rn << ", " << (st == 0 ? "y" : "acc") << i << ");\n";
                        if (!last) {
                            ro << "                " << out << i << " = __builtin_fma(ha, " << rn << ", y" << i << ");\n";
                            ro << "                xs[pb][" << (i - 1) << "][lane] = " << out << i << ";\n";
                        } else {
                            ro << "                xs[pb][" << (i - 1) << "][lane] = acc" << i << ";\n";
                        }
                        lists.push_back(split_lines(ro.str()));
                    }
                    so << interleave(lists);
                }
                o << resolve_ktab(so.str(), tables[w]);
            }
            o << "                __syncthreads();\n";
            if (!last) {
                for (int j : other) o << "                " << out << j << " = xs[pb][" << (j - 1) << "][lane];\n";
            } else {
                for (int i : own) o << "                y" << i << " = acc" << i << ";\n";
                for (int j : other) o << "                y" << j << " = xs[pb][" << (j - 1) << "][lane];\n";
            }
            o << "            }\n";
        }
        o << "        }\n";
        o << "        if (live) {\n            if (y_out) {\n";
        for (int d : own) o << "                y_out[" << (d - 1) << " * ld + m] = y" << d << ";\n";
        o << "            }\n            if (write_final) {\n"
          << "                f64* p = rec + qgs_rec_index(n_records - 1, n_records, backward) * " << ndim << " * ld + m;\n";
        for (int d : own) o << "                p[" << (d - 1) << " * ld] = y" << d << ";\n";
        o << "            }\n        }\n    }\n";
        g_ktab = nullptr;
    }
    o << "}\n";
    for (int w = 0; w < R; ++w) emit_ktable(out, kname + "_kt" + std::to_string(w), tables[w]);
    out << o.str();
}

// Tangent-linear / adjoint propagation along stored stage states.  One lane per (member, column):
// lane l = col*ld + member; tangent arrays are F[mode][col][member] = element d*(n_tg*ld) + l.
// share_x = C > 1: a workgroup of C wavefronts handles C columns of the same 64 members.  The stage states they all need
// (ndim values per member and stage, read from the record the trajectory pass wrote) go through LDS: every wavefront
// fetches 1/C of the NEXT stage's state at the start of a stage (the loads fly during the ~800 FMAs of the stage) and
// parks it in the other half of a double buffer at the end; one barrier per stage.  The plain kernel issues its ndim
// loads at the top of every stage and waits for them with nothing else to do (lone wavefront per SIMD: PMC, 19 % of
// the cycles in s_waitcnt), and reads every stage state once per column.
void emit_tgl_kernel(std::ostringstream &out, int ndim, const std::vector<std::vector<WX>> &tgl,
                     const std::vector<std::vector<WX>> &adj, int S, const CodegenOptions &opt,
                     const std::vector<std::pair<int, int>> &der, int share_x = 1, bool dense = false, bool pair_x = false)
{
    // dense: general lower-triangular tableau (tab = b[S], a[S*S]); the partial sums of the later stages' inputs are kept in
    // LDS exactly as in emit_rk_dense_kernel
    // pair_x (qgs_spec_tglp_s<S>): stage record in mode pairs as written by qgs_spec_rkstagesp_s<S> (see emit_rk_kernel)
    std::ostringstream o;
    KTable tables[2];
    const int C = dense ? 1 : std::max(1, share_x);
    const bool shx = C > 1;
    pair_x = pair_x && !shx && !dense;
    const std::string kname = dense ? "qgs_spec_tgld_s" + std::to_string(S)
                                    : (shx ? "qgs_spec_tglx" + std::to_string(C) + "_s" + std::to_string(S)
                                           : (pair_x ? "qgs_spec_tglp_s" : "qgs_spec_tgl_s") + std::to_string(S));
    o << "\n// tangent (adjoint=0) / adjoint (adjoint=1) model, " << S << "-stage RK, one lane per (member, column)";
    if (shx) o << ", " << C << " columns per workgroup sharing the stage states through LDS";
    o << "\n";
    if (dense) o << "// general lower-triangular tableau: partial sums of the later stages' tangent inputs in LDS\n";
    o << "extern \"C\" __global__ void __launch_bounds__(" << 64 * C << ", " << opt.min_waves_per_simd << ") " << kname << "(\n"
      << "    const f64* __restrict__ w_in_p,  // F[mode][col][member] at step `step_begin`\n"
      << "    f64* __restrict__ w_out_p,       // after step `step_end-1` (may be null)\n"
      << "    f64* __restrict__ rec,           // F[record][mode][col][member]\n"
      << "    const f64* __restrict__ stages,  // S[(step-step_begin)*" << S << "+stage][mode][member]\n"
      << "    const f64* __restrict__ dtime, const f64* __restrict__ tab,\n"
      << "    i64 n_traj, i64 ld, i64 n_tg, i64 step_begin, i64 step_end, i64 write_steps, i64 n_records,\n"
      << "    int backward, int write_final, int adjoint, f64 inverse)\n{\n";
    if (dense && S > 2) o << "    __shared__ f64 psw[" << (S - 2) << "][" << ndim << "][QGS_WAVE];\n";
    // park_v: the step-start vector v is the input of stage 0 and afterwards only the base of w_next_i = v_i + dt a k_i, read
    // once per row and stage.  Parked in LDS after stage 0 the kernel holds four vectors in registers instead of five and
    // the accumulation-register traffic (v_accvgpr moves are VALU slots) shrinks.
    const bool park_v = opt.tgl_park_v && S > 2 && !dense;
    if (park_v) o << "    __shared__ f64 vpk[" << C << "][" << ndim << "][QGS_WAVE];\n";
    if (shx) {
        o << "    __shared__ f64 xsh[2][" << ndim << "][QGS_WAVE];     // stage states of the 64 members, double-buffered\n";
        o << "    const int lane = threadIdx.x & 63;\n"
          << "    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n"
          << "    const i64 L = n_tg * ld;\n"
          << "    const i64 m0 = (i64)blockIdx.x * QGS_WAVE + lane;          // grid.x = ld / 64: m0 < ld\n"
          << "    const i64 c0 = (i64)blockIdx.y * " << C << " + wave;\n"
          << "    const bool live = (c0 < n_tg) && (m0 < n_traj);\n"
          << "    const i64 l = (c0 < n_tg ? c0 : n_tg - 1) * ld + m0;       // wavefronts past the last column shadow it (never store)\n"
          << "    const i64 m = m0 < n_traj ? m0 : n_traj - 1;\n"
          << "    const i64 g_total = (step_end - step_begin) * " << S << ";\n";
    } else {
        // Which (64-member group, column) a workgroup takes.  The n_tg column wavefronts of a member group read the same stage
        // states; workgroups go round-robin to the 8 XCDs (8 separate L2s), so when ld is a multiple of 64 XCD x takes the x-th
        // eighth of the member groups and runs the columns of a group back to back: the stage states then come from that
        // XCD's L2 instead of the Infinity Cache (grid = 8 * ceil(groups / 8) * n_tg, qgs_hip_api.hip launch_tgl()).
        o << "    const int lane = threadIdx.x;\n";
        o << "    const i64 L = n_tg * ld;\n"
          << "    i64 l0 = (i64)blockIdx.x * QGS_WAVE + threadIdx.x;\n"
          << "    if ((ld & 63) == 0) {\n"
          << "        const unsigned ng = (unsigned)(ld >> 6), per = (ng + 7u) >> 3, q = blockIdx.x >> 3;\n"
          << "        const unsigned grp = (blockIdx.x & 7u) * per + q / (unsigned)n_tg, colr = q % (unsigned)n_tg;\n"
          << "        if (grp >= ng) return;\n"
          << "        l0 = (i64)colr * ld + (i64)grp * QGS_WAVE + threadIdx.x;\n"
          << "    }\n"
          << "    const bool live = (l0 < L) && ((l0 % ld) < n_traj);\n"
          << "    const i64 l = (l0 < L) ? l0 : (L - 1);\n"
          << "    i64 m = l % ld; if (m >= n_traj) m = n_traj - 1;\n";
    }
    o << "    QGS_CLOCK_MARK(0)\n";
    o << "    " << decl_list("v", ndim) << "\n";
    for (int d = 1; d <= ndim; ++d) o << "    v" << d << " = w_in_p[" << (d - 1) << " * L + l];\n";
    for (int st = 0; st < S; ++st) o << "    const f64 tb" << st << " = tab[" << st << "];\n";
    if (dense) {
        for (int i = 1; i < S; ++i)
            for (int j = 0; j < i; ++j) o << "    const f64 ta" << i << "_" << j << " = tab[" << (S + i * S + j) << "];\n";
    } else {
        for (int st = 0; st + 1 < S; ++st) o << "    const f64 ta" << st << " = tab[" << (S + st) << "];\n";
    }
    emit_settle_loads(o, "    ", "v", all_rows(ndim));
    if (shx) {
        // first stage state: wavefront w brings the modes w, w + C, w + 2C, ... (slot q holds mode w + q*C)
        o << "    if (g_total > 0) {\n        const f64* sp0 = stages + m;\n";
        for (int q = 0; q * C < ndim; ++q) {
            const bool guard = (q + 1) * C > ndim;
            o << "        " << (guard ? "if (wave + " + std::to_string(q * C) + " < " + std::to_string(ndim) + ") " : "")
              << "xsh[0][wave + " << q * C << "][lane] = sp0[(i64)(wave + " << q * C << ") * ld];\n";
        }
        o << "    }\n    __syncthreads();\n";
    }
    o << "    QGS_REC_INIT\n";
    o << "    for (i64 ti = step_begin; ti < step_end; ++ti) {\n";
    o << "        const f64 dt = dtime[ti + 1] - dtime[ti];\n";
    o << "        if (ti == next_rec) {\n"
      << "            f64* p = rec + qgs_rec_index(iw, n_records, backward) * " << ndim << " * L + l;\n"
      << "            ++iw; next_rec += write_steps;\n"
      << "            if (live) {\n";
    for (int d = 1; d <= ndim; ++d) o << "                p[" << (d - 1) << " * L] = v" << d << ";\n";
    o << "            }\n        }\n";
    o << "        " << decl_list("acc", ndim) << "\n";
    if (S > 1) o << "        " << decl_list("wa", ndim) << "\n";
    if (S > 2) o << "        " << decl_list("wb", ndim) << "\n";
    if (park_v) for (int d = 1; d <= ndim; ++d) o << "        vpk[" << (shx ? "wave" : "0") << "][" << (d - 1) << "][lane] = v" << d << ";\n";
    for (int st = 0; st < S; ++st) {
        const std::string in = (st == 0) ? "v" : ((st % 2 == 1) ? "wa" : "wb");
        const std::string outn = (st % 2 == 0) ? "wa" : "wb";
        const bool last = (st == S - 1);
        o << "        {   // stage " << st << "\n";
        o << "            const f64 hb = dt * tb" << st << " * inverse;\n";      // inverse = +-1: exact
        if (dense) {
            for (int q = st + 1; q < S; ++q) o << "            const f64 h" << q << " = dt * ta" << q << "_" << st << " * inverse;\n";
        } else if (!last) o << "            const f64 ha = dt * ta" << st << " * inverse;\n";
        if (shx) {
            o << "            const i64 g = (ti - step_begin) * " << S << " + " << st << ";\n"
              << "            const int pb = (int)(g & 1);\n"
              << "            const f64* spn = stages + (g + 1 < g_total ? g + 1 : g) * " << ndim << " * ld + m;   // next stage state\n";
            for (int q = 0; q * C < ndim; ++q) {
                const bool guard = (q + 1) * C > ndim;
                o << "            f64 xn" << q << " = 0.0;\n";
                o << "            " << (guard ? "if (wave + " + std::to_string(q * C) + " < " + std::to_string(ndim) + ") " : "")
                  << "xn" << q << " = spn[(i64)(wave + " << q * C << ") * ld];\n";
            }
            for (int d = 1; d <= ndim; ++d) o << "            const f64 x" << d << " = xsh[pb][" << (d - 1) << "][lane];\n";
        } else {
            if (pair_x) {
                o << "            const f64* sp = stages + ((ti - step_begin) * " << S << " + " << st << ") * " << ndim << " * ld;\n";
                for (int d = 1; d + 1 <= ndim; d += 2)
                    o << "            const qgs_d2 xp" << d << " = *(const qgs_d2*)(sp + " << (d - 1) << " * ld + 2 * m); const f64 x" << d
                      << " = xp" << d << ".x, x" << (d + 1) << " = xp" << d << ".y;\n";
                if (ndim & 1) o << "            const f64 x" << ndim << " = sp[" << (ndim - 1) << " * ld + m];\n";
            } else {
                o << "            const f64* sp = stages + ((ti - step_begin) * " << S << " + " << st << ") * " << ndim << " * ld + m;\n";
                for (int d = 1; d <= ndim; ++d) o << "            const f64 x" << d << " = sp[" << (d - 1) << " * ld];\n";
            }
        }
        emit_derived(o, "            ", ndim, der, names("x"));
        for (int pass = 0; pass < 2; ++pass) {
            o << "            if (" << (pass == 0 ? "!adjoint" : "adjoint") << ") {\n";
            g_ktab = &tables[pass];
            o << "                kf64* kt = (kf64*)" << kname << "_kt" << pass << "; asm volatile(\"\" : \"+s\"(kt));\n";
            std::ostringstream so_all;
            std::vector<std::vector<std::string>> row_lines;
            for (int i = 1; i <= ndim; ++i) {                 // brace-less rows: the coefficient group vectors stay in scope
                std::ostringstream so;
                const std::string rn = "r" + std::to_string(i);
                emit_wx_row(so, "                ", pass == 0 ? tgl[i] : adj[i], rn, names("x"), names(in), opt,
                            pass * 100000 + st * 1000 + i);
                if (dense) {
                    so << "                acc" << i << " = __builtin_fma(hb, " << rn << ", " << (st == 0 ? "v" : "acc") << i << ");\n";
                    if (!last) {
                        const std::string base = (st == 0) ? "v" + std::to_string(i)
                                                           : "psw[" + std::to_string(st + 1 - 2) + "][" + std::to_string(i - 1) + "][lane]";
                        so << "                " << outn << i << " = __builtin_fma(h" << (st + 1) << ", " << rn << ", " << base << ");\n";
                        for (int q = st + 2; q < S; ++q) {
                            const std::string slot = "psw[" + std::to_string(q - 2) + "][" + std::to_string(i - 1) + "][lane]";
                            so << "                " << slot << " = __builtin_fma(h" << q << ", " << rn << ", "
                               << (st == 0 ? "v" + std::to_string(i) : slot) << ");\n";
                        }
                    }
                } else {
                    so << "                acc" << i << " = __builtin_fma(hb, " << rn << ", " << (st == 0 ? "v" : "acc") << i << ");\n";
                    if (!last) {
                        if (park_v && st > 0)
                            so << "                " << outn << i << " = __builtin_fma(ha, " << rn << ", vpk[" << (shx ? "wave" : "0") << "][" << (i - 1) << "][lane]);\n";
                        else so << "                " << outn << i << " = __builtin_fma(ha, " << rn << ", v" << i << ");\n";
                    }
                }
                row_lines.push_back(split_lines(so.str()));
            }
            {
                const int IW = std::max(1, opt.tgl_interleave);   // statements of IW consecutive rows round-robin (independent chains)
                for (size_t c0 = 0; c0 < row_lines.size(); c0 += IW) {
                    std::vector<std::vector<std::string>> grp(row_lines.begin() + c0, row_lines.begin() + std::min(row_lines.size(), c0 + IW));
                    so_all << interleave(grp);
                }
            }
            const std::ostringstream &so = so_all;
            o << resolve_ktab(so.str(), tables[pass], opt.tgl_coeff_dedupe);
            g_ktab = nullptr;
            o << "            }\n";
        }
        if (shx) {
            for (int q = 0; q * C < ndim; ++q) {
                const bool guard = (q + 1) * C > ndim;
                o << "            " << (guard ? "if (wave + " + std::to_string(q * C) + " < " + std::to_string(ndim) + ") " : "")
                  << "xsh[pb ^ 1][wave + " << q * C << "][lane] = xn" << q << ";\n";
            }
            o << "            __syncthreads();\n";
        }
        o << "        }\n";
    }
    for (int d = 1; d <= ndim; ++d) o << "        v" << d << " = acc" << d << ";\n";
    o << "    }\n";
    o << "    if (live) {\n        if (w_out_p) {\n";
    for (int d = 1; d <= ndim; ++d) o << "            w_out_p[" << (d - 1) << " * L + l] = v" << d << ";\n";
    o << "        }\n        if (write_final) {\n"
      << "            f64* p = rec + qgs_rec_index(n_records - 1, n_records, backward) * " << ndim << " * L + l;\n";
    for (int d = 1; d <= ndim; ++d) o << "            p[" << (d - 1) << " * L] = v" << d << ";\n";
    o << "        }\n    }\n    QGS_CLOCK_MARK(2)\n}\n";
    for (int pass = 0; pass < 2; ++pass) emit_ktable(out, kname + "_kt" + std::to_string(pass), tables[pass]);
    out << o.str();
}


// ---- LDS-resident stepper for systems that do not fit the register file (MAOOAM 6x6: ndim 228) ------------
// A workgroup of W wavefronts advances 64 members; the stage state lives in LDS as xs[mode][member] and wave w
// evaluates a contiguous block of rows.  The run-time-indexed generic kernel needs two LDS reads per tensor term
// (LDS-bound: measured 21 % of the fp64 rate).  Here the (j,k) pattern is compile-time knowledge, so the terms of
// a wave are reordered into "phases": a phase loads a small set of modes (<= cap) into registers once and then
// executes every term of the wave whose two factors are both in the set.  On the MAOOAM 6x6 tensor that is
// ~0.17 LDS reads per term instead of 2, and a product x_j*x_k needed by several rows of the wave is computed once
// (1.6 fp64 instructions per term instead of 2).
struct PTerm { int row, j, k; double c; };      // j <= k; j == 0: linear term c*x_k

struct Phase {
    std::vector<int> modes;                     // loaded at the head of the phase (ascending)
    std::vector<PTerm> terms;                   // sorted by (j, k, row)
};

std::vector<Phase> build_phases(int ndim, const std::vector<PTerm> &terms, int cap)
{
    typedef std::pair<int, int> Edge;
    std::map<Edge, std::vector<PTerm>> rem;
    for (const PTerm &t : terms) rem[{t.j, t.k}].push_back(t);
    std::vector<std::vector<int>> adj(ndim + 1);
    for (auto &kv : rem) {
        adj[kv.first.first].push_back(kv.first.second);
        if (kv.first.second != kv.first.first) adj[kv.first.second].push_back(kv.first.first);
    }
    auto count = [&](int a, int b) -> int {
        auto it = rem.find({std::min(a, b), std::max(a, b)});
        return it == rem.end() ? 0 : (int)it->second.size();
    };
    std::vector<Phase> phases;
    while (!rem.empty()) {
        std::vector<int> deg(ndim + 1, 0), gain(ndim + 1, 0);
        std::vector<char> in(ndim + 1, 0);
        for (auto &kv : rem) {
            deg[kv.first.first] += (int)kv.second.size();
            if (kv.first.second != kv.first.first) deg[kv.first.second] += (int)kv.second.size();
        }
        in[0] = 1;                                              // x_0 = 1 needs no register
        int n_in = 0;
        auto add = [&](int m) {
            in[m] = 1; ++n_in;
            for (int o : adj[m]) if (!in[o]) gain[o] += count(m, o);
        };
        for (int o : adj[0]) if (!in[o]) gain[o] += count(0, o);
        int seed = 0;
        for (int m = 1; m <= ndim; ++m)
            if (deg[m] > 0 && (seed == 0 || deg[m] > deg[seed])) seed = m;
        if (seed == 0) break;                                   // cannot happen: every remaining edge has a mode >= 1
        add(seed);
        while (n_in < cap) {
            int cand = 0, best = 0;
            for (int m = 1; m <= ndim; ++m) {
                if (in[m]) continue;
                const int sc = gain[m] + count(m, m);
                if (sc > best || (sc == best && sc > 0 && deg[m] > deg[cand])) { best = sc; cand = m; }
            }
            if (cand == 0 || best <= 0) break;
            add(cand);
        }
        Phase ph;
        for (int m = 1; m <= ndim; ++m) if (in[m]) ph.modes.push_back(m);
        for (auto it = rem.begin(); it != rem.end();) {
            if (in[it->first.first] && in[it->first.second]) {
                for (const PTerm &t : it->second) ph.terms.push_back(t);
                it = rem.erase(it);
            } else ++it;
        }
        // drop modes that ended up unused (a seed whose partners did not fit)
        std::vector<char> used(ndim + 1, 0);
        for (const PTerm &t : ph.terms) { used[t.j] = 1; used[t.k] = 1; }
        std::vector<int> keep;
        for (int m : ph.modes) if (used[m]) keep.push_back(m);
        ph.modes.swap(keep);
        if (ph.terms.empty()) break;                            // cannot happen for cap >= 2 (seed + one partner completes an edge)
        phases.push_back(std::move(ph));
    }
    return phases;
}

// ---- shared machinery of the LDS-resident kernels (stepper and tangent model) --------------------------------------
// Terms live in "node space": a node is one LDS-resident value (stepper: node m = mode m; tangent model: node j = w_j,
// node ndim + k = x_k); a term is c * node_j * node_k accumulated into row `row`, j == 0 meaning a single factor.
typedef std::vector<std::vector<PTerm>> RowTerms;           // [row] -> its terms

struct LdsStats { int64_t loads = 0, instr = 0, phases = 0, coef = 0; };

struct LdsNode { int64_t offset; int lane_kind; };          // byte offset in LDS without the lane part; which lane-offset variable
using NodeFn = std::function<LdsNode(int)>;

// fp64 instructions one wavefront spends per stage on the rows `own` (same rules as emit_lds_phases)
int64_t lds_wave_instr(int n_nodes, const RowTerms &rt, const std::vector<int> &own, int cap, bool group)
{
    std::vector<PTerm> terms;
    for (int i : own) terms.insert(terms.end(), rt[i].begin(), rt[i].end());
    int64_t n = 3 * (int64_t)own.size();
    for (const Phase &ph : build_phases(n_nodes, terms, cap)) {
        std::map<std::pair<int, double>, int> pieces;
        std::map<std::pair<int, int>, int> singles;
        for (const PTerm &t : ph.terms) {
            if (t.j == 0) { ++n; continue; }
            if (group) ++pieces[{t.row, std::fabs(t.c)}];
            else { ++singles[{t.j, t.k}]; ++n; }
        }
        if (group) {
            for (const PTerm &t : ph.terms) {
                if (t.j == 0) continue;
                if (pieces[{t.row, std::fabs(t.c)}] == 1) { ++singles[{t.j, t.k}]; ++n; }
            }
            for (auto &kv : pieces) if (kv.second > 1) n += kv.second + 1;
        }
        n += (int64_t)singles.size();
    }
    return n;
}

// Row blocks: neighbouring rows share most of their factors, so blocks are contiguous in a row sequence and balanced by
// cost; cheap rows (MAOOAM: the ocean rows) are first spread evenly through that sequence, otherwise one wavefront would
// own all of them and need twice the registers for its row state.  The estimate does not know how many products a
// wavefront can share between its rows, so the split is refined: count the instructions each block really needs,
// rescale the cost of its rows accordingly, split again.
std::vector<std::vector<int>> lds_partition(int n_rows, int n_nodes, const RowTerms &rt, int W, int cap, bool group)
{
    std::vector<int64_t> cost(n_rows + 1, 0);
    int64_t total = 0;
    for (int i = 1; i <= n_rows; ++i) {
        cost[i] = 1;
        for (const PTerm &t : rt[i]) cost[i] += (t.j == 0) ? 1 : 2;
        total += cost[i];
    }
    std::vector<int> seq;
    {
        std::vector<int> heavy, light;
        for (int i = 1; i <= n_rows; ++i) ((cost[i] * 2 * n_rows < total) ? light : heavy).push_back(i);
        int64_t heavy_total = 0, run = 0;
        for (int i : heavy) heavy_total += cost[i];
        size_t nl = 0;
        for (int i : heavy) {
            seq.push_back(i);
            run += cost[i];
            while (nl < light.size() && run * (int64_t)light.size() >= heavy_total * (int64_t)(nl + 1)) seq.push_back(light[nl++]);
        }
        while (nl < light.size()) seq.push_back(light[nl++]);
    }
    std::vector<std::vector<int>> owns;
    std::vector<double> c(cost.begin(), cost.end());
    double best_max = 0.0;
    for (int iter = 0; iter < 6; ++iter) {
        double tot = 0.0;
        for (int i = 1; i <= n_rows; ++i) tot += c[i];
        std::vector<std::vector<int>> cand(W);
        int w = 0;
        double run = 0.0;
        for (size_t q = 0; q < seq.size(); ++q) {
            cand[w].push_back(seq[q]);
            run += c[seq[q]];
            const size_t left = seq.size() - 1 - q;
            if (w + 1 < W && (run * W >= tot * (w + 1) || left <= (size_t)(W - 1 - w))) ++w;
        }
        double worst = 0.0;
        std::vector<double> actual(W, 0.0), est(W, 0.0);
        for (int v = 0; v < W; ++v) {
            std::sort(cand[v].begin(), cand[v].end());
            actual[v] = (double)lds_wave_instr(n_nodes, rt, cand[v], cap, group);
            for (int i : cand[v]) est[v] += c[i];
            worst = std::max(worst, actual[v]);
        }
        if (owns.empty() || worst < best_max) { owns = cand; best_max = worst; }
        for (int v = 0; v < W; ++v)
            if (est[v] > 0.0) for (int i : cand[v]) c[i] *= actual[v] / est[v];
    }
    return owns;
}

// The phases of one wavefront as straight-line code accumulating into k<row>.  `hook` is emitted in front of phase
// `hook_phase` (== phases.size(): behind the last one).
void emit_lds_phases(std::ostringstream &so, const char *ind, const std::vector<Phase> &phases, const NodeFn &node,
                     const std::vector<std::string> &lane_vars, const std::string &lds_ptr, bool group, int hook_phase,
                     const std::function<void(std::ostringstream &)> &hook, LdsStats &st, int order = 0)
{
    int ph_id = 0, prod_id = 0;
    for (const Phase &ph : phases) {
        if (ph_id == hook_phase) hook(so);
        const std::string sfx = "_" + std::to_string(ph_id++);
        // Opaque lane offset per phase: otherwise the compiler merges the reads of one value in different phases and keeps
        // it in a register (or scratch) in between.  ds_read offsets are 16-bit immediates, so every 64 KB window of the
        // LDS (and every lane-offset kind) gets its own base register.
        std::map<std::pair<int, int>, std::string> bases;
        for (int mo : ph.modes) {
            const LdsNode nd = node(mo);
            const std::pair<int, int> key{nd.lane_kind, (int)(nd.offset >> 16)};
            if (bases.count(key)) continue;
            const std::string name = "lb" + std::to_string(key.first) + "w" + std::to_string(key.second) + sfx;
            so << ind << "unsigned " << name << " = " << lane_vars[key.first];
            if (key.second) so << " + " << (int64_t)key.second * 65536 << "u";
            so << "; asm volatile(\"\" : \"+v\"(" << name << "));\n";
            bases[key] = name;
        }
        for (int mo : ph.modes) {
            const LdsNode nd = node(mo);
            so << ind << "const f64 x" << mo << sfx << " = *(const f64*)(" << lds_ptr << " + " << (nd.offset & 65535) << " + "
               << bases[{nd.lane_kind, (int)(nd.offset >> 16)}] << ");\n";
        }
        st.loads += (int64_t)ph.modes.size();
        ++st.phases;
        // Terms of one row with equal |coefficient| that fall into this phase are summed first
        // (c * (x_a x_b - x_c x_d ...): one fused multiply-add per term plus one for the coefficient); the
        // remaining single terms share their product between the rows of the wave that need it.
        std::map<std::pair<int, double>, std::vector<PTerm>> pieces;
        std::vector<PTerm> singles;
        for (const PTerm &t : ph.terms) {
            if (t.j == 0 || !group) singles.push_back(t);
            else pieces[{t.row, std::fabs(t.c)}].push_back(t);
        }
        // Order of the grouped statements inside a phase: by (row, |c|) (order 0), or by (|c|, row) (order 1).  Equal magnitudes
        // of different rows then sit next to each other -- the cos / sin partner rows of MAOOAM repeat their coefficients (219
        // of the 222 of rows 2 and 3 of the 6x6 model coincide) -- where the coefficient de-duplication of resolve_ktab (a
        // window of 16 entries) finds them: 15 175 instead of 15 472 table entries per workgroup-stage, 51.8 instead of 52.7 ms
        // (profiles/r03_lds228.md; sorting ALL statements of a phase by |c| gets 14 552 entries but separates the uses of the
        // shared products: 644 B of scratch, 65 ms -- not kept).
        std::vector<const std::vector<PTerm> *> piece_list;
        for (auto &kv : pieces) piece_list.push_back(&kv.second);
        if (order >= 1)
            std::stable_sort(piece_list.begin(), piece_list.end(), [](const std::vector<PTerm> *x, const std::vector<PTerm> *y) {
                return std::fabs((*x)[0].c) < std::fabs((*y)[0].c);
            });
        for (const std::vector<PTerm> *gp : piece_list) {
            const std::vector<PTerm> &g = *gp;
            if (g.size() == 1) { singles.push_back(g[0]); continue; }
            const std::string gname = "g" + std::to_string(prod_id++);
            const bool ref_neg = std::signbit(g[0].c);
            std::vector<Prod> ps;
            for (const PTerm &t : g)
                ps.push_back({std::signbit(t.c) != ref_neg, "x" + std::to_string(t.j) + sfx, "x" + std::to_string(t.k) + sfx});
            emit_group(so, ind, gname, ps);
            so << ind << coef_fma("k" + std::to_string(g[0].row), g[0].c, gname) << "\n";
            st.instr += (int64_t)g.size() + 1;
            ++st.coef;
        }
        std::sort(singles.begin(), singles.end(), [](const PTerm &x, const PTerm &y) {
            return x.j != y.j ? x.j < y.j : (x.k != y.k ? x.k < y.k : x.row < y.row);
        });
        size_t a = 0;
        while (a < singles.size()) {
            size_t b = a;
            while (b < singles.size() && singles[b].j == singles[a].j && singles[b].k == singles[a].k) ++b;
            const PTerm &t0 = singles[a];
            std::string factor;
            if (t0.j == 0) factor = "x" + std::to_string(t0.k) + sfx;
            else {
                const std::string pr = "x" + std::to_string(t0.j) + sfx + " * x" + std::to_string(t0.k) + sfx;
                if (b - a > 1) {
                    factor = "p" + std::to_string(prod_id++);
                    so << ind << "const f64 " << factor << " = " << pr << ";\n";
                } else factor = "(" + pr + ")";
                ++st.instr;
            }
            for (size_t q = a; q < b; ++q) {
                so << ind << coef_fma("k" + std::to_string(singles[q].row), singles[q].c, factor) << "\n";
                ++st.instr;
                ++st.coef;
            }
            a = b;
        }
    }
    if (hook_phase >= (int)phases.size()) hook(so);
}

// Derived monomials (rank-5 tensors) in the LDS-resident kernels: every derived value is one more LDS node behind the
// base ones.  The products are formed between two barriers right after a stage state has been published; a chain
// (q = x*x, r = q*x) stays inside one wavefront, the chains are spread over the wavefronts.
std::vector<std::vector<int>> lds_derived_shares(int nbase, const std::vector<std::pair<int, int>> &der, int W)
{
    const int nd = (int)der.size();
    std::vector<int> comp(nd);
    for (int n = 0; n < nd; ++n) comp[n] = n;
    std::function<int(int)> find = [&](int a) { return comp[a] == a ? a : comp[a] = find(comp[a]); };
    for (int n = 0; n < nd; ++n)
        for (int f : {der[n].first, der[n].second})
            if (f > nbase) comp[find(n)] = find(f - nbase - 1);
    std::map<int, std::vector<int>> groups;
    for (int n = 0; n < nd; ++n) groups[find(n)].push_back(n);
    std::vector<std::vector<int>> share(W);
    std::vector<std::pair<size_t, int>> order;
    for (auto &kv : groups) order.push_back({kv.second.size(), kv.first});
    std::sort(order.begin(), order.end(), [](const std::pair<size_t, int> &a, const std::pair<size_t, int> &b) {
        return a.first != b.first ? a.first > b.first : a.second < b.second;
    });
    for (auto &og : order) {
        int w = 0;
        for (int v = 1; v < W; ++v) if (share[v].size() < share[w].size()) w = v;
        for (int n : groups[og.second]) share[w].push_back(n);
    }
    for (auto &sv : share) std::sort(sv.begin(), sv.end());          // a derived value only refers to earlier ones
    return share;
}

// `value(node)`: expression reading a base node from LDS
void emit_lds_derived(std::ostringstream &o, const char *ind, int nbase, const std::vector<std::pair<int, int>> &der,
                      const std::vector<int> &mine, const std::function<std::string(int)> &value,
                      const std::function<std::string(int)> &slot)
{
    if (mine.empty()) return;
    o << ind << "{\n";
    auto operand = [&](int f) { return f > nbase ? "dq" + std::to_string(f) : value(f); };
    for (int n : mine) {
        const int id = nbase + 1 + n;
        o << ind << "    const f64 dq" << id << " = " << operand(der[n].first) << " * " << operand(der[n].second) << ";\n";
        o << ind << "    " << slot(id) << " = dq" << id << ";\n";
    }
    o << ind << "}\n";
}

// tend_kernel: the same kernel text, named qgs_spec_tendlds<W>, that leaves after the first tendency evaluation with
// f(y_in) in y_out (launched with a one-step grid and S = 1).  It is a kernel of its own because as a run-time mode of
// the stepper the extra exit path made the register allocator spill in the stepper (700 instead of 396 B of scratch,
// 60.9 instead of 55.1 ms for 65 536 members x 100 steps at ndim 228); a loop-free kernel built from the same phases
// spills far worse (the scheduler hoists the LDS reads of all phases: 14.6 KB of scratch, 30x slower) -- which is also why
// the exit is guarded by a run-time argument and not by something the compiler can prove.
// dense: general lower-triangular tableau (tab = b[S], a[S*S]), kernel qgs_spec_rkldsd<W>.  The input of stage q is
// P_q = y + dt * sum_{j<q} a_qj k_j: the next stage's input is completed in registers as before (its base is P_{st+1} instead
// of y), the partial sums of the stages after it are read-modify-written in a private global buffer pwork[workgroup][q][mode][64]
// (the LDS is full of stage state at these sizes; the buffer is L2 / Infinity-Cache resident).
void emit_rk_lds_kernel(std::ostringstream &out, int ndim, const std::vector<Row> &rows, const CodegenOptions &opt,
                        const std::vector<std::pair<int, int>> &der, bool tend_kernel = false, bool dense = false)
{
    const int W = opt.lds_waves, cap = std::max(2, opt.lds_cap);
    const int nnode = ndim + (int)der.size();
    const std::vector<std::vector<int>> dshare = lds_derived_shares(ndim, der, W);
    const std::function<std::string(int)> dval = [](int f) { return "xs[" + std::to_string(f - 1) + "][lane]"; };
    const std::string kname = std::string(tend_kernel ? "qgs_spec_tendlds" : (dense ? "qgs_spec_rkldsd" : "qgs_spec_rklds")) + std::to_string(W);
    RowTerms rt(ndim + 1);
    for (int i = 1; i <= ndim; ++i) {
        for (const Lin &l : rows[i].lin) rt[i].push_back({i, 0, l.k, l.c});
        for (const Bil &b : rows[i].bil) rt[i].push_back({i, std::min(b.j, b.k), std::max(b.j, b.k), b.c});
    }
    const std::vector<std::vector<int>> owns = lds_partition(ndim, nnode, rt, W, cap, opt.lds_group);
    const NodeFn node = [](int m) { return LdsNode{(int64_t)(m - 1) * 512, 0}; };
    // private buffers (step-start state, running sum, partial stage sums): a wavefront only ever touches its own rows, so they
    // are laid out [wavefront's rows, consecutively][64]: all of a wavefront's rows lie within +-4 KB of one or two base
    // addresses (the immediate offset range of global_load / global_store) instead of needing a 64-bit address per row
    std::vector<int> slot(ndim + 1, 0);
    {
        int q = 0;
        for (int w = 0; w < W; ++w) for (int d : owns[w]) slot[d] = q++;
    }
    std::ostringstream o;
    std::vector<KTable> tables(W);
    o << "\n// run-time stage count RK stepper, stage state in LDS, rows split over " << W << " wavefronts per 64 members,\n"
      << "// factors cached in registers per phase (cap " << cap << ")\n";
    o << "extern \"C\" __global__ void __launch_bounds__(" << 64 * W << ") " << kname << "(\n"
      << "    const f64* __restrict__ y_in,\n"
      << "    f64* __restrict__ y_out,        // final state, X[mode][member] (may be null)\n"
      << "    f64* __restrict__ ywork,        // private [workgroup][mode][64]: state at the start of the current step, re-read at\n"
      << "                                    // the end of every stage instead of being held in registers\n"
      << (dense ? "    f64* __restrict__ pwork,        // private [workgroup][stage][mode][64]: partial sums of the later stages' inputs\n" : "")
      << "    f64* __restrict__ rec, f64* __restrict__ stages,\n"
      << "    const f64* __restrict__ dtime, const f64* __restrict__ tab,\n"
      << "    i64 n_traj, i64 ld, i64 step_begin, i64 step_end, i64 write_steps, i64 n_records, int backward, int write_final, int S"
      << (tend_kernel ? ",\n    int tend_only)                   // always 1; a run-time value so that the stage loop stays a loop\n{\n" : ")\n{\n");
    o << "    __shared__ f64 xs[" << nnode << "][QGS_WAVE];";
    if (!der.empty()) o << "   // " << ndim << " variables + " << der.size() << " derived monomials";
    o << "\n";
    o << "    const int lane = threadIdx.x & 63;\n"
      << "    const unsigned lane8 = (unsigned)lane * 8u;\n"
      << "    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n"
      << "    const i64 m0 = (i64)blockIdx.x * QGS_WAVE + lane;\n"
      << "    const bool live = m0 < n_traj;\n"
      << "    const i64 m = live ? m0 : (n_traj - 1);\n"
      << "    f64* const yw = ywork + (i64)blockIdx.x * " << ndim * 64 << " + lane;   // row d of this member at yw[(d-1)*64]\n"
      << "    QGS_CLOCK_MARK(0)\n";
    if (dense) o << "    f64* const pw = pwork + (i64)blockIdx.x * S * " << ndim * 64 << " + lane;   // slot q of this workgroup at pw + q * " << ndim * 64 << "\n";
    LdsStats stats;
    for (int w = 0; w < W; ++w) {
        const std::vector<int> &own = owns[w];
        o << "    " << (w == 0 ? "if" : "else if") << " (wave == " << w << ") {   // rows:";
        for (int i : own) o << " " << i;
        o << "\n";
        const char *I2 = "        ", *I3 = "            ", *I4 = "                ";
        // acc<r>: running y + dt*sum b_i k_i; equals the state y at every step boundary
        for (int d : own) o << I2 << "f64 acc" << d << " = y_in[" << (d - 1) << " * ld + m];\n";
        for (int d : own) o << I2 << "xs[" << (d - 1) << "][lane] = acc" << d << "; yw[" << slot[d] * 64 << "] = acc" << d << ";\n";
        o << I2 << "__syncthreads();\n";
        if (!der.empty()) {
            emit_lds_derived(o, I2, ndim, der, dshare[w], dval, dval);
            o << I2 << "__syncthreads();\n";
        }
        o << I2 << "QGS_REC_INIT\n";
        o << I2 << "for (i64 ti = step_begin; ti < step_end; ++ti) {\n";
        o << I3 << "const f64 dt = dtime[ti + 1] - dtime[ti];\n";
        o << I3 << "if (ti == next_rec) {\n"
          << I4 << "i64 ldr = ld; asm volatile(\"\" : \"+s\"(ldr));   // keeps the row offsets out of the loop-invariant set\n"
          << I4 << "f64* p = rec + qgs_rec_index(iw, n_records, backward) * " << ndim << " * ldr + m;\n"
          << I4 << "++iw; next_rec += write_steps;\n"
          << I4 << "if (live) {\n";
        for (int d : own) o << I4 << "    p[" << (d - 1) << " * ldr] = " << "acc" << d << ";\n";
        o << I4 << "}\n" << I3 << "}\n";
        o << "#pragma nounroll\n";
        o << I3 << "for (int st = 0; st < S; ++st) {\n";
        o << I4 << "const bool last = (st == S - 1);\n";
        o << I4 << "const f64 hb = dt * tab[st];\n";
        if (dense) {
            o << I4 << "const f64 ha = last ? 0.0 : dt * tab[S + (st + 1) * S + st];\n"
              << I4 << "const f64* basep = (st == 0) ? yw : pw + (i64)(last ? st : st + 1) * " << ndim * 64 << ";   // P_{st+1}; P_1's base is y itself\n";
        } else o << I4 << "const f64 ha = last ? 0.0 : dt * tab[S + st];\n";
        // opaque per-stage values: the compiler must not hoist the re-reads of the step-start state out of the stage
        // loop (they would occupy registers for the whole step), nor turn the last-stage select into a branch that
        // sinks those loads to their use
        o << I4 << "const f64* ywp = " << (dense ? "basep" : "yw") << "; asm volatile(\"\" : \"+v\"(ywp));\n";
        o << I4 << "unsigned long long lastmask = last ? ~0ull : 0ull; asm volatile(\"\" : \"+v\"(lastmask));\n";
        o << I4 << "if (stages && live) {\n"
          << I4 << "    i64 ldr = ld; asm volatile(\"\" : \"+s\"(ldr));\n"
          << I4 << "    f64* sp = stages + ((ti - step_begin) * S + st) * " << ndim << " * ldr + m;\n";
        for (int d : own) o << I4 << "    sp[" << (d - 1) << " * ldr] = xs[" << (d - 1) << "][lane];\n";
        o << I4 << "}\n";
        g_ktab = &tables[w];
        o << I4 << "kf64* kt = (kf64*)" << kname << "_kt" << w << "; asm volatile(\"\" : \"+s\"(kt));\n";
        std::ostringstream so;
        std::vector<PTerm> terms;
        for (int i : own) {
            const Row &r = rows[i];
            if (r.has_c0 && r.c0 != 0.0) so << I4 << "f64 k" << i << " = " << lit(r.c0) << ";\n";
            else so << I4 << "f64 k" << i << " = 0.0;\n";
            terms.insert(terms.end(), rt[i].begin(), rt[i].end());
        }
        const std::vector<Phase> phases = build_phases(nnode, terms, cap);
        // step-start state of the own rows, consumed at the end of the stage
        const int hook_phase = std::max(0, (int)phases.size() - std::max(0, opt.lds_yload_ahead));
        emit_lds_phases(so, I4, phases, node, {"lane8"}, "(const char*)xs", opt.lds_group, hook_phase,
                        [&](std::ostringstream &h) {
                            for (int d : own) h << I4 << "const f64 yg" << d << " = ywp[" << slot[d] * 64 << "];\n";

                        }, stats, opt.lds_order);
        o << resolve_ktab(so.str(), tables[w], opt.lds_coeff_dedupe);
        g_ktab = nullptr;
        if (tend_kernel) {
            o << I4 << "if (tend_only) {          // uniform: every wavefront leaves here, nobody is left waiting at a barrier\n"
              << I4 << "    if (live) {\n";
            for (int d : own) o << I4 << "        y_out[" << (d - 1) << " * ld + m] = k" << d << ";\n";
            o << I4 << "    }\n" << I4 << "    return;\n" << I4 << "}\n";
        }
        if (dense) {
            // partial sums of the stages after the next one: P_q (+)= dt a_q,st k  (stage 0 starts them from y = yg)
            o << I4 << "for (int q = st + 2; q < S; ++q) {\n"
              << I4 << "    const f64 hq = dt * tab[S + q * S + st];\n"
              << I4 << "    f64* pq = pw + (i64)q * " << ndim * 64 << ";\n";
            for (int d : own)
                o << I4 << "    pq[" << slot[d] * 64 << "] = __builtin_fma(hq, k" << d << ", st == 0 ? yg" << d << " : pq[" << slot[d] * 64 << "]);\n";
            o << I4 << "}\n";
        }
        for (int d : own) {
            o << I4 << "acc" << d << " = __builtin_fma(hb, k" << d << ", acc" << d << ");\n";
            o << I4 << "k" << d << " = qgs_bitsel(lastmask, acc" << d << ", __builtin_fma(ha, k" << d << ", yg" << d << "));\n";
        }
        o << I4 << "__syncthreads();          // every wavefront is done reading the stage state\n";
        for (int d : own) o << I4 << "xs[" << (d - 1) << "][lane] = k" << d << ";\n";
        o << I4 << "__syncthreads();\n";
        if (!der.empty()) {                                  // derived monomials of the new stage state
            emit_lds_derived(o, I4, ndim, der, dshare[w], dval, dval);
            o << I4 << "__syncthreads();\n";
        }
        o << I3 << "}\n";
        // the new state is the start of the next step
        for (int d : own) o << I3 << "yw[" << slot[d] * 64 << "] = acc" << d << ";\n";
        o << I2 << "}\n";
        o << I2 << "if (live) {\n" << I3 << "if (y_out) {\n";
        for (int d : own) o << I4 << "y_out[" << (d - 1) << " * ld + m] = " << "acc" << d << ";\n";
        o << I3 << "}\n" << I3 << "if (write_final) {\n"
          << I4 << "f64* p = rec + qgs_rec_index(n_records - 1, n_records, backward) * " << ndim << " * ld + m;\n";
        for (int d : own) o << I4 << "p[" << (d - 1) << " * ld] = " << "acc" << d << ";\n";
        o << I3 << "}\n" << I2 << "}\n    }\n";
    }
    o << "    QGS_CLOCK_MARK(2)\n}\n";
    out << "// per stage and 64 members: " << stats.phases << " phases, " << stats.loads << " LDS reads, " << stats.instr
        << " fp64 instructions, " << stats.coef << " coefficient fetches\n";
    {
        size_t entries = 0;
        for (const KTable &t : tables) entries += t.vals.size();
        out << "// statement order " << opt.lds_order << ": " << entries << " coefficient table entries after de-duplication\n";
    }
    // (general-tableau flavour: built with -mllvm -disable-cgp, see kernel_compile_flags)
    for (int w = 0; w < W; ++w) emit_ktable(out, kname + "_kt" + std::to_string(w), tables[w]);
    out << o.str();
}

// LDS-resident tangent / adjoint model for large systems (same idea as the stepper above).  A workgroup of W wavefronts
// propagates 64 (member, column) pairs arranged as 16 members x 4 columns, so that the stage state of the 16 members
// (xs[mode][16], 28.5 KB at ndim 228) AND the tangent stage vector of the 64 pairs (ws[mode][64], 114 KB) fit the 160 KB
// LDS together (64 members x 1 column would need 2 x 114 KB).  Each wavefront owns a block of output rows of J w (or
// J^T w); its terms c * x_k * w_j are ordered into phases that cache <= cap LDS values in registers.
//   tangent  (J w)_i   = sum_{j,k} Tj_ijk x_k w_j        adjoint  (J^T w)_j = sum_{i,k} Tj_ijk x_k w_i
void emit_tgl_lds_kernel(std::ostringstream &out, int ndim, const std::vector<std::vector<WX>> &wx, bool adjoint,
                         const CodegenOptions &opt, const std::vector<std::pair<int, int>> &der)
{
    const int W = opt.lds_waves, cap = std::max(2, opt.lds_cap);
    // tile: MT members x (64 / MT) columns per workgroup.  16 x 4 by default; 8 x 8 halves the stage-state tile when the
    // derived monomials of a rank-5 model would not fit otherwise (dynamic-T MAOOAM 6x6: 230 variables + 118 monomials)
    const int MT = (opt.lds_tgl_members == 8) ? 8 : 16, MSH = (MT == 8) ? 3 : 4, NC = 64 / MT;
    const std::string sMT = std::to_string(MT), sMSK = std::to_string(MT - 1), sMSH = std::to_string(MSH), sNC = std::to_string(NC);
    const std::string kname = std::string(adjoint ? "qgs_spec_adjlds" : "qgs_spec_tgllds") + std::to_string(W) + (MT == 8 ? "m8" : "");
    const int nx = ndim + (int)der.size();                  // x nodes: the stage state and (rank 5) its derived monomials
    const int64_t xs_bytes = (int64_t)nx * MT * 8;
    const std::vector<std::vector<int>> dshare = lds_derived_shares(ndim, der, W);
    const std::function<std::string(int)> dval = [sMSK](int f) { return "xs[" + std::to_string(f - 1) + "][lane & " + sMSK + "]"; };
    RowTerms rt(ndim + 1);
    for (int i = 1; i <= ndim; ++i)
        for (const WX &t : wx[i]) {
            if (t.x == 0) rt[i].push_back({i, 0, t.w, t.c});                 // x_0 = 1: c * w_j
            else rt[i].push_back({i, t.w, ndim + t.x, t.c});                 // node w_j < node x_k
        }
    const std::vector<std::vector<int>> owns = lds_partition(ndim, ndim + nx, rt, W, cap, opt.lds_group);
    const NodeFn node = [ndim, xs_bytes, MT](int n) {
        return n <= ndim ? LdsNode{xs_bytes + (int64_t)(n - 1) * 512, 0} : LdsNode{(int64_t)(n - ndim - 1) * (MT * 8), 1};
    };
    // private step-start buffer laid out per wavefront behind an opaque base pointer, as in emit_rk_lds_kernel
    std::vector<int> slot(ndim + 1, 0);
    {
        int q = 0;
        for (int w = 0; w < W; ++w) for (int d : owns[w]) slot[d] = q++;
    }
    std::ostringstream o;
    std::vector<KTable> tables(W);
    o << "\n// " << (adjoint ? "adjoint" : "tangent") << " model, run-time stage count, " << MT << " members x " << NC << " columns per workgroup of " << W
      << " wavefronts,\n// stage state and tangent stage vector in LDS, factors cached in registers per phase (cap " << cap << ")\n";
    o << "extern \"C\" __global__ void __launch_bounds__(" << 64 * W << ") " << kname << "(\n"
      << "    const f64* __restrict__ w_in_p,  // F[mode][col][member] at step `step_begin`\n"
      << "    f64* __restrict__ w_out_p,       // after step `step_end-1` (may be null)\n"
      << "    f64* __restrict__ vwork,         // private [workgroup][mode][64]: tangent state at the start of the current step\n"
      << "    f64* __restrict__ rec,           // F[record][mode][col][member]\n"
      << "    const f64* __restrict__ stages,  // S[(step-step_begin)*S+stage][mode][member]\n"
      << "    const f64* __restrict__ dtime, const f64* __restrict__ tab,\n"
      << "    i64 n_traj, i64 ld, i64 n_tg, i64 step_begin, i64 step_end, i64 write_steps, i64 n_records,\n"
      << "    int backward, int write_final, f64 inverse, int S)\n{\n";
    o << "    __shared__ f64 lds_all[" << nx * MT + ndim * 64 << "];\n"
      << "    f64 (*xs)[" << MT << "] = (f64 (*)[" << MT << "])lds_all;                       // stage state of the " << MT << " members"
      << (der.empty() ? "" : " + derived monomials") << "\n"
      << "    f64 (*ws)[QGS_WAVE] = (f64 (*)[QGS_WAVE])(lds_all + " << nx * MT << ");   // tangent stage vector of the 64 pairs\n";
    o << "    const int lane = threadIdx.x & 63;\n"
      << "    const unsigned lane8 = (unsigned)lane * 8u, xl8 = (unsigned)(lane & " << sMSK << ") * 8u;\n"
      << "    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n"
      << "    const i64 L = n_tg * ld;\n"
      << "    const i64 mt = (i64)blockIdx.x * " << MT << ", m0 = mt + (lane & " << sMSK << "), c0 = (i64)blockIdx.y * " << NC << " + (lane >> " << sMSH << ");\n"
      << "    const bool live = (m0 < n_traj) && (c0 < n_tg);\n"
      << "    const i64 m = m0 < n_traj ? m0 : (n_traj - 1), col = c0 < n_tg ? c0 : (n_tg - 1);\n"
      << "    const i64 l = col * ld + m;                                  // this pair's lane of F[mode][col][member]\n"
      << "    f64* const vw = vwork + ((i64)blockIdx.y * gridDim.x + blockIdx.x) * " << ndim * 64 << " + lane;\n"
      << "    // stage-state tile of the " << MT << " members, loaded by the whole workgroup: element e -> (mode e / " << MT << ", member e % " << MT << ")\n"
      << "    i64 xm = mt + (threadIdx.x & " << sMSK << "); if (xm >= n_traj) xm = n_traj - 1;\n"
      << "#define QGS_LOAD_XS(sp) do { const f64* sp_ = (sp); \\\n"
      << "        for (int e = threadIdx.x; e < " << ndim * MT << "; e += " << 64 * W << ") xs[e >> " << sMSH << "][e & " << sMSK << "] = sp_[(i64)(e >> " << sMSH << ") * ld + xm]; } while (0)\n";
    LdsStats stats;
    for (int w = 0; w < W; ++w) {
        const std::vector<int> &own = owns[w];
        o << "    " << (w == 0 ? "if" : "else if") << " (wave == " << w << ") {   // rows:";
        for (int i : own) o << " " << i;
        o << "\n";
        const char *I2 = "        ", *I3 = "            ", *I4 = "                ";
        for (int d : own) o << I2 << "f64 acc" << d << " = w_in_p[" << (d - 1) << " * L + l];\n";
        for (int d : own) o << I2 << "ws[" << (d - 1) << "][lane] = acc" << d << "; vw[" << slot[d] * 64 << "] = acc" << d << ";\n";
        o << I2 << "if (step_begin < step_end) QGS_LOAD_XS(stages);\n";
        o << I2 << "__syncthreads();\n";
        if (!der.empty()) {
            emit_lds_derived(o, I2, ndim, der, dshare[w], dval, dval);
            o << I2 << "__syncthreads();\n";
        }
        o << I2 << "QGS_REC_INIT\n";
        o << I2 << "for (i64 ti = step_begin; ti < step_end; ++ti) {\n";
        o << I3 << "const f64 dt = dtime[ti + 1] - dtime[ti];\n";
        o << I3 << "if (ti == next_rec) {\n"
          << I4 << "i64 Lr = L; asm volatile(\"\" : \"+s\"(Lr));   // keeps the row offsets out of the loop-invariant set\n"
          << I4 << "f64* p = rec + qgs_rec_index(iw, n_records, backward) * " << ndim << " * Lr + l;\n"
          << I4 << "++iw; next_rec += write_steps;\n"
          << I4 << "if (live) {\n";
        for (int d : own) o << I4 << "    p[" << (d - 1) << " * Lr] = acc" << d << ";\n";
        o << I4 << "}\n" << I3 << "}\n";
        o << "#pragma nounroll\n";
        o << I3 << "for (int st = 0; st < S; ++st) {\n";
        o << I4 << "const bool last = (st == S - 1);\n";
        o << I4 << "const f64 hb = dt * tab[st] * inverse;\n";                      // inverse = +-1: exact
        o << I4 << "const f64 ha = last ? 0.0 : dt * tab[S + st] * inverse;\n";
        o << I4 << "const f64* vwp = vw; asm volatile(\"\" : \"+v\"(vwp));\n";
        o << I4 << "unsigned long long lastmask = last ? ~0ull : 0ull; asm volatile(\"\" : \"+v\"(lastmask));\n";
        g_ktab = &tables[w];
        o << I4 << "kf64* kt = (kf64*)" << kname << "_kt" << w << "; asm volatile(\"\" : \"+s\"(kt));\n";
        std::ostringstream so;
        std::vector<PTerm> terms;
        for (int i : own) {
            so << I4 << "f64 k" << i << " = 0.0;\n";
            terms.insert(terms.end(), rt[i].begin(), rt[i].end());
        }
        const std::vector<Phase> phases = build_phases(ndim + nx, terms, cap);
        const int hook_phase = std::max(0, (int)phases.size() - std::max(0, opt.lds_yload_ahead));
        emit_lds_phases(so, I4, phases, node, {"lane8", "xl8"}, "(const char*)lds_all", opt.lds_group, hook_phase,
                        [&](std::ostringstream &h) {
                            for (int d : own) h << I4 << "const f64 yg" << d << " = vwp[" << slot[d] * 64 << "];\n";
                        }, stats);
        o << resolve_ktab(so.str(), tables[w], opt.lds_coeff_dedupe);
        g_ktab = nullptr;
        for (int d : own) {
            o << I4 << "acc" << d << " = __builtin_fma(hb, k" << d << ", acc" << d << ");\n";
            o << I4 << "k" << d << " = qgs_bitsel(lastmask, acc" << d << ", __builtin_fma(ha, k" << d << ", yg" << d << "));\n";
        }
        o << I4 << "__syncthreads();          // every wavefront is done reading xs and ws\n";
        for (int d : own) o << I4 << "ws[" << (d - 1) << "][lane] = k" << d << ";\n";
        // stage state of the next stage (or of the first stage of the next step)
        o << I4 << "{\n"
          << I4 << "    const i64 nxt = (ti - step_begin) * S + st + 1;\n"
          << I4 << "    if (nxt < (step_end - step_begin) * S) QGS_LOAD_XS(stages + nxt * " << ndim << " * ld);\n"
          << I4 << "}\n";
        o << I4 << "__syncthreads();\n";
        if (!der.empty()) {                                  // derived monomials of the stage state just loaded
            emit_lds_derived(o, I4, ndim, der, dshare[w], dval, dval);
            o << I4 << "__syncthreads();\n";
        }
        o << I3 << "}\n";
        for (int d : own) o << I3 << "vw[" << slot[d] * 64 << "] = acc" << d << ";\n";
        o << I2 << "}\n";
        o << I2 << "if (live) {\n" << I3 << "if (w_out_p) {\n";
        for (int d : own) o << I4 << "w_out_p[" << (d - 1) << " * L + l] = acc" << d << ";\n";
        o << I3 << "}\n" << I3 << "if (write_final) {\n"
          << I4 << "f64* p = rec + qgs_rec_index(n_records - 1, n_records, backward) * " << ndim << " * L + l;\n";
        for (int d : own) o << I4 << "p[" << (d - 1) << " * L] = acc" << d << ";\n";
        o << I3 << "}\n" << I2 << "}\n    }\n";
    }
    o << "#undef QGS_LOAD_XS\n}\n";
    out << "// per stage and 64 (member, column) pairs: " << stats.phases << " phases, " << stats.loads << " LDS reads, " << stats.instr
        << " fp64 instructions, " << stats.coef << " coefficient fetches\n";
    // (built with -mllvm -disable-cgp, see kernel_compile_flags)
    for (int w = 0; w < W; ++w) emit_ktable(out, kname + "_kt" + std::to_string(w), tables[w]);
    out << o.str();
}

}  // namespace

// Batched Householder QR (dgeqr2 + dorg2r, the algorithm behind np.linalg.qr), fully unrolled for one shape, one workgroup per
// TILE OF M = 16 (or 8) CONSECUTIVE MEMBERS.  lane = (member mm = lane % M, column lane cc = lane / M), and lane (mm, cc) of
// wavefront w keeps, in registers, the columns c = L (s W + w) + cc of its member for the slots s = 0 .. P-1 (L = 64 / M column
// lanes, W wavefronts per workgroup: L P W >= n_cols).  A global access of a wavefront is then L whole runs of M members: with
// M = 16 every 128-byte line of A[row][col][m0 .. m0+15] belongs to one workgroup and moves in one instruction; with M = 8 a
// line is shared by two workgroups that the block index places next to each other on one XCD (one L2).  (Rounds 1-4 had one
// wavefront per member with lane = column: 36 lines of 8 bytes per instruction, the 16 members of a line spread over 16
// workgroups -- at 36 x 36 x 16 384 the L2 evicted part-written lines, WRITE_SIZE 5.8 x the matrix, 0.35 ms against a 0.043 ms
// HBM floor.)
//
// The pivot column of step j lives in the lanes (w_j, cc_j) of slot s_j, all compile-time constants; those M lanes form
// norm / beta / tau and publish v (and tau, 1 / (alpha - beta)) through a double-buffered LDS block vb[2][R + 2][M]; one barrier
// per step, after which every lane reads the M-member row v_i as one conflict-free ds_read_b64 (L lanes per address).
// LOOK-AHEAD: the wavefront that owns column j + 1 updates that slot first and forms and publishes pivot j + 1 in the same
// basic block as the update of its other slots, so the norm / sqrt / divisions are off the critical path when P > 1.  In the
// second phase (dorg2r) reflector j - 1 is published during step j: the reflectors are final by then.
// The arithmetic per column is statement for statement that of the one-wavefront kernel it replaces (same sums in the same
// order; `chains` > 1 splits the dot products into that many partial sums); the reflector is applied as
//   w = t (q_j + scale (v.q)),  q -= (w scale) v           with v unscaled, u = v * scale only ever formed in the pivot lanes.
// A wavefront none of whose columns is > j skips the step (uniform branch: no LDS reads); slots dead in every wavefront are
// not emitted.
QrPlan qr_plan(int n_rows, int n_cols, int members, int slots)
{
    // Which of the three layouts (codegen.h; measurements: profiles/r05_qr.md).  `members` / `slots` other than 0 are a developer
    // build's requests: members 4 = row design, 2 = grid design, 16 / 8 = tile design with that tile.
    const int row_slots = (n_cols + 15) / 16;
    const int row_regs = 2 * n_rows * row_slots + 24;          // row design: the matrices + temporaries (36 x 36: 216 + 12)
    auto row_plan = [&] {
        QrPlan p;
        p.members = 4; p.slots = row_slots; p.waves = 4; p.reload = false; p.chains = 1;
        // (row_regs 241 ... 256, e.g. 56 x 20 or 38 x 34, leave the compiler 12 ... 68 bytes of scratch at two wavefronts per SIMD: still
        // well ahead of one wavefront per SIMD, which has nothing to hide the latency of a dependent chain behind)
        p.one_wave_per_simd = row_regs > 256;
        return p;
    };
    auto grid_plan = [&] {
        // W wavefronts per member (4 W row groups), as few as keep the local rows x slots within ~110 registers
        QrPlan p;
        p.slots = row_slots;
        p.waves = 1;
        while (p.waves < 16 && 2 * ((n_rows + 4 * p.waves - 1) / (4 * p.waves)) * p.slots > 110) p.waves *= 2;
        p.row_groups = 4 * p.waves;
        p.members = std::max(1, std::min(4, 16 / p.waves));
        p.reload = false; p.chains = 1;
        return p;
    };
    if (members == 4 && row_regs <= 384) return row_plan();
    if (members == 2) return grid_plan();
    if (members == 0) {
        // 1. four matrices per wavefront from 13 columns on, while they fit the registers of a SIMD lane: two wavefronts per SIMD up to
        //    256 registers (n_rows x ceil(n_cols / 16) <= 116), one wavefront up to 384 with part of the matrices in accumulation
        //    registers (rows <= 64: beyond, the compiler's copies are the time).  16 384 matrices, against the next best design:
        //    36 x 36 0.13 ms (tile 0.215), 20 x 20 0.041 (tile 0.057), 36 x 20 0.072 (tile 0.089), 48 x 20 0.092 (grid 0.186),
        //    100 x 16 0.146 (grid 0.271), 40 x 40 0.24 (tile 0.42), 48 x 48 0.38 (tile 0.60), 64 x 20 0.19 (grid 0.27), 60 x 30 0.26
        //    (grid 0.39), 56 x 40 0.41 (grid 0.60), 60 x 44 0.54 (grid 0.67).  Thinner ones stay with the tile design (36 x 10 0.039
        //    against 0.034); 52 x 52 would want 440 registers and spills.
        if (n_cols > 12 && (row_regs <= 256 || (row_regs <= 384 && n_rows <= 64))) return row_plan();
        // 2. tall matrices, thin ones from 39 rows, and what is left up to 48 columns (64 x 40: 0.64 against the tile design's 0.97,
        //    64 x 48: 0.86 against 1.15; from 49 columns on the tile design is ahead: 52 x 52 0.71 against 0.89, 64 x 64 1.50 against 2.22)
        if (n_rows > 64 || (n_rows > 38 && n_cols <= 48)) return grid_plan();
    }
    if (n_rows > 64) return grid_plan();
    // 3. tile design.  Registers a lane needs: 2 R per slot for the columns + 2 R for the reflector + temporaries; what it may use: the
    //    512 of a SIMD lane shared by the wavefronts of one workgroup on that SIMD, at most 256
    auto make = [&](int M, int P, QrPlan &p) {
        const int L = 64 / M;
        p.members = M;
        p.slots = std::max(1, std::min(P, (n_cols + L - 1) / L));
        p.waves = (n_cols + L * p.slots - 1) / (L * p.slots);
        if (p.waves > 16) return false;
        p.slots = (n_cols + L * p.waves - 1) / (L * p.waves);         // (no slot that is empty in every wavefront)
        const int cap = std::min(256, 512 / ((p.waves + 3) / 4));
        p.reload = false;
        return 2 * n_rows * (p.slots + 1) + 30 <= cap;
    };
    QrPlan p;
    const int m_lo = (members == 8 || members == 16) ? members : 16, m_hi = (members == 8 || members == 16) ? members : 8;
    for (int M = m_lo; M >= m_hi; M -= 8)
        for (int P = slots > 0 ? slots : 4; P >= (slots > 0 ? slots : 1); --P)
            if (make(M, P, p)) return p;
    // nothing holds columns and reflector at once: one column per lane, the reflector read from LDS twice per step
    make(m_hi, slots > 0 ? slots : 1, p);
    p.reload = true;
    return p;
}

std::string qr_plan_signature(const QrPlan &p)
{
    std::ostringstream s;
    s << "m" << p.members << "p" << p.slots << "w" << p.waves << "c" << p.chains << "r" << (p.reload ? 1 : 0);
    if (p.row_groups > 0) s << "g" << p.row_groups;
    if (p.one_wave_per_simd) s << "o1";
    return s.str();
}

// `acc += (lane cc of the 16-lane row of src) * y` as `v_fmac_f64_dpp ... row_newbcast:cc` (full rate on gfx950), for the row and grid
// designs of the batched QR.  The compiler has no DPP form of the fp64 FMA to offer (`__builtin_amdgcn_update_dpp` on a double becomes
// a v_mov_b64_dpp in front of a plain FMA: twice the instructions), so these are inline assembly -- and the compiler does not look
// inside inline assembly for the hazard every DPP instruction has: a VGPR written by a VALU instruction must not be read as the DPP
// operand within the next two wait states.  Our own instructions never do that (checked below), but the register allocator may put
// a copy of `src` right in front of a statement (v_accvgpr_read_b32 out of the accumulation registers in the one-wavefront-per-SIMD
// kernels, a v_mov where it splits a live range): seen as wrong factors in a 64 x 20 developer plan, 113 such places.  So every
// statement starts with `s_nop 1`, whatever the allocator did before it, and holds a RUN of up to eight instructions, which makes that
// one wait per run instead of one per instruction: all operands of a statement are in their registers when it starts, and nothing of
// the compiler's comes between its instructions.
struct DppOp {
    std::string acc, src, y;
};
static void emit_dpp_fmacs(std::ostream &o, const std::string &ind, const std::vector<DppOp> &ops, int cc, size_t run = 8)
{
    for (size_t b = 0; b < ops.size(); b += run) {
        const size_t e = std::min(ops.size(), b + run);
        std::vector<std::string> outs, ins;
        auto index_of = [](const std::vector<std::string> &v, const std::string &n) {
            for (size_t k = 0; k < v.size(); ++k) if (v[k] == n) return (int)k;
            return -1;
        };
        for (size_t k = b; k < e; ++k) if (index_of(outs, ops[k].acc) < 0) outs.push_back(ops[k].acc);
        for (size_t k = b; k < e; ++k)
            for (const std::string *n : {&ops[k].src, &ops[k].y})
                if (index_of(outs, *n) < 0 && index_of(ins, *n) < 0) ins.push_back(*n);
        auto ref = [&](const std::string &n) {
            const int a = index_of(outs, n);
            return "%" + std::to_string(a >= 0 ? a : (int)outs.size() + index_of(ins, n));
        };
        o << ind << "asm volatile(\"s_nop 1";
        for (size_t k = b; k < e; ++k) {
            // (our own hazard: the DPP operand written by one of the two instructions before it)
            if ((k > b && ops[k - 1].acc == ops[k].src) || (k > b + 1 && ops[k - 2].acc == ops[k].src)) o << "\\n\\ts_nop 1";
            o << "\\n\\tv_fmac_f64_dpp " << ref(ops[k].acc) << ", " << ref(ops[k].src) << ", " << ref(ops[k].y) << " row_newbcast:" << cc
              << " row_mask:0xf bank_mask:0xf";
        }
        o << "\" :";
        for (size_t k = 0; k < outs.size(); ++k) o << (k ? ", " : " ") << "\"+v\"(" << outs[k] << ")";
        o << " :";
        for (size_t k = 0; k < ins.size(); ++k) o << (k ? ", " : " ") << "\"v\"(" << ins[k] << ")";
        o << ");\n";
    }
}

// Batched Householder QR, GRID design (plan.row_groups > 0): matrices too tall for the registers of one wavefront (rows > 64 ... 300,
// e.g. the 228 x n_vec bases of MAOOAM 6x6).  A member's matrix is spread over W wavefronts: lane = (row group g = 4 (wavefront % W) +
// lane / 16, column lane cc = lane % 16); group g keeps the rows g, g + NG, g + 2 NG, ... (NG = 4 W groups, dealt cyclically so that
// every group stays busy as the factorisation moves down) of the columns base_s + cc of every slot s.  Inside a group the pivot
// column reaches the other columns' lanes as in the row design (`v_fmac_f64_dpp row_newbcast`); what crosses groups is one number
// per column and step -- the dot product v.a_c -- summed across the four groups of a wavefront with two `__shfl_xor` and across the
// W wavefronts through a small LDS block, together with row j itself and the pivot's norm: one barrier per step, two LDS buffers.
// beta, tau and 1 / (alpha - beta) are formed by every lane from the same numbers in the same order, so they need no broadcast.
// (The round-1 kernel it replaces kept the whole matrix in LDS with ONE wavefront per member -- one wavefront per CU at 228 rows:
// 17 ms for 4 096 matrices of 228 x 40.)
// MW members share a workgroup (W MW <= 16 wavefronts); the matrices enter and leave through an LDS tile of NG rows at a time,
// towards global memory in runs of MW members per (row, column).
static GeneratedKernel generate_qr_grid_kernel(int n_rows, int n_cols, const QrPlan &plan)
{
    const int R = n_rows, C = n_cols, K = std::min(R, C), P = plan.slots, W = plan.waves, MW = plan.members, NG = plan.row_groups;
    if (NG != 4 * W || 16 * P < C || P < 1 || W < 1 || MW < 1 || W * MW > 16) throw std::runtime_error("batched QR: bad grid plan");
    const int L = (R + NG - 1) / NG;                  // local rows per group
    const int rem = C % 16;
    std::vector<int> base(P), width(P);
    for (int s = 0; s < P; ++s) {
        if (rem && s == P - 1) { base[s] = 0; width[s] = rem; }
        else { base[s] = rem + 16 * s; width[s] = 16; }
    }
    auto slot_of = [&](int c) { return c < rem ? P - 1 : (c - rem) / 16; };
    auto lane_of = [&](int c) { return c < rem ? c : (c - rem) % 16; };
    std::ostringstream o;
    const std::string I2 = "        ", I3 = "            ";
    auto q = [](int s, int l) { return "q" + std::to_string(s) + "_" + std::to_string(l); };
    auto fmac_b = [&](const std::string &ind, const std::string &acc, const std::string &src, const std::string &y, int cc) {
        emit_dpp_fmacs(o, ind, {{acc, src, y}}, cc);
    };
    // the same over the local rows l0 + 1 .. L - 1: acc(l) += (pivot lane of src(l)) * y(l)
    auto fmac_rows = [&](const std::string &ind, int l0, int cc, const std::function<DppOp(int)> &op) {
        std::vector<DppOp> ops;
        for (int l = l0 + 1; l < L; ++l) ops.push_back(op(l));
        emit_dpp_fmacs(o, ind, ops, cc);
    };
    auto live_slots = [&](int j, int sj) {
        std::vector<int> v;
        for (int s = 0; s < P; ++s) if (base[s] + width[s] - 1 > j && s != sj) v.push_back(s);
        if (base[sj] + width[sj] - 1 > j) v.push_back(sj);
        return v;
    };
    auto right_of = [&](int s, int j) -> std::string {
        if (base[s] > j) return "";
        return "(cc > " + std::to_string(j - base[s]) + ")";
    };
    const int TP = MW + 1;                            // member pitch of the tile (odd: conflict-free for MW = 4)
    o << "#ifndef QGS_SPEC_PRELUDE\n#define QGS_SPEC_PRELUDE\n" << PRELUDE << RECORD_HELPERS << "#endif\n";
    o << "// A[row][col][member] -> Q in place, diag(R) -> rdiag[col][member]; " << R << " x " << C << ": " << W << " wavefronts per member, " << NG
      << " row groups of " << L << " rows,\n// " << P << " column(s) per lane, " << MW << " member(s) per workgroup (" << qr_plan_signature(plan) << ")\n";
    o << "__device__ __forceinline__ f64 qgs_sum4(f64 x)      // sum over the four row groups of a wavefront, the same bits in all of them\n{\n"
      << "    x += __shfl_xor(x, 16);\n    x += __shfl_xor(x, 32);\n    return x;\n}\n";
    o << "extern \"C\" __global__ void __launch_bounds__(" << 64 * W * MW << ") qgs_spec_qr_" << R << "x" << C
      << "(f64* __restrict__ a, f64* __restrict__ rdiag, i64 n_traj, i64 ld)\n{\n";
    o << "    __shared__ f64 red[2][" << MW << "][" << W << "][" << P + 1 << "][16];   // per wavefront: partial v.a_c per slot and column lane; [P][0]: partial |x|^2\n"
      << "    __shared__ f64 piv[2][" << MW << "][" << P << "][16];            // row j of every slot (the pivot lane's entry: alpha, or tau in phase two)\n"
      << "    __shared__ f64 tile[" << NG * C << "][" << TP << "];\n";
    o << "    QGS_CLOCK_MARK(0)\n";
    o << "    const int tid = threadIdx.x, cc = tid & 15;\n"
      << "    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);\n"
      << "    const int mw = wave / " << W << ", ww = wave % " << W << ";           // member of the workgroup, wavefront of the member\n"
      << "    const int g = 4 * ww + ((tid >> 4) & 3);                         // row group: rows g, g + " << NG << ", ...\n"
      << "    const i64 m0 = (i64)blockIdx.x * " << MW << ";\n"
      << "    const bool live = m0 + mw < n_traj;\n"
      << "    // global side: lane = (member tm, pair tp): pass k moves the (row, column) pairs " << (64 * W) << " k + tp of a tile\n"
      << "    const int tm = tid % " << MW << ", tp = tid / " << MW << ";\n"
      << "    const bool tlive = m0 + tm < n_traj;\n";
    for (int s = 0; s < P; ++s) {
        o << "    f64";
        for (int l = 0; l < L; ++l) o << (l ? ", " : " ") << q(s, l);
        o << ";\n";
    }
    const int PASS = 64 * W;                          // pairs per pass (threads / MW)
    for (int l = 0; l < L; ++l) {                     // ---- in: local row l = the NG rows l NG .. of every group
        const int rows = std::min(NG, R - l * NG), pairs = rows * C, passes = (pairs + PASS - 1) / PASS;
        o << "    {   // rows " << l * NG << " .. " << l * NG + rows - 1 << " in\n"
          << I2 << "i64 ldw = ld; asm volatile(\"\" : \"+s\"(ldw));\n"
          << I2 << "const f64* const gp = a + (i64)tp * ldw + m0 + (tlive ? tm : 0);\n";
        for (int k = 0; k < passes; ++k) {
            const bool guard = PASS * k + PASS - 1 >= pairs;
            o << I2 << (guard ? "if (tp < " + std::to_string(pairs - PASS * k) + ") " : "") << "tile[" << PASS * k << " + tp][tm] = tlive ? gp[(i64)"
              << (l * NG * C + PASS * k) << " * ldw] : 0.0;\n";
        }
        o << I2 << "__syncthreads();\n";
        for (int s = 0; s < P; ++s) {
            o << I2 << q(s, l) << " = (" << (rows < NG ? "g < " + std::to_string(rows) + " && " : std::string())
              << (width[s] < 16 ? "cc < " + std::to_string(width[s]) : std::string("true")) << ") ? tile[g * " << C << " + " << base[s] << " + cc][mw] : 0.0;\n";
        }
        o << I2 << "__syncthreads();\n    }\n";
    }
    int step = 0;
    // the partial sums of step (j, slots): own rows below j of the pivot column times own rows of slot s
    auto partial_dots = [&](int j, int sj, int ccj, const std::vector<int> &slots, bool with_norm) {
        const int l0 = j / NG, gj = j % NG;
        o << I2 << "const bool below = g > " << gj << ";              // this group's row " << l0 << " is below row " << j << "\n";
        if (with_norm) {
            o << I2 << "f64 xn2 = 0.0;\n" << I2 << "{\n" << I3 << "const f64 e = below ? " << q(sj, l0) << " : 0.0;\n" << I3 << "xn2 = e * e;\n" << I2 << "}\n";
            for (int l = l0 + 1; l < L; ++l) o << I2 << "xn2 = __builtin_fma(" << q(sj, l) << ", " << q(sj, l) << ", xn2);\n";
        }
        for (int s : slots) {
            o << I2 << "f64 sd" << s << " = 0.0;\n";
            // (row l0 counts for the groups below the pivot row only)
            o << I2 << "{\n" << I3 << "f64 e = 0.0;\n";
            fmac_b(I3, "e", q(sj, l0), q(s, l0), ccj);
            o << I3 << "sd" << s << " = below ? e : 0.0;\n" << I2 << "}\n";
            fmac_rows(I2, l0, ccj, [&](int l) { return DppOp{"sd" + std::to_string(s), q(sj, l), q(s, l)}; });
        }
    };
    auto publish = [&](int j, int sj, int ccj, const std::vector<int> &slots, int B, bool with_norm) {
        const int l0 = j / NG, gj = j % NG;
        if (with_norm) o << I2 << "xn2 = qgs_sum4(xn2);\n";
        for (int s : slots) o << I2 << "sd" << s << " = qgs_sum4(sd" << s << ");\n";
        o << I2 << "if ((tid & 48) == 0) {                    // the wavefront's sums, once\n";
        for (int s : slots) o << I3 << "red[" << B << "][mw][ww][" << s << "][cc] = sd" << s << ";\n";
        if (with_norm) o << I3 << "if (cc == " << ccj << ") red[" << B << "][mw][ww][" << P << "][0] = xn2;\n";
        o << I2 << "}\n";
        o << I2 << "if (g == " << gj << ") {                     // row " << j << "\n";
        for (int s = 0; s < P; ++s) {
            const bool needed = std::find(slots.begin(), slots.end(), s) != slots.end() || s == sj;
            if (needed) o << I3 << "piv[" << B << "][mw][" << s << "][cc] = " << q(s, l0) << ";\n";
        }
        o << I2 << "}\n" << I2 << "__syncthreads();\n";
    };
    auto gather = [&](const std::string &name, int B, int s) {       // sum of the W wavefronts' partials, fixed order
        o << I2 << "f64 " << name << " = red[" << B << "][mw][0][" << s << "][cc];\n";
        for (int w = 1; w < W; ++w) o << I2 << name << " += red[" << B << "][mw][" << w << "][" << s << "][cc];\n";
    };
    // q -= (...) v on own rows below j, the pivot row itself in its group
    auto update = [&](int j, int sj, int ccj, int s, const std::string &nw, const std::string &wv) {
        const int l0 = j / NG, gj = j % NG;
        o << I2 << "{\n" << I3 << "f64 e = " << q(s, l0) << ";\n";
        fmac_b(I3, "e", q(sj, l0), nw, ccj);
        o << I3 << q(s, l0) << " = below ? e : ((g == " << gj << ") ? " << q(s, l0) << " - " << wv << " : " << q(s, l0) << ");\n" << I2 << "}\n";
        fmac_rows(I2, l0, ccj, [&](int l) { return DppOp{q(s, l), q(sj, l), nw}; });
    };
    for (int j = 0; j < K; ++j) {                   // ---- dgeqr2
        const int sj = slot_of(j), ccj = lane_of(j), l0 = j / NG, gj = j % NG, B = step & 1;
        const std::vector<int> slots = (j + 1 < C) ? live_slots(j, sj) : std::vector<int>();
        o << "    {   // column " << j << "\n";
        partial_dots(j, sj, ccj, slots, true);
        publish(j, sj, ccj, slots, B, true);
        o << I2 << "f64 xs = red[" << B << "][mw][0][" << P << "][0];\n";
        for (int w = 1; w < W; ++w) o << I2 << "xs += red[" << B << "][mw][" << w << "][" << P << "][0];\n";
        o << I2 << "const f64 alpha = piv[" << B << "][mw][" << sj << "][" << ccj << "];\n"
          << I2 << "f64 t = 0.0, beta = alpha, scale = 0.0;\n"
          << I2 << "if (xs != 0.0) {\n"
          << I2 << "    beta = -__builtin_copysign(__builtin_sqrt(__builtin_fma(alpha, alpha, xs)), alpha);\n"
          << I2 << "    t = (beta - alpha) / beta;\n"
          << I2 << "    scale = 1.0 / (alpha - beta);\n"
          << I2 << "}\n";
        for (int s : slots) {
            const std::string S = std::to_string(s), ro = right_of(s, j);
            gather("sum" + S, B, s);
            o << I2 << "f64 wv" << S << " = t * __builtin_fma(scale, sum" << S << ", piv[" << B << "][mw][" << s << "][cc]);\n";
            if (!ro.empty()) o << I2 << "wv" << S << " = " << ro << " ? wv" << S << " : 0.0;\n";
            o << I2 << "f64 nw" << S << " = -(wv" << S << " * scale);\n";
        }
        for (int s : slots) update(j, sj, ccj, s, "nw" + std::to_string(s), "wv" + std::to_string(s));
        // the pivot lanes keep u = v * scale below the diagonal, tau on it; diag(R) leaves
        o << I2 << "if (cc == " << ccj << ") {\n"
          << I3 << q(sj, l0) << " = below ? " << q(sj, l0) << " * scale : ((g == " << gj << ") ? t : " << q(sj, l0) << ");\n";
        for (int l = l0 + 1; l < L; ++l) o << I3 << q(sj, l) << " *= scale;\n";
        o << I3 << "if (g == " << gj << " && live) rdiag[(i64)" << j << " * ld + m0 + mw] = beta;\n";
        o << I2 << "}\n    }\n";
        ++step;
    }
    for (int j = K - 1; j >= 0; --j) {              // ---- dorg2r: tau_j sits on the diagonal of the pivot lanes
        const int sj = slot_of(j), ccj = lane_of(j), l0 = j / NG, gj = j % NG, B = step & 1;
        const std::vector<int> slots = live_slots(j, sj);
        o << "    {   // Q: reflector " << j << "\n";
        o << I2 << "const bool below = g > " << gj << ";\n";
        if (!slots.empty()) {
            // (partial_dots declares `below` itself: emit its body without the declaration)
            for (int s : slots) {
                o << I2 << "f64 sd" << s << " = 0.0;\n" << I2 << "{\n" << I3 << "f64 e = 0.0;\n";
                fmac_b(I3, "e", q(sj, l0), q(s, l0), ccj);
                o << I3 << "sd" << s << " = below ? e : 0.0;\n" << I2 << "}\n";
                fmac_rows(I2, l0, ccj, [&](int l) { return DppOp{"sd" + std::to_string(s), q(sj, l), q(s, l)}; });
            }
            publish(j, sj, ccj, slots, B, false);
            o << I2 << "const f64 t = piv[" << B << "][mw][" << sj << "][" << ccj << "];\n";
            for (int s : slots) {
                const std::string S = std::to_string(s), ro = right_of(s, j);
                gather("sum" + S, B, s);
                o << I2 << "f64 wv" << S << " = t * (piv[" << B << "][mw][" << s << "][cc] + sum" << S << ");\n";
                if (!ro.empty()) o << I2 << "wv" << S << " = " << ro << " ? wv" << S << " : 0.0;\n";
                o << I2 << "const f64 nw" << S << " = -wv" << S << ";\n";
            }
            for (int s : slots) update(j, sj, ccj, s, "nw" + std::to_string(s), "wv" + std::to_string(s));
            ++step;
        }
        // column j of Q = H_j e_j: (0 .. 0, 1 - t, -t u); tau is read from the pivot lane of the pivot row's group by DPP + shuffle-free:
        // every group needs it, so it travels through LDS when the step had no broadcast of its own
        if (slots.empty()) {
            o << I2 << "if (g == " << gj << " && cc == " << ccj << ") piv[" << B << "][mw][" << sj << "][" << ccj << "] = " << q(sj, l0) << ";\n"
              << I2 << "__syncthreads();\n"
              << I2 << "const f64 t = piv[" << B << "][mw][" << sj << "][" << ccj << "];\n";
            ++step;
        }
        o << I2 << "if (cc == " << ccj << ") {\n";
        for (int l = 0; l < l0; ++l) o << I3 << q(sj, l) << " = 0.0;\n";
        o << I3 << q(sj, l0) << " = below ? " << q(sj, l0) << " * -t : ((g == " << gj << ") ? 1.0 - t : 0.0);\n";
        for (int l = l0 + 1; l < L; ++l) o << I3 << q(sj, l) << " *= -t;\n";
        o << I2 << "}\n    }\n";
    }
    for (int l = 0; l < L; ++l) {                     // ---- out
        const int rows = std::min(NG, R - l * NG), pairs = rows * C, passes = (pairs + PASS - 1) / PASS;
        o << "    {   // rows " << l * NG << " .. " << l * NG + rows - 1 << " out\n"
          << I2 << "i64 ldw = ld; asm volatile(\"\" : \"+s\"(ldw));\n"
          << I2 << "f64* const gp = a + (i64)tp * ldw + m0 + (tlive ? tm : 0);\n";
        for (int s = 0; s < P; ++s)
            o << I2 << "if (" << (rows < NG ? "g < " + std::to_string(rows) + " && " : std::string())
              << (width[s] < 16 ? "cc < " + std::to_string(width[s]) : std::string("true")) << ") tile[g * " << C << " + " << base[s] << " + cc][mw] = " << q(s, l) << ";\n";
        o << I2 << "__syncthreads();\n";
        for (int k = 0; k < passes; ++k) {
            const bool guard = PASS * k + PASS - 1 >= pairs;
            o << I2 << "if (tlive" << (guard ? " && tp < " + std::to_string(pairs - PASS * k) : "") << ") gp[(i64)" << (l * NG * C + PASS * k)
              << " * ldw] = tile[" << PASS * k << " + tp][tm];\n";
        }
        o << I2 << "__syncthreads();\n    }\n";
    }
    o << "    QGS_CLOCK_MARK(2)\n}\n";
    GeneratedKernel gk;
    gk.source = o.str();
    return gk;
}

// Batched Householder QR, ROW design (plan.members == 4): lane = (member = lane / 16, column lane cc = lane % 16), and a lane keeps
// the columns c = 16 s + cc (s < P = ceil(n_cols / 16)) of its member: ONE wavefront holds four whole matrices.  The 16 lanes
// of a member are one DPP row, so the pivot column never leaves the registers: every product with v_i takes it from the pivot
// lane with `v_fmac_f64_dpp ... row_newbcast:cc_j` (full-rate on gfx950, profiles/r01_dpp_coefficients.txt), and tau / 1 / (alpha
// - beta) travel the same way.  No LDS, no barrier, no wait inside the factorisation: a straight line of fp64 VALU instructions.
// The tile design needs a barrier, an LDS round trip and 64 lanes x 8 bytes of LDS reads per wavefront and row in every one of
// its 2 (n_cols - 1) steps, and those -- not the arithmetic -- are its time (profiles/r05_qr.md).
// Four wavefronts (16 consecutive members) form a workgroup, and the matrices enter and leave through an LDS tile of `rc` rows
// at a time: towards global memory the workgroup is laid out like the tile design (lane = (member of 16, column group)), so every
// global access is whole 128-byte lines; towards the registers each wavefront reads / writes its own (4 members x 16 columns)
// view of the tile (row pitch 17 doubles: conflict-free both ways).
// Per step the slots are processed one after the other, the pivot's own slot last (its registers are the v of the others): one
// dot chain and one update in flight, which is what keeps the kernel inside 256 registers with 216 of them holding the matrices.
// Arithmetic: statement for statement that of the tile design (same sums in the same order, fused the same way).
static GeneratedKernel generate_qr_row_kernel(int n_rows, int n_cols, const QrPlan &plan)
{
    const int R = n_rows, C = n_cols, K = std::min(R, C), P = plan.slots;
    if (16 * P < C || P < 1) throw std::runtime_error("batched QR: the row plan does not cover the columns");
    // Which columns a slot holds.  When n_cols is not a multiple of 16 the short slot takes the FIRST n_cols % 16 columns, not the
    // last: a column is done with the first phase after its own step and enters the second phase only below its own index, so a
    // slot of low columns is live for a few steps (36 x 36: columns 0 .. 3 for 3 + 3 steps instead of columns 32 .. 35 for 34 +
    // 35) -- a quarter fewer dot / update instructions for the same work.
    const int rem = C % 16;
    std::vector<int> base(P), width(P);
    for (int s = 0; s < P; ++s) {
        if (rem && s == P - 1) { base[s] = 0; width[s] = rem; }
        else { base[s] = rem + 16 * s; width[s] = 16; }
    }
    auto slot_of = [&](int c) { return c < rem ? P - 1 : (c - rem) / 16; };
    auto lane_of = [&](int c) { return c < rem ? c : (c - rem) % 16; };
    std::ostringstream o;
    const std::string I1 = "    ", I2 = "        ", I3 = "            ";
    auto q = [](int s, int i) { return "q" + std::to_string(s) + "_" + std::to_string(i); };
    // sd = sum_{i > j} v_i q_i with v_i from lane cc_j: one chain, or plan.chains partial sums over interleaved rows
    const int NCH = std::max(1, std::min(4, plan.chains));
    auto dot_b = [&](int s, int sj, int j, int ccj) {
        const int n = R - j - 1, nch = std::max(1, std::min(NCH, n));
        o << I3 << "f64 sd = 0.0";
        for (int k = 1; k < nch; ++k) o << ", sd" << k << " = 0.0";
        o << ";\n";
        std::vector<DppOp> ops;
        for (int i = j + 1; i < R; ++i) {
            const int k = (i - j - 1) % nch;
            ops.push_back({k ? "sd" + std::to_string(k) : std::string("sd"), q(sj, i), q(s, i)});
        }
        emit_dpp_fmacs(o, I3, ops, ccj);
        if (nch == 2) o << I3 << "sd += sd1;\n";
        else if (nch == 3) o << I3 << "sd = (sd + sd1) + sd2;\n";
        else if (nch == 4) o << I3 << "sd = (sd + sd1) + (sd2 + sd3);\n";
    };
    // q_i += (v_i from lane cc_j) * nw on the rows below j
    auto update_b = [&](int s, int sj, int j, int ccj) {
        std::vector<DppOp> ops;
        for (int i = j + 1; i < R; ++i) ops.push_back({q(s, i), q(sj, i), "nw"});
        emit_dpp_fmacs(o, I3, ops, ccj);
    };
    auto live_slots = [&](int j, int sj) {              // slots with a column > j, the pivot's own slot last
        std::vector<int> v;
        for (int s = 0; s < P; ++s) if (base[s] + width[s] - 1 > j && s != sj) v.push_back(s);
        if (base[sj] + width[sj] - 1 > j) v.push_back(sj);
        return v;
    };
    // "this lane's column of slot s is > j": a comparison of the column lane with a constant (empty: always)
    // (`QGS_CC` re-derives the column lane from threadIdx.x behind an opaque copy every time: compared as a plain `cc`, the 2 x 36
    // lane masks are loop-invariant for the compiler, which keeps them in SGPR pairs, runs out of SGPRs and spills them into lanes
    // of VGPRs the matrices need)
    auto right_of = [&](int s, int j) -> std::string {
        if (base[s] > j) return "";
        return "(QGS_CC() > " + std::to_string(j - base[s]) + ")";
    };
    // rows per LDS tile: (row, column) pairs of a tile x 17 doubles, within 32 KB
    const int RC = std::max(1, std::min(R, (32 * 1024) / (C * 17 * 8)));
    o << "#ifndef QGS_SPEC_PRELUDE\n#define QGS_SPEC_PRELUDE\n" << PRELUDE << RECORD_HELPERS << "#endif\n";
    o << "// A[row][col][member] -> Q in place, diag(R) -> rdiag[col][member]; " << R << " x " << C << ", four members per wavefront,\n"
      << "// lane = (member, column lane of 16), " << P << " column(s) per lane (" << qr_plan_signature(plan) << "); in and out through an LDS tile of "
      << RC << " rows\n";
    o << "#ifdef QGS_QR_PROFILE\n#define QGS_QR_MARK(k) if (threadIdx.x == 0) prof[(i64)blockIdx.x * 160 + (k)] = (k) < 8 ? wall_clock64() : __builtin_amdgcn_s_memtime();\n"
      << "#define QGS_QR_PROF_ARG , unsigned long long* prof\n#else\n#define QGS_QR_MARK(k)\n#define QGS_QR_PROF_ARG\n#endif\n";
    o << "extern \"C\" __global__ void __launch_bounds__(256, " << (plan.one_wave_per_simd ? 1 : 2) << ") qgs_spec_qr_" << R << "x" << C
      << "(f64* __restrict__ a, f64* __restrict__ rdiag, i64 n_traj, i64 ld QGS_QR_PROF_ARG)\n{\n";
    o << "    __shared__ f64 tile[" << RC * C << "][17];          // [(row in the tile) * " << C << " + column][member of the workgroup's 16]\n";
    o << "    QGS_CLOCK_MARK(0)\n";
    o << "#define QGS_TID() ({ unsigned t_ = threadIdx.x; asm volatile(\"\" : \"+v\"(t_)); t_; })\n"
      << "#define QGS_CC() ((int)(QGS_TID() & 15u))\n";
    o << "    const i64 m0 = (i64)blockIdx.x * 16;\n    QGS_QR_MARK(0)\n";
    for (int s = 0; s < P; ++s) {
        o << "    f64";
        for (int i = 0; i < R; ++i) o << (i ? ", " : " ") << q(s, i);
        o << ";\n";
    }
    const std::string idx = std::string("        // column lane, member of the workgroup's 16 whose columns this lane keeps; global side: lane = (member tm of 16,\n")
                            + "        // pair group tg of 16): pass k moves the (row, column) pairs 16 k + tg of a tile\n"
                            + "        const int cc = QGS_CC(), ml = QGS_TID() >> 4, tm = cc, tg = ml;\n"
                            + "        const bool tlive = m0 + tm < n_traj;\n"
                            + "        // (the leading dimension behind an opaque copy per tile: the 84 row addresses are otherwise common to the way\n"
                            + "        // in and the way out, and the compiler keeps them in registers across the whole factorisation)\n"
                            + "        i64 ldw = ld; asm volatile(\"\" : \"+s\"(ldw));\n"
                            + "        f64* const gp = a + (i64)tg * ldw + m0 + (tlive ? tm : 0);\n";
    // way in: the global loads of tile k + 1 are issued before tile k goes through LDS (their values wait in registers that the
    // matrices do not need yet), so that the memory system always has a tile's worth of lines in flight per workgroup
    {
        o << "    {   // the matrices come in\n" << idx;
        auto issue = [&](int r0) {
            const int rows = std::min(RC, R - r0), pairs = rows * C, passes = (pairs + 15) / 16;
            for (int k = 0; k < passes; ++k) {
                const bool guard = 16 * k + 15 >= pairs;
                o << I2 << "const f64 g" << r0 << "_" << k << " = (tlive" << (guard ? " && tg < " + std::to_string(pairs - 16 * k) : "") << ") ? gp[(i64)"
                  << (r0 * C + 16 * k) << " * ldw] : 0.0;\n";
            }
        };
        issue(0);
        for (int r0 = 0; r0 < R; r0 += RC) {
            const int rows = std::min(RC, R - r0), pairs = rows * C, passes = (pairs + 15) / 16;
            if (r0 + RC < R) issue(r0 + RC);
            o << I2 << "// rows " << r0 << " .. " << r0 + rows - 1 << "\n";
            for (int k = 0; k < passes; ++k) {
                const bool guard = 16 * k + 15 >= pairs;
                o << I2 << (guard ? "if (tg < " + std::to_string(pairs - 16 * k) + ") " : "") << "tile[" << 16 * k << " + tg][tm] = g" << r0 << "_" << k << ";\n";
            }
            o << I2 << "__syncthreads();\n";
            for (int s = 0; s < P; ++s) {
                const bool guard = width[s] < 16;
                for (int i = 0; i < rows; ++i) {
                    o << I2 << q(s, r0 + i) << " = ";
                    if (guard) o << "(cc < " << width[s] << ") ? tile[" << i * C + base[s] << " + cc][ml] : 0.0;\n";
                    else o << "tile[" << i * C + base[s] << " + cc][ml];\n";
                }
            }
            o << I2 << "__syncthreads();\n";
        }
        o << "    }\n";
    }
    o << "    QGS_QR_MARK(1)\n";
    int step = 0;
    for (int j = 0; j < K; ++j) {                   // ---- dgeqr2
        const int sj = slot_of(j), ccj = lane_of(j);
        o << "    {   // column " << j << "\n";
        if (j + 1 < C) o << I2 << "QGS_QR_MARK(" << 8 + step++ << ")\n";
        // norm / beta / tau in every lane for its own column of slot sj: the values of lane cc_j are the ones that get used
        o << I2 << "f64 xn2 = 0.0;\n";
        for (int i = j + 1; i < R; ++i) o << I2 << "xn2 = __builtin_fma(" << q(sj, i) << ", " << q(sj, i) << ", xn2);\n";
        // beta = -sign(alpha) ||x||, tau = (beta - alpha) / beta = 1 + |alpha| / ||x||, scale = 1 / (alpha - beta), from ONE reciprocal
        // square root and ONE reciprocal, each a hardware estimate refined by two Newton steps (the IEEE-exact sqrt and two
        // divisions of the tile design cost 40 instructions and two dozen temporary registers per pivot: here the registers
        // are the bound)
        o << I2 << "f64 t = 0.0, scale = 0.0;\n"
          << I2 << "{\n"
          << I2 << "    const f64 alpha = " << q(sj, j) << ";\n"
          << I2 << "    f64 beta = alpha;\n"
          << I2 << "    if (xn2 != 0.0) {\n"
          << I2 << "        const f64 n2 = __builtin_fma(alpha, alpha, xn2);\n"
          << I2 << "        f64 r = __builtin_amdgcn_rsq(n2);\n"
          << I2 << "        r = __builtin_fma(0.5 * r, __builtin_fma(-n2 * r, r, 1.0), r);\n"
          << I2 << "        r = __builtin_fma(0.5 * r, __builtin_fma(-n2 * r, r, 1.0), r);\n"
          << I2 << "        f64 nrm = n2 * r;\n"
          << I2 << "        nrm = __builtin_fma(0.5 * r, __builtin_fma(-nrm, nrm, n2), nrm);\n"
          << I2 << "        beta = -__builtin_copysign(nrm, alpha);\n"
          << I2 << "        t = __builtin_fma(__builtin_fabs(alpha), r, 1.0);\n"
          << I2 << "        const f64 d = alpha - beta;\n"
          << I2 << "        f64 s = __builtin_amdgcn_rcp(d);\n"
          << I2 << "        s = __builtin_fma(s, __builtin_fma(-d, s, 1.0), s);\n"
          << I2 << "        scale = __builtin_fma(s, __builtin_fma(-d, s, 1.0), s);\n"
          << I2 << "    }\n"
          // diag(R) leaves at once (uniform row pointer + the member's byte offset); the pivot lanes' diagonal register is free from
          // here on (q_j enters this step only for columns > j, the upper triangle of R is not an output) and keeps tau_j
          << I2 << "    if (QGS_CC() == " << ccj << ") {\n"
          << I2 << "        if (m0 + (QGS_TID() >> 4) < n_traj) qgs_store_row(rdiag + (i64)" << j << " * ld + m0, (QGS_TID() >> 4) * 8u, beta);\n"
          << I2 << "        " << q(sj, j) << " = t;\n"
          << I2 << "    }\n"
          << I2 << "}\n";
        for (int s : live_slots(j, sj)) {
            const std::string ro = right_of(s, j);
            o << I2 << "{   // slot " << s << "\n";
            dot_b(s, sj, j, ccj);
            o << I3 << "f64 tm = " << q(s, j) << ", wv = 0.0, nw = 0.0;\n";
            emit_dpp_fmacs(o, I3, {{"tm", "scale", "sd"}, {"wv", "t", "tm"}}, ccj);   // tm = q_j + scale (v.q), wv = t tm
            if (!ro.empty()) o << I3 << "wv = " << ro << " ? wv : 0.0;\n";
            o << I3 << q(s, j) << " -= wv;\n";
            emit_dpp_fmacs(o, I3, {{"nw", "scale", "wv"}}, ccj);                      // w scale
            o << I3 << "nw = -nw;\n";
            update_b(s, sj, j, ccj);                                                   // q -= (w scale) v
            o << I2 << "}\n";
        }
        if (j + 1 < R) {
            o << I2 << "if (QGS_CC() == " << ccj << ") {            // the pivot lanes keep the reflector u = v * scale\n";
            for (int i = j + 1; i < R; ++i) o << I2 << "    " << q(sj, i) << " *= scale;\n";
            o << I2 << "}\n";
        }
        // Row j is final now.  In a slot whose columns are all right of j it holds entries of R's upper triangle, which nobody reads
        // again (diag(R) has left); the second phase wants zeros there (column c of Q is (0 .. 0, 1 - t, -t u) before the reflectors
        // left of c act on it, and they act on rows >= their own index).  Written as zeros HERE, unconditionally, the registers are
        // dead for the compiler from now until the second phase comes back to row j: that slack, growing by one row per step, is
        // what lets the kernel hold 3 x 36 doubles per lane within 256 registers without spilling.
        for (int s = 0; s < P; ++s)
            if (base[s] > j) o << I2 << q(s, j) << " = 0.0;\n";
        o << "    }\n";
    }
    o << "    QGS_QR_MARK(" << 8 + step << ")\n    QGS_QR_MARK(2)\n";
    for (int j = K - 1; j >= 0; --j) {              // ---- dorg2r: tau_j sits in the pivot lanes' q_j
        const int sj = slot_of(j), ccj = lane_of(j);
        o << "    {   // Q: reflector " << j << "\n";
        const std::vector<int> slots = live_slots(j, sj);
        if (!slots.empty()) o << I2 << "QGS_QR_MARK(" << 8 + ++step << ")\n";
        for (int s : slots) {
            const std::string ro = right_of(s, j);
            o << I2 << "{   // slot " << s << "\n";
            dot_b(s, sj, j, ccj);
            o << I3 << "const f64 tm = " << q(s, j) << " + sd;\n" << I3 << "f64 wv = 0.0;\n";
            emit_dpp_fmacs(o, I3, {{"wv", q(sj, j), "tm"}}, ccj);                     // t (q_j + u.q)
            if (!ro.empty()) o << I3 << "wv = " << ro << " ? wv : 0.0;\n";
            o << I3 << q(s, j) << " -= wv;\n" << I3 << "const f64 nw = -wv;\n";
            update_b(s, sj, j, ccj);
            o << I2 << "}\n";
        }
        o << I2 << "if (QGS_CC() == " << ccj << ") {            // column j of Q = H_j e_j: (0 .. 0, 1 - t, -t u)\n"
          << I2 << "    const f64 tj = " << q(sj, j) << ";\n";
        for (int i = base[sj]; i < j; ++i) o << I2 << "    " << q(sj, i) << " = 0.0;\n";      // (rows above the slot's first column: zeroed in the first phase)
        o << I2 << "    " << q(sj, j) << " = 1.0 - tj;\n";
        for (int i = j + 1; i < R; ++i) o << I2 << "    " << q(sj, i) << " *= -tj;\n";
        o << I2 << "}\n    }\n";
    }
    o << "    QGS_QR_MARK(" << 8 + step + 1 << ")\n    QGS_QR_MARK(3)\n";
    for (int r0 = 0; r0 < R; r0 += RC) {
        const int rows = std::min(RC, R - r0), pairs = rows * C, passes = (pairs + 15) / 16;
        o << "    {   // rows " << r0 << " .. " << r0 + rows - 1 << " out\n" << idx;
        for (int s = 0; s < P; ++s) {
            const bool guard = width[s] < 16;
            for (int i = 0; i < rows; ++i)
                o << I2 << (guard ? "if (cc < " + std::to_string(width[s]) + ") " : "") << "tile[" << i * C + base[s] << " + cc][ml] = " << q(s, r0 + i) << ";\n";
        }
        o << I2 << "__syncthreads();\n";
        for (int k = 0; k < passes; ++k) {
            const bool guard = 16 * k + 15 >= pairs;
            o << I2 << "if (tlive" << (guard ? " && tg < " + std::to_string(pairs - 16 * k) : "") << ") gp[(i64)" << (r0 * C + 16 * k)
              << " * ldw] = tile[" << 16 * k << " + tg][tm];\n";
        }
        o << I2 << "__syncthreads();\n    }\n";
    }
    o << "    QGS_QR_MARK(4)\n    QGS_QR_MARK(5)\n";
    o << "    QGS_CLOCK_MARK(2)\n}\n";
    GeneratedKernel g;
    g.source = o.str();
    return g;
}

GeneratedKernel generate_qr_kernel(int n_rows, int n_cols, const QrPlan &plan)
{
    if (plan.row_groups > 0) return generate_qr_grid_kernel(n_rows, n_cols, plan);
    if (plan.members == 4) return generate_qr_row_kernel(n_rows, n_cols, plan);
    const int R = n_rows, C = n_cols, K = std::min(R, C), P = plan.slots, W = plan.waves, NCH = std::max(1, plan.chains);
    const int M = plan.members, L = 64 / std::max(1, M);
    if ((M != 8 && M != 16) || P < 1 || W < 1 || W > 16 || L * P * W < C) throw std::runtime_error("batched QR: plan does not cover the columns");
    std::ostringstream o;
    const std::string I2 = "        ", I3 = "            ", I4 = "                ", I5 = "                    ";
    auto q = [](int s, int i) { return "q" + std::to_string(s) + "_" + std::to_string(i); };
    // sum_{i = lo .. R-1} a_i * b_i into `name`; one chain, or NCH partial sums added pairwise at the end
    auto dot = [&](const std::string &ind, const std::string &name, const std::function<std::string(int)> &a,
                   const std::function<std::string(int)> &b, int lo) {
        const int n = std::max(0, R - lo), nch = std::max(1, std::min(NCH, n));
        if (nch == 1) {
            o << ind << "f64 " << name << " = 0.0;\n";
            for (int i = lo; i < R; ++i) o << ind << name << " = __builtin_fma(" << a(i) << ", " << b(i) << ", " << name << ");\n";
            return;
        }
        for (int k = 0; k < nch; ++k) o << ind << "f64 " << name << "_" << k << " = 0.0;\n";
        for (int i = lo; i < R; ++i) {
            const std::string acc = name + "_" + std::to_string((i - lo) % nch);
            o << ind << acc << " = __builtin_fma(" << a(i) << ", " << b(i) << ", " << acc << ");\n";
        }
        std::vector<std::string> parts;
        for (int k = 0; k < nch; ++k) parts.push_back(name + "_" + std::to_string(k));
        while (parts.size() > 1) {
            std::vector<std::string> nx;
            for (size_t k = 0; k + 1 < parts.size(); k += 2) nx.push_back("(" + parts[k] + " + " + parts[k + 1] + ")");
            if (parts.size() & 1) nx.push_back(parts.back());
            parts.swap(nx);
        }
        o << ind << "const f64 " << name << " = " << parts[0] << ";\n";
    };
    struct Owner { int w, s, cc; };
    auto owner = [&](int j) { return Owner{(j / L) % W, (j / L) / W, j % L}; };
    // columns of slot s: L (s W + w) + cc; the slot is live at step j when some wavefront has a column > j in it
    auto slot_ever_live = [&](int s, int j) { return std::min(C - 1, L * (s * W + W - 1) + L - 1) > j; };
    // (uniform) "this wavefront has a column > j (and < C) in slot s"; empty when that holds for every wavefront
    auto slot_cond = [&](int s, int j) -> std::string {
        const bool all_gt = L * (s * W) + L - 1 > j, all_in = L * (s * W + W - 1) < C;
        if (all_gt && all_in) return "";
        std::ostringstream c;
        c << "(";
        if (!all_gt) c << L << " * (" << s * W << " + w) + " << L - 1 << " > " << j;
        if (!all_gt && !all_in) c << " && ";
        if (!all_in) c << L << " * (" << s * W << " + w) < " << C;
        c << ")";
        return c.str();
    };
    // ... in any of the slots (empty: always)
    auto any_cond = [&](const std::vector<int> &slots, int j) -> std::string {
        std::string c;
        for (int s : slots) {
            const std::string one = slot_cond(s, j);
            if (one.empty()) return "";
            c += (c.empty() ? "" : " || ") + one;
        }
        return c;
    };
    o << "#ifndef QGS_SPEC_PRELUDE\n#define QGS_SPEC_PRELUDE\n" << PRELUDE << RECORD_HELPERS << "#endif\n";
    o << "// A[row][col][member] -> Q in place, diag(R) -> rdiag[col][member]; " << R << " x " << C << ", " << M << " members per workgroup of " << W
      << " wavefronts,\n// lane = (member, column lane), " << P << " column(s) per lane (" << qr_plan_signature(plan) << ")\n";
    // (tools/ubench/qr_phases.cpp builds the kernel with -DQGS_QR_PROFILE: wave 0 notes the 100 MHz clock at the phase boundaries)
    o << "#ifdef QGS_QR_PROFILE\n#define QGS_QR_MARK(k) if (threadIdx.x == 0) prof[(i64)blockIdx.x * 160 + (k)] = (k) < 8 ? wall_clock64() : __builtin_amdgcn_s_memtime();\n"
      << "#define QGS_QR_PROF_ARG , unsigned long long* prof\n#else\n#define QGS_QR_MARK(k)\n#define QGS_QR_PROF_ARG\n#endif\n";
    o << "extern \"C\" __global__ void __launch_bounds__(" << 64 * W << ") qgs_spec_qr_" << R << "x" << C
      << "(f64* __restrict__ a, f64* __restrict__ rdiag, i64 n_traj, i64 ld QGS_QR_PROF_ARG)\n{\n";
    o << "    __shared__ f64 vb[2][" << R + 2 << "][" << M << "];      // rows j+1 .. R-1 of the reflector, [R] = tau, [R+1] = 1 / (alpha - beta)\n";
    o << "    const int lane = threadIdx.x & 63, mm = lane & " << M - 1 << ", cc = lane >> " << (M == 16 ? 4 : 3) << ";\n"
      << "    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n";
    if (M == 16)
        o << "    const i64 tile = blockIdx.x;\n";
    else   // the two half-line tiles 2 t, 2 t + 1 go to blocks b, b + 8: the same XCD (blocks go round-robin over the 8 XCDs), back to back
        o << "    const i64 tile = 2 * (8 * (i64)(blockIdx.x >> 4) + (blockIdx.x & 7)) + ((blockIdx.x >> 3) & 1);\n";
    o << "    if (tile * " << M << " >= n_traj) return;                // (the whole workgroup)\n"
      << "    const i64 m = tile * " << M << " + mm;\n"
      << "    const bool live = m < n_traj;\n"
      << "    const i64 ms = live ? m : 0;\n";
    for (int s = 0; s < P; ++s) {
        o << "    const int c" << s << " = " << L << " * (" << s * W << " + w) + cc;\n"
          << "    const bool col" << s << " = live && c" << s << " < " << C << ";\n"
          << "    f64* const ap" << s << " = a + (i64)(c" << s << " < " << C << " ? c" << s << " : 0) * ld + ms;\n";
        for (int i = 0; i < R; ++i)
            o << "    f64 " << q(s, i) << " = col" << s << " ? ap" << s << "[(i64)" << i * C << " * ld] : 0.0;\n";
        o << "    f64 tau" << s << " = 0.0;\n";
    }
    o << "    QGS_CLOCK_MARK(0)\n";
    o << "    QGS_QR_MARK(0)\n#ifdef QGS_QR_PROFILE\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n#endif\n    QGS_QR_MARK(1)\n";
    int step = 0;                                   // broadcast steps so far: step & 1 is the LDS buffer of the next one
    // pivot j: norm / beta / tau in the owner wavefront (caller has emitted `if (w == w_j)`), the pivot lanes publish the reflector
    // into buffer B when some column is left to update, and keep u = v * scale, beta, tau
    auto pivot = [&](const std::string &ind, int j, int B, bool publish) {
        const Owner ow = owner(j);
        const std::string in2 = ind + "    ";
        o << ind << "{   // pivot " << j << "\n";
        dot(in2, "xn2", [&](int i) { return q(ow.s, i); }, [&](int i) { return q(ow.s, i); }, j + 1);
        o << in2 << "const f64 alpha = " << q(ow.s, j) << ";\n"
          << in2 << "f64 pt = 0.0, beta = alpha, pscale = 0.0;\n"
          << in2 << "if (xn2 != 0.0) {\n"
          << in2 << "    beta = -__builtin_copysign(__builtin_sqrt(__builtin_fma(alpha, alpha, xn2)), alpha);\n"
          << in2 << "    pt = (beta - alpha) / beta;\n"
          << in2 << "    pscale = 1.0 / (alpha - beta);\n"
          << in2 << "}\n";
        o << in2 << "if (cc == " << ow.cc << ") {            // the pivot lanes\n";
        if (publish) {
            for (int i = j + 1; i < R; ++i) o << in2 << "    vb[" << B << "][" << i << "][mm] = " << q(ow.s, i) << ";\n";
            o << in2 << "    vb[" << B << "][" << R << "][mm] = pt;\n" << in2 << "    vb[" << B << "][" << R + 1 << "][mm] = pscale;\n";
        }
        o << in2 << "    tau" << ow.s << " = pt;\n"
          << in2 << "    if (live) rdiag[(i64)" << j << " * ld + m] = beta;\n"
          << in2 << "    " << q(ow.s, j) << " = beta;\n";
        for (int i = j + 1; i < R; ++i) o << in2 << "    " << q(ow.s, i) << " *= pscale;\n";
        o << in2 << "}\n" << ind << "}\n";
    };
    // second phase, reflector j: the pivot lanes publish u and tau into buffer B (when a column is left to update) and turn their
    // column into column j of Q = H_j e_j: (0 .. 0, 1 - t, -t u)
    auto publish_q = [&](const std::string &ind, int j, int B, bool publish) {
        const Owner ow = owner(j);
        o << ind << "if (w == " << ow.w << " && cc == " << ow.cc << ") {   // reflector " << j << "\n";
        if (publish) {
            for (int i = j + 1; i < R; ++i) o << ind << "    vb[" << B << "][" << i << "][mm] = " << q(ow.s, i) << ";\n";
            o << ind << "    vb[" << B << "][" << R << "][mm] = tau" << ow.s << ";\n";
        }
        for (int i = 0; i < j; ++i) o << ind << "    " << q(ow.s, i) << " = 0.0;\n";
        o << ind << "    " << q(ow.s, j) << " = 1.0 - tau" << ow.s << ";\n";
        for (int i = j + 1; i < R; ++i) o << ind << "    " << q(ow.s, i) << " *= -tau" << ow.s << ";\n";
        o << ind << "}\n";
    };
    // slot s updated by the reflector of step j (t, scale, v<i> in scope, or re-read from buffer B in reload mode)
    auto update = [&](const std::string &ind, int s, int j, int B, bool qr_phase) {
        const std::string S = std::to_string(s), in2 = ind + "    ";
        auto load_v = [&](const std::string &name) {
            for (int i = j + 1; i < R; ++i) o << in2 << "const f64 " << name << i << " = vb[" << B << "][" << i << "][mm];\n";
        };
        o << ind << "{   // slot " << s << "\n";
        std::string vn = "v";
        if (plan.reload) { vn = "va"; load_v(vn); }
        dot(in2, "sd", [&](int i) { return vn + std::to_string(i); }, [&](int i) { return q(s, i); }, j + 1);
        if (qr_phase)
            o << in2 << "const f64 wv = (c" << S << " > " << j << ") ? t * __builtin_fma(scale, sd, " << q(s, j) << ") : 0.0;\n"
              << in2 << q(s, j) << " -= wv;\n" << in2 << "const f64 wsc = -(wv * scale);\n";
        else
            o << in2 << "const f64 wv = (c" << S << " > " << j << ") ? t * (" << q(s, j) << " + sd) : 0.0;\n"
              << in2 << q(s, j) << " -= wv;\n" << in2 << "const f64 wsc = -wv;\n";
        if (plan.reload) {
            o << in2 << "asm volatile(\"\" ::: \"memory\");      // second pass over the reflector: read again, do not keep\n";
            vn = "vc"; load_v(vn);
        }
        for (int i = j + 1; i < R; ++i)
            o << in2 << q(s, i) << " = __builtin_fma(wsc, " << vn << i << ", " << q(s, i) << ");\n";
        o << ind << "}\n";
    };
    // one broadcast step: barrier, read the reflector, update; `ahead` emits the owner's look-ahead work after its first slot
    auto broadcast_step = [&](int j, bool qr_phase, int ahead_w, int ahead_s, const std::function<void(const std::string &)> &ahead) {
        const int B = step & 1;
        o << I2 << "QGS_QR_MARK(" << 8 + step << ")\n";
        o << I2 << "__syncthreads();\n";
        std::vector<int> slots;
        for (int s = 0; s < P; ++s) if (slot_ever_live(s, j)) slots.push_back(s);
        // the wavefront takes part when one of its slots still has a column > j (the owner of pivot j + 1 always has)
        {
            const std::string part = any_cond(slots, j);
            o << I2 << "if (" << (part.empty() ? std::string("true") : part) << ") {\n";
        }
        o << I3 << "const f64 t = vb[" << B << "][" << R << "][mm];\n";
        if (qr_phase) o << I3 << "const f64 scale = vb[" << B << "][" << R + 1 << "][mm];\n";
        if (!plan.reload)
            for (int i = j + 1; i < R; ++i) o << I3 << "const f64 v" << i << " = vb[" << B << "][" << i << "][mm];\n";
        auto others = [&](const std::string &ind, int skip) {
            for (int s : slots) {
                if (s == skip) continue;
                const std::string c = slot_cond(s, j);
                if (!c.empty()) o << ind << "if (" << c << ")\n";
                update(ind, s, j, B, qr_phase);
            }
        };
        others(I3, -1);
        o << I2 << "}\n";
        // (Forming pivot j + 1 BETWEEN the owner's slot updates, in one basic block with them, was measured: the interleaved form wants
        // more than 256 registers and spills, 0.46 instead of 0.22 ms at 36 x 36 -- profiles/r05_qr.md section 4.)
        if (ahead_w >= 0) {
            o << I2 << "if (w == " << ahead_w << ")\n";
            ahead(I2);
        }
        ++step;
    };
    // ---- dgeqr2: columns > j exist for j < C - 1; pivot j + 1 is formed during step j
    o << "    if (w == " << owner(0).w << ")\n";
    pivot("    ", 0, 0, C > 1);
    for (int j = 0; j + 1 < C; ++j) {
        o << "    {   // column " << j << "\n";
        const bool more = j + 1 < K;
        const Owner nx = owner(j + 1);
        const int Bn = (step + 1) & 1;
        broadcast_step(j, true, more ? nx.w : -1, more ? nx.s : -1, [&](const std::string &ind) { pivot(ind, j + 1, Bn, j + 2 < C); });
        o << "    }\n";
    }
    o << "    QGS_QR_MARK(2)\n";
    // ---- dorg2r: reflector j acts on the columns > j; reflector j - 1 is published during step j
    if (K - 1 >= 0 && !(K - 2 >= 0 && C > 1)) publish_q("    ", K - 1, 0, false);
    if (C > 1 && K >= 2) {
        // column K - 1 = C - 1 has nothing to its right: it only becomes a column of Q; reflector K - 2 is the first to be applied
        publish_q("    ", K - 1, 0, false);
        publish_q("    ", K - 2, step & 1, true);
        for (int j = K - 2; j >= 0; --j) {
            o << "    {   // Q: reflector " << j << "\n";
            const int Bn = (step + 1) & 1;
            o << I2 << "QGS_QR_MARK(" << 8 + step << ")\n";
            o << I2 << "__syncthreads();\n";
            // (no data dependence on this step's update: published before the update so that it is never waited for)
            if (j > 0) publish_q(I2, j - 1, Bn, true);
            // body without its own barrier
            {
                const int B = step & 1;
                std::vector<int> slots;
                for (int s = 0; s < P; ++s) if (slot_ever_live(s, j)) slots.push_back(s);
                const std::string part = any_cond(slots, j);
                o << I2 << "if (" << (part.empty() ? std::string("true") : part) << ") {\n";
                o << I3 << "const f64 t = vb[" << B << "][" << R << "][mm];\n";
                if (!plan.reload)
                    for (int i = j + 1; i < R; ++i) o << I3 << "const f64 v" << i << " = vb[" << B << "][" << i << "][mm];\n";
                for (int s : slots) {
                    const std::string c = slot_cond(s, j);
                    if (!c.empty()) o << I3 << "if (" << c << ")\n";
                    update(I3, s, j, B, false);
                }
                o << I2 << "}\n";
                ++step;
            }
            o << "    }\n";
        }
    }
    o << "    QGS_QR_MARK(" << 8 + step << ")\n    QGS_QR_MARK(3)\n";
    for (int s = 0; s < P; ++s) {
        o << "    if (col" << s << ") {\n";
        for (int i = 0; i < R; ++i) o << I2 << "ap" << s << "[(i64)" << i * C << " * ld] = " << q(s, i) << ";\n";
        o << "    }\n";
    }
    o << "    QGS_QR_MARK(4)\n#ifdef QGS_QR_PROFILE\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n#endif\n    QGS_QR_MARK(5)\n";
    o << "    QGS_CLOCK_MARK(2)\n}\n";
    GeneratedKernel g;
    g.source = o.str();
    return g;
}

bool tableau_is_subdiagonal(int s, const double *a)
{
    for (int i = 0; i < s; ++i)
        for (int j = 0; j < s; ++j)
            if (a[i * s + j] != 0.0 && j != i - 1) return false;
    return true;
}

bool kernel_uses_jacobian(Kernel k)
{
    switch (k) {
    case Kernel::Jac: case Kernel::Tgl: case Kernel::TglPair: case Kernel::TglX: case Kernel::TglDense: case Kernel::TglLds:
    case Kernel::AdjLds: return true;
    default: return false;
    }
}

std::string kernel_name(Kernel k, int S, const CodegenOptions &opt)
{
    switch (k) {
    case Kernel::Tend: return "qgs_spec_tend";
    case Kernel::Jac: return "qgs_spec_jac";
    case Kernel::Rk: return "qgs_spec_rk_s" + std::to_string(S);
    case Kernel::RkSplit: return "qgs_spec_rksplit" + std::to_string(opt.row_split) + "_s" + std::to_string(S);
    case Kernel::RkStages: return "qgs_spec_rkstages_s" + std::to_string(S);
    case Kernel::RkStagesPair: return "qgs_spec_rkstagesp_s" + std::to_string(S);
    case Kernel::TglPair: return "qgs_spec_tglp_s" + std::to_string(S);
    case Kernel::Tgl: return "qgs_spec_tgl_s" + std::to_string(S);
    case Kernel::RkLds: return "qgs_spec_rklds" + std::to_string(opt.lds_waves);
    case Kernel::TglLds: return "qgs_spec_tgllds" + std::to_string(opt.lds_waves) + (opt.lds_tgl_members == 8 ? "m8" : "");
    case Kernel::AdjLds: return "qgs_spec_adjlds" + std::to_string(opt.lds_waves) + (opt.lds_tgl_members == 8 ? "m8" : "");
    case Kernel::TglX: return "qgs_spec_tglx" + std::to_string(opt.tgl_share_x) + "_s" + std::to_string(S);
    case Kernel::RkRec: return "qgs_spec_rkr_s" + std::to_string(S);
    case Kernel::TendLds: return "qgs_spec_tendlds" + std::to_string(opt.lds_waves);
    case Kernel::RkDense: return "qgs_spec_rkd_s" + std::to_string(S);
    case Kernel::TglDense: return "qgs_spec_tgld_s" + std::to_string(S);
    case Kernel::RkLdsDense: return "qgs_spec_rkldsd" + std::to_string(opt.lds_waves);
    }
    return "";
}

// One kernel per translation unit: kernels compiled together share the register allocator's context and
// perturb each other (the plain stepper went from 276 to 324 VGPRs and 4.6 -> 4.7 ms when a 4-way split
// sibling was added to its module), so every kernel is generated, compiled and cached on its own.
// CodeGenPrepare: 3.3 of the 3.5 min the general-tableau LDS-resident stepper takes to compile at ndim 228, half of the 40 s of
// the LDS-resident tangent kernels, and the kernels come out the same without it (same registers, same scratch; 24.4 vs
// 24.5 ms for 16 384 members x 8 columns x 10 steps).
std::vector<std::string> kernel_compile_flags(Kernel k)
{
    if (k == Kernel::RkLdsDense || k == Kernel::TglLds || k == Kernel::AdjLds) return {"-mllvm", "-disable-cgp"};
    return {};
}

std::string options_signature(const CodegenOptions &o)
{
    std::ostringstream s;
    s << "w" << o.min_waves_per_simd << ",il" << o.interleave << ",til" << o.tgl_interleave << ",pv" << o.tgl_park_v << ",tp" << o.tgl_pair
      << ",td" << o.tgl_coeff_dedupe << ",sx" << o.tgl_share_x << ",sr" << o.rk_spread_rec << ",rs" << o.row_split << ",lw" << o.lds_waves
      << ",lm" << o.lds_tgl_members << ",lc" << o.lds_cap << ",lg" << o.lds_group << ",ld" << o.lds_coeff_dedupe << ",ly" << o.lds_yload_ahead
      << ",lo" << o.lds_order;
    return s.str();
}

GeneratedKernel generate_kernel(int ndim, const std::vector<Term> &tensor, const std::vector<Term> &jac_tensor, Kernel k, int S,
                                const CodegenOptions &opt, const Derived &der)
{
    GeneratedKernel gen;
    struct TablesGuard {
        std::vector<CoefTable> *old;
        explicit TablesGuard(std::vector<CoefTable> *t) : old(g_tables) { g_tables = t; }
        ~TablesGuard() { g_tables = old; g_ktab = nullptr; }
    } tables_guard(&gen.tables);
    std::ostringstream o;
    o << "#ifndef QGS_SPEC_PRELUDE\n#define QGS_SPEC_PRELUDE\n" << PRELUDE << RECORD_HELPERS << "#endif\n";
    // (only what this kernel is generated from: a kernel of the tendencies tensor is the same text whatever the Jacobian tensor is)
    if (kernel_uses_jacobian(k)) {
        o << "// ndim = " << ndim << ", Jacobian tensor: " << jac_tensor.size() << " entries\n";
        if (!der.j.empty()) o << "// derived monomials: " << der.j.size() << "\n";
    } else {
        o << "// ndim = " << ndim << ", tendencies tensor: " << tensor.size() << " entries\n";
        if (!der.t.empty()) o << "// derived monomials: " << der.t.size() << "\n";
    }
    const std::vector<Row> rows = build_rows(ndim, tensor);
    struct BaseGuard { int old; BaseGuard(int b) : old(g_ext_base) { g_ext_base = b; } ~BaseGuard() { g_ext_base = old; } } guard(ndim);
    switch (k) {
    case Kernel::Tend: emit_tend_kernel(o, ndim, rows, opt, der.t); break;
    case Kernel::Jac: emit_jac_kernel(o, ndim, jac_tensor, der.j); break;
    case Kernel::Rk: emit_rk_kernel(o, ndim, rows, S, false, opt, der.t); break;
    case Kernel::RkSplit: emit_rk_split_kernel(o, ndim, rows, S, opt.row_split, opt, der.t); break;
    case Kernel::RkStages: emit_rk_kernel(o, ndim, rows, S, true, opt, der.t); break;
    case Kernel::RkStagesPair: emit_rk_kernel(o, ndim, rows, S, true, opt, der.t, false, true); break;
    case Kernel::TglPair:
        emit_tgl_kernel(o, ndim, build_wx_rows(ndim, jac_tensor, false), build_wx_rows(ndim, jac_tensor, true), S, opt, der.j, 1, false, true);
        break;
    case Kernel::RkRec: emit_rk_kernel(o, ndim, rows, S, false, opt, der.t, true); break;
    case Kernel::RkDense: emit_rk_dense_kernel(o, ndim, rows, S, opt, der.t); break;
    case Kernel::TglDense:
        emit_tgl_kernel(o, ndim, build_wx_rows(ndim, jac_tensor, false), build_wx_rows(ndim, jac_tensor, true), S, opt, der.j, 1, true);
        break;
    case Kernel::Tgl:
        emit_tgl_kernel(o, ndim, build_wx_rows(ndim, jac_tensor, false), build_wx_rows(ndim, jac_tensor, true), S, opt, der.j);
        break;
    case Kernel::TglX:
        emit_tgl_kernel(o, ndim, build_wx_rows(ndim, jac_tensor, false), build_wx_rows(ndim, jac_tensor, true), S, opt, der.j,
                        opt.tgl_share_x);
        break;
    case Kernel::RkLds: emit_rk_lds_kernel(o, ndim, rows, opt, der.t); break;
    case Kernel::TendLds: emit_rk_lds_kernel(o, ndim, rows, opt, der.t, true); break;
    case Kernel::RkLdsDense: emit_rk_lds_kernel(o, ndim, rows, opt, der.t, false, true); break;
    case Kernel::TglLds: emit_tgl_lds_kernel(o, ndim, build_wx_rows(ndim, jac_tensor, false), false, opt, der.j); break;
    case Kernel::AdjLds: emit_tgl_lds_kernel(o, ndim, build_wx_rows(ndim, jac_tensor, true), true, opt, der.j); break;
    }
    gen.source = o.str();
    return gen;
}

std::vector<std::pair<Kernel, int>> kernel_list(int ndim, bool have_jac, const std::vector<int> &stages, const CodegenOptions &opt_in)
{
    CodegenOptions opt = opt_in;
    if (ndim < 2 * opt.row_split) opt.row_split = 1;      // too few rows to split
    std::vector<std::pair<Kernel, int>> l = {{Kernel::Tend, 0}};
    if (have_jac) l.push_back({Kernel::Jac, 0});
    for (int S : stages) {
        l.push_back({Kernel::Rk, S});
        if (opt.rk_spread_rec) l.push_back({Kernel::RkRec, S});
        if (opt.row_split > 1) l.push_back({Kernel::RkSplit, S});
        if (have_jac) {
            l.push_back({Kernel::RkStages, S});
            l.push_back({Kernel::Tgl, S});
            if (opt.tgl_pair) { l.push_back({Kernel::RkStagesPair, S}); l.push_back({Kernel::TglPair, S}); }
            if (opt.tgl_share_x > 1) l.push_back({Kernel::TglX, S});
        }
    }
    return l;
}

std::string generate_source(int ndim, const std::vector<Term> &tensor, const std::vector<Term> &jac_tensor,
                            const std::vector<int> &stages, const CodegenOptions &opt, const Derived &der)
{
    std::string all;
    for (auto &ks : kernel_list(ndim, !jac_tensor.empty(), stages, opt))
        all += generate_kernel(ndim, tensor, jac_tensor, ks.first, ks.second, opt, der).source + "\n";
    return all;
}

// ---- canonical form (codegen.h) ---------------------------------------------------------------------------------------------
namespace {

struct CoordKey {
    int i, j, k;
    bool operator==(const CoordKey &o) const { return i == o.i && j == o.j && k == o.k; }
};
struct CoordHash {
    size_t operator()(const CoordKey &c) const
    {
        uint64_t h = (uint64_t)(uint32_t)c.i * 0x9e3779b97f4a7c15ull;
        h = (h ^ (uint64_t)(uint32_t)c.j) * 0xc2b2ae3d27d4eb4full;
        h = (h ^ (uint64_t)(uint32_t)c.k) * 0x165667b19e3779f9ull;
        return (size_t)(h ^ (h >> 29));
    }
};

uint64_t magnitude_bits(double v)
{
    const double a = std::fabs(v);
    uint64_t u;
    std::memcpy(&u, &a, sizeof u);
    return u;                                   // the bit pattern, so that NaN payloads and 0.0 are classes like any other
}

std::vector<Term> merge_duplicates(const std::vector<Term> &in)
{
    std::vector<Term> out;
    std::unordered_map<CoordKey, size_t, CoordHash> where;
    where.reserve(in.size() * 2);
    for (const Term &t : in) {
        auto it = where.find(CoordKey{t.i, t.j, t.k});
        if (it == where.end()) { where.emplace(CoordKey{t.i, t.j, t.k}, out.size()); out.push_back(t); }
        else out[it->second].v += t.v;
    }
    return out;
}

}  // namespace

// Magnitudes that differ by at most `ulp` units in the last place are ONE magnitude (the one that appears first): see codegen.h
// Canonical.  Only normal numbers are ever merged (a relative bound: among subnormals a unit in the last place is not small
// against the value); zero, subnormals, infinities and NaNs are classes by their bit pattern.
bool magnitudes_close(double a, double b, int ulp)
{
    if (a == b) return true;
    if (ulp <= 0) return false;
    if (!std::isnormal(a) || !std::isnormal(b) || !(a > 0.0) || !(b > 0.0)) return false;
    const int64_t d = (int64_t)magnitude_bits(a) - (int64_t)magnitude_bits(b);
    return d >= -(int64_t)ulp && d <= (int64_t)ulp;
}

void canonicalize(const std::vector<Term> &terms, Canonical &c, int ulp)
{
    ulp = std::max(0, std::min(ulp, 64));
    c.terms = merge_duplicates(terms);
    c.magnitude.assign(1, 0.0);
    std::unordered_map<uint64_t, int> id;                  // exact magnitude (bit pattern) -> class
    std::map<double, int> reps;                            // class representatives (normal numbers), for the neighbourhood search
    id.emplace(magnitude_bits(0.0), 0);
    for (Term &t : c.terms) {
        const double a = std::fabs(t.v);
        auto it = id.find(magnitude_bits(t.v));
        int cls = -1;
        if (it != id.end()) cls = it->second;
        else if (ulp > 0 && std::isnormal(a)) {
            // the closest representative within the tolerance (the lower class id on a tie)
            auto hi = reps.lower_bound(a);
            int64_t best = (int64_t)ulp + 1;
            for (int side = 0; side < 2; ++side) {
                auto q = hi;
                if (side == 0) { if (q == reps.begin()) continue; --q; }
                else if (q == reps.end()) continue;
                if (!magnitudes_close(a, q->first, ulp)) continue;
                const int64_t d = std::llabs((int64_t)magnitude_bits(a) - (int64_t)magnitude_bits(q->first));
                if (d < best || (d == best && q->second < cls)) { best = d; cls = q->second; }
            }
        }
        if (cls < 0) {
            cls = (int)c.magnitude.size();
            c.magnitude.push_back(a);
            if (std::isnormal(a)) reps.emplace(a, cls);
        }
        id.emplace(magnitude_bits(t.v), cls);              // (the same bits again: straight to this class)
        t.v = std::copysign((double)cls, t.v);
    }
}

void Canonical::decode(const std::vector<double> &table, std::vector<double> &out) const
{
    out.resize(table.size());
    for (size_t n = 0; n < table.size(); ++n) {
        const double a = std::fabs(table[n]);
        const size_t id = (size_t)a;
        if (!(a == (double)id) || id >= magnitude.size()) throw std::logic_error("codegen: table entry is not a magnitude-class id");
        out[n] = std::copysign(magnitude[id], table[n]);
    }
}

// Greedy common-subexpression reduction of a set of monomials: while some monomial is longer than `target`, the pair
// of variables that occurs in most of them becomes a new variable (index ndim + 1 + n) and replaces one occurrence
// of the pair in each.  [a,a,a,m] for many m -> p = a*a, q = p*a, [q,m]: two products shared by all terms.
static void reduce_monomials(int ndim, std::vector<std::vector<int>> &mono, size_t target, std::vector<std::pair<int, int>> &derived)
{
    while (true) {
        std::map<std::pair<int, int>, int> count;
        for (const auto &f : mono) {
            if (f.size() <= target) continue;
            std::vector<std::pair<int, int>> seen;
            for (size_t a = 0; a < f.size(); ++a)
                for (size_t b = a + 1; b < f.size(); ++b) {
                    const std::pair<int, int> pr(f[a], f[b]);                 // f is sorted: f[a] <= f[b]
                    if (std::find(seen.begin(), seen.end(), pr) == seen.end()) { seen.push_back(pr); ++count[pr]; }
                }
        }
        if (count.empty()) break;
        std::pair<int, int> best = count.begin()->first;
        int best_n = 0;
        for (const auto &kv : count) if (kv.second > best_n) { best_n = kv.second; best = kv.first; }
        const int id = ndim + 1 + (int)derived.size();
        derived.push_back(best);
        for (auto &f : mono) {
            while (f.size() > target) {
                auto ia = std::find(f.begin(), f.end(), best.first);
                if (ia == f.end()) break;
                auto ib = std::find(best.first == best.second ? ia + 1 : f.begin(), f.end(), best.second);
                if (ib == f.end()) break;
                if (ib < ia) std::swap(ia, ib);
                f.erase(ib);
                f.erase(ia);
                f.insert(std::upper_bound(f.begin(), f.end(), id), id);
            }
        }
    }
}

void reduce_polynomial(int ndim, int rank, int64_t nnz, const int32_t *coo, const double *val, bool jacobian,
                       std::vector<Term> &out, std::vector<std::pair<int, int>> &derived)
{
    out.clear();
    derived.clear();
    const int first = jacobian ? 2 : 1;              // Jacobian entries: (i, j | k, l, m): the monomial starts at column 2
    if (rank == 3) {                                 // nothing to reduce: the entries are the terms, in the caller's order
        for (int64_t e = 0; e < nnz; ++e) out.push_back({coo[3 * e], coo[3 * e + 1], coo[3 * e + 2], val[e]});
        return;
    }
    // merge entries with equal (row, column, monomial): the Jacobian tensor holds every permutation separately
    std::map<std::vector<int>, size_t> where;
    std::vector<std::vector<int>> mono;
    std::vector<std::pair<int, int>> head;           // (i, j) -- j only for the Jacobian
    std::vector<double> v;
    for (int64_t e = 0; e < nnz; ++e) {
        const int32_t *c = coo + (int64_t)rank * e;
        if (c[0] < 1 || (jacobian && c[1] < 1)) continue;   // row 0 is the constant slot; Df drops row and column 0
        std::vector<int> f;
        for (int q = first; q < rank; ++q) if (c[q] != 0) f.push_back(c[q]);
        std::sort(f.begin(), f.end());
        std::vector<int> key = {c[0], jacobian ? c[1] : 0};
        key.insert(key.end(), f.begin(), f.end());
        auto it = where.find(key);
        if (it != where.end()) { v[it->second] += val[e]; continue; }
        where[key] = mono.size();
        mono.push_back(f);
        head.push_back({c[0], jacobian ? c[1] : 0});
        v.push_back(val[e]);
    }
    reduce_monomials(ndim, mono, jacobian ? 1 : 2, derived);
    for (size_t n = 0; n < mono.size(); ++n) {
        const auto &f = mono[n];
        if (jacobian) out.push_back({head[n].first, head[n].second, f.empty() ? 0 : f[0], v[n]});
        else out.push_back({head[n].first, f.size() == 2 ? f[0] : 0, f.empty() ? 0 : f.back(), v[n]});
    }
}

int64_t count_tendency_flops_instr(int ndim, const std::vector<Term> &tensor, const CodegenOptions &opt)
{
    const std::vector<Row> rows = build_rows(ndim, tensor);
    int64_t n = 0;
    for (int i = 1; i <= ndim; ++i) {
        const Row &r = rows[i];
        n += (int64_t)r.lin.size();
        auto groups = group_by_abs(r.bil);
        for (auto &g : groups) n += (int64_t)g.size() + 1;
    }
    return n;
}

}  // namespace qgs
