// codegen.cpp -- see codegen.h.  Emits HIP C++ source; everything is straight-line fp64 code built from the
// tensor, so the compiled kernels contain no tensor index loads at all: the coefficients arrive through the
// scalar unit (per-kernel __constant__ tables walked with s_load_dwordx16) and every VALU slot is a
// v_mul_f64 / v_fma_f64.  The tables are declared without initialisers: their contents are returned next to the source
// (GeneratedKernel::tables) and stored into the loaded module, so the source -- and the code object -- depend on the
// structure of the tensor only, not on its values.
#include "codegen_internal.h"

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

namespace detail {

std::string hexlit(double v)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%a", v);   // exact hex-float literal (C++17)
    return std::string(buf);
}

// Coefficient table mode: instead of a literal (two s_mov_b32 per use) a coefficient can be the next entry of
// a __constant__ table that the kernel walks sequentially (fetched eight at a time by s_load_dwordx16).
// Every stage emits the same rows in the same order, so one table per (kernel, wave partition) serves all stages.
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
std::string resolve_ktab(const std::string &text, KTable &t, bool dedupe)
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



// Emit the products of a group of (sign, left, right) factors into a temp `g`:
//   g = l0*r0; g = fma(+-l1, r1, g); ...

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

}  // namespace detail
using namespace detail;

bool tableau_is_subdiagonal(int s, const double *a)
{
    for (int i = 0; i < s; ++i)
        for (int j = 0; j < s; ++j)
            if (a[i * s + j] != 0.0 && j != i - 1) return false;
    return true;
}

bool tgl_asm_supported(int ndim, bool rank3, const CodegenOptions &opt)
{
    return rank3 && detail::tgl_asm_applies(ndim, 4, opt);
}

// (8 wavefronts x 256 registers: up to 32 rows per wavefront leave a factor cache of 45 values)
bool lds_tgl_asm_supported(int ndim, bool rank3, const CodegenOptions &opt)
{
    return rank3 && opt.lds_tgl_members == 16 && ndim <= 256 && (int64_t)opt.lds_asm_waves * 32 >= ndim;
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
    case Kernel::TglPair: return (opt.tgl_asm && S >= 2 && S <= 4 ? "qgs_spec_tglpa_s" : "qgs_spec_tglp_s") + std::to_string(S);
    case Kernel::Tgl: return "qgs_spec_tgl_s" + std::to_string(S);
    case Kernel::RkLds: return opt.lds_asm ? "qgs_spec_rkldsa" + std::to_string(opt.lds_asm_waves) : "qgs_spec_rklds" + std::to_string(opt.lds_waves);
    case Kernel::TglLds:
        if (opt.lds_tgl_asm) return "qgs_spec_tglldsa" + std::to_string(opt.lds_asm_waves);
        return "qgs_spec_tgllds" + std::to_string(opt.lds_waves) + (opt.lds_tgl_members == 8 ? "m8" : "");
    case Kernel::AdjLds:
        if (opt.lds_tgl_asm) return "qgs_spec_adjldsa" + std::to_string(opt.lds_asm_waves);
        return "qgs_spec_adjlds" + std::to_string(opt.lds_waves) + (opt.lds_tgl_members == 8 ? "m8" : "");
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
      << ",lo" << o.lds_order << ",la" << o.lds_asm << ":" << o.lds_asm_waves << ":" << o.lds_asm_cap << ":" << o.lds_asm_pingpong << ":"
      << o.lds_asm_lanes << ":" << o.lds_asm_chunk << ":" << o.lds_asm_vfree << ":" << o.lds_asm_sfree << ":" << o.lds_asm_mincap << ":" << o.lds_asm_coef
      << ":" << o.lds_asm_ring << ":" << o.lds_asm_progressive << ":" << o.lds_asm_merge
      << ":" << o.lds_asm_keep << ":" << o.lds_asm_fmac << ":" << o.lds_asm_skip << ",lt" << o.lds_tgl_asm << ",ta" << o.tgl_asm << ":" << o.tgl_asm_ring << ",ds" << o.asm_dpp_spacing;
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
        if (opt.tgl_asm && S >= 2 && S <= 4) {
            if (!der.j.empty() || !tgl_asm_applies(ndim, S, opt)) throw std::logic_error("codegen: the hand-scheduled tangent kernel does not exist for this model");
            emit_tgl_asm_kernel(o, ndim, build_wx_rows(ndim, jac_tensor, false), build_wx_rows(ndim, jac_tensor, true), S, opt);
            break;
        }
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
    case Kernel::RkLds:
        if (opt.lds_asm && !der.t.empty()) throw std::logic_error("codegen: the hand-scheduled LDS stepper takes rank-3 tensors only");
        if (opt.lds_asm) emit_rk_lds_asm_kernel(o, ndim, rows, opt);
        else emit_rk_lds_kernel(o, ndim, rows, opt, der.t);
        break;
    case Kernel::TendLds: emit_rk_lds_kernel(o, ndim, rows, opt, der.t, true); break;
    case Kernel::RkLdsDense: emit_rk_lds_kernel(o, ndim, rows, opt, der.t, false, true); break;
    case Kernel::TglLds: case Kernel::AdjLds: {
        const bool adjoint = (k == Kernel::AdjLds);
        if (opt.lds_tgl_asm && !lds_tgl_asm_supported(ndim, der.j.empty(), opt))
            throw std::logic_error("codegen: the hand-scheduled LDS tangent kernels take rank-3 tensors and tiles of 16 members only");
        if (opt.lds_tgl_asm) emit_tgl_lds_asm_kernel(o, ndim, build_wx_rows(ndim, jac_tensor, adjoint), adjoint, opt);
        else emit_tgl_lds_kernel(o, ndim, build_wx_rows(ndim, jac_tensor, adjoint), adjoint, opt, der.j);
        break;
    }
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
