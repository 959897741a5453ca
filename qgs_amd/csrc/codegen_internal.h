// codegen_internal.h -- what the translation units of the kernel generator share (namespace qgs::detail).  Not part of the
// library's interface: codegen.h is.
//
//   codegen.cpp           coefficient tables, rows, shared statement emitters, the prelude of every generated source, kernel
//                         names / options / dispatch (generate_kernel), canonical form of a tensor, rank-5 reduction
//   codegen_steppers.cpp  register-resident kernels: f, Df, fused RK steppers (sub-diagonal, general tableau, row-split)
//   codegen_tangent.cpp   register-resident tangent / adjoint kernels
//   codegen_tangent_asm.cpp  the pair kernel of the tangent model with a hand-scheduled body (inline assembly, fixed registers)
//   codegen_lds.cpp       LDS-resident kernels of large systems: phases, row partition, stepper, f, tangent / adjoint
//   codegen_lds_asm.cpp   the LDS-resident stepper with a hand-scheduled stage body (inline assembly, fixed registers)
//   codegen_qr.cpp        shape-specialised batched Householder QR
#pragma once
#include "codegen.h"

#include <cmath>
#include <functional>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace qgs {
namespace detail {

// ---- coefficient tables (codegen.cpp) ------------------------------------------------------------------------------------------
// Coefficient table mode: instead of a literal (two s_mov_b32 per use) a coefficient can be the next entry of
// a __constant__ table that the kernel walks sequentially (fetched eight at a time by s_load_dwordx16).
// Every stage emits the same rows in the same order, so one table per (kernel, wave partition) serves all stages.
struct KTable {
    std::vector<double> vals;
    size_t cursor = 0;
    size_t pad_to = 0;      // group-64 mode: the run-ahead loads may touch this many entries
};
extern thread_local KTable *g_ktab;
extern thread_local std::vector<CoefTable> *g_tables;      // tables of the kernel being generated (generate_kernel)
extern thread_local int g_ext_base;                        // first index of the derived monomials in the names of a kernel

std::string hexlit(double v);
std::string lit(double v);
std::string coef_fma(const std::string &acc, double c, const std::string &factor);
std::string resolve_ktab(const std::string &text, KTable &t, bool dedupe = false);
std::vector<std::string> split_lines(const std::string &text);
std::string interleave(const std::vector<std::vector<std::string>> &lists);
void emit_ktable(std::ostringstream &o, const std::string &name, const KTable &t);

// ---- rows of a tensor and the statements built from them (codegen.cpp) --------------------------------------------------------------
struct Bil { int j, k; double c; };
struct Lin { int k; double c; };

struct Row {
    double c0 = 0.0;
    bool has_c0 = false;
    std::vector<Lin> lin;
    std::vector<Bil> bil;
};
std::vector<Row> build_rows(int ndim, const std::vector<Term> &tensor);

using NameFn = std::function<std::string(int)>;
NameFn names(const std::string &prefix);

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
void emit_group(std::ostringstream &o, const char *indent, const std::string &g, const std::vector<Prod> &ps);

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

void emit_tend_row(std::ostringstream &o, const char *indent, const Row &row, const std::string &res, const NameFn &X,
                   const CodegenOptions &opt, int uid);

struct WX { int w, x; double c; };
std::vector<std::vector<WX>> build_wx_rows(int ndim, const std::vector<Term> &jac, bool adjoint);
void emit_wx_row(std::ostringstream &o, const char *indent, const std::vector<WX> &items, const std::string &res, const NameFn &X,
                 const NameFn &W, const CodegenOptions &opt, int uid);

void emit_derived(std::ostringstream &o, const char *indent, int ndim, const std::vector<std::pair<int, int>> &der, const NameFn &X);
std::string decl_list(const std::string &prefix, int ndim);
void emit_settle_loads(std::ostringstream &o, const char *indent, const std::string &prefix, const std::vector<int> &idx);
std::vector<int> all_rows(int ndim);

extern const char *PRELUDE;             // head of every generated source
extern const char *RECORD_HELPERS;

// ---- register-resident kernels (codegen_steppers.cpp, codegen_tangent.cpp) ----------------------------------------------------------
void emit_tend_kernel(std::ostringstream &out, int ndim, const std::vector<Row> &rows, const CodegenOptions &opt,
                      const std::vector<std::pair<int, int>> &der);
void emit_jac_kernel(std::ostringstream &out, int ndim, const std::vector<Term> &jac, const std::vector<std::pair<int, int>> &der);
void emit_rk_kernel(std::ostringstream &out, int ndim, const std::vector<Row> &rows, int S, bool store_stages,
                    const CodegenOptions &opt, const std::vector<std::pair<int, int>> &der, bool spread_rec = false,
                    bool pair_stages = false);
void emit_rk_dense_kernel(std::ostringstream &out, int ndim, const std::vector<Row> &rows, int S, const CodegenOptions &opt,
                          const std::vector<std::pair<int, int>> &der);
std::vector<int> partition_rows(int ndim, const std::vector<Row> &rows, int R, const CodegenOptions &opt);
void emit_rk_split_kernel(std::ostringstream &out, int ndim, const std::vector<Row> &rows, int S, int R,
                          const CodegenOptions &opt, const std::vector<std::pair<int, int>> &der);
void emit_tgl_kernel(std::ostringstream &out, int ndim, const std::vector<std::vector<WX>> &tgl,
                     const std::vector<std::vector<WX>> &adj, int S, const CodegenOptions &opt,
                     const std::vector<std::pair<int, int>> &der, int share_x = 1, bool dense = false, bool pair_x = false);

void emit_tgl_asm_kernel(std::ostringstream &out, int ndim, const std::vector<std::vector<WX>> &tgl,
                         const std::vector<std::vector<WX>> &adj, int S, const CodegenOptions &opt);
bool tgl_asm_applies(int ndim, int S, const CodegenOptions &opt);      // the hand-scheduled tangent kernel exists for this shape

// ---- LDS-resident kernels (codegen_lds.cpp, codegen_lds_asm.cpp) ---------------------------------------------------------------------
// Terms live in "node space": a node is one LDS-resident value (stepper: node m = mode m; tangent model: node j = w_j,
// node ndim + k = x_k); a term is c * node_j * node_k accumulated into row `row`, j == 0 meaning a single factor.
struct PTerm { int row, j, k; double c; };      // j <= k; j == 0: linear term c*x_k

struct Phase {
    std::vector<int> modes;                     // loaded at the head of the phase (ascending)
    std::vector<PTerm> terms;                   // sorted by (j, k, row)
};
typedef std::vector<std::vector<PTerm>> RowTerms;           // [row] -> its terms
struct LdsStats { int64_t loads = 0, instr = 0, phases = 0, coef = 0; };
struct LdsNode { int64_t offset; int lane_kind; };          // byte offset in LDS without the lane part; which lane-offset variable
using NodeFn = std::function<LdsNode(int)>;

std::vector<Phase> build_phases(int ndim, const std::vector<PTerm> &terms, int cap);
int64_t lds_wave_instr(int n_nodes, const RowTerms &rt, const std::vector<int> &own, int cap, bool group);
std::vector<std::vector<int>> lds_partition(int n_rows, int n_nodes, const RowTerms &rt, int W, int cap, bool group);
void emit_rk_lds_kernel(std::ostringstream &out, int ndim, const std::vector<Row> &rows, const CodegenOptions &opt,
                        const std::vector<std::pair<int, int>> &der, bool tend_kernel = false, bool dense = false);
void emit_rk_lds_asm_kernel(std::ostringstream &out, int ndim, const std::vector<Row> &rows, const CodegenOptions &opt);
void emit_tgl_lds_asm_kernel(std::ostringstream &out, int ndim, const std::vector<std::vector<WX>> &wx, bool adjoint, const CodegenOptions &opt);
void emit_tgl_lds_kernel(std::ostringstream &out, int ndim, const std::vector<std::vector<WX>> &wx, bool adjoint,
                         const CodegenOptions &opt, const std::vector<std::pair<int, int>> &der);

}  // namespace detail
}  // namespace qgs
