// codegen.h -- generator of tensor-specialised HIP kernel source (host side, plain C++).
//
// The qgs tendencies are dx_i = sum_{j,k} T_ijk x_j x_k with x_0 = 1
// (reference: qgs/functions/sparse_mul.py:48-81, qgs/functions/tendencies.py:111-115).
// On a CDNA4 wavefront with one ensemble member per lane the state must live in VGPRs, which
// cannot be indexed by the run-time (j,k) of a COO entry.  So for register-resident sizes the
// library turns the COO tensor into straight-line fp64 FMA code once per model (the same idea as
// the reference's own symbolic code export, but targeting gfx950 ISA through hiprtc) and
// compiles it when the model is created.  See DESIGN.md section 3.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace qgs {

struct Term {          // one COO entry (i, j, k, value); index 0 is the constant slot
    int i, j, k;
    double v;
};

// Always on (the alternatives were measured slower and removed in round 3): equal-|coefficient| bilinear terms of a row are
// factored, c * (m1 +- m2 ...); coefficients come from a __constant__ table walked by s_load_dwordx16 in a software pipeline
// of 16-coefficient groups (literal s_mov pairs: 7.4 instead of 4.6 ms for the stepper; compiler-placed loads: 5.75 ms).
//
// Coefficient VALUES are never part of the generated source (round 4): every coefficient is an entry of a `__constant__`
// table that the source declares without an initialiser; the generator returns the table contents next to the source and
// the library stores them into the loaded module (hipModuleGetGlobal).  What the source does depend on is the STRUCTURE of
// the tensor: the sparsity pattern, which coefficients of a row share a magnitude (they are factored), and their signs.
// The library therefore generates from the *canonical form* of a tensor (canonicalize() below), in which every coefficient
// is replaced by +-(id of its magnitude class): two parameter sets of one model -- the reference's run-time operands
// `coo` / `val` of sparse_mul3, qgs/functions/sparse_mul.py:48-81 -- give the same source, hence the same code object.
struct CodegenOptions {
    int min_waves_per_simd = 1;
    int interleave = 2;        // rows whose statements are interleaved in the row-split stepper (ILP)
    int tgl_interleave = 2;    // tangent kernel: rows whose statements are emitted round-robin (config 4, 100 calls: 0.949 -> 0.933 ms with park_v)
    bool tgl_park_v = true;    // tangent kernel: park the step-start vector in LDS after stage 0 (four register vectors instead of five)
    bool tgl_pair = true;      // stage record in mode pairs between qgs_spec_rkstagesp_s<S> and qgs_spec_tglp_s<S> (128-bit accesses)
    bool tgl_asm = false;      // the pair kernel of the tangent model (qgs_spec_tglp_s<S>) with a hand-scheduled body: qgs_spec_tglpa_s<S>
                               // (codegen_tangent_asm.cpp; rank-3 tensors, 2 - 4 stages, ndim <= 37).  Bitwise the same results; measured
                               // 0.977 against 0.935 ms per call at config 4 (profiles/r06_tgls.md): off, selectable with QGS_HIP_TGL_ASM=1
    int tgl_asm_ring = 6;      // ... coefficient chunks of 16 held in registers
    bool tgl_coeff_dedupe = true;  // tangent kernel: same de-duplication of coefficient fetches as lds_coeff_dedupe (config 4: 1.24 -> 1.20 ms)
    int tgl_share_x = 4;       // tangent kernel: columns (wavefronts) per workgroup that share the stage states of 64 members
                               // through LDS, next stage prefetched during the current one (1 = every wavefront loads its own)
    bool rk_spread_rec = true; // also emit qgs_spec_rkr_s<S> for write_steps == 1 (every step is a record): the 36 row stores of a step are
                               // spread over its stages, unconditional, scalar row pointer + lane offset (codegen.cpp emit_rk_kernel)
    int row_split = 4;         // also emit the row-split stepper with this many wavefronts per 64 members
    int lds_waves = 16;        // LDS-resident stepper (large ndim): wavefronts per 64 members
    int lds_tgl_members = 16;  // LDS-resident tangent kernels: members per workgroup tile (16 x 4 columns, or 8 x 8 columns)
    int lds_cap = 20;          // ... and modes cached in registers per phase (24 spills at 128 VGPRs: 63.6 ms vs 55.6 ms)
    bool lds_group = true;     // ... sum equal-|coefficient| terms of a row inside a phase first: 11 % fewer instructions and
                               //     25 % fewer coefficient fetches (needs the smaller factor cache above to stay spill-free)
    bool lds_coeff_dedupe = true; // ... a coefficient already present in the group of 16 being consumed is not fetched again
    int lds_yload_ahead = 2;   // ... and phases before the end of a stage at which the step-start state is re-read
    bool lds_tgl_asm = true;   // the LDS-resident tangent / adjoint kernels (rank 3, tiles of 16 members) with the same hand-scheduled stage body:
                               // qgs_spec_tglldsa<W> / qgs_spec_adjldsa<W>
    bool lds_asm = true;       // the LDS-resident stepper (rank 3) with a hand-scheduled stage body: qgs_spec_rkldsa<W> (codegen_lds_asm.cpp);
                               //     its own workgroup shape and phase size:
    int lds_asm_waves = 8;    //     wavefronts per 64 members (16: 128 registers per lane; 8: 256)
    int lds_asm_mincap = 10;   //     a wavefront takes at most as many rows as leave it a cache of this many modes
    int lds_asm_cap = 64;      //     modes cached per phase (one half of the factor cache when lds_asm_pingpong)
    bool lds_asm_pingpong = false;  // the LDS reads of phase p + 1 land in the idle half of the cache while phase p computes
    int lds_asm_lanes = 3;     //     statements whose instructions are emitted round-robin (independent dependency chains; three keep a sum two
                               //     instructions away from the DPP instruction that reads it)
    int lds_asm_coef = 1;      //     coefficients: 0 = scalar loads into two SGPR buffers; 1 = vector loads into a ring of registers + DPP broadcast
    bool lds_asm_progressive = true;  // (1, one cache set) statements ordered by the last factor they need, each instruction waits only
                               //     for the LDS reads it needs (in-order returns, nothing else on the counter)
    bool asm_dpp_spacing = true;   // both hand-scheduled kernels: two wait states between a VALU write of ANY register and a DPP instruction that
                               //     reads it (the compiler's rule); false: only for the DPP-shuffled operand, which these kernels never write by VALU
    int lds_asm_skip = 0;      //     developer timing experiments (wrong results): 1 = no barriers, 2 = no LDS waits, 4 = no vector-memory waits, 8 = no loads, 16 = no spacing s_nop, 32 = plain FMA for the DPP form
    bool lds_asm_fmac = true;  //     t = fma(a, b, t) as the two-address v_fmac_f64 (4 bytes instead of 8)
    bool lds_asm_merge = true; //     consecutive phases whose modes fit the cache together are one phase (the greedy cover's tail of 2 - 4-mode phases)
    bool lds_asm_keep = true;  //     (one cache set) a mode the previous phase left in a slot stays there and is not read again
    int lds_asm_ring = 3;      //     (1) chunks of 16 coefficients held in registers
    int lds_asm_chunk = 16;    //     coefficients per scalar-load chunk (two SGPR buffers of this size)
    int lds_asm_vfree = 24;    //     low VGPRs left to the compiler's frame code
    int lds_asm_sfree = 24;    //     low SGPRs left to the compiler's frame code
    int lds_order = 1;         // ... order of the grouped statements inside a phase of the stepper: 0 by (row, |c|), 1 by (|c|, row): partner
                               //     rows' equal coefficients become neighbours for the de-duplication (52.7 -> 51.8 ms, emit_lds_phases)
};

// Rank-5 tensors (QgsTensorDynamicT / QgsTensorT4, qgs/tensors/qgtensor.py:843-1363; contracted by sparse_mul5 /
// sparse_mul4, qgs/functions/sparse_mul.py:84-158): dx_i = sum T_ijklm x_j x_k x_l x_m.  The generated code keeps
// the bilinear form of every emitter: products of two variables that many monomials share become *derived
// variables* (index ndim+1+n = product of two earlier indices, evaluated once per tendency evaluation), until every
// tendency monomial has at most two factors and every Jacobian monomial at most one.
struct Derived {
    std::vector<std::pair<int, int>> t, j;     // for the tendencies tensor / for the Jacobian tensor
    bool empty() const { return t.empty() && j.empty(); }
};
// coo is (nnz, rank) row-major, rank 3 or 5.  `out` receives (i, j, k, v) terms over the extended index space
// (Jacobian: i, j = column, k = the single remaining x index), `derived` the products to evaluate first.
void reduce_polynomial(int ndim, int rank, int64_t nnz, const int32_t *coo, const double *val, bool jacobian,
                       std::vector<Term> &out, std::vector<std::pair<int, int>> &derived);

// Classification of a Butcher tableau (reference: integrate.py:214-219 uses the full matrix `a`).
// The register-resident kernels need a[i][j] != 0 only for j == i-1 (RK4, Heun, midpoint, Euler ...).
bool tableau_is_subdiagonal(int s, const double *a);

// Source of all specialised kernels of one model:
//   qgs_spec_tend            f(x) for an ensemble                       (tendencies.py:111-115)
//   qgs_spec_jac             Df(x) for an ensemble                      (tendencies.py:117-121)
//   qgs_spec_rk_s<S>         fused S-stage RK trajectory stepper        (integrate.py:182-223)
//   qgs_spec_rkr_s<S>        the same stepper for write_steps == 1 (every step a record): record stores spread over the step
//   qgs_spec_rkd_s<S>        S-stage RK with a general lower-triangular tableau (partial stage sums in LDS; optional stage store)
//   qgs_spec_tgld_s<S>       tangent / adjoint propagation for such a tableau
//   qgs_spec_rkstages_s<S>   same, also storing every stage state       (feeds the tangent kernel)
//   qgs_spec_rkstagesp_s<S> / qgs_spec_tglp_s<S>   the pair that exchanges the stage record in mode pairs (128-bit accesses)
//   qgs_spec_rklds<W>        large systems: stage state in LDS, W wavefronts per 64 members, factors cached in
//                            registers phase by phase; run-time stage count, optional stage store
//   qgs_spec_rkldsd<W>       the LDS-resident stepper for a general lower-triangular tableau
//   qgs_spec_tendlds<W>      f(x) with the same machinery (one evaluation)
//   qgs_spec_tgllds<W> / qgs_spec_adjlds<W>   tangent / adjoint model of large systems, 16 members x 4 columns per
//                            workgroup, stage state and tangent vector in LDS
//   qgs_spec_tgl_s<S>        tangent / adjoint propagation, one lane per (member, column)
//                                                                       (integrate.py:226-231, 555-614)
//   qgs_spec_tglx<C>_s<S>    same, C columns of 64 members per workgroup sharing the (double-buffered, prefetched) stage
//                            states through LDS
// `stages` lists the stage counts S to instantiate (sub-diagonal tableaus only).
// generate_source concatenates every kernel (inspection / offline builds); the library compiles one kernel per
// translation unit (generate_kernel), see codegen.cpp.
// One `__constant__` coefficient table of a generated kernel: `symbol[values.size()]`, declared uninitialised in the source.
struct CoefTable {
    std::string symbol;
    std::vector<double> values;          // whole 8-double blocks (zero padded)
};
struct GeneratedKernel {
    std::string source;
    std::vector<CoefTable> tables;       // in the values the tensors were given in (canonical tensors: magnitude-class ids)
};

// Canonical form of a tensor for the generator.  Entries with equal coordinates are merged (summed in their incoming order);
// then every value v becomes copysign(id(|v|), v) where id numbers the distinct magnitudes 1, 2, ... in order of first
// appearance (0.0 keeps id 0).  `magnitude[id]` restores the value: decode() maps a table of a generated kernel back to the
// coefficients of THIS tensor.  The tendencies tensor and the Jacobian tensor are canonicalised separately: a kernel reads
// one of them (kernel_uses_jacobian), and its cache entry depends on that one only.
//
// Magnitudes within `ulp` units in the last place of each other (default 2; normal numbers only) are ONE magnitude, the first
// to appear: the inner products of a spectral model reach analytically equal coefficients along different floating-point routes,
// and whether two such results agree in the last bit changes from one parameter value to the next (MAOOAM-36, kd = 0.0290 ...
// 0.0300: three different patterns of such splits in eleven values, a fourth at 0.031 -- taken literally, as many structures and
// sets of code objects where the model has one).  The specialised kernels therefore compute with the class representative:
// a perturbation of <= ulp * 2.2e-16 relative on a coefficient, far below the fp64 tolerances stated for the path.  The generic
// kernels, the contraction kernel and the reference's loops use the values as given (`val` is a run-time operand,
// qgs/functions/sparse_mul.py:76-81).  QGS_HIP_MAGNITUDE_ULP=0 makes every distinct value its own class (exact coefficients
// in the specialised kernels too, more structures in a parameter sweep); the value is part of the kernel-cache key.
struct Canonical {
    std::vector<Term> terms;
    std::vector<double> magnitude;       // magnitude[0] = 0.0
    void decode(const std::vector<double> &table, std::vector<double> &out) const;
};
constexpr int DEFAULT_MAGNITUDE_ULP = 2;
void canonicalize(const std::vector<Term> &terms, Canonical &out, int ulp = DEFAULT_MAGNITUDE_ULP);

enum class Kernel { Tend, Jac, Rk, RkSplit, RkStages, Tgl, RkLds, TglLds, AdjLds, TglX, RkRec, TendLds, RkDense, TglDense, RkLdsDense,
                    RkStagesPair, TglPair };
std::string kernel_name(Kernel k, int S, const CodegenOptions &opt);
// the hand-scheduled pair kernel of the tangent model exists for a model of this size (callers clear CodegenOptions::tgl_asm otherwise)
bool tgl_asm_supported(int ndim, bool rank3, const CodegenOptions &opt);
bool lds_tgl_asm_supported(int ndim, bool rank3, const CodegenOptions &opt);
bool kernel_uses_jacobian(Kernel k);     // its coefficients come from the Jacobian tensor (else: from the tendencies tensor)
GeneratedKernel generate_kernel(int ndim, const std::vector<Term> &tensor, const std::vector<Term> &jac_tensor, Kernel k, int S,
                                const CodegenOptions &opt, const Derived &der = Derived());
// extra compiler flags a kernel kind is built with (part of the cache key; e.g. "-mllvm -disable-cgp" for the LDS-resident
// tangent kernels, whose compilation is otherwise dominated by a pass that changes nothing in the result)
std::vector<std::string> kernel_compile_flags(Kernel k);
// every generator option as text (part of the cache key)
std::string options_signature(const CodegenOptions &opt);
std::vector<std::pair<Kernel, int>> kernel_list(int ndim, bool have_jac, const std::vector<int> &stages, const CodegenOptions &opt);
std::string generate_source(int ndim, const std::vector<Term> &tensor, const std::vector<Term> &jac_tensor,
                            const std::vector<int> &stages, const CodegenOptions &opt, const Derived &der = Derived());

// Batched Householder QR (dgeqr2 + dorg2r) fully unrolled for one matrix shape: kernel `qgs_spec_qr_<rows>x<cols>` (a, rdiag, n_traj,
// ld), n_cols <= n_rows <= 300, n_cols <= 64.  Replaces np.linalg.qr in the Benettin loops (qgs/toolbox/lyapunov.py:540-547, 599-628).
// Three layouts, chosen per shape by qr_plan (codegen.cpp has the reasons and profiles/r05_qr.md the measurements):
//   row design  (plan.members == 4): one wavefront = 4 members x 16 column lanes, all columns of a member in its lanes' registers,
//                the pivot column through DPP row broadcasts; no LDS or barrier in the factorisation.  Where 2 * rows * ceil(cols / 16)
//                registers fit one lane and >= 70 % of the column lanes are used (36 x 36, 38 x 38, 32 x 32 ...).
//                grid = ceil(n_traj / 16) workgroups of 256 threads.
//   tile design (plan.members == 16 | 8): 16 (8) members per workgroup, lane = (member, column lane), plan.slots columns per lane,
//                the reflector broadcast through LDS, one barrier per step.  Small and thin shapes, and 38 < rows <= 64.
//                grid = ceil(n_traj / 16) workgroups (members == 8: ceil(n_traj / 8) rounded up to a multiple of 16) of 64 * plan.waves threads.
//   grid design (plan.row_groups > 0): rows > 64: a member over plan.waves wavefronts = 4 * waves row groups (rows dealt cyclically),
//                DPP broadcast inside a group, one number per column and step across groups (shuffles + LDS), plan.members members
//                per workgroup.  grid = ceil(n_traj / plan.members) workgroups of 64 * plan.waves * plan.members threads.
struct QrPlan {
    int members = 16;     // members per workgroup (16: whole 128-byte lines; 8: half lines, two workgroups per line on one XCD)
    int slots = 1;        // columns per lane
    int waves = 1;        // wavefronts per workgroup ((64 / members) * slots * waves >= n_cols)
    int chains = 1;       // partial sums per dot product (1 = one left-to-right chain)
    bool reload = false;  // read the reflector from LDS once for the dot products and again for the update (large n_rows)
    int stagger = 0;      // row design: workgroups whose index has bit `stagger_bit` set start `stagger` x 8 128 cycles late (s_sleep), so that the
    int stagger_bit = 8;  //     two workgroups of a CU are not in their load / store phases at the same time (0: off)
    bool one_wave_per_simd = false;   // row design beyond 256 registers per lane: one wavefront per SIMD, part of the matrices in AGPRs
    int row_groups = 0;   // > 0: grid design for tall matrices (rows > 64): a member over `waves` wavefronts = 4 * waves row groups,
                          //      `members` members per workgroup
};
QrPlan qr_plan(int n_rows, int n_cols, int members = 0, int slots = 0);      // 0: the default choice
std::string qr_plan_signature(const QrPlan &plan);     // (part of the cache key)
GeneratedKernel generate_qr_kernel(int n_rows, int n_cols, const QrPlan &plan);

// Rough count of fp64 VALU instructions of one tendency evaluation in the generated code
// (used for the roofline note in the bench output and DESIGN.md).
int64_t count_tendency_flops_instr(int ndim, const std::vector<Term> &tensor, const CodegenOptions &opt);

}  // namespace qgs
