// codegen_qr.cpp -- generator of the shape-specialised batched Householder QR kernels.  See codegen.h.
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
using namespace detail;

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
    if (p.stagger > 0) s << "st" << p.stagger << "b" << p.stagger_bit;
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
    if (plan.stagger > 0)
        o << "    if ((blockIdx.x >> " << plan.stagger_bit << ") & 1u) {          // de-synchronise the workgroups that share a CU\n"
          << "        for (int k = 0; k < " << plan.stagger << "; ++k) __builtin_amdgcn_s_sleep(127);\n    }\n";
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

}  // namespace qgs
