// codegen_lds_asm.cpp -- the LDS-resident kernels of large rank-3 systems with a hand-scheduled stage body: the stepper
// (qgs_spec_rkldsa<W>) and the tangent / adjoint model (qgs_spec_tglldsa<W> / qgs_spec_adjldsa<W>).  See codegen.h /
// codegen_internal.h; the frames (workgroup shape, stage state in LDS, rows split over wavefronts, phases that cache a set of LDS
// values in registers) are those of emit_rk_lds_kernel / emit_tgl_lds_kernel in codegen_lds.cpp.
//
// Why.  Left to the compiler, the straight-line stage body of those kernels spills at every workgroup shape (420 B per lane at 16
// wavefronts, 700 - 1 100 B at 12 and 8: profiles/r06_lds228_variants.txt) -- the machine scheduler hoists LDS reads and
// coefficient fetches across the phases the generator laid out, and every `s_waitcnt lgkmcnt(0)` (LDS reads and scalar loads share
// one counter, scalar loads return out of order) waits for something that has only just been issued: 57 - 59 % of the
// wavefront-cycles of qgs_spec_rklds16 / qgs_spec_tgllds16 are waits.  The generator knows every live range, so here it allocates the
// registers itself and emits the stage body as ONE inline-assembly statement per wavefront (8 wavefronts x 256 registers):
//   * registers: the running sums `acc` (tied to the C++ variables by physical-register constraints), the stage sums `k`, a few
//     temporaries, the coefficient ring, and the factor cache -- one set: a value a phase shares with its predecessor stays in its
//     slot, the statements of a phase are ordered by the last LDS read they need and wait for exactly that read (developer
//     variant: two halves, phase p + 1 prefetched while phase p computes);
//   * coefficients: the wavefront's table in chunks of 16 doubles through vector memory, lane l of every row of 16 lanes loading
//     entry l into a ring of register pairs two chunks ahead, picked by `v_fmac_f64_dpp ... row_newbcast` (vector loads return in
//     order: the wait is precise; nothing but LDS reads is left on the LDS / scalar counter).  Developer variant: two SGPR buffers;
//   * the step-start state y is read back from the private global buffer into cache slots the last phase does not use, requested
//     when that phase starts, and not at all in the last stage;
//   * rows per wavefront: contiguous blocks found by dynamic programming over the instructions a block really needs;
//   * nothing is spilled, no lane moves, no scratch in the stage body.
// Everything outside the stage body (prologue, record stores, stage-state store / load for the tangent model, final stores) stays C++.
// profiles/r06_lds228.md and profiles/r06_tgllds.md have the measurements.
#include "codegen_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <sstream>
#include <stdexcept>

namespace qgs {
namespace detail {

namespace {

std::string vreg(int r) { return "v[" + std::to_string(r) + ":" + std::to_string(r + 1) + "]"; }
std::string sreg(int r) { return "s[" + std::to_string(r) + ":" + std::to_string(r + 1) + "]"; }

// One instruction of the stage body before the coefficients are placed.
struct AIns {
    enum Kind { MulT, FmaT, AccK, MovK, Raw } kind;
    int t = 0;              // temporary (VGPR number) written by MulT / FmaT, or read by AccK when src_temp
    int a = 0, b = 0;       // factor registers (MulT / FmaT)
    bool neg = false;       // MulT / FmaT: the product enters negated
    int row = 0;            // AccK / MovK: own-row index (0 .. R-1)
    double coef = 0.0;      // AccK / MovK: signed coefficient (a table entry)
    int src = 0;            // AccK: register multiplied by the coefficient
    std::string text;       // Raw
};

struct Statement { std::vector<AIns> ins; };

// Row blocks: contiguous in a row sequence (neighbouring rows share their factors: every departure from contiguity costs
// instructions -- two runs per wavefront + 4 %, four runs + 10 % on MAOOAM 6x6), balanced by the fp64 instructions a block really
// needs with the factor cache ITS row count leaves (cap_for), at most `max_rows` rows per block (the registers).  The sequence is
// the one of lds_partition: rows in order, the cheap ones (MAOOAM: the ocean rows) spread evenly through it.  Optimal contiguous
// split for the current cost estimate by dynamic programming, the estimate refined against the real counts a few times.
std::vector<std::vector<int>> asm_partition(int n_rows, int n_nodes, const RowTerms &rt, int W, const std::function<int(int)> &cap_for, int max_rows)
{
    std::vector<double> c(n_rows + 1, 0.0);
    double total = 0.0;
    for (int i = 1; i <= n_rows; ++i) {
        c[i] = 1;
        for (const PTerm &t : rt[i]) c[i] += (t.j == 0) ? 1 : 2;
        total += c[i];
    }
    std::vector<int> seq;
    {
        std::vector<int> heavy, light;
        for (int i = 1; i <= n_rows; ++i) ((c[i] * 2 * n_rows < total) ? light : heavy).push_back(i);
        double heavy_total = 0, run = 0;
        for (int i : heavy) heavy_total += c[i];
        size_t nl = 0;
        for (int i : heavy) {
            seq.push_back(i);
            run += c[i];
            while (nl < light.size() && run * (double)light.size() >= heavy_total * (double)(nl + 1)) seq.push_back(light[nl++]);
        }
        while (nl < light.size()) seq.push_back(light[nl++]);
    }
    const int n = (int)seq.size();
    if ((int64_t)W * max_rows < n) throw std::logic_error("codegen: the hand-scheduled LDS stepper has too few registers for this many rows per wavefront");
    std::vector<std::vector<int>> best;
    double best_max = 0.0;
    for (int iter = 0; iter < 8; ++iter) {
        std::vector<double> pre(n + 1, 0.0);
        for (int q = 0; q < n; ++q) pre[q + 1] = pre[q] + c[seq[q]];
        const double INF = 1e300;
        std::vector<std::vector<double>> dp(W + 1, std::vector<double>(n + 1, INF));
        std::vector<std::vector<int>> from(W + 1, std::vector<int>(n + 1, -1));
        dp[0][0] = 0.0;
        for (int w = 1; w <= W; ++w)
            for (int q = 1; q <= n; ++q)
                for (int q0 = std::max(0, q - max_rows); q0 < q; ++q0) {
                    if (dp[w - 1][q0] >= INF) continue;
                    const double v = std::max(dp[w - 1][q0], pre[q] - pre[q0]);
                    if (v < dp[w][q]) { dp[w][q] = v; from[w][q] = q0; }
                }
        if (from[W][n] < 0) throw std::logic_error("codegen: no row partition within the register plan");
        std::vector<std::vector<int>> cand(W);
        for (int w = W, q = n; w >= 1; --w) {
            const int q0 = from[w][q];
            cand[w - 1].assign(seq.begin() + q0, seq.begin() + q);
            q = q0;
        }
        double worst = 0.0;
        std::vector<double> actual(W, 0.0), est(W, 0.0);
        for (int v = 0; v < W; ++v) {
            std::sort(cand[v].begin(), cand[v].end());
            actual[v] = (double)lds_wave_instr(n_nodes, rt, cand[v], cap_for((int)cand[v].size()), true);
            for (int i : cand[v]) est[v] += c[i];
            worst = std::max(worst, actual[v]);
        }
        if (best.empty() || worst < best_max) { best = cand; best_max = worst; }
        for (int v = 0; v < W; ++v)
            if (est[v] > 0.0) for (int i : cand[v]) c[i] *= actual[v] / est[v];
    }
    return best;
}

// The two kernels that share the stage body below.  Stepper: nodes 1 .. ndim are the modes of the stage state, xs[mode][64].
// Tangent / adjoint model (frame of emit_tgl_lds_kernel: 16 members x 4 columns per workgroup): nodes 1 .. ndim are the components
// of the tangent stage vector, ws[mode][64] (what the wavefronts write at the end of a stage), nodes ndim + 1 .. 2 ndim the stage
// state of the 16 members, xs[mode][16], which the whole workgroup loads between two stage bodies.
struct AsmFrame {
    bool tangent = false, adjoint = false;
    int MT = 16;                                             // tangent: members per workgroup
};

void emit_lds_asm(std::ostringstream &out, int ndim, const RowTerms &rt, const std::vector<double> &c0, const AsmFrame &fr, const CodegenOptions &opt)
{
    const int W = opt.lds_asm_waves, cap = std::max(2, opt.lds_asm_cap), NL = std::max(1, opt.lds_asm_lanes);
    const bool pp = opt.lds_asm_pingpong && !fr.tangent;
    const bool dpp = opt.lds_asm_coef == 1 || fr.tangent;    // coefficients through vector memory + DPP broadcast (else: SGPR chunks)
    const int n_nodes = fr.tangent ? 2 * ndim : ndim;
    const int MT = fr.MT, NC = 64 / MT;
    const int CE = dpp ? 16 : ((opt.lds_asm_chunk == 8 || opt.lds_asm_chunk == 12) ? opt.lds_asm_chunk : 16);     // coefficients per chunk
    const int NR = dpp ? std::max(2, opt.lds_asm_ring) : 0;  // coefficient chunks in registers (dpp): chunk c in ring slot c % NR
    const int VT = std::min(256, (512 / ((W + 3) / 4)) / 8 * 8);                                     // registers a lane may have
    const std::string kname = (fr.tangent ? (fr.adjoint ? "qgs_spec_adjldsa" : "qgs_spec_tglldsa") : "qgs_spec_rkldsa") + std::to_string(W);
    if (ndim > 256) throw std::logic_error("codegen: the hand-scheduled LDS kernels address at most 256 rows of 64 lanes");
    auto is_x = [&](int n) { return fr.tangent && n > ndim; };                  // a stage-state node of the tangent frame
    // register plan of a wavefront with R rows: [0, VF) the compiler's, then acc, k, temporaries, address registers, (dpp: the
    // coefficient ring,) cache.  The cache is what the rows leave: wavefronts with few, long rows get the large cache their rows
    // profit from.
    const int VF = std::max(16, opt.lds_asm_vfree) / 2 * 2;
    struct Plan { int ACC0, K0, T0, LB1, L15, RING, C0, NS, half, cap; };
    auto plan_for = [&](int R) {
        Plan pl;
        pl.ACC0 = VF; pl.K0 = pl.ACC0 + 2 * R; pl.T0 = pl.K0 + 2 * R; pl.LB1 = pl.T0 + 2 * NL; pl.L15 = pl.LB1 + 1;
        pl.RING = (pl.L15 + 2) / 2 * 2;
        pl.C0 = pl.RING + 2 * NR;
        pl.NS = (VT - pl.C0) / 2;
        pl.half = pp ? pl.NS / 2 : pl.NS;
        pl.cap = std::min(cap, pl.half);
        return pl;
    };
    int max_rows = 0;                                        // the most rows that still leave a phase of `lds_asm_mincap` modes
    while (plan_for(max_rows + 1).cap >= std::max(4, opt.lds_asm_mincap)) ++max_rows;
    const std::vector<std::vector<int>> owns = asm_partition(ndim, n_nodes, rt, W, [&](int R) { return plan_for(R).cap; }, max_rows);
    std::vector<int> slot(ndim + 1, 0), slot0(W, 0);
    {
        int q = 0;
        for (int w = 0; w < W; ++w) { slot0[w] = q; for (int d : owns[w]) slot[d] = q++; }
    }
    // scalar plan: [0, SF) the compiler's, (SGPR mode: two coefficient buffers,) address pairs.  Every uniform input arrives in a
    // VGPR and is moved to an SGPR of this plan by v_readfirstlane_b32: under the SGPR pressure this statement creates, the compiler
    // handed VGPRs to "s"-constrained operands in some of the wavefront branches.
    const int SF = std::max(16, opt.lds_asm_sfree) / 4 * 4;
    const int BUF[2] = {SF, SF + 2 * CE}, YB = dpp ? SF : SF + 4 * CE, KT = YB + 2, YW = KT + 2, KB = dpp ? YW + 2 : YW, LAST = KB + 2;
    if (LAST + 1 > 96) throw std::logic_error("codegen: the hand-scheduled LDS stepper does not fit its scalar plan");

    std::ostringstream o;
    std::vector<KTable> tables(W);
    LdsStats stats;
    int64_t n_chunks = 0, n_extra_waits = 0, n_hazard_nops = 0;
    std::vector<int64_t> wave_instr(W, 0);
    if (!fr.tangent) {
        o << "\n// run-time stage count RK stepper, stage state in LDS, rows split over " << W << " wavefronts per 64 members, stage body\n"
          << "// hand-scheduled: " << VT << " registers per lane = " << VF << " for the frame + 2 x 2 per own row (acc, k) + " << 2 * NL
          << " temporaries + the factor cache in " << (pp ? "two halves" : "one set") << " (phases of <= " << cap << " modes); coefficients in chunks of "
          << CE << (dpp ? " through vector memory, " + std::to_string(NR) + " chunks in registers, broadcast by DPP" : " in two SGPR buffers") << "\n";
        o << "extern \"C\" __global__ void __launch_bounds__(" << 64 * W << ") " << kname << "(\n"
          << "    const f64* __restrict__ y_in,\n"
          << "    f64* __restrict__ y_out,        // final state, X[mode][member] (may be null)\n"
          << "    f64* __restrict__ ywork,        // private [workgroup][mode][64]: state at the start of the current step\n"
          << "    f64* __restrict__ rec, f64* __restrict__ stages,\n"
          << "    const f64* __restrict__ dtime, const f64* __restrict__ tab,\n"
          << "    i64 n_traj, i64 ld, i64 step_begin, i64 step_end, i64 write_steps, i64 n_records, int backward, int write_final, int S)\n{\n";
        o << "    __shared__ f64 xs[" << ndim << "][QGS_WAVE];\n";
        o << "    const int lane = threadIdx.x & 63;\n"
          << "    const unsigned lane8 = (unsigned)lane * 8u;\n"
          << "    const unsigned ldsaddr = (unsigned)(unsigned long long)(&xs[0][0]) + lane8;   // LDS byte address of xs[0][lane]\n"
          << "    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n"
          << "    const i64 m0 = (i64)blockIdx.x * QGS_WAVE + lane;\n"
          << "    const bool live = m0 < n_traj;\n"
          << "    const i64 m = live ? m0 : (n_traj - 1);\n"
          << "    f64* const ywg = ywork + (i64)blockIdx.x * " << ndim * 64 << ";              // this workgroup's block (uniform)\n"
          << "    QGS_CLOCK_MARK(0)\n";
    } else {
        const std::string sMT = std::to_string(MT), sMSK = std::to_string(MT - 1), sMSH = std::to_string(MT == 8 ? 3 : 4);
        o << "\n// " << (fr.adjoint ? "adjoint" : "tangent") << " model, run-time stage count, " << MT << " members x " << NC << " columns per workgroup of " << W
          << " wavefronts, stage state\n// and tangent stage vector in LDS, stage body hand-scheduled: " << VT << " registers per lane = " << VF
          << " for the frame + 2 x 2 per own row (acc, k) + " << 2 * NL << " temporaries\n// + the factor cache (phases of <= " << cap
          << " values); coefficients in chunks of 16 through vector memory, " << NR << " chunks in registers, broadcast by DPP\n";
        o << "extern \"C\" __global__ void __launch_bounds__(" << 64 * W << ") " << kname << "(\n"
          << "    const f64* __restrict__ w_in_p,  // F[mode][col][member] at step `step_begin`\n"
          << "    f64* __restrict__ w_out_p,       // after step `step_end-1` (may be null)\n"
          << "    f64* __restrict__ vwork,         // private [workgroup][mode][64]: tangent state at the start of the current step\n"
          << "    f64* __restrict__ rec,           // F[record][mode][col][member]\n"
          << "    const f64* __restrict__ stages,  // S[(step-step_begin)*S+stage][mode][member]\n"
          << "    const f64* __restrict__ dtime, const f64* __restrict__ tab,\n"
          << "    i64 n_traj, i64 ld, i64 n_tg, i64 step_begin, i64 step_end, i64 write_steps, i64 n_records,\n"
          << "    int backward, int write_final, f64 inverse, int S)\n{\n";
        o << "    __shared__ f64 lds_all[" << ndim * MT + ndim * 64 << "];\n"
          << "    f64 (*xs)[" << MT << "] = (f64 (*)[" << MT << "])lds_all;                       // stage state of the " << MT << " members\n"
          << "    f64 (*ws)[QGS_WAVE] = (f64 (*)[QGS_WAVE])(lds_all + " << ndim * MT << ");   // tangent stage vector of the 64 pairs\n";
        o << "    const int lane = threadIdx.x & 63;\n"
          << "    const unsigned lane8 = (unsigned)lane * 8u;\n"
          << "    const unsigned ldsaddr = (unsigned)(unsigned long long)(&ws[0][0]) + lane8;                       // LDS byte address of ws[0][lane]\n"
          << "    const unsigned xladdr = (unsigned)(unsigned long long)(&xs[0][0]) + (unsigned)(lane & " << sMSK << ") * 8u;   // of xs[0][member of this lane]\n"
          << "    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n"
          << "    const i64 L = n_tg * ld;\n"
          << "    const i64 mt = (i64)blockIdx.x * " << MT << ", m0 = mt + (lane & " << sMSK << "), c0 = (i64)blockIdx.y * " << NC << " + (lane >> " << sMSH << ");\n"
          << "    const bool live = (m0 < n_traj) && (c0 < n_tg);\n"
          << "    const i64 m = m0 < n_traj ? m0 : (n_traj - 1), col = c0 < n_tg ? c0 : (n_tg - 1);\n"
          << "    const i64 l = col * ld + m;                                  // this pair's lane of F[mode][col][member]\n"
          << "    f64* const ywg = vwork + ((i64)blockIdx.y * gridDim.x + blockIdx.x) * " << ndim * 64 << ";      // this workgroup's block (uniform)\n"
          << "    // stage-state tile of the " << MT << " members, loaded by the whole workgroup: element e -> (mode e / " << MT << ", member e % " << MT << ")\n"
          << "    i64 xm = mt + (threadIdx.x & " << sMSK << "); if (xm >= n_traj) xm = n_traj - 1;\n"
          << "#define QGS_LOAD_XS(sp) do { const f64* sp_ = (sp); \\\n"
          << "        for (int e = threadIdx.x; e < " << ndim * MT << "; e += " << 64 * W << ") xs[e >> " << sMSH << "][e & " << sMSK << "] = sp_[(i64)(e >> " << sMSH << ") * ld + xm]; } while (0)\n";
    }
    for (int w = 0; w < W; ++w) {
        const std::vector<int> &own = owns[w];
        const int R = (int)own.size();
        const Plan pl = plan_for(R);
        const int ACC0 = pl.ACC0, K0 = pl.K0, T0 = pl.T0, LB1 = pl.LB1, L15 = pl.L15, RING = pl.RING, C0 = pl.C0, NS = pl.NS, half = pl.half;
        std::map<int, int> ridx;
        for (int i = 0; i < R; ++i) ridx[own[i]] = i;
        auto ACC = [&](int i) { return ACC0 + 2 * i; };
        auto KR = [&](int i) { return K0 + 2 * i; };
        o << "    " << (w == 0 ? "if" : "else if") << " (wave == " << w << ") {   // rows:";
        for (int i : own) o << " " << i;
        o << "\n";
        const char *I2 = "        ", *I3 = "            ", *I4 = "                ";
        // F: the array the records and final values of this kernel live in, row stride FL, this lane's element at Fl
        const std::string FL = fr.tangent ? "L" : "ld", Fl = fr.tangent ? "l" : "m", FLr = fr.tangent ? "Lr" : "ldr";
        for (int d : own) o << I2 << "f64 acc" << d << " = " << (fr.tangent ? "w_in_p" : "y_in") << "[" << (d - 1) << " * " << FL << " + " << Fl << "];\n";
        o << I2 << "f64* const yw = ywg + " << slot0[w] * 64 << ";      // rows of this wavefront: consecutive 512-byte lines\n";
        for (int d : own) o << I2 << (fr.tangent ? "ws[" : "xs[") << (d - 1) << "][lane] = acc" << d << "; yw[" << (slot[d] - slot0[w]) * 64 << " + lane] = acc" << d << ";\n";
        if (fr.tangent) o << I2 << "if (step_begin < step_end) QGS_LOAD_XS(stages);\n";
        o << I2 << "__syncthreads();\n";
        o << I2 << "const unsigned ywlo = (unsigned)(unsigned long long)yw, ywhi = (unsigned)((unsigned long long)yw >> 32);\n";
        o << I2 << "QGS_REC_INIT\n";
        o << I2 << "const unsigned long long kt = (unsigned long long)(kf64*)" << kname << "_kt" << w << ";\n"
          << I2 << "const unsigned ktlo = (unsigned)kt, kthi = (unsigned)(kt >> 32);\n";
        o << I2 << "for (i64 ti = step_begin; ti < step_end; ++ti) {\n";
        o << I3 << "const f64 dt = dtime[ti + 1] - dtime[ti];\n";
        o << I3 << "if (ti == next_rec) {\n"
          << I4 << "i64 " << FLr << " = " << FL << "; asm volatile(\"\" : \"+s\"(" << FLr << "));   // keeps the row offsets out of the loop-invariant set\n"
          << I4 << "f64* p = rec + qgs_rec_index(iw, n_records, backward) * " << ndim << " * " << FLr << " + " << Fl << ";\n"
          << I4 << "++iw; next_rec += write_steps;\n"
          << I4 << "if (live) {\n";
        for (int d : own) o << I4 << "    p[" << (d - 1) << " * " << FLr << "] = " << "acc" << d << ";\n";
        o << I4 << "}\n" << I3 << "}\n";
        o << "#pragma nounroll\n";
        o << I3 << "for (int st = 0; st < S; ++st) {\n";
        o << I4 << "const int last = (st == S - 1);\n";
        if (fr.tangent) {
            o << I4 << "const f64 hb = dt * tab[st] * inverse;\n";                  // inverse = +-1: exact
            o << I4 << "const f64 ha = last ? 0.0 : dt * tab[S + st] * inverse;\n";
        } else {
            o << I4 << "const f64 hb = dt * tab[st];\n";
            o << I4 << "const f64 ha = last ? 0.0 : dt * tab[S + st];\n";
            o << I4 << "if (stages && live) {\n"
              << I4 << "    i64 ldr = ld; asm volatile(\"\" : \"+s\"(ldr));\n"
              << I4 << "    f64* sp = stages + ((ti - step_begin) * S + st) * " << ndim << " * ldr + m;\n";
            for (int d : own) o << I4 << "    sp[" << (d - 1) << " * ldr] = xs[" << (d - 1) << "][lane];\n";
            o << I4 << "}\n";
        }

        // ---- the stage body --------------------------------------------------------------------------------------------------
        std::vector<PTerm> terms;
        for (int i : own) terms.insert(terms.end(), rt[i].begin(), rt[i].end());
        std::vector<Phase> phases;
        for (Phase &ph : build_phases(n_nodes, terms, pl.cap)) {
            // the greedy cover ends in a tail of phases of 2 - 4 modes (isolated factor pairs): consecutive phases whose modes fit the
            // cache together are one phase (one round of LDS reads and one wait instead of ten)
            if (!phases.empty() && opt.lds_asm_merge) {
                std::vector<int> u;
                std::set_union(phases.back().modes.begin(), phases.back().modes.end(), ph.modes.begin(), ph.modes.end(), std::back_inserter(u));
                if ((int)u.size() <= pl.cap) {
                    phases.back().modes.swap(u);
                    phases.back().terms.insert(phases.back().terms.end(), ph.terms.begin(), ph.terms.end());
                    continue;
                }
            }
            phases.push_back(std::move(ph));
        }
        const int P = (int)phases.size();
        stats.phases += P;

        std::vector<std::string> body;                       // assembly lines
        std::map<int, long> last_valu_write;
        long slot_lines = 0, slot_count = 0;
        KTable &tab = tables[w];
        int chunk = 0;                                       // coefficient chunk being consumed
        std::vector<char> touched(R, 0);
        bool lds_pending = false;
        // Vector-memory operations in flight, oldest first.  Loads return in order, so "operation T has completed" is
        // `s_waitcnt vmcnt(number of operations issued after T)`.  The early reads of the step-start state are skipped in the last
        // stage (`maybe`): a wait for anything else does not count them -- it then waits for a few of them too where they were issued.
        struct VmOp { int id; bool maybe; };
        std::vector<VmOp> vmq;
        int vm_next = 0;
        auto vm_issue = [&](bool maybe) { vmq.push_back({vm_next, maybe}); return vm_next++; };
        auto vm_wait = [&](int id, bool count_maybe) {
            size_t at = vmq.size();
            for (size_t q = 0; q < vmq.size(); ++q) if (vmq[q].id == id) { at = q; break; }
            if (at == vmq.size()) return;                     // completed by an earlier wait
            int after = 0;
            for (size_t q = at + 1; q < vmq.size(); ++q) if (count_maybe || !vmq[q].maybe) ++after;
            body.push_back("s_waitcnt vmcnt(" + std::to_string(std::min(after, 63)) + ")");
            vmq.erase(vmq.begin(), vmq.begin() + (long)at + 1);
        };
        // -- coefficients, SGPR mode: chunk c in buffer c % 2, requested when chunk c - 1 starts being consumed
        auto issue_chunk = [&](int c) {
            const int base = BUF[c % 2];
            int off = 0;
            for (int n : (CE == 16 ? std::vector<int>{16, 16} : (CE == 12 ? std::vector<int>{16, 8} : std::vector<int>{16}))) {
                body.push_back("s_load_dwordx" + std::to_string(n) + " s[" + std::to_string(base + off) + ":" + std::to_string(base + off + n - 1) +
                               "], " + sreg(KT) + ", " + std::to_string((c * CE) * 8 + off * 4));
                off += n;
            }
        };
        // -- coefficients, DPP mode: chunk c = 16 doubles, lane l of every row of 16 lanes loads entry l; ring slot c % NR
        std::vector<int> ring_op(std::max(1, NR), -1);
        int kb_block = 0;
        auto issue_ring = [&](int c) {
            if (c / 32 != kb_block) {                         // 13-bit immediate offsets: a new base every 32 chunks
                kb_block = c / 32;
                body.push_back("s_add_u32 s" + std::to_string(KB) + ", s" + std::to_string(KT) + ", " + std::to_string(4096 * kb_block));
                body.push_back("s_addc_u32 s" + std::to_string(KB + 1) + ", s" + std::to_string(KT + 1) + ", 0");
            }
            body.push_back("global_load_dwordx2 " + vreg(RING + 2 * (c % NR)) + ", v" + std::to_string(L15) + ", " + sreg(KB) + " offset:" + std::to_string((c % 32) * 128));
            ring_op[c % NR] = vm_issue(false);
        };
        int reads_done = 0, reads_total = 0;                  // LDS read instructions of the current phase: known to have returned / issued
        auto boundary = [&]() {                              // chunk `chunk` is used up
            ++chunk;
            ++n_chunks;
            if (dpp) {
                issue_ring(chunk + NR - 1);                   // into the slot of the chunk just finished (chunks chunk .. chunk + NR - 2 are there or on their way)
                vm_wait(ring_op[chunk % NR], false);
            } else {
                body.push_back("s_waitcnt lgkmcnt(0)");          // (scalar loads share the counter with the LDS reads: both have landed)
                lds_pending = false;
                reads_done = reads_total;
                issue_chunk(chunk + 1);
            }
        };
        struct Coef { std::string sgpr; int reg = 0, lane = 0; bool neg = false; };
        auto coef_operand = [&](double c, bool allow_neg = true) -> Coef {
            Coef r;
            auto at = [&](size_t q, size_t lo) {
                r.neg = std::signbit(tab.vals[q]) != std::signbit(c);
                r.lane = (int)(q - lo);
                r.reg = RING + 2 * (chunk % std::max(1, NR));
                r.sgpr = std::string(r.neg ? "-" : "") + sreg(BUF[chunk % 2] + 2 * r.lane);
            };
            const size_t lo = (size_t)chunk * CE;
            for (size_t q = lo; q < tab.vals.size(); ++q)
                if (std::fabs(tab.vals[q]) == std::fabs(c) && c != 0.0 && (allow_neg || std::signbit(tab.vals[q]) == std::signbit(c))) { at(q, lo); return r; }
            if (tab.vals.size() == lo + (size_t)CE) boundary();
            tab.vals.push_back(c);
            ++stats.coef;
            at(tab.vals.size() - 1, (size_t)chunk * CE);
            return r;
        };
        // A DPP instruction must not read a register -- accumulator and second operand included -- that a VALU instruction wrote within
        // the last two wait states.  Two wavefronts per SIMD usually put an instruction of the partner in between, but a wavefront
        // also runs alone (its partner at a barrier or in a wait): the spacing is kept by construction.  Every line of the body is
        // one issue slot; the last VALU write of every register a DPP instruction may read is remembered; what the statement order
        // (three statements round-robin) does not space is spaced by s_nop.
        auto slot_now = [&]() {
            for (; slot_lines < (long)body.size(); ++slot_lines) {
                const std::string &ln = body[slot_lines];
                if (ln[0] == '.') continue;
                slot_count += (ln.compare(0, 6, "s_nop ") == 0) ? 1 + std::atoi(ln.c_str() + 6) : 1;
            }
            return slot_count;
        };
        auto valu_wrote = [&](int reg) { last_valu_write[reg] = slot_now(); };      // right after pushing the writing instruction
        auto dpp_reads = [&](int reg) {                       // right before pushing the DPP instruction
            auto it = last_valu_write.find(reg);
            if (it == last_valu_write.end()) return;
            const long between = slot_now() - it->second;
            if (between < 2 && opt.asm_dpp_spacing) { body.push_back("s_nop " + std::to_string(1 - between)); ++n_hazard_nops; }
        };
        auto acc_k = [&](int row, double c, int src) {        // k[row] += c * src
            const Coef cf = coef_operand(c);
            if (dpp) { dpp_reads(KR(row)); dpp_reads(src); }
            if (dpp)
                body.push_back("v_fmac_f64_dpp " + vreg(KR(row)) + ", " + (cf.neg ? "-" : "") + vreg(cf.reg) + ", " + vreg(src) + " row_newbcast:" +
                               std::to_string(cf.lane) + " row_mask:0xf bank_mask:0xf");
            else if (!touched[row]) body.push_back("v_mul_f64 " + vreg(KR(row)) + ", " + cf.sgpr + ", " + vreg(src));
            else body.push_back("v_fma_f64 " + vreg(KR(row)) + ", " + cf.sgpr + ", " + vreg(src) + ", " + vreg(KR(row)));
            touched[row] = 1;
            valu_wrote(KR(row));
            ++stats.instr;
            ++wave_instr[w];
        };
        auto place = [&](const AIns &in) {
            switch (in.kind) {
            case AIns::MulT:
                body.push_back("v_mul_f64 " + vreg(in.t) + ", " + (in.neg ? "-" : "") + vreg(in.a) + ", " + vreg(in.b));
                valu_wrote(in.t);
                ++stats.instr; ++wave_instr[w];
                break;
            case AIns::FmaT:
                // (the two-address form is a 4-byte instruction: a third less code to fetch for these statements)
                if (!in.neg && opt.lds_asm_fmac) body.push_back("v_fmac_f64 " + vreg(in.t) + ", " + vreg(in.a) + ", " + vreg(in.b));
                else body.push_back("v_fma_f64 " + vreg(in.t) + ", " + (in.neg ? "-" : "") + vreg(in.a) + ", " + vreg(in.b) + ", " + vreg(in.t));
                valu_wrote(in.t);
                ++stats.instr; ++wave_instr[w];
                break;
            case AIns::AccK: acc_k(in.row, in.coef, in.src); break;
            case AIns::MovK:
                if (dpp) {                                    // k = 0 (stage start) + c0 * 1.0
                    body.push_back("v_mov_b64 " + vreg(T0) + ", 1.0");
                    valu_wrote(T0);
                    acc_k(in.row, in.coef, T0);
                } else {
                    const Coef cf = coef_operand(in.coef, false);           // (a move takes no negation)
                    body.push_back("v_mov_b64 " + vreg(KR(in.row)) + ", " + cf.sgpr);
                    touched[in.row] = 1;
                }
                break;
            case AIns::Raw: body.push_back(in.text); break;
            }
        };
        auto cache_base = [&](int p) { return C0 + 2 * ((pp && (p & 1)) ? half : 0); };
        int n_read_instr = 0;                                 // LDS read instructions of the phase whose factors are being loaded
        struct ReadIns { int slot, m0, m1; };                 // modes m0 (, m1) into cache slot(s) `slot` (, slot + 1); m1 == 0: one mode
        auto emit_reads = [&](int base, const std::vector<ReadIns> &reads) {
            for (const ReadIns &r : reads) {
                if (r.m1)                                     // two rows per instruction: offsets in units of 64 doubles
                    body.push_back("ds_read2st64_b64 v[" + std::to_string(base + 2 * r.slot) + ":" + std::to_string(base + 2 * r.slot + 3) +
                                   "], %[lds] offset0:" + std::to_string(r.m0 - 1) + " offset1:" + std::to_string(r.m1 - 1));
                else if (is_x(r.m0))                          // stage state of this lane's member (28.5 KB at most: one immediate)
                    body.push_back("ds_read_b64 " + vreg(base + 2 * r.slot) + ", %[xl] offset:" + std::to_string((int64_t)(r.m0 - ndim - 1) * MT * 8));
                else {
                    const int64_t off = (int64_t)(r.m0 - 1) * 512;
                    body.push_back("ds_read_b64 " + vreg(base + 2 * r.slot) + ", " + (off >= 65536 ? "v" + std::to_string(LB1) : std::string("%[lds]")) +
                                   " offset:" + std::to_string(off & 65535));
                }
                stats.loads += r.m1 ? 2 : 1;
            }
            n_read_instr = (int)reads.size();
            lds_pending = !reads.empty();
        };
        auto lds_reads = [&](int p, const std::vector<int> &modes) {          // `modes` of phase p into its cache half, slot i = modes[i]
            std::vector<ReadIns> reads;
            size_t i = 0;
            for (; i + 1 < modes.size(); i += 2) reads.push_back({(int)i, modes[i], modes[i + 1]});
            if (i < modes.size()) reads.push_back({(int)i, modes[i], 0});
            emit_reads(cache_base(p), reads);
        };
        auto lds_wait = [&]() { body.push_back("s_waitcnt lgkmcnt(0)"); lds_pending = false; reads_done = reads_total; };
        // global address of own row i: YB pair = yw + 4096 * (i / 8), immediate offset 512 * (i % 8)
        int yb_block = -1;
        auto ybase = [&](int i) {
            if (i / 8 != yb_block) {
                yb_block = i / 8;
                if (yb_block == 0) body.push_back("s_mov_b64 " + sreg(YB) + ", " + sreg(YW));
                else {
                    body.push_back("s_add_u32 s" + std::to_string(YB) + ", s" + std::to_string(YW) + ", " + std::to_string(4096 * yb_block));
                    body.push_back("s_addc_u32 s" + std::to_string(YB + 1) + ", s" + std::to_string(YW + 1) + ", 0");
                }
            }
            return sreg(YB) + " offset:" + std::to_string(512 * (i % 8));
        };

        // -- stage start: uniform inputs, first coefficients, factors of phase 0
        body.push_back("v_readfirstlane_b32 s" + std::to_string(KT) + ", %[ktlo]");
        body.push_back("v_readfirstlane_b32 s" + std::to_string(KT + 1) + ", %[kthi]");
        body.push_back("v_readfirstlane_b32 s" + std::to_string(YW) + ", %[ywlo]");
        body.push_back("v_readfirstlane_b32 s" + std::to_string(YW + 1) + ", %[ywhi]");
        body.push_back("v_readfirstlane_b32 s" + std::to_string(LAST) + ", %[last]");
        body.push_back("v_add_u32 v" + std::to_string(LB1) + ", 0x10000, %[lds]");
        if (dpp) body.push_back("v_and_b32 v" + std::to_string(L15) + ", 0x78, %[lane8]");
        body.push_back("s_nop 4");                            // (an SGPR written by a VALU instruction is not an address at once)
        if (dpp) {
            body.push_back("s_mov_b64 " + sreg(KB) + ", " + sreg(KT));
            for (int c = 0; c < NR; ++c) issue_ring(c);        // every slot: chunk c + NR follows into the slot chunk c leaves
            for (int i = 0; i < R; ++i) { body.push_back("v_mov_b64 " + vreg(KR(i)) + ", 0"); touched[i] = 1; valu_wrote(KR(i)); }
        } else {
            issue_chunk(0);
        }
        bool start_pending = true;                            // the first coefficients have been requested, not waited for
        // where the step-start state lands.  Early: cache slots the last phases do not use (two halves: the idle half during the last
        // phase), requested when phase `y_phase` starts; late (what does not fit there): the cache after the last phase, in batches.
        std::vector<int> yreg(R, -1), yop(R, -1);
        std::vector<int> early_regs, late_regs;
        int y_phase = P;
        if (pp && P > 0) {
            const int idle = C0 + 2 * (((P - 1) & 1) ? 0 : half), act = C0 + 2 * (((P - 1) & 1) ? half : 0);
            for (int i = 0; i < half; ++i) { early_regs.push_back(idle + 2 * i); late_regs.push_back(act + 2 * i); }
            y_phase = P - 1;
        } else if (P > 0) {
            y_phase = P - 1;                                  // (registers: the slots the last phase leaves free, known once it is planned)
        } else {
            for (int i = 0; i < NS; ++i) late_regs.push_back(C0 + 2 * i);
        }
        int y_early = 0;
        auto settle_y_regs = [&]() {
            for (int l = 0; l < NL; ++l) late_regs.push_back(T0 + 2 * l);
            y_early = std::min(R, (int)early_regs.size());
            for (int i = 0; i < y_early; ++i) yreg[i] = early_regs[i];
        };
        if (pp || P == 0) settle_y_regs();
        auto y_loads = [&](int from, int to, bool maybe) {
            for (int i = from; i < to; ++i) {
                const std::string at = ybase(i);
                body.push_back("global_load_dwordx2 " + vreg(yreg[i]) + ", %[lane8], " + at);
                yop[i] = vm_issue(maybe);
            }
        };

        // statements of a phase, factors named by mode (registers are assigned once the load order is known)
        auto statements_of = [&](const Phase &ph) {
            std::map<std::pair<int, double>, std::vector<PTerm>> pieces;
            std::vector<PTerm> singles;
            for (const PTerm &t : ph.terms) {
                if (t.j == 0) singles.push_back(t);
                else pieces[{t.row, std::fabs(t.c)}].push_back(t);
            }
            std::vector<const std::vector<PTerm> *> piece_list;
            for (auto &kv : pieces) piece_list.push_back(&kv.second);
            std::stable_sort(piece_list.begin(), piece_list.end(), [](const std::vector<PTerm> *x, const std::vector<PTerm> *y) {
                return std::fabs((*x)[0].c) < std::fabs((*y)[0].c);
            });
            std::vector<Statement> sts;
            for (const std::vector<PTerm> *gp : piece_list) {
                const std::vector<PTerm> &g = *gp;
                if (g.size() == 1) { singles.push_back(g[0]); continue; }
                Statement s;
                const bool ref_neg = std::signbit(g[0].c);
                for (size_t n = 0; n < g.size(); ++n) {
                    AIns in;
                    in.kind = n == 0 ? AIns::MulT : AIns::FmaT;
                    in.a = g[n].j; in.b = g[n].k;
                    in.neg = std::signbit(g[n].c) != ref_neg;
                    s.ins.push_back(in);
                }
                AIns in; in.kind = AIns::AccK; in.row = ridx[g[0].row]; in.coef = g[0].c; in.src = -1;
                s.ins.push_back(in);
                sts.push_back(std::move(s));
            }
            std::sort(singles.begin(), singles.end(), [](const PTerm &x, const PTerm &y) {
                return x.j != y.j ? x.j < y.j : (x.k != y.k ? x.k < y.k : x.row < y.row);
            });
            for (size_t a = 0; a < singles.size();) {
                size_t b = a;
                while (b < singles.size() && singles[b].j == singles[a].j && singles[b].k == singles[a].k) ++b;
                Statement s;
                const PTerm &t0 = singles[a];
                if (t0.j != 0) {
                    AIns in; in.kind = AIns::MulT; in.a = t0.j; in.b = t0.k;
                    s.ins.push_back(in);
                }
                for (size_t q = a; q < b; ++q) {
                    AIns in; in.kind = AIns::AccK; in.row = ridx[singles[q].row]; in.coef = singles[q].c;
                    in.src = t0.j == 0 ? t0.k : -1;         // a mode, or the statement's temporary
                    s.ins.push_back(in);
                }
                sts.push_back(std::move(s));
                a = b;
            }
            return sts;
        };
        auto modes_of = [](const Statement &s) {
            std::vector<int> m;
            for (const AIns &in : s.ins) {
                if (in.kind == AIns::MulT || in.kind == AIns::FmaT) { m.push_back(in.a); m.push_back(in.b); }
                else if (in.kind == AIns::AccK && in.src > 0) m.push_back(in.src);
            }
            return m;
        };

        std::vector<int> resident(NS, 0);                      // single cache set: the mode every slot holds (0: none)
        for (int p = 0; p < P; ++p) {
            const Phase &ph = phases[p];
            std::vector<Statement> sts = statements_of(ph);
            const bool progressive = !pp && opt.lds_asm_progressive;
            std::map<int, int> slot_of, read_of;                // mode -> cache slot; mode -> index of the read instruction that brings it (-1: resident)
            std::vector<ReadIns> reads;
            if (pp) {
                // load order of the factors: by first use
                std::vector<int> order;
                for (const Statement &st : sts) for (int m : modes_of(st)) if (!slot_of.count(m)) { slot_of[m] = (int)order.size(); order.push_back(m); }
                for (auto &kv : slot_of) read_of[kv.first] = kv.second / 2;
                if (p == 0) lds_reads(0, order);
                if (lds_pending) { lds_wait(); ++n_extra_waits; }
            } else {
                // One cache set: a mode the previous phases left in a slot stays there (and is not read again); statements whose factors
                // are all resident come first -- they run while the new factors are on their way -- the others in the order their last
                // factor arrives; every instruction waits only for the reads it needs (LDS returns in order; DPP mode: nothing else on
                // the counter).
                std::vector<char> wanted(n_nodes + 1, 0);
                for (int m : ph.modes) wanted[m] = 1;
                std::vector<int> free_slots;
                for (int q = 0; q < NS; ++q) {
                    if (resident[q] && wanted[resident[q]] && opt.lds_asm_keep) { slot_of[resident[q]] = q; read_of[resident[q]] = -1; }
                    else { resident[q] = 0; free_slots.push_back(q); }
                }
                auto is_new = [&](const Statement &st) { for (int m : modes_of(st)) if (!slot_of.count(m)) return true; return false; };
                if (progressive) std::stable_partition(sts.begin(), sts.end(), [&](const Statement &st) { return !is_new(st); });
                std::vector<int> fresh;                          // new modes by first use
                {
                    std::map<int, int> seen;
                    for (const Statement &st : sts) for (int m : modes_of(st)) if (!slot_of.count(m) && !seen.count(m)) { seen[m] = 1; fresh.push_back(m); }
                }
                if (fresh.size() > free_slots.size()) throw std::logic_error("codegen: phase does not fit the factor cache");
                if (fr.tangent) {                                // every product has one factor of each kind: two vector components
                    std::vector<int> fw, fx, merged;             // (one instruction), then two state values, and so on
                    for (int m : fresh) (is_x(m) ? fx : fw).push_back(m);
                    for (size_t a = 0, b = 0; a < fw.size() || b < fx.size();) {
                        for (int n = 0; n < 2 && a < fw.size(); ++n) merged.push_back(fw[a++]);
                        for (int n = 0; n < 2 && b < fx.size(); ++n) merged.push_back(fx[b++]);
                    }
                    fresh.swap(merged);
                }
                size_t f = 0, q = 0;
                while (f < fresh.size()) {                       // adjacent free slots take two modes per instruction
                    if (f + 1 < fresh.size() && q + 1 < free_slots.size() && free_slots[q + 1] == free_slots[q] + 1 && !is_x(fresh[f]) && !is_x(fresh[f + 1])) {
                        reads.push_back({free_slots[q], fresh[f], fresh[f + 1]});
                        q += 2; f += 2;
                    } else {
                        reads.push_back({free_slots[q], fresh[f], 0});
                        q += 1; f += 1;
                    }
                }
                for (size_t r = 0; r < reads.size(); ++r) {
                    slot_of[reads[r].m0] = reads[r].slot; read_of[reads[r].m0] = (int)r; resident[reads[r].slot] = reads[r].m0;
                    if (reads[r].m1) { slot_of[reads[r].m1] = reads[r].slot + 1; read_of[reads[r].m1] = (int)r; resident[reads[r].slot + 1] = reads[r].m1; }
                }
                if (progressive) {
                    auto last_read = [&](const Statement &st) { int v = -1; for (int m : modes_of(st)) v = std::max(v, read_of[m]); return v; };
                    std::stable_sort(sts.begin(), sts.end(), [&](const Statement &x, const Statement &y) { return last_read(x) < last_read(y); });
                }
                emit_reads(C0, reads);
                reads_total = n_read_instr;
                reads_done = 0;
                if (!progressive && !reads.empty()) { lds_wait(); ++n_extra_waits; }
                if (p == P - 1) {                               // the step-start state: into the slots this last phase does not use
                    for (int q2 = 0; q2 < NS; ++q2) (resident[q2] && wanted[resident[q2]] ? late_regs : early_regs).push_back(C0 + 2 * q2);
                    settle_y_regs();
                }
            }
            if (start_pending) {                              // first phase: the first coefficients
                if (dpp) vm_wait(ring_op[0], false);
                else { lds_wait(); issue_chunk(1); }
                start_pending = false;
                for (int i = 0; i < R; ++i) {
                    if (c0[own[i]] != 0.0) { AIns in; in.kind = AIns::MovK; in.row = i; in.coef = c0[own[i]]; place(in); }
                }
            }
            const int base = cache_base(p);
            std::map<int, int> xreg;
            for (auto &kv : slot_of) xreg[kv.first] = base + 2 * kv.second;
            if (pp && p + 1 < P) {                             // the next phase's factors into the idle half, in ITS order of first use
                std::vector<Statement> nx = statements_of(phases[p + 1]);
                std::vector<int> no; std::map<int, int> seen;
                for (const Statement &s : nx) for (int m : modes_of(s)) if (!seen.count(m)) { seen[m] = 1; no.push_back(m); }
                // (the same order `assign` finds when that phase is emitted: sts are rebuilt from the same terms)
                lds_reads(p + 1, no);
            }
            if (p == y_phase && y_early > 0) {
                body.push_back("s_cmp_lg_u32 s" + std::to_string(LAST) + ", 0");
                body.push_back("s_cbranch_scc1 .Lqgs_ny%=");
                yb_block = -1;
                y_loads(0, y_early, true);
                body.push_back(".Lqgs_ny%=:");
            }
            // NL statements in flight, their instructions round-robin (independent fp64 dependency chains)
            if (pp) { reads_total = n_read_instr; reads_done = n_read_instr; }
            auto need = [&](int read) {                        // read instruction `read` of this phase has returned (-1: resident)
                const int q = read + 1;                        // read instructions needed
                if (q <= reads_done) return;
                // (4-bit counter: at most 15 may stay outstanding; SGPR mode: scalar loads may be on the counter too, and they return
                // out of order: everything)
                const int allow = dpp ? std::min(15, reads_total - q) : 0;
                body.push_back("s_waitcnt lgkmcnt(" + std::to_string(allow) + ")");
                reads_done = reads_total - allow;
                if (reads_done >= reads_total) lds_pending = false;
            };
            std::vector<size_t> cur(NL), pos(NL, 0);
            size_t next = 0;
            for (int l = 0; l < NL; ++l) cur[l] = next < sts.size() ? next++ : (size_t)-1;
            for (bool any = true; any;) {
                any = false;
                for (int l = 0; l < NL; ++l) {
                    if (cur[l] == (size_t)-1) continue;
                    AIns in = sts[cur[l]].ins[pos[l]++];
                    in.t = T0 + 2 * l;
                    if (in.kind == AIns::MulT || in.kind == AIns::FmaT) {
                        if (progressive) need(std::max(read_of[in.a], read_of[in.b]));
                        in.a = xreg[in.a]; in.b = xreg[in.b];
                    } else if (in.kind == AIns::AccK) {
                        if (in.src > 0) { if (progressive) need(read_of[in.src]); in.src = xreg[in.src]; }
                        else in.src = in.t;
                    }
                    place(in);
                    any = true;
                    if (pos[l] == sts[cur[l]].ins.size()) { cur[l] = next < sts.size() ? next++ : (size_t)-1; pos[l] = 0; }
                }
            }
            if (progressive && reads_done < reads_total) { lds_wait(); }      // (a phase without statements cannot happen; keeps the counter exact)
            lds_pending = pp && p + 1 < P;
        }
        if (start_pending) {                                  // no phase at all (cannot happen for a tensor with terms)
            if (dpp) vm_wait(ring_op[0], false); else { lds_wait(); issue_chunk(1); }
        }
        for (int i = 0; i < R; ++i)
            if (!touched[i]) body.push_back("v_mov_b64 " + vreg(KR(i)) + ", 0");
        // -- end of the stage: acc += hb k; next stage state = y + ha k (or the new state after the last stage)
        for (int i = 0; i < R; ++i)
            body.push_back("v_fma_f64 " + vreg(ACC(i)) + ", %[hb], " + vreg(KR(i)) + ", " + vreg(ACC(i)));
        body.push_back("s_cmp_lg_u32 s" + std::to_string(LAST) + ", 0");
        body.push_back("s_cbranch_scc1 .Lqgs_last%=");
        yb_block = -1;
        {
            const int B = (int)late_regs.size();
            int issued_to = std::min(R, y_early + B);
            for (int i = y_early; i < issued_to; ++i) yreg[i] = late_regs[(i - y_early) % B];
            y_loads(y_early, issued_to, false);
            for (int i = 0; i < R; ++i) {
                if (i >= issued_to) {                      // next batch: its registers are those of rows consumed a batch ago
                    const int to = std::min(R, issued_to + B);
                    for (int q = issued_to; q < to; ++q) yreg[q] = late_regs[(q - y_early) % B];
                    y_loads(issued_to, to, false);
                    issued_to = to;
                }
                vm_wait(yop[i], true);
                body.push_back("v_fma_f64 " + vreg(KR(i)) + ", %[ha], " + vreg(KR(i)) + ", " + vreg(yreg[i]));
            }
        }
        body.push_back("s_barrier");
        auto xs_write = [&](int reg0) {
            for (int i = 0; i < R; ++i) {
                const int64_t off = (int64_t)(own[i] - 1) * 512;
                body.push_back("ds_write_b64 " + (off >= 65536 ? "v" + std::to_string(LB1) : std::string("%[lds]")) + ", " + vreg(reg0 + 2 * i) +
                               " offset:" + std::to_string(off & 65535));
            }
        };
        xs_write(K0);
        body.push_back("s_branch .Lqgs_join%=");
        body.push_back(".Lqgs_last%=:");
        yb_block = -1;
        for (int i = 0; i < R; ++i) {
            const std::string at = ybase(i);
            body.push_back("global_store_dwordx2 %[lane8], " + vreg(ACC(i)) + ", " + at);
        }
        body.push_back("s_barrier");
        xs_write(ACC0);
        body.push_back(".Lqgs_join%=:");
        // nothing is left in flight (coefficient chunks requested ahead of the table's end included)
        body.push_back("s_waitcnt vmcnt(0) lgkmcnt(0)");
        if (!fr.tangent) body.push_back("s_barrier");         // (tangent frame: the workgroup loads the next stage state first)
        tab.pad_to = ((size_t)chunk + (dpp ? NR + 1 : 2)) * CE;

        o << I4 << "// " << R << " rows, cache " << NS << " slots, phases of <= " << pl.cap << " modes: " << P << " phases, " << wave_instr[w]
          << " fp64 instructions\n";
        o << I4 << "asm volatile(\n";
        for (const std::string &ln : body) {
            // (timing experiments of the developer build: the body without its barriers / LDS waits / vector-memory waits -- wrong results)
            if ((opt.lds_asm_skip & 1) && ln == "s_barrier") continue;
            if ((opt.lds_asm_skip & 2) && ln.compare(0, 18, "s_waitcnt lgkmcnt(") == 0) continue;
            if ((opt.lds_asm_skip & 4) && ln.compare(0, 16, "s_waitcnt vmcnt(") == 0 && ln.find("lgkmcnt") == std::string::npos) continue;
            if ((opt.lds_asm_skip & 8) && (ln.compare(0, 7, "ds_read") == 0 || ln.compare(0, 11, "global_load") == 0)) continue;      // no loads at all
            if ((opt.lds_asm_skip & 16) && ln.compare(0, 6, "s_nop ") == 0 && ln != "s_nop 4") continue;                                // no spacing
            if ((opt.lds_asm_skip & 32) && ln.find("_dpp ") != std::string::npos) {                                                     // plain FMA instead of the DPP form
                std::string t = ln.substr(0, ln.find(" row_newbcast"));
                t.replace(t.find("v_fmac_f64_dpp"), 14, "v_fmac_f64");
                o << I4 << "    \"" << t << "\\n\"\n";
                continue;
            }
            o << I4 << "    \"" << ln << "\\n\"\n";
        }
        o << I4 << "    :";
        for (int i = 0; i < R; ++i) o << (i ? ", " : " ") << "\"+{" << vreg(ACC(i)) << "}\"(acc" << own[i] << ")";
        o << "\n" << I4 << "    : [lds] \"v\"(ldsaddr), [lane8] \"v\"(lane8), [hb] \"v\"(hb), [ha] \"v\"(ha), [ktlo] \"v\"(ktlo), [kthi] \"v\"(kthi), "
          << "[ywlo] \"v\"(ywlo), [ywhi] \"v\"(ywhi), [last] \"v\"(last)" << (fr.tangent ? ", [xl] \"v\"(xladdr)" : "") << "\n";
        o << I4 << "    :";
        bool first = true;
        for (int r = K0; r < VT; ++r) { o << (first ? " " : ", ") << "\"v" << r << "\""; first = false; }
        for (int r = SF; r < LAST + 1; ++r) o << ", \"s" << r << "\"";
        o << ", \"scc\", \"memory\");\n";
        if (fr.tangent)                                          // stage state of the next stage (or of the first stage of the next step)
            o << I4 << "{\n"
              << I4 << "    const i64 nxt = (ti - step_begin) * S + st + 1;\n"
              << I4 << "    if (nxt < (step_end - step_begin) * S) QGS_LOAD_XS(stages + nxt * " << ndim << " * ld);\n"
              << I4 << "}\n"
              << I4 << "__syncthreads();\n";
        o << I3 << "}\n";
        o << I2 << "}\n";
        const std::string Fout = fr.tangent ? "w_out_p" : "y_out";
        o << I2 << "if (live) {\n" << I3 << "if (" << Fout << ") {\n";
        for (int d : own) o << I4 << Fout << "[" << (d - 1) << " * " << FL << " + " << Fl << "] = " << "acc" << d << ";\n";
        o << I3 << "}\n" << I3 << "if (write_final) {\n"
          << I4 << "f64* p = rec + qgs_rec_index(n_records - 1, n_records, backward) * " << ndim << " * " << FL << " + " << Fl << ";\n";
        for (int d : own) o << I4 << "p[" << (d - 1) << " * " << FL << "] = " << "acc" << d << ";\n";
        o << I3 << "}\n" << I2 << "}\n    }\n";
    }
    if (fr.tangent) o << "#undef QGS_LOAD_XS\n}\n";
    else o << "    QGS_CLOCK_MARK(2)\n}\n";
    out << "// per stage and 64 " << (fr.tangent ? "(member, column) pairs: " : "members: ") << stats.phases << " phases, " << stats.loads << " LDS reads, " << stats.instr
        << " fp64 instructions, " << stats.coef << " coefficient table entries in " << n_chunks + W << " chunks, "
        << n_extra_waits << " further waits for LDS reads, " << n_hazard_nops << " s_nop for the DPP read-after-write spacing\n";
    {
        int64_t mx = 0, sum = 0;
        for (int64_t v : wave_instr) { mx = std::max(mx, v); sum += v; }
        out << "// fp64 instructions of the wavefronts: max / mean = " << (sum ? (double)mx * W / (double)sum : 0.0) << "\n";
    }
    for (int w = 0; w < W; ++w) emit_ktable(out, kname + "_kt" + std::to_string(w), tables[w]);
    out << o.str();
}

}  // namespace

void emit_rk_lds_asm_kernel(std::ostringstream &out, int ndim, const std::vector<Row> &rows, const CodegenOptions &opt)
{
    RowTerms rt(ndim + 1);
    std::vector<double> c0(ndim + 1, 0.0);
    for (int i = 1; i <= ndim; ++i) {
        for (const Lin &l : rows[i].lin) rt[i].push_back({i, 0, l.k, l.c});
        for (const Bil &b : rows[i].bil) rt[i].push_back({i, std::min(b.j, b.k), std::max(b.j, b.k), b.c});
        if (rows[i].has_c0) c0[i] = rows[i].c0;
    }
    emit_lds_asm(out, ndim, rt, c0, AsmFrame{}, opt);
}

// (J w)_i = sum_{j,k} Tj_ijk x_k w_j, or the transpose: the rows of `wx` (build_wx_rows), factors as nodes w_j < x_k
void emit_tgl_lds_asm_kernel(std::ostringstream &out, int ndim, const std::vector<std::vector<WX>> &wx, bool adjoint, const CodegenOptions &opt)
{
    RowTerms rt(ndim + 1);
    for (int i = 1; i <= ndim; ++i)
        for (const WX &t : wx[i]) {
            if (t.x == 0) rt[i].push_back({i, 0, t.w, t.c});                 // x_0 = 1: c * w_j
            else rt[i].push_back({i, t.w, ndim + t.x, t.c});
        }
    AsmFrame fr;
    fr.tangent = true;
    fr.adjoint = adjoint;
    emit_lds_asm(out, ndim, rt, std::vector<double>(ndim + 1, 0.0), fr, opt);
}

}  // namespace detail
}  // namespace qgs
