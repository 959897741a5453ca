// codegen_tangent_asm.cpp -- the register-resident tangent / adjoint kernel (one lane per (member, column), stage record in mode
// pairs) with a hand-scheduled body: qgs_spec_tglpa_s<S>.  See codegen.h / codegen_internal.h; the arithmetic and the interface are
// those of qgs_spec_tglp_s<S> (emit_tgl_kernel in codegen_tangent.cpp).
//
// Why.  The compiler-scheduled kernel holds five 36-vectors (step-start vector, running sum, two tangent stage vectors, the stage
// state) in 380 registers: one wavefront per SIMD, 405 accumulation-register moves per column-step, and -- with nothing else to
// run -- every stage starts by waiting for its 36 stage-state loads (a quarter of the wave-cycles are waits).  Here the generator
// allocates the registers itself:
//   * architectural registers: two banks of 36 doubles for the tangent stage vectors (the input of a stage and its output; they swap
//     roles every stage), 36 doubles for the stage state, 13 doubles of temporaries: 242 registers, no spills;
//   * the stage state of the NEXT stage is requested at the start of a stage straight into the accumulation registers (vector loads
//     may target them) and moved over when the stage is done: the load latency is hidden behind ~900 fp64 instructions, at the price
//     of 72 v_accvgpr_read_b32 per stage (288 per step, the compiler's kernel spends 405 and still waits);
//   * the step-start vector v and the running sum live in LDS as pairs of rows (one 128-bit access per pair), read at the START of
//     the two rows that need them at their END;
//   * coefficients through vector loads into a ring of registers + DPP broadcast (codegen_lds_asm.cpp), requested NR - 1 chunks
//     ahead across stage and step boundaries, so no wait in the kernel is `lgkmcnt(0)` on a scalar load;
//   * all steps up to the next record are one assembly statement (a loop): the only exposed memory latency is that of its first
//     stage state.
#include "codegen_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <functional>
#include <map>
#include <sstream>
#include <stdexcept>

namespace qgs {
namespace detail {

namespace {

std::string vreg(int r) { return "v[" + std::to_string(r) + ":" + std::to_string(r + 1) + "]"; }
std::string vreg4(int r) { return "v[" + std::to_string(r) + ":" + std::to_string(r + 3) + "]"; }
std::string sreg(int r) { return "s[" + std::to_string(r) + ":" + std::to_string(r + 1) + "]"; }

// one instruction of a row's evaluation, registers still symbolic
struct TIns {
    enum Kind { Mul, Fma, Acc } kind;   // g = x * w; g = fma(+-x, w, g); r += c * (g | w)
    int x = 0, w = 0;                   // Mul / Fma: stage-state mode, tangent-vector mode
    bool neg = false;
    bool src_w = false;                 // Acc: the factor is the tangent component w (a linear term), else the temporary g
    int entry = 0;                      // Acc: coefficient table entry
    int tmp = 0;                        // which of the row's two product temporaries (statements alternate)
    bool cneg = false;                  // Acc: the entry enters negated
};

}  // namespace

bool tgl_asm_applies(int ndim, int S, const CodegenOptions &opt)
{
    return opt.tgl_asm && opt.tgl_pair && S >= 2 && S <= 4 && 6 * ndim + 20 + 2 * std::max(2, opt.tgl_asm_ring) + 3 <= 252;
}

void emit_tgl_asm_kernel(std::ostringstream &out, int ndim, const std::vector<std::vector<WX>> &tgl,
                         const std::vector<std::vector<WX>> &adj, int S, const CodegenOptions &opt)
{
    if (S < 2 || S > 4) throw std::logic_error("codegen: the hand-scheduled tangent kernel takes 2 to 4 stages");
    const int NR = std::max(2, opt.tgl_asm_ring), CE = 16;
    const int NP = (ndim + 1) / 2;                           // row pairs
    // architectural registers
    const int BANK[2] = {0, 2 * ndim}, X0 = 4 * ndim, T0 = 6 * ndim;
    const int R0 = T0, G0 = T0 + 4, SETA = T0 + 12, SETV = T0 + 16, RING = T0 + 20, L15 = RING + 2 * NR, MO8 = L15 + 1, LDSA = MO8 + 1, VEND = LDSA + 1;
    // (G0 .. G0 + 7: two product temporaries per row of the pair in flight -- a statement's sum is consumed by a DPP instruction, which
    // must not read a register a VALU instruction wrote within the last two wait states: the next statement starts in the other
    // temporary before the coefficient is applied to this one)
    if (VEND > 252) throw std::logic_error("codegen: the hand-scheduled tangent kernel does not fit 256 registers at this dimension");
    // scalar registers (all written here from v_readfirstlane_b32 of inputs that arrive in VGPRs)
    const int SB = 40, RUN = SB, XP = SB + 2, XQ = SB + 4, DTP = SB + 6, TAB = SB + 8, KT = SB + 10, KB = SB + 12, INV = SB + 14, STR = SB + 16,
              XSTG = SB + 17, HB = SB + 18, DT0 = SB + 20, TABV = SB + 24, HA = TABV + 16, DTS = HA + 2, SEND = DTS + 2;
    const std::string kname = "qgs_spec_tglpa_s" + std::to_string(S);

    std::ostringstream o;
    KTable tables[2];
    o << "\n// tangent (adjoint=0) / adjoint (adjoint=1) model, " << S << "-stage RK, one lane per (member, column), stage record in mode pairs;\n"
      << "// body hand-scheduled: " << VEND << " registers + " << 2 * ndim << " accumulation registers, coefficient ring of " << NR << " chunks\n";
    o << "extern \"C\" __global__ void __launch_bounds__(64, 1) " << kname << "(\n"
      << "    const f64* __restrict__ w_in_p,  // F[mode][col][member] at step `step_begin`\n"
      << "    f64* __restrict__ w_out_p,       // after step `step_end-1` (may be null)\n"
      << "    f64* __restrict__ rec,           // F[record][mode][col][member]\n"
      << "    const f64* __restrict__ stages,  // S[(step-step_begin)*" << S << "+stage][mode/2][member][2]\n"
      << "    const f64* __restrict__ dtime, const f64* __restrict__ tab,\n"
      << "    i64 n_traj, i64 ld, i64 n_tg, i64 step_begin, i64 step_end, i64 write_steps, i64 n_records,\n"
      << "    int backward, int write_final, int adjoint, f64 inverse)\n{\n";
    o << "    __shared__ qgs_d2 vpk[2 * " << NP << "][QGS_WAVE];      // [0, " << NP << "): step-start vector, [" << NP << ", ...): running sum; pairs of rows\n";
    o << "    __shared__ unsigned long long qargs[6];               // uniform inputs of the assembly statement (three registers of operands instead of fourteen)\n";
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
    o << "    QGS_CLOCK_MARK(0)\n";
    o << "    " << decl_list("v", ndim) << "\n";
    for (int d = 1; d <= ndim; ++d) o << "    v" << d << " = w_in_p[" << (d - 1) << " * L + l];\n";
    o << "    const unsigned ldsv = (unsigned)(unsigned long long)(&vpk[0][0]) + (unsigned)lane * 16u;\n"
      << "    const unsigned mo = (unsigned)m * 16u;                 // this member's pair inside a pair row of the stage record\n"
      << "    const unsigned strd = (unsigned)ld * 16u;              // bytes between pair rows\n"
      << "    const unsigned ldsq = (unsigned)(unsigned long long)(&qargs[0]);\n";
    o << "    QGS_REC_INIT\n";
    o << "    i64 ti = step_begin;\n";
    o << "    while (ti < step_end) {\n";
    o << "        if (ti == next_rec) {\n"
      << "            f64* p = rec + qgs_rec_index(iw, n_records, backward) * " << ndim << " * L + l;\n"
      << "            ++iw; next_rec += write_steps;\n"
      << "            if (live) {\n";
    for (int d = 1; d <= ndim; ++d) o << "                p[" << (d - 1) << " * L] = v" << d << ";\n";
    o << "            }\n        }\n";
    o << "        i64 run64 = step_end - ti;                          // steps up to the next record (or the end): one assembly statement\n"
      << "        if (next_rec > ti && next_rec - ti < run64) run64 = next_rec - ti;\n"
      << "        const unsigned run = (unsigned)(run64 > 1000000 ? 1000000 : run64);\n"
      << "        qargs[0] = (unsigned long long)(stages + (ti - step_begin) * " << S * ndim << " * ld);\n"
      << "        qargs[1] = (unsigned long long)(dtime + ti);\n"
      << "        qargs[2] = (unsigned long long)tab;\n"
      << "        qargs[4] = __builtin_bit_cast(unsigned long long, inverse);\n"
      << "        qargs[5] = (unsigned long long)strd | ((unsigned long long)run << 32);\n";

    for (int pass = 0; pass < 2; ++pass) {
        const std::vector<std::vector<WX>> &rowsrc = pass == 0 ? tgl : adj;
        KTable &tab = tables[pass];
        // ---- the rows as symbolic instruction lists; coefficient entries in order of first use (chunks of 16, de-duplicated inside a chunk)
        std::vector<std::vector<TIns>> prog(ndim + 1);
        {
            // the order in which the instructions will run: pair by pair, the two rows of a pair round-robin
            std::vector<std::vector<TIns>> raw(ndim + 1);
            std::vector<double> values;                       // coefficient of every Acc instruction (entry = index here until placed)
            for (int i = 1; i <= ndim; ++i) {
                std::vector<WX> lin, bil;
                for (const WX &a : rowsrc[i]) (a.x == 0 ? lin : bil).push_back(a);
                std::vector<TIns> lins, grp;
                for (const WX &a : lin) {
                    TIns t; t.kind = TIns::Acc; t.src_w = true; t.w = a.w; t.entry = (int)values.size();
                    values.push_back(a.c);
                    lins.push_back(t);
                }
                int k = 0;
                for (auto &g : group_by_abs(bil)) {
                    const bool ref_neg = std::signbit(g[0].c);
                    for (size_t n = 0; n < g.size(); ++n) {
                        TIns t; t.kind = n == 0 ? TIns::Mul : TIns::Fma; t.x = g[n].x; t.w = g[n].w; t.neg = std::signbit(g[n].c) != ref_neg; t.tmp = k & 1;
                        grp.push_back(t);
                    }
                    TIns t; t.kind = TIns::Acc; t.src_w = false; t.entry = (int)values.size(); t.tmp = k & 1;
                    values.push_back(g[0].c);
                    grp.push_back(t);
                    ++k;
                }
                // the coefficient of statement k is applied after the first instruction of statement k + 1 (which writes the other
                // temporary): two instructions of this row -- four issue slots with the partner row -- between the last write of a sum and
                // the DPP instruction that reads it
                for (size_t q = 0; q + 1 < grp.size(); ++q)
                    if (grp[q].kind == TIns::Acc && grp[q + 1].kind == TIns::Mul) { std::swap(grp[q], grp[q + 1]); ++q; }
                // linear terms (a DPP instruction each, all accumulating into r, and all BEFORE the first sum -- the order of the additions
                // into r is the compiler-scheduled kernel's, results stay bitwise equal): one behind every product instruction ahead of
                // the first accumulation of a sum, the rest in a row just before it (the hazard bookkeeping spaces those)
                size_t first_acc = grp.size();
                for (size_t q = 0; q < grp.size(); ++q) if (grp[q].kind == TIns::Acc) { first_acc = q; break; }
                std::vector<TIns> fin;
                size_t nl = 0;
                for (size_t q = 0; q < grp.size(); ++q) {
                    if (q == first_acc) while (nl < lins.size()) fin.push_back(lins[nl++]);
                    fin.push_back(grp[q]);
                    if (q < first_acc && nl < lins.size()) fin.push_back(lins[nl++]);
                }
                while (nl < lins.size()) fin.push_back(lins[nl++]);
                raw[i] = fin;
            }
            // placement in execution order: a coefficient whose magnitude already sits in the chunk being consumed is not stored again
            auto place = [&](TIns &t) {
                const double c = values[t.entry];
                const size_t lo = tab.vals.empty() ? 0 : (tab.vals.size() - 1) / CE * CE;
                for (size_t q = lo; q < tab.vals.size(); ++q)
                    if (std::fabs(tab.vals[q]) == std::fabs(c) && c != 0.0) { t.entry = (int)q; t.cneg = std::signbit(tab.vals[q]) != std::signbit(c); return; }
                t.entry = (int)tab.vals.size();
                t.cneg = false;
                tab.vals.push_back(c);
            };
            for (int q = 0; q < NP; ++q) {
                const int i0 = 2 * q + 1, i1 = std::min(ndim, 2 * q + 2);
                size_t p0 = 0, p1 = 0;
                while (p0 < raw[i0].size() || (i1 != i0 && p1 < raw[i1].size())) {
                    if (p0 < raw[i0].size()) { TIns t = raw[i0][p0++]; if (t.kind == TIns::Acc) place(t); prog[i0].push_back(t); }
                    if (i1 != i0 && p1 < raw[i1].size()) { TIns t = raw[i1][p1++]; if (t.kind == TIns::Acc) place(t); prog[i1].push_back(t); }
                }
            }
        }
        int NC = (int)((tab.vals.size() + CE - 1) / CE);
        NC = std::max(NR, (NC + NR - 1) / NR * NR);           // whole ring turns per stage: the slot of a chunk is the same in every stage
        tab.pad_to = (size_t)NC * CE;

        std::vector<std::string> body;
        struct VmOp { int id; };
        std::vector<VmOp> vmq;
        int vm_next = 0;
        auto vm_issue = [&]() { vmq.push_back({vm_next}); return vm_next++; };
        auto vm_wait = [&](int id) {
            size_t at = vmq.size();
            for (size_t q = 0; q < vmq.size(); ++q) if (vmq[q].id == id) { at = q; break; }
            if (at == vmq.size()) return;
            body.push_back("s_waitcnt vmcnt(" + std::to_string(std::min<int>((int)(vmq.size() - at - 1), 63)) + ")");
            vmq.erase(vmq.begin(), vmq.begin() + (long)at + 1);
        };
        // A DPP instruction must not read a register (its accumulator and its second operand included) that a VALU instruction wrote
        // within the last two wait states; a lone wavefront issues back to back, so this is kept by construction: every line of the
        // body is one issue slot, the last VALU write of the registers that DPP instructions read is remembered, and what the
        // instruction order does not space is spaced by s_nop.
        std::map<int, long> last_valu_write;
        auto slots = [&]() { long n = 0; for (const std::string &ln : body) if (ln[0] != '.') n += (ln.compare(0, 6, "s_nop ") == 0) ? 1 + std::atoi(ln.c_str() + 6) : 1; return n; };
        long slot_cache_lines = 0, slot_cache = 0;
        auto slot_now = [&]() {                               // issue slots so far (incremental over `body`)
            for (; slot_cache_lines < (long)body.size(); ++slot_cache_lines) {
                const std::string &ln = body[slot_cache_lines];
                if (ln[0] == '.') continue;
                slot_cache += (ln.compare(0, 6, "s_nop ") == 0) ? 1 + std::atoi(ln.c_str() + 6) : 1;
            }
            return slot_cache;
        };
        (void)slots;
        auto valu_wrote = [&](int reg) { last_valu_write[reg] = slot_now(); };      // call right after pushing the writing instruction
        auto dpp_reads = [&](int reg) {                       // call right before pushing the DPP instruction
            auto it = last_valu_write.find(reg);
            if (it == last_valu_write.end()) return;
            const long between = slot_now() - it->second;     // instructions issued since the write
            if (between < 2 && opt.asm_dpp_spacing) body.push_back("s_nop " + std::to_string(1 - between));
        };
        // coefficient ring: chunk k of a stage in slot k % NR (NC is a multiple of NR)
        std::vector<int> ring_op(NR, -1);
        auto issue_ring = [&](int k) {                        // chunk k % NC of the table
            const int kk = k % NC;
            body.push_back("global_load_dwordx2 " + vreg(RING + 2 * (kk % NR)) + ", v" + std::to_string(L15) + ", " + sreg(kk < 32 ? KT : KB) + " offset:" + std::to_string((kk % 32) * 128));
            ring_op[kk % NR] = vm_issue();
        };
        if (NC > 64) throw std::logic_error("codegen: coefficient table of the tangent kernel beyond the immediate offsets");
        std::vector<int> xop;
        auto issue_x = [&]() {                                // the stage state at XP into the accumulation registers; XP moves on by one stage
            body.push_back("s_mov_b64 " + sreg(XQ) + ", " + sreg(XP));
            xop.clear();
            for (int q = 0; q < ndim / 2; ++q) {
                body.push_back("global_load_dwordx4 a[" + std::to_string(4 * q) + ":" + std::to_string(4 * q + 3) + "], %[mo], " + sreg(XQ));
                xop.push_back(vm_issue());
                body.push_back("s_add_u32 s" + std::to_string(XQ) + ", s" + std::to_string(XQ) + ", s" + std::to_string(STR));
                body.push_back("s_addc_u32 s" + std::to_string(XQ + 1) + ", s" + std::to_string(XQ + 1) + ", 0");
            }
            if (ndim & 1) {                                   // odd last mode on its own: S[..][ndim - 1][member], member offset in doubles
                body.push_back("global_load_dwordx2 a[" + std::to_string(2 * (ndim - 1)) + ":" + std::to_string(2 * (ndim - 1) + 1) + "], v" + std::to_string(MO8) + ", " + sreg(XQ));
                xop.push_back(vm_issue());
            }
            body.push_back("s_add_u32 s" + std::to_string(XP) + ", s" + std::to_string(XP) + ", s" + std::to_string(XSTG));
            body.push_back("s_addc_u32 s" + std::to_string(XP + 1) + ", s" + std::to_string(XP + 1) + ", 0");
        };

        // ---- set-up
        auto rfl = [&](int sdst, int vsrc) { body.push_back("v_readfirstlane_b32 s" + std::to_string(sdst) + ", v" + std::to_string(vsrc)); };
        body.push_back("ds_read_b128 " + vreg4(R0) + ", %[ldsq]");                     // stage record, time grid
        body.push_back("ds_read_b128 " + vreg4(R0 + 4) + ", %[ldsq] offset:16");       // tableau, coefficient table
        body.push_back("ds_read_b128 " + vreg4(R0 + 8) + ", %[ldsq] offset:32");       // inverse, (stride, run)
        body.push_back("s_waitcnt lgkmcnt(0)");
        rfl(XP, R0); rfl(XP + 1, R0 + 1);
        rfl(DTP, R0 + 2); rfl(DTP + 1, R0 + 3);
        rfl(TAB, R0 + 4); rfl(TAB + 1, R0 + 5);
        rfl(KT, R0 + 6); rfl(KT + 1, R0 + 7);
        rfl(INV, R0 + 8); rfl(INV + 1, R0 + 9);
        rfl(STR, R0 + 10); rfl(RUN, R0 + 11);
        body.push_back("v_mbcnt_lo_u32_b32 v" + std::to_string(L15) + ", -1, 0");
        body.push_back("v_mbcnt_hi_u32_b32 v" + std::to_string(L15) + ", -1, v" + std::to_string(L15));
        body.push_back("v_and_b32 v" + std::to_string(L15) + ", 15, v" + std::to_string(L15));
        body.push_back("v_lshlrev_b32 v" + std::to_string(L15) + ", 3, v" + std::to_string(L15));
        body.push_back("v_lshrrev_b32 v" + std::to_string(MO8) + ", 1, %[mo]");
        body.push_back("v_add_u32 v" + std::to_string(LDSA) + ", " + std::to_string(NP * 1024) + ", %[ldsv]");
        body.push_back("s_nop 4");
        body.push_back("s_add_u32 s" + std::to_string(KB) + ", s" + std::to_string(KT) + ", 4096");
        body.push_back("s_addc_u32 s" + std::to_string(KB + 1) + ", s" + std::to_string(KT + 1) + ", 0");
        // bytes of one stage of the record: ndim * ld * 8 = strd * ndim / 2
        body.push_back("s_mul_i32 s" + std::to_string(XSTG) + ", s" + std::to_string(STR) + ", " + std::to_string(ndim));
        body.push_back("s_lshr_b32 s" + std::to_string(XSTG) + ", s" + std::to_string(XSTG) + ", 1");
        body.push_back("s_load_dwordx16 s[" + std::to_string(TABV) + ":" + std::to_string(TABV + 15) + "], " + sreg(TAB) + ", 0");
        issue_x();
        for (int k = 0; k < NR - 1; ++k) issue_ring(k);
        body.push_back(".Lqgs_step%=:");
        // ---- one step
        body.push_back("s_load_dwordx4 s[" + std::to_string(DT0) + ":" + std::to_string(DT0 + 3) + "], " + sreg(DTP) + ", 0");
        body.push_back("s_add_u32 s" + std::to_string(DTP) + ", s" + std::to_string(DTP) + ", 8");
        body.push_back("s_addc_u32 s" + std::to_string(DTP + 1) + ", s" + std::to_string(DTP + 1) + ", 0");
        body.push_back("s_waitcnt lgkmcnt(0)");
        body.push_back("v_mov_b64 " + vreg(G0) + ", " + sreg(DT0));
        body.push_back("v_add_f64 " + vreg(G0) + ", " + sreg(DT0 + 2) + ", -" + vreg(G0));             // dt = t[i + 1] - t[i]
        body.push_back("v_mul_f64 " + vreg(G0) + ", " + sreg(INV) + ", " + vreg(G0));                  // inverse = +-1: exact
        // (a lane read straight behind the fp64 instruction that produces the value returned a stale low word: measured, a relative
        // error of 1e-7 in dt b_s that differed from wavefront to wavefront; two wait states in between)
        body.push_back("s_nop 1");
        body.push_back("v_readfirstlane_b32 s" + std::to_string(DTS) + ", v" + std::to_string(G0));
        body.push_back("v_readfirstlane_b32 s" + std::to_string(DTS + 1) + ", v" + std::to_string(G0 + 1));
        // (canonical state of the vector-memory queue at the top of a step: the stage state of stage 0, then NR - 1 coefficient chunks)
        for (int st = 0; st < S; ++st) {
            const bool last = st == S - 1;
            const int IN = BANK[st & 1], OUT = BANK[(st + 1) & 1];
            vm_wait(xop.back());
            for (int r = 0; r < 2 * ndim; ++r) body.push_back("v_accvgpr_read_b32 v" + std::to_string(X0 + r) + ", a" + std::to_string(r));
            // The coefficient chunks requested ahead for this stage are older than the loads issued next: they are waited for first
            // (they left a stage ago), so that no later wait for a chunk has the new stage-state loads between itself and its chunk --
            // in the last step of the run those loads are not issued at all (the branch), and a wait must mean the same on both paths:
            // every wait below counts operations issued AFTER its target only, all of them coefficient chunks.
            for (int k = 0; k < NR - 1; ++k) vm_wait(ring_op[k % NR]);
            if (last) {                                       // the next step's first stage state, unless this is the last step of the run
                body.push_back("s_cmp_eq_u32 s" + std::to_string(RUN) + ", 1");
                body.push_back("s_cbranch_scc1 .Lqgs_nx%=");
            }
            issue_x();
            if (last) body.push_back(".Lqgs_nx%=:");
            // hb = dt b_s, ha = dt a_{s+1,s} (times inverse) as scalars
            body.push_back("v_mov_b64 " + vreg(G0) + ", " + sreg(DTS));
            body.push_back("v_mul_f64 " + vreg(G0 + 2) + ", " + sreg(TABV + 2 * st) + ", " + vreg(G0));
            body.push_back("s_nop 1");
            body.push_back("v_readfirstlane_b32 s" + std::to_string(HB) + ", v" + std::to_string(G0 + 2));
            body.push_back("v_readfirstlane_b32 s" + std::to_string(HB + 1) + ", v" + std::to_string(G0 + 3));
            if (!last) {
                body.push_back("v_mul_f64 " + vreg(G0 + 2) + ", " + sreg(TABV + 2 * (S + st)) + ", " + vreg(G0));
                body.push_back("s_nop 1");
                body.push_back("v_readfirstlane_b32 s" + std::to_string(HA) + ", v" + std::to_string(G0 + 2));
                body.push_back("v_readfirstlane_b32 s" + std::to_string(HA + 1) + ", v" + std::to_string(G0 + 3));
            }
            int chunk = -1;                                   // chunk of the table being consumed in this stage
            auto need_chunk = [&](int k) {
                while (chunk < k) {
                    ++chunk;
                    issue_ring(chunk + NR - 1);               // into the slot the previous chunk leaves (wraps into the next stage's first chunks)
                    vm_wait(ring_op[chunk % NR]);
                }
            };
            // (issue order inside `need_chunk`: the slot of chunk + NR - 1 is the slot of chunk - 1, whose last use has been issued)
            for (int q = 0; q < NP; ++q) {
                const int i0 = 2 * q + 1, i1 = std::min(ndim, 2 * q + 2);
                const bool two = i1 != i0;
                if (st > 0) body.push_back(std::string(two ? "ds_read_b128 " + vreg4(SETA) : "ds_read_b64 " + vreg(SETA)) + ", v" + std::to_string(LDSA) + " offset:" + std::to_string(q * 1024));
                if (st > 0 && !last) body.push_back(std::string(two ? "ds_read_b128 " + vreg4(SETV) : "ds_read_b64 " + vreg(SETV)) + ", %[ldsv] offset:" + std::to_string(q * 1024));
                body.push_back("v_mov_b64 " + vreg(R0) + ", 0"); valu_wrote(R0);
                if (two) { body.push_back("v_mov_b64 " + vreg(R0 + 2) + ", 0"); valu_wrote(R0 + 2); }
                size_t p0 = 0, p1 = 0;
                auto emit = [&](const TIns &t, int which) {
                    const int R = R0 + 2 * which, G = G0 + 4 * which + 2 * t.tmp;
                    if (t.kind == TIns::Mul) {
                        body.push_back("v_mul_f64 " + vreg(G) + ", " + (t.neg ? "-" : "") + vreg(X0 + 2 * (t.x - 1)) + ", " + vreg(IN + 2 * (t.w - 1)));
                        valu_wrote(G);
                    } else if (t.kind == TIns::Fma) {
                        body.push_back("v_fma_f64 " + vreg(G) + ", " + (t.neg ? "-" : "") + vreg(X0 + 2 * (t.x - 1)) + ", " + vreg(IN + 2 * (t.w - 1)) + ", " + vreg(G));
                        valu_wrote(G);
                    } else {
                        need_chunk(t.entry / CE);
                        const int src = t.src_w ? IN + 2 * (t.w - 1) : G;
                        dpp_reads(R); dpp_reads(src);
                        body.push_back("v_fmac_f64_dpp " + vreg(R) + ", " + (t.cneg ? "-" : "") + vreg(RING + 2 * ((t.entry / CE) % NR)) + ", " +
                                       vreg(src) + " row_newbcast:" + std::to_string(t.entry % CE) + " row_mask:0xf bank_mask:0xf");
                        valu_wrote(R);
                    }
                };
                while (p0 < prog[i0].size() || (two && p1 < prog[i1].size())) {
                    if (p0 < prog[i0].size()) emit(prog[i0][p0++], 0);
                    if (two && p1 < prog[i1].size()) emit(prog[i1][p1++], 1);
                }
                // the rows are done: running sum and next stage vector
                if (st > 0) body.push_back("s_waitcnt lgkmcnt(0)");
                for (int h = 0; h < (two ? 2 : 1); ++h) {
                    const int i = i0 + h, R = R0 + 2 * h, inreg = IN + 2 * (i - 1), outreg = OUT + 2 * (i - 1), accreg = SETA + 2 * h, vr = SETV + 2 * h;
                    if (st == 0) {
                        body.push_back("v_fma_f64 " + vreg(accreg) + ", " + sreg(HB) + ", " + vreg(R) + ", " + vreg(inreg));
                        if (!last) body.push_back("v_fma_f64 " + vreg(outreg) + ", " + sreg(HA) + ", " + vreg(R) + ", " + vreg(inreg));
                    } else if (!last) {
                        body.push_back("v_fma_f64 " + vreg(accreg) + ", " + sreg(HB) + ", " + vreg(R) + ", " + vreg(accreg));
                        body.push_back("v_fma_f64 " + vreg(outreg) + ", " + sreg(HA) + ", " + vreg(R) + ", " + vreg(vr));
                    } else {
                        body.push_back("v_fma_f64 " + vreg(outreg) + ", " + sreg(HB) + ", " + vreg(R) + ", " + vreg(accreg));     // the new state, in the other bank
                    }
                }
                if (st == 0 && S > 2)                          // the step-start vector of the pair: straight from its registers
                    body.push_back(std::string(two ? "ds_write_b128 %[ldsv], " + vreg4(IN + 2 * (i0 - 1)) : "ds_write_b64 %[ldsv], " + vreg(IN + 2 * (i0 - 1))) + " offset:" + std::to_string(q * 1024));
                if (!last)
                    body.push_back(std::string(two ? "ds_write_b128 v" + std::to_string(LDSA) + ", " + vreg4(SETA) : "ds_write_b64 v" + std::to_string(LDSA) + ", " + vreg(SETA)) + " offset:" + std::to_string(q * 1024));
            }
            // chunks of the table the rows did not reach (padding to whole ring turns): keep the ring turning
            need_chunk(NC - 1);
        }
        if (S & 1)                                            // odd stage count: the new state sits in the second bank
            for (int d = 0; d < ndim; ++d) body.push_back("v_mov_b64 " + vreg(BANK[0] + 2 * d) + ", " + vreg(BANK[1] + 2 * d));
        body.push_back("s_sub_u32 s" + std::to_string(RUN) + ", s" + std::to_string(RUN) + ", 1");
        body.push_back("s_cmp_lg_u32 s" + std::to_string(RUN) + ", 0");
        body.push_back("s_cbranch_scc1 .Lqgs_step%=");
        body.push_back("s_waitcnt vmcnt(0) lgkmcnt(0)");

        o << "        " << (pass == 0 ? "if (!adjoint) {" : "else {") << "\n";
        o << "            qargs[3] = (unsigned long long)(kf64*)" << kname << "_kt" << pass << ";\n";
        o << "            asm volatile(\n";
        for (const std::string &ln : body) o << "                \"" << ln << "\\n\"\n";
        o << "                :";
        for (int d = 1; d <= ndim; ++d) o << (d > 1 ? ", " : " ") << "\"+{" << vreg(BANK[0] + 2 * (d - 1)) << "}\"(v" << d << ")";
        o << "\n                : [ldsv] \"v\"(ldsv), [mo] \"v\"(mo), [ldsq] \"v\"(ldsq)\n";
        o << "                :";
        bool first = true;
        for (int r = BANK[1]; r < VEND; ++r) { o << (first ? " " : ", ") << "\"v" << r << "\""; first = false; }
        for (int r = 0; r < 2 * ndim; ++r) o << ", \"a" << r << "\"";
        for (int r = SB; r < SEND; ++r) o << ", \"s" << r << "\"";
        o << ", \"scc\", \"memory\");\n";
        o << "        }\n";
    }
    o << "        ti += run;\n";
    o << "    }\n";
    o << "    if (live) {\n        if (w_out_p) {\n";
    for (int d = 1; d <= ndim; ++d) o << "            w_out_p[" << (d - 1) << " * L + l] = v" << d << ";\n";
    o << "        }\n        if (write_final) {\n"
      << "            f64* p = rec + qgs_rec_index(n_records - 1, n_records, backward) * " << ndim << " * L + l;\n";
    for (int d = 1; d <= ndim; ++d) o << "            p[" << (d - 1) << " * L] = v" << d << ";\n";
    o << "        }\n    }\n";
    o << "    QGS_CLOCK_MARK(2)\n}\n";
    for (int pass = 0; pass < 2; ++pass) emit_ktable(out, kname + "_kt" + std::to_string(pass), tables[pass]);
    out << o.str();
}

}  // namespace detail
}  // namespace qgs
