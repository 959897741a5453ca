// codegen_lds.cpp -- emitters of the LDS-resident kernels of large systems (stepper, f, tangent / adjoint model).
// See codegen.h / codegen_internal.h.
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

// ---- LDS-resident stepper for systems that do not fit the register file (MAOOAM 6x6: ndim 228) ------------
// A workgroup of W wavefronts advances 64 members; the stage state lives in LDS as xs[mode][member] and wave w
// evaluates a contiguous block of rows.  The run-time-indexed generic kernel needs two LDS reads per tensor term
// (LDS-bound: measured 21 % of the fp64 rate).  Here the (j,k) pattern is compile-time knowledge, so the terms of
// a wave are reordered into "phases": a phase loads a small set of modes (<= cap) into registers once and then
// executes every term of the wave whose two factors are both in the set.  On the MAOOAM 6x6 tensor that is
// ~0.17 LDS reads per term instead of 2, and a product x_j*x_k needed by several rows of the wave is computed once
// (1.6 fp64 instructions per term instead of 2).


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
                        const std::vector<std::pair<int, int>> &der, bool tend_kernel, bool dense)
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

}  // namespace detail
}  // namespace qgs
