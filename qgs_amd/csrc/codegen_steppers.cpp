// codegen_steppers.cpp -- emitters of the register-resident kernels: f, Df, the fused RK steppers (sub-diagonal and general
// tableau) and the row-split stepper.  See codegen.h / codegen_internal.h.
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
                    const CodegenOptions &opt, const std::vector<std::pair<int, int>> &der, bool spread_rec,
                    bool pair_stages)
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

}  // namespace detail
}  // namespace qgs
