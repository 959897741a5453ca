// codegen_tangent.cpp -- emitter of the register-resident tangent / adjoint kernels.  See codegen.h / codegen_internal.h.
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
                     const std::vector<std::pair<int, int>> &der, int share_x, bool dense, bool pair_x)
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


}  // namespace detail
}  // namespace qgs
