// codegen_dump -- developer tool: print the specialised kernel source for a tensor given as text
// (lines "T i j k value" for the tendencies tensor and "J i j k value" for the Jacobian tensor; rank-5 tensors:
// "T5 i j k l m value" / "J5 i j k l m value").  The source is generated from the canonical form of the tensor, as the
// library does (codegen.h canonicalize): it holds no coefficient values; `tables` also prints the decoded coefficient tables.
#include "codegen.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s ndim tensor.txt [waves=N] [stages=4] [all]\n", argv[0]); return 2; }
    int ndim = std::atoi(argv[1]);
    std::vector<qgs::Term> T, J;
    FILE *f = std::fopen(argv[2], "r");
    if (!f) { std::perror("open"); return 1; }
    char kind[8]; int c[5]; double v;
    std::vector<int32_t> coo[2];
    std::vector<double> val[2];
    int rank = 3;
    while (std::fscanf(f, " %7s", kind) == 1) {
        const int r = kind[1] == '5' ? 5 : 3;
        rank = r;
        for (int q = 0; q < r; ++q) if (std::fscanf(f, "%d", &c[q]) != 1) return 1;
        if (std::fscanf(f, "%la", &v) != 1) return 1;
        const int w = kind[0] == 'T' ? 0 : 1;
        coo[w].insert(coo[w].end(), c, c + r);
        val[w].push_back(v);
    }
    std::fclose(f);
    qgs::Derived der;
    std::vector<double> magnitude;
    qgs::reduce_polynomial(ndim, rank, (int64_t)val[0].size(), coo[0].data(), val[0].data(), false, T, der.t);
    qgs::reduce_polynomial(ndim, rank, (int64_t)val[1].size(), coo[1].data(), val[1].data(), true, J, der.j);
    {
        qgs::Canonical ct, cj;                         // what the library generates from
        qgs::canonicalize(T, ct);
        qgs::canonicalize(J, cj);
        T = ct.terms;
        J = cj.terms;
        magnitude = ct.magnitude;
    }
    qgs::CodegenOptions opt;
    std::vector<int> stages = {4};
    bool all = false, tables = false;
    for (int a = 3; a < argc; ++a) {
        if (!std::strcmp(argv[a], "all")) all = true;
        if (!std::strcmp(argv[a], "tables")) tables = true;
        if (!std::strncmp(argv[a], "waves=", 6)) opt.min_waves_per_simd = std::atoi(argv[a] + 6);
        if (!std::strncmp(argv[a], "stages=", 7)) stages = {std::atoi(argv[a] + 7)};
        if (!std::strncmp(argv[a], "split=", 6)) opt.row_split = std::atoi(argv[a] + 6);
        if (!std::strncmp(argv[a], "ilv=", 4)) opt.interleave = std::atoi(argv[a] + 4);
        if (!std::strncmp(argv[a], "ldsorder=", 9)) opt.lds_order = std::atoi(argv[a] + 9);
        if (!std::strncmp(argv[a], "ldscap=", 7)) opt.lds_cap = std::atoi(argv[a] + 7);
        if (!std::strncmp(argv[a], "tglasm=", 7)) opt.tgl_asm = std::atoi(argv[a] + 7) != 0;
        if (!std::strncmp(argv[a], "tglring=", 8)) opt.tgl_asm_ring = std::atoi(argv[a] + 8);
        if (!std::strncmp(argv[a], "asm=", 4)) opt.lds_asm = std::atoi(argv[a] + 4) != 0;
        if (!std::strncmp(argv[a], "ldstglasm=", 10)) opt.lds_tgl_asm = std::atoi(argv[a] + 10) != 0;
        if (!std::strncmp(argv[a], "asmcoef=", 8)) opt.lds_asm_coef = std::atoi(argv[a] + 8);
        if (!std::strncmp(argv[a], "asmmerge=", 9)) opt.lds_asm_merge = std::atoi(argv[a] + 9) != 0;
        if (!std::strncmp(argv[a], "asmkeep=", 8)) opt.lds_asm_keep = std::atoi(argv[a] + 8) != 0;
        if (!std::strncmp(argv[a], "asmprog=", 8)) opt.lds_asm_progressive = std::atoi(argv[a] + 8) != 0;
        if (!std::strncmp(argv[a], "asmring=", 8)) opt.lds_asm_ring = std::atoi(argv[a] + 8);
        if (!std::strncmp(argv[a], "asmmincap=", 10)) opt.lds_asm_mincap = std::atoi(argv[a] + 10);
        if (!std::strncmp(argv[a], "asmwaves=", 9)) opt.lds_asm_waves = std::atoi(argv[a] + 9);
        if (!std::strncmp(argv[a], "asmcap=", 7)) opt.lds_asm_cap = std::atoi(argv[a] + 7);
        if (!std::strncmp(argv[a], "asmpp=", 6)) opt.lds_asm_pingpong = std::atoi(argv[a] + 6) != 0;
        if (!std::strncmp(argv[a], "asmlanes=", 9)) opt.lds_asm_lanes = std::atoi(argv[a] + 9);
        if (!std::strncmp(argv[a], "asmchunk=", 9)) opt.lds_asm_chunk = std::atoi(argv[a] + 9);
        if (!std::strncmp(argv[a], "asmvfree=", 9)) opt.lds_asm_vfree = std::atoi(argv[a] + 9);
        if (!std::strncmp(argv[a], "asmsfree=", 9)) opt.lds_asm_sfree = std::atoi(argv[a] + 9);
        if (!std::strncmp(argv[a], "ldswaves=", 9)) opt.lds_waves = std::atoi(argv[a] + 9);
        if (!std::strncmp(argv[a], "ldsyload=", 9)) opt.lds_yload_ahead = std::atoi(argv[a] + 9);
    }
    if (rank == 5) opt.row_split = 1;
    if (!qgs::tgl_asm_supported(ndim, der.j.empty(), opt)) opt.tgl_asm = false;
    if (!der.t.empty()) opt.lds_asm = false;           // (as the library: the hand-scheduled LDS stepper takes rank-3 tensors only)
    if (!qgs::lds_tgl_asm_supported(ndim, der.j.empty(), opt)) opt.lds_tgl_asm = false;
    std::fprintf(stderr, "ndim %d rank %d terms %zu jac terms %zu derived %zu / %zu tendency fp64 instr %lld\n", ndim, rank, T.size(),
                 J.size(), der.t.size(), der.j.size(), (long long)qgs::count_tendency_flops_instr(ndim, T, opt) + (long long)der.t.size());
    if (ndim <= 64) std::fputs(qgs::generate_source(ndim, T, J, stages, opt, der).c_str(), stdout);
    if (all) {
        // every other emitter too (the sanitizer job of tests/test_codegen_sanitizers.py): general-tableau, LDS-resident and QR kernels
        using K = qgs::Kernel;
        if (ndim <= 64)
            for (K k : {K::RkDense, K::TglDense}) std::fputs(qgs::generate_kernel(ndim, T, J, k, stages[0], opt, der).source.c_str(), stdout);
        for (K k : {K::RkLds, K::TendLds, K::RkLdsDense, K::TglLds, K::AdjLds}) std::fputs(qgs::generate_kernel(ndim, T, J, k, 0, opt, der).source.c_str(), stdout);
        std::fputs(qgs::generate_qr_kernel(ndim < 64 ? ndim : 64, ndim < 5 ? ndim : 5, qgs::qr_plan(ndim < 64 ? ndim : 64, ndim < 5 ? ndim : 5)).source.c_str(), stdout);
    }
    if (tables) {
        // the coefficient tables of the fused stepper, decoded back to this tensor's values
        qgs::Canonical canon;
        canon.magnitude = magnitude;
        const qgs::GeneratedKernel g = qgs::generate_kernel(ndim, T, J, ndim <= 64 ? qgs::Kernel::Rk : qgs::Kernel::RkLds, ndim <= 64 ? stages[0] : 0, opt, der);
        if (ndim > 64) std::fputs(g.source.c_str(), stdout);      // (ndim <= 64: printed above with the other kernels)
        for (const qgs::CoefTable &t : g.tables) {
            std::vector<double> v;
            canon.decode(t.values, v);
            std::printf("// table %s: %zu entries\n", t.symbol.c_str(), v.size());
            for (size_t n = 0; n < v.size(); ++n) std::printf("//   [%zu] %a\n", n, v[n]);
        }
    }
    return 0;
}
