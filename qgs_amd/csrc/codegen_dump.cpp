// codegen_dump -- developer tool: print the specialised kernel source for a tensor given as text
// (lines "T i j k value" for the tendencies tensor and "J i j k value" for the Jacobian tensor).
#include "codegen.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s ndim tensor.txt [nogroup] [waves=N] [stages=4]\n", argv[0]); return 2; }
    int ndim = std::atoi(argv[1]);
    std::vector<qgs::Term> T, J;
    FILE *f = std::fopen(argv[2], "r");
    if (!f) { std::perror("open"); return 1; }
    char kind; int i, j, k; double v;
    while (std::fscanf(f, " %c %d %d %d %la", &kind, &i, &j, &k, &v) == 5) (kind == 'T' ? T : J).push_back({i, j, k, v});
    std::fclose(f);
    qgs::CodegenOptions opt;
    std::vector<int> stages = {4};
    for (int a = 3; a < argc; ++a) {
        if (!std::strcmp(argv[a], "nogroup")) opt.group_coeff = false;
        if (!std::strncmp(argv[a], "waves=", 6)) opt.min_waves_per_simd = std::atoi(argv[a] + 6);
        if (!std::strncmp(argv[a], "stages=", 7)) stages = {std::atoi(argv[a] + 7)};
        if (!std::strncmp(argv[a], "split=", 6)) opt.row_split = std::atoi(argv[a] + 6);
        if (!std::strcmp(argv[a], "ktab")) opt.const_table = true;
        if (!std::strcmp(argv[a], "nopark")) opt.tgl_park_lds = false;
        if (!std::strncmp(argv[a], "kgroup=", 7)) opt.ktab_group = std::atoi(argv[a] + 7);
        if (!std::strncmp(argv[a], "tsplit=", 7)) opt.tgl_split = std::atoi(argv[a] + 7);
        if (!std::strncmp(argv[a], "ilv=", 4)) opt.interleave = std::atoi(argv[a] + 4);
    }
    std::fprintf(stderr, "ndim %d nnz %zu jnnz %zu tendency fp64 instr %lld\n", ndim, T.size(), J.size(),
                 (long long)qgs::count_tendency_flops_instr(ndim, T, opt));
    std::fputs(qgs::generate_source(ndim, T, J, stages, opt).c_str(), stdout);
    return 0;
}
