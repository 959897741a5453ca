// Developer tool: print the generated source of one specialised kernel for a tensor given as text
// (first line: ndim nnz; then "i j k value" per line, value as hex float or decimal).
//   g++ -O1 -std=c++17 -I qgs_amd/csrc tools/gen_kernel.cpp qgs_amd/csrc/codegen.cpp qgs_amd/csrc/codegen_steppers.cpp qgs_amd/csrc/codegen_tangent.cpp qgs_amd/csrc/codegen_lds.cpp qgs_amd/csrc/codegen_qr.cpp -o /tmp/gen_kernel
//   /tmp/gen_kernel tensor.txt rklds [stages] > k.hip && hipcc --offload-arch=gfx950 -c -Rpass-analysis=kernel-resource-usage k.hip
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "codegen.h"

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s tensor.txt tend|rk|rksplit|rkstages|rkstagesp|rkrec|tgl|tglp|rklds|tgllds|adjlds [S]\n", argv[0]); return 2; }
    FILE *f = std::fopen(argv[1], "r");
    if (!f) { std::perror(argv[1]); return 1; }
    int ndim; long nnz;
    if (std::fscanf(f, "%d %ld", &ndim, &nnz) != 2) return 1;
    std::vector<qgs::Term> T;
    for (long e = 0; e < nnz; ++e) {
        int i, j, k; char buf[64];
        if (std::fscanf(f, "%d %d %d %63s", &i, &j, &k, buf) != 4) return 1;
        T.push_back({i, j, k, std::strtod(buf, nullptr)});
    }
    std::fclose(f);
    qgs::CodegenOptions opt;
    if (const char *e = std::getenv("QGS_HIP_LDS_WAVES")) opt.lds_waves = std::atoi(e);
    if (const char *e = std::getenv("QGS_HIP_LDS_CAP")) opt.lds_cap = std::atoi(e);
    if (const char *e = std::getenv("QGS_HIP_LDS_YLOAD")) opt.lds_yload_ahead = std::atoi(e);
    if (const char *e = std::getenv("QGS_HIP_LDS_GROUP")) opt.lds_group = (*e == '1');
    const int S = argc > 3 ? std::atoi(argv[3]) : 4;
    qgs::Kernel k = qgs::Kernel::Tend;
    if (!std::strcmp(argv[2], "rk")) k = qgs::Kernel::Rk;
    else if (!std::strcmp(argv[2], "rksplit")) k = qgs::Kernel::RkSplit;
    else if (!std::strcmp(argv[2], "rkstages")) k = qgs::Kernel::RkStages;
    else if (!std::strcmp(argv[2], "rkrec")) k = qgs::Kernel::RkRec;
    else if (!std::strcmp(argv[2], "rkstagesp")) k = qgs::Kernel::RkStagesPair;
    else if (!std::strcmp(argv[2], "tglp")) k = qgs::Kernel::TglPair;
    else if (!std::strcmp(argv[2], "tgl")) k = qgs::Kernel::Tgl;
    else if (!std::strcmp(argv[2], "rklds")) k = qgs::Kernel::RkLds;
    else if (!std::strcmp(argv[2], "tgllds")) k = qgs::Kernel::TglLds;
    else if (!std::strcmp(argv[2], "adjlds")) k = qgs::Kernel::AdjLds;
    std::vector<qgs::Term> J;                       // Jacobian tensor = T + T.swapaxes(1, 2) (qgtensor.py:700-722)
    if (k == qgs::Kernel::TglLds || k == qgs::Kernel::AdjLds || k == qgs::Kernel::Tgl || k == qgs::Kernel::TglPair)
        for (const qgs::Term &t : T) { J.push_back(t); J.push_back({t.i, t.k, t.j, t.v}); }
    std::cout << qgs::generate_kernel(ndim, T, J, k, S, opt).source;
    return 0;
}
