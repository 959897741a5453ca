// Developer tool (build host): the generated source of the shape-specialised batched QR for one plan, see qr_phases.cpp.
//   qr_dump rows cols [members slots chains reload]      (0 / omitted: the default choice)
#include <cstdio>
#include <cstdlib>
#include "codegen.h"
int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s rows cols [members slots chains reload]\n", argv[0]); return 2; }
    const int R = std::atoi(argv[1]), C = std::atoi(argv[2]);
    qgs::QrPlan p = qgs::qr_plan(R, C, argc > 3 ? std::atoi(argv[3]) : 0, argc > 4 ? std::atoi(argv[4]) : 0);
    if (argc > 5 && std::atoi(argv[5]) > 0) p.chains = std::atoi(argv[5]);
    if (argc > 6) p.reload = std::atoi(argv[6]) != 0;
    std::fprintf(stderr, "%s members %d waves %d\n", qgs::qr_plan_signature(p).c_str(), p.members, p.waves);
    std::fputs(qgs::generate_qr_kernel(R, C, p).source.c_str(), stdout);
    return 0;
}
