// TEST INFRASTRUCTURE (oracle/_ref).  Drives the reference's OWN parameter-file reader — Core/src/Utils/GlobalStateParams.h,
// parameterFile.h, stringUtil*.h, compiled from where they lie under /root/reference (only the standard library is needed) —
// with the call sequence of GUI/src/HRBF_fusion.cpp:35-38 and prints every member it read.  tests/test_config.py compares
// include/hrbf_io.h (C++) and hrbffusion3d_amd/config.py (Python) with this output.
#include <cstdio>
#include <iomanip>
#include <iostream>
#include "GlobalStateParams.h"

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: ref_params GlobalStateParam.txt\n"); return 2; }
    ParameterFile pf(argv[1]);
    GlobalStateParam::getInstance().readMembers(pf);
    const GlobalStateParam &g = GlobalStateParam::get();
    std::cout << "----\n" << std::setprecision(9);
#define X(type, name) std::cout << #name << "\t" << #type << "\t[" << g.name << "]\n";
    X_GLOBAL_PARAM_FIELDS
#undef X
    return 0;
}
