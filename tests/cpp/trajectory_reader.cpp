// prints what hrbf_io.h's loadTrajectoryFile reads: "<count>" then one line per pose "<stamp> <tx> <ty> <tz>"
#include <cstdio>
#include "hrbf_io.h"
int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    std::vector<int64_t> stamps;
    const std::vector<hrbf_mi355::PoseCM> p = hrbf_mi355::loadTrajectoryFile(argv[1], argv[2], &stamps);
    printf("%zu\n", p.size());
    for (size_t i = 0; i < p.size(); ++i)
        printf("%lld %.9g %.9g %.9g\n", i < stamps.size() ? (long long)stamps[i] : -1LL, p[i].m[12], p[i].m[13], p[i].m[14]);
    return 0;
}
