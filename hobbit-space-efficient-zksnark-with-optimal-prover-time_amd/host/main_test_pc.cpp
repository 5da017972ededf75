// main_test_pc.cpp -- the reference's commented-out CLI hook `./pigeon <logN> 4 <K>`
// (src/main.cpp:1176: test_PC(1ULL<<atoi(argv[1]), atoi(argv[2]), atoi(argv[3]))) over the
// device-backed host mirror.  Build: see __graft_entry__.build_host().
#include <cstdio>
#include <cstdlib>
#include "hobbit_host.hpp"
int main(int argc, char **argv) {
    if (argc < 4) { printf("usage: %s <logN> 4 <K>\n", argv[0]); return 1; }
    init_hash();
    test_PC(1ULL << atoi(argv[1]), atoi(argv[2]), atoi(argv[3]));
    hobbit_host_shutdown();
    return 0;
}
