// main_test_pc.cpp -- the reference's commented-out CLI hook `./pigeon <logN> <option> <K>` (option 4: RS x expander, option 1: RS x RS)
// (src/main.cpp:1176: test_PC(1ULL<<atoi(argv[1]), atoi(argv[2]), atoi(argv[3]))) over the
// device-backed host mirror.  Build: see __graft_entry__.build_host().
#include <cstdio>
#include <cstdlib>
#include <string>
#include "hobbit_host.hpp"
int main(int argc, char **argv) {
    if (argc < 4) { printf("usage: %s <logN> <4 | 1> <K>   |   %s elastic <logN> <logB> <opt>\n", argv[0], argv[0]); return 1; }
    init_hash();
    if (std::string(argv[1]) == "elastic") {          // src/main.cpp:1177-1178: BUFFER_SPACE = 1<<argv[2]; test_Elastic_PC(1<<argv[1], argv[3])
        BUFFER_SPACE = 1ULL << atoi(argv[3]);
        test_Elastic_PC(1ULL << atoi(argv[2]), argc > 4 ? atoi(argv[4]) : 1);
    } else
        test_PC(1ULL << atoi(argv[1]), atoi(argv[2]), atoi(argv[3]));
    hobbit_host_shutdown();
    return 0;
}
