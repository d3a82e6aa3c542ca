// main.cpp -- the tester executable (reference: test/main.cpp:3-7 hard-codes 4096 x 4096).
// Usage: sparse_sgemv [M N]   (default 4096 4096; $SPMV_SEED makes the inputs reproducible)
#include <cstdlib>
#include "tester.hpp"

int main(int argc, char **argv)
{
    int m = 4096, n = 4096;
    if (argc >= 3) { m = std::atoi(argv[1]); n = std::atoi(argv[2]); }
    SparseSgemvTester tester(m, n);
    tester.RunTest();
    return tester.Mismatches() == 0 ? 0 : 1;
}
