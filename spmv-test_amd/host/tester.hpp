// tester.hpp -- host-side mirror of the reference's harness class
// (/root/reference/src/include/tester.hpp:11-57): same class name, same public surface
// (SparseSgemvTester(int m, int n), RunTest()), same flow.  Written fresh.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

class SparseSgemvTester {
public:
    SparseSgemvTester(int m, int n);
    ~SparseSgemvTester() = default;

    // random inputs -> CPU result -> every registered launcher -> compare (tester.cpp:15-34)
    void RunTest();

    // additions over the reference: reproducible inputs and a verdict the caller can act on
    void SetSeed(uint64_t seed) { seed_ = seed; seeded_ = true; }
    void SetSparsity(double a_zero_fraction, double x_zero_fraction) { a_zero_ = a_zero_fraction; x_zero_ = x_zero_fraction; }
    int Mismatches() const { return mismatches_; }

private:
    int m_, n_;
    uint64_t seed_ = 0;
    bool seeded_ = false;
    double a_zero_ = 0.5, x_zero_ = 0.5;  // tester.cpp:106,154
    int mismatches_ = 0;

    std::vector<float> A_host, X_host, Y_cpu_host;
    std::vector<std::vector<float>> Y_gpu_hosts;
    std::vector<std::string> names_;

    void GetRandomMatrix();
    void GetRandomVector();
    void SgemvCPU();
    void SgemvGPU();
    void CompareY();
};
