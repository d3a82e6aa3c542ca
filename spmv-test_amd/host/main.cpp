// main.cpp -- the tester executable (reference: test/main.cpp:3-7 hard-codes 4096 x 4096).
//   sparse_sgemv [M N]                          the reference flow on a random dense M x N matrix
//   sparse_sgemv --mtx FILE [--variant NAME] [--out Y.txt] [--save FILE.csrbin] [--parse-only]
//   sparse_sgemv --csrbin FILE [--variant NAME] [--out Y.txt]
//        y = A * ones through the C ABI on a matrix from disk (row f-4), checked against a host walk
// $SPMV_SEED makes the random inputs reproducible.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "kernel.hpp"
#include "mtx_io.hpp"
#include "tester.hpp"

static int variant_by_name(const std::string &n)
{
    for (int v = 0; v < SPMV_VARIANT_COUNT; ++v)
        if (n == spmv_variant_name(v)) return v;
    return -1;
}

static int run_file(int argc, char **argv)
{
    std::string mtx, bin, out, save, vname = "tiled";
    bool parse_only = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--mtx") mtx = next();
        else if (a == "--csrbin") bin = next();
        else if (a == "--out") out = next();
        else if (a == "--save") save = next();
        else if (a == "--variant") vname = next();
        else if (a == "--parse-only") parse_only = true;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    HostCsr m;
    std::string err = !mtx.empty() ? read_matrix_market(mtx, m) : read_csr_binary(bin, m);
    if (!err.empty()) { std::fprintf(stderr, "error: %s\n", err.c_str()); return 2; }
    std::printf("matrix %lld x %lld, nnz %lld\n", (long long)m.rows, (long long)m.cols, (long long)m.nnz());
    if (!save.empty() && !(err = write_csr_binary(save, m)).empty()) { std::fprintf(stderr, "error: %s\n", err.c_str()); return 2; }
    if (parse_only) return 0;
    const int variant = variant_by_name(vname);
    if (variant < 0) { std::fprintf(stderr, "unknown variant %s\n", vname.c_str()); return 2; }

    std::vector<float> x((size_t)m.cols, 1.0f), y((size_t)m.rows, NAN), ref((size_t)m.rows, 0.0f);
    for (int64_t r = 0; r < m.rows; ++r) {  // the harness's own CPU check, like SgemvCPU (tester.cpp:36-45)
        float acc = 0.0f;
        for (int32_t k = m.row_ptr[r]; k < m.row_ptr[r + 1]; ++k) acc += x[m.col_idx[k]] * m.vals[k];
        ref[r] = acc;
    }
    spmv_csr_t *A = nullptr;
    SPMV_CHECK(spmv_csr_create_host(m.rows, m.cols, m.nnz(), m.row_ptr.data(), m.col_idx.data(), m.vals.data(), &A));
    float ms = 0.0f;
    SPMV_CHECK(spmv_csr_run_host(A, variant, x.data(), y.data(), &ms));
    std::printf("spmv_csr_run<%s> took %g ms\n", spmv_variant_name(variant), ms);
    SPMV_CHECK(spmv_csr_destroy(A));
    int bad = 0;
    for (int64_t r = 0; r < m.rows; ++r) {
        float mag = 0.0f;
        for (int32_t k = m.row_ptr[r]; k < m.row_ptr[r + 1]; ++k) mag += std::fabs(m.vals[k]);
        if (!(std::fabs(ref[r] - y[r]) <= 1e-5f * mag + 1e-30f)) {
            if (bad < 16) std::fprintf(stderr, "[row %lld] cpu: %g, gpu: %g\n", (long long)r, ref[r], y[r]);
            ++bad;
        }
    }
    if (!out.empty() && !(err = write_vector_text(out, y)).empty()) { std::fprintf(stderr, "error: %s\n", err.c_str()); return 2; }
    std::printf(bad ? "====== %d MISMATCHES ======\n" : "========== OK ===========\n", bad);
    return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && std::strncmp(argv[1], "--", 2) == 0) return run_file(argc, argv);
    int m = 4096, n = 4096;
    if (argc >= 3) { m = std::atoi(argv[1]); n = std::atoi(argv[2]); }
    SparseSgemvTester tester(m, n);
    tester.RunTest();
    return tester.Mismatches() == 0 ? 0 : 1;
}
