// mtx_io.hpp -- on-disk formats around the path (the reference has none:
// it only ever multiplies a random dense matrix, src/tester.cpp:103-121).
//   Matrix Market coordinate files ("%%MatrixMarket matrix coordinate real|integer|pattern
//   general|symmetric|skew-symmetric") -> CSR with the conventions of include/spmv_hip.h
//   (row = output index, columns ascending inside a row, duplicates summed);
//   a raw binary CSR container (.csrbin) for fast reloads and for writing results back.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct HostCsr {
    int64_t rows = 0, cols = 0;
    std::vector<int32_t> row_ptr;  // rows + 1
    std::vector<int32_t> col_idx;
    std::vector<float> vals;
    int64_t nnz() const { return (int64_t)vals.size(); }
};

// Both return an empty string on success, else a message.
std::string read_matrix_market(const std::string &path, HostCsr &out);
std::string write_csr_binary(const std::string &path, const HostCsr &m);
std::string read_csr_binary(const std::string &path, HostCsr &out);
std::string write_vector_text(const std::string &path, const std::vector<float> &y);
