// ref_wrap.cpp -- extern "C" access to the REFERENCE's own CSRMatrix class
// (/root/reference/src/matrix_csr.cpp:5-23, src/include/matrix_csr.hpp:4-25),
// compiled unmodified from where it lies by oracle/Makefile into
// oracle/_ref/libref_formats.so.  TEST INFRASTRUCTURE ONLY: used to pin
// oracle/spmv_oracle.c and to generate tests/golden/*.npz in this container.
// /root/reference does not exist on the GPU box; nothing there needs this file.
#include <cstdint>
#include <cstring>
#include "matrix_csr.hpp"

extern "C" {

// Build the reference CSR; returns an opaque handle and the three array sizes
// exactly as the reference's *Size() getters report them (row_ptrs has N
// entries, no trailing sentinel -- matrix_csr.cpp:10-22).
void *ref_csr_build(int M, int N, float *A, int *n_row_ptrs, int *n_col_idxs, int *n_vals)
{
    CSRMatrix *c = new CSRMatrix(M, N, A);
    *n_row_ptrs = c->RowPtrsSize();
    *n_col_idxs = c->ColIdxsSize();
    *n_vals = c->ValuesSize();
    return c;
}

void ref_csr_copy(void *h, int *row_ptrs, int *col_idxs, float *vals)
{
    CSRMatrix *c = static_cast<CSRMatrix *>(h);
    std::memcpy(row_ptrs, c->GetRowPtrs(), sizeof(int) * (size_t)c->RowPtrsSize());
    std::memcpy(col_idxs, c->GetColIdxs(), sizeof(int) * (size_t)c->ColIdxsSize());
    std::memcpy(vals, c->GetValues(), sizeof(float) * (size_t)c->ValuesSize());
}

void ref_csr_free(void *h) { delete static_cast<CSRMatrix *>(h); }

}  // extern "C"
