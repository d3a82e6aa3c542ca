// ref_wrap.cpp -- extern "C" access to the REFERENCE's own format classes: CSRMatrix
// (/root/reference/src/matrix_csr.cpp:5-23, src/include/matrix_csr.hpp:4-25) and TCSRMatrix
// (src/tcsr.cpp:5-38, src/include/tcsr.hpp:4-23),
// compiled unmodified from where it lies by oracle/Makefile into
// oracle/_ref/libref_formats.so.  TEST INFRASTRUCTURE ONLY: used to pin
// oracle/spmv_oracle.c and to generate tests/golden/*.npz in this container.
// /root/reference does not exist on the GPU box; nothing there needs this file.
#include <cstdint>
#include <cstring>
#include "matrix_csr.hpp"
#include "tcsr.hpp"

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

// ---- the reference's tiled bitmap-CSR (/root/reference/src/tcsr.cpp:5-38, tcsr.hpp:4-23) ----
// blk_idx: exclusive prefix of nonzeros per 32x32 block (+1 sentinel), blocks ordered output
// strip outer / input block inner; bitmaps: word = output column inside the block, bit = input row.
void *ref_tcsr_build(int M, int N, float *A, int *n_blk_idx, int *n_bitmaps, int *n_vals)
{
    TCSRMatrix *t = new TCSRMatrix(M, N, A);
    *n_blk_idx = t->BlkIdxSize();
    *n_bitmaps = t->BitmapsSize();
    *n_vals = t->ValuesSize();
    return t;
}

void ref_tcsr_copy(void *h, int *blk_idx, uint32_t *bitmaps, float *vals)
{
    TCSRMatrix *t = static_cast<TCSRMatrix *>(h);
    std::memcpy(blk_idx, t->GetBlkIdx(), sizeof(int) * (size_t)t->BlkIdxSize());
    std::memcpy(bitmaps, t->GetBitmaps(), sizeof(uint32_t) * (size_t)t->BitmapsSize());
    std::memcpy(vals, t->GetValues(), sizeof(float) * (size_t)t->ValuesSize());
}

void ref_tcsr_free(void *h) { delete static_cast<TCSRMatrix *>(h); }

}  // extern "C"
