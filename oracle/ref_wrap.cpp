// ref_wrap.cpp -- extern "C" access to the REFERENCE's own format classes: CSRMatrix
// (/root/reference/src/matrix_csr.cpp:5-23, src/include/matrix_csr.hpp:4-25), TCSRMatrix
// (src/tcsr.cpp:5-38, src/include/tcsr.hpp:4-23), WSPMatrix (src/wsp.cpp:3-40), AWSPMatrix (src/awsp.cpp:3-49),
// AWSPRefMatrix (src/awsp_ref.cpp:4-58) and ASPMatrix (src/asp.cpp:3-14),
// compiled unmodified from where it lies by oracle/Makefile into
// oracle/_ref/libref_formats.so.  TEST INFRASTRUCTURE ONLY: used to pin
// oracle/spmv_oracle.c and to generate tests/golden/*.npz in this container.
// /root/reference does not exist on the GPU box; nothing there needs this file.
#include <cstdint>
#include <cstring>
#include "matrix_csr.hpp"
#include "tcsr.hpp"
#include "wsp.hpp"
#include "awsp.hpp"
#include "awsp_ref.hpp"
#include "asp.hpp"

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

// ---- the reference's bitmap formats of the wsp / awsp / awsp_ref / asp launchers -------------------------------
// One generic shape: build -> opaque handle + sizes + the public statistics the class exposes (stats[4]):
//   kind 0 WSPMatrix      stats = {nz_max_m, nz_max_n, 0, 0}                    (wsp.hpp:14)
//   kind 1 AWSPMatrix     stats = {nz_bk_max_, 0, 0, 0}                          (awsp.hpp:13)
//   kind 2 AWSPRefMatrix  stats = warp_nz_offset_[0..3] via GetWarpNZOffset()    (awsp_ref.hpp:13)
//   kind 3 ASPMatrix      no bitmaps, no statistics
struct RefFmt {
    int kind;
    WSPMatrix *wsp = nullptr;
    AWSPMatrix *awsp = nullptr;
    AWSPRefMatrix *ref = nullptr;
    ASPMatrix *asp = nullptr;
};

void *ref_fmt_build(int kind, int M, int N, float *A, int *n_bitmaps, int *n_vals, int *stats)
{
    RefFmt *f = new RefFmt{kind};
    stats[0] = stats[1] = stats[2] = stats[3] = 0;
    switch (kind) {
        case 0:
            f->wsp = new WSPMatrix(M, N, A);
            *n_bitmaps = f->wsp->BitmapsSize(); *n_vals = f->wsp->ValuesSize();
            stats[0] = f->wsp->nz_max_m; stats[1] = f->wsp->nz_max_n;
            break;
        case 1:
            f->awsp = new AWSPMatrix(M, N, A);
            *n_bitmaps = f->awsp->BitmapsSize(); *n_vals = f->awsp->ValuesSize();
            stats[0] = f->awsp->nz_bk_max_;
            break;
        case 2:
            f->ref = new AWSPRefMatrix(M, N, A);
            *n_bitmaps = f->ref->BitmapsSize(); *n_vals = f->ref->ValuesSize();
            std::memcpy(stats, f->ref->GetWarpNZOffset(), 4 * sizeof(int));
            break;
        default:
            f->asp = new ASPMatrix(M, N, A);
            *n_bitmaps = 0; *n_vals = f->asp->ValuesSize();
            break;
    }
    return f;
}

void ref_fmt_copy(void *h, uint32_t *bitmaps, float *vals)
{
    RefFmt *f = static_cast<RefFmt *>(h);
    switch (f->kind) {
        case 0:
            std::memcpy(bitmaps, f->wsp->GetBitmaps(), sizeof(uint32_t) * (size_t)f->wsp->BitmapsSize());
            std::memcpy(vals, f->wsp->GetValues(), sizeof(float) * (size_t)f->wsp->ValuesSize());
            break;
        case 1:
            std::memcpy(bitmaps, f->awsp->GetBitmaps(), sizeof(uint32_t) * (size_t)f->awsp->BitmapsSize());
            std::memcpy(vals, f->awsp->GetValues(), sizeof(float) * (size_t)f->awsp->ValuesSize());
            break;
        case 2:
            std::memcpy(bitmaps, f->ref->GetBitmaps(), sizeof(uint32_t) * (size_t)f->ref->BitmapsSize());
            std::memcpy(vals, f->ref->GetValues(), sizeof(float) * (size_t)f->ref->ValuesSize());
            break;
        default:
            std::memcpy(vals, f->asp->GetValues(), sizeof(float) * (size_t)f->asp->ValuesSize());
            break;
    }
}

void ref_fmt_free(void *h)
{
    RefFmt *f = static_cast<RefFmt *>(h);
    delete f->wsp; delete f->awsp; delete f->ref; delete f->asp;
    delete f;
}

}  // extern "C"
