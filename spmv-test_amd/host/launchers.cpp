// launchers.cpp -- the ten *_gemv_gpu launchers of include/kernel.hpp on top of the C ABI.
//
// Each follows the canonical shape of a reference launcher (e.g.
// /root/reference/src/kernels/csr_naive.cu:26-76): build the sparse format from the dense
// host matrix, move data to the device, run ONE timed kernel, copy Y back, release
// everything.  What changed underneath: the format is always CSR and is built on the device
// (spmv_csr_from_dense_host), and the kernel is one of the gfx950 variants of spmv_hip.h.
#include "kernel.hpp"

namespace {

void run_csr(int M, int N, float *A_host, float *X_host, float *Y_host, int variant)
{
    spmv_csr_t *csr = nullptr;
    SPMV_CHECK(spmv_csr_from_dense_host(M, N, A_host, nullptr, &csr));
    float kernel_ms = 0.0f;
    // prints "<stringified call> took <ms> ms" like the reference's macro (kernel.hpp:44)
    std::cout << "[" << spmv_variant_name(variant) << ", rows=" << N << ", cols=" << M << "] ";
    TIME_KERNEL(spmv_csr_run_host(csr, variant, X_host, Y_host, &kernel_ms));
    SPMV_CHECK(spmv_csr_destroy(csr));
}

void run_dense(int M, int N, float *A_host, float *X_host, float *Y_host, int mode)
{
    float kernel_ms = 0.0f;
    std::cout << "[dense mode " << mode << ", M=" << M << ", N=" << N << "] ";
    TIME_KERNEL(spmv_dense_gemv_host(M, N, A_host, X_host, Y_host, mode, &kernel_ms));
}

// the reference's csr_tiling launcher multiplies its tiled bitmap-CSR (csr_tiling.cu:116-166); the
// same format exists here for 32-aligned sizes (the only ones the reference supports); other sizes
// go through CSR with LDS x tiles
void run_tcsr(int M, int N, float *A_host, float *X_host, float *Y_host)
{
    if (M % 32 || N % 32) { run_csr(M, N, A_host, X_host, Y_host, SPMV_TILED); return; }
    spmv_tcsr_t *t = nullptr;
    SPMV_CHECK(spmv_tcsr_from_dense_host(M, N, A_host, nullptr, &t));
    float kernel_ms = 0.0f;
    std::cout << "[tcsr, rows=" << N << ", cols=" << M << "] ";
    TIME_KERNEL(spmv_tcsr_run_host(t, X_host, Y_host, &kernel_ms));
    SPMV_CHECK(spmv_tcsr_destroy(t));
}

// the reference's wsp / awsp / wsp_sm launchers multiply their own bitmap formats (wsp.cu:140-202, awsp.cu:319-388,
// wsp_sm.cu:213-265 on the AWSPRef arrays); the same formats exist here for 32-aligned sizes, other sizes go
// through the CSR variant named as the fallback
void run_bitmap(int M, int N, float *A_host, float *X_host, float *Y_host, int format, int fallback_variant)
{
    if (M % 32 || N % 32) { run_csr(M, N, A_host, X_host, Y_host, fallback_variant); return; }
    spmv_bitmap_t *b = nullptr;
    SPMV_CHECK(spmv_bitmap_from_dense_host(format, M, N, A_host, nullptr, &b));
    float kernel_ms = 0.0f;
    std::cout << "[bitmap format " << format << ", rows=" << N << ", cols=" << M << "] ";
    TIME_KERNEL(spmv_bitmap_run_host(b, X_host, Y_host, &kernel_ms));
    SPMV_CHECK(spmv_bitmap_destroy(b));
}

[[noreturn]] void bad_version(const char *name, int version)
{
    fprintf(stderr, "HIP error %s: unknown version %d\n", name, version);
    exit(EXIT_FAILURE);
}

}  // namespace

void naive_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host) { run_dense(M, N, A_host, X_host, Y_host, 0); }
void tiling_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host) { run_dense(M, N, A_host, X_host, Y_host, 1); }
void cublas_gemv_gpu(int M, int N, float *A, float *X, float *Y) { run_dense(M, N, A, X, Y, 2); }

void csr_naive_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host) { run_csr(M, N, A_host, X_host, Y_host, SPMV_SCALAR); }
void csr_tiling_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host) { run_tcsr(M, N, A_host, X_host, Y_host); }
void wsp_sm_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host) { run_bitmap(M, N, A_host, X_host, Y_host, SPMV_FMT_AWSP_REF, SPMV_TILED); }
void awsp_ref_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host) { run_csr(M, N, A_host, X_host, Y_host, SPMV_SCALAR); }

void wsp_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host, int version)
{
    switch (version) {
        case 0: run_bitmap(M, N, A_host, X_host, Y_host, SPMV_FMT_WSP, SPMV_WAVE); break;   // the WSP format itself
        case 1: run_csr(M, N, A_host, X_host, Y_host, SPMV_WAVE_PIPE); break;                // wavefront-per-64-rows on CSR
        default: bad_version("wsp_gemv_gpu", version);
    }
}

// the reference's ASP kernels keep A dense and skip the rows whose x is zero (asp.cu:20-26): version 0 runs the
// lanes-per-row CSR kernel, version 1 the input-major sweep that skips the segments of zero inputs (SPMV_XSKIP), version
// 2 -- the one the tester registers (tester.cpp:58) -- the dense split-M kernel with that activation-sparsity skip
void asp_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host, int version)
{
    if (version < 0 || version > 2) bad_version("asp_gemv_gpu", version);
    if (version == 2) run_dense(M, N, A_host, X_host, Y_host, 3);
    else if (version == 1) run_csr(M, N, A_host, X_host, Y_host, SPMV_XSKIP);   // the same skip on the compressed matrix
    else run_csr(M, N, A_host, X_host, Y_host, SPMV_VECTOR);
}

void awsp_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host, int version)
{
    // the adaptive slot adapts: version 2 (the reference's most developed kernel, awsp.cu:186-317) lets the library
    // choose between the LDS-tiled kernel and the panel sweep from the matrix itself (SPMV_AUTO)
    switch (version) {
        case 0: run_bitmap(M, N, A_host, X_host, Y_host, SPMV_FMT_AWSP, SPMV_ADAPTIVE); break;   // the AWSP format itself
        case 1: run_csr(M, N, A_host, X_host, Y_host, SPMV_TILED); break;
        case 2: run_csr(M, N, A_host, X_host, Y_host, SPMV_AUTO); break;
        default: bad_version("awsp_gemv_gpu", version);
    }
}
