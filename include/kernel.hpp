// kernel.hpp -- the reference's launcher surface, re-declared for the MI355X build.
//
// Mirrors /root/reference/src/include/kernel.hpp:8-17 symbol for symbol: the same ten C++
// free functions with the same (Itanium-mangled) signatures, so an unmodified
// src/tester.cpp (which includes "kernel.hpp" and calls five of them, tester.cpp:54-63) links
// against libspmv_launchers.so instead of the reference's src/kernels/*.cu.  Unlike the
// reference header this one pulls in no CUDA/HIP headers: the launchers reach the GPU only
// through the C ABI in spmv_hip.h, so a plain host compiler can build the tester.
//
// Semantics kept from the reference:
//   A_host  dense row-major M x N fp32 (read only), X_host M fp32, Y_host N fp32, fully
//   overwritten with  Y[i] = sum_j X[j] * A[j*N+i]  (README.md:29-35, tester.cpp:36-45).
//   Synchronous; callee owns and frees every device allocation; prints
//   "<launch> took <ms> ms" on stdout (kernel.hpp:44); a runtime failure prints
//   "HIP error ..." on stderr and exit(EXIT_FAILURE)s (kernel.hpp:21-28).
// Difference, on purpose: an unknown `version` is an error (the reference runs nothing and
// returns whatever a fresh cudaMalloc held, wsp.cu:187).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include "spmv_hip.h"

// GPU kernel launchers (reference: kernel.hpp:8-17) -> variant that backs each one
void tiling_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host);      // dense, LDS x tile
void cublas_gemv_gpu(int M, int N, float *A, float *X, float *Y);                      // dense, split-M (vendor slot)
void naive_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host);       // dense, thread per output
void csr_naive_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host);   // SPMV_SCALAR
void csr_tiling_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host);  // tiled bitmap-CSR (spmv_tcsr_*)
void wsp_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host, int version);   // 0 SPMV_WAVE, 1 SPMV_WAVE_PIPE
void asp_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host, int version);   // 0,1 SPMV_VECTOR; 2 dense + x==0 skip
void awsp_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host, int version);  // 0,1 SPMV_ADAPTIVE, 2 SPMV_TILED
void awsp_ref_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host);    // SPMV_SCALAR (reference order)
void wsp_sm_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host);      // SPMV_TILED

// error convention of the path (reference: CUDA_CHECK, kernel.hpp:21-28)
#define SPMV_CHECK(call)                                                                   \
    {                                                                                      \
        int err__ = (call);                                                                \
        if (err__ != SPMV_OK) {                                                            \
            fprintf(stderr, "HIP error %s:%d: %s\n", __FILE__, __LINE__, spmv_last_error()); \
            exit(EXIT_FAILURE);                                                            \
        }                                                                                  \
    }

// timing convention of the path (reference: TIME_KERNEL, kernel.hpp:31-48): `run_call` is an
// spmv_*_run_host(..., &ms) expression; the event-timed milliseconds it reports are printed
// in the reference's format.
#define TIME_KERNEL(run_call, ms_var)                                        \
    {                                                                        \
        SPMV_CHECK(run_call);                                                \
        std::cout << #run_call << " took " << (ms_var) << " ms" << std::endl; \
    }
