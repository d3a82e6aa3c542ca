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
void wsp_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host, int version);   // 0 WSP bitmap format (SPMV_WAVE if not 32-aligned), 1 SPMV_WAVE_PIPE
void asp_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host, int version);   // 0 SPMV_VECTOR; 1 SPMV_XSKIP; 2 dense + x==0 skip
void awsp_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host, int version);  // 0 AWSP bitmap format (SPMV_ADAPTIVE if not 32-aligned), 1 SPMV_TILED, 2 SPMV_AUTO
void awsp_ref_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host);    // SPMV_SCALAR (reference order)
void wsp_sm_gemv_gpu(int M, int N, float *A_host, float *X_host, float *Y_host);      // AWSPRef bitmap format (SPMV_TILED if not 32-aligned)

// error convention of the path (reference: CUDA_CHECK, kernel.hpp:21-28)
#define SPMV_CHECK(call)                                                                   \
    {                                                                                      \
        int err__ = (call);                                                                \
        if (err__ != SPMV_OK) {                                                            \
            fprintf(stderr, "HIP error %s:%d: %s\n", __FILE__, __LINE__, spmv_last_error()); \
            exit(EXIT_FAILURE);                                                            \
        }                                                                                  \
    }

// timing convention of the path (reference: TIME_KERNEL(kernel_call), kernel.hpp:31-48: ONE argument, the launch
// expression, timed with two events and printed as "<stringified call> took <ms> ms").  Here the launch expression
// is one of the spmv_*_run_host calls, which time their kernel between two HIP events on the launch stream
// themselves and report it through their last argument; by convention that argument is a float named
// `kernel_ms` in the calling scope.  The reference's figure is ONE COLD launch (module load included): the run_host calls
// keep the event time of their first launch too and the line prints it after the warm figure.
#define TIME_KERNEL(kernel_call)                                                     \
    {                                                                                \
        SPMV_CHECK(kernel_call);                                                     \
        std::cout << #kernel_call << " took " << kernel_ms << " ms (first launch, cold, as the reference times it: " \
                  << spmv_last_first_launch_ms() << " ms)" << std::endl;                                             \
    }

// the reference's name for the error macro (kernel.hpp:21-28), for code written against that header
#define CUDA_CHECK(call) SPMV_CHECK(call)
