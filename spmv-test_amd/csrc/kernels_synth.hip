// kernels_synth.hip -- device side of the counter-based synthetic CSR generator.
//
// Not reference behaviour: the reference only has random_device-seeded dense inputs
// (/root/reference/src/tester.cpp:103-121,151-167) that cannot reach the BASELINE sizes
// (a dense 1Mi x 1Mi matrix is 4 TiB).  This generator produces "synthetic CSR of stated
// (rows, cols, nnz)" directly on the device.  The specification is in DESIGN.md
// ("Synthetic workloads"); oracle/spmv_oracle.c holds an independent host statement of the
// same specification, and tests compare the two bit for bit.
//
// Element k of global row r (row length L, window [w0, w0+W)):
//     h    = hash(seed, r, k)
//     col  = w0 + floor(k*W/L) + (h >> 32) mod (floor((k+1)*W/L) - floor(k*W/L))
//     val  = (2*(h & 0xFFFFFF) + 1 - 2^24) * 2^-24          (odd/2^24: exact fp32, never 0)
// so columns ascend strictly inside a row, as CSRMatrix produces them (matrix_csr.cpp:12-20).
#include "spmv_internal.hpp"

namespace spmv {

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

__device__ __forceinline__ uint64_t hash3(uint64_t seed, uint64_t a, uint64_t b)
{
    uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ull * (a + 1));
    return mix64(z + 0xD1B54A32D192ED03ull * (b + 1));
}

__device__ __forceinline__ float unit_from_bits(uint64_t h)
{
    int32_t b = (int32_t)(h & 0xFFFFFFu);
    return (float)(2 * b + 1 - (1 << 24)) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(kBlock) void k_synth_fill(uint64_t seed, int64_t row0, int64_t n_local,
                                                       int64_t rows, int64_t cols, int64_t band,
                                                       const int32_t *__restrict__ row_ptr,
                                                       int32_t *__restrict__ col_idx,
                                                       float *__restrict__ vals, int64_t nnz_local)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nnz_local) return;
    // local row holding element i: last r with row_ptr[r] <= i
    int64_t lo = 0, hi = n_local;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)row_ptr[mid + 1] <= i) lo = mid + 1; else hi = mid;
    }
    const int64_t b = row_ptr[lo];
    const uint64_t len = (uint64_t)(row_ptr[lo + 1] - b);
    const uint64_t k = (uint64_t)(i - b);
    const int64_t row = row0 + lo;

    int64_t w0 = 0, W = cols;
    if (band > 0) {
        int64_t w = band > 8 * (int64_t)len ? band : 8 * (int64_t)len;
        if (w > cols) w = cols;
        int64_t centre = (int64_t)(((uint64_t)row * (uint64_t)cols) / (uint64_t)rows);
        int64_t s = centre - w / 2;
        if (s < 0) s = 0;
        if (s > cols - w) s = cols - w;
        w0 = s;
        W = w;
    }
    const uint64_t h = hash3(seed, (uint64_t)row, k);
    const uint64_t slo = (k * (uint64_t)W) / len;
    const uint64_t shi = ((k + 1) * (uint64_t)W) / len;
    uint64_t span = shi - slo;
    if (span < 1) span = 1;
    col_idx[i] = (int32_t)(w0 + (int64_t)slo + (int64_t)((h >> 32) % span));
    vals[i] = unit_from_bits(h);
}

__global__ __launch_bounds__(kBlock) void k_synth_x(uint64_t seed, int64_t j0, int64_t n, float *__restrict__ x)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    x[i] = unit_from_bits(hash3(seed ^ 0x5851F42D4C957F2Dull, (uint64_t)(j0 + i), 0));
}

int synth_fill(uint64_t seed, int64_t row0, int64_t n_local, int64_t rows, int64_t cols, int64_t band,
               const int32_t *d_row_ptr, int32_t *d_col_idx, float *d_vals, hipStream_t s)
{
    if (n_local == 0) return SPMV_OK;
    int32_t nnz32 = 0;
    SPMV_HIP_TRY(hipMemcpyAsync(&nnz32, d_row_ptr + n_local, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));
    const int64_t nnz = nnz32;
    if (nnz == 0) return SPMV_OK;
    const int64_t blocks = (nnz + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_synth_fill, dim3((unsigned)blocks), dim3(kBlock), 0, s, seed, row0, n_local, rows,
                       cols, band, d_row_ptr, d_col_idx, d_vals, nnz);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_synth_fill", __FILE__, __LINE__);
    return SPMV_OK;
}

int synth_x(uint64_t seed, int64_t j0, int64_t n, float *d_x, hipStream_t s)
{
    if (n == 0) return SPMV_OK;
    const int64_t blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_synth_x, dim3((unsigned)blocks), dim3(kBlock), 0, s, seed, j0, n, d_x);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_synth_x", __FILE__, __LINE__);
    return SPMV_OK;
}

}  // namespace spmv
