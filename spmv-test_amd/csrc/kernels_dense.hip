// kernels_dense.hip -- the dense side of the boundary.
//
//   dense_to_csr   device-side replacement of CSRMatrix::CSRMatrix
//                  (/root/reference/src/matrix_csr.cpp:5-23): dense row-major A[M][N] -> CSR of
//                  A^T; an element is kept iff (v != 0.0f) (-0.0f dropped, NaN kept, :15);
//                  column indices ascend inside a row (:12-20).  The reference does this as a
//                  single-threaded stride-N host scan (0.69 s at 4096^2); here it is a
//                  count / scan / fill over (column, 64-row slab) pairs: coalesced reads of A, the
//                  fill transposes 64 x 64 tiles through LDS so every run of a CSR row is one write.
//   dense_gemv     y[i] = sum_j x[j]*A[j*N+i]: the dense slots of the launcher API --
//                  naive_kernel (naive.cu:4-11), tiling_kernel (tiling_smem.cu:4-32) and the
//                  vendor slot cublas_gemv_gpu (cublas.cu:4-44).
#include <mutex>
#include "spmv_internal.hpp"

namespace spmv {

constexpr int kSlabRows = 64;  // A is cut into slabs of 64 rows; (output column, slab) pairs are counted and filled

// counts[i*S + s] = nonzeros of column i inside slab s (rows [64 s, 64 s + 64)): thread per column, coalesced reads
__global__ __launch_bounds__(kBlock) void k_dense_count(int M, int N, int S, const float *__restrict__ A,
                                                        int32_t *__restrict__ counts)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int s = blockIdx.y;
    if (i >= N) return;
    const int j0 = s * kSlabRows;
    const int j1 = j0 + kSlabRows < M ? j0 + kSlabRows : M;
    int cnt = 0;
    for (int j = j0; j < j1; ++j) cnt += (A[(size_t)j * N + i] != 0.0f) ? 1 : 0;
    counts[(size_t)i * S + s] = cnt;
}

// In-place exclusive scan of n int32 by one 1024-thread workgroup; total -> *total_out.
__global__ __launch_bounds__(1024) void k_exclusive_scan(int64_t n, int32_t *__restrict__ data,
                                                         int32_t *__restrict__ total_out)
{
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = t * per, e = (b + per < n) ? b + per : n;
    int64_t sum = 0;
    for (int64_t k = b; k < e; ++k) sum += data[k];
    part[t] = sum;
    __syncthreads();
    // Hillis-Steele over 1024 partials
    for (int o = 1; o < 1024; o <<= 1) {
        int64_t v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int64_t run = (t == 0) ? 0 : part[t - 1];
    for (int64_t k = b; k < e; ++k) {
        int32_t v = data[k];
        data[k] = (int32_t)run;
        run += v;
    }
    if (t == 1023) *total_out = (int32_t)part[1023];
}

// Large inputs: a three-step scan over tiles of kScanTile elements -- tile sums, the one-workgroup scan above over
// those sums, then a scan inside every tile started from its offset (the one-workgroup scan alone took 0.42 ms
// for the 262 144 slab counts of a 4096 x 4096 matrix, more than the two passes over the matrix together).
constexpr int kScanTile = 4096;   // 256 threads x 16 consecutive elements

__global__ __launch_bounds__(kBlock) void k_scan_tile_sums(int64_t n, const int32_t *__restrict__ data,
                                                           int32_t *__restrict__ sums)
{
    __shared__ int part[kBlock / kWave];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 16;
    int sum = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (base + i < n) sum += data[base + i];
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(kBlock) void k_scan_tiles(int64_t n, int32_t *__restrict__ data,
                                                       const int32_t *__restrict__ offsets)
{
    __shared__ int part[kBlock];
    const int t = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)t * 16;
    int v[16];
    int sum = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = base + i < n ? data[base + i] : 0;
        sum += v[i];
    }
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < kBlock; o <<= 1) {
        const int a = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += a;
        __syncthreads();
    }
    int run = offsets[blockIdx.x] + part[t] - sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (base + i < n) data[base + i] = run;
        run += v[i];
    }
}

int exclusive_scan_i32(int32_t *d_data, int64_t n, int32_t *d_total, hipStream_t s)
{
    hipError_t e;
    if (n <= 4 * kScanTile) {
        hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, s, n, d_data, d_total);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "k_exclusive_scan", __FILE__, __LINE__);
        return SPMV_OK;
    }
    const int64_t tiles = (n + kScanTile - 1) / kScanTile;
    DevPtr<int32_t> sums;
    SPMV_HIP_TRY(sums.alloc((size_t)tiles));
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)tiles), dim3(kBlock), 0, s, n, d_data, sums.p);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_scan_tile_sums", __FILE__, __LINE__);
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, s, tiles, sums.p, d_total);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_exclusive_scan", __FILE__, __LINE__);
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(kBlock), 0, s, n, d_data, sums.p);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_scan_tiles", __FILE__, __LINE__);
    SPMV_HIP_TRY(hipStreamSynchronize(s));   // `sums` is freed on return
    return SPMV_OK;
}

// row_ptr[i] = offs[i*S] for i < N, row_ptr[N] = nnz
__global__ void k_row_ptr_from_offsets(int N, int S, const int32_t *__restrict__ offs,
                                       const int32_t *__restrict__ total, int32_t *__restrict__ row_ptr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) row_ptr[i] = offs[(size_t)i * S];
    else if (i == N) row_ptr[N] = *total;
}

// One workgroup per (64 output columns) x (one slab of 64 rows): the 64 x 64 tile of A is read with coalesced
// rows into LDS, then every wave takes 16 of the columns: lane = row of the slab, the kept elements of a column are
// ranked with a ballot and written as ONE contiguous run (ascending row = ascending CSR column, matrix_csr.cpp:12-20).
__global__ __launch_bounds__(kBlock) void k_dense_fill(int M, int N, int S, const float *__restrict__ A,
                                                       const int32_t *__restrict__ offs,
                                                       int32_t *__restrict__ col_idx, float *__restrict__ vals)
{
    __shared__ float tile[kSlabRows][kWave + 1];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int i0 = blockIdx.x * kWave;
    const int s = blockIdx.y;
    const int j0 = s * kSlabRows;
    for (int jj = w; jj < kSlabRows; jj += kBlock / kWave) {
        const int j = j0 + jj;
        tile[jj][lane] = (j < M && i0 + lane < N) ? A[(size_t)j * N + i0 + lane] : 0.0f;
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    constexpr int kColsPerWave = kWave / (kBlock / kWave);
    for (int c = w * kColsPerWave; c < (w + 1) * kColsPerWave; ++c) {
        if (i0 + c >= N) break;   // wave-uniform
        const float v = tile[lane][c];
        const bool keep = v != 0.0f;   // rows past M were loaded as 0
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int32_t p = offs[(size_t)(i0 + c) * S + s] + __popcll(m & lt);
            vals[p] = v;
            col_idx[p] = j0 + lane;
        }
    }
}

static int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
    return SPMV_OK;
}

int dense_to_csr(int M, int N, const float *d_A, hipStream_t s, spmv_csr_t **out)
{
    int rc;
    DevPtr<int32_t> counts, total, row_ptr, col;
    DevPtr<float> val;
    const int S = M > 0 ? (M + kSlabRows - 1) / kSlabRows : 1;
    const size_t ncnt = (size_t)N * (size_t)S;
    if (S > 65535) {
        set_error("dense_to_csr: M = %d exceeds the 64 x 65535 rows one launch covers", M);
        return SPMV_ERR_INVALID;
    }
    SPMV_HIP_TRY(counts.alloc(ncnt));
    SPMV_HIP_TRY(total.alloc(1));
    SPMV_HIP_TRY(row_ptr.alloc((size_t)N + 1));
    SPMV_HIP_TRY(hipMemsetAsync(total.p, 0, sizeof(int32_t), s));
    if (N > 0) {
        const int gx = (N + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(k_dense_count, dim3(gx, S), dim3(kBlock), 0, s, M, N, S, d_A, counts.p);
        if ((rc = check_launch("k_dense_count"))) return rc;
        if ((rc = exclusive_scan_i32(counts.p, (int64_t)ncnt, total.p, s))) return rc;
    }
    hipLaunchKernelGGL(k_row_ptr_from_offsets, dim3((N + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, N, S,
                       counts.p, total.p, row_ptr.p);
    if ((rc = check_launch("k_row_ptr_from_offsets"))) return rc;
    int32_t nnz = 0;
    SPMV_HIP_TRY(hipMemcpyAsync(&nnz, total.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));
    SPMV_HIP_TRY(col.alloc((size_t)nnz));
    SPMV_HIP_TRY(val.alloc((size_t)nnz));
    if (N > 0 && nnz > 0) {
        hipLaunchKernelGGL(k_dense_fill, dim3((N + kWave - 1) / kWave, S), dim3(kBlock), 0, s, M, N, S, d_A, counts.p,
                           col.p, val.p);
        if ((rc = check_launch("k_dense_fill"))) return rc;
    }
    SPMV_HIP_TRY(hipStreamSynchronize(s));

    spmv_csr *h = new spmv_csr();
    h->rows = N;
    h->cols = M;
    h->nnz = nnz;
    h->owns_arrays = true;
    if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
    h->d_row_ptr = row_ptr.release();
    h->d_col_idx = col.release();
    h->d_vals = val.release();
    *out = h;
    return SPMV_OK;
}

// ---------------------------------------------------------------------------
// mode 0: thread per output, j ascending, unfused -> bit-identical to SgemvCPU (tester.cpp:36-45)
__global__ __launch_bounds__(kBlock) void k_gemv_naive(int M, int N, const float *__restrict__ A,
                                                       const float *__restrict__ x, float *__restrict__ y)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    float acc = 0.0f;
    for (int j = 0; j < M; ++j) {
        float p = x[j] * A[(size_t)j * N + i];
        acc = acc + p;
    }
    y[i] = acc;
}

// mode 1: same order, x staged through LDS 1024 entries at a time
__global__ __launch_bounds__(kBlock) void k_gemv_xtile(int M, int N, const float *__restrict__ A,
                                                       const float *__restrict__ x, float *__restrict__ y)
{
#pragma clang fp contract(off)
    __shared__ float xs[1024];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    float acc = 0.0f;
    for (int jb = 0; jb < M; jb += 1024) {
        __syncthreads();
        for (int t = threadIdx.x; t < 1024; t += kBlock) xs[t] = (jb + t < M) ? x[jb + t] : 0.0f;
        __syncthreads();
        const int jn = (M - jb) < 1024 ? (M - jb) : 1024;
        if (i < N)
            for (int j = 0; j < jn; ++j) {
                float p = xs[j] * A[(size_t)(jb + j) * N + i];
                acc = acc + p;
            }
    }
    if (i < N) y[i] = acc;
}

constexpr int kSlabs = 64;  // dense GEMV: row slabs of A processed in parallel per output column

__device__ __forceinline__ void slab_range(int M, int s, int &j0, int &j1)
{
    const int per = (M + kSlabs - 1) / kSlabs;
    j0 = s * per;
    j1 = j0 + per;
    if (j0 > M) j0 = M;
    if (j1 > M) j1 = M;
}

// mode 2: rows of A split into kSlabs slabs -> N/256 x kSlabs workgroups stream A with
// coalesced 256-B-per-wave rows; per-slab partials are combined in slab order.
// mode 3 (SKIP): the same with the reference's activation-sparsity idea (asp_kernel_v*,
// /root/reference/src/kernels/asp.cu:20-26: "if x_i != 0 load A and FMA"): x[j] is the same for every
// lane, so the test is wave-uniform and a zero x[j] skips the 1-KiB row segment of A entirely --
// with the tester's 50 %-zero x (tester.cpp:154) half of A is never read.
template <bool SKIP>
__global__ __launch_bounds__(kBlock) void k_gemv_split(int M, int N, const float *__restrict__ A,
                                                       const float *__restrict__ x, float *__restrict__ part)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int s = blockIdx.y;
    if (i >= N) return;
    int j0, j1;
    slab_range(M, s, j0, j1);
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int j = j0;
    for (; j + 3 < j1; j += 4) {
        const float x0 = x[j], x1 = x[j + 1], x2 = x[j + 2], x3 = x[j + 3];
        if (!SKIP || x0 != 0.0f) a0 = fmaf(x0, A[(size_t)j * N + i], a0);
        if (!SKIP || x1 != 0.0f) a1 = fmaf(x1, A[(size_t)(j + 1) * N + i], a1);
        if (!SKIP || x2 != 0.0f) a2 = fmaf(x2, A[(size_t)(j + 2) * N + i], a2);
        if (!SKIP || x3 != 0.0f) a3 = fmaf(x3, A[(size_t)(j + 3) * N + i], a3);
    }
    for (; j < j1; ++j) {
        const float xj = x[j];
        if (!SKIP || xj != 0.0f) a0 = fmaf(xj, A[(size_t)j * N + i], a0);
    }
    part[(size_t)s * N + i] = (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(kBlock) void k_gemv_combine(int N, const float *__restrict__ part, float *__restrict__ y)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    float acc = 0.0f;
    for (int s = 0; s < kSlabs; ++s) acc += part[(size_t)s * N + i];
    y[i] = acc;
}

size_t dense_gemv_workspace_bytes(int N, int mode)
{
    return (mode == 2 || mode == 3) ? sizeof(float) * (size_t)kSlabs * (size_t)(N > 0 ? N : 0) : 0;
}

// Allocation-free and asynchronous: modes 2/3 put their per-slab partials into the caller's workspace.
int dense_gemv_ws(int M, int N, const float *d_A, const float *d_x, float *d_y, int mode, void *d_ws, size_t ws_bytes,
                  hipStream_t s)
{
    if (N == 0) return SPMV_OK;
    const int blocks = (N + kBlock - 1) / kBlock;
    int rc;
    if (mode == 0) {
        hipLaunchKernelGGL(k_gemv_naive, dim3(blocks), dim3(kBlock), 0, s, M, N, d_A, d_x, d_y);
        return check_launch("k_gemv_naive");
    }
    if (mode == 1) {
        hipLaunchKernelGGL(k_gemv_xtile, dim3(blocks), dim3(kBlock), 0, s, M, N, d_A, d_x, d_y);
        return check_launch("k_gemv_xtile");
    }
    if (mode == 2 || mode == 3) {
        if (!d_ws || ws_bytes < dense_gemv_workspace_bytes(N, mode)) {
            set_error("spmv_dense_gemv_ws: mode %d needs %zu bytes of workspace, got %zu", mode,
                      dense_gemv_workspace_bytes(N, mode), d_ws ? ws_bytes : (size_t)0);
            return SPMV_ERR_INVALID;
        }
        float *part = static_cast<float *>(d_ws);
        if (mode == 3)
            hipLaunchKernelGGL(k_gemv_split<true>, dim3(blocks, kSlabs), dim3(kBlock), 0, s, M, N, d_A, d_x, part);
        else
            hipLaunchKernelGGL(k_gemv_split<false>, dim3(blocks, kSlabs), dim3(kBlock), 0, s, M, N, d_A, d_x, part);
        if ((rc = check_launch("k_gemv_split"))) return rc;
        hipLaunchKernelGGL(k_gemv_combine, dim3(blocks), dim3(kBlock), 0, s, N, part, d_y);
        return check_launch("k_gemv_combine");
    }
    set_error("spmv_dense_gemv: unknown mode %d", mode);
    return SPMV_ERR_VARIANT;
}

// ---------------------------------------------------------------------------
// The reference's ASP layout on the device (ASPMatrix, /root/reference/src/asp.cpp:3-14): A[M][N] re-tiled into 32 x 32
// blocks, column panel by column panel -- which makes every panel of 32 columns one contiguous M x 32 row-major array:
//   asp[(bn/32 * M + j) * 32 + c] = A[j * N + bn + c].
// k_asp_retile copies A into it (128-byte reads and writes); k_asp_gemv multiplies from it with the x == 0 skip of
// asp_kernel_v* (src/kernels/asp.cu:20-26) re-derived for 64 lanes: a wavefront owns one panel and takes TWO consecutive
// rows per instruction (lanes 0-31 row j, lanes 32-63 row j + 1: 256 contiguous bytes), each half skipping its row when
// its x is zero (a half that is masked off requests nothing); the halves meet through one shuffle at the end.  Rows split
// into the same kSlabs slabs as modes 2/3, partials combined in slab order by k_gemv_combine.
__global__ __launch_bounds__(kBlock) void k_asp_retile(int M, int N, const float *__restrict__ A, float *__restrict__ asp)
{
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int64_t panel = idx / ((int64_t)M * 32), rem = idx % ((int64_t)M * 32);
    const int j = (int)(rem >> 5), c = (int)(rem & 31);
    asp[idx] = A[(int64_t)j * N + panel * 32 + c];
}

__global__ __launch_bounds__(kBlock) void k_asp_gemv(int M, int N, const float *__restrict__ asp,
                                                     const float *__restrict__ x, float *__restrict__ part)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int panel = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int s = blockIdx.y;
    if (panel * 32 >= N) return;   // wave-uniform
    int j0, j1;
    slab_range(M, s, j0, j1);
    const int half = lane >> 5, c = lane & 31;
    const float *p = asp + (int64_t)panel * M * 32 + c;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int j = j0 + half;
    for (; j + 6 < j1; j += 8) {       // four row pairs in flight
        const float x0 = x[j], x1 = x[j + 2], x2 = x[j + 4], x3 = x[j + 6];
        if (x0 != 0.0f) a0 = fmaf(x0, p[(int64_t)j * 32], a0);
        if (x1 != 0.0f) a1 = fmaf(x1, p[(int64_t)(j + 2) * 32], a1);
        if (x2 != 0.0f) a2 = fmaf(x2, p[(int64_t)(j + 4) * 32], a2);
        if (x3 != 0.0f) a3 = fmaf(x3, p[(int64_t)(j + 6) * 32], a3);
    }
    for (; j < j1; j += 2) {
        const float xj = x[j];
        if (xj != 0.0f) a0 = fmaf(xj, p[(int64_t)j * 32], a0);
    }
    float acc = (a0 + a1) + (a2 + a3);
    acc += __shfl_down(acc, 32, kWave);
    if (half == 0) part[(size_t)s * N + panel * 32 + c] = acc;
}

int asp_retile(int M, int N, const float *d_A, float *d_asp, hipStream_t s)
{
    const int64_t n = (int64_t)M * N;
    if (n == 0) return SPMV_OK;
    hipLaunchKernelGGL(k_asp_retile, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, M, N, d_A, d_asp);
    return check_launch("k_asp_retile");
}

int asp_gemv_ws(int M, int N, const float *d_asp, const float *d_x, float *d_y, void *d_ws, size_t ws_bytes, hipStream_t s)
{
    if (N == 0) return SPMV_OK;
    if (!d_ws || ws_bytes < dense_gemv_workspace_bytes(N, 3)) {
        set_error("spmv_asp_gemv_ws: needs %zu bytes of workspace, got %zu", dense_gemv_workspace_bytes(N, 3),
                  d_ws ? ws_bytes : (size_t)0);
        return SPMV_ERR_INVALID;
    }
    float *part = static_cast<float *>(d_ws);
    const int panels = N / 32;
    hipLaunchKernelGGL(k_asp_gemv, dim3((panels + kBlock / kWave - 1) / (kBlock / kWave), kSlabs), dim3(kBlock), 0, s, M, N,
                       d_asp, d_x, part);
    if (int rc = check_launch("k_asp_gemv")) return rc;
    hipLaunchKernelGGL(k_gemv_combine, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, s, N, part, d_y);
    return check_launch("k_gemv_combine");
}

// The library-owned workspace behind spmv_dense_gemv (the entry without a workspace argument): one buffer per
// device, grown on demand, handed from call to call in STREAM ORDER -- a call on another stream than the last
// user's first makes its stream wait for the event recorded behind that user's combine kernel.  So the entry is
// asynchronous (no host wait) as long as the buffer is large enough; growing it waits for the last user once.
namespace {
struct DenseWorkspace {
    std::mutex mu;
    void *p = nullptr;
    size_t bytes = 0;
    hipStream_t last_stream = nullptr;
    hipEvent_t last_done = nullptr;
    bool used = false;
};
DenseWorkspace g_dense_ws[64];
}  // namespace

int dense_gemv(int M, int N, const float *d_A, const float *d_x, float *d_y, int mode, hipStream_t s)
{
    const size_t need = dense_gemv_workspace_bytes(N, mode);
    if (need == 0) return dense_gemv_ws(M, N, d_A, d_x, d_y, mode, nullptr, 0, s);
    int dev = 0;
    SPMV_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("spmv_dense_gemv: device id %d outside the workspace table", dev); return SPMV_ERR_INVALID; }
    DenseWorkspace &w = g_dense_ws[dev];
    std::lock_guard<std::mutex> lock(w.mu);
    if (!w.last_done) SPMV_HIP_TRY(hipEventCreateWithFlags(&w.last_done, hipEventDisableTiming));
    if (w.bytes < need) {
        if (w.used) SPMV_HIP_TRY(hipEventSynchronize(w.last_done));   // the old buffer's last reader has finished
        if (w.p) SPMV_HIP_TRY(hipFree(w.p));
        w.p = nullptr; w.bytes = 0;
        const size_t want = need < (1u << 20) ? (1u << 20) : need;
        SPMV_HIP_TRY(hipMalloc(&w.p, want));
        w.bytes = want;
        w.used = false;
    }
    if (w.used && w.last_stream != s) SPMV_HIP_TRY(hipStreamWaitEvent(s, w.last_done, 0));
    int rc = dense_gemv_ws(M, N, d_A, d_x, d_y, mode, w.p, w.bytes, s);
    if (rc) return rc;
    SPMV_HIP_TRY(hipEventRecord(w.last_done, s));
    w.last_stream = s;
    w.used = true;
    return SPMV_OK;
}

}  // namespace spmv
