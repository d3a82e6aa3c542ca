// kernels_tcsr.hip -- the reference's tiled bitmap-CSR format on gfx950.
//
// Format: exactly TCSRMatrix (/root/reference/src/tcsr.cpp:5-38, src/include/tcsr.hpp:4-23) --
//   32x32 blocks ordered output strip (block_x) outer, input block (block_y) inner;
//   bitmaps[b*32 + i] : word i of block b = output column block_x+i, bit j = input row block_y+j;
//   vals              : nonzeros in that bit order (column by column inside a block), no padding;
//   blk_idx[b]        : exclusive prefix of nonzeros per block, one trailing sentinel.
// It pays above a density of 1/32 (4 B + 1 bit per nonzero instead of CSR's 8 B): the regime of the
// reference's own tester (4096^2 at 50 %: 35.6 MB against 67.1 MB).
//
// The reference builds it with a host loop and multiplies with csr_tiling_kernel
// (src/kernels/csr_tiling.cu:24-114: 8 warps decompress a 32x32 tile into shared memory, three
// barriers per 32 rows, one warp computes).  Re-derived for wave64:
//   build : half a wavefront per block -- lane = output column, 32 coalesced 128-byte row reads,
//           popcount, one-workgroup scan, fill.
//   SpMV  : wavefront = 64 output columns x a segment of the input blocks, lane = output column;
//           block values staged in wavefront-private LDS with coalesced loads, each lane walks the
//           set bits of its own bitmap word (see k_tcsr_spmv).  No decompression, no barrier.
#include "spmv_internal.hpp"

using f4 = float __attribute__((ext_vector_type(4)));

struct spmv_tcsr {
    int M = 0, N = 0;
    int64_t nnz = 0, nblocks = 0;
    int32_t *d_blk_idx = nullptr;   // [nblocks+1]
    uint32_t *d_bitmaps = nullptr;  // [M*N/32]
    float *d_vals = nullptr;        // [nnz]
    int nseg = 0;                   // input-dimension segments per strip pair
    float *d_partial = nullptr;     // [nseg][N] partial sums, combined in segment order
};

namespace spmv {

// ---- build ----------------------------------------------------------------------------------
// half-wave per block: lane i (0..31) = output column block_x+i; 8 blocks per 256-thread workgroup
__global__ __launch_bounds__(kBlock) void k_tcsr_bits(int M, int N, const float *__restrict__ A,
                                                      uint32_t *__restrict__ bitmaps, int32_t *__restrict__ counts)
{
    const int64_t nby = M / 32, nblk = nby * (N / 32);
    const int64_t b = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int i = threadIdx.x & 31;
    uint32_t word = 0;
    if (b < nblk) {
        const int64_t bx = b / nby, by = b % nby;
        const float *p = A + (size_t)(by * 32) * N + bx * 32 + i;
#pragma unroll 8
        for (int j = 0; j < 32; ++j) word |= (p[(size_t)j * N] != 0.0f ? 1u : 0u) << j;
        bitmaps[b * 32 + i] = word;
    }
    int cnt = __popc(word);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 32);
    if (i == 0 && b < nblk) counts[b] = cnt;
}

__global__ __launch_bounds__(kBlock) void k_tcsr_fill(int M, int N, const float *__restrict__ A,
                                                      const uint32_t *__restrict__ bitmaps,
                                                      const int32_t *__restrict__ blk_idx, float *__restrict__ vals)
{
    const int64_t nby = M / 32, nblk = nby * (N / 32);
    const int64_t b = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int i = threadIdx.x & 31;
    const uint32_t word = b < nblk ? bitmaps[b * 32 + i] : 0u;
    // exclusive prefix of the per-column counts inside the half
    int inc = __popc(word);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        int v = __shfl_up(inc, o, 32);
        if (i >= o) inc += v;
    }
    if (b >= nblk) return;
    int64_t pos = (int64_t)blk_idx[b] + inc - __popc(word);
    const int64_t bx = b / nby, by = b % nby;
    const float *p = A + (size_t)(by * 32) * N + bx * 32 + i;
    for (int j = 0; j < 32; ++j) {
        const float v = p[(size_t)j * N];  // coalesced 128 B per half
        if ((word >> j) & 1u) vals[pos++] = v;
    }
}

__global__ void k_tcsr_sentinel(int64_t nblk, const int32_t *__restrict__ total, int32_t *__restrict__ blk_idx)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) blk_idx[nblk] = *total;
}

// ---- SpMV -------------------------------------------------------------------------------------
// Wavefront = 64 output columns (two adjacent strips, one per 32-lane half) x one segment of the
// input blocks; lane = output column.  Per 32x32 block the half stages the block's (contiguous)
// values into wavefront-private LDS with coalesced loads, every lane takes its own bitmap word
// (one coalesced 128-byte load per half), finds where its values start with a popcount prefix
// over the half, and walks the set bits of its word: acc += vals[off++] * x[j].  One accumulator
// per lane, no decompression, no workgroup barrier.  The input dimension is cut into kSeg segments
// per strip pair (more wavefronts than N/64 alone provides); their partial sums are combined in
// segment order by k_tcsr_combine.
constexpr int kTcsrWaves = kBlock / kWave;

template <bool DENSE>
__global__ __launch_bounds__(kBlock) void k_tcsr_spmv(int M, int N, int nseg, const int32_t *__restrict__ blk_idx,
                                                      const uint32_t *__restrict__ bitmaps,
                                                      const float *__restrict__ vals, const float *__restrict__ x,
                                                      float *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float vals_s[kTcsrWaves][2][1024 + 16];
    __shared__ float xs[kTcsrWaves][32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, i = lane & 31;
    const int nby = M / 32, nstrips = N / 32;
    const int strip = blockIdx.x * 2 + half;
    const bool live = strip < nstrips;  // N/32 may be odd: the last wavefront has one idle half
    const int seg = blockIdx.y * kTcsrWaves + wave;
    const int per = (nby + nseg - 1) / nseg;
    const int by0 = seg * per, by1 = (by0 + per < nby) ? by0 + per : nby;
    float *vs = vals_s[wave][half];
    float *xw = xs[wave];

    float acc = 0.0f;
    for (int by = by0; by < by1; ++by) {
        const int64_t b = (int64_t)(live ? strip : 0) * nby + by;
        const uint32_t word = live ? bitmaps[b * 32 + i] : 0u;
        const int base = blk_idx[b];
        const int cnt = live ? blk_idx[b + 1] - base : 0;
        // the block's values (contiguous, <= 1024 floats, start anywhere): 16-byte loads from the
        // 16-byte-aligned address below `base`, parked in LDS at the same alignment -- up to eight
        // loads per lane, all issued before the first store (a load-then-store loop would expose
        // one HBM round trip per iteration); value k of the block then sits at vs[shift + k]
        const int shift = base & 3;
        const int nvec = (shift + cnt + 3) >> 2;  // <= 257: the 9th trip covers a block of 1021..1024 values starting at shift 1..3
        f4 r[9];
        const f4 *v4 = reinterpret_cast<const f4 *>(vals + (base - shift));
#pragma unroll
        for (int q = 0; q < 9; ++q)
            if (i + 32 * q < nvec) r[q] = v4[i + 32 * q];
        if (half == 0) xw[i] = x[by * 32 + i];
#pragma unroll
        for (int q = 0; q < 9; ++q)
            if (i + 32 * q < nvec) *reinterpret_cast<f4 *>(vs + 4 * (i + 32 * q)) = r[q];
        // exclusive prefix of the per-column counts inside the half
        int inc = __popc(word);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            int v = __shfl_up(inc, o, 32);
            if (i >= o) inc += v;
        }
        int off = shift + inc - __popc(word);
        if (DENSE) {
            // all 32 bit positions, predicated: fixed trip count, 32 independent LDS reads in flight
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int k = off + __popc(word & ((1u << j) - 1u));
                const float v = ((word >> j) & 1u) ? vs[k] : 0.0f;
                acc = fmaf(v, xw[j], acc);
            }
        } else {
            // few bits per word: walk them
            uint32_t w = word;
            while (w) {
                const int j = __ffs(w) - 1;
                acc = fmaf(vs[off++], xw[j], acc);
                w &= w - 1;
            }
        }
    }
    if (live) partial[(size_t)seg * N + strip * 32 + i] = acc;
}

__global__ __launch_bounds__(kBlock) void k_tcsr_combine(int N, int nseg, const float *__restrict__ partial,
                                                         float *__restrict__ y)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    float s = 0.0f;
    for (int g = 0; g < nseg; ++g) s += partial[(size_t)g * N + i];
    y[i] = s;
}

static int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
    return SPMV_OK;
}

int tcsr_from_dense(int M, int N, const float *d_A, hipStream_t s, spmv_tcsr_t **out)
{
    const int64_t nblk = (int64_t)(M / 32) * (N / 32);
    DevPtr<int32_t> blk_idx, total;
    DevPtr<uint32_t> bitmaps;
    DevPtr<float> vals;
    SPMV_HIP_TRY(blk_idx.alloc((size_t)nblk + 1));
    SPMV_HIP_TRY(total.alloc(1));
    SPMV_HIP_TRY(bitmaps.alloc((size_t)nblk * 32));
    SPMV_HIP_TRY(hipMemsetAsync(total.p, 0, sizeof(int32_t), s));
    int rc;
    int32_t nnz = 0;
    if (nblk > 0) {
        const unsigned grid = (unsigned)((nblk + 7) / 8);
        hipLaunchKernelGGL(k_tcsr_bits, dim3(grid), dim3(kBlock), 0, s, M, N, d_A, bitmaps.p, blk_idx.p);
        if ((rc = check_launch("k_tcsr_bits"))) return rc;
        if ((rc = exclusive_scan_i32(blk_idx.p, nblk, total.p, s))) return rc;
        hipLaunchKernelGGL(k_tcsr_sentinel, dim3(1), dim3(64), 0, s, nblk, total.p, blk_idx.p);
        if ((rc = check_launch("k_tcsr_sentinel"))) return rc;
        SPMV_HIP_TRY(hipMemcpyAsync(&nnz, total.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        SPMV_HIP_TRY(hipStreamSynchronize(s));
        SPMV_HIP_TRY(vals.alloc((size_t)nnz + 4));  // k_tcsr_spmv reads whole 16-byte vectors around a block
        if (nnz > 0) {
            hipLaunchKernelGGL(k_tcsr_fill, dim3(grid), dim3(kBlock), 0, s, M, N, d_A, bitmaps.p, blk_idx.p, vals.p);
            if ((rc = check_launch("k_tcsr_fill"))) return rc;
        }
        SPMV_HIP_TRY(hipStreamSynchronize(s));
    } else {
        SPMV_HIP_TRY(hipMemsetAsync(blk_idx.p, 0, sizeof(int32_t), s));
        SPMV_HIP_TRY(vals.alloc(1));
        SPMV_HIP_TRY(hipStreamSynchronize(s));
    }
    // segments: enough wavefronts to fill the chip (>= 8 per CU) without shrinking a segment below 4 blocks
    int nseg = kTcsrWaves;
    const int pairs = (N / 32 + 1) / 2;
    while (pairs * nseg < 2048 && nseg * 2 <= 64 && (M / 32) / (nseg * 2) >= 4) nseg *= 2;
    DevPtr<float> partial;
    SPMV_HIP_TRY(partial.alloc((size_t)nseg * (size_t)(N ? N : 1)));
    spmv_tcsr *h = new spmv_tcsr();
    h->M = M; h->N = N; h->nnz = nnz; h->nblocks = nblk;
    h->nseg = nseg;
    h->d_partial = partial.release();
    h->d_blk_idx = blk_idx.release();
    h->d_bitmaps = bitmaps.release();
    h->d_vals = vals.release();
    *out = h;
    return SPMV_OK;
}

int tcsr_run(const spmv_tcsr &h, const float *d_x, float *d_y, hipStream_t s)
{
    if (h.N == 0) return SPMV_OK;
    const int pairs = (h.N / 32 + 1) / 2;
    const bool dense = h.nnz * 4 >= (int64_t)h.M * h.N;  // >= 8 of 32 bits set per word on average
    if (dense)
        hipLaunchKernelGGL(k_tcsr_spmv<true>, dim3(pairs, h.nseg / kTcsrWaves), dim3(kBlock), 0, s, h.M, h.N, h.nseg,
                           h.d_blk_idx, h.d_bitmaps, h.d_vals, d_x, h.d_partial);
    else
        hipLaunchKernelGGL(k_tcsr_spmv<false>, dim3(pairs, h.nseg / kTcsrWaves), dim3(kBlock), 0, s, h.M, h.N, h.nseg,
                           h.d_blk_idx, h.d_bitmaps, h.d_vals, d_x, h.d_partial);
    int rc = check_launch("k_tcsr_spmv");
    if (rc) return rc;
    hipLaunchKernelGGL(k_tcsr_combine, dim3((h.N + kBlock - 1) / kBlock), dim3(kBlock), 0, s, h.N, h.nseg, h.d_partial, d_y);
    return check_launch("k_tcsr_combine");
}

void tcsr_free(spmv_tcsr *h)
{
    if (!h) return;
    if (h->d_blk_idx) (void)hipFree(h->d_blk_idx);
    if (h->d_bitmaps) (void)hipFree(h->d_bitmaps);
    if (h->d_vals) (void)hipFree(h->d_vals);
    if (h->d_partial) (void)hipFree(h->d_partial);
    delete h;
}

void tcsr_dims(const spmv_tcsr &h, int *M, int *N) { *M = h.M; *N = h.N; }

int tcsr_sizes(const spmv_tcsr &h, int64_t *n_blk_idx, int64_t *n_bitmaps, int64_t *n_vals)
{
    if (n_blk_idx) *n_blk_idx = h.nblocks + 1;
    if (n_bitmaps) *n_bitmaps = h.nblocks * 32;
    if (n_vals) *n_vals = h.nnz;
    return SPMV_OK;
}

int tcsr_download(const spmv_tcsr &h, int32_t *blk_idx, uint32_t *bitmaps, float *vals)
{
    if (blk_idx) SPMV_HIP_TRY(hipMemcpy(blk_idx, h.d_blk_idx, sizeof(int32_t) * ((size_t)h.nblocks + 1), hipMemcpyDeviceToHost));
    if (bitmaps && h.nblocks)
        SPMV_HIP_TRY(hipMemcpy(bitmaps, h.d_bitmaps, sizeof(uint32_t) * (size_t)h.nblocks * 32, hipMemcpyDeviceToHost));
    if (vals && h.nnz) SPMV_HIP_TRY(hipMemcpy(vals, h.d_vals, sizeof(float) * (size_t)h.nnz, hipMemcpyDeviceToHost));
    return SPMV_OK;
}

}  // namespace spmv
