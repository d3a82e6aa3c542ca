// kernels_bitmap.hip -- the reference's bitmap formats WSP / AWSP / AWSPRef on gfx950.
//
// Formats, bit for bit what the reference's host classes produce (M, N multiples of 32, tester.cpp:9-10):
//   WSP      (WSPMatrix, /root/reference/src/wsp.cpp:3-40)        bit i*M + j of the bitmap = (input j, output i) is
//            kept; vals[i*nz_max_m + k] = k-th kept element of output column i, every column padded to the longest.
//   AWSP     (AWSPMatrix, src/awsp.cpp:3-49)                       word s*M + j = input row j inside output strip s
//            (outputs 32s..32s+31), bit c = output 32s+c; the kept elements of the 32x32 block b = s*(M/32) + j/32 in
//            (row, c) order at vals[b*nz_bk_max + k], every block padded to the fullest.
//   AWSPRef  (AWSPRefMatrix, src/awsp_ref.cpp:4-58)                the same bitmap; the kept elements of (strip s,
//            quarter q of the M inputs) in (row, c) order at vals[s*off[3] + (q ? off[q-1] : 0) + k], off = inclusive
//            prefix over q of the per-quarter maxima over all strips (warp_nz_offset_).
// The reference multiplies them with 32-lane kernels: wsp_kernel_v0/v1 (src/kernels/wsp.cu:4-138: a warp per output,
// 32 rows per step), awsp_kernel_v0/1/2 (awsp.cu:5-317) and awsp_ref_kernel (awsp_ref.cu:6-185: a lane per output, one
// row per step, `__popc(word & lanemask_lt)` as the in-word rank, a running value pointer, four warps per strip and a
// shared-memory sum of their partials), wsp_sm_kernel (wsp_sm.cu:6-211) on the AWSPRef arrays.  Re-derived for wave64:
//   * bitmap words are consumed 64 bits at a time: two rows of a strip (AWSP/AWSPRef: half-wave per row) or 64 rows of
//     an output column (WSP); the word pair is wave-uniform, so it arrives through the scalar cache;
//   * rank inside the pair = __popcll(pair & lanemask_lt), the running offset advances by __popcll(pair) on the
//     scalar unit; the value loads of one step are one compact run of <= 64 floats;
//   * the x == 0 skip of the reference (awsp_ref.cu:52, awsp.cu:127-134) is kept as a lane predicate;
//   * AWSP/AWSPRef: the M inputs are cut into the reference's four quarters, eight wavefronts share a (strip, quarter),
//     each finds where its rows' values start from the popcounts of the rows before it (LDS prefix), the quarter
//     partials are added in order by a second tiny kernel -- deterministic, no atomics.
// The builders run on the device from the dense matrix (the reference: single-threaded host loops, 0.18-0.32 s at
// 4096^2) and are checked bit for bit against the reference-built arrays in tests/golden/.
#include <climits>
#include "spmv_internal.hpp"

struct spmv_bitmap {
    int format = 0;                 // enum spmv_bitmap_format
    int M = 0, N = 0;
    int device = 0;
    int64_t n_bitmaps = 0, n_vals = 0;
    int32_t stats[4] = {0, 0, 0, 0};   // WSP {nz_max_m, nz_max_n}, AWSP {nz_bk_max_}, AWSPRef warp_nz_offset_[4]
    uint32_t *d_bitmaps = nullptr;
    float *d_vals = nullptr;
    float *d_partial = nullptr;     // AWSP/AWSPRef: [4][N] quarter partials
};

namespace spmv {

namespace {

constexpr int kQuarters = 4;        // awsp_ref.cpp:12 "warp_id < 4": quarters of the input dimension
constexpr int kRowWaves = 8;        // wavefronts that share one (strip, quarter)

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
    return SPMV_OK;
}

// ---- builders ---------------------------------------------------------------------------------------------------
// AWSP / AWSPRef bitmap: word s*M + j, bit c <-> A[j][32s + c] != 0.  A wavefront reads 64 consecutive floats of a row
// (two strips) and a 64-bit ballot is the two words.  cnt[s*M + j] = popcount (scanned next).
__global__ __launch_bounds__(kBlock) void k_rowwords(int M, int N, const float *__restrict__ A,
                                                     uint32_t *__restrict__ bitmaps, int32_t *__restrict__ cnt)
{
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int j = blockIdx.y * (kBlock / kWave) + w;
    const int c0 = blockIdx.x * kWave;            // first output column of this wavefront's pair of strips
    if (j >= M) return;
    const bool in = c0 + lane < N;
    const float v = in ? A[(size_t)j * N + c0 + lane] : 0.0f;
    const unsigned long long m = __ballot(v != 0.0f);
    if (lane == 0) {
        const int s = c0 >> 5;
        const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
        bitmaps[(size_t)s * M + j] = lo;
        cnt[(size_t)s * M + j] = __popc(lo);
        if (c0 + 32 < N) {
            bitmaps[(size_t)(s + 1) * M + j] = hi;
            cnt[(size_t)(s + 1) * M + j] = __popc(hi);
        }
    }
}

// WSP bitmap: word i*(M/32) + jw, bit b <-> A[32 jw + b][i] != 0: thread per output column, coalesced rows.
__global__ __launch_bounds__(kBlock) void k_colwords(int M, int N, const float *__restrict__ A,
                                                     uint32_t *__restrict__ bitmaps)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int jw = blockIdx.y;
    if (i >= N) return;
    uint32_t word = 0;
    const float *p = A + (size_t)jw * 32 * N + i;
#pragma unroll 8
    for (int b = 0; b < 32; ++b) word |= (p[(size_t)b * N] != 0.0f ? 1u : 0u) << b;
    bitmaps[(size_t)i * (M / 32) + jw] = word;
}

// maxima of the group sums of a scanned count array: group g = elements [g*len, (g+1)*len), key = g % nkeys.
// pre = exclusive scan of cnt with pre[n] = total.  AWSP: len 32, one key (nz_bk_max_); AWSPRef: len M/4, four keys.
__global__ void k_group_max(int64_t ngroups, int len, int nkeys, const int32_t *__restrict__ pre,
                            int32_t *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const int v = pre[(g + 1) * len] - pre[g * len];
    atomicMax(&out[g % nkeys], v);
}

__global__ void k_append_total(int64_t n, const int32_t *__restrict__ total, int32_t *__restrict__ pre)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) pre[n] = *total;
}

// values of AWSP (REF = false) / AWSPRef (REF = true): half a wavefront per input row of a strip
template <bool REF>
__global__ __launch_bounds__(kBlock) void k_rowvals(int M, int N, const float *__restrict__ A,
                                                    const uint32_t *__restrict__ bitmaps,
                                                    const int32_t *__restrict__ pre, const int32_t *__restrict__ stats,
                                                    float *__restrict__ vals)
{
    const int c = threadIdx.x & 31;
    const int j = blockIdx.y * (kBlock / 32) + (threadIdx.x >> 5);
    const int s = blockIdx.x;
    if (j >= M) return;
    const uint32_t word = bitmaps[(size_t)s * M + j];
    if (!((word >> c) & 1u)) return;
    const int rank = __popc(word & ((1u << c) - 1u));
    int64_t dst;
    if (REF) {
        const int Q = M / kQuarters, q = j / Q;
        dst = (int64_t)s * stats[3] + (q ? stats[q - 1] : 0) + (pre[(size_t)s * M + j] - pre[(size_t)s * M + q * Q]) + rank;
    } else {
        const int64_t b = (int64_t)s * (M / 32) + j / 32;
        dst = b * stats[0] + (pre[(size_t)s * M + j] - pre[(size_t)s * M + (j & ~31)]) + rank;
    }
    vals[dst] = A[(size_t)j * N + 32 * s + c];
}

// AWSPRef: the per-quarter maxima become their inclusive prefix (warp_nz_offset_, awsp_ref.cpp:33-40)
__global__ void k_prefix4(int32_t *__restrict__ stats)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int run = 0;
        for (int q = 0; q < kQuarters; ++q) { run += stats[q]; stats[q] = run; }
    }
}

// WSP values: row i of the CSR (output column i) copied to vals[i*nz_max_m ...], the rest stays zero
__global__ __launch_bounds__(kBlock) void k_wsp_pad(int N, int nz_max_m, const int32_t *__restrict__ row_ptr,
                                                    const float *__restrict__ csr_vals, float *__restrict__ vals)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int i = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (i >= N) return;
    const int b = row_ptr[i], e = row_ptr[i + 1];
    for (int k = b + lane; k < e; k += kWave) vals[(size_t)i * nz_max_m + (k - b)] = csr_vals[k];
}

__global__ void k_rowlen_max(int N, const int32_t *__restrict__ row_ptr, int32_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) atomicMax(out, row_ptr[i + 1] - row_ptr[i]);
}

// ---- multiply: AWSP / AWSPRef ------------------------------------------------------------------------------------
// Workgroup = (strip s, quarter q): kRowWaves wavefronts, each a contiguous run of that quarter's rows.  Lane =
// (row parity, output c).  A wavefront takes its rows 64 at a time: ONE coalesced load brings the 64 bitmap words (lane
// l holds the word of row g + l), one more the 64 x entries; the 32 steps of the group are unrolled, each step reads
// its two words and two x entries out of those registers with v_readlane (compile-time lanes: the reference's
// broadcast-by-__shfl_sync of a lane-held word, awsp_ref.cu:68-117, on the scalar unit), ranks itself with __popcll
// and issues its value load -- 32 compact loads in flight per wavefront before the first FMA.
__device__ __forceinline__ uint32_t lane_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float lane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

template <bool REF>
__global__ __launch_bounds__(kRowWaves *kWave) void k_rows_spmv(int M, int N, const uint32_t *__restrict__ bitmaps,
                                                                 const float *__restrict__ vals,
                                                                 const int32_t *__restrict__ stats,
                                                                 const float *__restrict__ x,
                                                                 float *__restrict__ partial)
{
    __shared__ int wave_nz[kRowWaves];
    __shared__ float sums[kRowWaves][32];
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = blockIdx.x, q = blockIdx.y;
    const int half = lane >> 5, c = lane & 31;
    // AWSPRef: quarter q = rows [q*M/4, (q+1)*M/4) (awsp_ref.cpp:12-14; M/4 is a multiple of 8), a wavefront takes an
    // even number of them.  AWSP has no quarters of its own: the row BLOCKS are dealt to four groups so that every
    // wavefront starts on a block boundary, where the value pointer restarts (awsp.cpp:38-46).
    int q0, q1, per;
    if (REF) {
        const int Q = M / kQuarters;
        q0 = q * Q; q1 = q0 + Q;
        per = ((Q + kRowWaves - 1) / kRowWaves + 1) & ~1;
    } else {
        const int nb = M / 32, Qb = (nb + kQuarters - 1) / kQuarters;
        q0 = (q * Qb < nb ? q * Qb : nb) * 32;
        q1 = ((q + 1) * Qb < nb ? (q + 1) * Qb : nb) * 32;
        per = (((q1 - q0) + kRowWaves - 1) / kRowWaves + 31) & ~31;
    }
    int j0 = q0 + w * per, j1 = j0 + per;
    if (j0 > q1) j0 = q1;
    if (j1 > q1) j1 = q1;
    const uint32_t *bm = bitmaps + (size_t)s * M;

    int64_t ptr = 0;
    if (REF) {
        // where this wavefront's values start inside the (strip, quarter) stream: popcount of the rows before it
        int mine = 0;
        for (int j = j0 + lane; j < j1; j += kWave) mine += __popc(bm[j]);
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) mine += __shfl_down(mine, o, kWave);
        if (lane == 0) wave_nz[w] = mine;
        __syncthreads();
        int before = 0;
        for (int k = 0; k < w; ++k) before += wave_nz[k];
        ptr = (int64_t)s * stats[3] + (q ? stats[q - 1] : 0) + before;
    }
    const int64_t nz_bk_max = REF ? 0 : stats[0];
    const unsigned long long lt = (1ull << lane) - 1ull;
    float acc = 0.0f;
    for (int g = j0; g < j1; g += kWave) {
        const int jl = g + lane;
        const uint32_t wreg = jl < j1 ? bm[jl] : 0u;      // rows past the range: no bits, nothing loaded
        const float xreg = jl < j1 ? x[jl] : 0.0f;
        float v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (!REF && (k & 15) == 0) ptr = ((int64_t)s * (M / 32) + ((g + 2 * k) >> 5)) * nz_bk_max;   // AWSP: a block restarts
            const unsigned long long W = (unsigned long long)lane_u32(wreg, 2 * k) | ((unsigned long long)lane_u32(wreg, 2 * k + 1) << 32);
            const float xk = half ? lane_f32(xreg, 2 * k + 1) : lane_f32(xreg, 2 * k);
            const bool bit = (W >> lane) & 1ull;
            const int64_t idx = ptr + __popcll(W & lt);
            ptr += __popcll(W);
            // awsp_ref.cu:52: nothing is loaded where x is 0
            v[k] = (bit && xk != 0.0f) ? __builtin_nontemporal_load(&vals[idx]) : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float xk = half ? lane_f32(xreg, 2 * k + 1) : lane_f32(xreg, 2 * k);
            acc = fmaf(xk, v[k], acc);
        }
    }
    acc += __shfl_down(acc, 32, kWave);                   // the two row parities
    if (lane < 32) sums[w][c] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = sums[0][c];
#pragma unroll
        for (int k = 1; k < kRowWaves; ++k) t += sums[k][c];       // wavefront order: deterministic
        partial[(size_t)q * N + 32 * s + c] = t;
    }
}

__global__ __launch_bounds__(kBlock) void k_quarter_combine(int N, const float *__restrict__ partial, float *__restrict__ y)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    float acc = partial[i];
#pragma unroll
    for (int q = 1; q < kQuarters; ++q) acc += partial[(size_t)q * N + i];
    y[i] = acc;
}

// ---- multiply: WSP --------------------------------------------------------------------------------------------------
// One wavefront per output column (wsp.cu:13 "one warp per output"), 64 input rows per step: the lane's bit of the
// 64-bit word pair, its rank below it, x[j] coalesced with holes, the values one compact run.  The column's words are
// fetched 64 at a time (2048 rows: one coalesced load, lane l holds word l) and handed to the 32 unrolled steps with
// v_readlane, so a wavefront has 32 x loads and 32 value loads in flight.
__global__ __launch_bounds__(kBlock) void k_wsp_spmv(int M, int N, int nz_max_m, const uint32_t *__restrict__ bitmaps,
                                                     const float *__restrict__ vals, const float *__restrict__ x,
                                                     float *__restrict__ y)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int i = blockIdx.x * (kBlock / kWave) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (i >= N) return;
    const int words = M / 32;
    const uint32_t *bw = bitmaps + (size_t)i * words;
    const float *v = vals + (size_t)i * nz_max_m;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int run = 0;
    float acc = 0.0f;
    for (int g = 0; g < words; g += kWave) {
        const uint32_t wreg = g + lane < words ? bw[g + lane] : 0u;
        float a[32], xs[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const unsigned long long W = (unsigned long long)lane_u32(wreg, 2 * k) | ((unsigned long long)lane_u32(wreg, 2 * k + 1) << 32);
            const bool bit = (W >> lane) & 1ull;          // a set bit implies the row exists (words past M are 0)
            const int idx = run + __popcll(W & lt);
            run += __popcll(W);
            xs[k] = bit ? x[(g + 2 * k) * 32 + lane] : 0.0f;
            a[k] = bit ? __builtin_nontemporal_load(&v[idx]) : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = fmaf(xs[k], a[k], acc);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, kWave);
    if (lane == 0) y[i] = acc;
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------
void bitmap_free(spmv_bitmap *h)
{
    if (!h) return;
    if (h->d_bitmaps) (void)hipFree(h->d_bitmaps);
    if (h->d_vals) (void)hipFree(h->d_vals);
    if (h->d_partial) (void)hipFree(h->d_partial);
    delete h;
}

int bitmap_from_dense(int format, int M, int N, const float *d_A, hipStream_t s, spmv_bitmap_t **out)
{
    int rc;
    spmv_bitmap *h = new spmv_bitmap();
    struct Guard { spmv_bitmap *&p; ~Guard() { if (p) bitmap_free(p); } } guard{h};
    h->format = format; h->M = M; h->N = N;
    if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
    const size_t nwords = (size_t)M * (size_t)N / 32;
    h->n_bitmaps = (int64_t)nwords;
    SPMV_HIP_TRY(hipMalloc((void **)&h->d_bitmaps, sizeof(uint32_t) * (nwords ? nwords : 1)));
    DevPtr<int32_t> d_stats;
    SPMV_HIP_TRY(d_stats.alloc(4));
    SPMV_HIP_TRY(hipMemsetAsync(d_stats.p, 0, 4 * sizeof(int32_t), s));

    if (format == SPMV_FMT_WSP) {
        // the values are the CSR of A^T with every row moved to a stride of nz_max_m (wsp.cpp:31-37)
        spmv_csr_t *csr = nullptr;
        if ((rc = dense_to_csr(M, N, d_A, s, &csr))) return rc;
        struct CsrGuard { spmv_csr_t *p; ~CsrGuard() { (void)spmv_csr_destroy(p); } } cg{csr};
        if (nwords) {
            hipLaunchKernelGGL(k_colwords, dim3((N + kBlock - 1) / kBlock, M / 32), dim3(kBlock), 0, s, M, N, d_A, h->d_bitmaps);
            if ((rc = check_launch("k_colwords"))) return rc;
        }
        if (N > 0) {
            hipLaunchKernelGGL(k_rowlen_max, dim3((N + 255) / 256), dim3(256), 0, s, N, csr->d_row_ptr, d_stats.p);
            if ((rc = check_launch("k_rowlen_max"))) return rc;
        }
        SPMV_HIP_TRY(hipMemcpyAsync(h->stats, d_stats.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        SPMV_HIP_TRY(hipStreamSynchronize(s));
        h->stats[1] = N;                                       // nz_max_n (wsp.cpp:7)
        h->n_vals = (int64_t)N * h->stats[0];
        SPMV_HIP_TRY(hipMalloc((void **)&h->d_vals, sizeof(float) * (size_t)(h->n_vals ? h->n_vals : 1)));
        SPMV_HIP_TRY(hipMemsetAsync(h->d_vals, 0, sizeof(float) * (size_t)h->n_vals, s));
        if (h->n_vals) {
            hipLaunchKernelGGL(k_wsp_pad, dim3((N + 3) / 4), dim3(kBlock), 0, s, N, h->stats[0], csr->d_row_ptr, csr->d_vals,
                               h->d_vals);
            if ((rc = check_launch("k_wsp_pad"))) return rc;
        }
        SPMV_HIP_TRY(hipStreamSynchronize(s));                 // the CSR is released on return
    } else {
        const bool ref = format == SPMV_FMT_AWSP_REF;
        DevPtr<int32_t> pre, total;
        SPMV_HIP_TRY(pre.alloc(nwords + 1));
        SPMV_HIP_TRY(total.alloc(1));
        SPMV_HIP_TRY(hipMemsetAsync(total.p, 0, sizeof(int32_t), s));
        if (nwords) {
            hipLaunchKernelGGL(k_rowwords, dim3((N + kWave - 1) / kWave, (M + 3) / 4), dim3(kBlock), 0, s, M, N, d_A,
                               h->d_bitmaps, pre.p);
            if ((rc = check_launch("k_rowwords"))) return rc;
            if ((rc = exclusive_scan_i32(pre.p, (int64_t)nwords, total.p, s))) return rc;
        }
        hipLaunchKernelGGL(k_append_total, dim3(1), dim3(1), 0, s, (int64_t)nwords, total.p, pre.p);
        if ((rc = check_launch("k_append_total"))) return rc;
        const int len = ref ? M / kQuarters : 32;
        const int64_t ngroups = len ? (int64_t)nwords / len : 0;
        if (ngroups) {
            hipLaunchKernelGGL(k_group_max, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, s, ngroups, len,
                               ref ? kQuarters : 1, pre.p, d_stats.p);
            if ((rc = check_launch("k_group_max"))) return rc;
        }
        if (ref) {
            hipLaunchKernelGGL(k_prefix4, dim3(1), dim3(1), 0, s, d_stats.p);
            if ((rc = check_launch("k_prefix4"))) return rc;
        }
        SPMV_HIP_TRY(hipMemcpyAsync(h->stats, d_stats.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        SPMV_HIP_TRY(hipStreamSynchronize(s));
        h->n_vals = ref ? (int64_t)(N / 32) * h->stats[3] : (int64_t)(M / 32) * (N / 32) * h->stats[0];
        if (h->n_vals >= (1ll << 31)) {
            set_error("bitmap format: %lld padded values exceed 2^31", (long long)h->n_vals);
            return SPMV_ERR_INVALID;
        }
        SPMV_HIP_TRY(hipMalloc((void **)&h->d_vals, sizeof(float) * (size_t)(h->n_vals ? h->n_vals : 1)));
        SPMV_HIP_TRY(hipMemsetAsync(h->d_vals, 0, sizeof(float) * (size_t)h->n_vals, s));
        if (nwords && h->n_vals) {
            const dim3 grid(N / 32, (M + 7) / 8);
            if (ref) hipLaunchKernelGGL(k_rowvals<true>, grid, dim3(kBlock), 0, s, M, N, d_A, h->d_bitmaps, pre.p, d_stats.p, h->d_vals);
            else hipLaunchKernelGGL(k_rowvals<false>, grid, dim3(kBlock), 0, s, M, N, d_A, h->d_bitmaps, pre.p, d_stats.p, h->d_vals);
            if ((rc = check_launch("k_rowvals"))) return rc;
        }
        // [4][N] quarter partials, then a device copy of the four statistics for the multiply kernels
        SPMV_HIP_TRY(hipMalloc((void **)&h->d_partial, sizeof(float) * ((size_t)kQuarters * (size_t)N + 4)));
        SPMV_HIP_TRY(hipMemcpyAsync(h->d_partial + (size_t)kQuarters * N, d_stats.p, 4 * sizeof(int32_t),
                                    hipMemcpyDeviceToDevice, s));
        SPMV_HIP_TRY(hipStreamSynchronize(s));                 // pre / d_stats are released on return
    }
    *out = h;
    guard.p = nullptr;
    return SPMV_OK;
}

int bitmap_run(const spmv_bitmap &h, const float *d_x, float *d_y, hipStream_t s)
{
    if (h.N == 0) return SPMV_OK;
    if (h.M == 0 || h.n_vals == 0) {
        SPMV_HIP_TRY(hipMemsetAsync(d_y, 0, sizeof(float) * (size_t)h.N, s));
        return SPMV_OK;
    }
    if (h.format == SPMV_FMT_WSP) {
        hipLaunchKernelGGL(k_wsp_spmv, dim3((h.N + 3) / 4), dim3(kBlock), 0, s, h.M, h.N, h.stats[0], h.d_bitmaps, h.d_vals,
                           d_x, d_y);
        return check_launch("k_wsp_spmv");
    }
    // the statistics live in the four words behind the partials (device copy made at build time)
    const int32_t *d_stats = reinterpret_cast<const int32_t *>(h.d_partial + (size_t)kQuarters * h.N);
    const dim3 grid(h.N / 32, kQuarters);
    if (h.format == SPMV_FMT_AWSP_REF)
        hipLaunchKernelGGL(k_rows_spmv<true>, grid, dim3(kRowWaves * kWave), 0, s, h.M, h.N, h.d_bitmaps, h.d_vals, d_stats, d_x,
                           h.d_partial);
    else
        hipLaunchKernelGGL(k_rows_spmv<false>, grid, dim3(kRowWaves * kWave), 0, s, h.M, h.N, h.d_bitmaps, h.d_vals, d_stats, d_x,
                           h.d_partial);
    int rc = check_launch("k_rows_spmv");
    if (rc) return rc;
    hipLaunchKernelGGL(k_quarter_combine, dim3((h.N + kBlock - 1) / kBlock), dim3(kBlock), 0, s, h.N, h.d_partial, d_y);
    return check_launch("k_quarter_combine");
}

void bitmap_info(const spmv_bitmap &h, int *format, int *M, int *N, int64_t *n_bitmaps, int64_t *n_vals, int32_t stats[4])
{
    if (format) *format = h.format;
    if (M) *M = h.M;
    if (N) *N = h.N;
    if (n_bitmaps) *n_bitmaps = h.n_bitmaps;
    if (n_vals) *n_vals = h.n_vals;
    if (stats) for (int i = 0; i < 4; ++i) stats[i] = h.stats[i];
}

int bitmap_download(const spmv_bitmap &h, uint32_t *bitmaps, float *vals)
{
    if (bitmaps && h.n_bitmaps)
        SPMV_HIP_TRY(hipMemcpy(bitmaps, h.d_bitmaps, sizeof(uint32_t) * (size_t)h.n_bitmaps, hipMemcpyDeviceToHost));
    if (vals && h.n_vals)
        SPMV_HIP_TRY(hipMemcpy(vals, h.d_vals, sizeof(float) * (size_t)h.n_vals, hipMemcpyDeviceToHost));
    return SPMV_OK;
}

int bitmap_device(const spmv_bitmap &h) { return h.device; }

}  // namespace spmv
