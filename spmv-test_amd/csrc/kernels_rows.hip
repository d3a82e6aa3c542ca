// kernels_rows.hip -- row-mapped CSR SpMV kernels for gfx950 (wave64).
//
//   k_scalar   thread per row, sequential, unfused mul+add (operands staged by the workgroup)
//              role of csr_naive_kernel (/root/reference/src/kernels/csr_naive.cu:6-23);
//              the same per-row operation order as SgemvCPU (src/tester.cpp:36-45),
//              so results are bit-identical to the CPU oracle.
//   k_wave     one 64-lane wavefront per row, lanes stride the row, __shfl_down tree
//              role of wsp_kernel_v0 (src/kernels/wsp.cu:4-56): "one warp per output,
//              butterfly reduce, lane 0 stores" -- re-derived for CSR and 64 lanes.
//   k_wave_bundle  a wavefront per 64 consecutive rows: coalesced stream of their nonzeros,
//              products parked in LDS, lane per short row, whole wave per long row
//              role of wsp_kernel_v1 (src/kernels/wsp.cu:59-138), the reference's pipelined version.
//   k_vector   G-lane groups per row (G = 2..32), the short-row member of the family
//              role of asp_kernel_v* (src/kernels/asp.cu:6-211: many outputs per block).
//
// All three are HBM/gather bound: no LDS, no MFMA (0.25 flop/byte).
#include <cstdlib>
#include "spmv_internal.hpp"

namespace spmv {

// One row by the whole wave IN THE ORACLE'S ORDER: 64 lane-consecutive products per trip (coalesced, each rounded
// once, the next trip's loads already in flight), then added in order -- lane 0's first -- through a readlane chain.
// Every lane returns the sum.
__device__ __forceinline__ float scalar_row_by_wave(int lane, int32_t b, int32_t e, const int32_t *__restrict__ col_idx,
                                                    const float *__restrict__ vals, const float *__restrict__ x)
{
#pragma clang fp contract(off)
    float acc = 0.0f;       // wave-uniform
    int32_t k = b + lane;
    float p = k < e ? x[col_idx[k]] * vals[k] : 0.0f;
    for (int32_t k0 = b; k0 < e; k0 += kWave) {
        const int32_t kn = k0 + kWave + lane;
        const float pn = kn < e ? x[col_idx[kn]] * vals[kn] : 0.0f;   // next trip, in flight during the chain
        const int n = e - k0 < kWave ? e - k0 : kWave;
        if (n == kWave) {
#pragma unroll
            for (int l = 0; l < kWave; ++l)
                acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), l));
        } else {
            for (int l = 0; l < n; ++l) acc = acc + __shfl(p, l);
        }
        p = pn;
    }
    return acc;
}

// ---------------------------------------------------------------------------
// k_scalar: ONE THREAD PER ROW, terms added in ascending k with an unfused multiply and add --
// exactly the arithmetic of the host loop, so y is bit-identical to SgemvCPU / the CSR walk.
// What is re-derived is only how the operands reach the thread: the 256 rows of a workgroup own one
// contiguous range of nonzeros, so when that range fits the LDS buffer the workgroup streams it
// with coalesced loads (lane-consecutive nonzeros), rounds each product x*val once, parks the
// products in LDS, and every thread then adds ITS row's products in order.  A naive thread-per-row
// loop reads col_idx/vals at a stride of one row per lane (4-byte loads, 64 cache lines per wave
// instruction); a workgroup whose rows hold more than kScalarCap nonzeros falls back to that loop.
// (Round 3, tried and dropped, A/B in one process -- profiles/r03_scalar_ab.jsonl: (1) a fast path for workgroups of <= 4096
// nonzeros with 16-byte streams kept in registers and a plan-free LDS window of x between the range's min and max column:
// config 2 band 8192 0.0525 ms against 0.0457, uniform 0.0994 against 0.0919 -- the window adds two barriers and a round trip
// to a kernel that is a chain of dependent round trips per workgroup; (2) the same without the window: 0.0483; (3) four
// loads in flight per trip: 0.0546; (4) 16 KiB of LDS for 8 workgroups per CU: 0.0460; (5) a pad word per 32 products
// against the stride-16 bank pattern of config 2's rows: 0.0449.  None moves it: 40 % of peak at config 2 band 8192.)
constexpr int kScalarCap = 8192;  // products staged per workgroup (32 KiB of LDS -> 5 workgroups per CU)

__global__ __launch_bounds__(kBlock) void k_scalar(int64_t rows, const int32_t *__restrict__ row_ptr,
                                                   const int32_t *__restrict__ col_idx,
                                                   const float *__restrict__ vals,
                                                   const float *__restrict__ x, float *__restrict__ y)
{
    // two roundings per term, like the host loop: HIP's __fmul_rn/__fadd_rn are plain operators
    // that hipcc would contract into v_fma_f32 under its default -ffp-contract=fast
#pragma clang fp contract(off)
    __shared__ float prod[kScalarCap];
    const int64_t r0 = (int64_t)blockIdx.x * kBlock;
    const int64_t r = r0 + threadIdx.x;
    const int64_t rend = (r0 + kBlock < rows) ? r0 + kBlock : rows;
    const int32_t wb = row_ptr[r0], we = row_ptr[rend];  // the workgroup's nonzero range
    const int32_t b = r < rows ? row_ptr[r] : 0, e = r < rows ? row_ptr[r + 1] : 0;
    if (we - wb <= kScalarCap) {
        for (int32_t k = wb + (int32_t)threadIdx.x; k < we; k += kBlock) {
            const float p = x[col_idx[k]] * vals[k];
            prod[k - wb] = p;
        }
        __syncthreads();
        if (r < rows) {
            float acc = 0.0f;
            for (int32_t k = b; k < e; ++k) acc = acc + prod[k - wb];
            y[r] = acc;
        }
    } else {
        // the workgroup's rows hold more than the LDS buffer: rows of up to 256 nonzeros by their own thread (the
        // plain loop), longer ones -- a power-law row can hold 65 536 -- one at a time by the whole wave, in order,
        // through the readlane chain of k_scalar_long below (c3: 39.6 ms -> see DESIGN.md)
        const int lane = threadIdx.x & (kWave - 1);
        const bool is_long = e - b > 256;
        if (r < rows && !is_long) {
            float acc = 0.0f;
            for (int32_t k = b; k < e; ++k) {
                const float p = x[col_idx[k]] * vals[k];
                acc = acc + p;
            }
            y[r] = acc;
        }
        unsigned long long todo = __ballot(is_long);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const float acc = scalar_row_by_wave(lane, __shfl(b, src), __shfl(e, src), col_idx, vals, x);
            if (lane == 0) y[r - lane + src] = acc;   // r - lane: the wave's first row
        }
    }
}

// k_scalar_long: the same arithmetic for matrices of LONG rows (mean > 64 nonzeros: the reference's own 4096 x 4096
// at 50 %), where a thread per row means 2048 dependent memory round trips per thread.  One wavefront per row: 64
// lane-consecutive products per trip (coalesced, each rounded once, the next trip's loads already in flight), then
// added IN ORDER -- lane 0's first -- through a readlane chain: the same sequence of roundings as the host loop,
// so y stays bit-identical to SgemvCPU.
__global__ __launch_bounds__(kBlock) void k_scalar_long(int64_t rows, const int32_t *__restrict__ row_ptr,
                                                        const int32_t *__restrict__ col_idx,
                                                        const float *__restrict__ vals,
                                                        const float *__restrict__ x, float *__restrict__ y)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (r >= rows) return;  // wave-uniform
    const float acc = scalar_row_by_wave(lane, row_ptr[r], row_ptr[r + 1], col_idx, vals, x);
    if (lane == 0) y[r] = acc;
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_reduce_sum(float v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;  // lane 0 holds the total
}

// one row by the whole wave: lanes stride the row (four 64-wide slices in flight per trip when PIPE), then the
// __shfl_down tree; every lane returns the sum
template <bool PIPE>
__device__ __forceinline__ float wave_row(int lane, int32_t b, int32_t e, const int32_t *__restrict__ col_idx,
                                          const float *__restrict__ vals, const float *__restrict__ x)
{
    float acc = 0.0f;
    int32_t k = b + lane;
    if (PIPE) {
        // 8 streamed loads then 4 gathers are issued before the first use
        float a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (; k + 3 * kWave < e; k += 4 * kWave) {
            int32_t c0 = col_idx[k], c1 = col_idx[k + kWave], c2 = col_idx[k + 2 * kWave],
                    c3 = col_idx[k + 3 * kWave];
            float v0 = vals[k], v1 = vals[k + kWave], v2 = vals[k + 2 * kWave], v3 = vals[k + 3 * kWave];
            float x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
            acc = fmaf(v0, x0, acc);
            a1 = fmaf(v1, x1, a1);
            a2 = fmaf(v2, x2, a2);
            a3 = fmaf(v3, x3, a3);
        }
        acc = (acc + a1) + (a2 + a3);
    }
    for (; k < e; k += kWave) acc = fmaf(vals[k], x[col_idx[k]], acc);
    return wave_reduce_sum(acc);
}

// SPMV_WAVE: one 64-lane wavefront per row, the literal re-derivation of wsp_kernel_v0 (PIPE = false).  PIPE = true
// is what SPMV_WAVE_PIPE runs on matrices of long rows (mean > 32 nonzeros, the reference's own 4096 x 4096 / 50 %
// regime): the same mapping with four slices in flight per trip.
template <bool PIPE>
__global__ __launch_bounds__(kBlock) void k_wave(int64_t rows, const int32_t *__restrict__ row_ptr,
                                                 const int32_t *__restrict__ col_idx,
                                                 const float *__restrict__ vals,
                                                 const float *__restrict__ x, float *__restrict__ y)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (r >= rows) return;  // wave-uniform
    const float acc = wave_row<PIPE>(lane, row_ptr[r], row_ptr[r + 1], col_idx, vals, x);
    if (lane == 0) y[r] = acc;
}

// SPMV_WAVE_PIPE (the slot of wsp_kernel_v1, the reference's unrolled / prefetching version): a wavefront owns 64
// consecutive rows.  When their nonzeros fit its LDS slice the wave streams the whole contiguous range with
// coalesced loads (lane-consecutive nonzeros, the whole wave busy whatever the row lengths), parks the products,
// and every lane then adds the products of ITS row in order; rows longer than a wavefront are added by all 64
// lanes with the __shfl_down tree instead.  A bundle that does not fit is taken in as many consecutive rows as do; a
// row that does not fit on its own goes through wave_row<true> (four slices in flight, straight from memory).
constexpr int kBundleCap = 2048;   // products per wave: 8 KiB of LDS, 32 KiB per workgroup
__global__ __launch_bounds__(kBlock) void k_wave_bundle(int64_t rows, const int32_t *__restrict__ row_ptr,
                                                        const int32_t *__restrict__ col_idx,
                                                        const float *__restrict__ vals,
                                                        const float *__restrict__ x, float *__restrict__ y)
{
    __shared__ float prod_all[kBlock / kWave][kBundleCap];
    const int lane = threadIdx.x & (kWave - 1);
    float *prod = prod_all[threadIdx.x >> 6];
    const int64_t r0 = ((int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * kWave;
    if (r0 >= rows) return;  // wave-uniform; no workgroup barrier below
    const int64_t r = r0 + lane;
    const bool live = r < rows;
    const int32_t b = row_ptr[live ? r : rows], e = row_ptr[live ? r + 1 : rows];
    const int n = (int)((rows - r0 < kWave) ? rows - r0 : kWave);
    // as many consecutive rows as fit the slice at a time (usually all 64); a row that does not fit on its own is
    // added straight from memory by the whole wave, four slices in flight
    int i0 = 0;
    while (i0 < n) {
        const int32_t sb = __shfl(b, i0);
        const unsigned long long fit = __ballot(lane >= i0 && lane < n && e - sb <= kBundleCap);   // e ascends
        const int cnt = __popcll(fit);
        if (cnt == 0) {
            const float acc = wave_row<true>(lane, sb, __shfl(e, i0), col_idx, vals, x);
            if (lane == 0) y[r0 + i0] = acc;
            ++i0;
            continue;
        }
        const int i1 = i0 + cnt;
        const int32_t se = __shfl(e, i1 - 1);
        {
            // four 64-wide slices per trip: 8 streamed loads, then 4 gathers, are issued before the first use (the
            // pipelining wsp_kernel_v1 adds to v0, wsp.cu:78-134, at wave64 width)
            int32_t k = sb + lane;
            for (; k + 3 * kWave < se; k += 4 * kWave) {
                const int32_t c0 = col_idx[k], c1 = col_idx[k + kWave], c2 = col_idx[k + 2 * kWave], c3 = col_idx[k + 3 * kWave];
                const float v0 = vals[k], v1 = vals[k + kWave], v2 = vals[k + 2 * kWave], v3 = vals[k + 3 * kWave];
                const float x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
                prod[k - sb] = v0 * x0;
                prod[k - sb + kWave] = v1 * x1;
                prod[k - sb + 2 * kWave] = v2 * x2;
                prod[k - sb + 3 * kWave] = v3 * x3;
            }
            for (; k < se; k += kWave) prod[k - sb] = vals[k] * x[col_idx[k]];
        }
        const bool mine = lane >= i0 && lane < i1;
        const bool is_long = mine && e - b > kWave;
        if (mine && !is_long) {
            float acc = 0.0f;
            for (int32_t k = b; k < e; ++k) acc += prod[k - sb];
            y[r] = acc;
        }
        unsigned long long todo = __ballot(is_long);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int32_t lb = __shfl(b, src), le = __shfl(e, src);
            float acc = 0.0f;
            for (int32_t k = lb + lane; k < le; k += kWave) acc += prod[k - sb];
            acc = wave_reduce_sum(acc);
            if (lane == 0) y[r0 + src] = acc;
        }
        i0 = i1;
    }
}

// ---------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(kBlock) void k_vector(int64_t rows, const int32_t *__restrict__ row_ptr,
                                                   const int32_t *__restrict__ col_idx,
                                                   const float *__restrict__ vals,
                                                   const float *__restrict__ x, float *__restrict__ y)
{
    constexpr int kRowsPerBlock = kBlock / G;
    const int sub = threadIdx.x % G;
    const int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.x / G;
    float acc = 0.0f;
    if (r < rows) {
        const int32_t b = row_ptr[r], e = row_ptr[r + 1];
        for (int32_t k = b + sub; k < e; k += G) acc = fmaf(vals[k], x[col_idx[k]], acc);
    }
    // all 64 lanes take part in the shuffles (rows past the end carry zeros)
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, G);
    if (sub == 0 && r < rows) y[r] = acc;
}

// ---------------------------------------------------------------------------
static int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
    return SPMV_OK;
}

static bool grid_ok(int64_t blocks)
{
    if (blocks > 0x7fffffffLL) {
        set_error("grid of %lld blocks exceeds the launch limit", (long long)blocks);
        return false;
    }
    return true;
}

int launch_scalar(const spmv_csr &h, const float *x, float *y, hipStream_t s)
{
    if (h.rows == 0) return SPMV_OK;
    int64_t blocks = (h.rows + kBlock - 1) / kBlock;
    if (!grid_ok(blocks)) return SPMV_ERR_INVALID;
    if (h.nnz > 64 * h.rows) {   // long rows: a wavefront per row, still in the oracle's order
        const int64_t wblocks = (h.rows + (kBlock / kWave) - 1) / (kBlock / kWave);
        if (!grid_ok(wblocks)) return SPMV_ERR_INVALID;
        hipLaunchKernelGGL(k_scalar_long, dim3((unsigned)wblocks), dim3(kBlock), 0, s, h.rows, h.d_row_ptr,
                           h.d_col_idx, h.d_vals, x, y);
        return check_launch("k_scalar_long");
    }
    hipLaunchKernelGGL(k_scalar, dim3((unsigned)blocks), dim3(kBlock), 0, s, h.rows, h.d_row_ptr,
                       h.d_col_idx, h.d_vals, x, y);
    return check_launch("k_scalar");
}

int launch_wave(const spmv_csr &h, const float *x, float *y, bool pipelined, hipStream_t s)
{
    if (h.rows == 0) return SPMV_OK;
    constexpr int kRowsPerBlock = kBlock / kWave;
    int64_t blocks = (h.rows + kRowsPerBlock - 1) / kRowsPerBlock;
    if (!grid_ok(blocks)) return SPMV_ERR_INVALID;
    // bundles of 64 rows pay off while they fit a wave's LDS slice (mean row length <= 32); beyond that a bundle
    // would serialise 64 long rows on one wave (4096 x 4096 at 50 %: 64 waves for the whole chip, 17x slower)
    const bool bundle = pipelined && h.nnz <= 32 * h.rows;
    if (pipelined && !bundle) {
        hipLaunchKernelGGL(k_wave<true>, dim3((unsigned)blocks), dim3(kBlock), 0, s, h.rows, h.d_row_ptr, h.d_col_idx,
                           h.d_vals, x, y);
    } else if (bundle) {
        const int64_t bundles = (h.rows + kWave - 1) / kWave;
        const int64_t bblocks = (bundles + (kBlock / kWave) - 1) / (kBlock / kWave);
        hipLaunchKernelGGL(k_wave_bundle, dim3((unsigned)(bblocks ? bblocks : 1)), dim3(kBlock), 0, s, h.rows,
                           h.d_row_ptr, h.d_col_idx, h.d_vals, x, y);
    } else {
        hipLaunchKernelGGL(k_wave<false>, dim3((unsigned)blocks), dim3(kBlock), 0, s, h.rows, h.d_row_ptr, h.d_col_idx,
                           h.d_vals, x, y);
    }
    return check_launch("k_wave");
}

int plan_vector(spmv_csr &h, hipStream_t)
{
    // lanes per row = smallest power of two >= mean row length, in [2, 32]
    double mean = h.rows > 0 ? (double)h.nnz / (double)h.rows : 0.0;
    int g = 2;
    while (g < 32 && (double)g < mean) g <<= 1;
    h.vector_width = g;
    return SPMV_OK;
}

template <int G>
static int launch_vector_g(const spmv_csr &h, const float *x, float *y, hipStream_t s)
{
    constexpr int kRowsPerBlock = kBlock / G;
    int64_t blocks = (h.rows + kRowsPerBlock - 1) / kRowsPerBlock;
    if (!grid_ok(blocks)) return SPMV_ERR_INVALID;
    hipLaunchKernelGGL(k_vector<G>, dim3((unsigned)blocks), dim3(kBlock), 0, s, h.rows, h.d_row_ptr,
                       h.d_col_idx, h.d_vals, x, y);
    return check_launch("k_vector");
}

int launch_vector(const spmv_csr &h, const float *x, float *y, hipStream_t s)
{
    if (h.rows == 0) return SPMV_OK;
    switch (h.vector_width) {
        case 2: return launch_vector_g<2>(h, x, y, s);
        case 4: return launch_vector_g<4>(h, x, y, s);
        case 8: return launch_vector_g<8>(h, x, y, s);
        case 16: return launch_vector_g<16>(h, x, y, s);
        case 32: return launch_vector_g<32>(h, x, y, s);
        default:
            set_error("SPMV_VECTOR used before spmv_csr_plan");
            return SPMV_ERR_NOT_PLANNED;
    }
}

// ---------------------------------------------------------------------------
// Structural validation (spmv_csr_validate): grid-stride over rows and elements, first offender by atomicMin.
__global__ __launch_bounds__(256) void k_validate(int64_t rows, int64_t cols, int64_t nnz,
                                                  const int32_t *__restrict__ row_ptr,
                                                  const int32_t *__restrict__ col_idx, int32_t *__restrict__ bad)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t r = t; r < rows; r += stride) {
        const int32_t a = row_ptr[r], b = row_ptr[r + 1];
        if (a > b || a < 0 || (int64_t)b > nnz) atomicMin(&bad[0], (int32_t)r);
    }
    for (int64_t k = t; k < nnz; k += stride) {
        const int32_t c = col_idx[k];
        if (c < 0 || (int64_t)c >= cols) atomicMin(&bad[1], (int32_t)k);
    }
    if (t == 0) {
        if (row_ptr[0] != 0) bad[2] = row_ptr[0];
        if ((int64_t)row_ptr[rows] != nnz) bad[3] = row_ptr[rows];
    }
}

int launch_validate(const spmv_csr *h, int32_t *d_bad4, hipStream_t stream)
{
    const int64_t work = h->nnz > h->rows ? h->nnz : h->rows;
    int64_t blocks = (work + 256 * 8 - 1) / (256 * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 32) blocks = 256 * 32;
    k_validate<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(h->rows, h->cols, h->nnz, h->d_row_ptr,
                                                                   h->d_col_idx, d_bad4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_validate launch", __FILE__, __LINE__);
    return SPMV_OK;
}

// ---------------------------------------------------------------------------
// spmv_csr_column_range: grid-stride min / max over col_idx
__global__ __launch_bounds__(256) void k_column_range(int64_t nnz, const int32_t *__restrict__ col_idx, int32_t *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int mn = 0x7fffffff, mx = -1;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride) {
        const int c = col_idx[k];
        mn = c < mn ? c : mn;
        mx = c > mx ? c : mx;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const int a = __shfl_down(mn, o, kWave), b = __shfl_down(mx, o, kWave);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & (kWave - 1)) == 0 && mx >= 0) {
        atomicMin(&out[0], mn);
        atomicMax(&out[1], mx);
    }
}

int launch_column_range(const spmv_csr &h, int32_t *d_out2, hipStream_t s)
{
    const int32_t init[2] = {0x7fffffff, -1};
    SPMV_HIP_TRY(hipMemcpyAsync(d_out2, init, sizeof init, hipMemcpyHostToDevice, s));
    if (h.nnz > 0) {
        int64_t blocks = (h.nnz + 256 * 16 - 1) / (256 * 16);
        if (blocks > 256 * 16) blocks = 256 * 16;
        k_column_range<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(h.nnz, h.d_col_idx, d_out2);
        return check_launch("k_column_range");
    }
    return SPMV_OK;
}

// ---------------------------------------------------------------------------
// Checksum of vals for the stale-plan guard (SPMV_CHECK_VALUES=1): position-mixed, so a permutation changes it too.
__global__ __launch_bounds__(256) void k_values_checksum(int64_t nnz, const float *__restrict__ vals,
                                                         unsigned long long *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0ull;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride) {
        unsigned long long v = (unsigned long long)__float_as_uint(vals[k]) + 0x9e3779b97f4a7c15ull * (unsigned long long)(k + 1);
        v ^= v >> 30; v *= 0xbf58476d1ce4e5b9ull; v ^= v >> 27; v *= 0x94d049bb133111ebull; v ^= v >> 31;
        acc += v;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(out, acc);
}

int values_checksum(const spmv_csr &h, hipStream_t s, uint64_t *out)
{
    DevPtr<unsigned long long> d;
    SPMV_HIP_TRY(d.alloc(1));
    SPMV_HIP_TRY(hipMemsetAsync(d.p, 0, sizeof(unsigned long long), s));
    if (h.nnz > 0) {
        int64_t blocks = (h.nnz + 256 * 16 - 1) / (256 * 16);
        if (blocks > 256 * 16) blocks = 256 * 16;
        k_values_checksum<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(h.nnz, h.d_vals, d.p);
        if (int rc = check_launch("k_values_checksum")) return rc;
    }
    unsigned long long v = 0;
    SPMV_HIP_TRY(hipMemcpyAsync(&v, d.p, sizeof v, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));
    *out = (uint64_t)v;
    return SPMV_OK;
}

bool check_values_env()
{
    static const bool on = [] { const char *e = getenv("SPMV_CHECK_VALUES"); return e && atoi(e) != 0; }();
    return on;
}

int stamp_values(const spmv_csr &h, hipStream_t s, ValuesStamp &st)
{
    st.gen = h.values_gen;
    st.have_sum = false;
    if (!check_values_env()) return SPMV_OK;
    if (int rc = values_checksum(h, s, &st.sum)) return rc;
    st.have_sum = true;
    return SPMV_OK;
}

int require_fresh_values(const spmv_csr &h, const ValuesStamp &st, hipStream_t s, const char *variant)
{
    if (st.gen != h.values_gen) {
        set_error("spmv_csr_run(%s): the plan holds a copy of vals taken before spmv_csr_values_changed; re-plan "
                  "(spmv_csr_plan rebuilds a stale plan, spmv_csr_plan_set always rebuilds)", variant);
        return SPMV_ERR_STALE_PLAN;
    }
    if (st.have_sum && check_values_env()) {
        uint64_t now = 0;
        if (int rc = values_checksum(h, s, &now)) return rc;
        if (now != st.sum) {
            set_error("spmv_csr_run(%s): SPMV_CHECK_VALUES: vals changed since the plan copied them (checksum %016llx, "
                      "planned with %016llx) and spmv_csr_values_changed was not called; re-plan", variant,
                      (unsigned long long)now, (unsigned long long)st.sum);
            return SPMV_ERR_STALE_PLAN;
        }
    }
    return SPMV_OK;
}

}  // namespace spmv
