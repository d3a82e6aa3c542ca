// kernels_adaptive.hip -- nnz-balanced CSR SpMV with LDS staging (SPMV_ADAPTIVE / SPMV_TILED).
//
// Role of the reference's adaptive family -- awsp_kernel_v0/1/2
// (/root/reference/src/kernels/awsp.cu:5-317), awsp_ref_kernel (awsp_ref.cu:6-185) -- and of its
// LDS-tiled kernels csr_tiling_kernel (csr_tiling.cu:24-114) / wsp_sm_kernel (wsp_sm.cu:6-211).
// Those walk 32x32 bitmap blocks with 32-lane warps; nothing of that survives here.  Re-derived
// for CSR on CDNA4:
//
//   * Work is split by NONZEROS, not rows: workgroup c streams nonzeros [c*T, (c+1)*T) with
//     16-byte-per-lane fully coalesced loads of col_idx and vals (each wave instruction covers
//     1 KiB contiguous).  Row lengths never unbalance the HBM stream.
//   * The T products vals[k]*x[col_idx[k]] are staged in LDS (padded one word per 32 so that
//     lane-per-row reads at power-of-two strides spread over the banks).
//   * Rows are then reduced adaptively inside the chunk: segments of <= kShortSeg products are
//     summed by one lane (sequential -> same order as the CPU oracle), longer ones by a whole
//     64-lane wavefront with a __shfl_down tree.
//   * A row that crosses chunk boundaries is finished deterministically: every chunk stores the
//     partial sum of the row it inherits in carry[c]; a tiny second kernel adds the carries to
//     the owner's partial in chunk order.  No float atomics (MI355X_MICROARCH.md: atomics run
//     memory-side at ~1.3 TB/s and are order-nondeterministic).
//   * Blocks are dealt round-robin over the 8 XCDs, so blockIdx is remapped to give every XCD one
//     contiguous range of chunks: neighbouring chunks gather from neighbouring parts of x and
//     share that XCD's 4 MiB L2.
//   * TILED additionally stages the chunk's column window of x in LDS (when the plan found it
//     <= kTileMaxCols wide) and gathers from LDS instead of L1/L2.
#include "spmv_internal.hpp"

namespace spmv {

__device__ __forceinline__ int pad_idx(int i) { return i + (i >> 5); }
constexpr int kProdWords = kChunk + (kChunk >> 5);
constexpr int kMaxLong = kChunk / (kShortSeg + 1) + 2;  // segments longer than kShortSeg per chunk

// chunk handled by this block: XCD j = blockIdx % 8 gets a contiguous range
__device__ __forceinline__ int xcd_chunk(int bid, int n)
{
    const int q = n / kXcds, rem = n % kXcds;
    const int j = bid % kXcds, idx = bid / kXcds;
    return j * q + (j < rem ? j : rem) + idx;
}

// ---------------------------------------------------------------------------
// plan: chunk_lb[c] = first row r with row_ptr[r] >= c*T   (c = 0..nchunks-1), chunk_lb[nchunks] = rows
__global__ void k_plan_chunks(int64_t rows, int nchunks, const int32_t *__restrict__ row_ptr,
                              int32_t *__restrict__ chunk_lb)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > nchunks) return;
    if (c == nchunks) { chunk_lb[c] = (int32_t)rows; return; }
    const int64_t key = (int64_t)c * kChunk;
    int64_t lo = 0, hi = rows;  // row_ptr[rows] = nnz > key, so the answer is <= rows
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)row_ptr[mid] < key) lo = mid + 1; else hi = mid;
    }
    chunk_lb[c] = (int32_t)lo;
}

// plan (TILED): column window of every chunk -> win[2c] = first column (aligned down to 4),
// win[2c+1] = window length in floats, 0 when wider than kTileMaxCols.
__global__ __launch_bounds__(kBlock) void k_plan_windows(int64_t nnz, int64_t cols, int nchunks,
                                                         const int32_t *__restrict__ col_idx,
                                                         int32_t *__restrict__ win)
{
    __shared__ int s_min[kBlock / kWave], s_max[kBlock / kWave];
    const int c = blockIdx.x;
    const int64_t base = (int64_t)c * kChunk;
    const int n = (int)((nnz - base) < kChunk ? (nnz - base) : kChunk);
    int mn = 0x7fffffff, mx = -1;
    for (int i = threadIdx.x; i < n; i += kBlock) {
        int v = col_idx[base + i];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        int a = __shfl_down(mn, o, kWave), b = __shfl_down(mx, o, kWave);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { s_min[threadIdx.x >> 6] = mn; s_max[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / kWave; ++w) {
            mn = s_min[w] < mn ? s_min[w] : mn;
            mx = s_max[w] > mx ? s_max[w] : mx;
        }
        int w0 = mn & ~3;
        int64_t len = (int64_t)mx + 1 - w0;
        win[2 * c] = w0;
        win[2 * c + 1] = (mx >= 0 && len <= kTileMaxCols) ? (int32_t)len : 0;
    }
}

// ---------------------------------------------------------------------------
template <bool TILED>
__global__ __launch_bounds__(kBlock) void k_adaptive(int64_t rows, int64_t nnz, int64_t cols, int nchunks,
                                                     const int32_t *__restrict__ row_ptr,
                                                     const int32_t *__restrict__ col_idx,
                                                     const float *__restrict__ vals,
                                                     const float *__restrict__ x, float *__restrict__ y,
                                                     const int32_t *__restrict__ chunk_lb,
                                                     float *__restrict__ carry,
                                                     const int32_t *__restrict__ win)
{
    __shared__ float prod[kProdWords];
    __shared__ int long_list[kMaxLong];
    __shared__ int long_count;
    __shared__ float xt[TILED ? kTileMaxCols : 1];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int c = xcd_chunk(blockIdx.x, nchunks);
    const int64_t base = (int64_t)c * kChunk;
    const int n = (int)((nnz - base) < kChunk ? (nnz - base) : kChunk);
    if (tid == 0) long_count = 0;
    int w0 = 0, wlen = 0;
    if (TILED) {
        w0 = win[2 * c];
        wlen = win[2 * c + 1];
    }

    // ---- 1. stream the chunk: 16 B per lane per load, all loads issued before first use
    constexpr int kVec = kNnzPerThread / 4;
    int4 cc[kVec];
    float4 vv[kVec];
    const bool full = (n == kChunk);
    if (full) {
        const int4 *c4 = reinterpret_cast<const int4 *>(col_idx + base);
        const float4 *v4 = reinterpret_cast<const float4 *>(vals + base);
#pragma unroll
        for (int j = 0; j < kVec; ++j) {
            cc[j] = c4[j * kBlock + tid];
            vv[j] = v4[j * kBlock + tid];
        }
    } else {
        // last chunk: guarded scalar loads, same element->lane map
#pragma unroll
        for (int j = 0; j < kVec; ++j) {
            const int i0 = (j * kBlock + tid) * 4;
            int ci[4];
            float vi[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = (i0 + q) < n;
                ci[q] = ok ? col_idx[base + i0 + q] : w0;  // padded lanes: any in-window column
                vi[q] = ok ? vals[base + i0 + q] : 0.0f;
            }
            cc[j] = make_int4(ci[0], ci[1], ci[2], ci[3]);
            vv[j] = make_float4(vi[0], vi[1], vi[2], vi[3]);
        }
    }

    // ---- 2. gather x (from the LDS window when TILED found one) and stage the products
    if (TILED) {
        for (int i = tid * 4; i < wlen; i += kBlock * 4) {
            if ((int64_t)w0 + i + 3 < cols) {
                float4 t = *reinterpret_cast<const float4 *>(x + w0 + i);
                xt[i] = t.x; xt[i + 1] = t.y; xt[i + 2] = t.z; xt[i + 3] = t.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((int64_t)w0 + i + q < cols && i + q < kTileMaxCols) xt[i + q] = x[w0 + i + q];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
        float4 xv;
        if (TILED && wlen > 0) {
            xv.x = xt[cc[j].x - w0]; xv.y = xt[cc[j].y - w0];
            xv.z = xt[cc[j].z - w0]; xv.w = xt[cc[j].w - w0];
        } else {
            xv.x = x[cc[j].x]; xv.y = x[cc[j].y]; xv.z = x[cc[j].z]; xv.w = x[cc[j].w];
        }
        const int i0 = (j * kBlock + tid) * 4;
        const int p0 = pad_idx(i0);  // i0 % 4 == 0: the four words never straddle a pad slot
        prod[p0] = vv[j].x * xv.x;
        prod[p0 + 1] = vv[j].y * xv.y;
        prod[p0 + 2] = vv[j].z * xv.z;
        prod[p0 + 3] = vv[j].w * xv.w;
    }
    __syncthreads();

    // ---- 3. segments of this chunk: t = 0 is the head (row inherited from chunk c-1, goes to
    //         carry[c]); t = 1..m are the rows that START in this chunk (go to y).
    const int lb0 = chunk_lb[c], lb1 = chunk_lb[c + 1];
    const int m = lb1 - lb0;
    const int64_t lim = base + n;
    for (int t = tid; t <= m; t += kBlock) {
        int s, e;
        float *dst;
        if (t == 0) {
            int64_t hp = row_ptr[lb0];  // lb0 <= rows and row_ptr[rows] = nnz
            s = 0;
            e = (int)((hp < lim ? hp : lim) - base);
            dst = carry + c;
        } else {
            const int64_t r = (int64_t)lb0 + t - 1;
            int64_t rb = row_ptr[r], re = row_ptr[r + 1];
            s = (int)(rb - base);
            e = (int)((re < lim ? re : lim) - base);
            dst = y + r;
        }
        if (e - s <= kShortSeg) {
            float acc = 0.0f;
            for (int i = s; i < e; ++i) acc += prod[pad_idx(i)];
            *dst = acc;
        } else {
            int slot = atomicAdd(&long_count, 1);
            long_list[slot] = t;
        }
    }
    __syncthreads();

    // ---- 4. long segments: one wavefront each
    const int nlong = long_count;
    for (int i = tid >> 6; i < nlong; i += kBlock / kWave) {
        const int t = long_list[i];
        int s, e;
        float *dst;
        if (t == 0) {
            int64_t hp = row_ptr[lb0];
            s = 0;
            e = (int)((hp < lim ? hp : lim) - base);
            dst = carry + c;
        } else {
            const int64_t r = (int64_t)lb0 + t - 1;
            int64_t rb = row_ptr[r], re = row_ptr[r + 1];
            s = (int)(rb - base);
            e = (int)((re < lim ? re : lim) - base);
            dst = y + r;
        }
        float acc = 0.0f;
        for (int k = s + lane; k < e; k += kWave) acc += prod[pad_idx(k)];
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, kWave);
        if (lane == 0) *dst = acc;
    }
}

// rows that continue past their owner chunk: y[r] += carry[c+1] + carry[c+2] + ... in chunk order
__global__ void k_carry_fixup(int nchunks, const int32_t *__restrict__ row_ptr,
                              const int32_t *__restrict__ chunk_lb, const float *__restrict__ carry,
                              float *__restrict__ y)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks - 1) return;
    const int lb0 = chunk_lb[c], lb1 = chunk_lb[c + 1];
    if (lb1 == lb0) return;  // no row starts in this chunk
    const int64_t r = (int64_t)lb1 - 1;
    const int64_t endp = row_ptr[r + 1];
    if (endp <= (int64_t)(c + 1) * kChunk) return;
    float s = y[r];
    for (int64_t c2 = c + 1; c2 * kChunk < endp; ++c2) s += carry[c2];
    y[r] = s;
}

// ---------------------------------------------------------------------------
static int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
    return SPMV_OK;
}

int plan_adaptive(spmv_csr &h, bool tiled, hipStream_t s)
{
    if (!h.planned_adaptive) {
        h.nchunks = (int)((h.nnz + kChunk - 1) / kChunk);
        if (h.nchunks > 0) {
            SPMV_HIP_TRY(hipMalloc((void **)&h.d_chunk_lb, sizeof(int32_t) * ((size_t)h.nchunks + 1)));
            SPMV_HIP_TRY(hipMalloc((void **)&h.d_carry, sizeof(float) * (size_t)h.nchunks));
            int blocks = (h.nchunks + 1 + kBlock - 1) / kBlock;
            hipLaunchKernelGGL(k_plan_chunks, dim3(blocks), dim3(kBlock), 0, s, h.rows, h.nchunks,
                               h.d_row_ptr, h.d_chunk_lb);
            int rc = check_launch("k_plan_chunks");
            if (rc) return rc;
        }
        h.planned_adaptive = true;
    }
    if (tiled && !h.planned_tiled) {
        if (h.nchunks > 0) {
            SPMV_HIP_TRY(hipMalloc((void **)&h.d_chunk_win, sizeof(int32_t) * 2 * (size_t)h.nchunks));
            hipLaunchKernelGGL(k_plan_windows, dim3(h.nchunks), dim3(kBlock), 0, s, h.nnz, h.cols,
                               h.nchunks, h.d_col_idx, h.d_chunk_win);
            int rc = check_launch("k_plan_windows");
            if (rc) return rc;
        }
        h.planned_tiled = true;
    }
    return SPMV_OK;
}

int launch_adaptive(const spmv_csr &h, const float *x, float *y, bool tiled, hipStream_t s)
{
    if (!h.planned_adaptive || (tiled && !h.planned_tiled)) {
        set_error("%s used before spmv_csr_plan", tiled ? "SPMV_TILED" : "SPMV_ADAPTIVE");
        return SPMV_ERR_NOT_PLANNED;
    }
    if (h.rows == 0) return SPMV_OK;
    if (h.nchunks == 0) {  // no nonzeros: y = 0
        SPMV_HIP_TRY(hipMemsetAsync(y, 0, sizeof(float) * (size_t)h.rows, s));
        return SPMV_OK;
    }
    if (tiled)
        hipLaunchKernelGGL(k_adaptive<true>, dim3(h.nchunks), dim3(kBlock), 0, s, h.rows, h.nnz, h.cols,
                           h.nchunks, h.d_row_ptr, h.d_col_idx, h.d_vals, x, y, h.d_chunk_lb, h.d_carry,
                           h.d_chunk_win);
    else
        hipLaunchKernelGGL(k_adaptive<false>, dim3(h.nchunks), dim3(kBlock), 0, s, h.rows, h.nnz, h.cols,
                           h.nchunks, h.d_row_ptr, h.d_col_idx, h.d_vals, x, y, h.d_chunk_lb, h.d_carry,
                           (const int32_t *)nullptr);
    int rc = check_launch("k_adaptive");
    if (rc) return rc;
    if (h.nchunks > 1) {
        int blocks = (h.nchunks - 1 + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(k_carry_fixup, dim3(blocks), dim3(kBlock), 0, s, h.nchunks, h.d_row_ptr,
                           h.d_chunk_lb, h.d_carry, y);
        rc = check_launch("k_carry_fixup");
    }
    return rc;
}

}  // namespace spmv
