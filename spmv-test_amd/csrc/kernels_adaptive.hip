// kernels_adaptive.hip -- nnz-balanced CSR SpMV with LDS staging (SPMV_ADAPTIVE / SPMV_TILED).
//
// Role of the reference's adaptive family -- awsp_kernel_v0/1/2
// (/root/reference/src/kernels/awsp.cu:5-317), awsp_ref_kernel (awsp_ref.cu:6-185) -- and of its
// LDS-tiled kernels csr_tiling_kernel (csr_tiling.cu:24-114) / wsp_sm_kernel (wsp_sm.cu:6-211).
// Those walk 32x32 bitmap blocks with 32-lane warps; nothing of that survives here.  Re-derived
// for CSR on CDNA4:
//
//   * Work is split by NONZEROS, not rows: workgroup c streams nonzeros [c*T, (c+1)*T),
//     T = 16 per lane, with 16-byte-per-lane non-temporal loads of col_idx and vals (each wave
//     instruction covers 1 KiB contiguous).  Row lengths never unbalance the HBM stream.
//   * The T products vals[k]*x[col_idx[k]] are staged in LDS (padded one word per 32 so that
//     lane-per-row reads at power-of-two strides spread over the banks).
//   * Rows are then reduced adaptively inside the chunk: segments of <= kShortSeg products are
//     summed by one lane (sequential -> same order as the CPU oracle); longer ones are queued in
//     LDS with their bounds and summed by 16-lane groups with a __shfl_down tree.
//   * A row that crosses chunk boundaries is finished deterministically: every chunk stores the
//     partial sum of the row it inherits in carry[c]; a tiny second kernel adds the carries to
//     the owner's partial in chunk order.  No float atomics (MI355X_MICROARCH.md: atomics run
//     memory-side at ~1.3 TB/s and are order-nondeterministic).
//   * Blocks are dealt round-robin over the 8 XCDs, so blockIdx is remapped to give every XCD one
//     contiguous range of chunks: neighbouring chunks gather from neighbouring parts of x and
//     share that XCD's 4 MiB L2.
//   * TILED stages the chunk's column window of x in LDS and gathers from LDS instead of L1/L2
//     (a 4-byte gather that misses L1 costs a whole L2 request and a 128-byte line).  The window
//     lives in the SAME LDS region the products are written to afterwards, so it costs no
//     occupancy.  The region grows with the workgroup: 256/512/1024 threads hold 4608/9216/18432
//     floats at the same LDS bytes per wave; the plan picks the smallest workgroup whose region
//     covers the windows of >= 90 % of the chunks, and a chunk whose window does not fit
//     gathers from global memory.
#include <cstdlib>
#include "spmv_internal.hpp"

namespace spmv {

using i4 = int __attribute__((ext_vector_type(4)));
using f4 = float __attribute__((ext_vector_type(4)));

constexpr int kGroup = 16;  // lanes that share one long segment

__host__ __device__ constexpr int chunk_of(int block) { return block * kNnzPerThread; }
__host__ __device__ constexpr int prod_words(int block) { return chunk_of(block) + (chunk_of(block) >> 5); }
// LDS region per workgroup: the padded product buffer (16.5 B/thread... 4224 floats per 256
// threads) rounded up to what still lets 2048 threads share a CU's 160 KiB next to the
// long-segment queue: 18 KiB per 256 threads = 4608 / 9216 / 18432 floats.
__host__ __device__ constexpr int region_words(int block) { return block * 18; }
static_assert(region_words(256) >= prod_words(256), "region must hold the products");
__host__ __device__ constexpr int max_long(int block) { return chunk_of(block) / (kShortSeg + 1) + 2; }

__device__ __forceinline__ int pad_idx(int i) { return i + (i >> 5); }

// chunk handled by this block: XCD j = blockIdx % 8 gets a contiguous range
__device__ __forceinline__ int xcd_chunk(int bid, int n)
{
    const int q = n / kXcds, rem = n % kXcds;
    const int j = bid % kXcds, idx = bid / kXcds;
    return j * q + (j < rem ? j : rem) + idx;
}

// ---------------------------------------------------------------------------
// plan: lb[c] = first row r with row_ptr[r] >= c*chunk   (c = 0..nchunks-1), lb[nchunks] = rows
__global__ void k_plan_chunks(int64_t rows, int nchunks, int chunk, const int32_t *__restrict__ row_ptr,
                              int32_t *__restrict__ lb)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > nchunks) return;
    if (c == nchunks) { lb[c] = (int32_t)rows; return; }
    const int64_t key = (int64_t)c * chunk;
    int64_t lo = 0, hi = rows;  // row_ptr[rows] = nnz > key, so the answer is <= rows
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)row_ptr[mid] < key) lo = mid + 1; else hi = mid;
    }
    lb[c] = (int32_t)lo;
}

// plan (TILED): column window of every chunk -> win[2c] = first column (aligned down to 4),
// win[2c+1] = window length in floats, 0 when wider than `cap`; stats[0] = widest staged window,
// stats[1] = number of chunks whose window fits.
__global__ __launch_bounds__(256) void k_plan_windows(int64_t nnz, int nchunks, int chunk,
                                                      const int32_t *__restrict__ col_idx,
                                                      int32_t *__restrict__ win, int32_t *__restrict__ stats, int cap)
{
    __shared__ int s_min[4], s_max[4];
    const int c = blockIdx.x;
    const int64_t base = (int64_t)c * chunk;
    const int n = (int)((nnz - base) < chunk ? (nnz - base) : chunk);
    int mn = 0x7fffffff, mx = -1;
    for (int i = threadIdx.x; i < n; i += 256) {
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
        for (int w = 1; w < 4; ++w) {
            mn = s_min[w] < mn ? s_min[w] : mn;
            mx = s_max[w] > mx ? s_max[w] : mx;
        }
        const int w0 = mn & ~3;
        const int64_t len = (int64_t)mx + 1 - w0;
        const int32_t wl = (mx >= 0 && len <= cap) ? (int32_t)len : 0;
        win[2 * c] = w0;
        win[2 * c + 1] = wl;
        if (wl > 0) {
            atomicMax(&stats[0], wl);
            atomicAdd(&stats[1], 1);
        }
    }
}

// ---------------------------------------------------------------------------
// Segment t of chunk c: t = 0 is the head (the row inherited from chunk c-1, summed into
// carry[c]); t = 1..m are the rows that START in this chunk (summed into y).
struct Segment {
    int s, e;
    float *dst;
};

__device__ __forceinline__ Segment make_segment(int t, int lb0, int64_t base, int64_t lim, int64_t rb, int64_t re,
                                                float *__restrict__ y, float *__restrict__ carry, int c)
{
    Segment g;
    if (t == 0) {  // rb = row_ptr[lb0]
        g.s = 0;
        g.e = (int)((rb < lim ? rb : lim) - base);
        g.dst = carry + c;
    } else {       // rb = row_ptr[r], re = row_ptr[r+1], r = lb0 + t - 1
        g.s = (int)(rb - base);
        g.e = (int)((re < lim ? re : lim) - base);
        g.dst = y + ((int64_t)lb0 + t - 1);
    }
    return g;
}

// BLOCK/256 workgroups of BLOCK threads fill a CU to the same LDS bytes and waves; the second
// launch-bounds argument keeps the kernel at <= 64 VGPRs so that they all fit (8 waves/SIMD).
template <int BLOCK, bool TILED>
__global__ __launch_bounds__(BLOCK, 8) void k_adaptive(int64_t rows, int64_t nnz, int64_t cols, int nchunks,
                                                       const int32_t *__restrict__ row_ptr,
                                                       const int32_t *__restrict__ col_idx,
                                                       const float *__restrict__ vals,
                                                       const float *__restrict__ x, float *__restrict__ y,
                                                       const int32_t *__restrict__ chunk_lb,
                                                       float *__restrict__ carry,
                                                       const int32_t *__restrict__ win)
{
    constexpr int kChunkT = chunk_of(BLOCK);
    // one dynamic LDS region, used twice: first as the x window (TILED), then -- after the
    // gathers have landed in registers -- as the product staging buffer.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int2 long_seg[max_long(BLOCK)];  // {segment id, first product | end product << 16}
    __shared__ int long_count;

    const int tid = threadIdx.x;
    const int c = xcd_chunk(blockIdx.x, nchunks);
    const int64_t base = (int64_t)c * kChunkT;
    const int n = (int)((nnz - base) < kChunkT ? (nnz - base) : kChunkT);
    const int64_t lim = base + n;
    if (tid == 0) long_count = 0;

    // chunk metadata first: the row_ptr prefetch below depends on it
    const int lb0 = chunk_lb[c], lb1 = chunk_lb[c + 1];
    const int m = lb1 - lb0;
    int w0 = 0, wlen = 0;
    if (TILED) {
        w0 = win[2 * c];
        wlen = win[2 * c + 1];
    }

    // ---- 1. stream the chunk: 16 B per lane per load (1 KiB contiguous per wave instruction),
    //         non-temporal (read once), all loads issued before the first use
    constexpr int kVec = kNnzPerThread / 4;
    i4 cc[kVec];
    f4 vv[kVec];
    const bool full = (n == kChunkT);
    if (full) {
        const i4 *c4 = reinterpret_cast<const i4 *>(col_idx + base);
        const f4 *v4 = reinterpret_cast<const f4 *>(vals + base);
#pragma unroll
        for (int j = 0; j < kVec; ++j) {
            cc[j] = __builtin_nontemporal_load(&c4[j * BLOCK + tid]);
            vv[j] = __builtin_nontemporal_load(&v4[j * BLOCK + tid]);
        }
    } else {
        // last chunk: guarded scalar loads, same element->lane map; padded lanes use an in-window
        // column and value 0 (their products are never read)
#pragma unroll
        for (int j = 0; j < kVec; ++j) {
            const int i0 = (j * BLOCK + tid) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = (i0 + q) < n;
                cc[j][q] = ok ? col_idx[base + i0 + q] : w0;
                vv[j][q] = ok ? vals[base + i0 + q] : 0.0f;
            }
        }
    }

    // ---- 2. row pointers of this lane's first segment, in flight together with the stream
    int64_t rb0 = 0, re0 = 0;
    if (tid <= m) {
        rb0 = row_ptr[tid == 0 ? lb0 : lb0 + tid - 1];  // lb0 <= rows and row_ptr[rows] = nnz
        re0 = tid == 0 ? 0 : row_ptr[lb0 + tid];
    }

    // ---- 3. gather x: from the LDS window when the plan found one, else through L1/L2
    f4 xv[kVec];
    if (TILED && wlen > 0) {
        for (int i = tid * 4; i < wlen; i += BLOCK * 4) {
            if ((int64_t)w0 + i + 3 < cols) {
                *reinterpret_cast<f4 *>(smem + i) = *reinterpret_cast<const f4 *>(x + w0 + i);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((int64_t)w0 + i + q < cols) smem[i + q] = x[w0 + i + q];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kVec; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[j][q] = smem[cc[j][q] - w0];
        __syncthreads();  // every gather has its value before the window is overwritten
    } else {
#pragma unroll
        for (int j = 0; j < kVec; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[j][q] = x[cc[j][q]];
    }

    // ---- 4. stage the products (padded one word per 32: lane-per-row reads spread over banks)
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
        const int p0 = pad_idx((j * BLOCK + tid) * 4);  // i0 % 4 == 0: never straddles a pad slot
        smem[p0] = vv[j][0] * xv[j][0];
        smem[p0 + 1] = vv[j][1] * xv[j][1];
        smem[p0 + 2] = vv[j][2] * xv[j][2];
        smem[p0 + 3] = vv[j][3] * xv[j][3];
    }
    __syncthreads();

    // ---- 5. short segments: one lane each, sequential (the oracle's order); longer ones are
    //         queued in LDS with their bounds so that phase 6 touches no global metadata
    for (int t = tid; t <= m; t += BLOCK) {
        int64_t rb, re;
        if (t == tid) { rb = rb0; re = re0; }
        else { rb = row_ptr[lb0 + t - 1]; re = row_ptr[lb0 + t]; }
        const Segment g = make_segment(t, lb0, base, lim, rb, re, y, carry, c);
        if (g.e - g.s <= kShortSeg) {
            float acc = 0.0f;
            int i = g.s;
            for (; i + 3 < g.e; i += 4) {
                const float a0 = smem[pad_idx(i)], a1 = smem[pad_idx(i + 1)], a2 = smem[pad_idx(i + 2)],
                            a3 = smem[pad_idx(i + 3)];
                acc = (((acc + a0) + a1) + a2) + a3;
            }
            for (; i < g.e; ++i) acc += smem[pad_idx(i)];
            *g.dst = acc;
        } else {
            const int slot = atomicAdd(&long_count, 1);
            long_seg[slot] = make_int2(t, g.s | (g.e << 16));  // s < 2^14, e <= 2^14
        }
    }
    __syncthreads();

    // ---- 6. long segments: one kGroup-lane group each (BLOCK/kGroup in parallel), lanes stride
    //         the segment, __shfl_down tree inside the group
    const int nlong = long_count;
    const int sub = tid & (kGroup - 1);
    for (int i = tid / kGroup; i < nlong; i += BLOCK / kGroup) {
        const int2 q = long_seg[i];
        const int qe = (int)((unsigned)q.y >> 16);
        float acc = 0.0f;
        for (int k = (q.y & 0xffff) + sub; k < qe; k += kGroup) acc += smem[pad_idx(k)];
#pragma unroll
        for (int o = kGroup / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, kGroup);
        if (sub == 0) {
            if (q.x == 0) carry[c] = acc;
            else y[(int64_t)lb0 + q.x - 1] = acc;
        }
    }
}

// rows that continue past their owner chunk: y[r] += carry[c+1] + carry[c+2] + ... in chunk order
__global__ void k_carry_fixup(int nchunks, int chunk, const int32_t *__restrict__ row_ptr,
                              const int32_t *__restrict__ chunk_lb, const float *__restrict__ carry,
                              float *__restrict__ y)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks - 1) return;
    const int lb0 = chunk_lb[c], lb1 = chunk_lb[c + 1];
    if (lb1 == lb0) return;  // no row starts in this chunk
    const int64_t r = (int64_t)lb1 - 1;
    const int64_t endp = row_ptr[r + 1];
    if (endp <= (int64_t)(c + 1) * chunk) return;
    float s = y[r];
    for (int64_t c2 = c + 1; c2 * chunk < endp; ++c2) s += carry[c2];
    y[r] = s;
}

// ---------------------------------------------------------------------------
static int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
    return SPMV_OK;
}

static void free_plan(ChunkPlan &p)
{
    if (p.d_lb) (void)hipFree(p.d_lb);
    if (p.d_carry) (void)hipFree(p.d_carry);
    if (p.d_win) (void)hipFree(p.d_win);
    p = ChunkPlan();
}

// chunk boundaries (+ column windows when `cap` > 0) for workgroups of `block` threads
static int build_plan(const spmv_csr &h, int block, int cap, hipStream_t s, ChunkPlan &p, int *fits)
{
    free_plan(p);
    p.block = block;
    const int chunk = chunk_of(block);
    p.nchunks = (int)((h.nnz + chunk - 1) / chunk);
    if (fits) *fits = 0;
    if (p.nchunks == 0) return SPMV_OK;
    SPMV_HIP_TRY(hipMalloc((void **)&p.d_lb, sizeof(int32_t) * ((size_t)p.nchunks + 1)));
    SPMV_HIP_TRY(hipMalloc((void **)&p.d_carry, sizeof(float) * (size_t)p.nchunks));
    hipLaunchKernelGGL(k_plan_chunks, dim3((p.nchunks + 1 + 255) / 256), dim3(256), 0, s, h.rows, p.nchunks, chunk,
                       h.d_row_ptr, p.d_lb);
    int rc = check_launch("k_plan_chunks");
    if (rc) return rc;
    if (cap > 0) {
        // [2*nchunks] windows + 2 words of statistics
        SPMV_HIP_TRY(hipMalloc((void **)&p.d_win, sizeof(int32_t) * (2 * (size_t)p.nchunks + 2)));
        int32_t *d_stats = p.d_win + 2 * (size_t)p.nchunks;
        SPMV_HIP_TRY(hipMemsetAsync(d_stats, 0, 2 * sizeof(int32_t), s));
        hipLaunchKernelGGL(k_plan_windows, dim3(p.nchunks), dim3(256), 0, s, h.nnz, p.nchunks, chunk, h.d_col_idx,
                           p.d_win, d_stats, cap);
        if ((rc = check_launch("k_plan_windows"))) return rc;
        int32_t stats[2] = {0, 0};
        SPMV_HIP_TRY(hipMemcpyAsync(stats, d_stats, sizeof stats, hipMemcpyDeviceToHost, s));
        SPMV_HIP_TRY(hipStreamSynchronize(s));
        p.window_max = stats[0];
        if (fits) *fits = stats[1];
    }
    return SPMV_OK;
}

int plan_adaptive(spmv_csr &h, bool tiled, hipStream_t s)
{
    if (!tiled) {
        if (h.plan_adaptive.block) return SPMV_OK;
        return build_plan(h, 256, 0, s, h.plan_adaptive, nullptr);
    }
    if (h.plan_tiled.block) return SPMV_OK;
    // smallest workgroup whose LDS region holds the x window of >= 90 % of the chunks; if even
    // 1024 threads do not reach 50 %, the matrix has no usable column locality: stay at 256.
    int forced = 0;
    if (const char *e = getenv("SPMV_TILED_BLOCK")) forced = atoi(e);  // tuning knob: 256 | 512 | 1024
    if (forced != 256 && forced != 512 && forced != 1024) forced = 0;
    const int cands[3] = {256, 512, 1024};
    for (int k = 0; k < 3; ++k) {
        const int block = cands[k];
        if (forced && block != forced) continue;
        int fits = 0;
        int rc = build_plan(h, block, region_words(block), s, h.plan_tiled, &fits);
        if (rc) return rc;
        if (forced || h.plan_tiled.nchunks == 0 || fits >= 0.9 * h.plan_tiled.nchunks) return SPMV_OK;
        if (block == 1024 && fits >= 0.5 * h.plan_tiled.nchunks) return SPMV_OK;
    }
    int fits = 0;
    return build_plan(h, 256, region_words(256), s, h.plan_tiled, &fits);
}

void destroy_plans(spmv_csr &h)
{
    free_plan(h.plan_adaptive);
    free_plan(h.plan_tiled);
}

template <int BLOCK, bool TILED>
static int launch_block(const spmv_csr &h, const ChunkPlan &p, const float *x, float *y, hipStream_t s)
{
    // dynamic LDS: the product buffer; a staged x window is never wider (plan cap = region)
    const size_t lds = sizeof(float) * (size_t)region_words(BLOCK);
    static bool attr_set = false;  // > 64 KiB of dynamic LDS needs the opt-in (BLOCK = 1024)
    if (!attr_set) {
        SPMV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_adaptive<BLOCK, TILED>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_adaptive<BLOCK, TILED>), dim3(p.nchunks), dim3(BLOCK), lds, s, h.rows, h.nnz, h.cols,
                       p.nchunks, h.d_row_ptr, h.d_col_idx, h.d_vals, x, y, p.d_lb, p.d_carry, p.d_win);
    return check_launch("k_adaptive");
}

int launch_adaptive(const spmv_csr &h, const float *x, float *y, bool tiled, hipStream_t s)
{
    const ChunkPlan &p = tiled ? h.plan_tiled : h.plan_adaptive;
    if (!p.block) {
        set_error("%s used before spmv_csr_plan", tiled ? "SPMV_TILED" : "SPMV_ADAPTIVE");
        return SPMV_ERR_NOT_PLANNED;
    }
    if (h.rows == 0) return SPMV_OK;
    if (p.nchunks == 0) {  // no nonzeros: y = 0
        SPMV_HIP_TRY(hipMemsetAsync(y, 0, sizeof(float) * (size_t)h.rows, s));
        return SPMV_OK;
    }
    int rc;
    if (!tiled) rc = launch_block<256, false>(h, p, x, y, s);
    else if (p.block == 256) rc = launch_block<256, true>(h, p, x, y, s);
    else if (p.block == 512) rc = launch_block<512, true>(h, p, x, y, s);
    else rc = launch_block<1024, true>(h, p, x, y, s);
    if (rc) return rc;
    if (p.nchunks > 1) {
        hipLaunchKernelGGL(k_carry_fixup, dim3((p.nchunks - 1 + 255) / 256), dim3(256), 0, s, p.nchunks,
                           chunk_of(p.block), h.d_row_ptr, p.d_lb, p.d_carry, y);
        rc = check_launch("k_carry_fixup");
    }
    return rc;
}

}  // namespace spmv
