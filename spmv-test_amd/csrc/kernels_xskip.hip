// kernels_xskip.hip -- SPMV_XSKIP: activation sparsity on the CSR side.
//
// The reference's actual idea (asp_kernel_v*, /root/reference/src/kernels/asp.cu:20-26; awsp_kernel_v1,
// awsp.cu:127-134; awsp_ref_kernel, awsp_ref.cu:52): where x[j] is zero, the elements of A that would be multiplied
// by it are never loaded.  A CSR whose rows are the OUTPUTS cannot do that at memory granularity -- the elements
// that meet x[j] are one per row, spread over the whole stream.  So the plan re-orders the matrix once into the
// input-major form (the rows of the dense A: for every input j the list of (output, value) pairs it feeds), cut into
// blocks of 1024 outputs so that the sums of a block live in LDS:
//     entries of (block b, input j) are one contiguous SEGMENT  erow16[k] = output - 1024 b,  evals[k] = value
//     seg_input[s], seg_ptr[s]   the non-empty segments of all blocks, block by block, inputs ascending
// and the multiply walks segments: a wavefront reads x[j] (one scalar), and if it is zero the segment -- 6 bytes per
// nonzero, contiguous -- is skipped whole; otherwise the lanes stride the segment and add x[j]*value into the
// wavefront's PRIVATE copy of the block's sums (plain LDS read-add-write: the outputs of one segment are distinct;
// LDS float atomics run at one lane per three clocks, tools/ubench_lds_atomic.hip, so the four wavefronts of a
// workgroup do not share a copy).  The copies are added in wavefront order, the slabs of a block (its segment list
// is cut into equal parts for parallelism) in slab order by a second small kernel: deterministic, no atomics.
// With the tester's 50 %-zero x (tester.cpp:154) half of the matrix is never read.
//
// Scope: the reference's own regime -- dense-ish matrices.  The plan counts nonzeros per (block, input) in a table
// of blocks x cols integers, so it refuses matrices where that table would exceed 2^27 entries; and the plain
// read-add-write needs rows without duplicate columns: rows must be sorted and duplicate-free (both checked).  The values are COPIED (re-plan
// after changing them), like SPMV_PANEL.
#include "spmv_internal.hpp"

namespace spmv {

namespace {

constexpr int kXR = 1024;           // outputs per block: 4 KiB of sums per wavefront copy (16 KiB per workgroup)
constexpr int kXWaves = 4;          // wavefronts (= private copies) per workgroup
constexpr int kXSlabMax = 64;       // slabs per block: the combine kernel adds that many partials per output
constexpr int64_t kXTableMax = 1ll << 27;

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
    return SPMV_OK;
}

// cnt[b*cols + j] = nonzeros of input j in output block b; dup[0] = 1 when a row holds the same column twice in a
// row, dup[1] = 1 when the columns of a row do not ascend (then a duplicate need not be adjacent: [3, 5, 3] -- two lanes
// of one segment would read-add-write the same LDS word, so the plan refuses unsorted rows outright)
__global__ __launch_bounds__(kBlock) void k_xs_count(int64_t rows, int64_t cols, const int32_t *__restrict__ row_ptr,
                                                     const int32_t *__restrict__ col_idx, int32_t *__restrict__ cnt,
                                                     int32_t *__restrict__ dup)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int64_t b = r / kXR;
    const int32_t s = row_ptr[r], e = row_ptr[r + 1];
    for (int32_t k = s + lane; k < e; k += kWave) {
        const int32_t j = col_idx[k];
        atomicAdd(&cnt[b * cols + j], 1);
        if (k + 1 < e) {
            const int32_t nx = col_idx[k + 1];
            if (nx == j) dup[0] = 1;
            if (nx < j) dup[1] = 1;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_xs_flags(int64_t n, const int32_t *__restrict__ cnt, int32_t *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) flag[i] = cnt[i] > 0 ? 1 : 0;
}

// non-empty (block, input) pairs -> segment list; block_seg[b] = first segment of block b
__global__ __launch_bounds__(kBlock) void k_xs_segments(int64_t nblocks, int64_t cols, const int32_t *__restrict__ cnt,
                                                        const int32_t *__restrict__ pos, const int32_t *__restrict__ segno,
                                                        const int32_t *__restrict__ nseg_total, const int32_t *__restrict__ nnz_total,
                                                        int32_t *__restrict__ seg_input, int32_t *__restrict__ seg_ptr,
                                                        int32_t *__restrict__ block_seg)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) {
        seg_ptr[*nseg_total] = *nnz_total;
        block_seg[nblocks] = *nseg_total;
    }
    if (i >= nblocks * cols) return;
    if (i % cols == 0) block_seg[i / cols] = segno[i];
    if (cnt[i] > 0) {
        seg_input[segno[i]] = (int32_t)(i % cols);
        seg_ptr[segno[i]] = pos[i];
    }
}

// entries into their segments (cursor = a copy of pos, advanced atomically: the order inside a segment is arbitrary --
// its outputs are distinct, and a sum only ever sees the inputs in ascending order)
__global__ __launch_bounds__(kBlock) void k_xs_fill(int64_t rows, int64_t cols, const int32_t *__restrict__ row_ptr,
                                                    const int32_t *__restrict__ col_idx, const float *__restrict__ vals,
                                                    int32_t *__restrict__ cursor, uint16_t *__restrict__ erow,
                                                    float *__restrict__ evals)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int64_t b = r / kXR;
    const uint16_t rl = (uint16_t)(r - b * kXR);
    const int32_t s = row_ptr[r], e = row_ptr[r + 1];
    for (int32_t k = s + lane; k < e; k += kWave) {
        const int32_t d = atomicAdd(&cursor[b * cols + col_idx[k]], 1);
        erow[d] = rl;
        evals[d] = vals[k];
    }
}

// ---- the multiply ---------------------------------------------------------------------------------------------------
// workgroup = (block b, slab t of its segments); wavefront w takes segments t0 + w, t0 + w + 4, ...
__global__ __launch_bounds__(kXWaves *kWave) void k_xskip(int64_t rows, int slabs, const int32_t *__restrict__ block_seg,
                                                          const int32_t *__restrict__ seg_input,
                                                          const int32_t *__restrict__ seg_ptr,
                                                          const uint16_t *__restrict__ erow,
                                                          const float *__restrict__ evals, const float *__restrict__ x,
                                                          float *__restrict__ out)
{
    __shared__ float ys[kXWaves][kXR];
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / slabs, t = blockIdx.x % slabs;
    const int sb0 = block_seg[b], sb1 = block_seg[b + 1];
    const int per = (sb1 - sb0 + slabs - 1) / slabs;
    int t0 = sb0 + t * per, t1 = t0 + per;
    if (t0 > sb1) t0 = sb1;
    if (t1 > sb1) t1 = sb1;
    float *mine = ys[w];
    for (int i = lane; i < kXR; i += kWave) mine[i] = 0.0f;
    for (int sg = t0 + w; sg < t1; sg += kXWaves) {
        const float xj = x[seg_input[sg]];                       // wave-uniform
        if (xj == 0.0f) continue;                                 // the whole segment is never read
        const int p0 = seg_ptr[sg], p1 = seg_ptr[sg + 1];
        int k = p0 + lane;
        for (; k + 7 * kWave < p1; k += 8 * kWave) {              // eight independent loads per array in flight
            int r[8];
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { r[u] = erow[k + u * kWave]; v[u] = __builtin_nontemporal_load(&evals[k + u * kWave]); }
#pragma unroll
            for (int u = 0; u < 8; ++u) mine[r[u]] += xj * v[u];
        }
        for (; k < p1; k += kWave) mine[erow[k]] += xj * evals[k];
    }
    __syncthreads();
    const int64_t row0 = (int64_t)b * kXR;
    const int n = (int)(rows - row0 < kXR ? rows - row0 : kXR);
    // slabs == 1: `out` is y itself; otherwise the partial of (b, t), added up by k_xs_combine
    float *dst = slabs == 1 ? out + row0 : out + ((int64_t)b * slabs + t) * kXR;
    for (int i = threadIdx.x; i < n; i += kXWaves * kWave) dst[i] = ((ys[0][i] + ys[1][i]) + ys[2][i]) + ys[3][i];
}

__global__ __launch_bounds__(kBlock) void k_xs_combine(int64_t rows, int slabs, const float *__restrict__ part,
                                                       float *__restrict__ y)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= rows) return;
    const int64_t b = r / kXR;
    const float *p = part + b * slabs * kXR + (r - b * kXR);
    float acc = 0.0f;
    int t = 0;
    for (; t + 7 < slabs; t += 8) {      // eight loads in flight, added in slab order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(t + u) * kXR];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; t < slabs; ++t) acc += p[(int64_t)t * kXR];
    y[r] = acc;
}

}  // namespace

void destroy_xskip(XskipPlan &p)
{
    if (p.d_block_seg) (void)hipFree(p.d_block_seg);
    if (p.d_seg_input) (void)hipFree(p.d_seg_input);
    if (p.d_seg_ptr) (void)hipFree(p.d_seg_ptr);
    if (p.d_erow) (void)hipFree(p.d_erow);
    if (p.d_evals) (void)hipFree(p.d_evals);
    if (p.d_part) (void)hipFree(p.d_part);
    p = XskipPlan();
}

int plan_xskip(spmv_csr &h, hipStream_t s)
{
    if (h.plan_xskip.ready && h.plan_xskip.stamp.gen == h.values_gen) return SPMV_OK;   // (a stale copy is rebuilt)
    destroy_xskip(h.plan_xskip);
    XskipPlan p;
    p.nblocks = (int)((h.rows + kXR - 1) / kXR);
    const int64_t table = (int64_t)p.nblocks * h.cols;
    if (table > kXTableMax) {
        set_error("spmv_csr_plan(xskip): %d output blocks x %lld inputs = %lld table entries > 2^27 -- this variant is for "
                  "dense-ish matrices (the reference's regime)", p.nblocks, (long long)h.cols, (long long)table);
        return SPMV_ERR_INVALID;
    }
    if (h.rows == 0 || h.nnz == 0) {
        p.ready = true;
        p.stamp.gen = h.values_gen;
        h.plan_xskip = p;
        return SPMV_OK;
    }
    int rc;
    DevPtr<int32_t> cnt, pos, flag, tot_nnz, tot_seg, dup;
    SPMV_HIP_TRY(cnt.alloc((size_t)table));
    SPMV_HIP_TRY(pos.alloc((size_t)table));
    SPMV_HIP_TRY(flag.alloc((size_t)table));
    SPMV_HIP_TRY(tot_nnz.alloc(1));
    SPMV_HIP_TRY(tot_seg.alloc(1));
    SPMV_HIP_TRY(dup.alloc(2));
    SPMV_HIP_TRY(hipMemsetAsync(cnt.p, 0, sizeof(int32_t) * (size_t)table, s));
    SPMV_HIP_TRY(hipMemsetAsync(dup.p, 0, 2 * sizeof(int32_t), s));
    const unsigned grows = (unsigned)((h.rows + 3) / 4), gtab = (unsigned)((table + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_xs_count, dim3(grows), dim3(kBlock), 0, s, h.rows, h.cols, h.d_row_ptr, h.d_col_idx, cnt.p, dup.p);
    if ((rc = check_launch("k_xs_count"))) return rc;
    SPMV_HIP_TRY(hipMemcpyAsync(pos.p, cnt.p, sizeof(int32_t) * (size_t)table, hipMemcpyDeviceToDevice, s));
    if ((rc = exclusive_scan_i32(pos.p, table, tot_nnz.p, s))) return rc;
    hipLaunchKernelGGL(k_xs_flags, dim3(gtab), dim3(kBlock), 0, s, table, cnt.p, flag.p);
    if ((rc = check_launch("k_xs_flags"))) return rc;
    if ((rc = exclusive_scan_i32(flag.p, table, tot_seg.p, s))) return rc;
    int32_t nseg = 0, has_dup[2] = {0, 0};
    SPMV_HIP_TRY(hipMemcpyAsync(&nseg, tot_seg.p, sizeof nseg, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipMemcpyAsync(has_dup, dup.p, sizeof has_dup, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));
    if (has_dup[0]) {
        set_error("spmv_csr_plan(xskip): a row holds the same column twice; this variant needs sorted, duplicate-free rows");
        return SPMV_ERR_INVALID;
    }
    if (has_dup[1]) {
        set_error("spmv_csr_plan(xskip): the columns of a row do not ascend; this variant needs sorted, duplicate-free rows "
                  "(a duplicate column in an unsorted row would go unnoticed)");
        return SPMV_ERR_INVALID;
    }
    p.nseg = nseg;
    DevPtr<int32_t> block_seg, seg_input, seg_ptr;
    DevPtr<uint16_t> erow;
    DevPtr<float> evals;
    SPMV_HIP_TRY(block_seg.alloc((size_t)p.nblocks + 1));
    SPMV_HIP_TRY(seg_input.alloc((size_t)nseg + 1));
    SPMV_HIP_TRY(seg_ptr.alloc((size_t)nseg + 1));
    SPMV_HIP_TRY(erow.alloc((size_t)h.nnz));
    SPMV_HIP_TRY(evals.alloc((size_t)h.nnz));
    hipLaunchKernelGGL(k_xs_segments, dim3(gtab), dim3(kBlock), 0, s, (int64_t)p.nblocks, h.cols, cnt.p, pos.p, flag.p, tot_seg.p,
                       tot_nnz.p, seg_input.p, seg_ptr.p, block_seg.p);
    if ((rc = check_launch("k_xs_segments"))) return rc;
    hipLaunchKernelGGL(k_xs_fill, dim3(grows), dim3(kBlock), 0, s, h.rows, h.cols, h.d_row_ptr, h.d_col_idx, h.d_vals, pos.p,
                       erow.p, evals.p);
    if ((rc = check_launch("k_xs_fill"))) return rc;
    // slabs per block: enough workgroups to fill the chip four times, at least 8 segments each, at most 64
    int slabs = (4 * device_cus(h.device) + p.nblocks - 1) / p.nblocks;
    const int most = (int)((nseg / p.nblocks + 7) / 8);
    if (slabs > most) slabs = most;
    if (slabs > kXSlabMax) slabs = kXSlabMax;
    if (slabs < 1) slabs = 1;
    p.slabs = slabs;
    if ((rc = stamp_values(h, s, p.stamp))) return rc;
    if (slabs > 1) SPMV_HIP_TRY(hipMalloc((void **)&p.d_part, sizeof(float) * (size_t)p.nblocks * slabs * kXR));
    SPMV_HIP_TRY(hipStreamSynchronize(s));   // the temporaries are freed on return
    p.d_block_seg = block_seg.release();
    p.d_seg_input = seg_input.release();
    p.d_seg_ptr = seg_ptr.release();
    p.d_erow = erow.release();
    p.d_evals = evals.release();
    p.ready = true;
    h.plan_xskip = p;
    return SPMV_OK;
}

int launch_xskip(const spmv_csr &h, const float *x, float *y, hipStream_t s)
{
    const XskipPlan &p = h.plan_xskip;
    if (!p.ready) {
        set_error("spmv_csr_run: variant xskip is not planned (call spmv_csr_plan first)");
        return SPMV_ERR_NOT_PLANNED;
    }
    if (int rc = require_fresh_values(h, p.stamp, s, "xskip")) return rc;
    if (h.rows == 0) return SPMV_OK;
    if (h.nnz == 0) {
        SPMV_HIP_TRY(hipMemsetAsync(y, 0, sizeof(float) * (size_t)h.rows, s));
        return SPMV_OK;
    }
    hipLaunchKernelGGL(k_xskip, dim3((unsigned)(p.nblocks * p.slabs)), dim3(kXWaves * kWave), 0, s, h.rows, p.slabs, p.d_block_seg,
                       p.d_seg_input, p.d_seg_ptr, p.d_erow, p.d_evals, x, p.slabs == 1 ? y : p.d_part);
    int rc = check_launch("k_xskip");
    if (rc || p.slabs == 1) return rc;
    hipLaunchKernelGGL(k_xs_combine, dim3((unsigned)((h.rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, h.rows, p.slabs,
                       p.d_part, y);
    return check_launch("k_xs_combine");
}

}  // namespace spmv
