// kernels_rows.hip -- row-mapped CSR SpMV kernels for gfx950 (wave64).
//
//   SPMV_SCALAR  a lane per row, terms added in ascending k, unfused mul+add: role of csr_naive_kernel
//              (/root/reference/src/kernels/csr_naive.cu:6-23); the same per-row operation order as SgemvCPU
//              (src/tester.cpp:36-45), so results are bit-identical to the CPU oracle.  Mean rows of up to 32 nonzeros:
//              k_wave_bundle<..., ORDERED> (below); up to 64: k_scalar (the workgroup's rows staged through LDS); longer:
//              k_scalar_long (a wavefront per row, readlane chain).
//   k_wave     one 64-lane wavefront per row, lanes stride the row, __shfl_down tree
//              role of wsp_kernel_v0 (src/kernels/wsp.cu:4-56): "one warp per output,
//              butterfly reduce, lane 0 stores" -- re-derived for CSR and 64 lanes.
//   k_wave_bundle + k_wave_pieces + k_wave_combine (SPMV_WAVE_PIPE)  a wavefront per 64 consecutive rows: coalesced
//              streams at full depth, products parked in LDS, x from a window in LDS, long rows in pieces for the whole
//              chip: role of wsp_kernel_v1 (src/kernels/wsp.cu:59-138), the reference's pipelined version.
//   k_vector   G-lane groups per row (G = 2..32), the short-row member of the family
//              role of asp_kernel_v* (src/kernels/asp.cu:6-211: many outputs per block).
//
// No MFMA anywhere (0.25 flop/byte): streams, gathers, LDS.
#include <cstdlib>
#include <vector>
#include "spmv_internal.hpp"

namespace spmv {

// A/B builds (tools/build_variant.sh): what a kernel costs without its gathers / its long rows / its in-order sums
#ifdef SPMV_R_NOGATHER
#define XG(c, k) x[((c) & 0) + ((k) & 1023)]
#else
#define XG(c, k) x[(c)]
#endif

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
            const float p = XG(col_idx[k], k) * vals[k];
            prod[k - wb] = p;
        }
        __syncthreads();
        if (r < rows) {
            float acc = 0.0f;
#ifdef SPMV_R_NOSUM
            if (e > b) acc = prod[b - wb];
#else
            for (int32_t k = b; k < e; ++k) acc = acc + prod[k - wb];
#endif
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

// SPMV_WAVE_PIPE (the slot of wsp_kernel_v1, the reference's unrolled / prefetching version, wsp.cu:59-138): a wavefront
// owns 64 consecutive rows; 8 or 16 wavefronts (512 or 1024 rows) make a workgroup.
//   * Rows of up to kBundleCap nonzeros: the wave takes them in runs of consecutive rows that hold at most kBundleCap
//     nonzeros together, streams that contiguous range with coalesced loads -- lane-consecutive nonzeros, the whole wave busy
//     whatever the row lengths, ALL of the run's loads issued before the first use and the next run's loads issued before
//     this one's sums (the pipelining wsp_kernel_v1 adds to v0, at the depth HBM latency asks of a wave64) -- parks the
//     products in its LDS slice, and every lane adds the products of ITS row in order; rows longer than a wavefront are
//     added by all 64 lanes with the __shfl_down tree.
//   * Longer rows are cut by the plan into pieces of kPieceLen nonzeros for k_wave_pieces: one wavefront per piece wherever
//     on the chip there is room (one coalesced stream + a wave sum into a word of the plan's scratch); k_wave_combine then
//     adds a row's pieces in order.
//   * x comes from a window in LDS -- the entries [lo, lo + wave_window) that hold every column of the block's rows
//     (the long ones excepted: theirs are gathered from memory), found by the plan -- where such a window exists, else
//     from memory; the columns of a windowed block are read as the plan's 16-bit offsets into the window (6 bytes per
//     nonzero instead of 8).
// The plan is a function of row_ptr and col_idx (not of the values), made by spmv_csr_plan or by the first run.
// Round 3's A/Bs (profiles/r03_rows_bounds.jsonl; config 3 with band 8192, 0.759 ms before, 0.386 ms now): the longest rows
// -- 65 536 nonzeros on ONE wavefront, 256 dependent trips -- were 0.40 ms of it; with the streams at full depth (0.16 ms
// without the gathers) the 4-byte gathers from L2 were 0.34 ms more, at one 128-byte line per lane and instruction: hence
// the window.  Tried in between and dropped: long rows by their own wave (the other seven wait with the workgroup's LDS), by
// the workgroup after a barrier with lists and counters in LDS (0.55-0.69 ms: barriers and bookkeeping), a window guessed
// without a plan from sampled columns (two more round trips in every workgroup), the pieces inside the bundle kernel
// (0.44 ms: its LDS leaves three workgroups per CU, too few loads in flight for the pieces), a window per long row (the
// synthetic band law gives a row of n nonzeros 8n columns: as many bytes of windows as of matrix), the pieces on a second
// stream beside the bundles (no overlap: 0.386 against 0.382 ms), the 16-bit offsets in a PAIR layout (a lane holds two
// consecutive nonzeros, 8-byte loads of values: config 2 0.0317 -> 0.0303 ms, but config 3 0.358 -> 0.374 and config 4
// 0.614 -> 0.640; with the runs aligned to even nonzeros config 2 fell to 0.040) -- in the slice layout the same offsets
// gain everywhere (config 2 0.0320 -> 0.0304, config 3 0.356 -> 0.350, config 4 0.618 -> 0.588) and are what ships.
#ifndef SPMV_BUNDLE_CAP
#define SPMV_BUNDLE_CAP 512
#endif
constexpr int kBundleCap = SPMV_BUNDLE_CAP;        // products per wave and run (4 bytes of LDS each) = nonzeros per piece
constexpr int kBundleSlices = kBundleCap / kWave;
// rows (= threads) of a workgroup that shares a window, and the entries of x it keeps in LDS: a band of 8192 + its rows +
// slack.  512 rows / 35 KiB (three workgroups per CU) for matrices of less than 2 Mi rows, where the grid is a few
// rounds of workgroups and whole rounds count (config 2: 0.0377 ms against 0.0413); 1024 rows / 37 KiB (two workgroups =
// 32 wavefronts per CU, half the window traffic) above (config 4: 0.632 against 0.716 ms, config 3: 0.372 against 0.391).
__host__ __device__ constexpr int wave_window(int block) { return block == 1024 ? 9472 : 8960; }
constexpr int64_t kWaveBigRows = 2 << 20;
#ifndef SPMV_PIECE_LEN
#define SPMV_PIECE_LEN 1024
#endif
constexpr int kPieceLen = SPMV_PIECE_LEN;          // nonzeros of a piece of a long row (one wavefront of k_wave_pieces)
static_assert(kBundleCap % (4 * kWave) == 0 && wave_window(512) % 32 == 0 && wave_window(1024) % 32 == 0, "bundle geometry");

// Workgroups are dealt round-robin over the 8 XCDs: block b of the grid takes work item xcd_item(b, n), so that every XCD
// gets ONE contiguous range of the n items -- neighbouring blocks of rows share lines of x, and a contiguous eighth of a
// banded matrix keeps its part of x in that XCD's 4 MiB of L2 (bijective for any n).
__device__ __forceinline__ int xcd_item(int bid, int n)
{
    const int q = n / kXcds, rem = n % kXcds;
    const int j = bid % kXcds, idx = bid / kXcds;
    return j * q + (j < rem ? j : rem) + idx;
}

// Buffer descriptors (wave-uniform base in scalar registers, 32-bit lane offsets, immediate slice offsets, reads past the
// end return 0): the streamed loads of a run share ONE offset register and need no predicates, and a gather's address is
// the column itself.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ const void *uniform_ptr(const void *p)
{
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (const void *)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ int32_t buf_i32(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0);
}

// the nonzeros [k0, k0 + len) (len <= kBundleCap, wave-uniform): all their loads issued at once, in groups of four slices
// that are skipped past the end; slot u of a lane = nonzero k0 + u*64 + lane (column 0, value 0 past the end)
template <int S>
__device__ __forceinline__ void bundle_loads(int lane, int64_t k0, int32_t len, const int32_t *__restrict__ col_idx,
                                             const float *__restrict__ vals, int32_t (&c)[S], float (&v)[S])
{
    constexpr int kBundleSlices = S;
    const __amdgpu_buffer_rsrc_t cr = rsrc_of(uniform_ptr(col_idx + k0), (uint32_t)len * 4u);
    const __amdgpu_buffer_rsrc_t vr = rsrc_of(uniform_ptr(vals + k0), (uint32_t)len * 4u);
    int lane4 = lane * 4;               // + the slice's immediate offset
    asm volatile("" : "+v"(lane4));     // (opaque: known bits would turn the add into an OR that is not folded into offset:)
#pragma unroll
    for (int t = 0; t < kBundleSlices; t += 4) {
        if (t * kWave < len) {
#pragma unroll
            for (int u = t; u < t + 4; ++u) {
                c[u] = buf_i32(cr, lane4 + u * (kWave * 4));
                v[u] = buf_f32(vr, lane4 + u * (kWave * 4));
            }
        } else {
#pragma unroll
            for (int u = t; u < t + 4; ++u) { c[u] = 0; v[u] = 0.0f; }
        }
    }
}
// the same with the columns as the plan's 16-bit offsets into the block's window (2 bytes per nonzero instead of 4):
// c[u] = offset of nonzero k0 + u*64 + lane (0 past the end)
template <int S>
__device__ __forceinline__ void bundle_loads16(int lane, int64_t k0, int32_t len, const uint16_t *__restrict__ col16,
                                               const float *__restrict__ vals, int32_t (&c)[S], float (&v)[S])
{
    const __amdgpu_buffer_rsrc_t cr = rsrc_of(uniform_ptr(col16 + k0), (uint32_t)len * 2u);
    const __amdgpu_buffer_rsrc_t vr = rsrc_of(uniform_ptr(vals + k0), (uint32_t)len * 4u);
    int lane2 = lane * 2, lane4 = lane * 4;
    asm volatile("" : "+v"(lane2), "+v"(lane4));
#pragma unroll
    for (int t = 0; t < S; t += 4) {
        if (t * kWave < len) {
#pragma unroll
            for (int u = t; u < t + 4; ++u) {
                c[u] = (int32_t)__builtin_amdgcn_raw_buffer_load_b16(cr, lane2 + u * (kWave * 2), 0, 0);
                v[u] = buf_f32(vr, lane4 + u * (kWave * 4));
            }
        } else {
#pragma unroll
            for (int u = t; u < t + 4; ++u) { c[u] = 0; v[u] = 0.0f; }
        }
    }
}
// v[u] *= x[c[u]]: from the window in LDS (win: entries [lo, lo + wlast] of x, every column of the caller's rows
// inside it -- the plan checked) or from memory, through a descriptor when x is shorter than 4 GiB (BUFX), else plain loads
template <bool BUFX, int S>
__device__ __forceinline__ void bundle_multiply(int lane, int32_t len, __amdgpu_buffer_rsrc_t xr, const float *__restrict__ x,
                                                const float *win, int32_t lo, uint32_t wlast, const int32_t (&c)[S], float (&v)[S])
{
    constexpr int kBundleSlices = S;
#pragma unroll
    for (int t = 0; t < kBundleSlices; t += 4) {
        if (t * kWave < len) {
            float xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#ifdef SPMV_R_NOGATHER
                const int32_t cc = (lo < 0 ? 0 : lo) + (((t + u) * kWave + lane) & 1023);
#else
                const int32_t cc = c[t + u];
#endif
                if (win) xv[u] = win[min((uint32_t)(cc - lo), wlast)];   // (slots past the end hold column 0; lo = 0: cc is an offset)
                else xv[u] = BUFX ? buf_f32(xr, cc << 2) : x[cc];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) v[t + u] *= xv[u];
        }
    }
}

// the workgroup's window: entries [lo, lo + wave_window(BLOCK)) of x into LDS (zeros past the end of x).  LDS-DMA
// (global_load_lds_dwordx4: no register in between, every 16-byte piece of a lane in flight at once), issued before the
// wave waits for its row pointers.
template <int BLOCK>
__device__ __forceinline__ void load_window(float *win, const float *__restrict__ x, int64_t lo, int64_t cols)
{
    if (lo + wave_window(BLOCK) + 3 < cols) {      // workgroup-uniform
        for (int i = threadIdx.x * 4; i < wave_window(BLOCK); i += BLOCK * 4)
            __builtin_amdgcn_global_load_lds(x + lo + i, win + i, 16, 0, 0);
    } else {
        for (int i = threadIdx.x * 4; i < wave_window(BLOCK); i += BLOCK * 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) win[i + q] = lo + i + q < cols ? x[lo + i + q] : 0.0f;
        }
    }
}

// A workgroup = BLOCK rows = BLOCK / 64 wavefronts.  MODE 0: x gathered from memory.  MODE 1: from the block's window in LDS where the
// plan has one (blk_lo >= 0).  MODE 2: the plan pass for MODE 1 -- the column span of the block's rows, the long ones
// excepted -> blk_lo.
// ORDERED (SPMV_SCALAR's use of this kernel): every row is added in ascending k with one rounding per product and per add,
// the host loop's arithmetic -- a row longer than a wavefront through a readlane chain over its products in LDS, a row
// longer than a run straight from memory by its wave (scalar_row_by_wave) instead of in pieces.
template <bool BUFX, int BLOCK, int MODE, bool ORDERED>
__global__ __launch_bounds__(BLOCK) void k_wave_bundle(int64_t rows, int64_t cols, uint32_t x_bytes,
                                                       const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col_idx,
                                                       const float *__restrict__ vals, const float *__restrict__ x,
                                                       float *__restrict__ y, int32_t *__restrict__ blk_lo,
                                                       uint16_t *__restrict__ col16)
{
    constexpr int kWaves = BLOCK / kWave;
    __shared__ float prod_all[MODE >= 2 ? 1 : kWaves][MODE >= 2 ? 1 : kBundleCap];
    constexpr int kWindow = wave_window(BLOCK);
    __shared__ __attribute__((aligned(16))) float win_lds[MODE == 1 ? kWindow : 4];
    __shared__ int32_t smin[kWaves], smax[kWaves];
    const int lane = threadIdx.x & (kWave - 1);
    // (readfirstlane: the compiler must SEE that the wave index, and with it every run's bounds and descriptor base, is
    // wave-uniform -- otherwise each buffer load of a run is wrapped in a waterfall loop over its descriptor)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *prod = prod_all[MODE >= 2 ? 0 : wave];
    const __amdgpu_buffer_rsrc_t xr = rsrc_of(x, x_bytes);
    const int blk = xcd_item((int)blockIdx.x, (int)gridDim.x);          // this workgroup's block of BLOCK rows
    const int64_t r0 = ((int64_t)blk * kWaves + wave) * kWave;
    const int64_t r = r0 + lane;
    const bool live = r < rows;
    const int32_t b = row_ptr[live ? r : rows], e = row_ptr[live ? r + 1 : rows];
    const int n = r0 < rows ? (int)((rows - r0 < kWave) ? rows - r0 : kWave) : 0;
    int32_t lo = -1;
    if (MODE == 1 || MODE == 3) lo = __builtin_amdgcn_readfirstlane(blk_lo[blk]);
    const float *win = (MODE == 1 && lo >= 0) ? win_lds : nullptr;
    if (MODE == 1 && win) load_window<BLOCK>(win_lds, x, lo, cols);        // (needs blk_lo only: out before b / e are waited for)
    const unsigned long long long_mask = __ballot(e - b > kBundleCap);     // k_wave_pieces' rows

    // the next run: the rows from `from` on that are not long and hold at most kBundleCap nonzeros together
    int i0 = 0, i1 = 0;
    int32_t sb = 0, len = 0;
    auto next_run = [&](int from) {
        i0 = from;
        while (i0 < n && ((long_mask >> i0) & 1ULL)) ++i0;
        i1 = i0;
        len = 0;
        if (i0 >= n) return;
        sb = __builtin_amdgcn_readfirstlane(__shfl(b, i0));
        const unsigned long long stop = long_mask >> i0;                   // the first long row after i0 ends the run
        const int room = stop ? __ffsll((long long)stop) - 1 : kWave;
        i1 = i0 + __popcll(__ballot(lane >= i0 && lane < n && lane < i0 + room && e - sb <= kBundleCap));   // e ascends
        len = __builtin_amdgcn_readfirstlane(__shfl(e, i1 - 1)) - sb;
    };
    int32_t c[kBundleSlices];
    float v[kBundleSlices];
    // a block with a window reads the plan's 16-bit offsets into it (MODE 1); everything else the 32-bit columns
    const bool off16 = MODE == 1 && win != nullptr && col16 != nullptr;
    auto run_loads = [&]() {
        if (off16) bundle_loads16(lane, sb, len, col16, vals, c, v);
        else bundle_loads(lane, sb, len, col_idx, vals, c, v);
    };
    next_run(0);
    run_loads();
    if (MODE == 1) {
        // s_barrier does not drain the vector-memory counter and LDS-DMA writes LDS when its load returns: every wave
        // waits for its own pieces of the window (vmcnt(0); expcnt / lgkmcnt left alone) before the barrier
        if (win) __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();                    // the window has landed
    }
    int32_t cmin = 0x7fffffff, cmax = -1;
    while (i0 < n) {
        if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < kBundleSlices; ++u) {
                if (u * kWave + lane < len) { cmin = min(cmin, c[u]); cmax = max(cmax, c[u]); }
            }
            next_run(i1);
            run_loads();
            continue;
        }
        if (MODE == 3) {        // the block's offsets into its window, where it has one
            if (lo >= 0) {
#pragma unroll
                for (int u = 0; u < kBundleSlices; ++u)
                    if (u * kWave + lane < len) col16[(int64_t)sb + u * kWave + lane] = (uint16_t)(c[u] - lo);
            }
            next_run(i1);
            run_loads();
            continue;
        }
        bundle_multiply<BUFX, kBundleSlices>(lane, len, xr, x, win, off16 ? 0 : lo, (uint32_t)(kWindow - 1), c, v);
#pragma unroll
        for (int t = 0; t < kBundleSlices; t += 4) {
            if (t * kWave < len) {
#pragma unroll
                for (int u = t; u < t + 4; ++u) prod[u * kWave + lane] = v[u];
            }
        }
        // this run's rows and range; the next run's loads go out under this run's sums
        const int a0 = i0, a1 = i1;
        const int32_t asb = sb;
        next_run(a1);
        run_loads();
        const bool mine = lane >= a0 && lane < a1;
        const bool is_long = mine && e - b > kWave;
        if (mine && !is_long) {
            float acc = 0.0f;
            int32_t k = b - asb;
            const int32_t ke = e - asb;
            for (; k + 3 < ke; k += 4) {       // four reads in flight, added in order
                const float p0 = prod[k], p1 = prod[k + 1], p2 = prod[k + 2], p3 = prod[k + 3];
                acc += p0;
                acc += p1;
                acc += p2;
                acc += p3;
            }
            for (; k < ke; ++k) acc += prod[k];
            y[r] = acc;
        }
        unsigned long long todo = __ballot(is_long);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int32_t lb = __shfl(b, src) - asb, le = __shfl(e, src) - asb;
            float acc;
            if (ORDERED) {
                acc = 0.0f;     // wave-uniform
                for (int32_t k0 = lb; k0 < le; k0 += kWave) {
                    const float pr = k0 + lane < le ? prod[k0 + lane] : 0.0f;
                    const int m = le - k0 < kWave ? le - k0 : kWave;
                    if (m == kWave) {
#pragma unroll
                        for (int l = 0; l < kWave; ++l)
                            acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pr), l));
                    } else {
                        for (int l = 0; l < m; ++l) acc = acc + __shfl(pr, l);
                    }
                }
            } else {
                float s0 = 0.0f, s1 = 0.0f;
                int32_t k = lb + lane;
                for (; k + kWave < le; k += 2 * kWave) {
                    s0 += prod[k];
                    s1 += prod[k + kWave];
                }
                if (k < le) s0 += prod[k];
                acc = wave_reduce_sum(s0 + s1);
            }
            if (lane == 0) y[r0 + src] = acc;
        }
        // (the next run's products overwrite prod only after every lane of this wave is past its reads: one wave,
        // program order)
    }
    if (ORDERED && MODE < 2) {
        unsigned long long todo = long_mask;
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const float acc = scalar_row_by_wave(lane, __shfl(b, src), __shfl(e, src), col_idx, vals, x);
            if (lane == 0) y[r0 + src] = acc;
        }
    }
    if (MODE == 2) {
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            cmin = min(cmin, __shfl_xor(cmin, o));
            cmax = max(cmax, __shfl_xor(cmax, o));
        }
        if (lane == 0) { smin[wave] = cmin; smax[wave] = cmax; }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < kWaves; ++i) { cmin = min(cmin, smin[i]); cmax = max(cmax, smax[i]); }
            int32_t w = 0;                                  // (no short nonzeros: any window will do)
            if (cmax >= cmin) {
                w = cmin & ~31;
                if (cmax - w >= kWindow) w = -1;
            }
            blk_lo[blk] = w;
        }
    }
}

// one wavefront per piece of kPieceLen nonzeros, wherever on the chip there is room: partial[p] = sum of its products (x from
// memory: the long rows' columns are the wide ones; no LDS, eight waves per SIMD, 2 x 16 loads in flight each)
// piece_base[p] >= 0: the piece's columns are read as the plan's 16-bit offsets from that column (col16; 6 bytes per
// nonzero), which the plan stores where a piece spans fewer than 65 536 columns.
template <bool BUFX>
__global__ __launch_bounds__(kBlock) void k_wave_pieces(int npieces, uint32_t x_bytes, const int32_t *__restrict__ piece_k0,
                                                        const int32_t *__restrict__ piece_len,
                                                        const int32_t *__restrict__ piece_base,
                                                        const uint16_t *__restrict__ col16,
                                                        const int32_t *__restrict__ col_idx, const float *__restrict__ vals,
                                                        const float *__restrict__ x, float *__restrict__ partial)
{
    constexpr int S = kPieceLen / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    const int p = xcd_item((int)blockIdx.x, (int)gridDim.x) * (kBlock / kWave) + (threadIdx.x >> 6);
    if (p >= npieces) return;   // wave-uniform
    const __amdgpu_buffer_rsrc_t xr = rsrc_of(x, x_bytes);
    const int32_t k0 = __builtin_amdgcn_readfirstlane(piece_k0[p]), len = __builtin_amdgcn_readfirstlane(piece_len[p]);
    const int32_t base = piece_base ? __builtin_amdgcn_readfirstlane(piece_base[p]) : -1;
    int32_t c[S];
    float v[S];
    if (base >= 0) {
        bundle_loads16<S>(lane, k0, len, col16, vals, c, v);
#pragma unroll
        for (int u = 0; u < S; ++u) c[u] += base;
    } else {
        bundle_loads<S>(lane, k0, len, col_idx, vals, c, v);
    }
    bundle_multiply<BUFX, S>(lane, len, xr, x, nullptr, 0, 0u, c, v);
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
    for (int t = 0; t < S; t += 4) {
        if (t * kWave < len) {      // (a slot past the end holds 0 * x[0], which is not 0 when x[0] is Inf or NaN)
            a0 += t * kWave + lane < len ? v[t] : 0.0f;
            a1 += (t + 1) * kWave + lane < len ? v[t + 1] : 0.0f;
            a2 += (t + 2) * kWave + lane < len ? v[t + 2] : 0.0f;
            a3 += (t + 3) * kWave + lane < len ? v[t + 3] : 0.0f;
        }
    }
    const float acc = wave_reduce_sum((a0 + a1) + (a2 + a3));
    if (lane == 0) partial[p] = acc;
}

// plan pass: one wavefront per piece -- its smallest column is its base; where the piece spans fewer than 65 536 columns the
// offsets from the base go to col16 (the entries of a long row's nonzeros, which the bundle pass leaves alone)
__global__ __launch_bounds__(kBlock) void k_wave_plan_piece16(int npieces, const int32_t *__restrict__ piece_k0,
                                                              const int32_t *__restrict__ piece_len,
                                                              const int32_t *__restrict__ col_idx, int32_t *__restrict__ piece_base,
                                                              uint16_t *__restrict__ col16)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int p = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (p >= npieces) return;
    const int32_t k0 = piece_k0[p], len = piece_len[p];
    int32_t cmin = 0x7fffffff, cmax = -1;
    for (int32_t k = lane; k < len; k += kWave) {
        const int32_t c = col_idx[k0 + k];
        cmin = min(cmin, c);
        cmax = max(cmax, c);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        cmin = min(cmin, __shfl_xor(cmin, o));
        cmax = max(cmax, __shfl_xor(cmax, o));
    }
    const bool fits = cmax - cmin < 65536;
    if (lane == 0) piece_base[p] = fits ? cmin : -1;
    if (fits)
        for (int32_t k = lane; k < len; k += kWave) col16[k0 + k] = (uint16_t)(col_idx[k0 + k] - cmin);
}

// one thread per long row: its pieces added in order
__global__ __launch_bounds__(kBlock) void k_wave_combine(int n_long, const int32_t *__restrict__ long_row,
                                                         const int32_t *__restrict__ long_first,
                                                         const float *__restrict__ partial, float *__restrict__ y)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_long) return;
    float acc = 0.0f;
    for (int p = long_first[i]; p < long_first[i + 1]; ++p) acc += partial[p];
    y[long_row[i]] = acc;
}

// ---- the plan of SPMV_WAVE_PIPE: the rows of more than kBundleCap nonzeros and their pieces, in row order
constexpr int kWavePlanBlock = 512;
// exclusive prefix of v over the workgroup's threads (total returned in *total by every thread)
__device__ __forceinline__ int block_exclusive_scan(int v, int *lds /* 16 + 1 */, int *total)
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == kWave - 1) lds[wave] = incl;
    __syncthreads();
    int base = 0, sum = 0;
    for (int w = 0; w < kWavePlanBlock / kWave; ++w) {
        if (w < wave) base += lds[w];
        sum += lds[w];
    }
    __syncthreads();
    *total = sum;
    return base + incl - v;
}
__global__ __launch_bounds__(kWavePlanBlock) void k_wave_plan_count(int64_t rows, const int32_t *__restrict__ row_ptr,
                                                                    int32_t *__restrict__ blk_long, int32_t *__restrict__ blk_pieces)
{
    __shared__ int lds[kWavePlanBlock / kWave];
    const int64_t r = (int64_t)blockIdx.x * kWavePlanBlock + threadIdx.x;
    const int32_t n = r < rows ? row_ptr[r + 1] - row_ptr[r] : 0;
    const int pieces = n > kBundleCap ? (n + kPieceLen - 1) / kPieceLen : 0;
    int tl, tp;
    (void)block_exclusive_scan(pieces ? 1 : 0, lds, &tl);
    (void)block_exclusive_scan(pieces, lds, &tp);
    if (threadIdx.x == 0) { blk_long[blockIdx.x] = tl; blk_pieces[blockIdx.x] = tp; }
}
__global__ __launch_bounds__(kWavePlanBlock) void k_wave_plan_fill(int64_t rows, const int32_t *__restrict__ row_ptr,
                                                                   const int32_t *__restrict__ blk_long,
                                                                   const int32_t *__restrict__ blk_pieces,
                                                                   int32_t *__restrict__ long_row, int32_t *__restrict__ long_first,
                                                                   int32_t *__restrict__ piece_k0, int32_t *__restrict__ piece_len)
{
    __shared__ int lds[kWavePlanBlock / kWave];
    const int64_t r = (int64_t)blockIdx.x * kWavePlanBlock + threadIdx.x;
    const int32_t b = r < rows ? row_ptr[r] : 0, e = r < rows ? row_ptr[r + 1] : 0;
    const int pieces = e - b > kBundleCap ? (e - b + kPieceLen - 1) / kPieceLen : 0;
    int tl, tp;
    const int li = blk_long[blockIdx.x] + block_exclusive_scan(pieces ? 1 : 0, lds, &tl);     // blk_*: exclusive prefixes
    const int pi = blk_pieces[blockIdx.x] + block_exclusive_scan(pieces, lds, &tp);
    if (pieces) {
        long_row[li] = (int32_t)r;
        long_first[li] = pi;
        for (int j = 0; j < pieces; ++j) {
            const int32_t k0 = b + j * kPieceLen;
            piece_k0[pi + j] = k0;
            piece_len[pi + j] = e - k0 < kPieceLen ? e - k0 : kPieceLen;
        }
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

void destroy_wave(WavePlan &p)
{
    (void)hipFree(p.d_long_row);
    (void)hipFree(p.d_long_first);
    (void)hipFree(p.d_piece_k0);
    (void)hipFree(p.d_piece_len);
    (void)hipFree(p.d_partial);
    (void)hipFree(p.d_blk_lo);
    (void)hipFree(p.d_col16);
    (void)hipFree(p.d_piece_base);
    p = WavePlan{};
}

template <int MODE, bool ORDERED, int BLOCK>
static void launch_bundle_block(const spmv_csr &h, const WavePlan &p, const float *x, float *y, int32_t *blk_lo, uint16_t *col16,
                                hipStream_t s)
{
    const dim3 grid((unsigned)p.blocks);
    if (h.cols < (1LL << 30))
        hipLaunchKernelGGL((k_wave_bundle<true, BLOCK, MODE, ORDERED>), grid, dim3(BLOCK), 0, s, h.rows, h.cols,
                           (uint32_t)(h.cols * 4), h.d_row_ptr, h.d_col_idx, h.d_vals, x, y, blk_lo, col16);
    else
        hipLaunchKernelGGL((k_wave_bundle<false, BLOCK, MODE, ORDERED>), grid, dim3(BLOCK), 0, s, h.rows, h.cols, 0u,
                           h.d_row_ptr, h.d_col_idx, h.d_vals, x, y, blk_lo, col16);
}
template <int MODE, bool ORDERED = false>
static void launch_bundle(const spmv_csr &h, const WavePlan &p, const float *x, float *y, int32_t *blk_lo, uint16_t *col16,
                          hipStream_t s)
{
    if (p.block_rows == 1024) launch_bundle_block<MODE, ORDERED, 1024>(h, p, x, y, blk_lo, col16, s);
    else launch_bundle_block<MODE, ORDERED, 512>(h, p, x, y, blk_lo, col16, s);
}

// The plan of SPMV_WAVE_PIPE, a function of row_ptr and col_idx (not of the values): the rows of more than kBundleCap
// nonzeros with their pieces, in row order (so that the pieces of a block of 512 rows are consecutive), and the window
// of x of every block where its columns fit one.  Waits for the stream.
int plan_wave(spmv_csr &h, hipStream_t s)
{
    if (h.plan_wave.ready) return SPMV_OK;
    destroy_wave(h.plan_wave);
    WavePlan &p = h.plan_wave;
    if (h.rows == 0 || h.nnz > 32 * h.rows) { p.ready = true; return SPMV_OK; }   // (long-row matrices run k_wave<true>)
    const int64_t nblk = (h.rows + kWavePlanBlock - 1) / kWavePlanBlock;     // (of the two list kernels)
    if (!grid_ok(nblk)) return SPMV_ERR_INVALID;
    const char *fb = getenv("SPMV_WAVE_BLOCK");
    p.block_rows = fb && atoi(fb) == 1024 ? 1024 : (fb && atoi(fb) == 512 ? 512 : (h.rows >= kWaveBigRows ? 1024 : 512));
    p.blocks = (h.rows + p.block_rows - 1) / p.block_rows;
    DevPtr<int32_t> d_bl, d_bp, blk_lo;
    SPMV_HIP_TRY(d_bl.alloc((size_t)nblk));
    SPMV_HIP_TRY(d_bp.alloc((size_t)nblk + 1));
    SPMV_HIP_TRY(blk_lo.alloc((size_t)p.blocks));
    hipLaunchKernelGGL(k_wave_plan_count, dim3((unsigned)nblk), dim3(kWavePlanBlock), 0, s, h.rows, h.d_row_ptr, d_bl.p, d_bp.p);
    if (int rc = check_launch("k_wave_plan_count")) return rc;
    std::vector<int32_t> bl((size_t)nblk), bp((size_t)nblk + 1);
    SPMV_HIP_TRY(hipMemcpyAsync(bl.data(), d_bl.p, sizeof(int32_t) * (size_t)nblk, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipMemcpyAsync(bp.data(), d_bp.p, sizeof(int32_t) * (size_t)nblk, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));
    int64_t nl = 0, np = 0;
    for (int64_t i = 0; i < nblk; ++i) {
        const int32_t l = bl[(size_t)i], q = bp[(size_t)i];
        bl[(size_t)i] = (int32_t)nl;
        bp[(size_t)i] = (int32_t)np;
        nl += l;
        np += q;
    }
    if (np > 0x7fffffffLL) { set_error("SPMV_WAVE_PIPE: %lld pieces", (long long)np); return SPMV_ERR_INVALID; }
    bp[(size_t)nblk] = (int32_t)np;
    p.n_long = (int)nl;
    p.pieces = (int)np;
    DevPtr<int32_t> lr, lf, k0, ln;
    DevPtr<float> part;
    SPMV_HIP_TRY(lr.alloc((size_t)nl));
    SPMV_HIP_TRY(lf.alloc((size_t)nl + 1));
    SPMV_HIP_TRY(k0.alloc((size_t)np));
    SPMV_HIP_TRY(ln.alloc((size_t)np));
    SPMV_HIP_TRY(part.alloc((size_t)np));
    SPMV_HIP_TRY(hipMemcpyAsync(d_bl.p, bl.data(), sizeof(int32_t) * (size_t)nblk, hipMemcpyHostToDevice, s));
    SPMV_HIP_TRY(hipMemcpyAsync(d_bp.p, bp.data(), sizeof(int32_t) * ((size_t)nblk + 1), hipMemcpyHostToDevice, s));
    SPMV_HIP_TRY(hipMemcpyAsync(lf.p + nl, &bp[(size_t)nblk], sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (nl) {
        hipLaunchKernelGGL(k_wave_plan_fill, dim3((unsigned)nblk), dim3(kWavePlanBlock), 0, s, h.rows, h.d_row_ptr, d_bl.p, d_bp.p,
                           lr.p, lf.p, k0.p, ln.p);
        if (int rc = check_launch("k_wave_plan_fill")) return rc;
    }
    p.d_long_row = lr.release();
    p.d_long_first = lf.release();
    p.d_piece_k0 = k0.release();
    p.d_piece_len = ln.release();
    p.d_partial = part.release();
    // the windows: the bundle kernel in its plan mode (the same runs and pieces, minimum and maximum column instead of products)
    launch_bundle<2>(h, p, nullptr, nullptr, blk_lo.p, nullptr, s);
    if (int rc = check_launch("k_wave_bundle<plan>")) return rc;
    std::vector<int32_t> wl((size_t)p.blocks);
    SPMV_HIP_TRY(hipMemcpyAsync(wl.data(), blk_lo.p, sizeof(int32_t) * (size_t)p.blocks, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));      // (also: bl / bp are host memory of this call)
    for (int64_t i = 0; i < p.blocks; ++i) p.win_blocks += wl[(size_t)i] >= 0;
    const char *c16env = getenv("SPMV_WAVE_COL16");
    p.windows = 2 * p.win_blocks >= p.blocks;
    if ((p.windows || np) && !(c16env && c16env[0] == '0')) {
        // 16-bit column offsets, 2 bytes per nonzero of device memory: into the block's window for the short rows of a block
        // that has one, from the piece's smallest column for the pieces of the long rows (where a piece spans fewer than
        // 65 536 columns); what neither covers stays unwritten and unread
        DevPtr<uint16_t> c16;
        DevPtr<int32_t> pbase;
        SPMV_HIP_TRY(c16.alloc((size_t)h.nnz));
        if (p.windows) {
            launch_bundle<3>(h, p, nullptr, nullptr, blk_lo.p, c16.p, s);
            if (int rc = check_launch("k_wave_bundle<col16>")) return rc;
        }
        if (np) {
            SPMV_HIP_TRY(pbase.alloc((size_t)np));
            hipLaunchKernelGGL(k_wave_plan_piece16, dim3((unsigned)((np + kBlock / kWave - 1) / (kBlock / kWave))), dim3(kBlock), 0, s,
                               (int)np, p.d_piece_k0, p.d_piece_len, h.d_col_idx, pbase.p, c16.p);
            if (int rc = check_launch("k_wave_plan_piece16")) return rc;
        }
        SPMV_HIP_TRY(hipStreamSynchronize(s));
        p.d_col16 = c16.release();
        p.d_piece_base = pbase.release();
    }
    p.d_blk_lo = blk_lo.release();
    p.ready = true;
    return SPMV_OK;
}

int launch_scalar(spmv_csr &h, const float *x, float *y, hipStream_t s)
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
    if (h.nnz <= 32 * h.rows) {
        // the bundle kernel of SPMV_WAVE_PIPE with its sums in the host loop's order: operands at the depth and from the x
        // windows that variant's plan provides (made here on the first run of a handle that was not planned)
        if (int rc = plan_wave(h, s)) return rc;
        const WavePlan &p = h.plan_wave;
        if (p.windows) launch_bundle<1, true>(h, p, x, y, p.d_blk_lo, p.d_col16, s);
        else launch_bundle<0, true>(h, p, x, y, nullptr, nullptr, s);
        return check_launch("k_wave_bundle<ordered>");
    }
    hipLaunchKernelGGL(k_scalar, dim3((unsigned)blocks), dim3(kBlock), 0, s, h.rows, h.d_row_ptr,
                       h.d_col_idx, h.d_vals, x, y);
    return check_launch("k_scalar");
}

int launch_wave(spmv_csr &h, const float *x, float *y, bool pipelined, hipStream_t s)
{
    if (h.rows == 0) return SPMV_OK;
    constexpr int kRowsPerBlock = kBlock / kWave;
    int64_t blocks = (h.rows + kRowsPerBlock - 1) / kRowsPerBlock;
    if (!grid_ok(blocks)) return SPMV_ERR_INVALID;
    // bundles of 64 rows pay off while they fit a wave's LDS slice (mean row length <= 32); beyond that a bundle
    // would serialise 64 long rows on one wave (4096 x 4096 at 50 %: 64 waves for the whole chip, 17x slower)
    // SPMV_WAVE on short rows (round 4): the same bundles WITHOUT the x window and the 16-bit offsets -- the plain kernel of
    // the pair (wsp_kernel_v0 beside v1): a wavefront per 64 rows instead of per row (rows of mean 32 left half the lanes of
    // a wavefront-per-row idle and made every trip one dependent gather), rows longer than a wavefront by the 64-lane
    // __shfl_down tree, the longest in pieces.  SPMV_WAVE_PIPE adds the window in LDS and the offsets into it.
    const bool plain_bundle = !pipelined && h.nnz <= 32 * h.rows && getenv("SPMV_WAVE_PER_ROW") == nullptr;
    const bool bundle = (pipelined && h.nnz <= 32 * h.rows) || plain_bundle;
    if (pipelined && !bundle) {
        hipLaunchKernelGGL(k_wave<true>, dim3((unsigned)blocks), dim3(kBlock), 0, s, h.rows, h.d_row_ptr, h.d_col_idx,
                           h.d_vals, x, y);
    } else if (bundle) {
        if (int rc = plan_wave(h, s)) return rc;        // (the first run of a handle that was not planned: allocates, waits)
        const WavePlan &p = h.plan_wave;
        // windows where at least half of the blocks have one (their 35 KiB leave two workgroups per CU); else every
        // gather goes to memory, from three workgroups per CU
        if (p.windows && !plain_bundle) launch_bundle<1>(h, p, x, y, p.d_blk_lo, p.d_col16, s);
        else launch_bundle<0>(h, p, x, y, nullptr, nullptr, s);
#ifndef SPMV_R_NOLONG
        if (p.n_long) {
            if (int rc = check_launch("k_wave_bundle")) return rc;
            const dim3 pgrid((unsigned)((p.pieces + kRowsPerBlock - 1) / kRowsPerBlock));
            if (h.cols < (1LL << 30))
                hipLaunchKernelGGL(k_wave_pieces<true>, pgrid, dim3(kBlock), 0, s, p.pieces, (uint32_t)(h.cols * 4), p.d_piece_k0,
                                   p.d_piece_len, p.d_piece_base, p.d_col16, h.d_col_idx, h.d_vals, x, p.d_partial);
            else
                hipLaunchKernelGGL(k_wave_pieces<false>, pgrid, dim3(kBlock), 0, s, p.pieces, 0u, p.d_piece_k0, p.d_piece_len,
                                   p.d_piece_base, p.d_col16, h.d_col_idx, h.d_vals, x, p.d_partial);
            if (int rc = check_launch("k_wave_pieces")) return rc;
            hipLaunchKernelGGL(k_wave_combine, dim3((unsigned)((p.n_long + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, p.n_long,
                               p.d_long_row, p.d_long_first, p.d_partial, y);
        }
#endif
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
