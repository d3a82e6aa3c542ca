// kernels_binned.hip -- SPMV_PANEL modes 4 and 5, "binned": y = A x for columns WITHOUT locality as two pure streams.
// Two flavours: mode 4 (the first: the SUM launch fetches the tiles from panel-major products; described right here) and
// mode 5 (the PRODUCT launch stores every product where the sum launch streams it; "the SCATTERED flavour" further down --
// what SPMV_AUTO takes since it turned out at least as fast everywhere the two were put side by side).
//
// Why.  The reference's only structure law is i.i.d. uniform positions
// (/root/reference/src/tester.cpp:103-121).  On such a matrix every gather design of this library pays one L2 line
// request per nonzero: the panel sweep (kernels_panel.hip) moves 1.4 x the algorithmic bytes at config 4 and 7.9 x at
// config 5's shard, bound by the 270 G line requests per second the L2s serve.  Here NOTHING is gathered from memory:
//
//   launch 1  k_bin_products   the nonzeros in PANEL-major order (16-bit column inside the panel + value: 6 B); a
//             workgroup stages its panel of x (2^15 columns, 128 KiB) in LDS by LDS-DMA, streams its share of the
//             panel's nonzeros with 16-byte loads and writes the products x[col] * val, 16 bytes per lane, in the same
//             order (4 B per nonzero).
//   launch 2  k_bin_sums       the rows in BINS of at most 8192 (4096); a wavefront owns a bin, its sums private in LDS.
//             Tile (bin b, panel p) is contiguous in the panel-major product array; the wavefront walks its bin's
//             tiles, 64 products and their 16-bit rows (2 B, bin-major: contiguous for the whole bin) per instruction,
//             eight instructions' loads in flight, and adds with a plain LDS read-add-write: rows ascend inside a
//             tile, so equal rows are neighbouring lanes -- folded in registers first (a shift and a compare find
//             them; pairs directly, longer runs by a segmented scan).  No atomics, no barrier, deterministic.
//
// 6 + 4 + 4 + 2 = 16 bytes per nonzero of perfectly sequential traffic, x and y once, 8 bytes of tables per tile
// -- against 8 B per nonzero plus a 128-byte line per gather.  The plan (device side, once): bins = row blocks of equal
// NONZERO counts (panel_row_blocks), a histogram of tiles, two prefix orders, one stable scatter.  The values are
// COPIED (like every layout of the panel family): spmv_csr_values_changed -> re-plan.
//
// Role: the sparse-scale counterpart of the reference's tiled format -- TCSRMatrix (src/tcsr.cpp:5-38) multiplied by
// csr_tiling_kernel (src/kernels/csr_tiling.cu:24-114): x tile in shared memory, tile values streamed.
#include <climits>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "spmv_internal.hpp"

namespace spmv {

namespace {

constexpr int kPwBits = 15;                // columns per panel: 128 KiB of x in LDS
constexpr int kPw = 1 << kPwBits;
constexpr int kProdThreads = 1024;         // one 16-wavefront workgroup per CU holds the panel
constexpr int kMaxPanels = 4096;           // (panel_tile_ptr's histogram) -> at most 2^27 columns
constexpr int kSumWaves = 4;               // wavefronts per workgroup of the sum launch
constexpr int kSpare = 512;                // spare sums per bin for its long rows (see k_bin_runs)


using u4 = unsigned __attribute__((ext_vector_type(4)));
using f4 = float __attribute__((ext_vector_type(4)));

int check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
    return SPMV_OK;
}

// ---- plan kernels ------------------------------------------------------------------------------------------------
// nonzeros of every panel: total[p] = sum over bins of the tile counts, rounded up to a multiple of 8 (16-byte loads)
__global__ __launch_bounds__(256) void k_bin_panel_totals(int nb, int np, int round, const int32_t *__restrict__ tile_ptr, int32_t *__restrict__ total)
{
    __shared__ int part[256];
    const int p = blockIdx.x;
    int sum = 0;
    for (int b = threadIdx.x; b < nb; b += 256) {
        const int32_t *tp = tile_ptr + (int64_t)b * (np + 1);
        sum += tp[p + 1] - tp[p];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[p] = (part[0] + round - 1) & ~(round - 1);
}
// pm[b * np + p] = pbase[p] + (nonzeros of panel p in the bins before b): one workgroup per panel scans the bins
__global__ __launch_bounds__(256) void k_bin_pm(int nb, int np, const int32_t *__restrict__ tile_ptr, const int32_t *__restrict__ pbase,
                                                int32_t *__restrict__ pm)
{
    __shared__ int part[256];
    __shared__ int carry;
    const int p = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) carry = pbase[p];
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int b = b0 + tid;
        int c = 0;
        if (b < nb) {
            const int32_t *tp = tile_ptr + (int64_t)b * (np + 1);
            c = tp[p + 1] - tp[p];
        }
        part[tid] = c;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int v = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (b < nb) pm[(int64_t)b * np + p] = carry + part[tid] - c;
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
}
// Stable scatter of one bin into both orders: a wave per bin, 64 nonzeros per step in CSR order; the rank of a nonzero
// among the same-panel nonzeros of its step comes from a ballot per distinct panel (the method of k_panel_fill).
__global__ __launch_bounds__(256) void k_bin_fill(int nb, int np, const int32_t *__restrict__ brow, const int32_t *__restrict__ row_ptr,
                                                  const int32_t *__restrict__ col_idx, const float *__restrict__ vals,
                                                  const uint16_t *__restrict__ rowloc, const int32_t *__restrict__ tile_ptr,
                                                  const int32_t *__restrict__ pm, uint16_t *__restrict__ c16, float *__restrict__ pvals,
                                                  uint16_t *__restrict__ r16)
{
    extern __shared__ int cursor_all[];     // 4 x np: as many workgroups per CU as the panel count allows (a wave walks its bin alone); 4 x 256 tags
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    if (b >= nb) return;
    int *cursor = cursor_all + w * np;
    volatile int *tags = cursor_all + 4 * np + w * 256;
    const int32_t *tp = tile_ptr + (int64_t)b * (np + 1);
    const int32_t *qp = pm + (int64_t)b * np;
    for (int p = lane; p < np; p += kWave) cursor[p] = tp[p];
    const int s = row_ptr[brow[b]], e = row_ptr[brow[b + 1]];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = s; base < e; base += kWave) {
        const int k = base + lane;
        const bool valid = k < e;
        int col = 0, rl = 0;
        float v = 0.0f;
        if (valid) {
            col = col_idx[k];
            rl = rowloc[k];
            v = vals[k];
        }
        const int p = col >> kPwBits;
        int dest = 0;
        const bool lone = lone_in_step(tags, valid, p, lane);       // the only nonzero of its panel in this step
        if (lone) {
            dest = cursor[p];
            cursor[p] = dest + 1;
        }
        unsigned long long todo = __ballot(valid && !lone);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int pl = __shfl(p, leader);
            const unsigned long long m = __ballot(valid && p == pl);
            const int first = cursor[pl];
            if (valid && p == pl) dest = first + __popcll(m & lt);
            if (lane == leader) cursor[pl] = first + __popcll(m);
            todo &= ~m;
        }
        if (valid) {
            r16[dest] = (uint16_t)rl;                         // bin-major
            const int q = qp[p] + (dest - tp[p]);             // the same rank inside the tile, panel-major
            c16[q] = (uint16_t)(col & (kPw - 1));
            pvals[q] = v;
        }
    }
}

// ---- launch 1: products in panel order ------------------------------------------------------------------------------
__global__ __launch_bounds__(kProdThreads) void k_bin_products(int splits, int64_t cols, const int32_t *__restrict__ pbase,
                                                               const uint16_t *__restrict__ c16, const float *__restrict__ pvals,
                                                               const float *__restrict__ x, float *__restrict__ prod)
{
    extern __shared__ __attribute__((aligned(16))) float xp[];      // the panel: kPw floats
    const int tid = threadIdx.x;
    const int p = blockIdx.x / splits, part = blockIdx.x - p * splits;
    const int64_t g0 = (int64_t)p << kPwBits;
    if (g0 + kPw + 3 < cols) {                                      // workgroup-uniform
#pragma unroll
        for (int i = tid * 4; i < kPw; i += kProdThreads * 4) __builtin_amdgcn_global_load_lds(x + g0 + i, xp + i, 16, 0, 0);
    } else {
        for (int i = tid; i < kPw; i += kProdThreads) xp[i] = g0 + i < cols ? x[g0 + i] : 0.0f;
    }
    const int a0 = pbase[p], b0 = pbase[p + 1];                     // multiples of 8
    const int units = (b0 - a0) >> 3, per = (units + splits - 1) / splits;
    const int a = a0 + part * per * 8;
    int b = a + per * 8;
    if (b > b0) b = b0;
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): this wave's pieces of the panel have landed
    __syncthreads();
    const int trips = (b - a + kProdThreads * 8 - 1) / (kProdThreads * 8);   // (counted on the scalar side, as in k_bs_products; no difference here)
    for (int trip = 0; trip < trips; ++trip) {
        const int k = a + trip * (kProdThreads * 8) + tid * 8;
        if (k >= b) continue;
        const u4 c = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(c16 + k));
        const f4 v0 = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(pvals + k));
        const f4 v1 = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(pvals + k + 4));
        f4 o0, o1;
        o0.x = xp[c.x & 0xffffu] * v0.x;
        o0.y = xp[c.x >> 16] * v0.y;
        o0.z = xp[c.y & 0xffffu] * v0.z;
        o0.w = xp[c.y >> 16] * v0.w;
        o1.x = xp[c.z & 0xffffu] * v1.x;
        o1.y = xp[c.z >> 16] * v1.y;
        o1.z = xp[c.w & 0xffffu] * v1.z;
        o1.w = xp[c.w >> 16] * v1.w;
        *reinterpret_cast<f4 *>(prod + k) = o0;
        *reinterpret_cast<f4 *>(prod + k + 4) = o1;
    }
}


// ==== the SCATTERED flavour (PanelPlan::scatter_mode): the products land in BIN-major order =============================
// The flavour above lets the sum launch FETCH tile (b, p) from the panel-major products: a piece of <= 64 E products of one
// tile per instruction group.  That is as good as the tiles are fat -- config 5's shard has 10-20 products per tile, so
// nine lanes in ten idle and every tile costs its table entries (4.1 ms against the sweep's 3.0).  Here the PRODUCT launch
// carries the permutation instead: it streams the panel-major entries (15-bit column + a "first of its tile" bit, value:
// 6 B) and stores each product at its bin-major position -- a tile is one contiguous RUN in both orders, so the position is
// the entry's own index plus its run's offset (4 B per run: run number = the block's first run + the flags counted up to
// the entry, a ballot per store), and the panel-major arrays are interleaved in blocks of 512 so that the 64 lanes of one
// store hold 64 CONSECUTIVE entries (16-byte loads per lane AND whole runs per store).  The sum launch is then a pure
// stream per bin: 4 B of product + a 16-bit ACCUMULATOR index, 256 entries per piece (four consecutive per lane), no tile
// table at all.  What keeps the read-add-write of a step free of collisions is decided when the plan is made: two entries
// of every lane form a step of 128; the j-th entry of a step that names a row already named in it gets the row's j-th
// spare accumulator (k_bs_accs), the spare accumulators join the row's own at the end of the bin.  A bin that needs more
// spare accumulators than there are (kPool) is flagged and adds with LDS atomics.
constexpr int kBlk = 512;                  // interleave block of the panel-major arrays: lane l of a wavefront holds entries l, 64 + l, ...
constexpr int kPool = 1024;                // spare accumulators of a bin (8192 + 1024 + 64 words: four wavefronts per CU)
constexpr int kBmPiece = 256;              // entries of the bin-major arrays per wavefront instruction group
constexpr int kRunBit = 1 << kPwBits;      // bit 15 of a panel-major column: this entry is the first of its run

__device__ __forceinline__ int interleaved(int q) { return (q & ~(kBlk - 1)) | ((q & 63) << 3) | ((q >> 6) & 7); }

// entries of every bin, rounded up to whole pieces (bbase is scanned afterwards)
__global__ void k_bs_counts(int nb, int np, const int32_t *__restrict__ tile_ptr, int32_t *__restrict__ bcnt, int32_t *__restrict__ bbase)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const int32_t *tp = tile_ptr + (int64_t)b * (np + 1);
    const int n = tp[np] - tp[0];
    bcnt[b] = n;
    bbase[b] = (n + kBmPiece - 1) & ~(kBmPiece - 1);
}


// Runs of the panel-major order: the nonempty tiles of a panel, bins ascending, then the panel's pad slots (one run slot per
// panel whether it has pad slots or not).  nruns[p] = the panel's run slots (scanned afterwards: first run of every panel).
__global__ __launch_bounds__(256) void k_bs_nruns(int nb, int np, const int32_t *__restrict__ tile_ptr, int32_t *__restrict__ nruns)
{
    __shared__ int part[256];
    const int p = blockIdx.x;
    int sum = 0;
    for (int b = threadIdx.x; b < nb; b += 256) {
        const int32_t *tp = tile_ptr + (int64_t)b * (np + 1);
        sum += tp[p + 1] > tp[p] ? 1 : 0;
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) nruns[p] = part[0] + 1;
}
// offset[run] = bin-major position - panel-major position of the run's entries; first_run[block of 512] = the run of the
// block's first entry (minus one where that entry starts a run: the product launch counts the flags INCLUSIVELY).  The pad
// slots of a panel are a run that lands behind the last bin (entries bm .. bm + 511: nothing reads them); its first slot
// gets the flag here (the fill kernel writes real entries only).  One workgroup per panel, bins scanned 256 at a time.
__global__ __launch_bounds__(256) void k_bs_runs(int nb, int np, int32_t bm, const int32_t *__restrict__ tile_ptr,
                                                 const int32_t *__restrict__ pm, const int32_t *__restrict__ pbase,
                                                 const int32_t *__restrict__ bbase, const int32_t *__restrict__ run0,
                                                 int32_t *__restrict__ offset, int32_t *__restrict__ first_run, uint16_t *__restrict__ c16)
{
    __shared__ int part[256];
    __shared__ int carry, last_end;
    const int p = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) { carry = 0; last_end = pbase[p]; }
    __syncthreads();
    const int r0 = run0[p];
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int b = b0 + tid;
        int c = 0, q0 = 0, d0 = 0;
        if (b < nb) {
            const int32_t *tp = tile_ptr + (int64_t)b * (np + 1);
            c = tp[p + 1] - tp[p];
            q0 = pm[(int64_t)b * np + p];
            d0 = bbase[b] + (tp[p] - tp[0]);
        }
        part[tid] = c > 0 ? 1 : 0;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int v = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (c > 0) {
            const int t = r0 + carry + part[tid] - 1;
            offset[t] = d0 - q0;
            for (int q = (q0 + kBlk - 1) & ~(kBlk - 1); q < q0 + c; q += kBlk) first_run[q / kBlk] = t - (q == q0 ? 1 : 0);
            atomicMax(&last_end, q0 + c);
        }
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
    if (tid == 0) {
        const int q0 = last_end, q1 = pbase[p + 1];                 // the pad slots
        if (q0 < q1) {
            const int t = r0 + carry;
            offset[t] = bm - q0;
            c16[interleaved(q0)] = (uint16_t)kRunBit;
            for (int q = (q0 + kBlk - 1) & ~(kBlk - 1); q < q1; q += kBlk) first_run[q / kBlk] = t - (q == q0 ? 1 : 0);
        }
    }
}

// k_bin_fill for this flavour, in one pass (SPMV_BS_FILL=1; the plan runs k_bs_group + k_bs_place below): rows at the bin-major
// (padded) position, the panel-major entry interleaved, the first entry of every tile flagged
__global__ __launch_bounds__(256) void k_bs_fill(int nb, int np, const int32_t *__restrict__ brow, const int32_t *__restrict__ row_ptr,
                                                 const int32_t *__restrict__ col_idx, const float *__restrict__ vals,
                                                 const uint16_t *__restrict__ rowloc, const int32_t *__restrict__ tile_ptr,
                                                 const int32_t *__restrict__ pm, const int32_t *__restrict__ bbase,
                                                 uint16_t *__restrict__ c16, float *__restrict__ pvals, uint16_t *__restrict__ acc)
{
    extern __shared__ int cursor_all[];     // 4 x np cursors, 4 x 256 tags
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    if (b >= nb) return;
    int *cursor = cursor_all + w * np;
    volatile int *tags = cursor_all + 4 * np + w * 256;
    const int32_t *tp = tile_ptr + (int64_t)b * (np + 1);
    const int32_t *qp = pm + (int64_t)b * np;
    for (int p = lane; p < np; p += kWave) cursor[p] = tp[p];
    const int s = row_ptr[brow[b]], e = row_ptr[brow[b + 1]];
    const int shift = bbase[b] - tp[0];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = s; base < e; base += kWave) {
        const int k = base + lane;
        const bool valid = k < e;
        int col = 0, rl = 0;
        float v = 0.0f;
        if (valid) {
            col = col_idx[k];
            rl = rowloc[k];
            v = vals[k];
        }
        const int p = col >> kPwBits;
        int dest = 0;
        const bool lone = lone_in_step(tags, valid, p, lane);       // the only nonzero of its panel in this step
        if (lone) {
            dest = cursor[p];
            cursor[p] = dest + 1;
        }
        unsigned long long todo = __ballot(valid && !lone);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int pl = __shfl(p, leader);
            const unsigned long long m = __ballot(valid && p == pl);
            const int first = cursor[pl];
            if (valid && p == pl) dest = first + __popcll(m & lt);
            if (lane == leader) cursor[pl] = first + __popcll(m);
            todo &= ~m;
        }
        if (valid) {
            acc[dest + shift] = (uint16_t)rl;                  // (k_bs_accs turns rows into accumulators)
            const int q = interleaved(qp[p] + (dest - tp[p]));
            c16[q] = (uint16_t)((col & (kPw - 1)) | (dest == tp[p] ? kRunBit : 0));   // the first entry of its tile starts a run
            pvals[q] = v;
        }
    }
}


// The fill in two passes (what plan_scatter runs; k_bs_fill above is the one-pass form, SPMV_BS_FILL=1).  One pass stores
// every nonzero of a bin at its tile -- 4096 tiles per bin at config 5's shard, 28 nonzeros each, filled over the whole
// life of the bin's wavefront: a 2-byte and a 4-byte store all over the panel-major arrays and a 2-byte store over the
// bin's region per nonzero, every line written a few bytes at a time (44 ms).  Two passes keep the writes of a tile close
// in time: (1) k_bs_group -- a wavefront per bin, the nonzeros into GROUPS of 64 panels inside the bin's own bin-major
// range (64 cursors: the lines being filled stay in the L2); (2) k_bs_place -- a wavefront per (group, bin), 64 tiles, a
// few hundred nonzeros: every tile is written within microseconds, by the XCD that also writes its neighbours in the panel.
// Inside a tile the nonzeros come in the order the LDS atomics of the two passes hand out places -- the same on every run,
// not the one-pass fill's ascending rows: this flavour needs no order inside a tile (k_bs_accs ranks what is there).
constexpr int kGroupBits = 6;              // panels per group: 64 (a lane's worth of cursors)
__global__ __launch_bounds__(256) void k_bs_group(int nb, int np, const int32_t *__restrict__ brow, const int32_t *__restrict__ row_ptr,
                                                  const int32_t *__restrict__ col_idx, const float *__restrict__ vals,
                                                  const uint16_t *__restrict__ rowloc, const int32_t *__restrict__ tile_ptr,
                                                  int32_t *__restrict__ tcol, float *__restrict__ tval, uint16_t *__restrict__ trow)
{
    __shared__ int cursor_all[4][64];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    if (b >= nb) return;
    int *cursor = cursor_all[w];
    const int32_t *tp = tile_ptr + (int64_t)b * (np + 1);
    const int ng = (np + 63) >> kGroupBits;                         // <= 64
    if (lane < ng) cursor[lane] = tp[lane << kGroupBits];
    const int s = row_ptr[brow[b]], e = row_ptr[brow[b + 1]];
    for (int base = s; base < e; base += kWave) {
        const int k = base + lane;
        const bool valid = k < e;
        int col = 0, rl = 0;
        float v = 0.0f;
        if (valid) {
            col = col_idx[k];
            rl = rowloc[k];
            v = vals[k];
        }
        const int g = col >> (kPwBits + kGroupBits);
        // (64 keys on 64 lanes: every key is held several times over and the ballot loop of the one-pass fill would take
        // ~40 trips per step -- an LDS atomic with return hands out the places; the order inside a group is then the order
        // the LDS serves the lanes in, the same on every run, and nothing downstream needs the rows of a tile ascending)
        const int dest = valid ? atomicAdd(&cursor[g], 1) : 0;
        if (valid) {
            tcol[dest] = col;
            tval[dest] = v;
            trow[dest] = (uint16_t)rl;
        }
    }
}
// pass 2: item (group g, bin b), b fastest, the items dealt to the XCDs in contiguous ranges.  src_*: the grouped copies of
// pass 1, or -- one group only -- the CSR arrays themselves (a bin's bin-major range IS its CSR range)
__global__ __launch_bounds__(256) void k_bs_place(int nb, int np, int nitems, const int32_t *__restrict__ src_col, const float *__restrict__ src_val,
                                                  const uint16_t *__restrict__ src_row, const int32_t *__restrict__ tile_ptr,
                                                  const int32_t *__restrict__ pm, const int32_t *__restrict__ bbase,
                                                  uint16_t *__restrict__ c16, float *__restrict__ pvals, uint16_t *__restrict__ acc)
{
    __shared__ int cursor_all[4][64];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int nwg = (int)gridDim.x, per8 = (nwg + 7) >> 3;
    const int wg = (int)(blockIdx.x & 7u) * per8 + (int)(blockIdx.x >> 3);   // (the grid is 8 per8 workgroups)
    const int item = wg * 4 + w;
    if (wg >= (nitems + 3) / 4 || item >= nitems) return;
    const int g = item / nb, b = item - g * nb;
    int *cursor = cursor_all[w];
    const int32_t *tp = tile_ptr + (int64_t)b * (np + 1);
    const int32_t *qp = pm + (int64_t)b * np;
    const int p0 = g << kGroupBits, p1 = p0 + 64 < np ? p0 + 64 : np;
    if (p0 + lane < p1) cursor[lane] = tp[p0 + lane];
    const int s = tp[p0], e = tp[p1];
    const int shift = bbase[b] - tp[0];
    for (int base = s; base < e; base += kWave) {
        const int k = base + lane;
        const bool valid = k < e;
        int col = 0, rl = 0;
        float v = 0.0f;
        if (valid) {
            col = src_col[k];
            rl = src_row[k];
            v = src_val[k];
        }
        const int p = col >> kPwBits, j = p - p0;
        int dest = valid ? atomicAdd(&cursor[j], 1) : 0;
        if (valid) {
            // The places come in CSR order, a row's nonzeros of the tile next to each other -- and a step of the sum launch takes
            // its 128 entries from a piece of 256 consecutive ones: a row with two nonzeros in the tile (rows of 33 and more over a
            // band of two million columns: a tenth of config 4's rows) would need a spare accumulator, a row of 32 in the tile 31
            // (config 4, band 2M: every bin over its 1024 and on the atomics, 2.3 ms).  Neighbours are dealt a piece apart
            // instead: the first 256 M entries of the tile (M = its entries / 256) as a 256 x M grid written column by column
            const int t0 = tp[p], len = tp[p + 1] - t0, m = len >> 8;
            const int r = dest - t0;
            if (m >= 2 && r < (m << 8)) {
                const int a = (int)((unsigned)r / (unsigned)m), bq = r - a * m;
                dest = t0 + (bq << 8) + a;
            }
            acc[dest + shift] = (uint16_t)rl;                  // (k_bs_accs turns rows into accumulators)
            const int q = interleaved(qp[p] + (dest - tp[p]));
            c16[q] = (uint16_t)((col & (kPw - 1)) | (dest == tp[p] ? kRunBit : 0));   // the first entry of its tile starts a run
            pvals[q] = v;
        }
    }
}

// Accumulators of one bin (a wavefront per bin).  A STEP of the sum launch holds 128 entries of the bin: entries 4 l + 2 h and
// 4 l + 2 h + 1 of piece (256 entries), l = 0..63, h = 0 | 1 -- the launch reads the 128 accumulators, adds, writes them back,
// so no accumulator may be named twice in a step.  occ = how many entries before this one in its step (first entries of the
// lanes, then second entries, lanes ascending) name the same row.  Pass A: the deepest occ of every row; then the spare
// accumulators are dealt out (occ of them per row) and the bin's list written; pass B: entry -> row (occ 0) or the row's spare
// accumulator number occ - 1.  LDS: tag[RB] (who wrote last) and deep[RB] (max occ, then the first spare).
template <int RB>
__global__ __launch_bounds__(kWave) void k_bs_accs(const int32_t *__restrict__ brow, const int32_t *__restrict__ bbase,
                                                   const int32_t *__restrict__ bcnt, uint16_t *__restrict__ acc,
                                                   int32_t *__restrict__ nlong, uint32_t *__restrict__ lrow, int32_t *__restrict__ stats)
{
    extern __shared__ int lds_i[];
    int *tag = lds_i, *deep = lds_i + RB;
    const int lane = threadIdx.x, b = blockIdx.x;
    const int nrows = brow[b + 1] - brow[b];
    const int base = bbase[b], n = bcnt[b];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int i = lane; i < RB; i += kWave) deep[i] = 0;
    __builtin_amdgcn_wave_barrier();
    // lanes of one instruction: how many lower lanes name the same row
    auto among = [&](bool valid, int r) {
        if (valid) reinterpret_cast<volatile int *>(tag)[r] = lane;
        __builtin_amdgcn_wave_barrier();
        const int wtag = valid ? reinterpret_cast<volatile int *>(tag)[r] : lane;
        unsigned long long losers = __ballot(valid && wtag != lane);
        int occ = 0;
        while (losers) {
            const int leader = __ffsll((long long)losers) - 1;
            const int rl = __shfl(r, leader);
            const unsigned long long m = __ballot(valid && r == rl);
            if (valid && r == rl) occ = __popcll(m & lt);
            losers &= ~m;
        }
        return occ;
    };
    // the two entries of a lane in one step: occ of both (the second ones count every first one of their row)
    auto occurrences = [&](bool v0, int r0, bool v1, int r1, int &occ0, int &occ1) {
        occ0 = among(v0, r0);                                       // (tag[r0] = some lane whose first entry names r0)
        __builtin_amdgcn_wave_barrier();
        const int t = v1 ? reinterpret_cast<volatile int *>(tag)[r1] & (kWave - 1) : 0;   // stale or never written: checked next
        const int rt = __shfl(v0 ? r0 : -1, t);                     // (every lane takes part: the lane asked may have no second entry)
        const bool cross = v1 && rt == r1;                          // a first entry of this step names my row
        int firsts = 0;
        unsigned long long todo = __ballot(cross);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int rl = __shfl(r1, leader);
            const int c = __popcll(__ballot(v0 && r0 == rl));
            const unsigned long long m = __ballot(v1 && r1 == rl);
            if (v1 && r1 == rl) firsts = c;
            todo &= ~m;
        }
        __builtin_amdgcn_wave_barrier();
        occ1 = firsts + among(v1, r1);
    };
    auto entry = [&](int e0, int h, int j, bool &valid, int &r) {
        const int i = e0 + 4 * lane + 2 * h + j;
        valid = i < n;
        r = valid ? (int)acc[base + i] : 0;
        return base + i;
    };
    int deepest = 0;
    for (int e0 = 0; e0 < n; e0 += kBmPiece)
        for (int h = 0; h < 2; ++h) {
            bool v0, v1;
            int r0, r1, occ0, occ1;
            entry(e0, h, 0, v0, r0);
            entry(e0, h, 1, v1, r1);
            occurrences(v0, r0, v1, r1, occ0, occ1);
            if (occ0 > 0) atomicMax(&deep[r0], occ0);
            if (occ1 > 0) atomicMax(&deep[r1], occ1);
            deepest = occ0 > deepest ? occ0 : deepest;
            deepest = occ1 > deepest ? occ1 : deepest;
        }
    if (__ballot(deepest > 0) == 0ull) {                            // no row twice in any step
        if (lane == 0) nlong[b] = 0;
        return;
    }
    __builtin_amdgcn_wave_barrier();
    // the spare accumulators, rows ascending; the list: row << 17 | first << 7 | count (at most 127)
    int used = 0, listed = 0;
    bool over = false;
    for (int r0 = 0; r0 < nrows && !over; r0 += kWave) {
        const int r = r0 + lane;
        const int need = r < nrows ? deep[r] : 0;
        int pre = need;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int t = __shfl_up(pre, d);
            if (lane >= d) pre += t;
        }
        const int total = __shfl(pre, kWave - 1);
        if (used + total > kPool) { over = true; break; }
        const int first = used + pre - need;
        const unsigned long long lm = __ballot(need > 0);
        if (need > 0) {
            deep[r] = first;
            lrow[(int64_t)b * kPool + listed + __popcll(lm & lt)] = ((uint32_t)r << 17) | ((uint32_t)first << 7) | (uint32_t)need;
        }
        used += total;
        listed += __popcll(lm);
    }
    if (over) {                                                     // flagged: the rows stay rows, the sum launch adds with atomics
        if (lane == 0) { nlong[b] = -1; atomicAdd(&stats[0], 1); }
        return;
    }
    if (lane == 0) { nlong[b] = listed; atomicAdd(&stats[1], listed); }
    __builtin_amdgcn_wave_barrier();
    for (int e0 = 0; e0 < n; e0 += kBmPiece)
        for (int h = 0; h < 2; ++h) {
            bool v0, v1;
            int r0, r1, occ0, occ1;
            const int i0 = entry(e0, h, 0, v0, r0);
            const int i1 = entry(e0, h, 1, v1, r1);
            occurrences(v0, r0, v1, r1, occ0, occ1);
            if (occ0 > 0) acc[i0] = (uint16_t)(RB + deep[r0] + occ0 - 1);
            if (occ1 > 0) acc[i1] = (uint16_t)(RB + deep[r1] + occ1 - 1);
        }
}

// launch 1 of this flavour: products in panel order, stored in bin order
__global__ __launch_bounds__(kProdThreads) void k_bs_products(int splits, int np, int64_t cols, const int32_t *__restrict__ pbase,
                                                              const uint16_t *__restrict__ c16, const float *__restrict__ pvals,
                                                              const int32_t *__restrict__ first_run, const int32_t *__restrict__ offset,
                                                              const float *__restrict__ x, float *__restrict__ prod)
{
    extern __shared__ __attribute__((aligned(16))) float xp[];      // the panel: kPw floats
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    int p = blockIdx.x / splits;
    const int part = blockIdx.x - p * splits;
#ifndef SPMV_BS_AB_NOXCD
    if (splits == 1) {
        // Workgroups go round the eight XCDs; an XCD takes the panels in CHUNKS of 32 consecutive ones (what its 32 CUs run at
        // a time), the chunks dealt round the XCDs.  Tiles (b, p) and (b, p + 1) are neighbours in the bin-major products and
        // share a 128-byte line: written through ONE L2 a few microseconds apart the line leaves it whole; written through two,
        // each L2 writes its part back under a byte mask (config 5's shard: 1.63 -> 1.36 ms,
        // profiles/r04_scattered_products_xcd.jsonl).  (First form: an XCD took a contiguous EIGHTH of the panels -- on a banded
        // shard whose nonzeros sit in the first eighth of the columns one XCD did all the work: 3.0 ms at config 5's shard with
        // a band of 200 000 ... 4M columns.)
        const int x8 = (int)(blockIdx.x & 7u), i = (int)(blockIdx.x >> 3);
        p = (((i >> 5) << 3) + x8) * 32 + (i & 31);
        if (p >= np) return;                                        // (the grid: 8 XCDs x 32 x ceil(np / 256) workgroups)
    }
#endif
    if (pbase[p] == pbase[p + 1]) return;                           // (an empty panel: nothing to stage x for)
    const int64_t g0 = (int64_t)p << kPwBits;
    if (g0 + kPw + 3 < cols) {                                      // workgroup-uniform
#pragma unroll
        for (int i = tid * 4; i < kPw; i += kProdThreads * 4) __builtin_amdgcn_global_load_lds(x + g0 + i, xp + i, 16, 0, 0);
    } else {
        for (int i = tid; i < kPw; i += kProdThreads) xp[i] = g0 + i < cols ? x[g0 + i] : 0.0f;
    }
    const int a0 = pbase[p], b0 = pbase[p + 1];                     // multiples of kBlk
    const int units = (b0 - a0) / kBlk, per = (units + splits - 1) / splits;
    const int a = a0 + part * per * kBlk;
    int b = a + per * kBlk;
    if (b > b0) b = b0;
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): this wave's pieces of the panel have landed
    __syncthreads();
    const unsigned long long le = (2ull << lane) - 1ull;            // this lane and the ones below
    // The trips are counted on the scalar side (a, b and a trip are multiples of 512: a wavefront is in or out as a whole): with
    // the lane's own k as the loop variable the compiler kept the loop under an exec mask and the product launch ran 903 us
    // where it now runs 720 (config 5's shard).  Every workgroup walks its panel bins ascending and the workgroups of an XCD
    // reach the same bins at about the same time -- which is what completes the shared lines in the L2: with the panels
    // started at different trips of the walk (tried against a suspected channel hot spot) the launch ran 923 us.
    // profiles/r04_scattered_products_ab_kernel_times.jsonl (the A/B builds behind it: the tree of commit 46593cb)
    const int trips = (b - a + kProdThreads * 8 - 1) / (kProdThreads * 8);
    for (int trip = 0; trip < trips; ++trip) {                      // a wavefront: one interleave block per trip
        const int k = a + trip * (kProdThreads * 8) + tid * 8;
        if (k >= b) continue;
        const int blk = __builtin_amdgcn_readfirstlane(k >> 9);
        const int t0 = first_run[blk];
        const u4 c = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(c16 + k));
        const f4 v0 = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(pvals + k));
        const f4 v1 = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(pvals + k + 4));
        const unsigned cw[8] = {c.x & 0xffffu, c.x >> 16, c.y & 0xffffu, c.y >> 16, c.z & 0xffffu, c.z >> 16, c.w & 0xffffu, c.w >> 16};
        const float vw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        int run = t0;                                               // (scalar) runs begun before store j's entries, from t0
        const int q = (k & ~(kBlk - 1)) + lane;                     // entry of store 0; store j: q + 64 j
#pragma unroll
        for (int j = 0; j < 8; ++j) {                               // store j of the 64 lanes: 64 consecutive entries of the panel
            const unsigned long long starts = __ballot((cw[j] & (unsigned)kRunBit) != 0u);
            const int d = offset[run + __popcll(starts & le)];
            run += __popcll(starts);
            const float o = xp[cw[j] & (unsigned)(kPw - 1)] * vw[j];
#ifdef SPMV_BS_NT_STORE
            __builtin_nontemporal_store(o, prod + (q + 64 * j + d));
#else
            prod[q + 64 * j + d] = o;
#endif
        }
    }
}

// launch 2 of this flavour: a wavefront per bin, its products and accumulator numbers one contiguous stream
template <int RB>
__global__ __launch_bounds__(kWave) void k_bs_sums(const int32_t *__restrict__ brow, const int32_t *__restrict__ bbase,
                                                   const int32_t *__restrict__ bcnt, const uint16_t *__restrict__ acc,
                                                   const float *__restrict__ prod, const int32_t *__restrict__ nlong,
                                                   const uint32_t *__restrict__ lrow, float *__restrict__ y, uint32_t prod_bytes,
                                                   uint32_t acc_bytes)
{
#ifndef SPMV_BS_DEPTH
#define SPMV_BS_DEPTH 4
#endif
#ifndef SPMV_BS_LOAD_AUX
#define SPMV_BS_LOAD_AUX 2      // cache-policy bits of the sum launch's loads: non-temporal (0: 398 -> 360 us at config 5's shard; the products are read once)
#endif
    constexpr int kDummy = RB + kPool, kD = SPMV_BS_DEPTH;           // pieces per register set (two sets in flight)
    extern __shared__ float sums[];                                 // RB + kPool + kWave
    const int lane = threadIdx.x, b = blockIdx.x;
    const int row0 = brow[b], nrows = brow[b + 1] - row0;
    const int base = bbase[b], pieces = (bcnt[b] + kBmPiece - 1) / kBmPiece, nl = nlong[b];
    for (int i = lane; i < kDummy; i += kWave) sums[i] = 0.0f;
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(prod), 0, (int)prod_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(acc), 0, (int)acc_bytes, 0x00020000);
    using u4v = decltype(__builtin_amdgcn_raw_buffer_load_b128(pr, 0, 0, 0));
    using u2v = decltype(__builtin_amdgcn_raw_buffer_load_b64(rr, 0, 0, 0));
    struct Set { u4v v[kD]; u2v a[kD]; };                           // as loaded: unpacking here would wait for the loads at once
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // (the table numbers are in: the waits below count piece loads only)
    auto issue = [&](int pi0, Set &c) {                             // no branch: a piece past the end reads on (slack) and counts for nothing
#pragma unroll
        for (int u = 0; u < kD; ++u) {
            const unsigned at = (unsigned)base + (unsigned)(pi0 + u) * kBmPiece;
            c.v[u] = __builtin_amdgcn_raw_buffer_load_b128(pr, lane * 16, (int)(at * 4u), SPMV_BS_LOAD_AUX);
            c.a[u] = __builtin_amdgcn_raw_buffer_load_b64(rr, lane * 8, (int)(at * 2u), SPMV_BS_LOAD_AUX);
            // the SAME order of loads before the loop and inside it: the wait counts are static, and a first trip that needs
            // "everything" (the scheduler had turned the sets round) makes every trip wait for everything
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto consume = [&](int pi0, Set &c, auto flagged) {
#pragma unroll
        for (int u = 0; u < kD; ++u) {
            const bool live = pi0 + u < pieces;
#pragma unroll
            for (int h = 0; h < 2; ++h) {                           // a step: two entries of every lane, 128 accumulators, none twice (k_bs_accs)
                const unsigned w = c.a[u][h];
                const int a0 = live ? (int)(w & 0xffffu) : kDummy, a1 = live ? (int)(w >> 16) : kDummy + 1;
                const float v0 = __uint_as_float(c.v[u][2 * h]), v1 = __uint_as_float(c.v[u][2 * h + 1]);
                if (decltype(flagged)::value) {
                    atomicAdd(&sums[a0], v0);
                    atomicAdd(&sums[a1], v1);
                } else {
                    const float s0 = sums[a0], s1 = sums[a1];       // both reads before either write: one LDS round trip per 128 entries
                    sums[a0] = s0 + v0;
                    sums[a1] = s1 + v1;
                }
            }
        }
    };
    // two register sets, both in flight at the top of the loop (as in k_bin_sums)
    auto stream = [&](auto flagged) {
        Set sa, sb;
        __builtin_amdgcn_sched_barrier(0);
        issue(0, sa);
        issue(kD, sb);
        for (int pi0 = 0; pi0 < pieces; pi0 += 2 * kD) {
            // (the scheduler is kept from moving a set's loads into the other set's sums: the waits would then drain both)
            __builtin_amdgcn_sched_barrier(0);
            consume(pi0, sa, flagged);
            __builtin_amdgcn_sched_barrier(0);
            issue(pi0 + 2 * kD, sa);
            __builtin_amdgcn_sched_barrier(0);
            consume(pi0 + kD, sb, flagged);
            __builtin_amdgcn_sched_barrier(0);
            issue(pi0 + 3 * kD, sb);
        }
    };
    if (nl >= 0) stream(std::false_type());
    else stream(std::true_type());                                  // flagged bin (wave-uniform): LDS atomics
    for (int i = lane; i < nl; i += kWave) {                        // the spare accumulators join their rows, in order
        const uint32_t w = lrow[(int64_t)b * kPool + i];
        const int row = (int)(w >> 17), first = RB + (int)((w >> 7) & 0x3ffu), cnt = (int)(w & 127u);
        float t = sums[row];
        for (int c = 0; c < cnt; ++c) t += sums[first + c];
        sums[row] = t;
    }
    for (int i = lane; i < nrows; i += kWave) y[row0 + i] = sums[i];
}

// ---- launch 2: sums per bin -----------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// lane l receives lane l-1's value (lane 0: `first`): DPP wave_shr:1, one VALU instruction instead of a trip through the LDS crossbar
__device__ __forceinline__ int shift_up1(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false); }

// The general fold (flagged tiles only): 64 (cnt) products of one tile, one per lane, rows ascending; equal rows are
// neighbouring lanes; their products are folded into the LAST lane of the run by a segmented scan, `row` becomes -1 on the
// lanes that have nothing to add any more.
__device__ __forceinline__ void fold_piece(int lane, int cnt, int &row, float &v)
{
    const bool valid = lane < cnt;
    if (!valid) row = -1 - lane;                                    // never equal to a neighbour
    const int prow = shift_up1(row, -2);
    const bool dup = valid && prow == row;
    const unsigned long long dm = __ballot(dup);
    if (dm != 0ull) {
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {                       // rows ascend: equal at distance d = one run
            const float tv = __shfl_up(v, d);
            const int tr = __shfl_up(row, d);
            if (lane >= d && tr == row) v += tv;
        }
        if (lane < kWave - 1 && ((dm >> (lane + 1)) & 1ull)) row = -1;   // not the last of its run
    }
    if (!valid) row = -1;
}

// A piece = up to 64 E consecutive products of ONE tile, E consecutive per lane (E = 4: a 16-byte load of products per
// lane; E = 2 for thin tiles).  The lane adds its E products into the sums one after the other, entry k of every lane in
// step k: rows ascend along the piece, so two products of one row are either in one lane (different steps) or in
// neighbouring lanes at its last and first entry (different steps again) -- as long as no row holds more than E products
// of the tile, no step ever holds a row twice, and the add is a plain LDS read-add-write with NO fold logic at all.  The
// plan flags the few tiles where a row does hold more (bit 31 of their pm entry): those go, 64 products at a time, one per
// lane, through the segmented scan above.  Dead entries (past the tile's end) add into a dummy word behind the sums.
#ifndef SPMV_BIN_LOAD_AUX
#define SPMV_BIN_LOAD_AUX 0     // cache-policy bits of the piece loads (A/B builds: 2 = non-temporal)
#endif
template <int RB, int E>
__global__ __launch_bounds__(kSumWaves *kWave) void k_bin_sums(int nb, int np, const int32_t *__restrict__ brow,
                                                               const int32_t *__restrict__ tile_ptr, const int32_t *__restrict__ pm,
                                                               const uint16_t *__restrict__ r16, const float *__restrict__ prod,
                                                               const int32_t *__restrict__ lptr, const uint32_t *__restrict__ lrow,
                                                               const int32_t *__restrict__ lcnt, float *__restrict__ y,
                                                               uint32_t prod_bytes, uint32_t r16_bytes)
{
    constexpr int kPiece = kWave * E, kShift = E == 4 ? 8 : 7, kD = E == 4 ? 4 : 8;   // pieces per register set
    constexpr int kStride = RB + kSpare + kWave;                    // a wave's sums, the spare sums of its long rows, the dummy word
    constexpr int kDummy = RB + kSpare;
    extern __shared__ float sums_all[];                             // kSumWaves x kStride
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * kSumWaves + wave;
    if (b >= nb) return;                                            // (no barrier below: a wave's sums are its own)
    float *sums = sums_all + wave * kStride;
    const int row0 = brow[b], nrows = brow[b + 1] - row0;
    for (int i = lane; i < kDummy; i += kWave) sums[i] = 0.0f;
    const int32_t *tp = tile_ptr + (int64_t)b * (np + 1);
    const int32_t *qp = pm + (int64_t)b * np;
    struct Set { float v[kD][E]; int r[kD][E]; int left[kD]; int q[kD], s[kD]; };
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(prod), 0, (int)prod_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(r16), 0, (int)r16_bytes, 0x00020000);
    for (int tb = 0; tb < np; tb += kWave) {                        // 64 tiles of the bin: a lane holds one tile's three numbers
        const int t = tb + lane;
        int s_l = 0, n_l = 0, q_l = 0;
        if (t < np) {
            s_l = tp[t];
            n_l = tp[t + 1] - s_l;
            q_l = qp[t];
        }
        const int c_l = (n_l + kPiece - 1) >> kShift;               // pieces of this tile
        const int pre = wave_inclusive_scan(c_l, lane);
        const int total = __builtin_amdgcn_readlane(pre, kWave - 1);
        const int excl = pre - c_l;
        // the three table numbers are in registers before the first piece is addressed: the waits further down then count
        // piece loads only (a wait for a table load issued on another control path would drain every piece in flight)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        // the loads of kD pieces, starting at piece ci0 of this batch of tiles.  No branch: a piece past the end loads a valid
        // address and counts for nothing (left = 0); lanes past a tile's end read on into the next tile (the arrays have
        // the slack) and add into the dummy word -- every path issues the same loads, the compiler's wait counters stay exact
        // and one register set really is in flight under the other
        auto issue = [&](int ci0, Set &c) {
#pragma unroll
            for (int u = 0; u < kD; ++u) {
                const int ci = ci0 + u;
                const bool live = ci < total;
                int tl = __popcll(__ballot(pre <= ci));             // the tile this piece belongs to: first lane with pre > ci
                tl = tl > kWave - 1 ? kWave - 1 : tl;
                const int s = __builtin_amdgcn_readlane(s_l, tl), n = __builtin_amdgcn_readlane(n_l, tl);
                const int q = __builtin_amdgcn_readlane(q_l, tl), ex = __builtin_amdgcn_readlane(excl, tl);
                const int off = live ? (ci - ex) << kShift : 0;
                c.left[u] = live ? n - off : 0;
                c.q[u] = q + off;                                   // (bit 31 of q: the tile is flagged)
                c.s[u] = s + off;
                // buffer loads: the piece's position is the SCALAR offset, the lane's share a constant vector offset -- no
                // address arithmetic per piece on the vector side (reads past the arrays' ends return 0)
                const unsigned qb = (unsigned)((q & 0x7fffffff) + off) * 4u, sb = (unsigned)(s + off) * 2u;
                if (E == 4) {
                    const auto pv = __builtin_amdgcn_raw_buffer_load_b128(pr, lane * 16, (int)qb, SPMV_BIN_LOAD_AUX);
                    c.v[u][0] = __uint_as_float(pv[0]); c.v[u][1] = __uint_as_float(pv[1]);
                    c.v[u][2] = __uint_as_float(pv[2]); c.v[u][3] = __uint_as_float(pv[3]);
                } else {
                    const auto pv = __builtin_amdgcn_raw_buffer_load_b64(pr, lane * 8, (int)qb, SPMV_BIN_LOAD_AUX);
                    c.v[u][0] = __uint_as_float(pv[0]); c.v[u][1] = __uint_as_float(pv[1]);
                }
                // (the rows of a lane: one load; a tile starts at any entry, so the address is 2-byte aligned only -- the
                // memory pipeline takes that, as it does for the global loads the compiler makes of a uint16_t pointer)
                if (E == 4) {
                    const auto rv = __builtin_amdgcn_raw_buffer_load_b64(rr, lane * 8, (int)sb, SPMV_BIN_LOAD_AUX);
                    c.r[u][0] = (int)(rv[0] & 0xffffu); c.r[u][1] = (int)(rv[0] >> 16);
                    c.r[u][2] = (int)(rv[1] & 0xffffu); c.r[u][3] = (int)(rv[1] >> 16);
                } else {
                    const unsigned rv = __builtin_amdgcn_raw_buffer_load_b32(rr, lane * 4, (int)sb, SPMV_BIN_LOAD_AUX);
                    c.r[u][0] = (int)(rv & 0xffffu); c.r[u][1] = (int)(rv >> 16);
                }
            }
        };
        auto consume = [&](Set &c) {
#pragma unroll
            for (int u = 0; u < kD; ++u) {
                const int left = c.left[u];
                if (c.q[u] < 0 && left > 0) {                       // flagged tile: one product per lane, folded (rare)
                    const int q = c.q[u] & 0x7fffffff, s = c.s[u];
                    const int m = left < kPiece ? left : kPiece;
                    for (int o = 0; o < m; o += kWave) {
                        const int cnt = m - o < kWave ? m - o : kWave;
                        int row = lane < cnt ? (int)r16[s + o + lane] : 0;
                        float v = lane < cnt ? prod[q + o + lane] : 0.0f;
                        fold_piece(lane, cnt, row, v);
                        if (row >= 0) sums[row] += v;
                    }
                    continue;
                }
#pragma unroll
                for (int k = 0; k < E; ++k) {                       // step k: entry k of every lane; a row is there at most once
                    const int a = E * lane + k < left ? c.r[u][k] : kDummy;
                    sums[a] += c.v[u][k];
                }
            }
        };
        // two register sets, both in flight at the top of the loop: a set is summed, then re-issued at once (its registers
        // are free), while the other set's loads are still out -- the waits inside consume() count the younger set's loads
        Set sa, sb;
        issue(0, sa);
        issue(kD, sb);
        for (int ci0 = 0; ci0 < total; ci0 += 2 * kD) {
            consume(sa);
            issue(ci0 + 2 * kD, sa);
            consume(sb);
            issue(ci0 + 3 * kD, sb);
        }
    }
    // the long rows of the bin: their spare sums join the row's own, in slot order
    for (int i = lptr[b] + lane; i < lptr[b + 1]; i += kWave) {
        const uint32_t w = lrow[i];
        const int row = (int)(w >> 16), base = RB + (int)(w & 0xffffu), n = lcnt[i];
        float t = sums[row];
        for (int c = 0; c < n; ++c) t += sums[base + c];
        sums[row] = t;
    }
    for (int i = lane; i < nrows; i += kWave) y[row0 + i] = sums[i];
}

// ---- plan: long rows --------------------------------------------------------------------------------------------------
// A row that holds m > E products of some tile would meet itself in a step of the sum launch.  Its products number
// E, E+1, ... of that tile go to SPARE sums instead (slot j / E - 1 of the row's spare sums: again E consecutive products per
// slot, never two in one step), and the spare sums join the row's own at the end of the bin.
// pass 1: maxrun[row] = the most products any tile holds of the row (where more than E); one thread per tile
__global__ __launch_bounds__(256) void k_bin_runs(int np, int e, const int32_t *__restrict__ brow, const int32_t *__restrict__ tile_ptr,
                                                  const uint16_t *__restrict__ r16, int32_t *__restrict__ maxrun)
{
    const int b = blockIdx.x, row0 = brow[b];
    const int32_t *tp = tile_ptr + (int64_t)b * (np + 1);
    for (int p = threadIdx.x; p < np; p += 256) {
        const int s0 = tp[p], s1 = tp[p + 1];
        int i = s0;
        while (i < s1) {
            const int r = r16[i];
            int j = i + 1;
            while (j < s1 && r16[j] == r) ++j;
            if (j - i > e) atomicMax(&maxrun[row0 + r], j - i);
            i = j;
        }
    }
}
// pass 2: per bin, the spare sums of its long rows: maxrun[row] becomes the row's first spare slot (-1: none); spare[b] =
// slots the bin needs (a bin that needs more than kSpare is flagged whole: its tiles go through the fold); nlong[b] = its long rows
__global__ __launch_bounds__(256) void k_bin_spare(int e, const int32_t *__restrict__ brow, int32_t *__restrict__ maxrun,
                                                   int32_t *__restrict__ spare, int32_t *__restrict__ nlong)
{
    __shared__ int part[256], part2[256];
    __shared__ int carry, carry2;
    const int b = blockIdx.x, tid = threadIdx.x, row0 = brow[b], nrows = brow[b + 1] - row0;
    if (tid == 0) { carry = 0; carry2 = 0; }
    __syncthreads();
    for (int r0 = 0; r0 < nrows; r0 += 256) {
        const int r = r0 + tid;
        const int m = r < nrows ? maxrun[row0 + r] : 0;
        const int ex = m > e ? (m + e - 1) / e - 1 : 0;
        part[tid] = ex;
        part2[tid] = ex > 0;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int v = tid >= d ? part[tid - d] : 0, v2 = tid >= d ? part2[tid - d] : 0;
            __syncthreads();
            part[tid] += v;
            part2[tid] += v2;
            __syncthreads();
        }
        if (r < nrows) maxrun[row0 + r] = ex > 0 ? (carry + part[tid] - ex) | (ex << 16) : -1;   // first slot | slots << 16
        __syncthreads();
        if (tid == 255) { carry += part[255]; carry2 += part2[255]; }
        __syncthreads();
    }
    if (tid == 0) { spare[b] = carry; nlong[b] = carry > kSpare ? 0 : carry2; }
}
// pass 3: the rows of the products that go to spare sums are rewritten (bins within the budget); the tiles of the others flagged;
// the bin's list of long rows (row << 16 | first slot, slots) written from lptr[b] on, rows ascending
__global__ __launch_bounds__(256) void k_bin_rewrite(int np, int e, int rb, const int32_t *__restrict__ brow, const int32_t *__restrict__ tile_ptr,
                                                     const int32_t *__restrict__ slot, const int32_t *__restrict__ spare,
                                                     const int32_t *__restrict__ lptr, uint16_t *__restrict__ r16, int32_t *__restrict__ pm,
                                                     uint32_t *__restrict__ lrow, int32_t *__restrict__ lcnt, int32_t *__restrict__ flagged)
{
    __shared__ int cursor;
    const int b = blockIdx.x, row0 = brow[b], nrows = brow[b + 1] - row0;
    const int32_t *tp = tile_ptr + (int64_t)b * (np + 1);
    const bool over = spare[b] > kSpare;
    if (over) {                                                     // more long rows than spare sums: every tile with a long run folds
        for (int p = threadIdx.x; p < np; p += 256) {
            const int s0 = tp[p], s1 = tp[p + 1];
            bool hit = false;
            for (int i = s0; i + e < s1 && !hit; ++i) hit = r16[i] == r16[i + e];
            if (hit) {
                pm[(int64_t)b * np + p] |= (int32_t)0x80000000;
                atomicAdd(flagged, 1);
            }
        }
        return;
    }
    if (spare[b] == 0) return;
    for (int p = threadIdx.x; p < np; p += 256) {
        const int s0 = tp[p], s1 = tp[p + 1];
        int i = s0;
        while (i < s1) {
            const int r = r16[i];
            int j = i + 1;
            while (j < s1 && r16[j] == r) ++j;
            if (j - i > e) {
                const int first = slot[row0 + r] & 0xffff;
                for (int k = i + e; k < j; ++k) r16[k] = (uint16_t)(rb + first + (k - i) / e - 1);
            }
            i = j;
        }
    }
    // the list, in row order (one thread: a bin has a handful of long rows)
    if (threadIdx.x == 0) {
        int at = lptr[b];
        for (int r = 0; r < nrows; ++r) {
            const int w = slot[row0 + r];
            if (w >= 0) {
                lrow[at] = ((uint32_t)r << 16) | (uint32_t)(w & 0xffff);
                lcnt[at] = w >> 16;
                ++at;
            }
        }
        cursor = at;
    }
    (void)cursor;
}

}  // namespace

void destroy_binned(PanelPlan &p)
{
    if (p.d_c16) (void)hipFree(p.d_c16);
    if (p.d_r16) (void)hipFree(p.d_r16);
    if (p.d_pm) (void)hipFree(p.d_pm);
    if (p.d_pbase) (void)hipFree(p.d_pbase);
    if (p.d_prod) (void)hipFree(p.d_prod);
    if (p.d_lptr) (void)hipFree(p.d_lptr);
    if (p.d_lrow) (void)hipFree(p.d_lrow);
    if (p.d_lcnt) (void)hipFree(p.d_lcnt);
    if (p.d_offset) (void)hipFree(p.d_offset);
    if (p.d_first_run) (void)hipFree(p.d_first_run);
    if (p.d_bbase) (void)hipFree(p.d_bbase);
    if (p.d_bcnt) (void)hipFree(p.d_bcnt);
    if (p.d_nlong) (void)hipFree(p.d_nlong);
    p.d_offset = p.d_first_run = p.d_bbase = p.d_bcnt = p.d_nlong = nullptr;
    p.scatter_mode = false;
    p.d_lptr = p.d_lcnt = nullptr;
    p.d_lrow = nullptr;
    p.d_c16 = p.d_r16 = nullptr;
    p.d_pm = p.d_pbase = nullptr;
    p.d_prod = nullptr;
    p.binned_mode = false;
}

// mean nonzeros of a (bin, panel) tile: what a piece of the sum launch can hope to hold
double binned_tile_nonzeros(const spmv_csr &h, int bin_rows)
{
    if (h.rows <= 0 || h.cols <= 0) return 0.0;
    const double nb = (double)((h.rows + bin_rows - 1) / bin_rows), np = (double)((h.cols + kPw - 1) >> kPwBits);
    return (double)h.nnz / (nb * np);
}


// the rest of plan_binned for the scattered flavour: bin-major positions, the fill, the accumulators
static int plan_scatter(spmv_csr &h, PanelPlan &p, int rb, int32_t padded, DevPtr<int32_t> &brow, DevPtr<int32_t> &tiles,
                        DevPtr<int32_t> &pm, DevPtr<int32_t> &pbase, hipStream_t s)
{
    const int nb = p.nblocks, np = p.npanels;
    int rc;
    DevPtr<int32_t> bbase, bcnt, total, nlong, stats, run0, offset, first_run;
    DevPtr<uint32_t> lrow;
    DevPtr<uint16_t> rowloc, c16, acc;
    DevPtr<float> pvals, prod;
    SPMV_HIP_TRY(bbase.alloc((size_t)nb + 1));
    SPMV_HIP_TRY(bcnt.alloc((size_t)nb));
    SPMV_HIP_TRY(total.alloc(1));
    k_bs_counts<<<dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s>>>(nb, np, tiles.p, bcnt.p, bbase.p);
    if ((rc = check("k_bs_counts"))) return rc;
    if ((rc = exclusive_scan_i32(bbase.p, nb, total.p, s))) return rc;
    int32_t bm = 0;
    SPMV_HIP_TRY(hipMemcpyAsync(&bm, total.p, sizeof bm, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipMemcpyAsync(bbase.p + nb, total.p, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));
    if (bm < 0 || (int64_t)bm > (1ll << 30) - 1024) {   // ((bm + 528) * 4 bytes must fit the buffer descriptors' 32 bits)
        set_error("spmv_csr_plan(panel, binned, scattered products): %d bins pad nnz %lld beyond 2^30 entries", nb, (long long)h.nnz);
        return SPMV_ERR_INVALID;
    }
    const size_t nslot = (size_t)padded + 8, bslot = (size_t)bm + kBlk + 16;       // (entries bm ...: where the pad slots of the panels store)
    SPMV_HIP_TRY(c16.alloc(nslot));
    SPMV_HIP_TRY(pvals.alloc(nslot));
    SPMV_HIP_TRY(prod.alloc(bslot));
    SPMV_HIP_TRY(acc.alloc(bslot));
    SPMV_HIP_TRY(rowloc.alloc((size_t)h.nnz + 8));
    SPMV_HIP_TRY(nlong.alloc((size_t)nb));
    SPMV_HIP_TRY(lrow.alloc((size_t)nb * kPool));
    SPMV_HIP_TRY(stats.alloc(2));
    // the runs of the panel-major order (nonempty tiles + one slot per panel for its pad slots)
    SPMV_HIP_TRY(run0.alloc((size_t)np + 1));
    k_bs_nruns<<<dim3((unsigned)np), dim3(256), 0, s>>>(nb, np, tiles.p, run0.p);
    if ((rc = check("k_bs_nruns"))) return rc;
    if ((rc = exclusive_scan_i32(run0.p, np, total.p, s))) return rc;
    int32_t nruns = 0;
    SPMV_HIP_TRY(hipMemcpyAsync(&nruns, total.p, sizeof nruns, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));
    SPMV_HIP_TRY(offset.alloc((size_t)nruns + 1));
    SPMV_HIP_TRY(first_run.alloc((size_t)padded / kBlk + 1));
    SPMV_HIP_TRY(hipMemsetAsync(offset.p, 0, sizeof(int32_t) * ((size_t)nruns + 1), s));
    SPMV_HIP_TRY(hipMemsetAsync(first_run.p, 0, sizeof(int32_t) * ((size_t)padded / kBlk + 1), s));
    SPMV_HIP_TRY(hipMemsetAsync(c16.p, 0, sizeof(uint16_t) * nslot, s));     // the pad slots of every panel: column 0, value 0 ...
    SPMV_HIP_TRY(hipMemsetAsync(pvals.p, 0, sizeof(float) * nslot, s));
    SPMV_HIP_TRY(hipMemsetD16Async(reinterpret_cast<hipDeviceptr_t>(acc.p), (unsigned short)(rb + kPool), bslot, s));   // pad entries: the dummy word
    SPMV_HIP_TRY(hipMemsetAsync(prod.p, 0, sizeof(float) * bslot, s));
    SPMV_HIP_TRY(hipMemsetAsync(stats.p, 0, sizeof(int32_t) * 2, s));
    int32_t st[2] = {0, 0};
    if (h.nnz > 0) {
        if ((rc = panel_rowloc(h, brow.p, nb, rowloc.p, s))) return rc;
        bool one_pass = false;
        if (const char *e = getenv("SPMV_BS_FILL")) one_pass = atoi(e) == 1;   // (A/B runs: the one-pass fill)
        if (one_pass) {
            k_bs_fill<<<dim3((unsigned)((nb + 3) / 4)), dim3(256), sizeof(int) * 4 * ((size_t)np + 256), s>>>(
                nb, np, brow.p, h.d_row_ptr, h.d_col_idx, h.d_vals, rowloc.p, tiles.p, pm.p, bbase.p, c16.p, pvals.p, acc.p);
            if ((rc = check("k_bs_fill"))) return rc;
        } else {
            const int ng = (np + 63) >> kGroupBits;
            DevPtr<int32_t> tcol;
            DevPtr<float> tval;
            DevPtr<uint16_t> trow;
            const int32_t *src_col = h.d_col_idx;
            const float *src_val = h.d_vals;
            const uint16_t *src_row = rowloc.p;
            if (ng > 1) {                                           // (one group: the bins' CSR ranges are grouped as they are)
                SPMV_HIP_TRY(tcol.alloc((size_t)h.nnz + 8));
                SPMV_HIP_TRY(tval.alloc((size_t)h.nnz + 8));
                SPMV_HIP_TRY(trow.alloc((size_t)h.nnz + 8));
                k_bs_group<<<dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, s>>>(nb, np, brow.p, h.d_row_ptr, h.d_col_idx, h.d_vals, rowloc.p,
                                                                                 tiles.p, tcol.p, tval.p, trow.p);
                if ((rc = check("k_bs_group"))) return rc;
                src_col = tcol.p; src_val = tval.p; src_row = trow.p;
            }
            const int64_t nitems64 = (int64_t)ng * nb;
            if (nitems64 > INT_MAX / 2) { set_error("spmv_csr_plan(panel, binned, scattered products): %lld fill items", (long long)nitems64); return SPMV_ERR_INVALID; }
            const int nitems = (int)nitems64, nwg = (nitems + 3) / 4;
            k_bs_place<<<dim3(8u * (unsigned)((nwg + 7) / 8)), dim3(256), 0, s>>>(nb, np, nitems, src_col, src_val, src_row, tiles.p, pm.p,
                                                                                 bbase.p, c16.p, pvals.p, acc.p);
            if ((rc = check("k_bs_place"))) return rc;
            SPMV_HIP_TRY(hipStreamSynchronize(s));                  // the grouped copies are freed here
        }
        k_bs_runs<<<dim3((unsigned)np), dim3(256), 0, s>>>(nb, np, bm, tiles.p, pm.p, pbase.p, bbase.p, run0.p, offset.p, first_run.p, c16.p);
        if ((rc = check("k_bs_runs"))) return rc;
        const size_t lds = sizeof(int) * 2 * (size_t)rb;
        if (rb == 16384) {
            static LdsOptIn optin;
            if ((rc = optin.ensure(reinterpret_cast<const void *>(&k_bs_accs<16384>), h.device, (int)lds))) return rc;
            k_bs_accs<16384><<<dim3((unsigned)nb), dim3(kWave), lds, s>>>(brow.p, bbase.p, bcnt.p, acc.p, nlong.p, lrow.p, stats.p);
        } else if (rb == 8192) {
            static LdsOptIn optin;
            if ((rc = optin.ensure(reinterpret_cast<const void *>(&k_bs_accs<8192>), h.device, (int)lds))) return rc;
            k_bs_accs<8192><<<dim3((unsigned)nb), dim3(kWave), lds, s>>>(brow.p, bbase.p, bcnt.p, acc.p, nlong.p, lrow.p, stats.p);
        } else {
            k_bs_accs<4096><<<dim3((unsigned)nb), dim3(kWave), lds, s>>>(brow.p, bbase.p, bcnt.p, acc.p, nlong.p, lrow.p, stats.p);
        }
        if ((rc = check("k_bs_accs"))) return rc;
        SPMV_HIP_TRY(hipMemcpyAsync(st, stats.p, sizeof st, hipMemcpyDeviceToHost, s));
    } else {
        SPMV_HIP_TRY(hipMemsetAsync(nlong.p, 0, sizeof(int32_t) * (size_t)nb, s));
    }
    // the product launch: at least two rounds of CUs over the panels that HOLD something (a banded shard fills a fraction of
    // its panels), their streams shared by `splits` workgroups where they are fewer
    const int cus = device_cus(h.device);
    {
        std::vector<int32_t> hb((size_t)np + 1);
        SPMV_HIP_TRY(hipMemcpyAsync(hb.data(), pbase.p, sizeof(int32_t) * ((size_t)np + 1), hipMemcpyDeviceToHost, s));
        SPMV_HIP_TRY(hipStreamSynchronize(s));
        int holding = 0;
        for (int q = 0; q < np; ++q) holding += hb[(size_t)q + 1] > hb[(size_t)q] ? 1 : 0;
        if (holding < 1) holding = 1;
        p.splits = holding >= 2 * cus ? 1 : (2 * cus + holding - 1) / holding;
    }
    if (const char *e = getenv("SPMV_BINNED_SPLITS")) { const int v = atoi(e); if (v > 0) p.splits = v; }
    if ((rc = stamp_values(h, s, p.stamp))) return rc;
    SPMV_HIP_TRY(hipStreamSynchronize(s));   // the temporaries (tiles, pm, rowloc) are freed on return
    p.padded = padded;
    p.bm_entries = bm;
    p.flagged_tiles = st[0];                 // (bins, in this flavour)
    p.long_rows = st[1];
    p.d_c16 = c16.release();
    p.d_pvals = pvals.release();
    p.d_offset = offset.release();
    p.d_first_run = first_run.release();
    p.runs = nruns;
    p.d_prod = prod.release();
    p.d_r16 = acc.release();
    p.d_pbase = pbase.release();
    p.d_bbase = bbase.release();
    p.d_bcnt = bcnt.release();
    p.d_nlong = nlong.release();
    p.d_lrow = lrow.release();
    p.d_brow = brow.release();
    p.ready = true;
    return SPMV_OK;
}

// want_rows: 0 = the rule (8192 rows per bin where that still leaves two bins per resident wavefront, else 4096), 4096 | 8192
// scatter: the flavour whose product launch stores in bin order (thin tiles)
int plan_binned(spmv_csr &h, PanelPlan &p, int want_rows, bool scatter, hipStream_t s)
{
    if (scatter && want_rows != 0 && want_rows != 4096 && want_rows != 8192 && want_rows != 16384) {
        set_error("spmv_csr_plan(panel, binned, scattered products): rows per bin %d (4096, 8192 or 16384)", want_rows);
        return SPMV_ERR_INVALID;
    }
    if (!scatter && want_rows != 0 && want_rows != 1024 && want_rows != 2048 && want_rows != 4096 && want_rows != 8192) {
        set_error("spmv_csr_plan(panel, binned): rows per bin %d (1024, 2048, 4096 or 8192)", want_rows);
        return SPMV_ERR_INVALID;
    }
    const int64_t np64 = (h.cols + kPw - 1) >> kPwBits;
    if (np64 > kMaxPanels) {
        set_error("spmv_csr_plan(panel, binned): %lld columns need more than %d panels of %d columns", (long long)h.cols, kMaxPanels, kPw);
        return SPMV_ERR_INVALID;
    }
    const int np = np64 < 1 ? 1 : (int)np64;
    if (h.nnz > (1ll << 30) - 8ll * np - 512) {     // (the products are addressed through a buffer descriptor: 4 GiB)
        set_error("spmv_csr_plan(panel, binned): nnz %lld beyond the layout's 2^30 nonzeros per handle", (long long)h.nnz);
        return SPMV_ERR_INVALID;
    }
    const int cus = device_cus(h.device);
    int rb = want_rows;
    if (rb == 0) rb = scatter ? 8192 : 4096;     // eight wavefronts per CU (8192: four; measured slower at every size tried: DESIGN.md); scattered: fatter tiles win
    p.binned_mode = true;
    p.scatter_mode = scatter;
    p.bin_rows = rb;
    p.pw_bits = kPwBits;
    p.npanels = np;
    p.splits = 1;
    if (h.rows == 0) {
        p.stamp.gen = h.values_gen;
        p.stamp.have_sum = false;
        p.ready = true;
        return SPMV_OK;
    }
    // bins: equal nonzero counts, at most rb rows, a whole number of rounds of the wavefronts the sum launch keeps resident
    // (wavefronts that carry equal loads finish together; the cuts aim at 0.8 rb rows so that hardly any needs splitting)
    const int64_t slots = (int64_t)cus * ((rb + kSpare + kWave) * 4 * kSumWaves <= 80 * 1024 ? 2 : 1) * kSumWaves;
    int64_t nb0 = slots * ((h.rows * 5 / 4 + (int64_t)rb * slots - 1) / ((int64_t)rb * slots));
    if (scatter) {
        // as many rows per bin as the cuts allow (fat runs for the product launch) -- but a bin per wavefront the sum launch
        // keeps resident at least: config 3 (4Mi rows) 576 bins of 8192 rows ran 0.81 ms, 1024 bins of 4096 0.52
        // (profiles/r04_scattered_bins.jsonl; more bins than that change little: the launch is not bound by its rounds)
        nb0 = (h.rows * 9 / 8 + rb - 1) / rb;
        const int64_t resident = (int64_t)cus * (rb == 4096 ? 7 : rb == 8192 ? 4 : 2);
        if (nb0 < resident) nb0 = resident;
        if (const char *e = getenv("SPMV_BS_BINS")) { const long long v = atoll(e); if (v > 0) nb0 = v; }   // (A/B runs)
    }
    if (nb0 > h.rows) nb0 = h.rows;
    DevPtr<int32_t> brow;
    int rc = panel_row_blocks(h, nb0, rb, s, brow, &p.nblocks);
    if (rc) return rc;
    const int nb = p.nblocks;
    if ((int64_t)nb * (np + 1) > (int64_t)INT_MAX / 2) {
        set_error("spmv_csr_plan(panel, binned): %d bins x %d panels is more tiles than the tables hold", nb, np);
        return SPMV_ERR_INVALID;
    }
    DevPtr<int32_t> tiles, pm, pbase, total;
    DevPtr<uint16_t> rowloc, c16, r16;
    DevPtr<float> pvals, prod;
    SPMV_HIP_TRY(tiles.alloc((size_t)nb * (size_t)(np + 1)));
    SPMV_HIP_TRY(pm.alloc((size_t)nb * (size_t)np));
    SPMV_HIP_TRY(pbase.alloc((size_t)np + 1));
    SPMV_HIP_TRY(total.alloc(1));
    if ((rc = panel_tile_ptr(h, brow.p, nb, kPwBits, np, tiles.p, s))) return rc;
    k_bin_panel_totals<<<dim3((unsigned)np), dim3(256), 0, s>>>(nb, np, scatter ? kBlk : 8, tiles.p, pbase.p);
    if ((rc = check("k_bin_panel_totals"))) return rc;
    if ((rc = exclusive_scan_i32(pbase.p, np, total.p, s))) return rc;
    int32_t padded = 0;
    SPMV_HIP_TRY(hipMemcpyAsync(&padded, total.p, sizeof padded, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipMemcpyAsync(pbase.p + np, total.p, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));
    k_bin_pm<<<dim3((unsigned)np), dim3(256), 0, s>>>(nb, np, tiles.p, pbase.p, pm.p);
    if ((rc = check("k_bin_pm"))) return rc;
    if (scatter) return plan_scatter(h, p, rb, padded, brow, tiles, pm, pbase, s);
    const size_t nslot = (size_t)padded + 264;                             // (a piece reads up to 256 entries past a tile's end)
    SPMV_HIP_TRY(c16.alloc(nslot));
    SPMV_HIP_TRY(pvals.alloc(nslot));
    SPMV_HIP_TRY(prod.alloc(nslot));
    SPMV_HIP_TRY(r16.alloc((size_t)h.nnz + 264));
    SPMV_HIP_TRY(hipMemsetAsync(r16.p + h.nnz, 0, sizeof(uint16_t) * 264, s));
    SPMV_HIP_TRY(rowloc.alloc((size_t)h.nnz + 8));
    SPMV_HIP_TRY(hipMemsetAsync(c16.p, 0, sizeof(uint16_t) * nslot, s));     // (the pad slots of every panel: column 0, value 0)
    SPMV_HIP_TRY(hipMemsetAsync(pvals.p, 0, sizeof(float) * nslot, s));
    if (h.nnz > 0) {
        if ((rc = panel_rowloc(h, brow.p, nb, rowloc.p, s))) return rc;
        k_bin_fill<<<dim3((unsigned)((nb + 3) / 4)), dim3(256), sizeof(int) * 4 * ((size_t)np + 256), s>>>(
            nb, np, brow.p, h.d_row_ptr, h.d_col_idx, h.d_vals, rowloc.p, tiles.p, pm.p, c16.p, pvals.p, r16.p);
        if ((rc = check("k_bin_fill"))) return rc;
    }
    // the product launch: every panel's stream shared by `splits` workgroups so that the launch is at least two rounds of CUs
    p.splits = np >= 2 * cus ? 1 : (2 * cus + np - 1) / np;
    if (const char *e = getenv("SPMV_BINNED_SPLITS")) { const int v = atoi(e); if (v > 0) p.splits = v; }
    // fat tiles (a hundred products or more): four products per lane and piece; thin ones: two
    p.wide_pieces = (double)h.nnz >= 96.0 * (double)nb * (double)np;
    if (const char *e = getenv("SPMV_BINNED_WIDE")) { if (*e) p.wide_pieces = atoi(e) != 0; }
    // rows that hold more products of a tile than a lane takes: spare sums (k_bin_runs); bins with too many of them: the fold
    int32_t flagged = 0, nlong_total = 0;
    DevPtr<int32_t> maxrun, spare, lptr, lcnt;
    DevPtr<uint32_t> lrow;
    const int e = p.wide_pieces ? 4 : 2;
    SPMV_HIP_TRY(lptr.alloc((size_t)nb + 1));
    SPMV_HIP_TRY(hipMemsetAsync(lptr.p, 0, sizeof(int32_t) * ((size_t)nb + 1), s));
    if (h.nnz > 0) {
        SPMV_HIP_TRY(maxrun.alloc((size_t)h.rows));
        SPMV_HIP_TRY(spare.alloc((size_t)nb));
        SPMV_HIP_TRY(hipMemsetAsync(maxrun.p, 0, sizeof(int32_t) * (size_t)h.rows, s));
        k_bin_runs<<<dim3((unsigned)nb), dim3(256), 0, s>>>(np, e, brow.p, tiles.p, r16.p, maxrun.p);
        if ((rc = check("k_bin_runs"))) return rc;
        k_bin_spare<<<dim3((unsigned)nb), dim3(256), 0, s>>>(e, brow.p, maxrun.p, spare.p, lptr.p);
        if ((rc = check("k_bin_spare"))) return rc;
        if ((rc = exclusive_scan_i32(lptr.p, nb, total.p, s))) return rc;
        SPMV_HIP_TRY(hipMemcpyAsync(&nlong_total, total.p, sizeof nlong_total, hipMemcpyDeviceToHost, s));
        SPMV_HIP_TRY(hipMemcpyAsync(lptr.p + nb, total.p, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        SPMV_HIP_TRY(hipStreamSynchronize(s));
    }
    SPMV_HIP_TRY(lrow.alloc((size_t)nlong_total + 1));
    SPMV_HIP_TRY(lcnt.alloc((size_t)nlong_total + 1));
    if (h.nnz > 0) {
        SPMV_HIP_TRY(hipMemsetAsync(total.p, 0, sizeof(int32_t), s));
        k_bin_rewrite<<<dim3((unsigned)nb), dim3(256), 0, s>>>(np, e, rb, brow.p, tiles.p, maxrun.p, spare.p, lptr.p, r16.p, pm.p, lrow.p,
                                                               lcnt.p, total.p);
        if ((rc = check("k_bin_rewrite"))) return rc;
        SPMV_HIP_TRY(hipMemcpyAsync(&flagged, total.p, sizeof flagged, hipMemcpyDeviceToHost, s));
    }
    if ((rc = stamp_values(h, s, p.stamp))) return rc;
    SPMV_HIP_TRY(hipStreamSynchronize(s));   // the temporaries are freed on return
    p.padded = padded;
    p.flagged_tiles = flagged;
    p.long_rows = nlong_total;
    p.d_lptr = lptr.release();
    p.d_lrow = lrow.release();
    p.d_lcnt = lcnt.release();
    p.d_c16 = c16.release();
    p.d_pvals = pvals.release();
    p.d_prod = prod.release();
    p.d_r16 = r16.release();
    p.d_pm = pm.release();
    p.d_pbase = pbase.release();
    p.d_tile_ptr = tiles.release();
    p.d_brow = brow.release();
    p.ready = true;
    return SPMV_OK;
}

template <int RB, int E>
static int launch_sums_e(const spmv_csr &h, const PanelPlan &p, float *y, hipStream_t s)
{
    const size_t lds = sizeof(float) * (size_t)(RB + kSpare + kWave) * kSumWaves;
    const dim3 grid((unsigned)((p.nblocks + kSumWaves - 1) / kSumWaves)), block(kSumWaves * kWave);
    static LdsOptIn optin;
    if (int rc = optin.ensure(reinterpret_cast<const void *>(&k_bin_sums<RB, E>), h.device, (int)lds)) return rc;
    k_bin_sums<RB, E><<<grid, block, lds, s>>>(p.nblocks, p.npanels, p.d_brow, p.d_tile_ptr, p.d_pm, p.d_r16, p.d_prod, p.d_lptr,
                                               p.d_lrow, p.d_lcnt, y, (uint32_t)((p.padded + 264) * 4), (uint32_t)((h.nnz + 264) * 2));
    return check("k_bin_sums");
}
template <int RB>
static int launch_sums(const spmv_csr &h, const PanelPlan &p, float *y, hipStream_t s)
{
    return p.wide_pieces ? launch_sums_e<RB, 4>(h, p, y, s) : launch_sums_e<RB, 2>(h, p, y, s);
}

template <int RB>
static int launch_bs_sums(const spmv_csr &h, const PanelPlan &p, float *y, hipStream_t s)
{
    const size_t lds = sizeof(float) * (size_t)(RB + kPool + kWave);
    static LdsOptIn optin;
    if (int rc = optin.ensure(reinterpret_cast<const void *>(&k_bs_sums<RB>), h.device, (int)lds)) return rc;
    k_bs_sums<RB><<<dim3((unsigned)p.nblocks), dim3(kWave), lds, s>>>(p.d_brow, p.d_bbase, p.d_bcnt, p.d_r16, p.d_prod, p.d_nlong, p.d_lrow, y,
                                                                      (uint32_t)((p.bm_entries + kBlk + 16) * 4), (uint32_t)((p.bm_entries + kBlk + 16) * 2));
    return check("k_bs_sums");
}

int launch_binned(const spmv_csr &h, const PanelPlan &p, const float *x, float *y, hipStream_t s)
{
    if (p.nblocks == 0) return SPMV_OK;   // no rows
    if (p.scatter_mode) {
        const size_t lds = sizeof(float) * (size_t)kPw;
        static LdsOptIn optin;
        if (int rc = optin.ensure(reinterpret_cast<const void *>(&k_bs_products), h.device, (int)lds)) return rc;
        const unsigned grid = p.splits == 1 ? 256u * (unsigned)((p.npanels + 255) / 256) : (unsigned)(p.npanels * p.splits);
        k_bs_products<<<dim3(grid), dim3(kProdThreads), lds, s>>>(p.splits, p.npanels, h.cols, p.d_pbase, p.d_c16,
                                                                                              p.d_pvals, p.d_first_run, p.d_offset, x, p.d_prod);
        if (int rc = check("k_bs_products")) return rc;
        return p.bin_rows == 16384 ? launch_bs_sums<16384>(h, p, y, s) : p.bin_rows == 8192 ? launch_bs_sums<8192>(h, p, y, s) : launch_bs_sums<4096>(h, p, y, s);
    }
    {
        const size_t lds = sizeof(float) * (size_t)kPw;
        static LdsOptIn optin;
        if (int rc = optin.ensure(reinterpret_cast<const void *>(&k_bin_products), h.device, (int)lds)) return rc;
        k_bin_products<<<dim3((unsigned)(p.npanels * p.splits)), dim3(kProdThreads), lds, s>>>(p.splits, h.cols, p.d_pbase, p.d_c16,
                                                                                               p.d_pvals, x, p.d_prod);
        if (int rc = check("k_bin_products")) return rc;
    }
    return p.bin_rows == 8192 ? launch_sums<8192>(h, p, y, s) : p.bin_rows == 4096 ? launch_sums<4096>(h, p, y, s) :
           p.bin_rows == 2048 ? launch_sums<2048>(h, p, y, s) : launch_sums<1024>(h, p, y, s);
}

}  // namespace spmv
