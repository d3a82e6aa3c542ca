// kernels_panel.hip -- SPMV_PANEL: a (row block x column panel) sweep for matrices whose columns have no locality.
//
// Why: with columns drawn from all of a large x (tens of MiB), every CSR kernel of this library -- and every
// rocSPARSE algorithm -- pays one L2-missing fabric request per nonzero (~56 G/s on MI355X).
// Here the rows are cut into blocks of equal NONZERO count (at most 9728 rows; as many blocks as fill every
// resident wave slot), and the nonzeros of a block are re-ordered, once, by COLUMN PANEL (2^pw_bits columns,
// 512 KiB of x by default).  One wavefront owns the block: its output sums live in LDS for the whole sweep, it
// streams its re-ordered nonzeros front to back, and because every wave of the launch walks the panels in the
// same order at the same pace (equal loads), the slice of x they are all
// gathering from stays in each XCD's 4 MiB L2.  The grid of one launch is exactly what is resident on the chip (a
// second launch takes the next set of row blocks): a wave that starts late is out of step with the others, and the
// L2 then holds several panels instead of one (measured with tools/ubench_panel.hip: 2.3 ms instead of 1.2 ms on
// the 16Mi x 16Mi / 2^28-nonzero uniform workload).  Low occupancy is deliberate for the same reason -- 4 waves per
// CU, each with dozens of sixteen-byte loads and gathers in flight, beat 16 or 32 waves per CU.
//
// The reference's tiled format (TCSRMatrix, src/tcsr.cpp:5-38, multiplied by csr_tiling_kernel,
// src/kernels/csr_tiling.cu:24-114: 32 x 32 tiles, x tile and tile values in shared memory) is the same idea at
// dense-matrix scale; this is its sparse-matrix, cache-sized counterpart.
//
// Layout (PanelPlan): for row block b (rows [brow[b], brow[b+1])) the nonzeros keep their CSR range
// [row_ptr[brow[b]], row_ptr[brow[b+1]]) but are stably sorted by panel:
//   packed[k] = row_in_block << 18 | join << 17 | column_in_panel     pvals[k] = value   (8 B per nonzero, as CSR)
//   (join: the nonzero 4 places earlier in the same tile has the same row -- the two would meet in one instruction)
//   tile_ptr[b*(np+1) + p] = first k of panel p in block b (tile_ptr[..+np] = end of the block)
// Inside a tile rows ascend, so the lanes of one instruction normally hold distinct rows and the sum into LDS is a
// plain read-add-write (an LDS float atomic per nonzero costs more than the whole rest of the kernel: measured);
// the plan marks the nonzeros whose left neighbour in an instruction holds the same row; a group of four
// instructions that holds such a mark, or straddles two tiles, adds with LDS atomics instead.  No barriers; a wave's
// sums are touched by that wave alone, in instruction order.
// The values are COPIED at plan time (the other variants read the live vals array): re-plan after changing them.
#include <climits>
#include <cstdlib>
#include "spmv_internal.hpp"

namespace spmv {

namespace {

// (the SPMV_PANEL_* macros exist for A/B builds, tools/explore.py; the shipped values are the defaults)
#ifndef SPMV_PANEL_RW
#define SPMV_PANEL_RW 9728
#endif
#ifndef SPMV_PANEL_WG_PER_CU
#define SPMV_PANEL_WG_PER_CU 2
#endif
constexpr int kRw = SPMV_PANEL_RW;         // most rows of one block (equal-nonzero cuts vary): 38 KiB of LDS per wave
constexpr int kRwTarget = kRw == 9728 ? 8192 : kRw * 27 / 32;   // rows per wave block the launch count is sized for
constexpr int kWgPerCu = SPMV_PANEL_WG_PER_CU;   // workgroups of two wavefronts resident per CU: 2 x 76 KiB of LDS
constexpr int kColBits = 17;               // column_in_panel field of packed[]
constexpr unsigned kColMask = (1u << kColBits) - 1;
constexpr unsigned kJoinBit = 1u << kColBits;   // this nonzero and the one 4 places before it in its tile share a row
constexpr int kRowShift = kColBits + 1;
constexpr int kWavesPerWg = 2;             // 76 KiB of LDS per workgroup -> 2 workgroups = 4 waves per CU
// 16-byte vectors per lane, step and array: a step of the sweep is 256 x VEC nonzeros of a wavefront's stream.  Two
// instantiations, chosen when the plan is made (PanelPlan::step_vecs): 8 where the tiles are fat (+4 % over 4 at configs 3
// and 4), 4 where a step of 2048 would span many panels -- config 5's shard, 128 nonzeros per tile: the gathers of ONE step
// then cover 16 panels = 8 MiB of x, twice an L2, and the wavefronts of an XCD stop sharing lines (3.02 -> 2.41 ms,
// profiles/r04_panel_step_size.jsonl; 6: 2.61, 3: 2.44, 2: 2.53, 1: 2.92)
constexpr int kVecMax = 8;
constexpr int kStepMax = kWave * 4 * kVecMax;
constexpr int kMaxPanels = 4096;

using u4 = unsigned __attribute__((ext_vector_type(4)));
using f4 = float __attribute__((ext_vector_type(4)));

// ---- plan kernels -----------------------------------------------------------------------------------------
// Row blocks with equal NONZERO counts (waves that carry equal loads stay in step): cut[b] = first row whose
// row_ptr reaches b * nnz / nb.
__global__ void k_panel_cuts(int64_t rows, int64_t nnz, int nb, const int32_t *__restrict__ row_ptr,
                             int32_t *__restrict__ cut)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    if (b == nb) {
        cut[b] = (int32_t)rows;
        return;
    }
    const int64_t target = nnz * b / nb;
    int64_t lo = 0, hi = rows;   // first r in [0, rows] with row_ptr[r] >= target
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)row_ptr[mid] >= target) hi = mid; else lo = mid + 1;
    }
    cut[b] = (int32_t)lo;
}
// a cut of more than `cap` rows (a stretch of short or empty rows) is split evenly: nsub[b] pieces
__global__ void k_panel_nsub(int nb, int cap, const int32_t *__restrict__ cut, int32_t *__restrict__ nsub)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb) nsub[b] = (cut[b + 1] - cut[b] + cap - 1) / cap;
}
__global__ void k_panel_brow(int64_t rows, int nb, int cap, const int32_t *__restrict__ cut, const int32_t *__restrict__ off,
                             const int32_t *__restrict__ total, int32_t *__restrict__ brow)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) brow[*total] = (int32_t)rows;
    if (b >= nb) return;
    const int len = cut[b + 1] - cut[b];
    const int n = (len + cap - 1) / cap;
    if (n == 0) return;
    const int step = (len + n - 1) / n;
    for (int i = 0; i < n; ++i) brow[off[b] + i] = cut[b] + i * step;
}

// tile_ptr of one row block: histogram of the panels of its nonzeros, then an exclusive scan.
__global__ __launch_bounds__(256) void k_panel_tiles(const int32_t *__restrict__ brow,
                                                     const int32_t *__restrict__ row_ptr,
                                                     const int32_t *__restrict__ col_idx, int pw_bits, int np,
                                                     int32_t *__restrict__ tile_ptr)
{
    __shared__ int hist[kMaxPanels];
    __shared__ int part[256];
    const int tid = threadIdx.x;
    const int s = row_ptr[brow[blockIdx.x]], e = row_ptr[brow[blockIdx.x + 1]];
    for (int p = tid; p < np; p += 256) hist[p] = 0;
    __syncthreads();
    for (int k = s + tid; k < e; k += 256) atomicAdd(&hist[col_idx[k] >> pw_bits], 1);
    __syncthreads();
    // thread t scans entries [t*per, (t+1)*per)
    const int per = (np + 255) / 256;
    int sum = 0;
    for (int i = 0; i < per; ++i) {
        const int p = tid * per + i;
        if (p < np) sum += hist[p];
    }
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = s + part[tid] - sum;
    int32_t *tp = tile_ptr + (int64_t)blockIdx.x * (np + 1);
    for (int i = 0; i < per; ++i) {
        const int p = tid * per + i;
        if (p < np) {
            tp[p] = run;
            run += hist[p];
        }
    }
    if (tid == 0) tp[np] = e;
}

// row_in_block of every nonzero (temporary, 2 B per nonzero): one workgroup per row block, one lane per row,
// long rows by the whole wave.
__global__ __launch_bounds__(256) void k_panel_rowloc(const int32_t *__restrict__ brow,
                                                      const int32_t *__restrict__ row_ptr,
                                                      uint16_t *__restrict__ rowloc)
{
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int row0 = brow[blockIdx.x], row1 = brow[blockIdx.x + 1];
    for (int r0 = row0 + w * kWave; r0 < row1; r0 += 4 * kWave) {
        const int r = r0 + lane;
        int a = 0, b = 0;
        if (r < row1) {
            a = row_ptr[r];
            b = row_ptr[r + 1];
        }
        const uint16_t mine = (uint16_t)(r - row0);
        const bool is_long = b - a > 64;
        if (!is_long)
            for (int k = a; k < b; ++k) rowloc[k] = mine;
        unsigned long long todo = __ballot(is_long);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int la = __shfl(a, src), lb = __shfl(b, src);
            const uint16_t lr = (uint16_t)__shfl((int)mine, src);
            for (int k = la + lane; k < lb; k += kWave) rowloc[k] = lr;
        }
    }
}

// Stable scatter of one row block into panel order: one wave per block, 64 nonzeros per step in CSR order; the
// rank of a nonzero among the same-panel nonzeros of its step comes from a ballot per distinct panel.
__global__ __launch_bounds__(256) void k_panel_fill(int nblocks, const int32_t *__restrict__ brow,
                                                    const int32_t *__restrict__ row_ptr,
                                                    const int32_t *__restrict__ col_idx,
                                                    const float *__restrict__ vals,
                                                    const uint16_t *__restrict__ rowloc, int pw_bits, int np,
                                                    const int32_t *__restrict__ tile_ptr,
                                                    uint32_t *__restrict__ packed, float *__restrict__ pvals)
{
    __shared__ int cursor_all[4][kMaxPanels];
    __shared__ int tags_all[4][256];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int wb = blockIdx.x * 4 + w;
    if (wb >= nblocks) return;
    int *cursor = cursor_all[w];
    volatile int *tags = tags_all[w];
    const int32_t *tp = tile_ptr + (int64_t)wb * (np + 1);
    for (int p = lane; p < np; p += kWave) cursor[p] = tp[p];
    const int s = row_ptr[brow[wb]], e = row_ptr[brow[wb + 1]];
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned cmask = (1u << pw_bits) - 1u;
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
        const int p = col >> pw_bits;
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
            packed[dest] = ((unsigned)rl << kRowShift) | ((unsigned)col & cmask);
            pvals[dest] = v;
        }
    }
}

// join bits: nonzero k and nonzero k-4 of the SAME tile have the same row.  Lanes of one multiply instruction hold
// stream positions 4 apart, so an instruction without join bits (and inside one tile) holds 64 distinct rows.
__global__ __launch_bounds__(256) void k_panel_joins(const int32_t *__restrict__ brow, int np,
                                                     const int32_t *__restrict__ row_ptr,
                                                     const int32_t *__restrict__ tile_ptr, uint32_t *__restrict__ packed)
{
    const int s = row_ptr[brow[blockIdx.x]], e = row_ptr[brow[blockIdx.x + 1]];
    const int32_t *tp = tile_ptr + (int64_t)blockIdx.x * (np + 1);
    for (int k = s + 4 + (int)threadIdx.x; k < e; k += 256) {
        int lo = 0, hi = np;                     // tile of k: the last p with tp[p] <= k
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (tp[mid] <= k) lo = mid; else hi = mid;
        }
        if (k - 4 >= tp[lo] && (packed[k - 4] >> kRowShift) == (packed[k] >> kRowShift)) atomicOr(&packed[k], kJoinBit);
    }
}

// ---- the multiply -------------------------------------------------------------------------------------------
template <int kVec>
struct StepRegs {     // one step of the stream, as loaded
    u4 c[kVec];
    f4 v[kVec];
};
template <int kVec>
struct GatherRegs {   // one step between its gathers and its sums
    float xv[kVec][4];
    float val[kVec][4];
    int row[kVec][4];
    bool simple[kVec];     // wave-uniform: group j holds 4 x 64 distinct rows
    bool straddle[kVec];   // wave-uniform: group j spans two tiles
    bool interior;
};

template <int kVec>
__device__ __forceinline__ void panel_load(const u4 *__restrict__ c4, const f4 *__restrict__ v4, int base, int lane,
                                           StepRegs<kVec> &r)
{
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
        const int i = (base >> 2) + j * kWave + lane;
        r.c[j] = __builtin_nontemporal_load(&c4[i]);
        r.v[j] = __builtin_nontemporal_load(&v4[i]);
    }
}

template <int kVec>
__global__ __launch_bounds__(kWave *kWavesPerWg) void k_panel(int wb0, int wb1, const int32_t *__restrict__ brow,
                                                              const int32_t *__restrict__ row_ptr,
                                                              const int32_t *__restrict__ tile_ptr,
                                                              const uint32_t *__restrict__ packed,
                                                              const float *__restrict__ pvals,
                                                              const float *__restrict__ x, float *__restrict__ y,
                                                              int np, int pw_bits)
{
    constexpr int kStep = kWave * 4 * kVec;
    __shared__ float ys_all[kWavesPerWg][kRw];
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int wb = wb0 + blockIdx.x * kWavesPerWg + w;
    if (wb >= wb1) return;   // no barrier anywhere below: the waves of a workgroup are independent
    float *ys = ys_all[w];
    const int row0 = brow[wb], row1 = brow[wb + 1];
    const int n = row1 - row0;
    for (int i = lane; i < n; i += kWave) ys[i] = 0.0f;
    const int s = row_ptr[row0], e = row_ptr[row1];
    const int32_t *tp = tile_ptr + (int64_t)wb * (np + 1);
    const u4 *c4 = reinterpret_cast<const u4 *>(packed);
    const f4 *v4 = reinterpret_cast<const f4 *>(pvals);
    const int k0 = s & ~3;
    const int nsteps = (e - k0 + kStep - 1) / kStep;
    int pdone = 0;   // boundaries tp[1..pdone] lie at or before the current step

    // Two stages per step, one step apart, so that a step's 32 gathers per lane are in flight while the previous
    // step is being summed (one wave per SIMD: nothing else hides their latency).
    //   gather(step): panel of every element, issue the x loads, keep what the sums need (the stream registers are
    //                 free for the next load right after);
    //   sum(step):    add into the wave's LDS sums.
    auto gather = [&](int base, const StepRegs<kVec> &r, GatherRegs<kVec> &g) {
        // panel of every element: pdone + the boundaries inside this step that lie at or before it
        int pv[kVec][4];
#pragma unroll
        for (int j = 0; j < kVec; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) pv[j][q] = pdone;
        int p = pdone + 1;
        while (p < np) {
            const int bnd = tp[__builtin_amdgcn_readfirstlane(p)];
            if ((int64_t)bnd >= (int64_t)base + kStep) break;
#pragma unroll
            for (int j = 0; j < kVec; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) pv[j][q] += (base + (j * kWave + lane) * 4 + q >= bnd) ? 1 : 0;
            ++p;
        }
        pdone = p - 1;
        g.interior = base >= s && (int64_t)base + kStep <= (int64_t)e;
#pragma unroll
        for (int j = 0; j < kVec; ++j) {
            // the four instructions of group j cover 256 consecutive stream positions: one tile, no join bits,
            // all valid -> 64 distinct rows each
            const unsigned joins = (r.c[j][0] | r.c[j][1] | r.c[j][2] | r.c[j][3]) & kJoinBit;
            g.straddle[j] = __builtin_amdgcn_readlane(pv[j][0], 0) != __builtin_amdgcn_readlane(pv[j][3], 63);
            g.simple[j] = g.interior && !g.straddle[j] && __ballot(joins != 0u) == 0ull;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t col = ((int64_t)pv[j][q] << pw_bits) | (int64_t)(r.c[j][q] & kColMask);
                if (g.interior) {
                    g.xv[j][q] = x[col];
                } else {
                    const int k = base + (j * kWave + lane) * 4 + q;
                    g.xv[j][q] = (k >= s && k < e) ? x[col] : 0.0f;
                }
                g.row[j][q] = (int)(r.c[j][q] >> kRowShift);
                g.val[j][q] = r.v[j][q];
            }
        }
    };
    auto sum = [&](int base, const GatherRegs<kVec> &g) {
#pragma unroll
        for (int j = 0; j < kVec; ++j) {
            if (g.simple[j]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) ys[g.row[j][q]] = ys[g.row[j][q]] + g.val[j][q] * g.xv[j][q];
            } else {
                // a tile boundary, a repeated row or the ragged end of the stream: LDS atomics (one wave, in
                // instruction order).  Cheaper than sorting the cases out lane by lane -- measured, DESIGN.md.
                const int kq = base + (j * kWave + lane) * 4;
                const bool all4 = g.interior || (kq >= s && kq + 3 < e);
                if (all4 && g.row[j][0] == g.row[j][3] && !g.straddle[j]) {
                    // my four consecutive nonzeros are one row (rows ascend inside a tile): one add
                    atomicAdd(&ys[g.row[j][0]], (g.val[j][0] * g.xv[j][0] + g.val[j][1] * g.xv[j][1]) +
                                                    (g.val[j][2] * g.xv[j][2] + g.val[j][3] * g.xv[j][3]));
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (g.interior || (kq + q >= s && kq + q < e))
                            atomicAdd(&ys[g.row[j][q]], g.val[j][q] * g.xv[j][q]);
                }
            }
        }
    };

    StepRegs<kVec> s0, s1;
    GatherRegs<kVec> g0, g1;
    auto step_base = [&](int st) { return k0 + st * kStep; };
    if (nsteps > 0) {
        panel_load(c4, v4, step_base(0), lane, s0);
        if (nsteps > 1) panel_load(c4, v4, step_base(1), lane, s1);
        gather(step_base(0), s0, g0);
        if (nsteps > 2) panel_load(c4, v4, step_base(2), lane, s0);
    }
    for (int st = 0; st < nsteps; st += 2) {
        if (st + 1 < nsteps) {
            gather(step_base(st + 1), s1, g1);
            if (st + 3 < nsteps) panel_load(c4, v4, step_base(st + 3), lane, s1);
        }
        sum(step_base(st), g0);
        if (st + 2 < nsteps) {
            gather(step_base(st + 2), s0, g0);
            if (st + 4 < nsteps) panel_load(c4, v4, step_base(st + 4), lane, s0);
        }
        if (st + 1 < nsteps) sum(step_base(st + 1), g1);
    }
    for (int i = lane; i < n; i += kWave) y[row0 + i] = ys[i];
}

// ---- the same sweep with the panel of x staged in LDS (PanelPlan::lds_mode) -----------------------------------------
// For a SMALL x (config 2: 4 MiB) the panel does not have to come through L2 gather by gather -- one 128-byte line per
// nonzero, 0.28 T gathers/s at best (tools/ubench_gather.hip) -- every workgroup can copy ALL of x through its LDS,
// 16 Ki columns at a time with 16-byte LDS-DMA loads (22-27 TB/s chip-wide, same tool), and gather from LDS.  That is
// cols x 4 bytes of L2->LDS traffic per workgroup whatever it multiplies, so it pays when the workgroups are few and
// the panels well filled.  MEASURED: it does not pay even there -- config 2 (uniform columns) 0.135 ms against 0.087 ms
// for the L2 sweep above: 64 panel steps of a barrier, 64 KiB of DMA and ONE 64-nonzero instruction per wavefront each
// expose a memory round trip per step (a deeper-pipelined form with the stream prefetched three groups ahead and the
// tile bounds on the scalar unit measured 0.186 ms: the compiler parks the prefetched group behind the next panel's
// DMA).  So the library never chooses this mode; it documents the experiment and stays selectable and tested.
// Workgroup = 16 wavefronts, each with a row block of <= 384 rows whose sums live in its slice of LDS (same layout
// and plan as above with 2^14-column panels); two panel buffers: panel p+1 streams in while panel p is multiplied,
// one barrier per panel.  Lanes hold CONSECUTIVE nonzeros of the tile (rows ascend), equal rows are folded by a
// segmented shuffle scan and the last lane of each run adds into LDS with a plain read-add-write -- no join bits, no
// LDS atomics (0.33 lane-updates per clock, tools/ubench_lds_atomic.hip).
constexpr int kLdsPwBits = 14;
constexpr int kLdsW = 1 << kLdsPwBits;      // columns per staged panel: 64 KiB
constexpr int kLdsWaves = 16;
constexpr int kRwLds = 384;                 // most rows of a wavefront's block: 1.5 KiB of sums
constexpr int kRwLdsTarget = 256;

__global__ __launch_bounds__(kLdsWaves *kWave) void k_panel_lds(int nblocks, int64_t cols, const int32_t *__restrict__ brow,
                                                                 const int32_t *__restrict__ tile_ptr,
                                                                 const uint32_t *__restrict__ packed,
                                                                 const float *__restrict__ pvals,
                                                                 const float *__restrict__ x, float *__restrict__ y, int np)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *ys = lds + 2 * kLdsW + w * kRwLds;
    const int wb = blockIdx.x * kLdsWaves + w;
    const bool live = wb < nblocks;                         // wave-uniform; idle wavefronts still stage and meet the barriers
    const int row0 = live ? brow[wb] : 0, row1 = live ? brow[wb + 1] : 0;
    const int n = row1 - row0;
    for (int i = lane; i < kRwLds; i += kWave) ys[i] = 0.0f;
    const int32_t *tp = tile_ptr + (int64_t)(live ? wb : 0) * (np + 1);

    auto stage = [&](int p) {
        float *dst = lds + (p & 1) * kLdsW;
        const int64_t g0 = (int64_t)p << kLdsPwBits;
        if (g0 + kLdsW + 3 < cols) {                        // workgroup-uniform
#pragma unroll
            for (int i = tid * 4; i < kLdsW; i += kLdsWaves * kWave * 4) __builtin_amdgcn_global_load_lds(x + g0 + i, dst + i, 16, 0, 0);
        } else {
            for (int i = tid; i < kLdsW; i += kLdsWaves * kWave) dst[i] = g0 + i < cols ? x[g0 + i] : 0.0f;
        }
    };
    stage(0);
    for (int p = 0; p < np; ++p) {
        __builtin_amdgcn_s_waitcnt(0);                      // this lane's pieces of panel p (and its stream loads) have landed
        __syncthreads();                                    // ... everyone's; and everyone is done reading buffer (p+1)&1
        if (p + 1 < np) stage(p + 1);
        if (!live) continue;
        const float *xp = lds + (p & 1) * kLdsW;
        const int k0 = tp[p], k1 = tp[p + 1];
        for (int kb = k0; kb < k1; kb += kWave) {
            const int k = kb + lane;
            const bool on = k < k1;
            const uint32_t word = on ? packed[k] : 0xFFFFFFFFu;
            const float v = on ? pvals[k] : 0.0f;
            const int row = (int)(word >> kRowShift);       // off lanes: a row id no real row has
            float prod = on ? v * xp[word & (kLdsW - 1)] : 0.0f;
            // rows ascend inside a tile: fold the runs of equal rows (inclusive segmented scan), the last lane of a run adds
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const float t = __shfl_up(prod, d, kWave);
                const int r = __shfl_up(row, d, kWave);
                if (lane >= d && r == row) prod += t;
            }
            const int next = __shfl_down(row, 1, kWave);
            if (on && (lane == kWave - 1 || next != row)) ys[row] += prod;
        }
    }
    __syncthreads();
    for (int i = lane; i < n; i += kWave) y[row0 + i] = ys[i];
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
    return SPMV_OK;
}

}  // namespace

// Row blocks of equal NONZERO counts (nb0 cuts of row_ptr), any cut of more than `cap` rows split evenly:
// brow[0..nblocks] = first row of every block.  Shared with kernels_colsort.hip.
int panel_row_blocks(const spmv_csr &h, int64_t nb0, int cap, hipStream_t s, DevPtr<int32_t> &brow, int32_t *nblocks_out)
{
    DevPtr<int32_t> cut, nsub, total;
    SPMV_HIP_TRY(cut.alloc((size_t)nb0 + 1));
    SPMV_HIP_TRY(nsub.alloc((size_t)nb0));
    SPMV_HIP_TRY(total.alloc(1));
    const unsigned gb = (unsigned)((nb0 + 1 + 255) / 256);
    k_panel_cuts<<<dim3(gb), dim3(256), 0, s>>>(h.rows, h.nnz, (int)nb0, h.d_row_ptr, cut.p);
    int rc = check_launch("k_panel_cuts");
    if (rc) return rc;
    k_panel_nsub<<<dim3(gb), dim3(256), 0, s>>>((int)nb0, cap, cut.p, nsub.p);
    if ((rc = check_launch("k_panel_nsub"))) return rc;
    if ((rc = exclusive_scan_i32(nsub.p, nb0, total.p, s))) return rc;
    int32_t nblocks = 0;
    SPMV_HIP_TRY(hipMemcpyAsync(&nblocks, total.p, sizeof nblocks, hipMemcpyDeviceToHost, s));
    SPMV_HIP_TRY(hipStreamSynchronize(s));
    SPMV_HIP_TRY(brow.alloc((size_t)nblocks + 1));
    k_panel_brow<<<dim3(gb), dim3(256), 0, s>>>(h.rows, (int)nb0, cap, cut.p, nsub.p, total.p, brow.p);
    if ((rc = check_launch("k_panel_brow"))) return rc;
    SPMV_HIP_TRY(hipStreamSynchronize(s));   // the temporaries are freed on return
    *nblocks_out = nblocks;
    return SPMV_OK;
}

int panel_rowloc(const spmv_csr &h, const int32_t *d_brow, int nblocks, uint16_t *d_rowloc, hipStream_t s)
{
    k_panel_rowloc<<<dim3((unsigned)nblocks), dim3(256), 0, s>>>(d_brow, h.d_row_ptr, d_rowloc);
    return check_launch("k_panel_rowloc");
}

// tile_ptr[b * (np + 1) + p] = first position, in block-major / panel-minor order, of the nonzeros of row block b whose
// column lies in panel p (np <= 4096 panels of 2^pw_bits columns); entry np = end of the block.  Shared with kernels_binned.hip.
int panel_tile_ptr(const spmv_csr &h, const int32_t *d_brow, int nblocks, int pw_bits, int np, int32_t *d_tile_ptr, hipStream_t s)
{
    if (np > kMaxPanels) { set_error("panel_tile_ptr: %d panels (at most %d)", np, kMaxPanels); return SPMV_ERR_INVALID; }
    k_panel_tiles<<<dim3((unsigned)nblocks), dim3(256), 0, s>>>(d_brow, h.d_row_ptr, h.d_col_idx, pw_bits, np, d_tile_ptr);
    return check_launch("k_panel_tiles");
}

void destroy_panel(PanelPlan &p)
{
    destroy_binned(p);
    destroy_colsort(p);
    if (p.d_packed) (void)hipFree(p.d_packed);
    if (p.d_pvals) (void)hipFree(p.d_pvals);
    if (p.d_tile_ptr) (void)hipFree(p.d_tile_ptr);
    if (p.d_brow) (void)hipFree(p.d_brow);
    p = PanelPlan();
}

// one launch = one set of co-resident waves sweeping in step: 2 workgroups (4 waves) per CU
static int resident_waves(int device) { return device_cus(device) * kWgPerCu * kWavesPerWg; }

// spmv_csr_plan: idempotent like the other variants (spmv_csr_plan_set always re-plans: the way to refresh the copied values)
// ... unless the caller has announced new values (spmv_csr_values_changed): the copy is then rebuilt as it was planned
int plan_panel(spmv_csr &h, hipStream_t s) { return refresh_panel(h, h.plan_panel, s); }

// `dst` is one of the handle's two panel plans: plan_panel (SPMV_PANEL) or plan_auto_panel (what SPMV_AUTO chose: its
// own, so that planning AUTO never disturbs a PANEL plan the caller made, and the other way round)
int refresh_panel(spmv_csr &h, PanelPlan &dst, hipStream_t s)
{
    if (!dst.ready) return build_panel(h, dst, 0, 0, 0, s);
    if (dst.stamp.gen == h.values_gen) return SPMV_OK;
    if (dst.binned_mode) return build_panel(h, dst, dst.bin_rows, 0, dst.scatter_mode ? 5 : 4, s);
    if (dst.sorted_mode) return build_panel(h, dst, dst.sb_rows, dst.sb_waves, 3, s);
    return build_panel(h, dst, dst.pw_bits, dst.waves_per_launch, dst.lds_mode ? 2 : 1, s);
}

int plan_panel_with(spmv_csr &h, int want_bits, int want_waves, int want_mode, hipStream_t s)
{
    return build_panel(h, h.plan_panel, want_bits, want_waves, want_mode, s);
}

// want_mode: 0 = the rule below, 1 = panels through L2 (k_panel), 2 = panels staged in LDS (k_panel_lds),
//            3 = sorted blocks (kernels_colsort.hip; SPMV_PANEL_SORTED=1 makes it the rule's answer),
//            4 = binned (kernels_binned.hip: products streamed in panel order, summed per row block; want_bits = rows per bin),
//            5 = binned, the products stored in bin order by the product launch (thin tiles)
int build_panel(spmv_csr &h, PanelPlan &dst, int want_bits, int want_waves, int want_mode, hipStream_t s)
{
    destroy_panel(dst);
    PanelPlan p;
    if (want_mode == 4 || want_mode == 5) {      // binned: two streaming launches, no gather from memory (kernels_binned.hip)
        const int rc = plan_binned(h, p, want_bits, want_mode == 5, s);
        if (rc) { destroy_panel(p); return rc; }
        dst = p;
        return SPMV_OK;
    }
    {
        bool sorted = false;
        if (const char *e = getenv("SPMV_PANEL_SORTED")) sorted = atoi(e) != 0;
        if (want_mode == 3 || (want_mode == 0 && sorted)) {
            if (h.nnz > (int64_t)INT_MAX / 17 * 16 - 4096) {
                set_error("spmv_csr_plan(panel, sorted blocks): nnz %lld too close to 2^31 for one handle", (long long)h.nnz);
                return SPMV_ERR_INVALID;
            }
            const int rc = plan_colsort(h, p, want_mode == 3 ? want_bits : 0, want_mode == 3 ? want_waves : 0, s);
            if (rc) { destroy_panel(p); return rc; }
            dst = p;
            return SPMV_OK;
        }
    }
    {
        // LDS mode (k_panel_lds) is never chosen by the library: it was built to test whether staging the panels of a
        // SMALL x in LDS beats gathering them through L2 and measured slower where it had
        // its best chance -- config 2, uniform columns: 0.135 ms against 0.087 ms; config 3: 3.5 ms against 0.86 ms
        // ("Uniform columns").  It stays selectable (params[6] = 2, SPMV_PANEL_LDS=1) and tested.
        bool lds = false;
        if (const char *e = getenv("SPMV_PANEL_LDS")) lds = atoi(e) != 0;
        if (want_mode == 1) lds = false;
        if (want_mode == 2) lds = true;
        p.lds_mode = lds;
    }
    if (h.nnz > (int64_t)INT_MAX - 4 * kStepMax) {
        set_error("spmv_csr_plan(panel): nnz %lld too close to 2^31 for one handle", (long long)h.nnz);
        return SPMV_ERR_INVALID;
    }
    int bits = kColBits;   // 128Ki columns = 512 KiB of x per panel
    if (const char *e = getenv("SPMV_PANEL_BITS")) bits = atoi(e);
    if (want_bits > 0) bits = want_bits;
    if (bits < 8) bits = 8;
    if (bits > kColBits) bits = kColBits;
    while (bits < kColBits && ((h.cols + (1ll << bits) - 1) >> bits) > kMaxPanels) ++bits;
    if (p.lds_mode) {
        bits = kLdsPwBits;
        if (((h.cols + kLdsW - 1) >> kLdsPwBits) > kMaxPanels) {
            set_error("spmv_csr_plan(panel, LDS mode): %lld columns need more than %d panels of 2^%d columns", (long long)h.cols,
                      kMaxPanels, kLdsPwBits);
            return SPMV_ERR_INVALID;
        }
    }
    p.pw_bits = bits;
    p.npanels = (int)((h.cols + (1ll << bits) - 1) >> bits);
    if (p.npanels < 1) p.npanels = 1;
    if (p.npanels > kMaxPanels) {
        set_error("spmv_csr_plan(panel): %lld columns need more than %d panels of 2^%d columns", (long long)h.cols,
                  kMaxPanels, kColBits);
        return SPMV_ERR_INVALID;
    }
    // row blocks: equal nonzero counts, as many as fill the resident wave slots of every launch
    p.waves_per_launch = resident_waves(h.device);
    if (const char *e = getenv("SPMV_PANEL_WAVES")) {
        const int v = atoi(e);
        if (v > 0) p.waves_per_launch = v;
    }
    if (want_waves > 0) p.waves_per_launch = want_waves;
    if (p.lds_mode) p.waves_per_launch = device_cus(h.device) * kLdsWaves;   // one 16-wavefront workgroup per CU and round
    if (h.rows == 0) {
        p.ready = true;
        p.stamp.gen = h.values_gen;
        dst = p;
        return SPMV_OK;
    }
    const int64_t launches0 =
        (h.rows + (int64_t)p.waves_per_launch * kRwTarget - 1) / ((int64_t)p.waves_per_launch * kRwTarget);
    int64_t nb0 = launches0 * p.waves_per_launch;
    const int cap = p.lds_mode ? kRwLds : kRw;
    if (p.lds_mode) nb0 = (h.rows + kRwLdsTarget - 1) / kRwLdsTarget;
    if (nb0 > h.rows) nb0 = h.rows;
    DevPtr<int32_t> brow;
    int rc = panel_row_blocks(h, nb0, cap, s, brow, &p.nblocks);
    if (rc) return rc;
    // the step of the sweep: 1024 nonzeros where a step of 2048 would span eight tiles or more (see kVecMax)
    p.step_vecs = (double)h.nnz < 256.0 * (double)p.nblocks * (double)p.npanels ? 4 : 8;
    if (const char *e = getenv("SPMV_PANEL_STEP")) { const int v = atoi(e); if (v == 4 || v == 8) p.step_vecs = v; }

    DevPtr<uint32_t> packed;
    DevPtr<float> pvals;
    DevPtr<int32_t> tiles;
    DevPtr<uint16_t> rowloc;
    const size_t slots = (size_t)h.nnz + 2 * kStepMax + 8;   // the last step of a block reads past its end
    SPMV_HIP_TRY(packed.alloc(slots));
    SPMV_HIP_TRY(pvals.alloc(slots));
    SPMV_HIP_TRY(tiles.alloc((size_t)p.nblocks * (size_t)(p.npanels + 1)));
    SPMV_HIP_TRY(rowloc.alloc((size_t)h.nnz));
    SPMV_HIP_TRY(hipMemsetAsync(packed.p + h.nnz, 0, sizeof(uint32_t) * (slots - (size_t)h.nnz), s));
    SPMV_HIP_TRY(hipMemsetAsync(pvals.p + h.nnz, 0, sizeof(float) * (slots - (size_t)h.nnz), s));
    k_panel_tiles<<<dim3((unsigned)p.nblocks), dim3(256), 0, s>>>(brow.p, h.d_row_ptr, h.d_col_idx, p.pw_bits,
                                                                   p.npanels, tiles.p);
    if ((rc = check_launch("k_panel_tiles"))) return rc;
    if (h.nnz > 0) {
        k_panel_rowloc<<<dim3((unsigned)p.nblocks), dim3(256), 0, s>>>(brow.p, h.d_row_ptr, rowloc.p);
        if ((rc = check_launch("k_panel_rowloc"))) return rc;
        k_panel_fill<<<dim3((unsigned)((p.nblocks + 3) / 4)), dim3(256), 0, s>>>(
            p.nblocks, brow.p, h.d_row_ptr, h.d_col_idx, h.d_vals, rowloc.p, p.pw_bits, p.npanels, tiles.p, packed.p,
            pvals.p);
        if ((rc = check_launch("k_panel_fill"))) return rc;
        if (!p.lds_mode) {   // the LDS kernel folds equal rows itself
            k_panel_joins<<<dim3((unsigned)p.nblocks), dim3(256), 0, s>>>(brow.p, p.npanels, h.d_row_ptr, tiles.p,
                                                                           packed.p);
            if ((rc = check_launch("k_panel_joins"))) return rc;
        }
    }
    if ((rc = stamp_values(h, s, p.stamp))) return rc;
    SPMV_HIP_TRY(hipStreamSynchronize(s));   // the temporaries are freed on return
    p.d_packed = packed.release();
    p.d_pvals = pvals.release();
    p.d_tile_ptr = tiles.release();
    p.d_brow = brow.release();
    p.ready = true;
    dst = p;
    return SPMV_OK;
}

int panel_launches(const PanelPlan &p)
{
    if (p.binned_mode) return p.nblocks ? 2 : 0;
    if (p.lds_mode || p.sorted_mode) return p.nblocks ? 1 : 0;
    return p.nblocks && p.waves_per_launch ? (p.nblocks + p.waves_per_launch - 1) / p.waves_per_launch : 0;
}

int launch_panel(const spmv_csr &h, const float *x, float *y, hipStream_t s) { return launch_panel_plan(h, h.plan_panel, x, y, s); }

int launch_panel_plan(const spmv_csr &h, const PanelPlan &p, const float *x, float *y, hipStream_t s)
{
    if (!p.ready) {
        set_error("spmv_csr_run: variant panel is not planned (call spmv_csr_plan first)");
        return SPMV_ERR_NOT_PLANNED;
    }
    if (int rc = require_fresh_values(h, p.stamp, s, "panel")) return rc;
    if (p.binned_mode) return launch_binned(h, p, x, y, s);
    if (p.sorted_mode) return launch_colsort(h, p, x, y, s);
    if (p.lds_mode) {
        if (p.nblocks == 0) return SPMV_OK;
        const size_t lds = sizeof(float) * (size_t)(2 * kLdsW + kLdsWaves * kRwLds);
        static LdsOptIn optin;
        if (int rc = optin.ensure(reinterpret_cast<const void *>(&k_panel_lds), h.device, (int)lds)) return rc;
        k_panel_lds<<<dim3((unsigned)((p.nblocks + kLdsWaves - 1) / kLdsWaves)), dim3(kLdsWaves * kWave), lds, s>>>(
            p.nblocks, h.cols, p.d_brow, p.d_tile_ptr, p.d_packed, p.d_pvals, x, y, p.npanels);
        return check_launch("k_panel_lds");
    }
    // equal shares: 1536 blocks on 1024 slots run as 768 + 768, not 1024 + 512
    const int launches = panel_launches(p);
    if (launches == 0) return SPMV_OK;   // no rows
    const int share = (p.nblocks + launches - 1) / launches;
    for (int b0 = 0; b0 < p.nblocks; b0 += share) {
        const int b1 = b0 + share < p.nblocks ? b0 + share : p.nblocks;
        const int g = (b1 - b0 + kWavesPerWg - 1) / kWavesPerWg;
        if (p.step_vecs == 4)
            k_panel<4><<<dim3((unsigned)g), dim3(kWave * kWavesPerWg), 0, s>>>(b0, b1, p.d_brow, h.d_row_ptr, p.d_tile_ptr,
                                                                               p.d_packed, p.d_pvals, x, y, p.npanels, p.pw_bits);
        else
            k_panel<8><<<dim3((unsigned)g), dim3(kWave * kWavesPerWg), 0, s>>>(b0, b1, p.d_brow, h.d_row_ptr, p.d_tile_ptr,
                                                                               p.d_packed, p.d_pvals, x, y, p.npanels, p.pw_bits);
        const int rc = check_launch("k_panel");
        if (rc) return rc;
    }
    return SPMV_OK;
}

}  // namespace spmv
