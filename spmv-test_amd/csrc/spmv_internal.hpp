// spmv_internal.hpp -- shared declarations of libspmv_hip.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include "spmv_hip.h"

namespace spmv {

// ---- error plumbing -------------------------------------------------------
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define SPMV_HIP_TRY(call)                                                   \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) return ::spmv::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

// Device allocation that frees itself unless release()d into a handle.
template <typename T>
struct DevPtr {
    T *p = nullptr;
    DevPtr() = default;
    DevPtr(const DevPtr &) = delete;
    DevPtr &operator=(const DevPtr &) = delete;
    ~DevPtr() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc((void **)&p, sizeof(T) * (count ? count : 1)); }
    T *release() { T *q = p; p = nullptr; return q; }
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): a function-local static of this type
// remembers the devices a kernel has been opted in on (one bit per device id; ids >= 64 set it on every launch).
struct LdsOptIn {
    std::atomic<uint64_t> done{0};
    int ensure(const void *fn, int device, int bytes);
};
int device_cus(int device);   // compute units of a device (cached per device id; 256 if the query fails)
// SPMV_OK when `device` (where a handle's arrays live) is the calling thread's current device
int require_current(int device, const char *what);

// ---- geometry constants ----------------------------------------------------
constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kBlock = 256;        // 4 waves: one per SIMD of a CU
constexpr int kXcds = 8;           // MI355X: 8 XCDs, blocks dealt round-robin over them

// ADAPTIVE / TILED chunking: every lane streams kNnzPerThread consecutive-by-4 nonzeros
// (col_idx + vals = 8 B each) with 16-byte loads; a workgroup of B threads owns B*16 nonzeros.
// (the SPMV_T_* macros exist for A/B builds, tools/explore.py; the shipped values are the defaults)
#ifndef SPMV_T_NPT
#define SPMV_T_NPT 16
#endif
#ifndef SPMV_T_SHORT
#define SPMV_T_SHORT 32
#endif
constexpr int kNnzPerThread = SPMV_T_NPT;
constexpr int kShortSeg = SPMV_T_SHORT;            // row segments up to this long: one lane sums them

// Chunk boundaries (and, for TILED, column windows) for workgroups of `block` threads.
struct ChunkPlan {
    int block = 0;             // 0 = not planned; 256 | 512 | 1024 threads per workgroup
    int nchunks = 0;           // ceil(nnz / (block*16))
    int32_t *d_lb = nullptr;   // [nchunks+1] first row whose row_ptr >= c*chunk
    float *d_carry = nullptr;  // [nchunks]   partial sum of the row continued from chunk c-1
    int32_t *d_win = nullptr;  // [2*nchunks+2] TILED: first column, window length (0 = not staged); stats
    // 16-bit columns (TILED): chunks whose span is fully staged also carry col - w0 as uint16
    uint16_t *d_col16 = nullptr;   // [nchunks * chunk], lane layout of k_tiled16
    int32_t *d_list16 = nullptr;   // [n16] chunks run by k_tiled16
    int32_t *d_list32 = nullptr;   // [nchunks - n16 - nsorted] the others, run by the 32-bit body
    int n16 = 0;
    // sorted chunks (TILED): spans of several LDS regions gather x in column order instead of staging it
    uint32_t *d_perm = nullptr;        // [nsorted * chunk] position << 18 | column - w0 of the sorted chunks (list order), read INSTEAD of col_idx
    int32_t *d_list_sorted = nullptr;  // [nsorted]
    int nsorted = 0;                   // chunks run by the sorted body
    int nsorted_marked = 0;            // chunks the window pass marked (a few may still switch to a block list)
    int sorted_from = 0;               // -1: sorted where the modelled cost is lower; 0: never; n: from n passes on
    int long_piece_chunks = 0;         // chunks of three passes or more that hold a piece of a long row (> 512 nonzeros)
    double model_cost = 0.0;           // modelled time per nonzero of this plan (PlanCost units; compares block sizes)
    int32_t *d_blk = nullptr;      // [64 * nchunks] ids of the staged 1024-column blocks of the chunks that use a list
    int nblk_chunks = 0;           // how many chunks do
    int maxpass = 0;
    bool col16_wanted = false;     // the plan asked for the 16-bit copy (it is kept only where enough chunks qualify)
    int spanning_rows = 0;         // rows that continue past their owner chunk (0: no fix-up launch)
    int region = 0;                // floats of the dynamic LDS region (x slice, then products)
    bool persist = false;      // persistent software-pipelined launch (measured slower: DESIGN.md section 4)
    int staged_single = 0;     // TILED: chunks whose whole column span is staged in one pass
    int staged_full = 0;       // TILED: chunks staged completely (any number of passes)
};

// Plans that COPY the values (PANEL, XSKIP) remember which state of vals they copied: the handle's generation counter
// (spmv_csr_values_changed bumps it) and, under SPMV_CHECK_VALUES=1, a checksum of the array itself.
struct ValuesStamp {
    uint64_t gen = 0;
    uint64_t sum = 0;
    bool have_sum = false;
};

// SPMV_PANEL (kernels_panel.hip): row blocks of at most 8192 rows and equal nonzero counts, each block's nonzeros stably sorted by column panel.
struct PanelPlan {
    bool ready = false;
    bool lds_mode = false;         // panels of 2^14 columns staged in LDS by 16-wavefront workgroups (small x) instead of gathered through L2
    int pw_bits = 0;               // log2(columns per panel)
    int npanels = 0;
    int nblocks = 0;               // row blocks (<= 8192 rows, equal nonzero counts) = wavefronts of work
    int waves_per_launch = 0;      // what is resident at once: one launch = one sweep in step
    int step_vecs = 8;             // 16-byte vectors per lane and step of the sweep (8 | 4: thin tiles, kernels_panel.hip kVecMax)
    uint32_t *d_packed = nullptr;  // [nnz + slack] row_in_block << 18 | join << 17 | column_in_panel (kernels_panel.hip)
    float *d_pvals = nullptr;      // [nnz + slack] values in the same order (a COPY: re-plan after changing vals)
    int32_t *d_tile_ptr = nullptr; // [nblocks * (npanels + 1)]
    int32_t *d_brow = nullptr;     // [nblocks + 1] first row of every block
    ValuesStamp stamp;             // which state of vals d_pvals is a copy of
    // sorted blocks (mode 3, kernels_colsort.hip): blocks of <= 4096 rows streamed in column order, d_packed / d_pvals in
    // units of 256 slots (four groups of 64 nonzeros with distinct rows)
    bool sorted_mode = false;
    int32_t *d_ubeg = nullptr;     // [nblocks + 1] first unit of every block
    int32_t *d_usimple = nullptr;  // [nblocks] leading units of the block that hold groups (64 distinct rows per instruction)
    int32_t *d_uend = nullptr;     // [nblocks] one past the last unit the block uses (groups, then its row-sorted tail)
    int32_t *d_ubase = nullptr;    // [units] first column of every unit (packed holds offsets from it)
    int32_t *d_tbeg = nullptr;     // [nblocks] first tail unit of the block in d_trow
    uint16_t *d_trow = nullptr;    // [tail_units * 256] rows of the tail units (their packed words are whole columns)
    int64_t tail_units = 0;
    int sb_rows = 0, sb_waves = 0;   // rows per block (4096 | 8192), wavefronts per workgroup (8 | 4)
    int64_t wide_blocks = 0;       // blocks whose short rows span more than 32768 lines of x (4 MiB)
    int64_t units = 0;
    int64_t lines = 0;             // occupied 128-byte lines of x, summed over the blocks (plan statistic)
    int64_t tail = 0;              // nonzeros in the tails (long rows, what could not be grouped)
    // binned (mode 4, kernels_binned.hip): the nonzeros twice over -- in PANEL-major order (d_c16 + d_pvals: what the
    // product launch streams, its panel of x in LDS) and in BIN-major order (d_r16: what the sum launch streams beside the
    // products); tile (bin b, panel p) is contiguous in both, rows ascending inside it
    bool binned_mode = false;
    int bin_rows = 0;              // most rows of a bin (4096 | 8192): a wavefront's private sums in LDS
    uint16_t *d_c16 = nullptr;     // [padded] column - panel * 2^pw_bits, panel-major (every panel padded to a multiple of 8)
    uint16_t *d_r16 = nullptr;     // [nnz] row - brow[bin], bin-major (d_tile_ptr positions)
    int32_t *d_pm = nullptr;       // [nblocks * npanels] panel-major position of tile (b, p)
    int32_t *d_pbase = nullptr;    // [npanels + 1] panel-major position of every panel's first entry
    float *d_prod = nullptr;       // [padded] scratch of a run: the products, panel-major
    int64_t padded = 0;            // entries of the panel-major arrays
    int splits = 1;                // workgroups per panel of the product launch
    bool wide_pieces = false;      // the sum launch takes four products per lane (fat tiles) instead of two
    int flagged_tiles = 0;         // tiles that go through the fold (bins whose long rows need more spare sums than there are)
    int long_rows = 0;             // rows with spare sums (more products in some tile than a lane takes)
    int32_t *d_lptr = nullptr;     // [nblocks + 1] first long row of every bin in d_lrow / d_lcnt
    uint32_t *d_lrow = nullptr;    // [long_rows] row in bin << 16 | first spare slot
    int32_t *d_lcnt = nullptr;     // [long_rows] spare slots of the row
    // ... its scattered flavour: the product launch stores in bin order (d_offset, d_first_run; bit 15 of d_c16), the sum launch streams d_prod + d_r16 (here:
    // accumulator numbers) bin by bin; d_lrow = [nblocks * 1024] row << 17 | first spare accumulator << 7 | how many
    bool scatter_mode = false;
    int32_t *d_offset = nullptr;   // [runs] bin-major minus panel-major position of a run (a nonempty tile, or a panel's pad slots)
    int32_t *d_first_run = nullptr;   // [padded / 512] run of every 512-entry block's first entry (minus one where it starts a run)
    int64_t runs = 0;
    int32_t *d_bbase = nullptr;    // [nblocks + 1] first entry of every bin in d_prod / d_r16 (multiples of 256)
    int32_t *d_bcnt = nullptr;     // [nblocks] entries of the bin
    int32_t *d_nlong = nullptr;    // [nblocks] rows with spare accumulators (-1: the bin adds with LDS atomics)
    int64_t bm_entries = 0;        // entries of the bin-major arrays
};

// SPMV_XSKIP (kernels_xskip.hip): the matrix in input-major segments per block of 1024 outputs
struct XskipPlan {
    bool ready = false;
    int nblocks = 0, nseg = 0, slabs = 0;
    int32_t *d_block_seg = nullptr;   // [nblocks+1] first segment of every output block
    int32_t *d_seg_input = nullptr;   // [nseg+1] input (column) of a segment
    int32_t *d_seg_ptr = nullptr;     // [nseg+1] first entry of a segment
    uint16_t *d_erow = nullptr;       // [nnz] output - 1024 * block
    float *d_evals = nullptr;         // [nnz] values in segment order (a COPY: re-plan after changing vals)
    float *d_part = nullptr;          // [nblocks * slabs * 1024] slab partials (slabs > 1)
    ValuesStamp stamp;                // which state of vals d_evals is a copy of
};

// SPMV_WAVE_PIPE (kernels_rows.hip): the rows too long for a wavefront's bundle, cut into pieces for the whole chip
struct WavePlan {
    bool ready = false;
    int n_long = 0, pieces = 0;
    int32_t *d_long_row = nullptr;    // [n_long] the rows, ascending
    int32_t *d_long_first = nullptr;  // [n_long + 1] first piece of every row
    int32_t *d_piece_k0 = nullptr;    // [pieces] first nonzero of a piece
    int32_t *d_piece_len = nullptr;   // [pieces] its length (<= 1024)
    float *d_partial = nullptr;       // [pieces] scratch of a run: the pieces' sums
    int block_rows = 512;             // rows of a bundle workgroup (512 | 1024)
    int32_t *d_blk_lo = nullptr;      // [blocks] first entry of the x window of every block of block_rows rows, -1: none
    uint16_t *d_col16 = nullptr;      // [nnz] 16-bit column offsets: from blk_lo for windowed blocks' short rows, from piece_base for pieces (null: not built)
    int32_t *d_piece_base = nullptr;  // [pieces] smallest column of a piece whose offsets are in d_col16, -1: 32-bit columns
    bool windows = false;             // the bundle kernel stages windows (at least half of the blocks have one)
    int64_t blocks = 0, win_blocks = 0;   // blocks of block_rows rows, and how many have a window
};

}  // namespace spmv

struct spmv_tcsr;    // kernels_tcsr.hip
struct spmv_bitmap;  // kernels_bitmap.hip

// The opaque handle of include/spmv_hip.h.
struct spmv_csr {
    int64_t rows = 0, cols = 0, nnz = 0;
    const int32_t *d_row_ptr = nullptr;
    const int32_t *d_col_idx = nullptr;
    const float *d_vals = nullptr;
    bool owns_arrays = false;
    int device = 0;

    // plan state
    int vector_width = 0;          // SPMV_VECTOR: lanes per row (2..32), 0 = not planned
    spmv::ChunkPlan plan_adaptive; // SPMV_ADAPTIVE: 256-thread workgroups
    spmv::ChunkPlan plan_tiled;    // SPMV_TILED: workgroup size chosen from the column windows
    spmv::PanelPlan plan_panel;    // SPMV_PANEL
    spmv::PanelPlan plan_auto_panel;   // SPMV_AUTO where it resolved to the panel family (its own: see refresh_panel)
    bool auto_made_tiled = false;  // SPMV_AUTO made the TILED plan it looked at (and may release it)
    spmv::XskipPlan plan_xskip;    // SPMV_XSKIP
    spmv::WavePlan plan_wave;      // SPMV_WAVE_PIPE
    int auto_variant = -1;         // SPMV_AUTO: the variant its plan chose (-1 = not planned)
    uint64_t values_gen = 0;       // bumped by spmv_csr_values_changed: plans that copied vals before that are stale
};

namespace spmv {

// ---- kernel launchers (each enqueues on `s`, returns a status) -------------
int launch_scalar(spmv_csr &h, const float *x, float *y, hipStream_t s);
int launch_wave(spmv_csr &h, const float *x, float *y, bool pipelined, hipStream_t s);
int plan_wave(spmv_csr &h, hipStream_t s);
void destroy_wave(WavePlan &p);
int launch_vector(const spmv_csr &h, const float *x, float *y, hipStream_t s);
int launch_adaptive(const spmv_csr &h, const float *x, float *y, bool tiled, hipStream_t s);
int launch_panel(const spmv_csr &h, const float *x, float *y, hipStream_t s);

int plan_xskip(spmv_csr &h, hipStream_t s);
int launch_xskip(const spmv_csr &h, const float *x, float *y, hipStream_t s);
void destroy_xskip(XskipPlan &p);
int plan_panel(spmv_csr &h, hipStream_t s);
void destroy_panel(PanelPlan &p);
int panel_launches(const PanelPlan &p);
int plan_vector(spmv_csr &h, hipStream_t s);
int plan_adaptive(spmv_csr &h, bool tiled, hipStream_t s);
// TILED with exactly these parameters (spmv_csr_plan_set); block 256|512|1024, maxpass >= 1
int plan_tiled_with(spmv_csr &h, int block, int maxpass, bool col16, hipStream_t s);
int plan_adaptive_with(spmv_csr &h, int block, hipStream_t s);
int plan_panel_with(spmv_csr &h, int pw_bits, int waves_per_launch, int mode, hipStream_t s);   // 0 = library default
int build_panel(spmv_csr &h, PanelPlan &dst, int pw_bits, int waves_per_launch, int mode, hipStream_t s);
int refresh_panel(spmv_csr &h, PanelPlan &dst, hipStream_t s);
int launch_panel_plan(const spmv_csr &h, const PanelPlan &p, const float *x, float *y, hipStream_t s);
// kernels_panel.hip helpers shared with kernels_colsort.hip
int panel_row_blocks(const spmv_csr &h, int64_t nb0, int cap, hipStream_t s, DevPtr<int32_t> &brow, int32_t *nblocks);
int panel_rowloc(const spmv_csr &h, const int32_t *d_brow, int nblocks, uint16_t *d_rowloc, hipStream_t s);
// kernels_colsort.hip: SPMV_PANEL mode 3
int plan_colsort(spmv_csr &h, PanelPlan &p, int want_rows, int want_waves, hipStream_t s);
double colsort_model_cost(const PanelPlan &p, int64_t nnz);
double colsort_cost(int rows_per_block, double lines_per_nnz, double tail_frac);
int colsort_probe(const spmv_csr &h, hipStream_t s, double *long_frac, double *wide_frac, double *lines_per_nnz);
int launch_colsort(const spmv_csr &h, const PanelPlan &p, const float *x, float *y, hipStream_t s);
void destroy_colsort(PanelPlan &p);
// kernels_binned.hip: SPMV_PANEL mode 4
int panel_tile_ptr(const spmv_csr &h, const int32_t *d_brow, int nblocks, int pw_bits, int np, int32_t *d_tile_ptr, hipStream_t s);
int plan_binned(spmv_csr &h, PanelPlan &p, int want_rows, bool scatter, hipStream_t s);
int launch_binned(const spmv_csr &h, const PanelPlan &p, const float *x, float *y, hipStream_t s);
void destroy_binned(PanelPlan &p);
double binned_tile_nonzeros(const spmv_csr &h, int bin_rows);
void destroy_plans(spmv_csr &h);
void drop_tiled_plan(spmv_csr &h);   // SPMV_AUTO resolved to another variant: the TILED plan it looked at is released

int dense_to_csr(int M, int N, const float *d_A, hipStream_t s, spmv_csr_t **out);
int dense_gemv(int M, int N, const float *d_A, const float *d_x, float *d_y, int mode, hipStream_t s);
// the reference's ASP layout (kernels_dense.hip): re-tile, and the multiply from it with the x == 0 skip
int asp_retile(int M, int N, const float *d_A, float *d_asp, hipStream_t s);
int asp_gemv_ws(int M, int N, const float *d_asp, const float *d_x, float *d_y, void *d_ws, size_t ws_bytes, hipStream_t s);
size_t dense_gemv_workspace_bytes(int N, int mode);
int dense_gemv_ws(int M, int N, const float *d_A, const float *d_x, float *d_y, int mode, void *d_ws, size_t ws_bytes,
                  hipStream_t s);

// kernels_rows.hip: structural check of a CSR (bad[0] first row with row_ptr[r] > row_ptr[r+1] or outside [0,nnz],
// bad[1] first element with a column outside [0,cols), bad[2]/bad[3] row_ptr[0] / row_ptr[rows] when wrong)
int launch_validate(const spmv_csr *h, int32_t *d_bad4, hipStream_t stream);
// order-sensitive 64-bit checksum of vals (a device pass + a host wait: debug aid behind SPMV_CHECK_VALUES=1)
int values_checksum(const spmv_csr &h, hipStream_t s, uint64_t *out);
bool check_values_env();   // SPMV_CHECK_VALUES=1 (read once)
// stamp a plan that has just copied vals / refuse to run one whose copy is out of date (SPMV_ERR_STALE_PLAN)
int stamp_values(const spmv_csr &h, hipStream_t s, ValuesStamp &st);
int require_fresh_values(const spmv_csr &h, const ValuesStamp &st, hipStream_t s, const char *variant);
// min / max of col_idx (kernels_rows.hip): d_out[0] = min (INT_MAX when nnz = 0), d_out[1] = max (-1)
int launch_column_range(const spmv_csr &h, int32_t *d_out2, hipStream_t s);
// in-place exclusive scan of n int32 (one 1024-thread workgroup); the total goes to *d_total
int exclusive_scan_i32(int32_t *d_data, int64_t n, int32_t *d_total, hipStream_t s);

int tcsr_from_dense(int M, int N, const float *d_A, hipStream_t s, spmv_tcsr_t **out);
int tcsr_run(const spmv_tcsr &h, const float *d_x, float *d_y, hipStream_t s);
int tcsr_sizes(const spmv_tcsr &h, int64_t *n_blk_idx, int64_t *n_bitmaps, int64_t *n_vals);
int tcsr_download(const spmv_tcsr &h, int32_t *blk_idx, uint32_t *bitmaps, float *vals);
void tcsr_free(spmv_tcsr *h);
void tcsr_dims(const spmv_tcsr &h, int *M, int *N);

// kernels_bitmap.hip: the reference's WSP / AWSP / AWSPRef formats
int bitmap_from_dense(int format, int M, int N, const float *d_A, hipStream_t s, spmv_bitmap_t **out);
int bitmap_run(const spmv_bitmap &h, const float *d_x, float *d_y, hipStream_t s);
void bitmap_info(const spmv_bitmap &h, int *format, int *M, int *N, int64_t *n_bitmaps, int64_t *n_vals, int32_t stats[4]);
int bitmap_download(const spmv_bitmap &h, uint32_t *bitmaps, float *vals);
int bitmap_device(const spmv_bitmap &h);
void bitmap_free(spmv_bitmap *h);

int synth_fill(uint64_t seed, int64_t row0, int64_t n_local, int64_t rows, int64_t cols, int64_t band,
               const int32_t *d_row_ptr, int32_t *d_col_idx, float *d_vals, hipStream_t s);
int synth_x(uint64_t seed, int64_t j0, int64_t n, float *d_x, hipStream_t s);


#if defined(__HIPCC__)
// Stable scatter by key, 64 entries per step (k_panel_fill, k_bin_fill, k_bs_fill): which lanes hold a key that NO other lane of
// the step holds?  They take their cursor and bump it themselves; only the keys held twice go through the ballot loop -- on
// uniform columns over thousands of panels that is one or two trips instead of 64.  `tags` = 256 words of the wavefront's
// own LDS, hashed by the key's low bits: two keys in one slot both go to the loop (conservative, never wrong).
__device__ __forceinline__ bool lone_in_step(volatile int *tags, bool valid, int key, int lane)
{
    const int slot = key & 255;
    if (valid) tags[slot] = lane;
    __builtin_amdgcn_wave_barrier();
    const int first = valid ? tags[slot] : lane;
    __builtin_amdgcn_wave_barrier();
    if (valid && first != lane) tags[slot] = kWave;                 // somebody else's slot too: nobody in it is alone
    __builtin_amdgcn_wave_barrier();
    const int second = valid ? tags[slot] : kWave;
    __builtin_amdgcn_wave_barrier();
    return valid && second == lane;
}
#endif

}  // namespace spmv
