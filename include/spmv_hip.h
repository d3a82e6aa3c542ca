/*
 * spmv_hip.h -- C ABI of libspmv_hip.so: the MI355X (gfx950) fp32 CSR SpMV hot path.
 *
 * This is the drop-in boundary underneath the reference's launcher surface
 * (/root/reference/src/include/kernel.hpp:8-17).  The reference launchers are
 * C++ free functions that take a dense host matrix; include/kernel.hpp in this
 * repo re-declares them unchanged and spmv-test_amd/host/launchers.cpp
 * implements them on top of the entry points below.  Every entry point is
 * extern "C", takes plain pointers and sizes, and returns 0 on success or a
 * negative spmv_status code; spmv_last_error() then holds a message.  Nothing
 * here falls back to the CPU: without a HIP device every compute entry point
 * fails with SPMV_ERR_NO_DEVICE.
 *
 * Conventions (from matrix_csr.cpp:8-22):
 *   CSR row i   = output index i in [0, rows)   (column i of the dense A)
 *   column idx  = input index j in [0, cols)    (row j of the dense A)
 *   y[i] = sum_k vals[k] * x[col_idx[k]],  k in [row_ptr[i], row_ptr[i+1])
 *   row_ptr has rows+1 int32 entries (row_ptr[rows] == nnz); the reference's
 *   CSRMatrix omits the last one and csr_naive.cu:15 substitutes nnz for it.
 *   nnz < 2^31 per handle; larger problems are row-block shards, one handle each.
 */
#ifndef SPMV_HIP_H
#define SPMV_HIP_H

#include <stdint.h>

#if defined(__GNUC__)
#define SPMV_API __attribute__((visibility("default")))
#else
#define SPMV_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spmv_csr spmv_csr_t;

enum spmv_status {
    SPMV_OK = 0,
    SPMV_ERR_NO_DEVICE = -1, /* no HIP device visible: there is no CPU path      */
    SPMV_ERR_INVALID = -2,   /* bad argument (null, negative size, misaligned)   */
    SPMV_ERR_HIP = -3,       /* a HIP runtime call failed (message has details)  */
    SPMV_ERR_VARIANT = -4,   /* unknown variant id (reference: silently no-op,   */
                             /* wsp.cu:187 -- here it is reported)               */
    SPMV_ERR_NOT_PLANNED = -5,
    SPMV_ERR_STALE_PLAN = -6 /* the variant's plan holds a COPY of vals taken before   */
                             /* spmv_csr_values_changed (or, under SPMV_CHECK_VALUES=1,*/
                             /* before the array changed): re-plan                     */
};

/* Kernel variants.  The right-hand column is the reference slot each one
 * replaces (file:line in /root/reference) -- same role, re-derived for CSR on
 * 64-lane wavefronts, not a translation of the bitmap kernels. */
enum spmv_variant {
    SPMV_SCALAR = 0,    /* thread per row, k ascending, mul+add unfused:           */
                        /*   csr_naive_kernel      src/kernels/csr_naive.cu:6-23   */
                        /*   (bit-identical to SgemvCPU, src/tester.cpp:36-45).    */
                        /* Shares SPMV_WAVE_PIPE's plan (operands only; see there). */
    SPMV_WAVE = 1,      /* 64-lane wavefronts + __shfl_down reduction:             */
                        /*   wsp_kernel_v0         src/kernels/wsp.cu:4-56         */
                        /* Rows of mean > 32 nonzeros: a wavefront per row.  Shorter */
                        /* rows (round 4): a wavefront per 64 rows as in            */
                        /* SPMV_WAVE_PIPE -- coalesced streams, a lane per short row, */
                        /* the wave + __shfl_down per longer one, the longest in     */
                        /* pieces -- but x gathered from memory: no window in LDS,   */
                        /* no 16-bit offsets (the plain kernel of the pair).  Shares */
                        /* SPMV_WAVE_PIPE's plan there (the long rows' pieces).      */
    SPMV_WAVE_PIPE = 2, /* a wavefront per 64 rows: their nonzeros streamed        */
                        /* coalesced with all loads of a run in flight, products    */
                        /* parked in LDS, a lane per short row, the wave +          */
                        /* __shfl_down per longer one; rows of more than 512        */
                        /* nonzeros in pieces, a wavefront per piece; x from a      */
                        /* window in LDS per block of rows where the columns fit:   */
                        /*   wsp_kernel_v1         src/kernels/wsp.cu:59-138       */
                        /* Plan: the long rows and the windows, a function of       */
                        /* row_ptr and col_idx (NOT of the values).  A handle that  */
                        /* was not planned plans on its first run (allocates and    */
                        /* waits for the stream once: call spmv_csr_plan before     */
                        /* capturing a graph).                                      */
    SPMV_VECTOR = 3,    /* 2..32-lane groups per row, width from mean row length:  */
                        /*   asp_kernel_v0/1/2     src/kernels/asp.cu:6-211        */
    SPMV_ADAPTIVE = 4,  /* nnz-balanced chunks, products staged in LDS, per-chunk  */
                        /* adaptive row reduction, deterministic carry fix-up:     */
                        /*   awsp_kernel_v0/1/2    src/kernels/awsp.cu:5-317,      */
                        /*   awsp_ref_kernel       src/kernels/awsp_ref.cu:6-185   */
    SPMV_TILED = 5,     /* ADAPTIVE + the chunk's window of x staged in LDS:       */
                        /*   csr_tiling_kernel     src/kernels/csr_tiling.cu:24-114,*/
                        /*   wsp_sm_kernel         src/kernels/wsp_sm.cu:6-211     */
    SPMV_PANEL = 6,     /* (row block x column panel) sweep for columns without       */
                        /* locality: a wavefront keeps its block's sums in LDS and all */
                        /* resident waves walk the panels of x in step, so the panel   */
                        /* stays in L2.  The sparse-scale counterpart of the reference's*/
                        /* tiled format: TCSRMatrix src/tcsr.cpp:5-38 + csr_tiling_kernel*/
                        /* src/kernels/csr_tiling.cu:24-114.  Its plan COPIES the values */
                        /* (re-plan after changing them).                               */
    SPMV_AUTO = 7,      /* the library chooses at plan time: SPMV_TILED where its plan can stage the    */
                        /* chunks' x windows in LDS, SPMV_PANEL where it cannot (columns without        */
                        /* locality) and x is larger than one XCD's L2.  spmv_csr_plan_describe names   */
                        /* the choice.  Role: the reference's adaptive slot awsp_gemv_gpu               */
                        /* (src/kernels/awsp.cu:319-388), which the launcher of that name now runs.     */
    SPMV_XSKIP = 8,     /* activation sparsity on the CSR side: the plan re-orders the matrix into input-major */
                        /* segments per block of 1024 outputs; a run skips, whole, the segments of the inputs    */
                        /* whose x is zero (6 contiguous bytes per nonzero never read) and sums the rest in LDS. */
                        /*   the `x_i != 0` skip of asp_kernel_v* (src/kernels/asp.cu:20-26), awsp_kernel_v1     */
                        /*   (awsp.cu:127-134), awsp_ref_kernel (awsp_ref.cu:52).  For dense-ish matrices (the    */
                        /*   reference's regime): the plan refuses when ceil(rows/1024) x cols exceeds 2^27, and  */
                        /*   rows must be sorted and duplicate-free.  Its plan COPIES the values, like PANEL.  */
    SPMV_VARIANT_COUNT = 9
};

/* ---- runtime ---------------------------------------------------------- */
SPMV_API int spmv_device_count(void);            /* >= 0; 0 when no HIP device is usable */
SPMV_API const char *spmv_last_error(void);      /* thread-local, never NULL             */
SPMV_API const char *spmv_variant_name(int variant);
/* The event time of the FIRST launch of the last spmv_*_run_host call on this thread: cold, code-object load and first-touch
 * included -- the figure the reference's TIME_KERNEL prints (kernel.hpp:31-48 times one cold launch).  *kernel_ms of those
 * calls is the second, warm launch.  The launchers of include/kernel.hpp print both. */
SPMV_API float spmv_last_first_launch_ms(void);

/* ---- matrix handles ---------------------------------------------------
 * replaces: CSRMatrix (src/matrix_csr.cpp:5-23, src/include/matrix_csr.hpp:4-25)
 * and the cudaMalloc/cudaMemcpy block of every launcher (e.g. csr_naive.cu:36-52). */

/* Copy host CSR arrays to the current device (owned by the handle). */
SPMV_API int spmv_csr_create_host(int64_t rows, int64_t cols, int64_t nnz, const int32_t *row_ptr,
                         const int32_t *col_idx, const float *vals, spmv_csr_t **out);

/* Borrow device arrays (caller keeps them alive; col_idx/vals 16-byte aligned). */
SPMV_API int spmv_csr_create_device(int64_t rows, int64_t cols, int64_t nnz, const int32_t *d_row_ptr,
                           const int32_t *d_col_idx, const float *d_vals, spmv_csr_t **out);

/* Dense row-major host A[M][N] -> CSR of A^T, built ON THE DEVICE with the
 * reference's semantics (keep iff value != 0.0f, columns ascending):
 * replaces the O(MN) host scan of matrix_csr.cpp:5-23.  rows = N, cols = M. */
SPMV_API int spmv_csr_from_dense_host(int M, int N, const float *A_host, void *stream, spmv_csr_t **out);

/* Same with A already resident on the device. */
SPMV_API int spmv_csr_from_dense_device(int M, int N, const float *d_A, void *stream, spmv_csr_t **out);

/* Copy the handle's CSR arrays back (row_ptr: rows+1 entries).  Any pointer may be NULL. */
SPMV_API int spmv_csr_download(const spmv_csr_t *h, int32_t *row_ptr, int32_t *col_idx, float *vals);

/* Check the arrays on the device: row_ptr[0] == 0, row_ptr non-decreasing, row_ptr[rows] == nnz, every column
 * index in [0, cols).  SPMV_ERR_INVALID + spmv_last_error() names the first offending row / element.  Both
 * spmv_csr_create_* call it (one pass over row_ptr and col_idx, synchronous): the kernels index x and their LDS
 * windows with these numbers unchecked, so a malformed matrix must never reach them.  The reference has no
 * counterpart (its CSR only ever comes from its own dense scan, matrix_csr.cpp:5-23). */
SPMV_API int spmv_csr_validate(const spmv_csr_t *h, void *stream);

SPMV_API int spmv_csr_dims(const spmv_csr_t *h, int64_t *rows, int64_t *cols, int64_t *nnz);

/* The smallest and the largest column index the matrix references (one device pass over col_idx, synchronous):
 * *col_min = cols, *col_max = -1 for a matrix without nonzeros.  A row block's x footprint -- what a sharded multiply has
 * to make available to the rank that holds it (include/spmv_dist.h: spmv_dist_pipe_set_footprint). */
SPMV_API int spmv_csr_column_range(const spmv_csr_t *h, int64_t *col_min, int64_t *col_max, void *stream);
SPMV_API int spmv_csr_destroy(spmv_csr_t *h);

/* ---- the hot path ------------------------------------------------------
 * spmv_csr_plan: one-off device-side preprocessing a variant needs (chunk
 * boundaries, column windows, for SPMV_TILED also a 16-bit copy of the column
 * indices; workgroup size and pass budget follow from the chunk statistics,
 * with SPMV_AUTOTUNE=1 from timed trial launches instead); SPMV_SCALAR, SPMV_WAVE
 * and SPMV_WAVE_PIPE share one (see the enum; a no-op for SPMV_WAVE on rows of mean > 32).  A handle belongs to the device that was current when it was
 * created: plan and run fail with SPMV_ERR_INVALID under another current device.  Excluded from the timed SpMV like the reference
 * excludes its host format build from TIME_KERNEL (e.g. wsp.cu:146 vs :167).
 * A plan snapshots the sparsity PATTERN (row_ptr, col_idx): with borrowed
 * arrays (spmv_csr_create_device) the pattern must not change afterwards;
 * vals are read live on every run and may be updated freely -- except by
 * SPMV_PANEL and SPMV_XSKIP, whose plans re-order the nonzeros and keep their own copy
 * of the values (after changing them re-plan with spmv_csr_plan_set, which always
 * rebuilds; spmv_csr_plan is idempotent; the panel layout also replaces col_idx and
 * vals byte for byte in what a run reads, and is limited to 2^29 columns).
 * spmv_csr_run: enqueue y = A x on `stream` (a hipStream_t, NULL = default).
 * Asynchronous; d_x has cols floats, d_y has rows floats and is fully
 * overwritten.  No allocation, no synchronisation: graph-capturable -- with
 * one exception: SPMV_SCALAR / SPMV_WAVE_PIPE on a handle that spmv_csr_plan
 * has not seen make their plan on the first run (an allocation and a wait). */
SPMV_API int spmv_csr_plan(spmv_csr_t *h, int variant, void *stream);
SPMV_API int spmv_csr_run(spmv_csr_t *h, int variant, const float *d_x, float *d_y, void *stream);

/* Tell the handle that the caller has rewritten vals (borrowed arrays, spmv_csr_create_device).  The variants that
 * read vals live need nothing; the plans of SPMV_PANEL and SPMV_XSKIP hold a re-ordered COPY of the values, and from
 * this call on spmv_csr_run of those variants (and of SPMV_AUTO where it resolved to one of them) fails with
 * SPMV_ERR_STALE_PLAN instead of multiplying with the old values, until spmv_csr_plan (which rebuilds a stale plan
 * with the parameters it had) or spmv_csr_plan_set has run again.  The library cannot see a write it is not told
 * about; for hunting one down set SPMV_CHECK_VALUES=1 in the environment: plans then also keep a checksum of vals and
 * every run of those variants recomputes it first (one pass over vals and a host wait per run -- a debug aid, not
 * graph-capturable).  Runs of one handle must be stream-ordered: plans own scratch buffers (slab partials, carries)
 * that two concurrent runs of the same handle on different streams would share. */
SPMV_API int spmv_csr_values_changed(spmv_csr_t *h);

/* What decides a plan's chunk cuts and with them the order of every fp32 sum, as numbers a caller can carry from
 * one handle to another (rank 0 to the other ranks of a job, one run to the next):
 *   params[0] variant actually planned (SPMV_AUTO resolves to SPMV_TILED or SPMV_PANEL)
 *   params[1] threads per workgroup (ADAPTIVE/TILED: 256 | 512 | 1024; a chunk is 16x that many nonzeros)
 *   params[2] staging-pass budget (TILED)        params[3] 1 = keep the 16-bit column copy where it pays (TILED)
 *   params[4] log2(columns per panel) (PANEL)    params[5] wavefronts per launch (PANEL)
 *   params[6] PANEL layout: 1 = panel sweep, x gathered through L2; 2 = the sweep with x staged in LDS; 3 = sorted blocks
 *             (params[4] = rows per block 4096 | 8192, params[5] = wavefronts per workgroup); 4 = binned, the sum launch fetches
 *             the tiles (params[4] = rows per bin 1024 ... 8192); 5 = binned, the product launch stores in bin order
 *             (params[4] = rows per bin 4096 | 8192 | 16384); 0 on input: the library's rule                 params[7] 0
 * spmv_csr_plan (the default) derives them from the matrix alone -- no timing -- so two handles of one matrix
 * already agree; handles of DIFFERENT row blocks of one matrix may not, and with SPMV_AUTOTUNE=1 nothing is
 * guaranteed.  spmv_csr_plan_set plans with exactly these numbers (replacing any existing plan of that variant),
 * spmv_csr_plan_like copies them from `src`.  Row blocks whose first nonzero offsets are multiples of the chunk
 * size, planned alike, give y bit-identical to the whole matrix planned alike (ADAPTIVE/TILED). */
SPMV_API int spmv_csr_plan_get(const spmv_csr_t *h, int variant, int32_t params[8]);
SPMV_API int spmv_csr_plan_set(spmv_csr_t *h, int variant, const int32_t params[8], void *stream);
SPMV_API int spmv_csr_plan_like(spmv_csr_t *dst, const spmv_csr_t *src, int variant, void *stream);

/* Bytes of plan data the variant reads per run.  Chunk boundaries, carries and windows come on top
 * of the CSR arrays; the 16-bit column offsets of SPMV_TILED REPLACE the 4-byte col_idx reads of the
 * chunks that have them (2 bytes per nonzero instead of 4), so a tiled run can move fewer HBM
 * bytes than the CSR-algorithmic count. */
SPMV_API int64_t spmv_csr_plan_bytes(const spmv_csr_t *h, int variant);

/* One-line description of the plan ("block=512 maxpass=4 chunks=32768 single=31080 col16=31080 ...")
 * written to buf (NUL-terminated, truncated to n). */
SPMV_API int spmv_csr_plan_describe(const spmv_csr_t *h, int variant, char *buf, int n);

/* Run `iters` back-to-back launches on `stream` between two HIP events
 * recorded on that same stream; returns the mean milliseconds per launch.
 * replaces: TIME_KERNEL (src/include/kernel.hpp:31-48). */
SPMV_API int spmv_csr_time(spmv_csr_t *h, int variant, const float *d_x, float *d_y, int iters,
                  void *stream, float *ms_per_launch);

/* Host-buffer convenience used by the C++ launchers: upload x (cols floats),
 * plan if needed, one untimed launch (code-object load, caches), one launch
 * between HIP events, download y (rows floats).  Synchronous.  *kernel_ms (may
 * be NULL) receives the event time of the second launch (the reference times a
 * single cold launch, kernel.hpp:31-48).  The dense and tcsr *_host calls do
 * the same.
 * replaces: the malloc/memcpy/TIME_KERNEL/memcpy/free body of a reference
 * launcher (e.g. csr_naive.cu:36-73). */
SPMV_API int spmv_csr_run_host(spmv_csr_t *h, int variant, const float *x_host, float *y_host,
                               float *kernel_ms);

/* ---- dense baselines (reference slots cublas / naive / tiling) ---------
 * y[i] = sum_j x[j] * A[j*N+i] on the dense device matrix.
 * replaces: cublas_gemv_gpu (cublas.cu:4-44), naive_kernel (naive.cu:4-11),
 * tiling_kernel (tiling_smem.cu:4-32).  mode 0 = thread per output (naive),
 * 1 = LDS-staged x tile (tiling), 2 = split-M wave-coalesced (vendor slot),
 * 3 = mode 2 + activation sparsity: rows of A whose x[j] is 0 are not read
 *     (asp_kernel_v0/1/2, src/kernels/asp.cu:20-26). */
SPMV_API int spmv_dense_gemv(int M, int N, const float *d_A, const float *d_x, float *d_y, int mode,
                    void *stream);

/* Modes 2 and 3 split M over 64 row slabs and need 64*N floats for the per-slab partial sums.
 * spmv_dense_gemv_ws takes that workspace from the caller (spmv_dense_gemv_workspace_bytes says how much; 0 for
 * modes 0/1): no allocation, no host wait, graph-capturable.  spmv_dense_gemv without the argument uses a
 * buffer the library keeps per device and hands from call to call in stream order (asynchronous too; it waits on
 * the host only when the buffer has to grow). */
SPMV_API int64_t spmv_dense_gemv_workspace_bytes(int N, int mode);
SPMV_API int spmv_dense_gemv_ws(int M, int N, const float *d_A, const float *d_x, float *d_y, int mode,
                                void *d_workspace, int64_t workspace_bytes, void *stream);

/* Host-buffer form of the same (upload A and x, run, download y). */
SPMV_API int spmv_dense_gemv_host(int M, int N, const float *A_host, const float *x_host, float *y_host,
                                  int mode, float *kernel_ms);

/* ---- the reference's ASP layout (the uncompressed one of its activation-sparsity launcher) -----
 * replaces: ASPMatrix (src/asp.cpp:3-14, src/include/asp.hpp) and the read pattern of asp_kernel_v0/1/2
 * (src/kernels/asp.cu:6-211).  spmv_asp_retile: d_A[M][N] row-major -> d_asp (M*N floats), the same array bit for
 * bit: 32 x 32 blocks, column panel by column panel (asp[(bn/32 * M + j) * 32 + c] = A[j*N + bn + c]); M and N
 * multiples of 32 (tester.cpp:10-11).  spmv_asp_gemv_ws: y = A^T x from that layout, a row of a panel skipped
 * when its x is zero (asp.cu:20-26); workspace as for spmv_dense_gemv_ws mode 3; asynchronous, no allocation.
 * (asp_gemv_gpu of include/kernel.hpp runs dense mode 3 on the row-major matrix -- the same skip without the copy.) */
SPMV_API int spmv_asp_retile(int M, int N, const float *d_A, float *d_asp, void *stream);
SPMV_API int spmv_asp_gemv_ws(int M, int N, const float *d_asp, const float *d_x, float *d_y, void *d_workspace,
                              int64_t workspace_bytes, void *stream);

/* ---- the reference's tiled bitmap-CSR format (dense-ish matrices, density > 1/32) -----
 * replaces: TCSRMatrix (src/tcsr.cpp:5-38, src/include/tcsr.hpp:4-23) and csr_tiling_kernel
 * with its launcher (src/kernels/csr_tiling.cu:24-166).  Same arrays, bit for bit: blk_idx
 * (exclusive nonzero prefix per 32x32 block + sentinel), bitmaps (word = output column inside
 * the block, bit = input row), vals (unpadded, bitmap order).  M and N must be multiples of 32
 * (the reference asserts it, src/tester.cpp:9-10); anything else is SPMV_ERR_INVALID.
 * Built on the device from the dense matrix; y[i] = sum_j x[j]*A[j*N+i] as everywhere. */
typedef struct spmv_tcsr spmv_tcsr_t;
SPMV_API int spmv_tcsr_from_dense_host(int M, int N, const float *A_host, void *stream, spmv_tcsr_t **out);
SPMV_API int spmv_tcsr_from_dense_device(int M, int N, const float *d_A, void *stream, spmv_tcsr_t **out);
SPMV_API int spmv_tcsr_sizes(const spmv_tcsr_t *h, int64_t *n_blk_idx, int64_t *n_bitmaps, int64_t *n_vals);
SPMV_API int spmv_tcsr_download(const spmv_tcsr_t *h, int32_t *blk_idx, uint32_t *bitmaps, float *vals);
SPMV_API int spmv_tcsr_run(const spmv_tcsr_t *h, const float *d_x, float *d_y, void *stream);
SPMV_API int spmv_tcsr_run_host(const spmv_tcsr_t *h, const float *x_host, float *y_host, float *kernel_ms);
SPMV_API int spmv_tcsr_destroy(spmv_tcsr_t *h);

/* ---- the reference's bitmap formats of its wsp / awsp / awsp_ref launchers (density > 1/32) -------------------
 * replaces: WSPMatrix (src/wsp.cpp:3-40) + wsp_kernel_v0/v1 (src/kernels/wsp.cu:4-138),
 *           AWSPMatrix (src/awsp.cpp:3-49) + awsp_kernel_v0/1/2 (src/kernels/awsp.cu:5-317),
 *           AWSPRefMatrix (src/awsp_ref.cpp:4-58) + awsp_ref_kernel / wsp_sm_kernel (awsp_ref.cu:6-185, wsp_sm.cu:6-211).
 * The arrays are the reference's, bit for bit (built on the device from the dense matrix); the multiply is re-derived
 * for 64-lane wavefronts (64-bit word pairs, __popcll ranks, running value offsets, the reference's x == 0 skip).
 * M and N must be multiples of 32.  stats[4] = what the reference classes expose: WSP {nz_max_m, nz_max_n, 0, 0},
 * AWSP {nz_bk_max_, 0, 0, 0}, AWSPRef warp_nz_offset_[0..3]. */
enum spmv_bitmap_format { SPMV_FMT_WSP = 0, SPMV_FMT_AWSP = 1, SPMV_FMT_AWSP_REF = 2, SPMV_FMT_COUNT = 3 };
typedef struct spmv_bitmap spmv_bitmap_t;
SPMV_API int spmv_bitmap_from_dense_host(int format, int M, int N, const float *A_host, void *stream, spmv_bitmap_t **out);
SPMV_API int spmv_bitmap_from_dense_device(int format, int M, int N, const float *d_A, void *stream, spmv_bitmap_t **out);
SPMV_API int spmv_bitmap_sizes(const spmv_bitmap_t *h, int64_t *n_bitmaps, int64_t *n_vals, int32_t stats[4]);
SPMV_API int spmv_bitmap_download(const spmv_bitmap_t *h, uint32_t *bitmaps, float *vals);
SPMV_API int spmv_bitmap_run(const spmv_bitmap_t *h, const float *d_x, float *d_y, void *stream);
SPMV_API int spmv_bitmap_run_host(const spmv_bitmap_t *h, const float *x_host, float *y_host, float *kernel_ms);
SPMV_API int spmv_bitmap_destroy(spmv_bitmap_t *h);

/* ---- synthetic CSR of stated (rows, cols, nnz) ---------------------------
 * Counter-based generator (DESIGN.md "Synthetic workloads"): element k of
 * global row r is a pure function of (seed, r, k, row length, band), so the
 * host can regenerate any row.  d_row_ptr (n_local+1 entries, rebased to 0)
 * gives the lengths of global rows [row0, row0+n_local).  band = 0: columns
 * uniform over [0, cols); band > 0: inside a diagonal band of that width. */
SPMV_API int spmv_synth_fill(uint64_t seed, int64_t row0, int64_t n_local, int64_t rows, int64_t cols,
                    int64_t band, const int32_t *d_row_ptr, int32_t *d_col_idx, float *d_vals,
                    void *stream);
SPMV_API int spmv_synth_x(uint64_t seed, int64_t j0, int64_t n, float *d_x, void *stream);

/* ---- measurement aids: kernels of KNOWN traffic + a marker dispatch ------------------------------------------
 * The metric of this path is rocprofv3's FETCH_SIZE / WRITE_SIZE over the kernel time (role of the reference's
 * profile.sh:18-20, a profiler around the executable).  The counters are uncalibrated outside 16-byte streams on
 * gfx950, so a profiled process launches these beside the SpMV kernels and corrects per access class
 * (bench.py --traffic-child; tools/summarize_profile.py).  spmv_calib_stream reads exactly `bytes` (a multiple of 16,
 * buffer 16-byte aligned) with 16-byte loads.  spmv_calib_gather reads n_lines DISTINCT 128-byte lines of a table of
 * table_lines lines (a power of two; 128 bytes each), one lane per line, neighbouring lanes far apart; touch = 1: one
 * word per line, 2: one word in each 64-byte half, 4: one word in each 32-byte sector.  spmv_calib_marker launches an
 * empty kernel (spmv::k_marker) of `id` workgroups of 64 threads: a cut mark in a per-dispatch counter file.
 * d_sink: one float the kernels never write in practice. */
SPMV_API int spmv_calib_stream(const void *d_src, int64_t bytes, float *d_sink, void *stream);
SPMV_API int spmv_calib_gather(const float *d_table, int64_t table_lines, int64_t n_lines, int touch, float *d_sink,
                               void *stream);
SPMV_API int spmv_calib_store(float *d_dst, int64_t bytes, int width, void *stream);
SPMV_API int spmv_calib_marker(int id, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_HIP_H */
