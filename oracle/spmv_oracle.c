/*
 * spmv_oracle.c -- CPU restatement of the reference's fp32 CSR SpMV path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * The product path (spmv-test_amd/) never links or calls it and has no CPU
 * fallback.
 *
 * What is restated, and from where (all citations into /root/reference):
 *   oracle_csr_count / oracle_csr_fill   src/matrix_csr.cpp:5-23 (CSRMatrix ctor)
 *   oracle_wsp_* / oracle_awsp_* / oracle_awsp_ref_* / oracle_asp_values
 *                                        src/wsp.cpp:3-40, src/awsp.cpp:3-49, src/awsp_ref.cpp:4-58,
 *                                        src/asp.cpp:3-14 (the bitmap formats; pinned like the CSR builder)
 *   oracle_sgemv_dense                   src/tester.cpp:36-45    (SgemvCPU, THE oracle)
 *   oracle_spmv_csr_seq                  src/kernels/csr_naive.cu:13-22 (the per-row loop,
 *                                        walked on the host in the same order)
 *   oracle_compare                       src/tester.cpp:74-88    (CompareY, abs 1e-3)
 *
 * Pinning status:
 *   - The CSR builder is pinned bit-for-bit against the reference's own
 *     CSRMatrix class, compiled unmodified from /root/reference/src/matrix_csr.cpp
 *     into oracle/_ref/libref_formats.so (see oracle/Makefile, tests/test_oracle_ref.py)
 *     and through the committed fixtures tests/golden/ (.npz files) that the same
 *     reference build produced (tests/golden/make_golden.py).
 *   - SgemvCPU lives in src/tester.cpp, which cannot be compiled in this image
 *     without stand-ins for <cuda_runtime.h>/<cublas_v2.h> (kernel.hpp:2,5), so
 *     it is NOT executed here: oracle_sgemv_dense restates its 8-line loop from
 *     the text.  Its equivalence to the CSR walk (zeros contribute exactly +-0)
 *     is checked on every fixture.  The reference's own tests hold no golden
 *     vectors (random_device-seeded inputs, src/tester.cpp:107,155).
 *
 * Arithmetic: fp32, one rounding for the product and one for the add, in
 * ascending k -- what g++ -O2 emits for tester.cpp:40-42 on x86-64 (no FMA
 * without -mfma).  Compiled with -ffp-contract=off so this never changes.
 *
 * The synth_* functions are NOT reference behaviour: they restate this
 * project's own counter-based synthetic-matrix specification (DESIGN.md,
 * "Synthetic workloads") so any row of a multi-hundred-million-nnz device
 * matrix can be regenerated on the host and checked.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__GNUC__)
#define ORACLE_API __attribute__((visibility("default")))
#else
#define ORACLE_API
#endif

/* ------------------------------------------------------------------ */
/* matrix_csr.cpp:5-23 -- dense row-major A[M][N]  ->  CSR of A^T.     */
/* Row i of the CSR is output i = column i of A; column index j is the */
/* input index.  An element is kept iff (value != 0.0f): -0.0f is      */
/* dropped, NaN is kept (matrix_csr.cpp:15).                           */
/* The reference's row_pointers has N entries and no sentinel; here    */
/* row_ptr has N+1 entries, row_ptr[N] = nnz (the value csr_naive.cu:15 */
/* substitutes for the missing entry).                                 */
/* ------------------------------------------------------------------ */
ORACLE_API int64_t oracle_csr_count(int M, int N, const float *A, int32_t *row_ptr)
{
    int64_t cur = 0;
    for (int i = 0; i < N; i++) {
        row_ptr[i] = (int32_t)cur;
        for (int j = 0; j < M; j++) {
            float v = A[(size_t)j * N + i];
            if (v != 0.0f) cur++;
        }
    }
    row_ptr[N] = (int32_t)cur;
    return cur;
}

ORACLE_API void oracle_csr_fill(int M, int N, const float *A, int32_t *col_idx, float *vals)
{
    size_t p = 0;
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < M; j++) {
            float v = A[(size_t)j * N + i];
            if (v != 0.0f) {
                vals[p] = v;
                col_idx[p] = j;
                p++;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* tcsr.cpp:5-38 -- tiled bitmap-CSR.  Blocks of 32 outputs x 32 inputs, */
/* ordered output strip (block_x) outer, input block (block_y) inner;   */
/* inside a block bit index i*32+j = (output column i, input row j), so  */
/* word i of the block belongs to output block_x+i and bit j to input    */
/* block_y+j; values are appended in that bit order; blk_idx is the      */
/* exclusive prefix of nonzeros per block with one trailing sentinel.    */
/* M and N must be multiples of 32 (the reference asserts it,            */
/* tester.cpp:9-10, and indexes out of range otherwise).                 */
/* Returns nnz; vals may be NULL for a counting pass.                    */
/* ------------------------------------------------------------------ */
ORACLE_API int64_t oracle_tcsr_build(int M, int N, const float *A, int32_t *blk_idx, uint32_t *bitmaps,
                                     float *vals)
{
    int64_t value_index = 0, bit_index = 0, nblk = 0;
    memset(bitmaps, 0, sizeof(uint32_t) * ((size_t)M * N / 32));
    blk_idx[nblk++] = 0;
    for (int bx = 0; bx < N; bx += 32) {
        for (int by = 0; by < M; by += 32) {
            for (int i = 0; i < 32; i++) {
                for (int j = 0; j < 32; j++) {
                    float v = A[(size_t)(by + j) * N + (bx + i)];
                    if (v != 0.0f) {
                        if (vals) vals[value_index] = v;
                        value_index++;
                        bitmaps[bit_index / 32] |= 1u << (bit_index % 32);
                    }
                    bit_index++;
                }
            }
            blk_idx[nblk++] = (int32_t)value_index;
        }
    }
    return value_index;
}

/* ------------------------------------------------------------------ */
/* The bitmap formats of the reference's wsp / awsp / awsp_ref / asp   */
/* launchers, restated as closed-form index maps (the reference grows   */
/* vectors of vectors and pads afterwards; the arrays are the same).   */
/* Every builder comes as a pair: *_bitmaps fills the bitmap and        */
/* returns the statistic that sizes the value array, *_values fills the */
/* zero-padded value array.  "kept" means (v != 0.0f), as everywhere.   */
/* M and N multiples of 32 (tester.cpp:9-10).                           */
/* ------------------------------------------------------------------ */

/* wsp.cpp:3-40 (WSPMatrix).  Bit (i*M + j) of the bitmap = element (input j, output i): one run of M bits per */
/* output column.  Values: the kept elements of column i, j ascending, at vals[i*nz_max_m + k]; nz_max_m = the    */
/* longest column (wsp.hpp:14, public), nz_max_n = N.  Returns nz_max_m.                                          */
ORACLE_API int32_t oracle_wsp_bitmaps(int M, int N, const float *A, uint32_t *bitmaps)
{
    int32_t longest = 0;
    memset(bitmaps, 0, sizeof(uint32_t) * ((size_t)M * N / 32));
    for (int i = 0; i < N; i++) {
        int32_t len = 0;
        for (int j = 0; j < M; j++) {
            if (A[(size_t)j * N + i] != 0.0f) {
                const size_t bit = (size_t)i * M + j;
                bitmaps[bit >> 5] |= 1u << (bit & 31);
                len++;
            }
        }
        if (len > longest) longest = len;
    }
    return longest;
}

ORACLE_API void oracle_wsp_values(int M, int N, const float *A, int32_t nz_max_m, float *vals)
{
    memset(vals, 0, sizeof(float) * (size_t)N * (size_t)nz_max_m);
    for (int i = 0; i < N; i++) {
        float *dst = vals + (size_t)i * nz_max_m;
        for (int j = 0; j < M; j++) {
            const float v = A[(size_t)j * N + i];
            if (v != 0.0f) *dst++ = v;
        }
    }
}

/* awsp.cpp:3-49 (AWSPMatrix).  32x32 blocks, output strip outer, input block inner: block b = (bn/32)*(M/32) + */
/* bm/32.  Word 32*b + r of the bitmap = input row bm+r, bit c = output bn+c.  Values of a block in that (r, c)    */
/* order at vals[b*nz_bk_max + k], every block padded to the fullest one.  Returns nz_bk_max (awsp.hpp:13).        */
ORACLE_API int32_t oracle_awsp_bitmaps(int M, int N, const float *A, uint32_t *bitmaps)
{
    int32_t fullest = 0;
    size_t word = 0;
    for (int bn = 0; bn < N; bn += 32) {
        for (int bm = 0; bm < M; bm += 32) {
            int32_t cnt = 0;
            for (int r = 0; r < 32; r++, word++) {
                uint32_t w = 0;
                const float *row = A + (size_t)(bm + r) * N + bn;
                for (int c = 0; c < 32; c++)
                    if (row[c] != 0.0f) { w |= 1u << c; cnt++; }
                bitmaps[word] = w;
            }
            if (cnt > fullest) fullest = cnt;
        }
    }
    return fullest;
}

ORACLE_API void oracle_awsp_values(int M, int N, const float *A, int32_t nz_bk_max, float *vals)
{
    const size_t nblk = (size_t)(M / 32) * (size_t)(N / 32);
    memset(vals, 0, sizeof(float) * nblk * (size_t)nz_bk_max);
    size_t b = 0;
    for (int bn = 0; bn < N; bn += 32) {
        for (int bm = 0; bm < M; bm += 32, b++) {
            float *dst = vals + b * (size_t)nz_bk_max;
            for (int r = 0; r < 32; r++) {
                const float *row = A + (size_t)(bm + r) * N + bn;
                for (int c = 0; c < 32; c++)
                    if (row[c] != 0.0f) *dst++ = row[c];
            }
        }
    }
}

/* awsp_ref.cpp:4-58 (AWSPRefMatrix).  Word s*M + j of the bitmap = input row j inside output strip s (outputs     */
/* 32s..32s+31), bit c = output 32s+c.  The M inputs are cut into four quarters ("warps", M/4 rows each); the kept */
/* elements of (strip s, quarter q) in (row, c) order start at vals[s*off[3] + (q ? off[q-1] : 0)], where off[q] is */
/* the inclusive prefix over q of the per-quarter maxima over all strips (warp_nz_offset_, awsp_ref.cpp:33-40).    */
ORACLE_API void oracle_awsp_ref_bitmaps(int M, int N, const float *A, uint32_t *bitmaps, int32_t off[4])
{
    int32_t most[4] = {0, 0, 0, 0};
    const int Q = M / 4;
    for (int s = 0; s < N / 32; s++) {
        for (int q = 0; q < 4; q++) {
            int32_t cnt = 0;
            for (int j = q * Q; j < (q + 1) * Q; j++) {
                uint32_t w = 0;
                const float *row = A + (size_t)j * N + 32 * s;
                for (int c = 0; c < 32; c++)
                    if (row[c] != 0.0f) { w |= 1u << c; cnt++; }
                bitmaps[(size_t)s * M + j] = w;
            }
            if (cnt > most[q]) most[q] = cnt;
        }
    }
    int32_t run = 0;
    for (int q = 0; q < 4; q++) { run += most[q]; off[q] = run; }
}

ORACLE_API void oracle_awsp_ref_values(int M, int N, const float *A, const int32_t off[4], float *vals)
{
    const int Q = M / 4;
    memset(vals, 0, sizeof(float) * (size_t)(N / 32) * (size_t)off[3]);
    for (int s = 0; s < N / 32; s++) {
        for (int q = 0; q < 4; q++) {
            float *dst = vals + (size_t)s * off[3] + (q ? off[q - 1] : 0);
            for (int j = q * Q; j < (q + 1) * Q; j++) {
                const float *row = A + (size_t)j * N + 32 * s;
                for (int c = 0; c < 32; c++)
                    if (row[c] != 0.0f) *dst++ = row[c];
            }
        }
    }
}

/* asp.cpp:3-14 (ASPMatrix): the dense matrix re-tiled, block b (as in AWSP) holds its 32x32 elements row-major. */
ORACLE_API void oracle_asp_values(int M, int N, const float *A, float *vals)
{
    size_t p = 0;
    for (int bn = 0; bn < N; bn += 32)
        for (int bm = 0; bm < M; bm += 32)
            for (int r = 0; r < 32; r++)
                for (int c = 0; c < 32; c++) vals[p++] = A[(size_t)(bm + r) * N + bn + c];
}

/* tester.cpp:36-45 -- y[i] = sum_j x[j] * A[j*N+i], fp32, j ascending. */
ORACLE_API void oracle_sgemv_dense(int M, int N, const float *A, const float *X, float *Y)
{
    for (int i = 0; i < N; i++) {
        float acc = 0.0f;
        for (int j = 0; j < M; j++) {
            acc += X[j] * A[(size_t)j * N + i];
        }
        Y[i] = acc;
    }
}

/* csr_naive.cu:13-22 walked on the host: acc += X[col[k]] * val[k]. */
static void spmv_rows(int64_t r0, int64_t r1, const int32_t *row_ptr, const int32_t *col_idx,
                      const float *vals, const float *x, float *y)
{
    for (int64_t r = r0; r < r1; r++) {
        float acc = 0.0f;
        for (int32_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
            acc += x[col_idx[k]] * vals[k];
        }
        y[r] = acc;
    }
}

ORACLE_API void oracle_spmv_csr_seq(int64_t rows, const int32_t *row_ptr, const int32_t *col_idx,
                                    const float *vals, const float *x, float *y)
{
    spmv_rows(0, rows, row_ptr, col_idx, vals, x, y);
}

/* Row-parallel version for the CPU baseline: per-row order is unchanged, so
 * the result is bit-identical to oracle_spmv_csr_seq for any thread count.
 * Rows are split at equal-nnz boundaries. */
typedef struct {
    int64_t r0, r1;
    const int32_t *row_ptr, *col_idx;
    const float *vals, *x;
    float *y;
} mt_job_t;

static void *mt_worker(void *p)
{
    mt_job_t *j = (mt_job_t *)p;
    spmv_rows(j->r0, j->r1, j->row_ptr, j->col_idx, j->vals, j->x, j->y);
    return NULL;
}

static int64_t lower_bound_i32(const int32_t *a, int64_t n, int64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        if ((int64_t)a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

ORACLE_API int oracle_spmv_csr_mt(int64_t rows, const int32_t *row_ptr, const int32_t *col_idx,
                                  const float *vals, const float *x, float *y, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256];
    mt_job_t job[256];
    int64_t nnz = row_ptr[rows];
    int64_t prev = 0;
    for (int t = 0; t < threads; t++) {
        int64_t target = nnz * (t + 1) / threads;
        int64_t r1 = (t == threads - 1) ? rows : lower_bound_i32(row_ptr, rows + 1, target);
        if (r1 > rows) r1 = rows;
        if (r1 < prev) r1 = prev;
        job[t] = (mt_job_t){prev, r1, row_ptr, col_idx, vals, x, y};
        prev = r1;
    }
    for (int t = 0; t < threads; t++) {
        if (pthread_create(&tid[t], NULL, mt_worker, &job[t]) != 0) return -1;
    }
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    return 0;
}

/* fp64 walk + sum of |x*val| per row: the error budget for reordered fp32
 * sums (condition-aware bound, DESIGN.md "Tolerance"). */
ORACLE_API void oracle_spmv_csr_f64(int64_t rows, const int32_t *row_ptr, const int32_t *col_idx,
                                    const float *vals, const float *x, double *y, double *abs_sum)
{
    for (int64_t r = 0; r < rows; r++) {
        double acc = 0.0, mag = 0.0;
        for (int32_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
            double p = (double)x[col_idx[k]] * (double)vals[k];
            acc += p;
            mag += fabs(p);
        }
        y[r] = acc;
        if (abs_sum) abs_sum[r] = mag;
    }
}

/* tester.cpp:74-88 -- count of entries with |cpu - gpu| > tol (the reference
 * prints them and carries on; it uses tol = 1e-3). */
ORACLE_API int64_t oracle_compare(int64_t n, const float *y_cpu, const float *y_gpu, float tol)
{
    int64_t bad = 0;
    for (int64_t i = 0; i < n; i++) {
        float d = y_cpu[i] - y_gpu[i];
        if (!(fabsf(d) <= tol)) bad++;
    }
    return bad;
}

/* ------------------------------------------------------------------ */
/* Synthetic-matrix specification (this project's, not the reference's) */
/* ------------------------------------------------------------------ */
static inline uint64_t synth_mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

static inline uint64_t synth_hash(uint64_t seed, uint64_t a, uint64_t b)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (a + 1);
    z = synth_mix64(z);
    z += 0xD1B54A32D192ED03ull * (b + 1);
    return synth_mix64(z);
}

/* 24 hash bits -> odd multiple of 2^-24 in (-1,1): exact in fp32, never 0. */
static inline float synth_unit(uint64_t h)
{
    int32_t b = (int32_t)(h & 0xFFFFFFu);
    return (float)(2 * b + 1 - (1 << 24)) * (1.0f / 16777216.0f);
}

ORACLE_API float oracle_synth_x(uint64_t seed, int64_t j)
{
    return synth_unit(synth_hash(seed ^ 0x5851F42D4C957F2Dull, (uint64_t)j, 0));
}

/* Column window of a row: [w0, w0+W).  band == 0: the whole row [0, cols).
 * band  > 0: W = max(band, 8*len) clipped to cols, centred on the diagonal
 * position row*cols/rows and clamped into [0, cols-W]. */
static inline void synth_window(int64_t row, int64_t rows, int64_t cols, int64_t len, int64_t band,
                                int64_t *w0, int64_t *W)
{
    if (band <= 0) { *w0 = 0; *W = cols; return; }
    int64_t w = band > 8 * len ? band : 8 * len;
    if (w > cols) w = cols;
    int64_t centre = (int64_t)(((uint64_t)row * (uint64_t)cols) / (uint64_t)rows);
    int64_t s = centre - w / 2;
    if (s < 0) s = 0;
    if (s > cols - w) s = cols - w;
    *w0 = s; *W = w;
}

/* Element k of a row of length len: one column drawn uniformly inside
 * stratum k of the row's window -> columns ascending and distinct. */
ORACLE_API void oracle_synth_row(uint64_t seed, int64_t row, int64_t rows, int64_t cols,
                                 int64_t len, int64_t band, int32_t *col_idx, float *vals)
{
    int64_t w0, W;
    synth_window(row, rows, cols, len, band, &w0, &W);
    for (int64_t k = 0; k < len; k++) {
        uint64_t h = synth_hash(seed, (uint64_t)row, (uint64_t)k);
        int64_t lo = (int64_t)(((uint64_t)k * (uint64_t)W) / (uint64_t)len);
        int64_t hi = (int64_t)(((uint64_t)(k + 1) * (uint64_t)W) / (uint64_t)len);
        int64_t span = hi - lo;
        if (span < 1) span = 1;
        col_idx[k] = (int32_t)(w0 + lo + (int64_t)((h >> 32) % (uint64_t)span));
        vals[k] = synth_unit(h);
    }
}

/* Fill rows [r0, r1) of a matrix whose row_ptr (rebased so row_ptr[0] is the
 * offset of row r0 inside col_idx/vals) is given; global row ids r0.. are
 * used for hashing so a shard equals the same rows of the whole matrix. */
ORACLE_API void oracle_synth_fill(uint64_t seed, int64_t r0, int64_t r1, int64_t rows, int64_t cols,
                                  int64_t band, const int32_t *row_ptr, int32_t *col_idx, float *vals)
{
    for (int64_t r = r0; r < r1; r++) {
        int64_t b = row_ptr[r - r0], e = row_ptr[r - r0 + 1];
        oracle_synth_row(seed, r, rows, cols, e - b, band, col_idx + b, vals + b);
    }
}

ORACLE_API void oracle_synth_x_fill(uint64_t seed, int64_t j0, int64_t j1, float *x)
{
    for (int64_t j = j0; j < j1; j++) x[j - j0] = oracle_synth_x(seed, j);
}
