// capi.hip -- the extern "C" entry points of include/spmv_hip.h.
#include <cstdarg>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include "spmv_internal.hpp"

namespace spmv {

static thread_local char g_err[512] = "";
static thread_local float g_first_launch_ms = 0.0f;   // the FIRST (cold) launch of the last *_run_host call on this thread

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    set_error("HIP error %s:%d: %s (%s)", file, line, hipGetErrorString(e), what);
    return SPMV_ERR_HIP;
}

static int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device visible (hipGetDeviceCount: %s); libspmv_hip has no CPU path",
                  e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return SPMV_ERR_NO_DEVICE;
    }
    return SPMV_OK;
}

int LdsOptIn::ensure(const void *fn, int device, int bytes)
{
    const uint64_t bit = (device >= 0 && device < 64) ? (1ull << device) : 0ull;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return SPMV_OK;
    SPMV_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (bit) done.fetch_or(bit, std::memory_order_release);
    return SPMV_OK;
}

int device_cus(int device)
{
    static std::atomic<int> cache[64];
    const bool cached = device >= 0 && device < 64;
    if (cached) {
        const int c = cache[device].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        cus = 256;
    }
    if (cached) cache[device].store(cus, std::memory_order_relaxed);
    return cus;
}

int require_current(int device, const char *what)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: hipGetDevice failed", what);
        return SPMV_ERR_HIP;
    }
    if (cur != device) {
        set_error("%s: the handle lives on device %d but the current device is %d (hipSetDevice first)", what, device, cur);
        return SPMV_ERR_INVALID;
    }
    return SPMV_OK;
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace spmv

using namespace spmv;

extern "C" {

int spmv_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char *spmv_last_error(void) { return g_err; }

float spmv_last_first_launch_ms(void) { return g_first_launch_ms; }

const char *spmv_variant_name(int variant)
{
    switch (variant) {
        case SPMV_SCALAR: return "scalar";
        case SPMV_WAVE: return "wave";
        case SPMV_WAVE_PIPE: return "wave_pipe";
        case SPMV_VECTOR: return "vector";
        case SPMV_ADAPTIVE: return "adaptive";
        case SPMV_TILED: return "tiled";
        case SPMV_PANEL: return "panel";
        case SPMV_AUTO: return "auto";
        case SPMV_XSKIP: return "xskip";
        default: return "unknown";
    }
}

static int check_dims(int64_t rows, int64_t cols, int64_t nnz)
{
    if (rows < 0 || cols < 0 || nnz < 0 || rows >= (1LL << 31) || cols >= (1LL << 31) || nnz >= (1LL << 31)) {
        set_error("bad dimensions rows=%lld cols=%lld nnz=%lld (each must be in [0, 2^31))", (long long)rows,
                  (long long)cols, (long long)nnz);
        return SPMV_ERR_INVALID;
    }
    return SPMV_OK;
}

int spmv_csr_validate(const spmv_csr_t *h, void *stream)
{
    if (!h) { set_error("spmv_csr_validate: null handle"); return SPMV_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    const int32_t init[4] = {INT32_MAX, INT32_MAX, 0, (int32_t)h->nnz};
    int32_t bad[4];
    DevPtr<int32_t> d_bad;
    SPMV_HIP_TRY(d_bad.alloc(4));
    SPMV_HIP_TRY(hipMemcpyAsync(d_bad.p, init, sizeof init, hipMemcpyHostToDevice, st));
    int rc = launch_validate(h, d_bad.p, st);
    if (rc) return rc;
    SPMV_HIP_TRY(hipMemcpyAsync(bad, d_bad.p, sizeof bad, hipMemcpyDeviceToHost, st));
    SPMV_HIP_TRY(hipStreamSynchronize(st));
    if (bad[2] != 0 || (int64_t)bad[3] != h->nnz) {
        set_error("malformed CSR: row_ptr[0]=%d row_ptr[rows]=%d, expected 0 and nnz=%lld", bad[2], bad[3],
                  (long long)h->nnz);
        return SPMV_ERR_INVALID;
    }
    if (bad[0] != INT32_MAX) {
        set_error("malformed CSR: row_ptr decreases or leaves [0, nnz] at row %d", bad[0]);
        return SPMV_ERR_INVALID;
    }
    if (bad[1] != INT32_MAX) {
        set_error("malformed CSR: column index of element %d is outside [0, %lld)", bad[1], (long long)h->cols);
        return SPMV_ERR_INVALID;
    }
    return SPMV_OK;
}

int spmv_csr_create_host(int64_t rows, int64_t cols, int64_t nnz, const int32_t *row_ptr,
                         const int32_t *col_idx, const float *vals, spmv_csr_t **out)
{
    if (!out || !row_ptr || (nnz > 0 && (!col_idx || !vals))) {
        set_error("spmv_csr_create_host: null argument");
        return SPMV_ERR_INVALID;
    }
    int rc = check_dims(rows, cols, nnz);
    if (rc) return rc;
    if ((rc = require_device())) return rc;
    if (row_ptr[0] != 0 || row_ptr[rows] != nnz) {
        set_error("spmv_csr_create_host: row_ptr[0]=%d row_ptr[rows]=%d, expected 0 and nnz=%lld", row_ptr[0],
                  row_ptr[rows], (long long)nnz);
        return SPMV_ERR_INVALID;
    }
    DevPtr<int32_t> rp, ci;
    DevPtr<float> va;
    SPMV_HIP_TRY(rp.alloc((size_t)rows + 1));
    SPMV_HIP_TRY(ci.alloc((size_t)nnz));
    SPMV_HIP_TRY(va.alloc((size_t)nnz));
    SPMV_HIP_TRY(hipMemcpy(rp.p, row_ptr, sizeof(int32_t) * ((size_t)rows + 1), hipMemcpyHostToDevice));
    if (nnz > 0) {
        SPMV_HIP_TRY(hipMemcpy(ci.p, col_idx, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice));
        SPMV_HIP_TRY(hipMemcpy(va.p, vals, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice));
    }
    spmv_csr *h = new (std::nothrow) spmv_csr();
    if (!h) { set_error("out of host memory"); return SPMV_ERR_INVALID; }
    h->rows = rows; h->cols = cols; h->nnz = nnz;
    h->owns_arrays = true;
    if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
    h->d_row_ptr = rp.release(); h->d_col_idx = ci.release(); h->d_vals = va.release();
    if ((rc = spmv_csr_validate(h, nullptr))) {
        spmv_csr_destroy(h);
        return rc;
    }
    *out = h;
    return SPMV_OK;
}

int spmv_csr_create_device(int64_t rows, int64_t cols, int64_t nnz, const int32_t *d_row_ptr,
                           const int32_t *d_col_idx, const float *d_vals, spmv_csr_t **out)
{
    if (!out || !d_row_ptr || (nnz > 0 && (!d_col_idx || !d_vals))) {
        set_error("spmv_csr_create_device: null argument");
        return SPMV_ERR_INVALID;
    }
    int rc = check_dims(rows, cols, nnz);
    if (rc) return rc;
    if ((rc = require_device())) return rc;
    if (!aligned16(d_col_idx) || !aligned16(d_vals)) {
        set_error("spmv_csr_create_device: col_idx and vals must be 16-byte aligned");
        return SPMV_ERR_INVALID;
    }
    spmv_csr *h = new (std::nothrow) spmv_csr();
    if (!h) { set_error("out of host memory"); return SPMV_ERR_INVALID; }
    h->rows = rows; h->cols = cols; h->nnz = nnz;
    h->d_row_ptr = d_row_ptr; h->d_col_idx = d_col_idx; h->d_vals = d_vals;
    h->owns_arrays = false;
    if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
    if ((rc = spmv_csr_validate(h, nullptr))) {
        delete h;
        return rc;
    }
    *out = h;
    return SPMV_OK;
}

int spmv_csr_from_dense_device(int M, int N, const float *d_A, void *stream, spmv_csr_t **out)
{
    if (!out || M < 0 || N < 0 || (!d_A && (int64_t)M * N > 0)) {
        set_error("spmv_csr_from_dense_device: bad argument");
        return SPMV_ERR_INVALID;
    }
    int rc = require_device();
    if (rc) return rc;
    return dense_to_csr(M, N, d_A, (hipStream_t)stream, out);
}

int spmv_csr_from_dense_host(int M, int N, const float *A_host, void *stream, spmv_csr_t **out)
{
    if (!out || M < 0 || N < 0 || (!A_host && (int64_t)M * N > 0)) {
        set_error("spmv_csr_from_dense_host: bad argument");
        return SPMV_ERR_INVALID;
    }
    int rc = require_device();
    if (rc) return rc;
    const size_t bytes = sizeof(float) * (size_t)M * (size_t)N;
    float *d_A = nullptr;
    SPMV_HIP_TRY(hipMalloc((void **)&d_A, bytes ? bytes : 4));
    hipError_t e = hipMemcpyAsync(d_A, A_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { (void)hipFree(d_A); return hip_fail(e, "hipMemcpyAsync(A)", __FILE__, __LINE__); }
    rc = dense_to_csr(M, N, d_A, (hipStream_t)stream, out);
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d_A);
    return rc;
}

int spmv_csr_download(const spmv_csr_t *h, int32_t *row_ptr, int32_t *col_idx, float *vals)
{
    if (!h) { set_error("spmv_csr_download: null handle"); return SPMV_ERR_INVALID; }
    if (row_ptr)
        SPMV_HIP_TRY(hipMemcpy(row_ptr, h->d_row_ptr, sizeof(int32_t) * ((size_t)h->rows + 1), hipMemcpyDeviceToHost));
    if (col_idx && h->nnz)
        SPMV_HIP_TRY(hipMemcpy(col_idx, h->d_col_idx, sizeof(int32_t) * (size_t)h->nnz, hipMemcpyDeviceToHost));
    if (vals && h->nnz)
        SPMV_HIP_TRY(hipMemcpy(vals, h->d_vals, sizeof(float) * (size_t)h->nnz, hipMemcpyDeviceToHost));
    return SPMV_OK;
}

int spmv_csr_dims(const spmv_csr_t *h, int64_t *rows, int64_t *cols, int64_t *nnz)
{
    if (!h) { set_error("spmv_csr_dims: null handle"); return SPMV_ERR_INVALID; }
    if (rows) *rows = h->rows;
    if (cols) *cols = h->cols;
    if (nnz) *nnz = h->nnz;
    return SPMV_OK;
}

int spmv_csr_column_range(const spmv_csr_t *h, int64_t *col_min, int64_t *col_max, void *stream)
{
    if (!h || !col_min || !col_max) { set_error("spmv_csr_column_range: null argument"); return SPMV_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    DevPtr<int32_t> d;
    SPMV_HIP_TRY(d.alloc(2));
    int rc = launch_column_range(*h, d.p, st);
    if (rc) return rc;
    int32_t out[2] = {0, 0};
    SPMV_HIP_TRY(hipMemcpyAsync(out, d.p, sizeof out, hipMemcpyDeviceToHost, st));
    SPMV_HIP_TRY(hipStreamSynchronize(st));
    *col_min = out[1] < 0 ? h->cols : (int64_t)out[0];
    *col_max = (int64_t)out[1];
    return SPMV_OK;
}

int spmv_csr_destroy(spmv_csr_t *h)
{
    if (!h) return SPMV_OK;
    int rc = SPMV_OK;
    auto fr = [&](const void *p) {
        if (p && hipFree(const_cast<void *>(p)) != hipSuccess) rc = SPMV_ERR_HIP;
    };
    if (h->owns_arrays) { fr(h->d_row_ptr); fr(h->d_col_idx); fr(h->d_vals); }
    destroy_plans(*h);
    delete h;
    if (rc) set_error("hipFree failed in spmv_csr_destroy");
    return rc;
}

// SPMV_AUTO.  TILED's plan is made first (cheap: a few passes over col_idx, no trial launches) and priced (model_cost,
// in units of "a chunk that streams 8 bytes per nonzero with cache-resident gathers").
//   1. It stages (nearly) everything in one or two passes (model_cost <= 1.20: bands up to ~60 000 columns at config 4): TILED.
//   2. Otherwise the sorted-blocks layout of SPMV_PANEL (kernels_colsort.hip) is tried where a look at the matrix says it can
//      work -- short rows (at most a tenth of the nonzeros in rows of more than 256), blocks of 4096 rows whose window of
//      x fits an L2 -- and taken when its own model prices it below TILED (wide bands: 65 536 columns and beyond; uniform
//      columns over a few MiB of x: config 2).
//   3. Otherwise, when TILED could stage or sort less than half of the chunks -- it gathers most of x through L2/fabric,
//      one request per nonzero -- and x fills one XCD's 4 MiB L2 or more, the panel sweep (1.2x at config 2's 4 MiB, 3-4x at
//      64 MiB: DESIGN.md section 4); below that size x sits in every L2 anyway and the row-major kernel keeps its lead.
static int plan_auto(spmv_csr &h, hipStream_t s)
{
    if (h.auto_variant == SPMV_PANEL) return refresh_panel(h, h.plan_auto_panel, s);   // idempotent, except that a stale copy of vals is rebuilt
    if (h.auto_variant >= 0) return SPMV_OK;
    h.auto_made_tiled = h.plan_tiled.block == 0;
    int rc = plan_adaptive(h, true, s);
    if (rc) return rc;
    // a TILED plan the caller made stays; one made here for a look is released when another variant is chosen (up to
    // 6 bytes per nonzero of column copies nobody would read)
    auto release_tiled = [&]() { if (h.auto_made_tiled) drop_tiled_plan(h); h.auto_made_tiled = false; };
    const ChunkPlan &p = h.plan_tiled;
    const double cost_tiled = p.model_cost;
    const bool little_staged = p.nchunks > 0 && 2 * ((int64_t)p.staged_full + p.nsorted) < p.nchunks &&
                               2 * (int64_t)p.nblk_chunks < p.nchunks;
    const bool x_beyond_l2 = h.cols * (int64_t)sizeof(float) >= (4ll << 20);   // an L2 also holds the stream passing through
    // (a workgroup per 4096 rows, one or two per CU: fewer blocks than three quarters of the CUs leave the chip idle)
    bool try_sorted = p.nchunks > 0 && cost_tiled > 1.20 && 2 * (int64_t)p.nblk_chunks < p.nchunks &&
                      h.rows >= 4096ll * (3 * device_cus(h.device) / 4);
    if (const char *e = getenv("SPMV_AUTO_SORTED_BLOCKS")) try_sorted = try_sorted && atoi(e) != 0;   // 0: never (A/B runs)
    if (try_sorted) {
        double long_frac = 0.0, wide_frac = 0.0, lines_est = 0.0;
        if ((rc = colsort_probe(h, s, &long_frac, &wide_frac, &lines_est))) return rc;
        // (the estimate of the lines per nonzero prices the layout before it is built: 5 % slack for what it cannot see)
        if (long_frac <= 0.10 && wide_frac <= 0.10 && 0.95 * colsort_cost(4096, lines_est, long_frac) < cost_tiled) {
            // (the layout takes 8192-row blocks where the 4096-row ones touch more than 0.27 lines of x per nonzero: where the
            // probe's estimate is clearly beyond that, the 4096-row build -- 23 ms at config 4 -- is skipped)
            const bool big = lines_est > 0.30 && h.rows >= 4096ll * 16 * device_cus(h.device);
            rc = build_panel(h, h.plan_auto_panel, big ? 8192 : 0, big ? 4 : 0, 3, s);
            if (rc == SPMV_OK) {
                const PanelPlan &pp = h.plan_auto_panel;
                const bool fits = (double)pp.tail <= 0.10 * (double)h.nnz && 10 * pp.wide_blocks <= pp.nblocks;
                if (fits && colsort_model_cost(pp, h.nnz) < 0.98 * cost_tiled) {   // (a tie goes to TILED: it reads vals live)
                    h.auto_variant = SPMV_PANEL;
                    release_tiled();
                    return SPMV_OK;
                }
                destroy_panel(h.plan_auto_panel);
            } else if (rc != SPMV_ERR_INVALID) {
                return rc;   // INVALID: outside the layout's limits -> on with the other candidates
            }
        }
    }
    if (little_staged && x_beyond_l2) {
        // x beyond every L2 (each gather would be a 128-byte line from the fabric) and tiles that still hold a line or
        // two of products: the binned layout -- two streaming launches, nothing gathered from memory
        // (measured, profiles/r04_binned_*.jsonl: config 4 uniform 1.33 -> 1.06 ms at 83 nonzeros per tile, config 3 uniform
        // 0.86 -> 0.52 ms at 663; config 5's shard 3.0 -> 4.1 ms at 10: the sum launch works on nearly empty pieces)
        // Thinner tiles (config 5's shard: 14 at 4096 rows per bin, 28 at 8192): the flavour whose product launch stores in bin
        // order -- the sum launch then streams whatever the tiles hold (3.0 -> 1.12 ms there, the sweep with its shorter step 2.4).
        // Constant rows of 8 / 10 / 12 / 16 / 24 on 16Mi x 16Mi (64 ... 192 nonzeros per 4096 rows x 32768 columns): 0.55 / 0.67 /
        // 0.78 / 1.01 / 1.50 ms against the fetching flavour's 0.69 / 0.78 / 0.87 / 1.06 / 1.53 (profiles/
        // r04_binned_flavours_by_tile.jsonl), config 4 itself (128) 0.97 against 1.01, config 3 (1024, a third of the nonzeros in
        // long rows) 0.52 against 0.54 once it has a bin per resident wavefront: the fetching flavour (mode 4) is never the
        // rule's answer any more; it stays selectable
        const bool big_x = h.cols * (int64_t)sizeof(float) >= (8ll << 20);
        const double tile = binned_tile_nonzeros(h, 4096);
        int try_binned = big_x && tile >= 4.0 ? 5 : 0;
        if (const char *e = getenv("SPMV_AUTO_BINNED")) try_binned = atoi(e) != 0 ? try_binned : 0;   // 0: never (A/B runs)
        if (try_binned) {
            rc = build_panel(h, h.plan_auto_panel, 0, 0, try_binned, s);
            // (bins that ran out of spare accumulators add with LDS atomics, four times slower: rows with dozens of nonzeros in
            // one tile -- long rows over a band of a few million columns -- are the fetching flavour's case)
            if (rc == SPMV_OK && try_binned == 5 && 32 * (int64_t)h.plan_auto_panel.flagged_tiles > h.plan_auto_panel.nblocks)
                rc = build_panel(h, h.plan_auto_panel, 0, 0, 4, s);
            if (rc == SPMV_OK) {
                h.auto_variant = SPMV_PANEL;
                release_tiled();
                return SPMV_OK;
            }
            if (rc != SPMV_ERR_INVALID) return rc;   // INVALID: outside the layout's limits -> the sweep
        }
        rc = build_panel(h, h.plan_auto_panel, 0, 0, 1, s);
        if (rc == SPMV_OK) {
            h.auto_variant = SPMV_PANEL;
            release_tiled();
            return SPMV_OK;
        }
        if (rc != SPMV_ERR_INVALID) return rc;   // INVALID: outside the panel layout's limits -> stay with TILED
    }
    h.auto_variant = SPMV_TILED;
    return SPMV_OK;
}

int spmv_csr_plan(spmv_csr_t *h, int variant, void *stream)
{
    if (!h) { set_error("spmv_csr_plan: null handle"); return SPMV_ERR_INVALID; }
    if (int rc = require_current(h->device, "spmv_csr_plan")) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
        case SPMV_AUTO: return plan_auto(*h, s);
        case SPMV_WAVE: return h->nnz <= 32 * h->rows ? plan_wave(*h, s) : SPMV_OK;   // short rows: bundles (the long rows' pieces)
        case SPMV_SCALAR:        // (the x windows of the bundle kernel, which SPMV_SCALAR runs with ordered sums)
        case SPMV_WAVE_PIPE: return plan_wave(*h, s);
        case SPMV_VECTOR: return plan_vector(*h, s);
        case SPMV_ADAPTIVE: return plan_adaptive(*h, false, s);
        case SPMV_TILED: return plan_adaptive(*h, true, s);
        case SPMV_PANEL: return plan_panel(*h, s);
        case SPMV_XSKIP: return plan_xskip(*h, s);
        default:
            set_error("spmv_csr_plan: unknown variant %d", variant);
            return SPMV_ERR_VARIANT;
    }
}

int spmv_csr_run(spmv_csr_t *h, int variant, const float *d_x, float *d_y, void *stream)
{
    if (!h || (!d_x && h->cols > 0) || (!d_y && h->rows > 0)) {
        set_error("spmv_csr_run: null argument");
        return SPMV_ERR_INVALID;
    }
    if (!aligned16(d_x)) {
        set_error("spmv_csr_run: x must be 16-byte aligned");
        return SPMV_ERR_INVALID;
    }
    if (int rc = require_current(h->device, "spmv_csr_run")) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (variant == SPMV_AUTO) {
        if (h->auto_variant < 0) { set_error("SPMV_AUTO used before spmv_csr_plan"); return SPMV_ERR_NOT_PLANNED; }
        if (h->auto_variant == SPMV_PANEL) return launch_panel_plan(*h, h->plan_auto_panel, d_x, d_y, s);
        variant = h->auto_variant;
    }
    switch (variant) {
        case SPMV_SCALAR: return launch_scalar(*h, d_x, d_y, s);
        case SPMV_WAVE: return launch_wave(*h, d_x, d_y, false, s);
        case SPMV_WAVE_PIPE: return launch_wave(*h, d_x, d_y, true, s);
        case SPMV_VECTOR: return launch_vector(*h, d_x, d_y, s);
        case SPMV_ADAPTIVE: return launch_adaptive(*h, d_x, d_y, false, s);
        case SPMV_TILED: return launch_adaptive(*h, d_x, d_y, true, s);
        case SPMV_PANEL: return launch_panel(*h, d_x, d_y, s);
        case SPMV_XSKIP: return launch_xskip(*h, d_x, d_y, s);
        default:
            set_error("spmv_csr_run: unknown variant %d", variant);
            return SPMV_ERR_VARIANT;
    }
}

int spmv_csr_values_changed(spmv_csr_t *h)
{
    if (!h) { set_error("spmv_csr_values_changed: null handle"); return SPMV_ERR_INVALID; }
    ++h->values_gen;
    return SPMV_OK;
}

int spmv_csr_plan_get(const spmv_csr_t *h, int variant, int32_t params[8])
{
    if (!h || !params) { set_error("spmv_csr_plan_get: null argument"); return SPMV_ERR_INVALID; }
    for (int i = 0; i < 8; ++i) params[i] = 0;
    const bool is_auto = variant == SPMV_AUTO;
    if (variant == SPMV_AUTO) {
        if (h->auto_variant < 0) { set_error("spmv_csr_plan_get: SPMV_AUTO is not planned"); return SPMV_ERR_NOT_PLANNED; }
        variant = h->auto_variant;
    }
    const PanelPlan &panel = is_auto ? h->plan_auto_panel : h->plan_panel;
    params[0] = variant;
    switch (variant) {
        case SPMV_SCALAR: case SPMV_WAVE: case SPMV_WAVE_PIPE: return SPMV_OK;
        case SPMV_VECTOR: params[1] = h->vector_width; return SPMV_OK;
        case SPMV_ADAPTIVE:
        case SPMV_TILED: {
            const ChunkPlan &p = variant == SPMV_TILED ? h->plan_tiled : h->plan_adaptive;
            if (!p.block) { set_error("spmv_csr_plan_get: variant %d is not planned", variant); return SPMV_ERR_NOT_PLANNED; }
            params[1] = p.block; params[2] = p.maxpass; params[3] = p.col16_wanted ? 1 : 0;
            return SPMV_OK;
        }
        case SPMV_XSKIP:
            if (!h->plan_xskip.ready) { set_error("spmv_csr_plan_get: xskip is not planned"); return SPMV_ERR_NOT_PLANNED; }
            params[1] = h->plan_xskip.slabs;
            return SPMV_OK;
        case SPMV_PANEL:
            if (!panel.ready) { set_error("spmv_csr_plan_get: panel is not planned"); return SPMV_ERR_NOT_PLANNED; }
            params[4] = panel.pw_bits; params[5] = panel.waves_per_launch;
            params[6] = panel.binned_mode ? (panel.scatter_mode ? 5 : 4) : panel.sorted_mode ? 3 : (panel.lds_mode ? 2 : 1);
            if (panel.sorted_mode) { params[4] = panel.sb_rows; params[5] = panel.sb_waves; }
            if (panel.binned_mode) { params[4] = panel.bin_rows; params[5] = 0; }
            return SPMV_OK;
        default: set_error("spmv_csr_plan_get: unknown variant %d", variant); return SPMV_ERR_VARIANT;
    }
}

int spmv_csr_plan_set(spmv_csr_t *h, int variant, const int32_t params[8], void *stream)
{
    if (!h || !params) { set_error("spmv_csr_plan_set: null argument"); return SPMV_ERR_INVALID; }
    if (int rc = require_current(h->device, "spmv_csr_plan_set")) return rc;
    hipStream_t s = (hipStream_t)stream;
    int target = variant;
    if (variant == SPMV_AUTO) {
        target = params[0];
        if (target != SPMV_TILED && target != SPMV_PANEL) {
            set_error("spmv_csr_plan_set: SPMV_AUTO resolves to tiled or panel, not %d", target);
            return SPMV_ERR_INVALID;
        }
    } else if (params[0] != variant) {
        set_error("spmv_csr_plan_set: params describe variant %d, not %d", params[0], variant);
        return SPMV_ERR_INVALID;
    }
    int rc;
    switch (target) {
        case SPMV_SCALAR: case SPMV_WAVE: case SPMV_WAVE_PIPE: rc = SPMV_OK; break;
        case SPMV_VECTOR: {
            const int w = params[1];
            if (w != 2 && w != 4 && w != 8 && w != 16 && w != 32) { set_error("spmv_csr_plan_set: lanes per row %d", w); return SPMV_ERR_INVALID; }
            h->vector_width = w;
            rc = SPMV_OK;
            break;
        }
        case SPMV_ADAPTIVE: rc = plan_adaptive_with(*h, params[1], s); break;
        case SPMV_TILED: rc = plan_tiled_with(*h, params[1], params[2], params[3] != 0, s); break;
        case SPMV_PANEL: rc = build_panel(*h, variant == SPMV_AUTO ? h->plan_auto_panel : h->plan_panel, params[4], params[5], params[6], s); break;
        case SPMV_XSKIP: destroy_xskip(h->plan_xskip); rc = plan_xskip(*h, s); break;   // always rebuilt: the values are a copy
        default: set_error("spmv_csr_plan_set: unknown variant %d", target); return SPMV_ERR_VARIANT;
    }
    if (rc == SPMV_OK && variant == SPMV_AUTO) h->auto_variant = target;
    return rc;
}

int spmv_csr_plan_like(spmv_csr_t *dst, const spmv_csr_t *src, int variant, void *stream)
{
    int32_t params[8];
    int rc = spmv_csr_plan_get(src, variant, params);
    return rc ? rc : spmv_csr_plan_set(dst, variant, params, stream);
}

int64_t spmv_csr_plan_bytes(const spmv_csr_t *h, int variant)
{
    if (!h) return 0;
    const PanelPlan &panel = variant == SPMV_AUTO ? h->plan_auto_panel : h->plan_panel;
    if (variant == SPMV_AUTO) variant = h->auto_variant;
    switch (variant) {
        case SPMV_ADAPTIVE:  // chunk_lb read + carry written and re-read
            return (int64_t)(h->plan_adaptive.nchunks + 1) * 4 + (int64_t)h->plan_adaptive.nchunks * 8;
        case SPMV_TILED:     // + the two window words per chunk, the chunk lists, the 16-bit offsets and the sorted words
            return (int64_t)(h->plan_tiled.nchunks + 1) * 4 + (int64_t)h->plan_tiled.nchunks * 16 +
                   ((h->plan_tiled.n16 || h->plan_tiled.nsorted) ? (int64_t)h->plan_tiled.nchunks * 4 : 0) +
                   (int64_t)h->plan_tiled.n16 * 2 * h->plan_tiled.block * kNnzPerThread +
                   (int64_t)h->plan_tiled.nsorted * 4 * h->plan_tiled.block * kNnzPerThread +
                   (int64_t)h->plan_tiled.nblk_chunks * 256 * 4;   // block lists: up to 256 ids per chunk that has one
        case SPMV_XSKIP:     // segment list + slab partials; erow16/evals (6 B per nonzero) REPLACE col_idx/vals (8 B)
            return (int64_t)h->plan_xskip.nseg * 8 + ((int64_t)h->plan_xskip.nblocks + 1) * 4 +
                   (h->plan_xskip.slabs > 1 ? (int64_t)h->plan_xskip.nblocks * h->plan_xskip.slabs * 1024 * 8 : 0);
        case SPMV_PANEL:     // tile_ptr; packed/pvals REPLACE col_idx/vals byte for byte (sorted blocks: + the empty slots)
            if (panel.binned_mode && panel.scatter_mode)   // panel-major: column 2 + value 4, an offset per run; bin-major: product 4 + accumulator 2; the lists
                return panel.padded * (2 + 4) + panel.padded / 512 * 4 + panel.runs * 4 + panel.bm_entries * (4 + 2) +
                       (int64_t)panel.nblocks * (1024 * 4 + 16) + ((int64_t)panel.npanels + 1) * 4;
            if (panel.binned_mode)   // two tables per tile, 16-bit columns and rows, the products written and read back; pvals REPLACES vals
                return (int64_t)panel.nblocks * (2 * (int64_t)panel.npanels + 2) * 4 + panel.padded * (2 + 4 + 4) + h->nnz * 2;
            if (panel.sorted_mode)   // unit bases, block tables, the rows of the tail units, the empty slots of the units in use
                return panel.units * 4 + (int64_t)panel.nblocks * 20 + panel.tail_units * 512;
            return (int64_t)panel.nblocks * (panel.npanels + 1) * 4 + ((int64_t)panel.nblocks + 1) * 4;
        case SPMV_WAVE:      // (short rows: the same plan without the windows and the offsets; long rows: none)
            if (h->nnz > 32 * h->rows) return 0;
            return (int64_t)h->plan_wave.n_long * 8 + (int64_t)h->plan_wave.pieces * 16;
        case SPMV_SCALAR:    // (the same plan; the ordered kernel does not use the pieces)
        case SPMV_WAVE_PIPE: // the long rows' list, piece table read, partial sums written and re-read (col16 REPLACES 4 of col_idx's bytes with 2)
            return (int64_t)h->plan_wave.n_long * 8 + (int64_t)h->plan_wave.pieces * 16 + h->plan_wave.blocks * 8;
        default: return 0;
    }
}

int spmv_csr_plan_describe(const spmv_csr_t *h, int variant, char *buf, int n)
{
    if (!h || !buf || n <= 0) { set_error("spmv_csr_plan_describe: bad argument"); return SPMV_ERR_INVALID; }
    const PanelPlan *panel = &h->plan_panel;
    if (variant == SPMV_AUTO) {   // "auto -> <variant>: <that variant's plan>"
        if (h->auto_variant < 0) { snprintf(buf, (size_t)n, "not planned"); return SPMV_OK; }
        const int w = snprintf(buf, (size_t)n, "auto -> %s: ", spmv_variant_name(h->auto_variant));
        if (w < 0 || w >= n) return SPMV_OK;
        if (h->auto_variant != SPMV_PANEL) return spmv_csr_plan_describe(h, h->auto_variant, buf + w, n - w);
        panel = &h->plan_auto_panel;
        variant = SPMV_PANEL;
        buf += w;
        n -= w;
    }
    const ChunkPlan *p = variant == SPMV_ADAPTIVE ? &h->plan_adaptive : (variant == SPMV_TILED ? &h->plan_tiled : nullptr);
    if (variant == SPMV_VECTOR) snprintf(buf, (size_t)n, "lanes_per_row=%d", h->vector_width);
    else if (variant == SPMV_WAVE && h->nnz > 32 * h->rows) snprintf(buf, (size_t)n, "no plan (a wavefront per row)");
    else if (variant == SPMV_WAVE_PIPE || variant == SPMV_SCALAR || variant == SPMV_WAVE) {
        if (!h->plan_wave.ready) snprintf(buf, (size_t)n, "not planned (the first run plans)");
        else snprintf(buf, (size_t)n, "long_rows=%d pieces=%d block_rows=%d blocks=%lld blocks_with_x_window=%lld col16=%d", h->plan_wave.n_long,
                      h->plan_wave.pieces, h->plan_wave.block_rows, (long long)h->plan_wave.blocks, (long long)h->plan_wave.win_blocks,
                      h->plan_wave.d_col16 ? 1 : 0);
    }
    else if (variant == SPMV_PANEL && panel->ready && panel->binned_mode && panel->scatter_mode)
        snprintf(buf, (size_t)n, "binned scattered_products bins=%d rows_per_bin=%d panels=%d panel_columns=%d nonzeros_per_tile=%.1f rows_with_spare_sums=%d flagged_bins=%d product_workgroups_per_panel=%d padded=%lld bin_entries=%lld",
                 panel->nblocks, panel->bin_rows, panel->npanels, 1 << panel->pw_bits,
                 panel->nblocks ? (double)h->nnz / ((double)panel->nblocks * panel->npanels) : 0.0, panel->long_rows, panel->flagged_tiles,
                 panel->splits, (long long)panel->padded, (long long)panel->bm_entries);
    else if (variant == SPMV_PANEL && panel->ready && panel->binned_mode)
        snprintf(buf, (size_t)n, "binned bins=%d rows_per_bin=%d panels=%d panel_columns=%d nonzeros_per_tile=%.1f products_per_lane=%d long_rows=%d flagged_tiles=%d product_workgroups_per_panel=%d padded=%lld",
                 panel->nblocks, panel->bin_rows, panel->npanels, 1 << panel->pw_bits,
                 panel->nblocks ? (double)h->nnz / ((double)panel->nblocks * panel->npanels) : 0.0, panel->wide_pieces ? 4 : 2, panel->long_rows, panel->flagged_tiles, panel->splits,
                 (long long)panel->padded);
    else if (variant == SPMV_PANEL && panel->ready && panel->sorted_mode)
        snprintf(buf, (size_t)n, "sorted_blocks=%d rows_per_block=%d wavefronts=%d lines_per_nonzero=%.3f tail_nonzeros=%lld wide_blocks=%lld model_cost=%.3f",
                 panel->nblocks, panel->sb_rows, panel->sb_waves,
                 h->nnz ? (double)panel->lines / (double)h->nnz : 0.0, (long long)panel->tail,
                 (long long)panel->wide_blocks, colsort_model_cost(*panel, h->nnz));
    else if (variant == SPMV_PANEL && panel->ready)
        snprintf(buf, (size_t)n, "panel_columns=%d panels=%d row_blocks=%d waves_per_launch=%d launches=%d x_panels_in=%s nonzeros_per_step=%d",
                 1 << panel->pw_bits, panel->npanels, panel->nblocks,
                 panel->waves_per_launch, panel_launches(*panel), panel->lds_mode ? "LDS" : "L2", panel->lds_mode ? 64 : 256 * panel->step_vecs);
    else if (variant == SPMV_XSKIP && h->plan_xskip.ready)
        snprintf(buf, (size_t)n, "output_blocks=%d segments=%d slabs_per_block=%d", h->plan_xskip.nblocks, h->plan_xskip.nseg,
                 h->plan_xskip.slabs);
    else if (!p) snprintf(buf, (size_t)n, "no plan");
    else if (!p->block) snprintf(buf, (size_t)n, "not planned");
    else
        snprintf(buf, (size_t)n, "block=%d region=%d maxpass=%d chunks=%d staged_single=%d staged_full=%d col16_chunks=%d sorted_chunks=%d block_list_chunks=%d spanning_rows=%d persist=%d model_cost=%.3f",
                 p->block, p->region, p->maxpass, p->nchunks, p->staged_single, p->staged_full, p->n16, p->nsorted,
                 p->nblk_chunks, p->spanning_rows, p->persist ? 1 : 0, p->model_cost);
    return SPMV_OK;
}

int spmv_csr_time(spmv_csr_t *h, int variant, const float *d_x, float *d_y, int iters, void *stream,
                  float *ms_per_launch)
{
    if (iters <= 0 || !ms_per_launch) { set_error("spmv_csr_time: bad argument"); return SPMV_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t t0, t1;
    SPMV_HIP_TRY(hipEventCreate(&t0));
    SPMV_HIP_TRY(hipEventCreate(&t1));
    int rc = SPMV_OK;
    SPMV_HIP_TRY(hipEventRecord(t0, s));
    for (int i = 0; i < iters && rc == SPMV_OK; ++i) rc = spmv_csr_run(h, variant, d_x, d_y, s);
    SPMV_HIP_TRY(hipEventRecord(t1, s));
    SPMV_HIP_TRY(hipEventSynchronize(t1));
    float ms = 0.0f;
    SPMV_HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    *ms_per_launch = ms / (float)iters;
    return rc;
}

// RAII for the host-buffer conveniences
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4); }
};
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};
}  // namespace

int spmv_csr_run_host(spmv_csr_t *h, int variant, const float *x_host, float *y_host, float *kernel_ms)
{
    if (!h || (!x_host && h->cols > 0) || (!y_host && h->rows > 0)) {
        set_error("spmv_csr_run_host: null argument");
        return SPMV_ERR_INVALID;
    }
    int rc = spmv_csr_plan(h, variant, nullptr);
    if (rc) return rc;
    DevBuf dx, dy;
    EventPair ev;
    SPMV_HIP_TRY(dx.alloc(sizeof(float) * (size_t)h->cols));
    SPMV_HIP_TRY(dy.alloc(sizeof(float) * (size_t)h->rows));
    SPMV_HIP_TRY(hipMemcpy(dx.p, x_host, sizeof(float) * (size_t)h->cols, hipMemcpyHostToDevice));
    SPMV_HIP_TRY(hipEventCreate(&ev.a));
    SPMV_HIP_TRY(hipEventCreate(&ev.b));
    // two launches: the reference's TIME_KERNEL (kernel.hpp:31-48) times a single COLD launch, code object load
    // included -- that figure is kept (spmv_last_first_launch_ms); *kernel_ms is the second launch: the kernel
    SPMV_HIP_TRY(hipEventRecord(ev.a, nullptr));
    rc = spmv_csr_run(h, variant, (const float *)dx.p, (float *)dy.p, nullptr);
    SPMV_HIP_TRY(hipEventRecord(ev.b, nullptr));
    SPMV_HIP_TRY(hipEventSynchronize(ev.b));
    if (rc) return rc;
    SPMV_HIP_TRY(hipEventElapsedTime(&g_first_launch_ms, ev.a, ev.b));   // what the reference's macro would have printed
    SPMV_HIP_TRY(hipDeviceSynchronize());
    SPMV_HIP_TRY(hipEventRecord(ev.a, nullptr));
    rc = spmv_csr_run(h, variant, (const float *)dx.p, (float *)dy.p, nullptr);
    SPMV_HIP_TRY(hipEventRecord(ev.b, nullptr));
    SPMV_HIP_TRY(hipEventSynchronize(ev.b));
    if (rc) return rc;
    float ms = 0.0f;
    SPMV_HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
    if (kernel_ms) *kernel_ms = ms;
    SPMV_HIP_TRY(hipDeviceSynchronize());
    SPMV_HIP_TRY(hipMemcpy(y_host, dy.p, sizeof(float) * (size_t)h->rows, hipMemcpyDeviceToHost));
    return SPMV_OK;
}

int spmv_dense_gemv_host(int M, int N, const float *A_host, const float *x_host, float *y_host, int mode,
                         float *kernel_ms)
{
    if (M < 0 || N < 0 || ((int64_t)M * N > 0 && (!A_host || !x_host)) || (N > 0 && !y_host)) {
        set_error("spmv_dense_gemv_host: bad argument");
        return SPMV_ERR_INVALID;
    }
    int rc = require_device();
    if (rc) return rc;
    DevBuf dA, dx, dy;
    EventPair ev;
    SPMV_HIP_TRY(dA.alloc(sizeof(float) * (size_t)M * (size_t)N));
    SPMV_HIP_TRY(dx.alloc(sizeof(float) * (size_t)M));
    SPMV_HIP_TRY(dy.alloc(sizeof(float) * (size_t)N));
    SPMV_HIP_TRY(hipMemcpy(dA.p, A_host, sizeof(float) * (size_t)M * (size_t)N, hipMemcpyHostToDevice));
    SPMV_HIP_TRY(hipMemcpy(dx.p, x_host, sizeof(float) * (size_t)M, hipMemcpyHostToDevice));
    SPMV_HIP_TRY(hipEventCreate(&ev.a));
    SPMV_HIP_TRY(hipEventCreate(&ev.b));
    SPMV_HIP_TRY(hipEventRecord(ev.a, nullptr));
    rc = dense_gemv(M, N, (const float *)dA.p, (const float *)dx.p, (float *)dy.p, mode, nullptr);   // the first, cold launch
    SPMV_HIP_TRY(hipEventRecord(ev.b, nullptr));
    SPMV_HIP_TRY(hipEventSynchronize(ev.b));
    if (rc) return rc;
    SPMV_HIP_TRY(hipEventElapsedTime(&g_first_launch_ms, ev.a, ev.b));
    SPMV_HIP_TRY(hipDeviceSynchronize());
    SPMV_HIP_TRY(hipEventRecord(ev.a, nullptr));
    rc = dense_gemv(M, N, (const float *)dA.p, (const float *)dx.p, (float *)dy.p, mode, nullptr);
    SPMV_HIP_TRY(hipEventRecord(ev.b, nullptr));
    SPMV_HIP_TRY(hipEventSynchronize(ev.b));
    if (rc) return rc;
    float ms = 0.0f;
    SPMV_HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
    if (kernel_ms) *kernel_ms = ms;
    SPMV_HIP_TRY(hipDeviceSynchronize());
    SPMV_HIP_TRY(hipMemcpy(y_host, dy.p, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost));
    return SPMV_OK;
}

static int tcsr_check_dims(int M, int N, const void *A, const void *out)
{
    if (!out || M < 0 || N < 0 || (M % 32) || (N % 32) || (!A && (int64_t)M * N > 0) || (int64_t)M * N >= (1LL << 36)) {
        set_error("spmv_tcsr_from_dense: M and N must be non-negative multiples of 32 (got %d x %d)", M, N);
        return SPMV_ERR_INVALID;
    }
    return require_device();
}

int spmv_tcsr_from_dense_device(int M, int N, const float *d_A, void *stream, spmv_tcsr_t **out)
{
    int rc = tcsr_check_dims(M, N, d_A, out);
    if (rc) return rc;
    return tcsr_from_dense(M, N, d_A, (hipStream_t)stream, out);
}

int spmv_tcsr_from_dense_host(int M, int N, const float *A_host, void *stream, spmv_tcsr_t **out)
{
    int rc = tcsr_check_dims(M, N, A_host, out);
    if (rc) return rc;
    DevBuf dA;
    const size_t bytes = sizeof(float) * (size_t)M * (size_t)N;
    SPMV_HIP_TRY(dA.alloc(bytes));
    SPMV_HIP_TRY(hipMemcpyAsync(dA.p, A_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    rc = tcsr_from_dense(M, N, (const float *)dA.p, (hipStream_t)stream, out);
    (void)hipStreamSynchronize((hipStream_t)stream);
    return rc;
}

int spmv_tcsr_sizes(const spmv_tcsr_t *h, int64_t *n_blk_idx, int64_t *n_bitmaps, int64_t *n_vals)
{
    if (!h) { set_error("spmv_tcsr_sizes: null handle"); return SPMV_ERR_INVALID; }
    return tcsr_sizes(*h, n_blk_idx, n_bitmaps, n_vals);
}

int spmv_tcsr_download(const spmv_tcsr_t *h, int32_t *blk_idx, uint32_t *bitmaps, float *vals)
{
    if (!h) { set_error("spmv_tcsr_download: null handle"); return SPMV_ERR_INVALID; }
    return tcsr_download(*h, blk_idx, bitmaps, vals);
}

int spmv_tcsr_run(const spmv_tcsr_t *h, const float *d_x, float *d_y, void *stream)
{
    if (!h || !d_x || !d_y) { set_error("spmv_tcsr_run: null argument"); return SPMV_ERR_INVALID; }
    return tcsr_run(*h, d_x, d_y, (hipStream_t)stream);
}

int spmv_tcsr_run_host(const spmv_tcsr_t *h, const float *x_host, float *y_host, float *kernel_ms)
{
    if (!h || !x_host || !y_host) { set_error("spmv_tcsr_run_host: null argument"); return SPMV_ERR_INVALID; }
    int64_t nb = 0, nw = 0, nv = 0;
    tcsr_sizes(*h, &nb, &nw, &nv);
    int M = 0, N = 0;
    tcsr_dims(*h, &M, &N);
    DevBuf dx, dy;
    EventPair ev;
    SPMV_HIP_TRY(dx.alloc(sizeof(float) * (size_t)M));
    SPMV_HIP_TRY(dy.alloc(sizeof(float) * (size_t)N));
    SPMV_HIP_TRY(hipMemcpy(dx.p, x_host, sizeof(float) * (size_t)M, hipMemcpyHostToDevice));
    SPMV_HIP_TRY(hipEventCreate(&ev.a));
    SPMV_HIP_TRY(hipEventCreate(&ev.b));
    SPMV_HIP_TRY(hipEventRecord(ev.a, nullptr));
    int rc = tcsr_run(*h, (const float *)dx.p, (float *)dy.p, nullptr);   // the first, cold launch
    SPMV_HIP_TRY(hipEventRecord(ev.b, nullptr));
    SPMV_HIP_TRY(hipEventSynchronize(ev.b));
    if (rc) return rc;
    SPMV_HIP_TRY(hipEventElapsedTime(&g_first_launch_ms, ev.a, ev.b));
    SPMV_HIP_TRY(hipDeviceSynchronize());
    SPMV_HIP_TRY(hipEventRecord(ev.a, nullptr));
    rc = tcsr_run(*h, (const float *)dx.p, (float *)dy.p, nullptr);
    SPMV_HIP_TRY(hipEventRecord(ev.b, nullptr));
    SPMV_HIP_TRY(hipEventSynchronize(ev.b));
    if (rc) return rc;
    float ms = 0.0f;
    SPMV_HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
    if (kernel_ms) *kernel_ms = ms;
    SPMV_HIP_TRY(hipMemcpy(y_host, dy.p, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost));
    return SPMV_OK;
}

int spmv_tcsr_destroy(spmv_tcsr_t *h)
{
    tcsr_free(h);
    return SPMV_OK;
}

static int bitmap_check_dims(int format, int M, int N, const void *A, const void *out)
{
    if (format < 0 || format >= SPMV_FMT_COUNT) { set_error("spmv_bitmap_from_dense: unknown format %d", format); return SPMV_ERR_VARIANT; }
    if (!out || M < 0 || N < 0 || (M % 32) || (N % 32) || (!A && (int64_t)M * N > 0) || (int64_t)M * N >= (1LL << 36)) {
        set_error("spmv_bitmap_from_dense: M and N must be non-negative multiples of 32 (got %d x %d)", M, N);
        return SPMV_ERR_INVALID;
    }
    return require_device();
}

int spmv_bitmap_from_dense_device(int format, int M, int N, const float *d_A, void *stream, spmv_bitmap_t **out)
{
    int rc = bitmap_check_dims(format, M, N, d_A, out);
    if (rc) return rc;
    return bitmap_from_dense(format, M, N, d_A, (hipStream_t)stream, out);
}

int spmv_bitmap_from_dense_host(int format, int M, int N, const float *A_host, void *stream, spmv_bitmap_t **out)
{
    int rc = bitmap_check_dims(format, M, N, A_host, out);
    if (rc) return rc;
    DevBuf dA;
    const size_t bytes = sizeof(float) * (size_t)M * (size_t)N;
    SPMV_HIP_TRY(dA.alloc(bytes));
    SPMV_HIP_TRY(hipMemcpyAsync(dA.p, A_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    rc = bitmap_from_dense(format, M, N, (const float *)dA.p, (hipStream_t)stream, out);
    (void)hipStreamSynchronize((hipStream_t)stream);
    return rc;
}

int spmv_bitmap_sizes(const spmv_bitmap_t *h, int64_t *n_bitmaps, int64_t *n_vals, int32_t stats[4])
{
    if (!h) { set_error("spmv_bitmap_sizes: null handle"); return SPMV_ERR_INVALID; }
    bitmap_info(*h, nullptr, nullptr, nullptr, n_bitmaps, n_vals, stats);
    return SPMV_OK;
}

int spmv_bitmap_download(const spmv_bitmap_t *h, uint32_t *bitmaps, float *vals)
{
    if (!h) { set_error("spmv_bitmap_download: null handle"); return SPMV_ERR_INVALID; }
    return bitmap_download(*h, bitmaps, vals);
}

int spmv_bitmap_run(const spmv_bitmap_t *h, const float *d_x, float *d_y, void *stream)
{
    if (!h || !d_x || !d_y) { set_error("spmv_bitmap_run: null argument"); return SPMV_ERR_INVALID; }
    if (int rc = require_current(bitmap_device(*h), "spmv_bitmap_run")) return rc;
    return bitmap_run(*h, d_x, d_y, (hipStream_t)stream);
}

int spmv_bitmap_run_host(const spmv_bitmap_t *h, const float *x_host, float *y_host, float *kernel_ms)
{
    if (!h || !x_host || !y_host) { set_error("spmv_bitmap_run_host: null argument"); return SPMV_ERR_INVALID; }
    int M = 0, N = 0;
    bitmap_info(*h, nullptr, &M, &N, nullptr, nullptr, nullptr);
    DevBuf dx, dy;
    EventPair ev;
    SPMV_HIP_TRY(dx.alloc(sizeof(float) * (size_t)M));
    SPMV_HIP_TRY(dy.alloc(sizeof(float) * (size_t)N));
    SPMV_HIP_TRY(hipMemcpy(dx.p, x_host, sizeof(float) * (size_t)M, hipMemcpyHostToDevice));
    SPMV_HIP_TRY(hipEventCreate(&ev.a));
    SPMV_HIP_TRY(hipEventCreate(&ev.b));
    SPMV_HIP_TRY(hipEventRecord(ev.a, nullptr));
    int rc = bitmap_run(*h, (const float *)dx.p, (float *)dy.p, nullptr);   // the first, cold launch
    SPMV_HIP_TRY(hipEventRecord(ev.b, nullptr));
    SPMV_HIP_TRY(hipEventSynchronize(ev.b));
    if (rc) return rc;
    SPMV_HIP_TRY(hipEventElapsedTime(&g_first_launch_ms, ev.a, ev.b));
    SPMV_HIP_TRY(hipDeviceSynchronize());
    SPMV_HIP_TRY(hipEventRecord(ev.a, nullptr));
    rc = bitmap_run(*h, (const float *)dx.p, (float *)dy.p, nullptr);
    SPMV_HIP_TRY(hipEventRecord(ev.b, nullptr));
    SPMV_HIP_TRY(hipEventSynchronize(ev.b));
    if (rc) return rc;
    float ms = 0.0f;
    SPMV_HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
    if (kernel_ms) *kernel_ms = ms;
    SPMV_HIP_TRY(hipMemcpy(y_host, dy.p, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost));
    return SPMV_OK;
}

int spmv_bitmap_destroy(spmv_bitmap_t *h)
{
    bitmap_free(h);
    return SPMV_OK;
}

int spmv_dense_gemv(int M, int N, const float *d_A, const float *d_x, float *d_y, int mode, void *stream)
{
    if (M < 0 || N < 0 || ((int64_t)M * N > 0 && (!d_A || !d_x)) || (N > 0 && !d_y)) {
        set_error("spmv_dense_gemv: bad argument");
        return SPMV_ERR_INVALID;
    }
    int rc = require_device();
    if (rc) return rc;
    return dense_gemv(M, N, d_A, d_x, d_y, mode, (hipStream_t)stream);
}

int64_t spmv_dense_gemv_workspace_bytes(int N, int mode) { return (int64_t)dense_gemv_workspace_bytes(N, mode); }

int spmv_dense_gemv_ws(int M, int N, const float *d_A, const float *d_x, float *d_y, int mode, void *d_workspace,
                       int64_t workspace_bytes, void *stream)
{
    if (M < 0 || N < 0 || ((int64_t)M * N > 0 && (!d_A || !d_x)) || (N > 0 && !d_y) || workspace_bytes < 0) {
        set_error("spmv_dense_gemv_ws: bad argument");
        return SPMV_ERR_INVALID;
    }
    int rc = require_device();
    if (rc) return rc;
    return dense_gemv_ws(M, N, d_A, d_x, d_y, mode, d_workspace, (size_t)workspace_bytes, (hipStream_t)stream);
}

int spmv_asp_retile(int M, int N, const float *d_A, float *d_asp, void *stream)
{
    if (M < 0 || N < 0 || (M % 32) || (N % 32) || ((int64_t)M * N > 0 && (!d_A || !d_asp))) {
        set_error("spmv_asp_retile: bad argument (M and N are multiples of 32, like the reference's tester asserts)");
        return SPMV_ERR_INVALID;
    }
    int rc = require_device();
    if (rc) return rc;
    return asp_retile(M, N, d_A, d_asp, (hipStream_t)stream);
}

int spmv_asp_gemv_ws(int M, int N, const float *d_asp, const float *d_x, float *d_y, void *d_workspace, int64_t workspace_bytes,
                     void *stream)
{
    if (M < 0 || N < 0 || (M % 32) || (N % 32) || ((int64_t)M * N > 0 && (!d_asp || !d_x)) || (N > 0 && !d_y) ||
        workspace_bytes < 0) {
        set_error("spmv_asp_gemv_ws: bad argument");
        return SPMV_ERR_INVALID;
    }
    int rc = require_device();
    if (rc) return rc;
    return asp_gemv_ws(M, N, d_asp, d_x, d_y, d_workspace, (size_t)workspace_bytes, (hipStream_t)stream);
}

int spmv_synth_fill(uint64_t seed, int64_t row0, int64_t n_local, int64_t rows, int64_t cols, int64_t band,
                    const int32_t *d_row_ptr, int32_t *d_col_idx, float *d_vals, void *stream)
{
    if (n_local < 0 || row0 < 0 || row0 + n_local > rows || cols <= 0 || cols >= (1LL << 31) ||
        rows >= (1LL << 31) || !d_row_ptr) {
        set_error("spmv_synth_fill: bad argument");
        return SPMV_ERR_INVALID;
    }
    int rc = require_device();
    if (rc) return rc;
    return synth_fill(seed, row0, n_local, rows, cols, band, d_row_ptr, d_col_idx, d_vals, (hipStream_t)stream);
}

int spmv_synth_x(uint64_t seed, int64_t j0, int64_t n, float *d_x, void *stream)
{
    if (n < 0 || j0 < 0 || (n > 0 && !d_x)) { set_error("spmv_synth_x: bad argument"); return SPMV_ERR_INVALID; }
    int rc = require_device();
    if (rc) return rc;
    return synth_x(seed, j0, n, d_x, (hipStream_t)stream);
}

}  // extern "C"
