// dist.cpp -- libspmv_dist.so: the C ABI of include/spmv_dist.h on RCCL (one spmv_dist_t per rank).
//
// No counterpart in the reference (single GPU, no collective: SURVEY.md section 5); this is BASELINE.json's config 5
// for a C++ caller: row blocks, x broadcast once, y slices concatenated by an all-gather over xGMI.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "spmv_dist.h"

struct spmv_dist {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0, device = 0;
    std::vector<int64_t> bounds;   // world+1
    int64_t cols = 0;
    bool uniform = false;          // all blocks equal: one in-place ncclAllGather
    int32_t *d_params = nullptr;   // 8 ints for spmv_dist_plan_like_root
};

namespace {

// the message behind the last non-zero status of this library on this thread (spmv_dist_last_error)
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define DIST_NCCL(call)                                                                              \
    do {                                                                                             \
        ncclResult_t r__ = (call);                                                                   \
        if (r__ != ncclSuccess) return fail(SPMV_ERR_HIP, "RCCL error %s:%d: %s (%s)", __FILE__, __LINE__, ncclGetErrorString(r__), #call); \
    } while (0)
#define DIST_HIP(call)                                                                               \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess) return fail(SPMV_ERR_HIP, "HIP error %s:%d: %s (%s)", __FILE__, __LINE__, hipGetErrorString(e__), #call); \
    } while (0)

int require_device(const spmv_dist *d, const char *what)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != d->device)
        return fail(SPMV_ERR_INVALID, "%s: rank %d lives on device %d but the current device is %d", what, d->rank, d->device, cur);
    return SPMV_OK;
}

}  // namespace

extern "C" {

const char *spmv_dist_last_error(void) { return g_err; }

int spmv_dist_get_unique_id(void *id128)
{
    if (!id128) return fail(SPMV_ERR_INVALID, "spmv_dist_get_unique_id: null");
    static_assert(sizeof(ncclUniqueId) == SPMV_DIST_ID_BYTES, "id size");
    ncclUniqueId id;
    DIST_NCCL(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return SPMV_OK;
}

int spmv_dist_init(int world, int rank, const void *id128, spmv_dist_t **out)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail(SPMV_ERR_INVALID, "spmv_dist_init: bad argument");
    if (spmv_device_count() < 1) return fail(SPMV_ERR_NO_DEVICE, "spmv_dist_init: no HIP device");
    spmv_dist *d = new spmv_dist();
    d->world = world; d->rank = rank;
    if (hipGetDevice(&d->device) != hipSuccess) d->device = 0;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclResult_t r = ncclCommInitRank(&d->comm, world, id, rank);
    if (r != ncclSuccess) { delete d; return fail(SPMV_ERR_HIP, "ncclCommInitRank: %s", ncclGetErrorString(r)); }
    if (hipMalloc((void **)&d->d_params, 8 * sizeof(int32_t)) != hipSuccess) { ncclCommDestroy(d->comm); delete d; return fail(SPMV_ERR_HIP, "hipMalloc"); }
    *out = d;
    return SPMV_OK;
}

int spmv_dist_init_all(int ndev, const int *devices, spmv_dist_t **out)
{
    if (!out || ndev < 1) return fail(SPMV_ERR_INVALID, "spmv_dist_init_all: bad argument");
    if (spmv_device_count() < ndev) return fail(SPMV_ERR_NO_DEVICE, "spmv_dist_init_all: %d devices asked, %d visible", ndev, spmv_device_count());
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; ++i) devs[i] = devices ? devices[i] : i;
    std::vector<ncclComm_t> comms(ndev);
    DIST_NCCL(ncclCommInitAll(comms.data(), ndev, devs.data()));
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int i = 0; i < ndev; ++i) {
        spmv_dist *d = new spmv_dist();
        d->comm = comms[i]; d->world = ndev; d->rank = i; d->device = devs[i];
        DIST_HIP(hipSetDevice(devs[i]));
        DIST_HIP(hipMalloc((void **)&d->d_params, 8 * sizeof(int32_t)));
        out[i] = d;
    }
    (void)hipSetDevice(prev);
    return SPMV_OK;
}

int spmv_dist_group_start(void) { DIST_NCCL(ncclGroupStart()); return SPMV_OK; }
int spmv_dist_group_end(void) { DIST_NCCL(ncclGroupEnd()); return SPMV_OK; }

int spmv_dist_rank(const spmv_dist_t *d, int *world, int *rank, int *device)
{
    if (!d) return fail(SPMV_ERR_INVALID, "spmv_dist_rank: null handle");
    if (world) *world = d->world;
    if (rank) *rank = d->rank;
    if (device) *device = d->device;
    return SPMV_OK;
}

int spmv_dist_set_partition(spmv_dist_t *d, const int64_t *row_bounds, int64_t cols)
{
    if (!d || !row_bounds || cols < 0) return fail(SPMV_ERR_INVALID, "spmv_dist_set_partition: bad argument");
    if (row_bounds[0] != 0) return fail(SPMV_ERR_INVALID, "spmv_dist_set_partition: row_bounds[0] must be 0");
    for (int r = 0; r < d->world; ++r)
        if (row_bounds[r + 1] < row_bounds[r]) return fail(SPMV_ERR_INVALID, "spmv_dist_set_partition: bounds decrease at rank %d", r);
    d->bounds.assign(row_bounds, row_bounds + d->world + 1);
    d->cols = cols;
    d->uniform = true;
    for (int r = 1; r < d->world; ++r)
        if (d->bounds[r + 1] - d->bounds[r] != d->bounds[1] - d->bounds[0]) d->uniform = false;
    return SPMV_OK;
}

int spmv_dist_broadcast_x(spmv_dist_t *d, float *d_x, int root, void *stream)
{
    if (!d || (!d_x && d->cols > 0) || root < 0 || root >= d->world) return fail(SPMV_ERR_INVALID, "spmv_dist_broadcast_x: bad argument");
    if (int rc = require_device(d, "spmv_dist_broadcast_x")) return rc;
    if (d->cols == 0) return SPMV_OK;
    DIST_NCCL(ncclBroadcast(d_x, d_x, (size_t)d->cols, ncclFloat, root, d->comm, (hipStream_t)stream));
    return SPMV_OK;
}

int spmv_dist_plan_like_root(spmv_dist_t *d, spmv_csr_t *shard, int variant, int root, void *stream)
{
    if (!d || !shard || root < 0 || root >= d->world) return fail(SPMV_ERR_INVALID, "spmv_dist_plan_like_root: bad argument");
    if (int rc = require_device(d, "spmv_dist_plan_like_root")) return rc;
    hipStream_t s = (hipStream_t)stream;
    int32_t params[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (d->rank == root) {
        int rc = spmv_csr_plan(shard, variant, stream);
        if (rc == SPMV_OK) rc = spmv_csr_plan_get(shard, variant, params);
        if (rc) return fail(rc, "spmv_dist_plan_like_root (root): %s", spmv_last_error());
        DIST_HIP(hipMemcpyAsync(d->d_params, params, sizeof params, hipMemcpyHostToDevice, s));
    }
    DIST_NCCL(ncclBroadcast(d->d_params, d->d_params, 8, ncclInt32, root, d->comm, s));
    if (d->rank != root) {
        DIST_HIP(hipMemcpyAsync(params, d->d_params, sizeof params, hipMemcpyDeviceToHost, s));
        DIST_HIP(hipStreamSynchronize(s));
        const int rc = spmv_csr_plan_set(shard, variant, params, stream);
        if (rc) return fail(rc, "spmv_dist_plan_like_root (rank %d): %s", d->rank, spmv_last_error());
    }
    return SPMV_OK;
}

int spmv_dist_allgather_y(spmv_dist_t *d, float *d_y_full, void *stream)
{
    if (!d || d->bounds.empty()) return fail(SPMV_ERR_INVALID, "spmv_dist_allgather_y: no partition set");
    if (int rc = require_device(d, "spmv_dist_allgather_y")) return rc;
    if (d->world == 1 || d->bounds.back() == 0) return SPMV_OK;
    if (!d_y_full) return fail(SPMV_ERR_INVALID, "spmv_dist_allgather_y: null y");
    hipStream_t s = (hipStream_t)stream;
    if (d->uniform) {
        // in place: rank r's slot already holds its slice (sendbuff == recvbuff + rank * count)
        const size_t n = (size_t)(d->bounds[1] - d->bounds[0]);
        DIST_NCCL(ncclAllGather(d_y_full + d->bounds[d->rank], d_y_full, n, ncclFloat, d->comm, s));
        return SPMV_OK;
    }
    // unequal blocks: all-gather-v as one grouped broadcast per owner
    DIST_NCCL(ncclGroupStart());
    for (int r = 0; r < d->world; ++r) {
        const size_t n = (size_t)(d->bounds[r + 1] - d->bounds[r]);
        if (n == 0) continue;
        float *p = d_y_full + d->bounds[r];
        ncclResult_t e = ncclBroadcast(p, p, n, ncclFloat, r, d->comm, s);
        if (e != ncclSuccess) { (void)ncclGroupEnd(); return fail(SPMV_ERR_HIP, "ncclBroadcast(rank %d): %s", r, ncclGetErrorString(e)); }
    }
    DIST_NCCL(ncclGroupEnd());
    return SPMV_OK;
}

int spmv_dist_spmv(spmv_dist_t *d, spmv_csr_t *shard, int variant, const float *d_x, float *d_y_full, void *stream)
{
    if (!d || !shard || d->bounds.empty()) return fail(SPMV_ERR_INVALID, "spmv_dist_spmv: bad argument or no partition");
    if (int rc = require_device(d, "spmv_dist_spmv")) return rc;
    int64_t rows = 0, cols = 0;
    (void)spmv_csr_dims(shard, &rows, &cols, nullptr);
    if (rows != d->bounds[d->rank + 1] - d->bounds[d->rank] || cols != d->cols)
        return fail(SPMV_ERR_INVALID, "spmv_dist_spmv: shard is %lld x %lld, the partition gives rank %d %lld x %lld", (long long)rows,
                    (long long)cols, d->rank, (long long)(d->bounds[d->rank + 1] - d->bounds[d->rank]), (long long)d->cols);
    if (rows > 0) {
        const int rc = spmv_csr_run(shard, variant, d_x, d_y_full + d->bounds[d->rank], stream);
        if (rc) return fail(rc, "spmv_dist_spmv: %s", spmv_last_error());
    }
    return spmv_dist_allgather_y(d, d_y_full, stream);
}

int spmv_dist_destroy(spmv_dist_t *d)
{
    if (!d) return SPMV_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(d->device);
    if (d->d_params) (void)hipFree(d->d_params);
    if (d->comm) (void)ncclCommDestroy(d->comm);
    (void)hipSetDevice(prev);
    delete d;
    return SPMV_OK;
}

}  // extern "C"
