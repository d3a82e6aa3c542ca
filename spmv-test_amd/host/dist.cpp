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
    int32_t *d_params = nullptr;   // status + 8 ints for spmv_dist_plan_like_root
    bool local = false;            // spmv_dist_init_local: no communicator (SPMV_DIST_PEER_STORE only)
};

// One rank's side of the pipelined step (include/spmv_dist.h): the side stream the exchanges run on and, per block group,
// the event behind its product (the exchange waits for it) and the event behind its exchange (the next step's product waits).
struct spmv_dist_pipe {
    spmv_dist *d = nullptr;
    int S = 0, exchange = SPMV_DIST_ALLGATHER;
    int64_t sub_rows = 0, cols = 0;
    hipStream_t comm = nullptr;
    std::vector<hipEvent_t> ev_mult, ev_done;
    std::vector<char> pending;     // ev_done[s] has been recorded at least once
    // PEER_STORE: the OTHER ranks write this rank's y_full.  ev_release marks the point of this rank's stream behind which the
    // readers of the last step's y_full lie (spmv_dist_pipe_release, or the start of this rank's own next step): a peer's
    // stores of the next step wait for it.
    hipEvent_t ev_release = nullptr;
    bool released = false;
    // PEER_STORE (spmv_dist_pipe_link): every rank's pipe and y_full, index = rank
    std::vector<spmv_dist_pipe *> peers;
    std::vector<float *> peer_y;
    // footprint exchange (spmv_dist_pipe_set_footprint): rank q needs the rows of need_k intervals of y; none set = everything
    std::vector<int64_t> need_lo, need_hi;   // [world * need_k]
    int need_k = 0;
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
    if (hipMalloc((void **)&d->d_params, 9 * sizeof(int32_t)) != hipSuccess) { ncclCommDestroy(d->comm); delete d; return fail(SPMV_ERR_HIP, "hipMalloc"); }
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
        DIST_HIP(hipMalloc((void **)&d->d_params, 9 * sizeof(int32_t)));
        out[i] = d;
    }
    (void)hipSetDevice(prev);
    return SPMV_OK;
}

int spmv_dist_init_local(int nranks, const int *devices, spmv_dist_t **out)
{
    if (!out || nranks < 1) return fail(SPMV_ERR_INVALID, "spmv_dist_init_local: bad argument");
    const int ndev = spmv_device_count();
    if (ndev < 1) return fail(SPMV_ERR_NO_DEVICE, "spmv_dist_init_local: no HIP device");
    for (int i = 0; i < nranks; ++i) {
        const int dev = devices ? devices[i] : i % ndev;
        if (dev < 0 || dev >= ndev) return fail(SPMV_ERR_INVALID, "spmv_dist_init_local: device %d of rank %d is not visible", dev, i);
        spmv_dist *d = new spmv_dist();
        d->world = nranks; d->rank = i; d->device = dev; d->local = true;
        out[i] = d;
    }
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
    if (d->local) return fail(SPMV_ERR_INVALID, "spmv_dist_broadcast_x: local ranks have no communicator (copy x with hipMemcpyPeerAsync)");
    if (int rc = require_device(d, "spmv_dist_broadcast_x")) return rc;
    if (d->cols == 0) return SPMV_OK;
    DIST_NCCL(ncclBroadcast(d_x, d_x, (size_t)d->cols, ncclFloat, root, d->comm, (hipStream_t)stream));
    return SPMV_OK;
}

int spmv_dist_plan_like_root(spmv_dist_t *d, spmv_csr_t *shard, int variant, int root, void *stream)
{
    if (!d || !shard || root < 0 || root >= d->world) return fail(SPMV_ERR_INVALID, "spmv_dist_plan_like_root: bad argument");
    if (d->local) return fail(SPMV_ERR_INVALID, "spmv_dist_plan_like_root: local ranks carry plans with spmv_csr_plan_get / _set");
    if (int rc = require_device(d, "spmv_dist_plan_like_root")) return rc;
    hipStream_t s = (hipStream_t)stream;
    // nine numbers travel: the root's status first.  A root whose plan fails (XSKIP's table limit, PANEL's column limit on
    // its shard only) still takes part in the broadcast -- the other ranks are already inside it and would wait for ever
    //  -- and every rank returns the root's error.
    int32_t msg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int root_rc = SPMV_OK;
    if (d->rank == root) {
        root_rc = spmv_csr_plan(shard, variant, stream);
        if (root_rc == SPMV_OK) root_rc = spmv_csr_plan_get(shard, variant, msg + 1);
        if (root_rc) (void)fail(root_rc, "spmv_dist_plan_like_root (root): %s", spmv_last_error());
        msg[0] = root_rc;
        DIST_HIP(hipMemcpyAsync(d->d_params, msg, sizeof msg, hipMemcpyHostToDevice, s));
    }
    DIST_NCCL(ncclBroadcast(d->d_params, d->d_params, 9, ncclInt32, root, d->comm, s));
    if (d->rank == root) return root_rc;
    DIST_HIP(hipMemcpyAsync(msg, d->d_params, sizeof msg, hipMemcpyDeviceToHost, s));
    DIST_HIP(hipStreamSynchronize(s));
    if (msg[0] != SPMV_OK) return fail(msg[0], "spmv_dist_plan_like_root (rank %d): the root's plan failed with status %d", d->rank, msg[0]);
    const int rc = spmv_csr_plan_set(shard, variant, msg + 1, stream);
    if (rc) return fail(rc, "spmv_dist_plan_like_root (rank %d): %s", d->rank, spmv_last_error());
    return SPMV_OK;
}

int spmv_dist_allgather_y(spmv_dist_t *d, float *d_y_full, void *stream)
{
    if (!d || d->bounds.empty()) return fail(SPMV_ERR_INVALID, "spmv_dist_allgather_y: no partition set");
    if (int rc = require_device(d, "spmv_dist_allgather_y")) return rc;
    if (d->world == 1 || d->bounds.back() == 0) return SPMV_OK;
    if (d->local) return fail(SPMV_ERR_INVALID, "spmv_dist_allgather_y: local ranks have no communicator (use a SPMV_DIST_PEER_STORE pipe)");
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

// ---- the pipelined step -------------------------------------------------------------------------------------------------
int spmv_dist_pipe_create(spmv_dist_t *d, int S, int64_t sub_rows, int64_t cols, int exchange, spmv_dist_pipe_t **out)
{
    if (!d || !out || S < 1 || S > 4096 || sub_rows < 0 || cols < 0 || exchange < SPMV_DIST_ALLGATHER || exchange > SPMV_DIST_PEER_STORE)
        return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_create: bad argument");
    if (d->local && exchange != SPMV_DIST_PEER_STORE)
        return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_create: local ranks (spmv_dist_init_local) exchange by peer stores only");
    if ((int64_t)S * d->world * sub_rows >= (1ll << 40)) return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_create: too many rows");
    if (int rc = require_device(d, "spmv_dist_pipe_create")) return rc;
    spmv_dist_pipe *p = new spmv_dist_pipe();
    p->d = d; p->S = S; p->sub_rows = sub_rows; p->cols = cols; p->exchange = exchange;
    p->ev_mult.assign(S, nullptr);
    p->ev_done.assign(S, nullptr);
    p->pending.assign(S, 0);
    hipError_t e = hipStreamCreateWithFlags(&p->comm, hipStreamNonBlocking);
    for (int s = 0; s < S && e == hipSuccess; ++s) {
        e = hipEventCreateWithFlags(&p->ev_mult[s], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_done[s], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_release, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)spmv_dist_pipe_destroy(p);
        return fail(SPMV_ERR_HIP, "spmv_dist_pipe_create: %s", hipGetErrorString(e));
    }
    *out = p;
    return SPMV_OK;
}

int spmv_dist_pipe_link(spmv_dist_pipe_t *const *pipes, float *const *d_y_full, int n)
{
    if (!pipes || !d_y_full || n < 1) return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_link: bad argument");
    for (int r = 0; r < n; ++r) {
        const spmv_dist_pipe *p = pipes[r];
        if (!p || !d_y_full[r] || p->d->world != n || p->d->rank != r || p->exchange != SPMV_DIST_PEER_STORE || p->S != pipes[0]->S ||
            p->sub_rows != pipes[0]->sub_rows)
            return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_link: entry %d is not rank %d's PEER_STORE pipe of this %d-rank job", r, r, n);
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int r = 0; r < n; ++r) {
        spmv_dist_pipe *p = pipes[r];
        p->peers.assign(pipes, pipes + n);
        p->peer_y.assign(d_y_full, d_y_full + n);
        for (int q = 0; q < n; ++q) {
            const int a = p->d->device, b = pipes[q]->d->device;
            if (a == b) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) {
                (void)hipSetDevice(prev);
                return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_link: device %d cannot access device %d", a, b);
            }
            (void)hipSetDevice(a);
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                (void)hipSetDevice(prev);
                return fail(SPMV_ERR_HIP, "hipDeviceEnablePeerAccess(%d -> %d): %s", a, b, hipGetErrorString(e));
            }
            (void)hipGetLastError();
        }
    }
    (void)hipSetDevice(prev);
    return SPMV_OK;
}

int spmv_dist_pipe_set_footprint(spmv_dist_pipe_t *p, const int64_t *need_lo, const int64_t *need_hi, int per_rank)
{
    if (!p || !need_lo || !need_hi || per_rank < 1 || per_rank > 4096) return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_set_footprint: bad argument");
    if (p->exchange == SPMV_DIST_ALLGATHER)
        return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_set_footprint: an all-gather moves whole slices (use SPMV_DIST_P2P or SPMV_DIST_PEER_STORE)");
    const int n = p->d->world * per_rank;
    for (int i = 0; i < n; ++i)
        if (need_lo[i] < 0 || need_hi[i] < need_lo[i]) return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_set_footprint: bad interval %d", i);
    p->need_lo.assign(need_lo, need_lo + n);
    p->need_hi.assign(need_hi, need_hi + n);
    p->need_k = per_rank;
    return SPMV_OK;
}

namespace {

// how many parts of rank `owner`'s slot of group s rank `peer` needs (1 = the whole slot, without a footprint), and part k
inline int needed_parts(const spmv_dist_pipe *p) { return p->need_k > 0 ? p->need_k : 1; }
inline void needed_part(const spmv_dist_pipe *p, int s, int owner, int peer, int k, int64_t *a, int64_t *b)
{
    const int64_t r0 = ((int64_t)s * p->d->world + owner) * p->sub_rows, r1 = r0 + p->sub_rows;
    *a = r0;
    *b = r1;
    if (p->need_k == 0) return;
    const int64_t lo = p->need_lo[(size_t)peer * p->need_k + k], hi = p->need_hi[(size_t)peer * p->need_k + k];
    if (lo > *a) *a = lo;
    if (hi < *b) *b = hi;
    if (*b < *a) *b = *a;
}

// the exchange of block group s on the pipe's side stream (which already waits for whatever must come first)
int exchange_group(spmv_dist_pipe *p, int s, float *d_y_full)
{
    spmv_dist *d = p->d;
    const int world = d->world;
    if (world == 1 || p->sub_rows == 0) return SPMV_OK;
    float *grp = d_y_full + (int64_t)s * world * p->sub_rows;
    float *mine = grp + (int64_t)d->rank * p->sub_rows;
    const size_t n = (size_t)p->sub_rows;
    switch (p->exchange) {
        case SPMV_DIST_ALLGATHER:      // in place: rank r's slot already holds its slice (sendbuff == recvbuff + rank * count)
            DIST_NCCL(ncclAllGather(mine, grp, n, ncclFloat, d->comm, p->comm));
            return SPMV_OK;
        case SPMV_DIST_P2P: {
            DIST_NCCL(ncclGroupStart());
            for (int q = 0; q < world; ++q) {
                if (q == d->rank) continue;
                // what q needs of my slot goes out, what I need of q's slot comes in: both sides compute both ranges from the
                // same footprints, so the send and the receive of a pair agree in size (and are both skipped when empty)
                // (a row inside two of a rank's intervals -- a band wider than a block -- travels twice: harmless, and both sides agree)
                ncclResult_t e = ncclSuccess;
                for (int k = 0; k < needed_parts(p) && e == ncclSuccess; ++k) {
                    int64_t sa, sb, ra, rb;
                    needed_part(p, s, d->rank, q, k, &sa, &sb);
                    needed_part(p, s, q, d->rank, k, &ra, &rb);
                    if (sb > sa) e = ncclSend(d_y_full + sa, (size_t)(sb - sa), ncclFloat, q, d->comm, p->comm);
                    if (e == ncclSuccess && rb > ra) e = ncclRecv(d_y_full + ra, (size_t)(rb - ra), ncclFloat, q, d->comm, p->comm);
                }
                if (e != ncclSuccess) { (void)ncclGroupEnd(); return fail(SPMV_ERR_HIP, "ncclSend/ncclRecv(peer %d): %s", q, ncclGetErrorString(e)); }
            }
            DIST_NCCL(ncclGroupEnd());
            return SPMV_OK;
        }
        default: {                     // PEER_STORE: my slice into the same place of every peer's y_full
            if ((int)p->peers.size() != world) return fail(SPMV_ERR_INVALID, "SPMV_DIST_PEER_STORE: spmv_dist_pipe_link has not been called");
            if (p->peer_y[d->rank] != d_y_full) return fail(SPMV_ERR_INVALID, "SPMV_DIST_PEER_STORE: y_full differs from the linked buffer");
            for (int k = 1; k < world; ++k) {   // start with the next rank: the owners do not all hit one peer at once
                const int q = (d->rank + k) % world;
                // q's readers of the previous step's y_full come first (its release mark, if it has set one)
                if (p->peers[q]->released) DIST_HIP(hipStreamWaitEvent(p->comm, p->peers[q]->ev_release, 0));
                for (int part = 0; part < needed_parts(p); ++part) {
                    int64_t a, b;
                    needed_part(p, s, d->rank, q, part, &a, &b);   // (the whole slot without a footprint)
                    if (b > a)
                        DIST_HIP(hipMemcpyPeerAsync(p->peer_y[q] + a, p->peers[q]->d->device, d_y_full + a, d->device,
                                                    (size_t)(b - a) * sizeof(float), p->comm));
                }
            }
            return SPMV_OK;
        }
    }
}

}  // namespace

int spmv_dist_pipe_step(spmv_dist_pipe_t *p, spmv_csr_t *const *blocks, int variant, const float *d_x, float *d_y_full, void *stream)
{
    if (!p || !blocks || !d_y_full) return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_step: null argument");
    spmv_dist *d = p->d;
    if (int rc = require_device(d, "spmv_dist_pipe_step")) return rc;
    hipStream_t st = (hipStream_t)stream;
    // peer stores: whatever this rank's stream holds so far (the readers of the last step's y_full) precedes the stores the
    // peers enqueue from here on.  (A peer whose step was enqueued BEFORE this call could not wait for it: where the
    // ranks of one process are stepped one after the other, spmv_dist_pipe_release on every rank first closes that gap.)
    if (p->exchange == SPMV_DIST_PEER_STORE && d->world > 1) {
        DIST_HIP(hipEventRecord(p->ev_release, st));
        p->released = true;
    }
    for (int s = 0; s < p->S; ++s) {
        int64_t rows = 0, cols = 0;
        if (!blocks[s] || spmv_csr_dims(blocks[s], &rows, &cols, nullptr) != SPMV_OK || rows != p->sub_rows || cols != p->cols)
            return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_step: block %d is not %lld x %lld", s, (long long)p->sub_rows, (long long)p->cols);
        // the product overwrites the slot the previous step's exchange of this group read from
        if (p->pending[s]) DIST_HIP(hipStreamWaitEvent(st, p->ev_done[s], 0));
        float *slot = d_y_full + ((int64_t)s * d->world + d->rank) * p->sub_rows;
        if (rows > 0) {
            const int rc = spmv_csr_run(blocks[s], variant, d_x, slot, stream);
            if (rc) return fail(rc, "spmv_dist_pipe_step (block %d): %s", s, spmv_last_error());
        }
        if (d->world > 1) {
            DIST_HIP(hipEventRecord(p->ev_mult[s], st));
            DIST_HIP(hipStreamWaitEvent(p->comm, p->ev_mult[s], 0));   // exchange s starts when product s is done
            if (int rc = exchange_group(p, s, d_y_full)) return rc;
            DIST_HIP(hipEventRecord(p->ev_done[s], p->comm));
            p->pending[s] = 1;
        }
    }
    return SPMV_OK;
}

int spmv_dist_pipe_exchange_only(spmv_dist_pipe_t *p, float *d_y_full, void *stream)
{
    if (!p || !d_y_full) return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_exchange_only: null argument");
    spmv_dist *d = p->d;
    if (int rc = require_device(d, "spmv_dist_pipe_exchange_only")) return rc;
    if (d->world == 1) return SPMV_OK;
    hipStream_t st = (hipStream_t)stream;
    DIST_HIP(hipEventRecord(p->ev_mult[0], st));
    DIST_HIP(hipStreamWaitEvent(p->comm, p->ev_mult[0], 0));
    for (int s = 0; s < p->S; ++s) {
        if (int rc = exchange_group(p, s, d_y_full)) return rc;
        DIST_HIP(hipEventRecord(p->ev_done[s], p->comm));
        p->pending[s] = 1;
    }
    return SPMV_OK;
}

int spmv_dist_pipe_finish(spmv_dist_pipe_t *p, void *stream)
{
    if (!p) return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_finish: null handle");
    hipStream_t st = (hipStream_t)stream;
    // RCCL: this rank's exchanges deliver everyone's slices.  Peer stores: a rank's y_full is filled by the OTHER ranks'
    // side streams, so it waits for every rank's exchanges (events of one process order streams across devices).
    if (p->exchange == SPMV_DIST_PEER_STORE && !p->peers.empty()) {
        for (spmv_dist_pipe *q : p->peers)
            for (int s = 0; s < q->S; ++s)
                if (q->pending[s]) DIST_HIP(hipStreamWaitEvent(st, q->ev_done[s], 0));
        return SPMV_OK;
    }
    for (int s = 0; s < p->S; ++s)
        if (p->pending[s]) DIST_HIP(hipStreamWaitEvent(st, p->ev_done[s], 0));
    return SPMV_OK;
}

int spmv_dist_pipe_release(spmv_dist_pipe_t *p, void *stream)
{
    if (!p) return fail(SPMV_ERR_INVALID, "spmv_dist_pipe_release: null handle");
    if (int rc = require_device(p->d, "spmv_dist_pipe_release")) return rc;
    if (p->exchange != SPMV_DIST_PEER_STORE || p->d->world == 1) return SPMV_OK;   // RCCL: the receives are enqueued by this rank itself
    DIST_HIP(hipEventRecord(p->ev_release, (hipStream_t)stream));
    p->released = true;
    return SPMV_OK;
}

int spmv_dist_pipe_destroy(spmv_dist_pipe_t *p)
{
    if (!p) return SPMV_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(p->d->device);
    if (p->comm) (void)hipStreamSynchronize(p->comm);
    for (hipEvent_t e : p->ev_mult) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->ev_done) if (e) (void)hipEventDestroy(e);
    if (p->ev_release) (void)hipEventDestroy(p->ev_release);
    if (p->comm) (void)hipStreamDestroy(p->comm);
    (void)hipSetDevice(prev);
    delete p;
    return SPMV_OK;
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
