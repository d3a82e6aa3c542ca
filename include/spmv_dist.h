/*
 * spmv_dist.h -- C ABI of libspmv_dist.so: the row-block exchange of the sharded SpMV over RCCL.
 *
 * The reference is single-GPU (last row: no collective anywhere); BASELINE.json's config 5
 * partitions the matrix in row blocks over the 8 GPUs of a node, broadcasts the dense vector once and concatenates
 * the output slices with an all-gather.  This header is that layer for a C / C++ caller -- the world of the
 * reference's tester (src/tester.cpp:47-72 calls plain C++ functions) -- on top of spmv_hip.h: one spmv_csr_t per
 * row block, one spmv_dist_t per rank, RCCL underneath (ncclAllGather over the xGMI mesh; with unequal blocks one
 * grouped ncclBroadcast per rank).  The Python plumbing of bench.py (spmv-test_amd/dist.py, torch.distributed)
 * does the same for the benchmark harness.
 *
 * Two process models, as SURVEY section 8e lists them:
 *   one process per GPU     rank 0 calls spmv_dist_get_unique_id and hands the 128 bytes to the other processes
 *                           (a file, a pipe, MPI, a socket -- the caller's business); every rank calls spmv_dist_init
 *                           under its own current device;
 *   one process, all GPUs   spmv_dist_init_all creates the ranks of all listed devices at once (ncclCommInitAll);
 *                           the caller then drives every rank from one thread, setting the device before each call and
 *                           bracketing the calls of one step with spmv_dist_group_start / _end.
 * Every call returns 0 or a negative spmv_status; spmv_dist_last_error() has the message (thread-local).
 */
#ifndef SPMV_DIST_H
#define SPMV_DIST_H

#include "spmv_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spmv_dist spmv_dist_t;

#define SPMV_DIST_ID_BYTES 128   /* = NCCL_UNIQUE_ID_BYTES */

SPMV_API const char *spmv_dist_last_error(void);
SPMV_API int spmv_dist_get_unique_id(void *id128);
SPMV_API int spmv_dist_init(int world, int rank, const void *id128, spmv_dist_t **out);
SPMV_API int spmv_dist_init_all(int ndev, const int *devices, spmv_dist_t **out /* ndev handles */);
SPMV_API int spmv_dist_group_start(void);
SPMV_API int spmv_dist_group_end(void);
SPMV_API int spmv_dist_rank(const spmv_dist_t *d, int *world, int *rank, int *device);

/* Row blocks: rank r owns global rows [row_bounds[r], row_bounds[r+1]); every rank holds all `cols` entries of x.
 * (spmv-test_amd/partition.py::balanced_row_bounds cuts by nonzero count; equal blocks use one in-place all-gather.) */
SPMV_API int spmv_dist_set_partition(spmv_dist_t *d, const int64_t *row_bounds /* world+1 */, int64_t cols);

/* The one-off distribution of the dense vector (north_star: "the dense vector broadcast once"). */
SPMV_API int spmv_dist_broadcast_x(spmv_dist_t *d, float *d_x, int root, void *stream);

/* Make this rank's plan of `variant` the one rank `root` made (spmv_csr_plan_get there, a broadcast of the eight
 * numbers, spmv_csr_plan_set here): all row blocks then cut their chunks alike.  Collective, and it waits on the host
 * for the broadcast: for the one-process-per-GPU model.  With all ranks in one process carry the numbers with
 * spmv_csr_plan_get / spmv_csr_plan_set directly. */
SPMV_API int spmv_dist_plan_like_root(spmv_dist_t *d, spmv_csr_t *shard, int variant, int root, void *stream);

/* One step: y_full[row_bounds[rank] ...] = shard * x on `stream`, then the slices of all ranks are concatenated into
 * d_y_full (rows_total floats) on every rank.  Asynchronous on `stream`; collective. */
SPMV_API int spmv_dist_spmv(spmv_dist_t *d, spmv_csr_t *shard, int variant, const float *d_x, float *d_y_full, void *stream);

/* The exchange alone (slices already in place). */
SPMV_API int spmv_dist_allgather_y(spmv_dist_t *d, float *d_y_full, void *stream);

SPMV_API int spmv_dist_destroy(spmv_dist_t *d);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_DIST_H */
