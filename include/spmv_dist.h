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

/* ---- the pipelined step (round 3): block-cyclic row blocks, the exchange of one block group under the multiply of the next
 *
 * The global row space is cut into S * world equal blocks of sub_rows rows; rank p owns blocks s*world + p, s = 0..S-1.
 * Group s -- one block per rank -- is contiguous in the natural row order, so its concatenation lands in y_full in place:
 * y_full[(s*world + p) * sub_rows ...] is rank p's slot of group s.  spmv_dist_pipe_step enqueues, for s = 0..S-1, the
 * product of the rank's block s on `stream` and -- on a side stream of the pipe that waits for that product only -- the
 * exchange of group s, which therefore runs while the CUs multiply block s+1.  Steps pipeline across calls: the product
 * of block s waits only for the previous step's exchange of group s (it overwrites the slot that exchange read), so the
 * exchange never drains inside a run of steps; spmv_dist_pipe_finish orders `stream` behind everything outstanding.
 * (What bench.py --gpus N does with torch.distributed, spmv-test_amd/dist.py:PipelinedSpmv, for a C / C++ caller.)
 *
 * Exchange of a group:
 *   SPMV_DIST_ALLGATHER   one in-place ncclAllGather (RCCL's collective: rings / trees over the xGMI mesh)
 *   SPMV_DIST_P2P         one grouped ncclSend + ncclRecv pair per peer: every slice travels its owner's DIRECT xGMI link
 *                         to each peer, all seven links of a GPU busy at once (the all-pairs schedule of SURVEY section 5)
 *   SPMV_DIST_PEER_STORE  no RCCL: the owner copies its slice into every peer's y_full (hipMemcpyPeerAsync over xGMI; the
 *                         "direct peer-store" of SURVEY section 8e "Overlap").  One-process model only: the ranks must
 *                         be linked with spmv_dist_pipe_link, which hands every pipe its peers' buffers and events. */
typedef struct spmv_dist_pipe spmv_dist_pipe_t;
enum spmv_dist_exchange { SPMV_DIST_ALLGATHER = 0, SPMV_DIST_P2P = 1, SPMV_DIST_PEER_STORE = 2 };

SPMV_API int spmv_dist_pipe_create(spmv_dist_t *d, int S, int64_t sub_rows, int64_t cols, int exchange, spmv_dist_pipe_t **out);
/* PEER_STORE: all pipes of the job (index = rank), each with the y_full it will be stepped with.  Peer access is enabled
 * between the devices that differ; ranks may share a device (rehearsal on one GPU: the copies are then device-local). */
SPMV_API int spmv_dist_pipe_link(spmv_dist_pipe_t *const *pipes, float *const *d_y_full, int n);
/* blocks[s] = the handle of this rank's s-th block (rows [(s*world + rank) * sub_rows, +sub_rows), all `cols` columns). */
SPMV_API int spmv_dist_pipe_step(spmv_dist_pipe_t *p, spmv_csr_t *const *blocks, int variant, const float *d_x, float *d_y_full,
                                 void *stream);
/* The S exchanges of one step without the products (what the exchange alone costs). */
SPMV_API int spmv_dist_pipe_exchange_only(spmv_dist_pipe_t *p, float *d_y_full, void *stream);
/* Footprint exchange (optional; the default concatenates all of y on every rank, which is what the north-star fixes).
 * A rank's blocks reference only some columns of x; where y becomes the next x (an iteration), rank q needs y only on its
 * own column footprint -- for a band of 8192 columns that is each of its blocks' rows plus 4096 on either side instead of
 * the other ranks' 7 x 64 MiB.  A rank's footprint is given as `per_rank` intervals (one per block: the blocks of a rank
 * lie world blocks apart, one interval around all of them would span nearly everything): rank q needs rows
 * [need_lo[q*per_rank + k], need_hi[q*per_rank + k]), k = 0..per_rank-1.  After this call the exchanges of SPMV_DIST_P2P and
 * SPMV_DIST_PEER_STORE pipes move, from every owner to every peer, only the rows of the group's slot that fall inside one of
 * the PEER's intervals (nothing where they do not meet); y_full is then complete on a rank inside its footprint and on its
 * own rows only.  The arrays must be the same on every rank (spmv_csr_column_range of every block, all-gathered by the
 * caller).  Not for SPMV_DIST_ALLGATHER. */
SPMV_API int spmv_dist_pipe_set_footprint(spmv_dist_pipe_t *p, const int64_t *need_lo, const int64_t *need_hi, int per_rank);
SPMV_API int spmv_dist_pipe_finish(spmv_dist_pipe_t *p, void *stream);
/* SPMV_DIST_PEER_STORE only (a no-op for the RCCL exchanges, whose receives a rank enqueues itself behind its own work):
 * the OTHER ranks write this rank's y_full, so the readers of one step's y_full -- a norm, the copy that makes y the next x,
 * enqueued on `stream` after spmv_dist_pipe_finish -- have to be ordered before the peers' stores of the next step.
 * spmv_dist_pipe_release marks that point of `stream`; every store a peer enqueues into this rank's y_full AFTER the call
 * waits for it.  spmv_dist_pipe_step sets the mark itself at its start, which covers callers that step the ranks
 * concurrently (one host thread per rank); a caller that steps the ranks of a process one after the other calls
 * spmv_dist_pipe_release on EVERY rank (readers enqueued) before the first rank's next step -- or double-buffers y_full. */
SPMV_API int spmv_dist_pipe_release(spmv_dist_pipe_t *p, void *stream);
SPMV_API int spmv_dist_pipe_destroy(spmv_dist_pipe_t *p);

/* Ranks WITHOUT an RCCL communicator, for SPMV_DIST_PEER_STORE only: one process drives them all; `devices` may name one
 * device several times (the whole pipeline -- block-cyclic blocks, plan hand-over, peer stores, event ordering -- then
 * runs on a single GPU, which is how the tests exercise world > 1 on a one-GPU box). */
SPMV_API int spmv_dist_init_local(int nranks, const int *devices, spmv_dist_t **out /* nranks handles */);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_DIST_H */
