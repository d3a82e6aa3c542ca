// dist_selftest.cpp -- bin/spmv_dist_selftest: the sharded SpMV of include/spmv_dist.h, all ranks in ONE process
// (spmv_dist_init_all = ncclCommInitAll, the first process model of SURVEY.md section 8e), checked against the whole
// matrix multiplied by a single handle.  Also the worked example of INTEGRATION.md for a C++ caller.
//
//   spmv_dist_selftest [--ranks N] [--rows-per-rank R] [--band B] [--unequal] [--variant tiled|adaptive|scalar|auto] [--steps K]
//                      [--pipeline S] [--exchange allgather|p2p|peer] [--local] [--footprint] [--no-verify] [--vary-x]
// N defaults to the number of visible GPUs.  Matrix: (N*R)^2, 16 nonzeros per row (config 2's law), generated on
// the devices by spmv_synth_fill; with --unequal the blocks hold R-17, R+17, ... rows (the all-gather-v path).
// --pipeline S (round 3): the pipelined step of spmv_dist.h -- every rank's R rows as S block-cyclic blocks, the exchange
// of block group s on a side stream under the multiply of block s+1 -- with RCCL's all-gather, a grouped send/recv pair per
// peer, or peer stores (hipMemcpyPeerAsync, no RCCL).  --local: ranks without a communicator (spmv_dist_init_local), as
// many as asked on the visible devices round-robin -- the whole pipeline with world > 1 on a ONE-GPU box (peer stores only).
// --footprint: the optional footprint exchange -- every rank receives only the rows of y its own columns reference.
// --vary-x (with --pipeline, verified runs): four more steps with x doubled from step to step, a slow reader of y_full
// on every rank's stream between the steps (a copy behind a large fill) and spmv_dist_pipe_release before the next step:
// every snapshot must equal 2^t times the single-handle y, bit for bit -- the peer stores of step t+1 may not overtake the
// readers of step t.
// --no-verify (with --pipeline): timing only, no single-handle reference -- the whole matrix of 8 x 16Mi rows has 2^31
// nonzeros, one more than a handle takes (bench.py --exchange all times the peer-store pipeline at that size).
// Prints one JSON line; exit code 0 iff every rank's y is bit-identical to the single-handle result.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "spmv_dist.h"

#define OK_HIP(call)                                                                                  \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) { fprintf(stderr, "HIP error %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(EXIT_FAILURE); } \
    } while (0)
#define OK_SPMV(call)                                                                                 \
    do {                                                                                              \
        int rc__ = (call);                                                                            \
        if (rc__ != SPMV_OK) { fprintf(stderr, "error %s:%d: %s | %s\n", __FILE__, __LINE__, spmv_last_error(), spmv_dist_last_error()); exit(EXIT_FAILURE); } \
    } while (0)

struct Rank {
    int device = 0;
    int64_t r0 = 0, r1 = 0;
    int32_t *d_rp = nullptr, *d_ci = nullptr;
    float *d_va = nullptr, *d_x = nullptr, *d_y = nullptr;
    spmv_csr_t *A = nullptr;
    hipStream_t stream = nullptr;
    // --pipeline: S blocks of sub rows each (block-cyclic: block s of rank p = global block s*world + p)
    std::vector<int32_t *> b_rp, b_ci;
    std::vector<float *> b_va;
    std::vector<spmv_csr_t *> blocks;
    spmv_dist_pipe_t *pipe = nullptr;
};

static int run_pipeline(int ndev, int64_t per, int64_t band, int variant, int steps, int S, int exchange, bool local, bool footprint, bool verify, bool vary_x);

int main(int argc, char **argv)
{
    int ndev = spmv_device_count();
    int64_t per = 1 << 18, band = 4096;
    bool unequal = false, local = false, footprint = false, verify = true, vary_x = false;
    int variant = SPMV_TILED, steps = 20, pipeline = 0, exchange = SPMV_DIST_ALLGATHER;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--ranks") ndev = atoi(next());
        else if (a == "--rows-per-rank") per = atoll(next());
        else if (a == "--band") band = atoll(next());
        else if (a == "--steps") steps = atoi(next());
        else if (a == "--unequal") unequal = true;
        else if (a == "--local") local = true;
        else if (a == "--footprint") footprint = true;
        else if (a == "--no-verify") verify = false;
        else if (a == "--vary-x") vary_x = true;
        else if (a == "--pipeline") pipeline = atoi(next());
        else if (a == "--exchange") {
            std::string v = next();
            if (v == "allgather") exchange = SPMV_DIST_ALLGATHER;
            else if (v == "p2p") exchange = SPMV_DIST_P2P;
            else if (v == "peer") exchange = SPMV_DIST_PEER_STORE;
            else { fprintf(stderr, "--exchange allgather|p2p|peer\n"); return 2; }
        }
        else if (a == "--variant") {
            std::string v = next();
            variant = v == "adaptive" ? SPMV_ADAPTIVE : v == "scalar" ? SPMV_SCALAR : v == "auto" ? SPMV_AUTO : SPMV_TILED;
        } else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (ndev < 1 || (!local && ndev > spmv_device_count())) { fprintf(stderr, "HIP error: %d ranks asked, %d devices visible\n", ndev, spmv_device_count()); return EXIT_FAILURE; }
    if (local && spmv_device_count() < 1) { fprintf(stderr, "HIP error: no device visible\n"); return EXIT_FAILURE; }
    if (pipeline > 0) return run_pipeline(ndev, per, band, variant, steps, pipeline, exchange, local, footprint, verify, vary_x);
    if (local) { fprintf(stderr, "--local needs --pipeline S --exchange peer\n"); return 2; }
    const uint64_t seed = 20251031;
    const int64_t rows = per * ndev, cols = rows;
    const int nnz_row = 16;

    std::vector<int64_t> bounds(ndev + 1, 0);
    for (int r = 0; r < ndev; ++r) {
        int64_t n = per;
        if (unequal && ndev > 1) n += (r % 2 ? 17 : -17) * (r + 1 < ndev || ndev % 2 == 0 ? 1 : 0);
        bounds[r + 1] = bounds[r] + n;
    }
    bounds[ndev] = rows;

    std::vector<spmv_dist_t *> dist(ndev, nullptr);
    OK_SPMV(spmv_dist_init_all(ndev, nullptr, dist.data()));
    std::vector<Rank> rk(ndev);
    for (int r = 0; r < ndev; ++r) {
        Rank &R = rk[r];
        OK_SPMV(spmv_dist_rank(dist[r], nullptr, nullptr, &R.device));
        OK_HIP(hipSetDevice(R.device));
        OK_HIP(hipStreamCreate(&R.stream));
        OK_SPMV(spmv_dist_set_partition(dist[r], bounds.data(), cols));
        R.r0 = bounds[r]; R.r1 = bounds[r + 1];
        const int64_t n = R.r1 - R.r0, nnz = n * nnz_row;
        std::vector<int32_t> rp(n + 1);
        for (int64_t i = 0; i <= n; ++i) rp[i] = (int32_t)(i * nnz_row);
        OK_HIP(hipMalloc((void **)&R.d_rp, sizeof(int32_t) * (n + 1)));
        OK_HIP(hipMalloc((void **)&R.d_ci, sizeof(int32_t) * (nnz ? nnz : 1)));
        OK_HIP(hipMalloc((void **)&R.d_va, sizeof(float) * (nnz ? nnz : 1)));
        OK_HIP(hipMalloc((void **)&R.d_x, sizeof(float) * cols));
        OK_HIP(hipMalloc((void **)&R.d_y, sizeof(float) * rows));
        OK_HIP(hipMemcpy(R.d_rp, rp.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
        OK_HIP(hipMemset(R.d_y, 0xff, sizeof(float) * rows));                       // NaNs: every row must be overwritten
        OK_SPMV(spmv_synth_fill(seed, R.r0, n, rows, cols, band, R.d_rp, R.d_ci, R.d_va, R.stream));
        OK_SPMV(spmv_csr_create_device(n, cols, nnz, R.d_rp, R.d_ci, R.d_va, &R.A));
        if (r == 0) OK_SPMV(spmv_synth_x(seed, 0, cols, R.d_x, R.stream));
        OK_HIP(hipStreamSynchronize(R.stream));
    }
    // the dense vector, broadcast once
    OK_SPMV(spmv_dist_group_start());
    for (int r = 0; r < ndev; ++r) { OK_HIP(hipSetDevice(rk[r].device)); OK_SPMV(spmv_dist_broadcast_x(dist[r], rk[r].d_x, 0, rk[r].stream)); }
    OK_SPMV(spmv_dist_group_end());
    // every block planned like block 0 (all ranks live in this process: the numbers travel by hand)
    int32_t params[8];
    OK_HIP(hipSetDevice(rk[0].device));
    OK_SPMV(spmv_csr_plan(rk[0].A, variant, rk[0].stream));
    OK_SPMV(spmv_csr_plan_get(rk[0].A, variant, params));
    for (int r = 1; r < ndev; ++r) { OK_HIP(hipSetDevice(rk[r].device)); OK_SPMV(spmv_csr_plan_set(rk[r].A, variant, params, rk[r].stream)); }

    auto step = [&]() {
        OK_SPMV(spmv_dist_group_start());
        for (int r = 0; r < ndev; ++r) {
            OK_HIP(hipSetDevice(rk[r].device));
            OK_SPMV(spmv_dist_spmv(dist[r], rk[r].A, variant, rk[r].d_x, rk[r].d_y, rk[r].stream));
        }
        OK_SPMV(spmv_dist_group_end());
    };
    auto sync_all = [&]() { for (int r = 0; r < ndev; ++r) { OK_HIP(hipSetDevice(rk[r].device)); OK_HIP(hipStreamSynchronize(rk[r].stream)); } };
    step();
    sync_all();

    // the whole matrix through ONE handle on device 0, planned alike
    OK_HIP(hipSetDevice(rk[0].device));
    const int64_t nnz_all = rows * nnz_row;
    std::vector<int32_t> rp_all(rows + 1);
    for (int64_t i = 0; i <= rows; ++i) rp_all[i] = (int32_t)(i * nnz_row);
    int32_t *d_rp, *d_ci;
    float *d_va, *d_yref;
    OK_HIP(hipMalloc((void **)&d_rp, sizeof(int32_t) * (rows + 1)));
    OK_HIP(hipMalloc((void **)&d_ci, sizeof(int32_t) * nnz_all));
    OK_HIP(hipMalloc((void **)&d_va, sizeof(float) * nnz_all));
    OK_HIP(hipMalloc((void **)&d_yref, sizeof(float) * rows));
    OK_HIP(hipMemcpy(d_rp, rp_all.data(), sizeof(int32_t) * (rows + 1), hipMemcpyHostToDevice));
    OK_SPMV(spmv_synth_fill(seed, 0, rows, rows, cols, band, d_rp, d_ci, d_va, rk[0].stream));
    spmv_csr_t *whole = nullptr;
    OK_SPMV(spmv_csr_create_device(rows, cols, nnz_all, d_rp, d_ci, d_va, &whole));
    OK_SPMV(spmv_csr_plan_set(whole, variant, params, rk[0].stream));
    OK_SPMV(spmv_csr_run(whole, variant, rk[0].d_x, d_yref, rk[0].stream));
    OK_HIP(hipStreamSynchronize(rk[0].stream));
    std::vector<float> yref(rows), y(rows);
    OK_HIP(hipMemcpy(yref.data(), d_yref, sizeof(float) * rows, hipMemcpyDeviceToHost));
    int64_t differing = 0;
    for (int r = 0; r < ndev; ++r) {
        OK_HIP(hipSetDevice(rk[r].device));
        OK_HIP(hipMemcpy(y.data(), rk[r].d_y, sizeof(float) * rows, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < rows; ++i) differing += std::memcmp(&y[i], &yref[i], sizeof(float)) != 0;
    }

    // timing: K steps (multiply + exchange), multiply alone, exchange alone -- events on rank 0's stream
    hipEvent_t e0, e1;
    OK_HIP(hipSetDevice(rk[0].device));
    OK_HIP(hipEventCreate(&e0));
    OK_HIP(hipEventCreate(&e1));
    auto timed = [&](auto &&body) {
        sync_all();
        OK_HIP(hipSetDevice(rk[0].device));
        OK_HIP(hipEventRecord(e0, rk[0].stream));
        for (int k = 0; k < steps; ++k) body();
        OK_HIP(hipSetDevice(rk[0].device));
        OK_HIP(hipEventRecord(e1, rk[0].stream));
        sync_all();
        float ms = 0;
        OK_HIP(hipEventElapsedTime(&ms, e0, e1));
        return ms / steps;
    };
    const float step_ms = timed(step);
    const float mult_ms = timed([&]() {
        for (int r = 0; r < ndev; ++r) {
            OK_HIP(hipSetDevice(rk[r].device));
            OK_SPMV(spmv_csr_run(rk[r].A, variant, rk[r].d_x, rk[r].d_y + rk[r].r0, rk[r].stream));
        }
    });
    const float xchg_ms = timed([&]() {
        OK_SPMV(spmv_dist_group_start());
        for (int r = 0; r < ndev; ++r) { OK_HIP(hipSetDevice(rk[r].device)); OK_SPMV(spmv_dist_allgather_y(dist[r], rk[r].d_y, rk[r].stream)); }
        OK_SPMV(spmv_dist_group_end());
    });
    char plan[256];
    OK_HIP(hipSetDevice(rk[0].device));
    OK_SPMV(spmv_csr_plan_describe(rk[0].A, variant, plan, sizeof plan));
    const double bytes = 8.0 * nnz_all + 4.0 * (rows + ndev) + 4.0 * rows + 4.0 * cols * ndev;
    printf("{\"world\": %d, \"process_model\": \"one process, ncclCommInitAll\", \"rows\": %lld, \"nnz\": %lld, \"band\": %lld, "
           "\"unequal_blocks\": %s, \"variant\": \"%s\", \"rows_differing_from_single_handle\": %lld, \"step_ms\": %.5f, "
           "\"multiply_only_ms\": %.5f, \"exchange_only_ms\": %.5f, \"aggregate_GBs\": %.1f, \"plan\": \"%s\"}\n",
           ndev, (long long)rows, (long long)nnz_all, (long long)band, unequal ? "true" : "false", spmv_variant_name(variant),
           (long long)differing, step_ms, mult_ms, xchg_ms, bytes / (step_ms * 1e-3) / 1e9, plan);
    for (int r = 0; r < ndev; ++r) {
        OK_HIP(hipSetDevice(rk[r].device));
        (void)spmv_csr_destroy(rk[r].A);
        (void)spmv_dist_destroy(dist[r]);
    }
    (void)spmv_csr_destroy(whole);
    return differing == 0 ? 0 : 1;
}


// ---- the pipelined step: S block-cyclic blocks per rank, exchange of group s under the multiply of block s+1 ------------
static int run_pipeline(int ndev, int64_t per, int64_t band, int variant, int steps, int S, int exchange, bool local, bool footprint, bool verify, bool vary_x)
{
    if (footprint && exchange == SPMV_DIST_ALLGATHER) { fprintf(stderr, "--footprint needs --exchange p2p or peer\n"); return 2; }
    if (S < 1 || per % S) { fprintf(stderr, "--rows-per-rank must be a multiple of --pipeline\n"); return 2; }
    if (local && exchange != SPMV_DIST_PEER_STORE) { fprintf(stderr, "--local ranks exchange by peer stores only (--exchange peer)\n"); return 2; }
    const uint64_t seed = 20251031;
    const int64_t sub = per / S, rows = per * ndev, cols = rows;
    const int nnz_row = 16;
    std::vector<spmv_dist_t *> dist(ndev, nullptr);
    if (local) OK_SPMV(spmv_dist_init_local(ndev, nullptr, dist.data()));
    else OK_SPMV(spmv_dist_init_all(ndev, nullptr, dist.data()));
    std::vector<Rank> rk(ndev);
    std::vector<int32_t> rp(sub + 1);
    for (int64_t i = 0; i <= sub; ++i) rp[i] = (int32_t)(i * nnz_row);
    for (int r = 0; r < ndev; ++r) {
        Rank &R = rk[r];
        OK_SPMV(spmv_dist_rank(dist[r], nullptr, nullptr, &R.device));
        OK_HIP(hipSetDevice(R.device));
        OK_HIP(hipStreamCreate(&R.stream));
        OK_HIP(hipMalloc((void **)&R.d_x, sizeof(float) * cols));
        OK_HIP(hipMalloc((void **)&R.d_y, sizeof(float) * rows));
        OK_HIP(hipMemset(R.d_y, 0xff, sizeof(float) * rows));                       // NaNs: every row must be overwritten
        R.b_rp.assign(S, nullptr); R.b_ci.assign(S, nullptr); R.b_va.assign(S, nullptr); R.blocks.assign(S, nullptr);
        for (int s = 0; s < S; ++s) {
            const int64_t row0 = ((int64_t)s * ndev + r) * sub, nnz = sub * nnz_row;
            OK_HIP(hipMalloc((void **)&R.b_rp[s], sizeof(int32_t) * (sub + 1)));
            OK_HIP(hipMalloc((void **)&R.b_ci[s], sizeof(int32_t) * (nnz ? nnz : 1)));
            OK_HIP(hipMalloc((void **)&R.b_va[s], sizeof(float) * (nnz ? nnz : 1)));
            OK_HIP(hipMemcpy(R.b_rp[s], rp.data(), sizeof(int32_t) * (sub + 1), hipMemcpyHostToDevice));
            OK_SPMV(spmv_synth_fill(seed, row0, sub, rows, cols, band, R.b_rp[s], R.b_ci[s], R.b_va[s], R.stream));
            OK_SPMV(spmv_csr_create_device(sub, cols, nnz, R.b_rp[s], R.b_ci[s], R.b_va[s], &R.blocks[s]));
        }
        if (r == 0) OK_SPMV(spmv_synth_x(seed, 0, cols, R.d_x, R.stream));
        OK_HIP(hipStreamSynchronize(R.stream));
        OK_SPMV(spmv_dist_pipe_create(dist[r], S, sub, cols, exchange, &R.pipe));
    }
    // the dense vector, distributed once: RCCL broadcast, or (local ranks) a copy from rank 0
    if (local) {
        for (int r = 1; r < ndev; ++r) {
            OK_HIP(hipSetDevice(rk[r].device));
            OK_HIP(hipMemcpyPeer(rk[r].d_x, rk[r].device, rk[0].d_x, rk[0].device, sizeof(float) * cols));
        }
    } else {
        OK_SPMV(spmv_dist_group_start());
        for (int r = 0; r < ndev; ++r) {
            OK_HIP(hipSetDevice(rk[r].device));
            int64_t bounds_dummy[2] = {0, 0};
            (void)bounds_dummy;
            std::vector<int64_t> b(ndev + 1);
            for (int q = 0; q <= ndev; ++q) b[q] = per * q;
            OK_SPMV(spmv_dist_set_partition(dist[r], b.data(), cols));
            OK_SPMV(spmv_dist_broadcast_x(dist[r], rk[r].d_x, 0, rk[r].stream));
        }
        OK_SPMV(spmv_dist_group_end());
    }
    if (exchange == SPMV_DIST_PEER_STORE) {
        std::vector<spmv_dist_pipe_t *> pipes(ndev);
        std::vector<float *> ys(ndev);
        for (int r = 0; r < ndev; ++r) { pipes[r] = rk[r].pipe; ys[r] = rk[r].d_y; }
        OK_SPMV(spmv_dist_pipe_link(pipes.data(), ys.data(), ndev));
    }
    // --footprint: every rank's column footprint (spmv_csr_column_range over its blocks), known to all, set on every pipe:
    // the exchanges then move only what the receiving rank's columns reference
    std::vector<int64_t> need_lo((size_t)ndev * S, 0), need_hi((size_t)ndev * S, rows);
    if (footprint) {
        for (int r = 0; r < ndev; ++r) {
            OK_HIP(hipSetDevice(rk[r].device));
            for (int s = 0; s < S; ++s) {       // one interval per block: a rank's blocks lie world blocks apart
                int64_t a, b;
                OK_SPMV(spmv_csr_column_range(rk[r].blocks[s], &a, &b, rk[r].stream));
                need_lo[(size_t)r * S + s] = b < 0 ? 0 : a;
                need_hi[(size_t)r * S + s] = b < 0 ? 0 : b + 1;
            }
        }
        for (int r = 0; r < ndev; ++r) OK_SPMV(spmv_dist_pipe_set_footprint(rk[r].pipe, need_lo.data(), need_hi.data(), S));
    }
    // every block planned like rank 0's first
    int32_t params[8];
    OK_HIP(hipSetDevice(rk[0].device));
    OK_SPMV(spmv_csr_plan(rk[0].blocks[0], variant, rk[0].stream));
    OK_SPMV(spmv_csr_plan_get(rk[0].blocks[0], variant, params));
    for (int r = 0; r < ndev; ++r)
        for (int s = 0; s < S; ++s) {
            if (r == 0 && s == 0) continue;
            OK_HIP(hipSetDevice(rk[r].device));
            OK_SPMV(spmv_csr_plan_set(rk[r].blocks[s], variant, params, rk[r].stream));
        }
    const bool rccl = exchange != SPMV_DIST_PEER_STORE;
    auto step = [&]() {
        if (rccl) OK_SPMV(spmv_dist_group_start());
        for (int r = 0; r < ndev; ++r) {
            OK_HIP(hipSetDevice(rk[r].device));
            OK_SPMV(spmv_dist_pipe_step(rk[r].pipe, rk[r].blocks.data(), variant, rk[r].d_x, rk[r].d_y, rk[r].stream));
        }
        if (rccl) OK_SPMV(spmv_dist_group_end());
    };
    auto finish_all = [&]() {
        for (int r = 0; r < ndev; ++r) { OK_HIP(hipSetDevice(rk[r].device)); OK_SPMV(spmv_dist_pipe_finish(rk[r].pipe, rk[r].stream)); }
        for (int r = 0; r < ndev; ++r) { OK_HIP(hipSetDevice(rk[r].device)); OK_HIP(hipStreamSynchronize(rk[r].stream)); }
    };
    step(); step(); step();      // three in a row: the cross-step ordering (product s waits for the last exchange of group s)
    finish_all();

    // the whole matrix through ONE handle on device 0, planned alike (--no-verify: timing only)
    OK_HIP(hipSetDevice(rk[0].device));
    const int64_t nnz_all = rows * nnz_row;
    spmv_csr_t *whole = nullptr;
    int64_t differing = 0, untouched = 0;
    if (verify) {
        std::vector<int32_t> rp_all(rows + 1);
        for (int64_t i = 0; i <= rows; ++i) rp_all[i] = (int32_t)(i * nnz_row);
        int32_t *d_rp, *d_ci;
        float *d_va, *d_yref;
        OK_HIP(hipMalloc((void **)&d_rp, sizeof(int32_t) * (rows + 1)));
        OK_HIP(hipMalloc((void **)&d_ci, sizeof(int32_t) * nnz_all));
        OK_HIP(hipMalloc((void **)&d_va, sizeof(float) * nnz_all));
        OK_HIP(hipMalloc((void **)&d_yref, sizeof(float) * rows));
        OK_HIP(hipMemcpy(d_rp, rp_all.data(), sizeof(int32_t) * (rows + 1), hipMemcpyHostToDevice));
        OK_SPMV(spmv_synth_fill(seed, 0, rows, rows, cols, band, d_rp, d_ci, d_va, rk[0].stream));
        OK_SPMV(spmv_csr_create_device(rows, cols, nnz_all, d_rp, d_ci, d_va, &whole));
        OK_SPMV(spmv_csr_plan_set(whole, variant, params, rk[0].stream));
        OK_SPMV(spmv_csr_run(whole, variant, rk[0].d_x, d_yref, rk[0].stream));
        OK_HIP(hipStreamSynchronize(rk[0].stream));
        std::vector<float> yref(rows), y(rows);
        OK_HIP(hipMemcpy(yref.data(), d_yref, sizeof(float) * rows, hipMemcpyDeviceToHost));
        for (int r = 0; r < ndev; ++r) {
            OK_HIP(hipSetDevice(rk[r].device));
            OK_HIP(hipMemcpy(y.data(), rk[r].d_y, sizeof(float) * rows, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < rows; ++i) {
                // with a footprint a rank holds y on its own rows and inside its footprint; the rest keeps the NaN it was preset to
                const bool own = (i / sub) % ndev == r;
                bool needed = !footprint || own;
                for (int s = 0; s < S && !needed; ++s) needed = i >= need_lo[(size_t)r * S + s] && i < need_hi[(size_t)r * S + s];
                if (needed) differing += std::memcmp(&y[i], &yref[i], sizeof(float)) != 0;
                else untouched += y[i] != y[i];
            }
        }
    }

    // --vary-x: x doubles from step to step, y_full is read (slowly) between the steps; the stores of the next step wait
    int64_t vary_bad = -1;
    if (vary_x && verify && !footprint) {
        vary_bad = 0;
        constexpr int T = 4;
        std::vector<float> x0(cols), yref_h(rows), snap_h(rows);
        std::vector<std::vector<float>> xt(T, std::vector<float>(cols));     // one host buffer per step: the uploads are asynchronous
        OK_HIP(hipSetDevice(rk[0].device));
        OK_HIP(hipMemcpy(x0.data(), rk[0].d_x, sizeof(float) * cols, hipMemcpyDeviceToHost));
        {   // the single-handle y for x0 (the verify block above left it on the host only inside its scope: recompute)
            float *d_yr = nullptr;
            OK_HIP(hipMalloc((void **)&d_yr, sizeof(float) * rows));
            OK_SPMV(spmv_csr_run(whole, variant, rk[0].d_x, d_yr, rk[0].stream));
            OK_HIP(hipStreamSynchronize(rk[0].stream));
            OK_HIP(hipMemcpy(yref_h.data(), d_yr, sizeof(float) * rows, hipMemcpyDeviceToHost));
            OK_HIP(hipFree(d_yr));
        }
        const size_t ballast = (size_t)256 << 20;
        std::vector<float *> snaps((size_t)ndev * T, nullptr);
        std::vector<void *> fill(ndev, nullptr);
        for (int r = 0; r < ndev; ++r) {
            OK_HIP(hipSetDevice(rk[r].device));
            OK_HIP(hipMalloc(&fill[r], ballast));
            for (int t = 0; t < T; ++t) OK_HIP(hipMalloc((void **)&snaps[(size_t)r * T + t], sizeof(float) * rows));
        }
        finish_all();
        for (int t = 0; t < T; ++t) {
            const float scale = (float)(1 << t);
            for (int64_t i = 0; i < cols; ++i) xt[t][i] = x0[i] * scale;
            for (int r = 0; r < ndev; ++r) {
                OK_HIP(hipSetDevice(rk[r].device));
                OK_HIP(hipMemcpyAsync(rk[r].d_x, xt[t].data(), sizeof(float) * cols, hipMemcpyHostToDevice, rk[r].stream));
            }
            step();
            for (int r = 0; r < ndev; ++r) { OK_HIP(hipSetDevice(rk[r].device)); OK_SPMV(spmv_dist_pipe_finish(rk[r].pipe, rk[r].stream)); }
            for (int r = 0; r < ndev; ++r) {          // the slow reader of this step's y_full, then the release mark
                OK_HIP(hipSetDevice(rk[r].device));
                OK_HIP(hipMemsetAsync(fill[r], t, ballast, rk[r].stream));
                OK_HIP(hipMemcpyAsync(snaps[(size_t)r * T + t], rk[r].d_y, sizeof(float) * rows, hipMemcpyDeviceToDevice, rk[r].stream));
                OK_SPMV(spmv_dist_pipe_release(rk[r].pipe, rk[r].stream));
            }
        }
        finish_all();
        for (int r = 0; r < ndev; ++r) {
            OK_HIP(hipSetDevice(rk[r].device));
            for (int t = 0; t < T; ++t) {
                OK_HIP(hipMemcpy(snap_h.data(), snaps[(size_t)r * T + t], sizeof(float) * rows, hipMemcpyDeviceToHost));
                const float scale = (float)(1 << t);
                for (int64_t i = 0; i < rows; ++i) {
                    const float want = yref_h[i] * scale;
                    vary_bad += std::memcmp(&snap_h[i], &want, sizeof(float)) != 0;
                }
                OK_HIP(hipFree(snaps[(size_t)r * T + t]));
            }
            OK_HIP(hipFree(fill[r]));
            OK_HIP(hipMemcpy(rk[r].d_x, x0.data(), sizeof(float) * cols, hipMemcpyHostToDevice));
        }
        differing += vary_bad;
    }

    hipEvent_t e0, e1;
    OK_HIP(hipSetDevice(rk[0].device));
    OK_HIP(hipEventCreate(&e0));
    OK_HIP(hipEventCreate(&e1));
    auto timed = [&](auto &&body) {
        finish_all();
        OK_HIP(hipSetDevice(rk[0].device));
        OK_HIP(hipEventRecord(e0, rk[0].stream));
        for (int k = 0; k < steps; ++k) body();
        for (int r = 0; r < ndev; ++r) { OK_HIP(hipSetDevice(rk[r].device)); OK_SPMV(spmv_dist_pipe_finish(rk[r].pipe, rk[r].stream)); }
        OK_HIP(hipSetDevice(rk[0].device));
        OK_HIP(hipEventRecord(e1, rk[0].stream));
        for (int r = 0; r < ndev; ++r) { OK_HIP(hipSetDevice(rk[r].device)); OK_HIP(hipStreamSynchronize(rk[r].stream)); }
        float ms = 0;
        OK_HIP(hipEventElapsedTime(&ms, e0, e1));
        return ms / steps;
    };
    const float step_ms = timed(step);
    const float mult_ms = timed([&]() {
        for (int r = 0; r < ndev; ++r) {
            OK_HIP(hipSetDevice(rk[r].device));
            for (int s = 0; s < S; ++s)
                OK_SPMV(spmv_csr_run(rk[r].blocks[s], variant, rk[r].d_x, rk[r].d_y + ((int64_t)s * ndev + r) * sub, rk[r].stream));
        }
    });
    const float xchg_ms = timed([&]() {
        if (rccl) OK_SPMV(spmv_dist_group_start());
        for (int r = 0; r < ndev; ++r) { OK_HIP(hipSetDevice(rk[r].device)); OK_SPMV(spmv_dist_pipe_exchange_only(rk[r].pipe, rk[r].d_y, rk[r].stream)); }
        if (rccl) OK_SPMV(spmv_dist_group_end());
    });
    char plan[256];
    OK_HIP(hipSetDevice(rk[0].device));
    OK_SPMV(spmv_csr_plan_describe(rk[0].blocks[0], variant, plan, sizeof plan));
    const double bytes = 8.0 * nnz_all + 4.0 * (rows + (double)ndev * S) + 4.0 * rows + 4.0 * cols * ndev;
    const char *xname = exchange == SPMV_DIST_ALLGATHER ? "allgather" : exchange == SPMV_DIST_P2P ? "p2p" : "peer";
    printf("{\"world\": %d, \"process_model\": \"%s\", \"pipeline_blocks_per_rank\": %d, \"exchange\": \"%s\", \"rows\": %lld, "
           "\"nnz\": %lld, \"band\": %lld, \"variant\": \"%s\", \"footprint_exchange\": %s, \"rows_never_sent_to_a_rank_that_does_not_need_them\": %lld, "
           "\"rows_differing_from_single_handle\": %lld, \"verified\": %s, \"vary_x_snapshot_words_differing\": %lld, \"step_ms\": %.5f, "
           "\"multiply_only_ms\": %.5f, \"exchange_only_ms\": %.5f, \"aggregate_GBs\": %.1f, \"plan\": \"%s\"}\n",
           ndev, local ? "one process, local ranks (no communicator), devices round-robin" : "one process, ncclCommInitAll", S, xname,
           (long long)rows, (long long)nnz_all, (long long)band, spmv_variant_name(variant), footprint ? "true" : "false",
           (long long)untouched, (long long)differing, verify ? "true" : "false", (long long)vary_bad, step_ms, mult_ms, xchg_ms,
           bytes / (step_ms * 1e-3) / 1e9, plan);
    for (int r = 0; r < ndev; ++r) {
        OK_HIP(hipSetDevice(rk[r].device));
        (void)spmv_dist_pipe_destroy(rk[r].pipe);
        for (int s = 0; s < S; ++s) (void)spmv_csr_destroy(rk[r].blocks[s]);
        (void)spmv_dist_destroy(dist[r]);
    }
    if (whole) (void)spmv_csr_destroy(whole);
    return differing == 0 ? 0 : 1;
}
