"""CPU, world_size 2, gloo: the row-block exchange (broadcast x once, multiply, all-gather y).

The library has no CPU compute path, so the local product is a checker-backed stand-in here;
what is under test is the sharding, the buffers and the collective pattern of dist.ShardedSpmv."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, balanced, q):
    import __graft_entry__ as ge
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg, orc = ge.load_package(), ge.load_oracle()
        W, P = pkg.workloads, pkg.partition
        nb = 3 if balanced else 2     # 3 blocks: the equal-nnz cut falls inside a block -> unequal row counts
        w = W.Workload("t", nb * W.BLOCK_ROWS, nb * W.BLOCK_ROWS, "powerlaw", 32, band=0)
        L = W.row_lengths(w)
        rp_global = np.concatenate([[0], np.cumsum(L)]).astype(np.int64)
        bounds = P.balanced_row_bounds(rp_global, world) if balanced else P.equal_row_bounds(w.rows, world)
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        rp = P.shard_row_ptr(rp_global, r0, r1)
        assert np.array_equal(rp, W.row_ptr(w, r0, r1 - r0))
        ci, va = orc.synth_fill(w.seed, r0, r1, w.rows, w.cols, w.band, rp)

        def local_spmv(x, y_local):     # stand-in for CsrMatrix.run on this rank's shard
            y_local[:r1 - r0] = torch.from_numpy(orc.spmv(rp, ci, va, x.numpy()))

        sh = pkg.dist.ShardedSpmv(bounds, w.cols, local_spmv, torch.device("cpu"))
        if rank == 0:
            sh.x.copy_(torch.from_numpy(orc.synth_x(w.seed, 0, w.cols)))
        sh.broadcast_x(0)
        y = sh.step().numpy().copy()
        # the whole problem on one rank
        rp_all = rp_global.astype(np.int32)
        ci_all, va_all = orc.synth_fill(w.seed, 0, w.rows, w.rows, w.cols, w.band, rp_all)
        y_ref = orc.spmv(rp_all, ci_all, va_all, orc.synth_x(w.seed, 0, w.cols))
        q.put((rank, bool(np.array_equal(y.view(np.uint32), y_ref.view(np.uint32))), sh.uniform))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,balanced", [(2, False), (2, True), (4, True)], ids=["world2-equal", "world2-by-nnz", "world4-by-nnz"])
def test_row_block_exchange(built, world, balanced):
    """world 2 and world 4 (VERDICT next-7d: four ranks with UNEQUAL blocks -- the nnz-balanced cut of a power-law
    matrix -- through the padded-slot all-gather), bit-exact against the whole problem on one rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, balanced, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert [r[1] for r in res] == [True] * world
    assert all(r[2] == (not balanced) for r in res)      # equal rows -> in-place gather path


def _worker_pipe(rank, world, port, S, q, exchange="allgather"):
    import __graft_entry__ as ge
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg, orc = ge.load_package(), ge.load_oracle()
        W = pkg.workloads
        sub = W.BLOCK_ROWS
        w = W.Workload("t", S * world * sub, S * world * sub, "mixed", 16, band=4096)
        blocks = []
        for s in range(S):
            b = s * world + rank
            r0, r1 = b * sub, (b + 1) * sub
            rp = W.row_ptr(w, r0, sub)
            ci, va = orc.synth_fill(w.seed, r0, r1, w.rows, w.cols, w.band, rp)
            blocks.append((rp, ci, va))

        def make(s):
            rp, ci, va = blocks[s]
            return lambda x, y: y.copy_(torch.from_numpy(orc.spmv(rp, ci, va, x.numpy())))

        sh = pkg.dist.PipelinedSpmv(S, sub, w.cols, [make(s) for s in range(S)], torch.device("cpu"), exchange=exchange)
        assert sh.owned_blocks() == [s * world + rank for s in range(S)]
        if rank == 0:
            sh.x.copy_(torch.from_numpy(orc.synth_x(w.seed, 0, w.cols)))
        sh.broadcast_x(0)
        sh.step()
        y = sh.finish().numpy().copy()
        sh.exchange_only()                    # the all-gathers alone (bench.py's exchange_only_ms) leave y unchanged
        assert np.array_equal(sh.finish().numpy(), y)
        rp_all = W.row_ptr(w)
        ci_all, va_all = orc.synth_fill(w.seed, 0, w.rows, w.rows, w.cols, w.band, rp_all)
        y_ref = orc.spmv(rp_all, ci_all, va_all, orc.synth_x(w.seed, 0, w.cols))
        q.put((rank, bool(np.array_equal(y.view(np.uint32), y_ref.view(np.uint32)))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("S", [1, 3])
def test_two_rank_block_cyclic_pipelined_exchange(built, S):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipe, args=(r, 2, port, S, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert [r[1] for r in res] == [True, True]


@pytest.mark.parametrize("world", [2, 3])
def test_block_cyclic_exchange_by_direct_send_recv(built, world):
    """The same pipeline with the block groups concatenated by one isend/irecv pair per peer (bench.py --exchange p2p:
    the all-pairs schedule over the direct xGMI links) instead of the all-gather collective: bit-exact, world 2 and 3."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipe, args=(r, world, port, 2, q, "p2p")) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert [r[1] for r in res] == [True] * world
