"""One-process-per-GPU sharded SpMV: y = A x with A cut into row blocks (partition.py).

Exchange pattern (BASELINE.json north_star): x is broadcast once, every rank multiplies its row
block, and the output slices are concatenated on every rank with ONE all-gather over RCCL
(``torch.distributed`` backend "nccl" on ROCm; "gloo" in the CPU tests).  y is the only data that
moves per step: rows_local * 4 bytes out, (world-1) slices in, over the xGMI mesh.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.distributed as dist


class ShardedSpmv:
    """Holds one rank's shard and the buffers of the exchange.

    ``local_spmv(x_full, y_local)`` enqueues the rank's row block product; in the product it is
    ``CsrMatrix.run`` bound to a variant, in the CPU (gloo) tests it is a checker-backed stand-in,
    because this library has no CPU compute path.
    """

    def __init__(self, bounds: Sequence[int], cols: int, local_spmv: Callable, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if len(bounds) != self.world + 1:
            raise ValueError("bounds must have world+1 entries")
        self.bounds = [int(b) for b in bounds]
        self.rows = self.bounds[-1]
        self.cols = cols
        self.r0, self.r1 = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.rows_local = self.r1 - self.r0
        sizes = [self.bounds[p + 1] - self.bounds[p] for p in range(self.world)]
        self.max_rows = max(sizes)
        self.uniform = all(s == self.max_rows for s in sizes)
        self.local_spmv = local_spmv
        self.x = torch.zeros(cols, dtype=torch.float32, device=device)
        # the gather buffer: world slots of max_rows; with equal blocks it IS the full y
        self.y_slots = torch.zeros(self.world * self.max_rows, dtype=torch.float32, device=device)
        self.y_local = self.y_slots[self.rank * self.max_rows:(self.rank + 1) * self.max_rows]
        self.y_full = self.y_slots if self.uniform else torch.zeros(self.rows, dtype=torch.float32, device=device)

    def broadcast_x(self, src: int = 0) -> None:
        """The one-off distribution of the dense vector."""
        if self.x.device.type == "cuda" and dist.get_backend(self.group) != "nccl":
            host = self.x.cpu()
            dist.broadcast(host, src=src, group=self.group)
            self.x.copy_(host)
        else:
            dist.broadcast(self.x, src=src, group=self.group)

    def multiply(self) -> None:
        """Local row block only (no communication)."""
        self.local_spmv(self.x, self.y_local)

    def gather(self) -> torch.Tensor:
        """Concatenate the slices on every rank; returns the full y."""
        if self.world > 1:
            backend = dist.get_backend(self.group)
            if self.y_slots.device.type == "cuda" and backend != "nccl":
                # rehearsal path (gloo ranks sharing one GPU): stage through the host
                slots = torch.empty(self.y_slots.shape, dtype=torch.float32)
                dist.all_gather_into_tensor(slots, self.y_local.cpu(), group=self.group)
                self.y_slots.copy_(slots)
            else:
                # in-place all-gather: rank r's slot already holds its slice (RCCL accepts
                # sendbuff == recvbuff + rank*count); gloo on CPU wants a separate input
                src = self.y_local.clone() if self.y_slots.device.type == "cpu" else self.y_local
                dist.all_gather_into_tensor(self.y_slots, src, group=self.group)
        if not self.uniform:
            for p in range(self.world):
                n = self.bounds[p + 1] - self.bounds[p]
                self.y_full[self.bounds[p]:self.bounds[p + 1]] = self.y_slots[p * self.max_rows:p * self.max_rows + n]
        return self.y_full

    def step(self) -> torch.Tensor:
        self.multiply()
        return self.gather()


class _Works:
    """The works of one batched isend/irecv, waited for as one."""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()


class PipelinedSpmv:
    """Block-cyclic row blocks with the all-gather of one block group overlapped with the multiply
    of the next (SURVEY section 8e "Overlap").

    The global row space is cut into ``S * world`` equal blocks of ``sub_rows`` rows; rank p owns
    blocks ``s * world + p`` for s = 0..S-1.  Group s (one block per rank) is contiguous in the
    natural row order, so ONE in-place all-gather per group lands it in ``y_full`` with no
    permutation copy:  y_full[(s*world + p)*sub_rows : ...] is rank p's slot of group s.
    A step issues, for s = 0..S-1: the local product of block s on the compute stream, then --
    on a side stream that waits for that product only -- the all-gather of group s; RCCL moves
    group s over xGMI while the CUs multiply block s+1.  S = 1 degenerates to the plain
    multiply-then-gather.

    ``local_spmvs[s](x_full, y_slot)`` enqueues the product of this rank's s-th block on the
    CURRENT stream (the product binds ``CsrMatrix.run`` of one handle per block).
    """

    def __init__(self, S: int, sub_rows: int, cols: int, local_spmvs, device, group=None, exchange: str = "allgather"):
        self.group = group
        # how a block group is concatenated: "allgather" = one in-place all_gather_into_tensor (RCCL's collective: rings
        # over the xGMI mesh); "p2p" = one batched isend/irecv pair per peer (grouped ncclSend/ncclRecv: every slice
        # travels its owner's DIRECT link to each peer -- the all-pairs schedule of SURVEY section 5, 7 links busy at
        # once).  Same bytes, same result; which is faster is for an 8-GPU node to say (bench.py --exchange).
        if exchange not in ("allgather", "p2p"):
            raise ValueError("exchange must be 'allgather' or 'p2p'")
        self.exchange = exchange
        # one process, no process group: every block is local and nothing is exchanged (bench --scaling strong, N = 1)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if len(local_spmvs) != S:
            raise ValueError("need one local product per owned block")
        self.S, self.sub_rows, self.cols = S, sub_rows, cols
        self.rows = S * self.world * sub_rows
        self.local_spmvs = list(local_spmvs)
        self.device = torch.device(device)
        self.x = torch.zeros(cols, dtype=torch.float32, device=device)
        self.y_full = torch.zeros(self.rows, dtype=torch.float32, device=device)
        self.cuda = self.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=device) if self.cuda else None
        self._pending = []

    def owned_blocks(self):
        """Global block ids this rank owns, in the order of ``local_spmvs``."""
        return [s * self.world + self.rank for s in range(self.S)]

    def block_rows(self, s: int):
        b = s * self.world + self.rank
        return b * self.sub_rows, (b + 1) * self.sub_rows

    def broadcast_x(self, src: int = 0) -> None:
        if self.cuda and dist.get_backend(self.group) != "nccl":
            host = self.x.cpu()
            dist.broadcast(host, src=src, group=self.group)
            self.x.copy_(host)
        else:
            dist.broadcast(self.x, src=src, group=self.group)

    def _group_view(self, s: int):
        n = self.world * self.sub_rows
        return self.y_full[s * n:(s + 1) * n]

    def _gather_group(self, s: int):
        grp = self._group_view(s)
        mine = grp[self.rank * self.sub_rows:(self.rank + 1) * self.sub_rows]
        if self.world == 1:
            return None
        backend = dist.get_backend(self.group)
        if self.cuda and backend != "nccl":      # rehearsal: gloo ranks sharing one GPU, via the host
            host = torch.empty(grp.shape, dtype=torch.float32)
            dist.all_gather_into_tensor(host, mine.cpu(), group=self.group)
            grp.copy_(host)
            return None
        if self.exchange == "p2p":
            ops = []
            for peer in range(self.world):
                if peer == self.rank:
                    continue
                slot = grp[peer * self.sub_rows:(peer + 1) * self.sub_rows]
                ops.append(dist.P2POp(dist.isend, mine, peer, group=self.group))
                ops.append(dist.P2POp(dist.irecv, slot, peer, group=self.group))
            return _Works(dist.batch_isend_irecv(ops))
        src = mine if self.cuda else mine.clone()
        return dist.all_gather_into_tensor(grp, src, group=self.group, async_op=True)

    def step(self) -> torch.Tensor:
        """One y = A x: S products, S overlapped all-gathers.  Everything is ENQUEUED when this returns.
        Steps pipeline across the call boundary: the product of block s only waits for the previous
        step's gather of group s (which reads the slot it is about to overwrite), so the tail gathers
        of one step overlap the head products of the next.  Call ``finish()`` before reading
        ``y_full`` on the current stream (or at the end of a timed region)."""
        if not self.cuda:
            for s in range(self.S):
                a, b = self.block_rows(s)
                self.local_spmvs[s](self.x, self.y_full[a:b])
                w = self._gather_group(s)
                if w is not None:
                    w.wait()
            return self.y_full
        compute = torch.cuda.current_stream(self.device)
        if not self._pending:
            self._pending = [None] * self.S
        for s in range(self.S):
            a, b = self.block_rows(s)
            if self._pending[s] is not None:
                compute.wait_event(self._pending[s])         # last step's gather of this slot is done
            self.local_spmvs[s](self.x, self.y_full[a:b])
            if self.world > 1:
                ev = torch.cuda.Event()
                ev.record(compute)
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(ev)          # gather s starts when product s is done
                    w = self._gather_group(s)
                    if w is not None:
                        w.wait()                             # orders comm_stream behind the collective
                    done = torch.cuda.Event()
                    done.record(self.comm_stream)
                self._pending[s] = done
        return self.y_full

    def exchange_only(self) -> None:
        """The S all-gathers of one step without the products (what the exchange alone costs; bench.py prints it
        beside ``multiply_only_ms``).  Enqueued on the side stream like in ``step``; call ``finish()`` after."""
        if self.world == 1:
            return
        if not self.cuda:
            for s in range(self.S):
                w = self._gather_group(s)
                if w is not None:
                    w.wait()
            return
        compute = torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(compute)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            for s in range(self.S):
                w = self._gather_group(s)
                if w is not None:
                    w.wait()

    def finish(self) -> torch.Tensor:
        """Order the current stream behind every outstanding gather; y_full is then complete."""
        if self.cuda and self.world > 1:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        return self.y_full
