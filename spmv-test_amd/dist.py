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
