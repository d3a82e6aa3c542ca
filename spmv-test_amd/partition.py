"""Row-block partitioning of a CSR matrix across the GPUs of one node.

The reference is single-GPU; this layer is the multi-GPU extension BASELINE.json asks for:
rows are independent, so the matrix is cut into contiguous row blocks, one per rank, each block
keeping GLOBAL column indices (every rank holds all of x) and a row_ptr rebased to local int32
offsets -- which is also how a > 2^31-nonzero matrix (config 5: 2Gi nnz) fits int32 indexing.
Block boundaries are chosen by nonzero count, not row count, so power-law inputs balance.
"""
from __future__ import annotations

import numpy as np


def balanced_row_bounds(row_ptr: np.ndarray, parts: int, align: int = 1) -> np.ndarray:
    """Boundaries b[0..parts] (b[0]=0, b[parts]=rows) so that every block holds ~nnz/parts nonzeros.

    ``row_ptr`` is the global int64 (or int32) array with rows+1 entries.  Boundary p is the first
    row whose starting offset reaches p*nnz/parts, rounded to a multiple of ``align`` rows.
    """
    row_ptr = np.asarray(row_ptr)
    rows = len(row_ptr) - 1
    nnz = int(row_ptr[-1])
    b = np.zeros(parts + 1, dtype=np.int64)
    b[parts] = rows
    for p in range(1, parts):
        target = (nnz * p) // parts
        r = int(np.searchsorted(row_ptr[:-1], target, side="left"))
        if align > 1:
            r = min(rows, ((r + align // 2) // align) * align)
        b[p] = r
    return np.maximum.accumulate(b)


def equal_row_bounds(rows: int, parts: int) -> np.ndarray:
    return np.array([(rows * p) // parts for p in range(parts + 1)], dtype=np.int64)


def shard_row_ptr(row_ptr: np.ndarray, r0: int, r1: int) -> np.ndarray:
    """Local int32 row_ptr (r1-r0+1 entries, starting at 0) of rows [r0, r1)."""
    seg = np.asarray(row_ptr[r0:r1 + 1], dtype=np.int64)
    local = seg - seg[0]
    if local[-1] >= (1 << 31):
        raise ValueError(f"rows [{r0},{r1}) hold {int(local[-1])} nonzeros >= 2^31: use more parts")
    return local.astype(np.int32)


def shard_arrays(row_ptr, col_idx, vals, r0: int, r1: int):
    """Host-side cut of a whole CSR into the shard of rows [r0, r1): (row_ptr32, col_idx, vals)."""
    k0, k1 = int(row_ptr[r0]), int(row_ptr[r1])
    return shard_row_ptr(row_ptr, r0, r1), col_idx[k0:k1], vals[k0:k1]
