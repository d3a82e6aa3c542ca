"""ctypes binding of include/spmv_dist.h (lib/libspmv_dist.so): the row-block exchange over RCCL below Python.

bench.py --backend native drives the C++ pipeline through this module -- the same calls a C++ caller of the reference's
tester world would make (spmv_dist_init with a unique id handed over by the launcher, spmv_dist_pipe_step per step) -- while
torch.distributed (gloo, CPU) only carries the 128-byte id, the plan numbers, the barrier and the max-over-ranks of the
timing.  No data-path byte goes through torch.
"""
from __future__ import annotations

import ctypes as C
import os

from . import capi

ALLGATHER, P2P, PEER_STORE = 0, 1, 2
EXCHANGES = {"allgather": ALLGATHER, "p2p": P2P, "peer": PEER_STORE}
ID_BYTES = 128

_vp = C.c_void_p
_PP = C.POINTER(C.c_void_p)
# every symbol include/spmv_dist.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "spmv_dist_last_error": (C.c_char_p, []),
    "spmv_dist_get_unique_id": (C.c_int, [_vp]),
    "spmv_dist_init": (C.c_int, [C.c_int, C.c_int, _vp, _PP]),
    "spmv_dist_init_all": (C.c_int, [C.c_int, C.POINTER(C.c_int), _PP]),
    "spmv_dist_init_local": (C.c_int, [C.c_int, C.POINTER(C.c_int), _PP]),
    "spmv_dist_group_start": (C.c_int, []),
    "spmv_dist_group_end": (C.c_int, []),
    "spmv_dist_rank": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "spmv_dist_set_partition": (C.c_int, [_vp, C.POINTER(C.c_int64), C.c_int64]),
    "spmv_dist_broadcast_x": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "spmv_dist_plan_like_root": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp]),
    "spmv_dist_spmv": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp]),
    "spmv_dist_allgather_y": (C.c_int, [_vp, _vp, _vp]),
    "spmv_dist_destroy": (C.c_int, [_vp]),
    "spmv_dist_pipe_create": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int64, C.c_int, _PP]),
    "spmv_dist_pipe_link": (C.c_int, [_PP, _PP, C.c_int]),
    "spmv_dist_pipe_step": (C.c_int, [_vp, _PP, C.c_int, _vp, _vp, _vp]),
    "spmv_dist_pipe_exchange_only": (C.c_int, [_vp, _vp, _vp]),
    "spmv_dist_pipe_set_footprint": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]),
    "spmv_dist_pipe_finish": (C.c_int, [_vp, _vp]),
    "spmv_dist_pipe_release": (C.c_int, [_vp, _vp]),
    "spmv_dist_pipe_destroy": (C.c_int, [_vp]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        capi.lib()                                   # libspmv_hip.so first: libspmv_dist.so binds to the same instance
        if not capi.DIST_LIB_PATH.exists():
            raise FileNotFoundError(f"{capi.DIST_LIB_PATH} is missing: run __graft_entry__.build()")
        l = C.CDLL(os.fspath(capi.DIST_LIB_PATH), mode=C.RTLD_LOCAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class DistError(RuntimeError):
    pass


def check(status: int) -> None:
    if status != 0:
        raise DistError(f"libspmv_dist status {status}: {lib().spmv_dist_last_error().decode(errors='replace')} | "
                        f"{capi.lib().spmv_last_error().decode(errors='replace')}")


def unique_id() -> bytes:
    buf = C.create_string_buffer(ID_BYTES)
    check(lib().spmv_dist_get_unique_id(buf))
    return buf.raw


class NativePipeline:
    """One rank of the pipelined sharded SpMV of include/spmv_dist.h (one process per GPU): the same surface as
    dist.PipelinedSpmv (x, y_full, step, finish, exchange_only, broadcast_x, block_rows), RCCL called from C++."""

    def __init__(self, world: int, rank: int, id128: bytes, S: int, sub_rows: int, cols: int, handles, variant: int, device,
                 exchange: str = "allgather"):
        import torch
        if exchange not in ("allgather", "p2p"):
            raise ValueError("one process per GPU exchanges with RCCL: 'allgather' or 'p2p'")
        self.world, self.rank, self.S, self.sub_rows, self.cols = world, rank, S, sub_rows, cols
        self.variant = variant
        self.handles = list(handles)
        self.rows = S * world * sub_rows
        self.device = torch.device(device)
        self.x = torch.zeros(cols, dtype=torch.float32, device=device)
        self.y_full = torch.zeros(self.rows, dtype=torch.float32, device=device)
        self._d = C.c_void_p()
        check(lib().spmv_dist_init(world, rank, C.c_char_p(id128), C.byref(self._d)))
        bounds = (C.c_int64 * (world + 1))(*[p * S * sub_rows for p in range(world + 1)])
        check(lib().spmv_dist_set_partition(self._d, bounds, cols))      # (only broadcast_x reads it: cols)
        self._p = C.c_void_p()
        check(lib().spmv_dist_pipe_create(self._d, S, sub_rows, cols, EXCHANGES[exchange], C.byref(self._p)))
        self._blocks = (C.c_void_p * S)(*[h._h for h in self.handles])

    def block_rows(self, s: int):
        b = s * self.world + self.rank
        return b * self.sub_rows, (b + 1) * self.sub_rows

    def broadcast_x(self, src: int = 0) -> None:
        check(lib().spmv_dist_broadcast_x(self._d, self.x.data_ptr(), src, capi._stream_handle()))

    def set_footprint(self, need_lo, need_hi) -> None:
        """The optional footprint exchange: ``need_lo/need_hi[q][k]`` = rows of y rank q's k-th block references (the same
        lists on every rank); 'p2p' only."""
        per_rank = len(need_lo[0])
        flat_lo = [int(v) for row in need_lo for v in row]
        flat_hi = [int(v) for row in need_hi for v in row]
        n = len(flat_lo)
        check(lib().spmv_dist_pipe_set_footprint(self._p, (C.c_int64 * n)(*flat_lo), (C.c_int64 * n)(*flat_hi), per_rank))

    def step(self):
        check(lib().spmv_dist_pipe_step(self._p, self._blocks, self.variant, self.x.data_ptr(), self.y_full.data_ptr(),
                                        capi._stream_handle()))
        return self.y_full

    def exchange_only(self) -> None:
        check(lib().spmv_dist_pipe_exchange_only(self._p, self.y_full.data_ptr(), capi._stream_handle()))

    def finish(self):
        check(lib().spmv_dist_pipe_finish(self._p, capi._stream_handle()))
        return self.y_full

    def release(self) -> None:
        """The readers of y_full enqueued so far come before the peers' stores of the next step (peer stores only)."""
        check(lib().spmv_dist_pipe_release(self._p, capi._stream_handle()))

    def close(self) -> None:
        if self._p:
            import torch
            torch.cuda.synchronize()
            check(lib().spmv_dist_pipe_destroy(self._p))
            check(lib().spmv_dist_destroy(self._d))
            self._p = self._d = C.c_void_p()
