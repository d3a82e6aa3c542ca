"""ctypes binding of include/spmv_hip.h (libspmv_hip.so).

Every call goes to the HIP library; there is no Python or CPU implementation behind these
wrappers.  A non-zero status raises :class:`SpmvError` with ``spmv_last_error()``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "lib" / "libspmv_hip.so"
LAUNCHERS_PATH = PKG_DIR / "lib" / "libspmv_launchers.so"
TESTER_PATH = PKG_DIR / "bin" / "sparse_sgemv"
DIST_LIB_PATH = PKG_DIR / "lib" / "libspmv_dist.so"            # include/spmv_dist.h (RCCL; not loaded by this module)
DIST_SELFTEST_PATH = PKG_DIR / "bin" / "spmv_dist_selftest"

# enum spmv_variant
SCALAR, WAVE, WAVE_PIPE, VECTOR, ADAPTIVE, TILED, PANEL, AUTO, XSKIP = range(9)
# the variants that accept ANY CSR matrix ...
VARIANTS = {"scalar": SCALAR, "wave": WAVE, "wave_pipe": WAVE_PIPE, "vector": VECTOR,
            "adaptive": ADAPTIVE, "tiled": TILED, "panel": PANEL, "auto": AUTO}
# ... and with the one that is limited to dense-ish matrices (its plan refuses the others)
ALL_VARIANTS = dict(VARIANTS, xskip=XSKIP)

# enum spmv_status
OK, ERR_NO_DEVICE, ERR_INVALID, ERR_HIP, ERR_VARIANT, ERR_NOT_PLANNED, ERR_STALE_PLAN = 0, -1, -2, -3, -4, -5, -6

# every symbol include/spmv_hip.h declares: name -> (restype, argtypes)
_i32p, _f32p, _vp = C.c_void_p, C.c_void_p, C.c_void_p   # raw addresses (host or device)
_H = C.c_void_p                                           # spmv_csr_t*
_HP = C.POINTER(C.c_void_p)
SIGNATURES = {
    "spmv_device_count": (C.c_int, []),
    "spmv_last_error": (C.c_char_p, []),
    "spmv_variant_name": (C.c_char_p, [C.c_int]),
    "spmv_last_first_launch_ms": (C.c_float, []),
    "spmv_csr_create_host": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, _i32p, _i32p, _f32p, _HP]),
    "spmv_csr_create_device": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, _i32p, _i32p, _f32p, _HP]),
    "spmv_csr_from_dense_host": (C.c_int, [C.c_int, C.c_int, _f32p, _vp, _HP]),
    "spmv_csr_from_dense_device": (C.c_int, [C.c_int, C.c_int, _f32p, _vp, _HP]),
    "spmv_csr_download": (C.c_int, [_H, _i32p, _i32p, _f32p]),
    "spmv_csr_validate": (C.c_int, [_H, C.c_void_p]),
    "spmv_csr_dims": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "spmv_csr_column_range": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _vp]),
    "spmv_csr_destroy": (C.c_int, [_H]),
    "spmv_csr_plan": (C.c_int, [_H, C.c_int, _vp]),
    "spmv_csr_run": (C.c_int, [_H, C.c_int, _f32p, _f32p, _vp]),
    "spmv_csr_values_changed": (C.c_int, [_H]),
    "spmv_csr_plan_get": (C.c_int, [_H, C.c_int, C.POINTER(C.c_int32)]),
    "spmv_csr_plan_set": (C.c_int, [_H, C.c_int, C.POINTER(C.c_int32), _vp]),
    "spmv_csr_plan_like": (C.c_int, [_H, _H, C.c_int, _vp]),
    "spmv_csr_plan_bytes": (C.c_int64, [_H, C.c_int]),
    "spmv_csr_plan_describe": (C.c_int, [_H, C.c_int, C.c_char_p, C.c_int]),
    "spmv_csr_time": (C.c_int, [_H, C.c_int, _f32p, _f32p, C.c_int, _vp, C.POINTER(C.c_float)]),
    "spmv_csr_run_host": (C.c_int, [_H, C.c_int, _f32p, _f32p, C.POINTER(C.c_float)]),
    "spmv_dense_gemv": (C.c_int, [C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_int, _vp]),
    "spmv_dense_gemv_workspace_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "spmv_dense_gemv_ws": (C.c_int, [C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_int, _vp, C.c_int64, _vp]),
    "spmv_asp_retile": (C.c_int, [C.c_int, C.c_int, _f32p, _f32p, _vp]),
    "spmv_asp_gemv_ws": (C.c_int, [C.c_int, C.c_int, _f32p, _f32p, _f32p, _vp, C.c_int64, _vp]),
    "spmv_dense_gemv_host": (C.c_int, [C.c_int, C.c_int, _f32p, _f32p, _f32p, C.c_int, C.POINTER(C.c_float)]),
    "spmv_tcsr_from_dense_host": (C.c_int, [C.c_int, C.c_int, _f32p, _vp, _HP]),
    "spmv_tcsr_from_dense_device": (C.c_int, [C.c_int, C.c_int, _f32p, _vp, _HP]),
    "spmv_tcsr_sizes": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "spmv_tcsr_download": (C.c_int, [_H, _i32p, _i32p, _f32p]),
    "spmv_tcsr_run": (C.c_int, [_H, _f32p, _f32p, _vp]),
    "spmv_tcsr_run_host": (C.c_int, [_H, _f32p, _f32p, C.POINTER(C.c_float)]),
    "spmv_tcsr_destroy": (C.c_int, [_H]),
    "spmv_bitmap_from_dense_host": (C.c_int, [C.c_int, C.c_int, C.c_int, _f32p, _vp, _HP]),
    "spmv_bitmap_from_dense_device": (C.c_int, [C.c_int, C.c_int, C.c_int, _f32p, _vp, _HP]),
    "spmv_bitmap_sizes": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "spmv_bitmap_download": (C.c_int, [_H, _i32p, _f32p]),
    "spmv_bitmap_run": (C.c_int, [_H, _f32p, _f32p, _vp]),
    "spmv_bitmap_run_host": (C.c_int, [_H, _f32p, _f32p, C.POINTER(C.c_float)]),
    "spmv_bitmap_destroy": (C.c_int, [_H]),
    "spmv_synth_fill": (C.c_int, [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                  _i32p, _i32p, _f32p, _vp]),
    "spmv_synth_x": (C.c_int, [C.c_uint64, C.c_int64, C.c_int64, _f32p, _vp]),
    "spmv_calib_stream": (C.c_int, [_vp, C.c_int64, _f32p, _vp]),
    "spmv_calib_gather": (C.c_int, [_f32p, C.c_int64, C.c_int64, C.c_int, _f32p, _vp]),
    "spmv_calib_store": (C.c_int, [_f32p, C.c_int64, C.c_int, _vp]),
    "spmv_calib_marker": (C.c_int, [C.c_int, _vp]),
}


class SpmvError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libspmv_hip status {status}: {message}")
        self.status = status


_lib = None


def use_library(path) -> None:
    """Development aid (tools/explore.py A/B runs): bind to another build of libspmv_hip.so.
    Handles created before the switch must already be closed."""
    global _lib, LIB_PATH
    LIB_PATH = Path(path).resolve()
    _lib = None


def lib() -> C.CDLL:
    """Load libspmv_hip.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C spmv-test_amd`); there is no fallback implementation")
        # torch bundles its own libamdhip64.so.7; importing it first makes this library bind to
        # the same HIP runtime instance, so torch device pointers and streams are valid here.
        import torch  # noqa: F401
        l = C.CDLL(os.fspath(LIB_PATH), mode=C.RTLD_LOCAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(status: int) -> None:
    if status != OK:
        raise SpmvError(status, lib().spmv_last_error().decode(errors="replace"))


def device_count() -> int:
    return lib().spmv_device_count()


def _ptr(t) -> int:
    """Address of a torch tensor / numpy array / None."""
    if t is None:
        return 0
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def _stream_handle(stream=None) -> int:
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream


class CsrMatrix:
    """Owner of one ``spmv_csr_t`` handle (a whole matrix or one row-block shard)."""

    def __init__(self, handle: int, keepalive=()):
        self._h = C.c_void_p(handle)
        self._keep = tuple(keepalive)
        r, c, z = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().spmv_csr_dims(self._h, C.byref(r), C.byref(c), C.byref(z)))
        self.rows, self.cols, self.nnz = r.value, c.value, z.value

    # -- constructors ---------------------------------------------------------
    @classmethod
    def from_host(cls, rows, cols, row_ptr, col_idx, vals):
        import numpy as np
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.float32)
        h = C.c_void_p()
        check(lib().spmv_csr_create_host(rows, cols, int(col_idx.size), _ptr(row_ptr), _ptr(col_idx),
                                         _ptr(vals), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_device(cls, rows, cols, row_ptr, col_idx, vals):
        """Borrow torch device tensors (int32, int32, float32); they are kept alive by this object."""
        h = C.c_void_p()
        check(lib().spmv_csr_create_device(rows, cols, int(col_idx.numel()), _ptr(row_ptr), _ptr(col_idx),
                                           _ptr(vals), C.byref(h)))
        return cls(h.value, keepalive=(row_ptr, col_idx, vals))

    @classmethod
    def from_dense_host(cls, A):
        """Dense row-major numpy A[M][N] -> CSR of A^T on the device (matrix_csr.cpp:5-23 semantics)."""
        import numpy as np
        A = np.ascontiguousarray(A, dtype=np.float32)
        M, N = A.shape
        h = C.c_void_p()
        check(lib().spmv_csr_from_dense_host(M, N, _ptr(A), 0, C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_dense_device(cls, A):
        M, N = A.shape
        h = C.c_void_p()
        check(lib().spmv_csr_from_dense_device(M, N, _ptr(A), _stream_handle(), C.byref(h)))
        return cls(h.value)

    # -- hot path ---------------------------------------------------------------
    def plan(self, variant: int, stream=None) -> None:
        check(lib().spmv_csr_plan(self._h, variant, _stream_handle(stream)))

    def run(self, variant: int, x, y, stream=None) -> None:
        """Enqueue y = A x on torch's current stream (or ``stream``).  x, y: float32 device tensors."""
        assert x.numel() >= self.cols and y.numel() >= self.rows
        check(lib().spmv_csr_run(self._h, variant, _ptr(x), _ptr(y), _stream_handle(stream)))

    def column_range(self, stream=None):
        """(smallest, largest) column index referenced; (cols, -1) for a matrix without nonzeros."""
        lo, hi = C.c_int64(), C.c_int64()
        check(lib().spmv_csr_column_range(self._h, C.byref(lo), C.byref(hi), _stream_handle(stream)))
        return lo.value, hi.value

    def values_changed(self) -> None:
        """The caller rewrote vals (borrowed arrays): plans that hold a copy of them are stale from here on."""
        check(lib().spmv_csr_values_changed(self._h))

    def time(self, variant: int, x, y, iters: int, stream=None) -> float:
        """Mean ms per launch over ``iters`` launches, HIP events on the launch stream."""
        ms = C.c_float()
        check(lib().spmv_csr_time(self._h, variant, _ptr(x), _ptr(y), iters, _stream_handle(stream),
                                  C.byref(ms)))
        return ms.value

    def run_host(self, variant: int, x, y) -> float:
        import numpy as np
        assert x.dtype == np.float32 and y.dtype == np.float32
        ms = C.c_float()
        check(lib().spmv_csr_run_host(self._h, variant, _ptr(x), _ptr(y), C.byref(ms)))
        return ms.value

    def plan_params(self, variant: int):
        """The eight numbers that fix a plan's chunk cuts (spmv_csr_plan_get)."""
        a = (C.c_int32 * 8)()
        check(lib().spmv_csr_plan_get(self._h, variant, a))
        return list(a)

    def plan_set(self, variant: int, params, stream=None) -> None:
        a = (C.c_int32 * 8)(*params)
        check(lib().spmv_csr_plan_set(self._h, variant, a, _stream_handle(stream)))

    def plan_like(self, other: "CsrMatrix", variant: int, stream=None) -> None:
        check(lib().spmv_csr_plan_like(self._h, other._h, variant, _stream_handle(stream)))

    def plan_describe(self, variant: int) -> str:
        buf = C.create_string_buffer(256)
        check(lib().spmv_csr_plan_describe(self._h, variant, buf, 256))
        return buf.value.decode()

    def plan_bytes(self, variant: int) -> int:
        return lib().spmv_csr_plan_bytes(self._h, variant)

    def download(self):
        import numpy as np
        rp = np.empty(self.rows + 1, np.int32)
        ci = np.empty(self.nnz, np.int32)
        va = np.empty(self.nnz, np.float32)
        check(lib().spmv_csr_download(self._h, _ptr(rp), _ptr(ci), _ptr(va)))
        return rp, ci, va

    def close(self) -> None:
        if self._h:
            check(lib().spmv_csr_destroy(self._h))
            self._h = C.c_void_p()
            self._keep = ()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TcsrMatrix:
    """Owner of one ``spmv_tcsr_t``: the reference's tiled bitmap-CSR (tcsr.cpp:5-38) on the device."""

    def __init__(self, handle: int, M: int, N: int):
        self._h = C.c_void_p(handle)
        self.M, self.N = M, N
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().spmv_tcsr_sizes(self._h, C.byref(a), C.byref(b), C.byref(c)))
        self.n_blk_idx, self.n_bitmaps, self.nnz = a.value, b.value, c.value

    @classmethod
    def from_dense_host(cls, A):
        import numpy as np
        A = np.ascontiguousarray(A, dtype=np.float32)
        M, N = A.shape
        h = C.c_void_p()
        check(lib().spmv_tcsr_from_dense_host(M, N, _ptr(A), 0, C.byref(h)))
        return cls(h.value, M, N)

    @classmethod
    def from_dense_device(cls, A):
        M, N = A.shape
        h = C.c_void_p()
        check(lib().spmv_tcsr_from_dense_device(M, N, _ptr(A), _stream_handle(), C.byref(h)))
        return cls(h.value, M, N)

    def download(self):
        import numpy as np
        bi = np.empty(self.n_blk_idx, np.int32)
        bm = np.empty(self.n_bitmaps, np.uint32)
        va = np.empty(self.nnz, np.float32)
        check(lib().spmv_tcsr_download(self._h, _ptr(bi), _ptr(bm), _ptr(va)))
        return bi, bm, va

    def run(self, x, y, stream=None) -> None:
        assert x.numel() >= self.M and y.numel() >= self.N
        check(lib().spmv_tcsr_run(self._h, _ptr(x), _ptr(y), _stream_handle(stream)))

    def run_host(self, x, y) -> float:
        ms = C.c_float()
        check(lib().spmv_tcsr_run_host(self._h, _ptr(x), _ptr(y), C.byref(ms)))
        return ms.value

    def close(self) -> None:
        if self._h:
            check(lib().spmv_tcsr_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


BITMAP_FORMATS = {"wsp": 0, "awsp": 1, "awsp_ref": 2}      # enum spmv_bitmap_format


class BitmapMatrix:
    """Owner of one ``spmv_bitmap_t``: the reference's WSP / AWSP / AWSPRef format (wsp.cpp, awsp.cpp,
    awsp_ref.cpp) on the device."""

    def __init__(self, handle: int, fmt: str, M: int, N: int):
        self._h = C.c_void_p(handle)
        self.fmt, self.M, self.N = fmt, M, N
        a, b = C.c_int64(), C.c_int64()
        st = (C.c_int32 * 4)()
        check(lib().spmv_bitmap_sizes(self._h, C.byref(a), C.byref(b), st))
        self.n_bitmaps, self.n_vals, self.stats = a.value, b.value, list(st)

    @classmethod
    def from_dense_host(cls, fmt: str, A):
        import numpy as np
        A = np.ascontiguousarray(A, dtype=np.float32)
        M, N = A.shape
        h = C.c_void_p()
        check(lib().spmv_bitmap_from_dense_host(BITMAP_FORMATS[fmt], M, N, _ptr(A), 0, C.byref(h)))
        return cls(h.value, fmt, M, N)

    @classmethod
    def from_dense_device(cls, fmt: str, A):
        M, N = A.shape
        h = C.c_void_p()
        check(lib().spmv_bitmap_from_dense_device(BITMAP_FORMATS[fmt], M, N, _ptr(A), _stream_handle(), C.byref(h)))
        return cls(h.value, fmt, M, N)

    def download(self):
        import numpy as np
        bm = np.empty(self.n_bitmaps, np.uint32)
        va = np.empty(self.n_vals, np.float32)
        check(lib().spmv_bitmap_download(self._h, _ptr(bm), _ptr(va)))
        return bm, va

    def run(self, x, y, stream=None) -> None:
        assert x.numel() >= self.M and y.numel() >= self.N
        check(lib().spmv_bitmap_run(self._h, _ptr(x), _ptr(y), _stream_handle(stream)))

    def run_host(self, x, y) -> float:
        ms = C.c_float()
        check(lib().spmv_bitmap_run_host(self._h, _ptr(x), _ptr(y), C.byref(ms)))
        return ms.value

    def close(self) -> None:
        if self._h:
            check(lib().spmv_bitmap_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dense_gemv(A, x, y, mode: int, stream=None, workspace=None) -> None:
    """y = A^T x on the dense device matrix; with ``workspace`` (a device tensor of at least
    ``dense_gemv_workspace_bytes(N, mode)`` bytes) through the allocation-free entry."""
    M, N = A.shape
    if workspace is None:
        check(lib().spmv_dense_gemv(M, N, _ptr(A), _ptr(x), _ptr(y), mode, _stream_handle(stream)))
    else:
        check(lib().spmv_dense_gemv_ws(M, N, _ptr(A), _ptr(x), _ptr(y), mode, _ptr(workspace),
                                       workspace.numel() * workspace.element_size(), _stream_handle(stream)))


def dense_gemv_workspace_bytes(N: int, mode: int) -> int:
    return lib().spmv_dense_gemv_workspace_bytes(N, mode)


def asp_retile(A, out, stream=None) -> None:
    """The reference's ASPMatrix layout of the dense device matrix A[M][N] into ``out`` (M*N floats)."""
    M, N = A.shape
    check(lib().spmv_asp_retile(M, N, _ptr(A), _ptr(out), _stream_handle(stream)))


def asp_gemv(M: int, N: int, asp, x, y, workspace, stream=None) -> None:
    """y = A^T x from the ASP layout, rows with x == 0 skipped; ``workspace``: dense_gemv_workspace_bytes(N, 3) bytes."""
    check(lib().spmv_asp_gemv_ws(M, N, _ptr(asp), _ptr(x), _ptr(y), _ptr(workspace),
                                 workspace.numel() * workspace.element_size(), _stream_handle(stream)))


def synth_fill(seed, row0, n_local, rows, cols, band, row_ptr, col_idx, vals, stream=None) -> None:
    check(lib().spmv_synth_fill(seed, row0, n_local, rows, cols, band, _ptr(row_ptr), _ptr(col_idx),
                                _ptr(vals), _stream_handle(stream)))


def synth_x(seed, j0, n, x, stream=None) -> None:
    check(lib().spmv_synth_x(seed, j0, n, _ptr(x), _stream_handle(stream)))


# -- measurement aids (kernels of known traffic for calibrating rocprofv3's counters; include/spmv_hip.h) ----------
def calib_stream(src, nbytes: int, sink, stream=None) -> None:
    check(lib().spmv_calib_stream(_ptr(src), nbytes, _ptr(sink), _stream_handle(stream)))


def calib_gather(table, table_lines: int, n_lines: int, touch: int, sink, stream=None) -> None:
    check(lib().spmv_calib_gather(_ptr(table), table_lines, n_lines, touch, _ptr(sink), _stream_handle(stream)))


def calib_store(dst, nbytes: int, width: int, stream=None) -> None:
    check(lib().spmv_calib_store(_ptr(dst), nbytes, width, _stream_handle(stream)))


def calib_marker(ident: int, stream=None) -> None:
    check(lib().spmv_calib_marker(ident, _stream_handle(stream)))
