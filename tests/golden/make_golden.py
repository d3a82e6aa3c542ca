#!/usr/bin/env python3
"""Generate tests/golden/*.npz -- run in the build container only (needs /root/reference).

The CSR arrays in every fixture come from the REFERENCE's own CSRMatrix class
(/root/reference/src/matrix_csr.cpp:5-23), compiled unmodified into
oracle/_ref/libref_formats.so by oracle/Makefile and called through oracle/ref_wrap.cpp.
They are stored in the reference's layout: ``ref_row_ptrs`` has N entries and NO trailing
sentinel (matrix_csr.cpp:10-22).

``y_dense`` is the dense loop of SgemvCPU (src/tester.cpp:36-45).  tester.cpp itself cannot be compiled in
this image without stand-in CUDA headers, so the vector is not produced by reference code; it is computed
TWICE by statements of that 8-line loop that share no code -- ``numpy_sgemv`` below (numpy float32, one
rounding for the product and one for the add, j ascending; nothing from oracle/) and
oracle/spmv_oracle.c:oracle_sgemv_dense (C, gcc -O2 -ffp-contract=off) -- and written only when the two
agree bit for bit.  The stored vector is the numpy one (``y_source``); tests/test_oracle.py then checks the C
oracle against it, so that check is no longer the oracle against itself.

For 32-aligned inputs the fixture also holds the reference's bitmap formats of the same matrix (WSPMatrix,
AWSPMatrix, AWSPRefMatrix; src/wsp.cpp, src/awsp.cpp, src/awsp_ref.cpp) with the statistics the classes expose
(``nz_max_m``, ``nz_bk_max_``, ``warp_nz_offset_[4]``), and a checksum of ASPMatrix's re-tiled values.

Fixtures are data only (inputs and expected outputs).  Usage:  python tests/golden/make_golden.py
"""
import ctypes
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402


def ref_csr(A):
    lib = ctypes.CDLL(str(ROOT / "oracle" / "_ref" / "libref_formats.so"))
    lib.ref_csr_build.restype = ctypes.c_void_p
    lib.ref_csr_build.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3
    lib.ref_csr_copy.argtypes = [ctypes.c_void_p] * 4
    lib.ref_csr_free.argtypes = [ctypes.c_void_p]
    A = np.ascontiguousarray(A, np.float32)
    M, N = A.shape
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    h = lib.ref_csr_build(M, N, A.ctypes.data, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    rp = np.empty(a.value, np.int32)
    ci = np.empty(b.value, np.int32)
    va = np.empty(c.value, np.float32)
    lib.ref_csr_copy(h, rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
    lib.ref_csr_free(h)
    return rp, ci, va


def ref_tcsr(A):
    """The reference's TCSRMatrix (src/tcsr.cpp:5-38) on A; needs M, N multiples of 32."""
    lib = ctypes.CDLL(str(ROOT / "oracle" / "_ref" / "libref_formats.so"))
    lib.ref_tcsr_build.restype = ctypes.c_void_p
    lib.ref_tcsr_build.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3
    lib.ref_tcsr_copy.argtypes = [ctypes.c_void_p] * 4
    lib.ref_tcsr_free.argtypes = [ctypes.c_void_p]
    A = np.ascontiguousarray(A, np.float32)
    M, N = A.shape
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    h = lib.ref_tcsr_build(M, N, A.ctypes.data, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    bi = np.empty(a.value, np.int32)
    bm = np.empty(b.value, np.uint32)
    va = np.empty(c.value, np.float32)
    lib.ref_tcsr_copy(h, bi.ctypes.data, bm.ctypes.data, va.ctypes.data)
    lib.ref_tcsr_free(h)
    return bi, bm, va


def ref_fmt(kind, A):
    """kind 0 WSPMatrix, 1 AWSPMatrix, 2 AWSPRefMatrix, 3 ASPMatrix -> (bitmaps, vals, stats[4]) from the reference build."""
    lib = ctypes.CDLL(str(ROOT / "oracle" / "_ref" / "libref_formats.so"))
    lib.ref_fmt_build.restype = ctypes.c_void_p
    lib.ref_fmt_build.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                  ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    lib.ref_fmt_copy.argtypes = [ctypes.c_void_p] * 3
    lib.ref_fmt_free.argtypes = [ctypes.c_void_p]
    A = np.ascontiguousarray(A, np.float32)
    M, N = A.shape
    nb, nv = ctypes.c_int(), ctypes.c_int()
    st = np.zeros(4, np.int32)
    h = lib.ref_fmt_build(kind, M, N, A.ctypes.data, ctypes.byref(nb), ctypes.byref(nv), st.ctypes.data)
    bm = np.empty(nb.value, np.uint32)
    va = np.empty(nv.value, np.float32)
    lib.ref_fmt_copy(h, bm.ctypes.data, va.ctypes.data)
    lib.ref_fmt_free(h)
    return bm, va, st


def numpy_sgemv(A, x):
    """SgemvCPU (src/tester.cpp:36-45) stated in numpy, sharing nothing with oracle/: for every output i,
    acc = 0; for j ascending: acc = fl32(acc + fl32(x[j] * A[j][i])).  Vectorised over i only."""
    A = np.ascontiguousarray(A, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    acc = np.zeros(A.shape[1], np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        for j in range(A.shape[0]):
            p = np.multiply(x[j], A[j], dtype=np.float32)
            acc = np.add(acc, p, dtype=np.float32)
    return acc


def tester_style(M, N, a_zero, x_zero, seed):
    """Inputs in the style of tester.cpp:103-121,151-167 with a fixed PCG64 seed."""
    rng = np.random.Generator(np.random.PCG64(seed))
    A = rng.uniform(-1.0, 1.0, size=(M, N)).astype(np.float32)
    A[rng.random(size=(M, N)) < a_zero] = 0.0
    x = rng.uniform(-1.0, 1.0, size=M).astype(np.float32)
    x[rng.random(size=M) < x_zero] = 0.0
    return A, x


def edge_case(seed):
    """Empty CSR rows, an all-zero input row, -0.0f (dropped by `!= 0.0f`), denormals, huge/tiny."""
    M, N = 96, 80
    A, x = tester_style(M, N, 0.7, 0.3, seed)
    A[:, 5] = 0.0            # CSR row 5 empty
    A[:, 79] = 0.0           # last CSR row empty (exercises the missing-sentinel convention)
    A[:, 0] = 0.0            # first CSR row empty
    A[17, :] = 0.0           # input 17 unused
    A[3, 10] = -0.0          # must be dropped
    A[4, 10] = np.float32(1e-41)   # denormal: kept
    A[6, 11] = np.float32(3e38)
    A[7, 11] = np.float32(-3e38)   # cancellation of huge terms
    A[8, 12] = np.float32(1e-30)
    x[6] = 1.0
    x[7] = 1.0
    x[4] = 1.0
    return A, x


def main():
    orc = ge.load_oracle()
    specs = {
        "g128_half": tester_style(128, 128, 0.5, 0.5, 1),        # the tester's own regime, small
        "g256x384_10pct": tester_style(256, 384, 0.9, 0.0, 2),   # M != N
        "g1024_1pct": tester_style(1024, 1024, 0.99, 0.0, 3),    # config-1 density
        "g64x2048_dense_rows": tester_style(2048, 64, 0.2, 0.5, 4),  # few long CSR rows (M=2048 inputs)
        "gedge": edge_case(5),
    }
    for name, (A, x) in specs.items():
        M, N = A.shape
        rp, ci, va = ref_csr(A)
        y = numpy_sgemv(A, x)
        y_c = orc.sgemv_dense(A, x)
        assert np.array_equal(y.view(np.uint32), y_c.view(np.uint32)), f"{name}: the two statements of SgemvCPU differ"
        out = dict(M=np.int32(M), N=np.int32(N), x=x, ref_row_ptrs=rp, ref_col_idxs=ci, ref_vals=va,
                   y_dense=y,
                   y_source=np.array("numpy_sgemv (tests/golden/make_golden.py) restating src/tester.cpp:36-45; "
                                     "bit-equal to oracle_sgemv_dense (oracle/spmv_oracle.c) when written"),
                   csr_source=np.array("reference CSRMatrix, src/matrix_csr.cpp:5-23, via oracle/_ref"))
        if M % 32 == 0 and N % 32 == 0:   # the reference's tiled bitmap-CSR of the same matrix
            bi, bm, tv = ref_tcsr(A)
            out.update(tcsr_blk_idx=bi, tcsr_bitmaps=bm, tcsr_vals=tv,
                       tcsr_source=np.array("reference TCSRMatrix, src/tcsr.cpp:5-38, via oracle/_ref"))
            for kind, key in ((0, "wsp"), (1, "awsp"), (2, "awsp_ref")):
                bm, fv, st = ref_fmt(kind, A)
                out[f"{key}_bitmaps"], out[f"{key}_vals"], out[f"{key}_stats"] = bm, fv, st
            _, av, _ = ref_fmt(3, A)       # ASPMatrix is the dense matrix re-tiled: its size and a checksum suffice
            out["asp_checksum"] = np.array([av.size, int(av.view(np.uint32).astype(np.uint64).sum() & 0xFFFFFFFFFFFF),
                                            int((av.view(np.uint32).astype(np.uint64) * (np.arange(av.size, dtype=np.uint64) % 65521 + 1)).sum()
                                                & 0xFFFFFFFFFFFF)], np.int64)
            out["bitmap_formats_source"] = np.array("reference WSPMatrix / AWSPMatrix / AWSPRefMatrix / ASPMatrix "
                                                    "(src/wsp.cpp, awsp.cpp, awsp_ref.cpp, asp.cpp) via oracle/_ref; "
                                                    "stats = {nz_max_m, nz_max_n} / {nz_bk_max_} / warp_nz_offset_[4]")
        if A.size <= 128 * 128:
            out["A"] = A      # keep the dense matrix where it is small (needed for -0.0f)
        np.savez_compressed(HERE / f"{name}.npz", **out)
        print(f"{name}: {M}x{N} nnz={len(va)} -> {(HERE / (name + '.npz')).stat().st_size} bytes")


if __name__ == "__main__":
    main()
