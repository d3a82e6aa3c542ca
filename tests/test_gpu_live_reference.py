"""GPU (-m gpu): the device-side format builders against the REFERENCE ITSELF, live -- oracle/_ref/libref_formats.so is
the reference's own CSRMatrix / TCSRMatrix / WSPMatrix / AWSPMatrix / AWSPRefMatrix compiled unmodified in the build
container (oracle/Makefile); it travels to the GPU box with the other built libraries, so these tests run fresh random
inputs through the reference's classes and through the HIP builders side by side and compare the arrays bit for bit.
(The golden fixtures under tests/golden pin the same classes on a fixed set of matrices; this is the open-ended form.)
Skipped where the library was not built (no /root/reference at build time)."""
import ctypes
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REF_LIB = Path(__file__).resolve().parent.parent / "oracle" / "_ref" / "libref_formats.so"
needs_ref = pytest.mark.skipif(not REF_LIB.exists(), reason="oracle/_ref not built (no /root/reference at build time)")


def _ref():
    lib = ctypes.CDLL(str(REF_LIB))
    ip = ctypes.POINTER(ctypes.c_int)
    lib.ref_csr_build.restype = ctypes.c_void_p
    lib.ref_csr_build.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ip, ip, ip]
    lib.ref_csr_copy.argtypes = [ctypes.c_void_p] * 4
    lib.ref_csr_free.argtypes = [ctypes.c_void_p]
    lib.ref_tcsr_build.restype = ctypes.c_void_p
    lib.ref_tcsr_build.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ip, ip, ip]
    lib.ref_tcsr_copy.argtypes = [ctypes.c_void_p] * 4
    lib.ref_tcsr_free.argtypes = [ctypes.c_void_p]
    lib.ref_fmt_build.restype = ctypes.c_void_p
    lib.ref_fmt_build.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ip, ip, ip]
    lib.ref_fmt_copy.argtypes = [ctypes.c_void_p] * 3
    lib.ref_fmt_free.argtypes = [ctypes.c_void_p]
    return lib


def _random_dense(M, N, zero, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    A = rng.uniform(-1, 1, size=(M, N)).astype(np.float32)
    A[rng.random(size=(M, N)) < zero] = 0.0
    A.flat[::13] = -0.0                 # `!= 0.0f` drops it
    if A.size > 40:
        A.flat[17] = np.float32(1e-45)  # a denormal is kept
        A.flat[29] = np.nan             # NaN != 0: kept
    return np.ascontiguousarray(A)


@needs_ref
@pytest.mark.parametrize("M,N,zero", [(64, 96, 0.5), (300, 70, 0.95), (33, 1, 0.0), (1, 40, 0.3), (1000, 257, 0.7),
                                      (4096, 4096, 0.5), (257, 4099, 0.99)])
def test_device_csr_builder_against_the_live_reference(pkg, gpu, M, N, zero):
    """spmv_csr_from_dense_device against CSRMatrix (matrix_csr.cpp:5-23) on a fresh matrix."""
    import torch
    lib = _ref()
    A = _random_dense(M, N, zero, M * 1009 + N)
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    h = lib.ref_csr_build(M, N, A.ctypes.data, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    rp = np.empty(a.value, np.int32); ci = np.empty(b.value, np.int32); va = np.empty(c.value, np.float32)
    lib.ref_csr_copy(h, rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
    lib.ref_csr_free(h)
    m = pkg.capi.CsrMatrix.from_dense_device(torch.from_numpy(A).to(gpu))
    drp, dci, dva = m.download()
    assert (m.rows, m.cols, m.nnz) == (N, M, c.value)
    assert a.value == N and np.array_equal(drp[:-1], rp) and drp[-1] == c.value      # the reference keeps N row starts
    assert np.array_equal(dci, ci)
    assert np.array_equal(dva.view(np.uint32), va.view(np.uint32))
    m.close()


@needs_ref
@pytest.mark.parametrize("M,N,zero", [(32, 32, 0.0), (64, 96, 0.5), (2048, 96, 0.9), (96, 4096, 0.97), (1056, 160, 0.3),
                                      (4096, 4096, 0.5)])
def test_device_tcsr_builder_against_the_live_reference(pkg, gpu, M, N, zero):
    """spmv_tcsr_from_dense_device against TCSRMatrix (tcsr.cpp:5-38)."""
    import torch
    lib = _ref()
    A = _random_dense(M, N, zero, M * 31 + N)
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    h = lib.ref_tcsr_build(M, N, A.ctypes.data, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    bi = np.empty(a.value, np.int32); bm = np.empty(b.value, np.uint32); va = np.empty(c.value, np.float32)
    lib.ref_tcsr_copy(h, bi.ctypes.data, bm.ctypes.data, va.ctypes.data)
    lib.ref_tcsr_free(h)
    T = pkg.capi.TcsrMatrix.from_dense_device(torch.from_numpy(A).to(gpu))
    dbi, dbm, dva = T.download()
    assert np.array_equal(dbi, bi) and np.array_equal(dbm, bm)
    assert np.array_equal(dva.view(np.uint32), va.view(np.uint32))
    T.close()


@needs_ref
@pytest.mark.parametrize("fmt,kind", [("wsp", 0), ("awsp", 1), ("awsp_ref", 2)])
@pytest.mark.parametrize("M,N,zero", [(32, 32, 0.0), (2048, 96, 0.9), (96, 4096, 0.97), (1056, 160, 0.3), (4096, 4096, 0.5)])
def test_device_bitmap_builders_against_the_live_reference(pkg, gpu, fmt, kind, M, N, zero):
    """spmv_bitmap_from_dense_device against WSPMatrix / AWSPMatrix / AWSPRefMatrix (wsp.cpp:3-40, awsp.cpp:3-49,
    awsp_ref.cpp:4-58): bitmaps, padded values and the statistics the classes expose."""
    import torch
    lib = _ref()
    A = _random_dense(M, N, zero, M * 7 + N + kind)
    nb, nv = ctypes.c_int(), ctypes.c_int()
    stats = (ctypes.c_int * 4)()
    h = lib.ref_fmt_build(kind, M, N, A.ctypes.data, ctypes.byref(nb), ctypes.byref(nv), stats)
    bm = np.empty(nb.value, np.uint32); va = np.empty(nv.value, np.float32)
    lib.ref_fmt_copy(h, bm.ctypes.data, va.ctypes.data)
    lib.ref_fmt_free(h)
    B = pkg.capi.BitmapMatrix.from_dense_device(fmt, torch.from_numpy(A).to(gpu))
    assert B.stats == [int(v) for v in stats]
    dbm, dva = B.download()
    assert np.array_equal(dbm, bm) and len(dva) == len(va)
    assert np.array_equal(dva.view(np.uint32), va.view(np.uint32))
    B.close()


@needs_ref
@pytest.mark.parametrize("M,N,zero", [(32, 32, 0.0), (2048, 96, 0.9), (96, 4096, 0.97), (1056, 160, 0.3), (4096, 4096, 0.5)])
def test_device_asp_layout_against_the_live_reference(pkg, oracle, gpu, M, N, zero):
    """spmv_asp_retile against ASPMatrix (asp.cpp:3-14), bit for bit; then spmv_asp_gemv_ws from that layout -- the
    x == 0 skip of asp_kernel_v* (asp.cu:20-26) with the tester's 50 %-zero x -- against the CSR oracle of the same
    matrix."""
    import torch
    from _util import assert_close_to_oracle
    lib = _ref()
    capi = pkg.capi
    A = _random_dense(M, N, zero, M * 5 + N)
    A[np.isnan(A)] = 0.5                                  # (keep the product comparable: no NaN in the values here)
    nb, nv = ctypes.c_int(), ctypes.c_int()
    stats = (ctypes.c_int * 4)()
    h = lib.ref_fmt_build(3, M, N, A.ctypes.data, ctypes.byref(nb), ctypes.byref(nv), stats)
    ref = np.empty(nv.value, np.float32)
    lib.ref_fmt_copy(h, None, ref.ctypes.data)
    lib.ref_fmt_free(h)
    assert nv.value == M * N
    dA = torch.from_numpy(A).to(gpu)
    d_asp = torch.empty(M * N, dtype=torch.float32, device=gpu)
    capi.asp_retile(dA, d_asp)
    torch.cuda.synchronize()
    assert np.array_equal(d_asp.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # the multiply from the layout
    rng = np.random.default_rng(M + N)
    x = rng.uniform(-1, 1, size=M).astype(np.float32)
    x[rng.random(M) < 0.5] = 0.0
    dx = torch.from_numpy(x).to(gpu)
    dy = torch.full((N,), float("nan"), device=gpu)
    ws = torch.empty(capi.dense_gemv_workspace_bytes(N, 3), dtype=torch.uint8, device=gpu)
    capi.asp_gemv(M, N, d_asp, dx, dy, ws)
    torch.cuda.synchronize()
    rp, ci, va = oracle.csr_from_dense(A)
    y64, mag = oracle.spmv_f64(rp, ci, va, x)
    assert_close_to_oracle(dy.cpu().numpy(), y64, mag, f"asp layout {M}x{N}")
    with pytest.raises(capi.SpmvError):
        capi.asp_retile(torch.zeros(33, 32, device=gpu), torch.zeros(33 * 32, device=gpu))
