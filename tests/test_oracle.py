"""CPU: pin oracle/spmv_oracle.c against the committed golden vectors and, where this container has
it, against the reference's own CSRMatrix build (oracle/_ref)."""
import ctypes
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
REF_LIB = ROOT / "oracle" / "_ref" / "libref_formats.so"


def test_csr_builder_matches_reference_fixture(oracle, golden):
    """matrix_csr.cpp:5-23 -- bit-exact on indices AND values, reference layout (no sentinel)."""
    rp, ci, va = oracle.csr_from_dense(golden.A)
    assert rp[-1] == len(golden.vals)
    assert np.array_equal(rp[:-1], golden.ref_row_ptrs)
    assert len(golden.ref_row_ptrs) == golden.N            # N entries, not N+1
    assert np.array_equal(ci, golden.col_idx)
    assert np.array_equal(va.view(np.uint32), golden.vals.view(np.uint32))


def test_dense_loop_matches_fixture(oracle, golden):
    y = oracle.sgemv_dense(golden.A, golden.x)
    assert np.array_equal(y.view(np.uint32), golden.y.view(np.uint32))


def test_csr_walk_equals_dense_loop(oracle, golden):
    """the sequential CSR walk is bit-identical to SgemvCPU (zeros add exactly +-0)."""
    y = oracle.spmv(golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    # the only representable difference is the sign of a zero (an empty row gives +0 both ways)
    assert np.array_equal(y, golden.y)
    nz = golden.y != 0
    assert np.array_equal(y[nz].view(np.uint32), golden.y[nz].view(np.uint32))


@pytest.mark.parametrize("threads", [2, 3, 8])
def test_multithreaded_walk_is_bit_identical(oracle, golden, threads):
    y1 = oracle.spmv(golden.row_ptr, golden.col_idx, golden.vals, golden.x, threads=1)
    yt = oracle.spmv(golden.row_ptr, golden.col_idx, golden.vals, golden.x, threads=threads)
    assert np.array_equal(y1.view(np.uint32), yt.view(np.uint32))


def test_f64_walk_bounds_fp32_error(oracle, golden):
    y = oracle.spmv(golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    y64, mag = oracle.spmv_f64(golden.row_ptr, golden.col_idx, golden.vals, golden.x)
    rows_len = np.diff(golden.row_ptr)
    # sequential fp32 sum of n terms: |err| <= n * eps * sum|terms|
    bound = (rows_len + 1) * 2.0 ** -24 * mag + 1e-45
    assert np.all(np.abs(y.astype(np.float64) - y64) <= bound)


def test_edge_fixture_semantics():
    from conftest import load_golden
    g = load_golden("gedge")
    lens = np.diff(g.row_ptr)
    assert lens[0] == 0 and lens[5] == 0 and lens[79] == 0          # empty CSR rows, incl. first/last
    assert g.A[3, 10] == 0 and np.signbit(g.A[3, 10])                # the -0.0f is in the dense input
    r10 = slice(g.row_ptr[10], g.row_ptr[11])
    assert 3 not in g.col_idx[r10] and 4 in g.col_idx[r10]           # -0.0f dropped, denormal kept
    assert g.y[5] == 0 and g.y[79] == 0


def test_compare_counts_like_reference(oracle):
    a = np.array([0.0, 1.0, 2.0, np.nan], np.float32)
    b = np.array([0.0005, 1.002, 2.0, 0.0], np.float32)
    assert oracle.L.oracle_compare(4, a.ctypes.data, b.ctypes.data, ctypes.c_float(1e-3)) == 2


@pytest.mark.skipif(not REF_LIB.exists(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("shape,zero", [((64, 96), 0.5), ((300, 70), 0.95), ((33, 1), 0.0), ((1, 40), 0.3)])
def test_csr_builder_matches_live_reference(oracle, shape, zero):
    """Run the reference's CSRMatrix (compiled from /root/reference by oracle/Makefile) on fresh inputs."""
    lib = ctypes.CDLL(str(REF_LIB))
    lib.ref_csr_build.restype = ctypes.c_void_p
    lib.ref_csr_build.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3
    lib.ref_csr_copy.argtypes = [ctypes.c_void_p] * 4
    lib.ref_csr_free.argtypes = [ctypes.c_void_p]
    rng = np.random.Generator(np.random.PCG64(hash(shape) & 0xFFFF))
    A = rng.uniform(-1, 1, size=shape).astype(np.float32)
    A[rng.random(size=shape) < zero] = 0.0
    A.flat[::7] = -0.0
    M, N = shape
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    h = lib.ref_csr_build(M, N, A.ctypes.data, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    rp = np.empty(a.value, np.int32); ci = np.empty(b.value, np.int32); va = np.empty(c.value, np.float32)
    lib.ref_csr_copy(h, rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
    lib.ref_csr_free(h)
    orp, oci, ova = oracle.csr_from_dense(A)
    assert a.value == N and np.array_equal(orp[:-1], rp) and orp[-1] == c.value
    assert np.array_equal(oci, ci) and np.array_equal(ova.view(np.uint32), va.view(np.uint32))


def test_tcsr_builder_matches_reference_fixture(oracle, golden):
    """tcsr.cpp:5-38 -- blk_idx, bitmaps and values bit for bit (32-aligned fixtures)."""
    if golden.tcsr is None:
        pytest.skip("fixture is not 32-aligned: the reference's TCSR is undefined for it")
    bi, bm, va = oracle.tcsr_from_dense(golden.A)
    rbi, rbm, rva = golden.tcsr
    assert len(rbi) == (golden.M // 32) * (golden.N // 32) + 1
    assert np.array_equal(bi, rbi) and np.array_equal(bm, rbm)
    assert np.array_equal(va.view(np.uint32), rva.view(np.uint32))
    # the same nonzeros as the CSR of the same matrix, in a different order
    assert len(va) == len(golden.vals) and np.array_equal(np.sort(va), np.sort(golden.vals))
